cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/tools/sweep_driver.py 256 0 3 > $R/gpurun_out/pmc_$tag.log 2>&1 || echo "group $grp failed"
done
cd $R
for d in gpurun_out/pmc_*/; do echo "== $d"; python tools/pmc_summary.py $d; done > gpurun_out/pmc_summary.txt 2>&1
rm -rf gpurun_out/pmc_*/
cat gpurun_out/pmc_summary.txt
