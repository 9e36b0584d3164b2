python -m pytest tests/test_gpu_part.py tests/test_gpu_slabs.py tests/test_gpu_failures.py -q -m gpu 2>&1 | tail -4
python - <<'PY'
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import numpy as np, time
import part_check as pc
from cmc_fluid_solver_amd import capi, grids
pc.timing(256, capi.SWEEP_AUTO)
# a 256^3 box as 8 slabs on one card: time per step of one slab thread (all 8 share the card: ~sum of the slabs' work)
g = grids.box(256, h=1.0/255)
params = capi.fluid_params(np.float32, 200.0, 0.72, 1.4)
for xs in (capi.XSOLVE_REDUCED, capi.XSOLVE_PIPELINED):
    grp = capi.LocalGroup(g, params, 8, np.float32)
    def work(r, sv):
        sv.set_option(capi.OPT_XSOLVE, xs)
        for i in range(2):
            sv.UpdateBoundaries(); sv.TimeStep(0.1, 4, 2, False)
        t0 = time.perf_counter()
        for i in range(5):
            sv.UpdateBoundaries(); sv.TimeStep(0.1, 4, 2, False)
        return (time.perf_counter() - t0) / 5, sv.last_sweep_kernels()
    res = grp.run(work)
    print("8 slabs of the 256^3 box on ONE card, xsolve %d: %.2f ms per step (all slabs share the card), ran %s" % (xs, max(r[0] for r in res) * 1e3, res[0][1]))
    grp.close()
PY
