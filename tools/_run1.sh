python tools/part_phases.py 256 > gpurun_out/phases_v0.log 2>&1
for o in 0 1; do for v in 0 1 2 4 5 6; do echo "== order $o variant $v"; FS3D_PART_ORDER=$o FS3D_PART_VARIANT=$v python - <<'PY'
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import part_check as pc
from cmc_fluid_solver_amd import capi, grids
pc.check(grids.box_with_obstacle(256, 16, 48, h=0.004), "obstacle 256x16x48", dirs=(0,))
pc.timing(256, capi.SWEEP_AUTO)
PY
done; done > gpurun_out/variants.log 2>&1
cat gpurun_out/phases_v0.log; grep -v obstacle gpurun_out/variants.log
