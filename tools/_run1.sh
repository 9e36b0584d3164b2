python - <<'PY'
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import os
import part_check as pc
from cmc_fluid_solver_amd import capi, grids
os.environ["FS3D_PART_VARIANT"]="10"
pc.check(grids.box_with_obstacle(256, 16, 48, h=0.004), "obstacle 256x16x48 v10", dirs=(0,))
pc.check(grids.box_with_obstacle(12, 256, 100, h=0.004), "obstacle 12x256x100 v10", dirs=(1,))
PY
for v in 0 10 11 0 10 11; do echo "== variant $v"; FS3D_PART_VARIANT=$v python - <<'PY'
import sys; sys.path.insert(0,'tools'); sys.argv=['x']
import part_check as pc
from cmc_fluid_solver_amd import capi, grids
pc.timing(256, capi.SWEEP_AUTO)
PY
done 2>&1 | grep "timing\|variant"
