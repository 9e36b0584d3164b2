"""A few merged sweeps per direction at full size, for rocprofv3 runs (kernel trace / PMC).
python tools/sweep_driver.py [size] [kernel id] [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
g = grids.box(n, h=1.0 / (n - 1))
s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
for i in range(2):
    s.UpdateBoundaries(); s.TimeStep(0.1, 4, 2, False)
for d in (0, 1, 2):
    for r in range(reps):
        s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
print("ran", s.last_sweep_kernels())
s.close()
