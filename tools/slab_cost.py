"""What one rank of an N-GPU x-slab run computes per step, measured on ONE card: the middle slab of the 256^3 box (256/N planes)
as a context of its own, no communicator (a world of one: exchanges are no-ops, the cross-slab X sweep runs its kernels --
interface elimination, reduced system, slab solve -- on the slab alone; the numbers are times, the fields are not a solution).
python tools/slab_cost.py [size]   (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids
from cmc_fluid_solver_amd.slab import slab_range

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = grids.box(n, h=1.0 / (n - 1))
params = capi.fluid_params(np.float32, 200.0, 0.72, 1.4)
base = None
for nr in (1, 2, 4, 8):
    xa, xb = slab_range(n, nr // 2, nr)
    s = capi.Solver(g, params, np.float32, x_range=(xa, xb))
    if nr > 1:
        s.set_option(capi.OPT_XSOLVE, capi.XSOLVE_REDUCED)
    for i in range(3):
        s.time_step_async(0.1, 4, 2)
    s.synchronize()
    s.enable_timing(True)
    t0 = time.perf_counter()
    K = 10
    for i in range(K):
        s.time_step_async(0.1, 4, 2)
    s.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    ms, cnt = s.last_step_timing()
    s.enable_timing(False)
    per = [m / max(c, 1) for m, c in zip(ms, cnt)]
    base = base or wall
    print("N=%d: slab of %d planes: %.3f ms/step (x%.2f of the whole grid's; ideal x%.3f); per launch Z %.4f Y %.4f X %.4f other %.4f ms; launches per step %s; ran %s"
          % (nr, xb - xa, wall, wall / base, 1.0 / nr, per[0], per[1], per[2], per[3], [c // K if c > K else c for c in cnt], s.last_sweep_kernels()), flush=True)
    s.close()
