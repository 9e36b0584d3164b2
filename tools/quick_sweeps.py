"""Quick per-direction sweep timing + PIPE-vs-LINE agreement at N^3 (measurement aid for kernel experiments).
   python tools/quick_sweeps.py [N] ; FS3D_LIB_PATH selects an experimental library build."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = np.float32
g = grids.box(n, n, n, h=1.0 / n)
params = capi.fluid_params(dtype, 200.0, 0.72, 1.4)
base = [np.ascontiguousarray(a, dtype) for a in (g.vx, g.vy, g.vz, g.T)]
cur, tmp = grids.perturb(base, seed=1), grids.perturb(base, seed=2)
res = {}
for kern in (capi.SWEEP_LINE, capi.SWEEP_PIPE):
    s = capi.Solver(g, params, dtype)
    s.set_option(capi.OPT_SWEEP_KERNEL, kern)
    s.upload_layer(capi.LAYER_CUR, cur); s.upload_layer(capi.LAYER_TEMP, tmp)
    for d in (0, 1, 2):
        s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT)
        if kern == capi.SWEEP_PIPE:
            reps = 10
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT)
            s.synchronize()
            print("PIPE dir %d: %.3f ms per sweep (host-timed, synchronous calls)" % (d, (time.perf_counter() - t0) / reps * 1e3))
        res[(kern, d)] = s.download_layer(capi.LAYER_NEXT)
    s.close()
ok = True
for d in (0, 1, 2):
    for v in range(4):
        if not np.array_equal(res[(capi.SWEEP_LINE, d)][v], res[(capi.SWEEP_PIPE, d)][v]):
            ok = False
            print("MISMATCH dir %d field %d" % (d, v))
print("PIPE == LINE:", ok)
sys.exit(0 if ok else 1)
