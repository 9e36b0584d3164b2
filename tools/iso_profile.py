import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from cmc_fluid_solver_amd import capi, grids
names = ["P rows", "wait turn", "F forward", "wait all F + B turn", "B backward", "drain", "O stores"]
for nx in [int(a) for a in sys.argv[1:]] or (4, 64, 256):
    g = grids.box(nx, 256, 256, h=1.0 / 255)
    s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
    for i in range(2):
        s.time_step_async(0.1, 1, 1)
    s.synchronize()
    for d, dn in ((1, "Y"), (2, "Z")):
        st = s.profile_sweep(d, 0.1).astype(np.int64)
        st = st[:, [w for w in range(8) if st[:, w, 7].any()], :]
        dur = np.diff(st, axis=2)
        life = st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)
        print("nx=%d dir %s: %d bundles, mean life %d | %s" % (nx, dn, len(st), life.mean(), "  ".join("%s %d" % (nm.split()[0], dur[:, :, k].mean()) for k, nm in enumerate(names) if k in (0, 2, 4, 6))))
    s.close()
