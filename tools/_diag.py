import sys, os; sys.path.insert(0,'/root/repo')
import numpy as np
from cmc_fluid_solver_amd import capi, grids
def rel(a,b):
    a=np.asarray(a,np.float64); b=np.asarray(b,np.float64); return float(np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300))
params = capi.fluid_params(np.float32, 200.0, 0.72, 1.4)
for n, nr, kern in ((128, 4, capi.SWEEP_AUTO), (128, 4, capi.SWEEP_EXACT), (256, 8, capi.SWEEP_AUTO), (64, 2, capi.SWEEP_AUTO)):
    g = grids.box(n, h=1.0/(n-1))
    s = capi.Solver(g, params, np.float32); s.set_option(capi.OPT_SWEEP_KERNEL, kern)
    for i in range(2):
        s.UpdateBoundaries(); s.TimeStep(0.1, 4, 2, True)
    A = s.download_layer(capi.LAYER_CUR); ran1 = s.last_sweep_kernels(); s.close()
    grp = capi.LocalGroup(g, params, nr, np.float32)
    def work(r, sv):
        sv.set_option(capi.OPT_SWEEP_KERNEL, kern)
        for i in range(2):
            sv.UpdateBoundaries(); sv.TimeStep(0.1, 4, 2, True)
        return sv.download_layer(capi.LAYER_CUR), sv.last_sweep_kernels()
    res = grp.run(work)
    full = [np.concatenate([res[r][0][v] for r in range(nr)], axis=0) for v in range(4)]
    print(n, nr, kern, "single ran", ran1, "slab ran", res[0][1], "rel", ["%.1e" % rel(a, b) for a, b in zip(full, A)])
    d = np.abs(full[3].astype(np.float64) - A[3])
    print("   T max abs diff %.2e at" % d.max(), np.unravel_index(d.argmax(), d.shape), "per-plane max:", ["%.0e" % d[i].max() for i in range(0, n, max(1, n // 16))])
    grp.close()
