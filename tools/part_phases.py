"""In-kernel phase stamps of the partition sweep kernel (s_memtime ticks): mean time a wave spends in each phase,
mean life of a workgroup, and the launch's span.  python tools/part_phases.py [size]   (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = grids.box(n, h=1.0 / (n - 1))
s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
for i in range(2):
    s.UpdateBoundaries(); s.TimeStep(0.1, 4, 2, False)
names = ["codes", "P rows", "E eliminate", "barrier 1", "R interface", "S solve", "O store"]
for d in (0, 1):
    for rep in range(2):
        st = s.profile_sweep(d, 0.1, max_blocks=8192).astype(np.int64)      # [blocks, 8 waves, 8 stamps]
    st = st[(st[:, 0, 0] > 0)]
    ph = np.diff(st, axis=2)                                               # [blocks, waves, 7]
    life = st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)
    print("dir %d: %d workgroups, ran %s; mean wave ticks per phase: %s" % (
        d, len(st), s.last_sweep_kernels()["XYZ"[d]], ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, ph.mean(axis=(0, 1))))))
    print("        workgroup life mean %.0f  min %.0f  max %.0f ticks; per-wave-id P: %s  E: %s" % (
        life.mean(), life.min(), life.max(), np.round(ph[:, :, 1].mean(axis=0)).astype(int).tolist(),
        np.round(ph[:, :, 2].mean(axis=0)).astype(int).tolist()))
s.close()
