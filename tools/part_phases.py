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
    st = st[:, :7]                                                         # wave 7 stamps the 100 MHz real-time clock (timeline below)
    ph = np.diff(st, axis=2)                                               # [blocks, waves, 7]
    life = st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)
    print("dir %d: %d workgroups, ran %s; mean wave ticks per phase: %s" % (
        d, len(st), s.last_sweep_kernels()["XYZ"[d]], ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, ph.mean(axis=(0, 1))))))
    print("        workgroup life mean %.0f  min %.0f  max %.0f ticks; per-wave-id P: %s  E: %s" % (
        life.mean(), life.min(), life.max(), np.round(ph[:, :, 1].mean(axis=0)).astype(int).tolist(),
        np.round(ph[:, :, 2].mean(axis=0)).astype(int).tolist()))
s.close()


def timeline(d, bins=24):
    """Launch timeline: number of resident workgroups that load (codes + P), compute (E .. S) or store (O) per time bin."""
    s2 = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
    for i in range(2):
        s2.UpdateBoundaries(); s2.TimeStep(0.1, 4, 2, False)
    for rep in range(2):
        st = s2.profile_sweep(d, 0.1, max_blocks=8192).astype(np.int64)
    s2.close()
    ids = np.nonzero(st[:, 0, 0] > 0)[0]
    st = st[ids]
    if os.environ.get("FS3D_TIMELINE_DUMP"):
        np.save(os.environ["FS3D_TIMELINE_DUMP"] + "_dir%d.npy" % d, st)
    rel = st[:, 7, :] - st[:, 7, 0].min()                                   # wave 7: s_memrealtime, 10 ns units, chip-wide
    a = rel[:, 0]; p_end = rel[:, 2]; o_beg = rel[:, 6]; e = rel[:, 7]
    span = e.max()
    print("dir %d timeline: span %d x 10 ns, %d workgroups; per bin: resident / loading / computing / storing" % (d, span, len(st)))
    for b in range(bins):
        t = (b + 0.5) * span / bins
        res = (a <= t) & (t < e)
        ld = res & (t < p_end); stg = res & (t >= o_beg); cp = res & ~ld & ~stg
        print("   %5.1f%%  %4d  %4d  %4d  %4d" % (100 * (b + 0.5) / bins, res.sum(), ld.sum(), cp.sum(), stg.sum()))


if os.environ.get("FS3D_TIMELINE"):
    for d in (0, 1):
        timeline(d)
