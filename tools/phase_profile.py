#!/usr/bin/env python3
"""Per-phase timing of the pipelined sweep kernel from in-kernel s_memtime stamps
(fs3d_profile_sweep).  Usage: python tools/phase_profile.py [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmc_fluid_solver_amd import capi, grids

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = grids.box(n, h=1.0 / (n - 1))
s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
for i in range(2):
    s.time_step_async(0.1, 4, 2)
s.synchronize()
names = ["P rows", "wait turn", "F forward", "wait all F + B turn", "B backward", "drain", "O stores"]
for d, dn in ((0, "X"), (1, "Y"), (2, "Z")):
    st = s.profile_sweep(d, 0.1).astype(np.int64)
    st = st[:, [w for w in range(8) if st[:, w, 7].any()], :]          # [blocks, waves, 8]
    dur = np.diff(st, axis=2)                                # [blocks, waves, 7]
    life = st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)
    t0 = st[:, :, 0].min()
    print("dir %s: %d blocks, kernel span %.1f us (100 MHz ticks? raw %d), mean block life %d ticks" % (
        dn, len(st), 0, st[:, :, 7].max() - t0, life.mean()))
    for k, nm in enumerate(names):
        print("   %-22s mean %9.0f  per wave: %s" % (nm, dur[:, :, k].mean(),
              " ".join("%7.0f" % v for v in dur[:, :, k].mean(axis=0))))
