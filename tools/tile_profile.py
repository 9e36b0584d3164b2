#!/usr/bin/env python3
"""Phase durations of the pipelined sweep kernel split by lane tile (tiles that hold wall lines take the masked paths).
Usage: python tools/tile_profile.py [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmc_fluid_solver_amd import capi, grids

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = grids.box(n, h=1.0 / (n - 1))
s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
s.time_step_async(0.1, 1, 1); s.synchronize()
names = ["P", "wait", "F", "w", "B", "d", "O"]
for d, dn in ((0, "X"), (1, "Y"), (2, "Z")):
    st = s.profile_sweep(d, 0.1).astype(np.int64)
    nb = len(st); b = np.arange(nb)
    q, x, slot = nb >> 3, b & 7, b >> 3
    lb = x * q + slot                                   # kernel's XCD-aware id (nb % 8 == 0)
    n_o = n
    tile = lb // n_o
    dur = np.diff(st, axis=2)
    life = st[:, :, 7].max(axis=1) - st[:, :, 0].min(axis=1)
    for t in range(nb // n_o):
        m = tile == t
        print("dir %s tile %d: life %6d | %s" % (dn, t, life[m].mean(), "  ".join("%s %6d" % (nm, dur[m][:, :, k].mean()) for k, nm in enumerate(names) if k in (0, 2, 4, 6))))
