#!/usr/bin/env python3
"""Single sweeps with and without the fused merge, host-timed.  Usage: python tools/merge_cost.py [size ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmc_fluid_solver_amd import capi, grids
for n in [int(a) for a in sys.argv[1:]] or [256, 512]:
    g = grids.box(n, h=1.0 / (n - 1))
    s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
    s.UpdateBoundaries(); s.TimeStep(0.1, 1, 1, False)
    for merge in (False, True):
        for d in (0, 1, 2):
            for i in range(2):
                s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=merge)
            t0 = time.perf_counter()
            for i in range(5):
                s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=merge)
            print("n=%d dir %d merge=%d: %.3f ms" % (n, d, merge, (time.perf_counter() - t0) / 5 * 1e3))
    s.close()
