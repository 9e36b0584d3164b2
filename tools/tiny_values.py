#!/usr/bin/env python3
"""How many field values of a run started from rest lie in (0, 2^-100) -- the operands that send a bundle of the fp32
pipe kernel to its full-division pass.  Usage: python tools/tiny_values.py size steps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cmc_fluid_solver_amd import capi, grids
n, steps = int(sys.argv[1]), int(sys.argv[2])
g = grids.box(n, h=1.0 / (n - 1))
s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
dt = 0.1 if n <= 256 else 0.1
for i in range(steps):
    s.UpdateBoundaries(); s.TimeStep(dt, 4, 2, False)
for name, layer in (("cur", capi.LAYER_CUR), ("temp", capi.LAYER_TEMP)):
    for v, a in enumerate(s.download_layer(layer)):
        x = np.abs(a)
        tiny = (x > 0) & (x < 2.0 ** -100)
        print("%s field %d: %9d tiny of %d (%.2f %%), exact zeros %d, max %.3g; lines (along x) with a tiny value: %d of %d" % (
            name, v, tiny.sum(), a.size, 100.0 * tiny.mean(), (x == 0).sum(), x.max(), tiny.any(axis=0).sum(), n * n))
