"""Does the power-of-two plane stride of the 256^3 box cost the X sweep something (rows of one line 256 KiB apart)?
The same kernels on boxes whose plane is not a power of two.  python tools/stride_check.py   (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids

DIMS = ((256, 256, 256), (256, 256, 256), (256, 248, 256), (248, 256, 256), (256, 252, 256), (252, 256, 256))
if os.environ.get('STRIDE_SMALL_PLANES'):
    DIMS = ((256, 256, 256), (256, 256, 256), (256, 128, 256), (256, 256, 128), (256, 128, 128), (256, 64, 256), (256, 512, 128))
for dims in DIMS:
    g = grids.box(*dims, h=1.0 / 255)
    s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
    for i in range(2):
        s.UpdateBoundaries(); s.TimeStep(0.1, 4, 2, False)
    out = {}
    for d in (0, 1, 2):
        s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
        s.enable_timing(True)
        for _ in range(10):
            s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
        ms, cnt = s.last_step_timing()
        k = {2: 0, 1: 1, 0: 2}[d]
        out["XYZ"[d]] = ms[k] / max(cnt[k], 1)
        s.enable_timing(False)
    cells = dims[0] * dims[1] * dims[2] / 1e6
    print("%s: ns per cell  X %.4f  Y %.4f  Z %.4f   (ms X %.4f Y %.4f Z %.4f) ran %s" % (dims, out["X"] / cells * 1e3 / 1e3 * 1e3, out["Y"] / cells, out["Z"] / cells,
          out["X"], out["Y"], out["Z"], s.last_sweep_kernels()), flush=True)
    s.close()
