"""sha256 of the fields after two steps on a masked grid: run it with two builds (FS3D_LIB_PATH) to see whether a kernel change
kept the bits (packed math did, other contractions do not).  python tools/hash_fields.py   (GPU box)"""
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
from cmc_fluid_solver_amd import capi, grids
g = grids.box_with_obstacle(136, 150, 64, h=0.01)
s = capi.Solver(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), np.float32)
for i in range(2):
    s.UpdateBoundaries(); s.TimeStep(0.1, 2, 2, True)
h = hashlib.sha256()
for a in s.download_layer(capi.LAYER_CUR): h.update(np.ascontiguousarray(a).tobytes())
print('HASH', h.hexdigest()[:16], s.last_sweep_kernels())
