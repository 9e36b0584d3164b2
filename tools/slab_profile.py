"""Kernel timings of the slab (multi-GPU) code path on ONE card: N slab contexts joined by the in-process
transport.  With FS3D_XBLOCKS=1 the cross-slab X sweep is strictly serial, so every k_xsweep_* launch runs
alone on the card and its rocprofv3 duration is what a rank of an N-GPU job would see for that kernel.
  rocprofv3 --kernel-trace --stats -d gpurun_out/slab -- python3 tools/slab_profile.py 256 8
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = grids.box(n, n, n, h=1.0 / n)
grp = capi.LocalGroup(g, capi.fluid_params(np.float32, 200.0, 0.72, 1.4), nranks, np.float32)


def work(rank, s):
    s.UpdateBoundaries()
    s.TimeStep(0.1, 4, 2, False)
    t0 = time.perf_counter()
    for _ in range(steps):
        s.TimeStep(0.1, 4, 2, False)
    return (time.perf_counter() - t0) / steps


print("ms/step per slab thread (all slabs share the card):", ["%.2f" % (t * 1e3) for t in grp.run(work)])
grp.close()
