"""Partition kernels against the sequential fp32 recurrence as the systems get stiff (diffusion number D = nu dt / h^2, i.e. weak
diagonal dominance b / (|a| + |c|) - 1 = 1.5 / D): deviation of both from the fp64 solution on the Shape3D sphere (h = 1 mm) for a
range of dt.  python tools/stiffness_check.py   (GPU box)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cmc_fluid_solver_amd import capi, shape3d
from test_shape3d import icosphere


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def vrel(A, B):
    num = sum(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2 for a, b in zip(A[:3], B[:3]))
    den = sum(np.linalg.norm(np.asarray(b, np.float64)) ** 2 for b in B[:3])
    return float(np.sqrt(num / max(den, 1e-300)))


f32 = lambda x: float(np.float32(x))
v, f = icosphere(11.0, (40.0, 42.0, 45.0), subdiv=1)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "s.txt")
    shape3d.write_mesh(p, [(v, f)])
    nodes, _ = shape3d.load_shape3d(p, f32(0.001), f32(0.001), f32(0.001), align=True)
for dt in (0.0005, 0.002, 0.005, 0.01, 0.02, 0.05, 0.2):
    res = {}
    for tag, dtype, kernel in (("e32", np.float32, capi.SWEEP_EXACT), ("e64", np.float64, capi.SWEEP_EXACT), ("part", np.float32, capi.SWEEP_PART)):
        s = capi.Solver(nodes, capi.fluid_params(dtype, f32(200.0), f32(0.72), f32(1.4)), dtype)
        s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
        for i in range(5):
            s.UpdateBoundaries(); s.TimeStep(dtype(dt), 2, 2, False)
        res[tag] = s.download_layer(capi.LAYER_CUR)
        s.close()
    D = 1.0 / (200.0 * 0.72) * dt / 1e-6
    print("dt %-7g D_T %7.1f: vs fp64 -- velocity: partition %.2e, sequential %.2e; T: partition %.2e, sequential %.2e; |v|max %.2e" % (
        dt, D, vrel(res["part"], res["e64"]), vrel(res["e32"], res["e64"]), rel(res["part"][3], res["e64"][3]), rel(res["e32"][3], res["e64"][3]),
        max(float(np.abs(a).max()) for a in res["e64"][:3])), flush=True)
