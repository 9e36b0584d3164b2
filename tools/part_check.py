"""Partition kernel (FS3D_SWEEP_PART) against the CPU oracle on small grids (rel-L2 per field) and its sweep
times at full size next to the exact pipe kernel.  Run on the GPU box: python tools/part_check.py [size]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmc_fluid_solver_amd import capi, grids
from oracle import oracle as O

DT = 0.1


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-300))


def check(g, name, dirs=(0, 1, 2), kernel=capi.SWEEP_PART):
    dtype = np.float32
    params = capi.fluid_params(dtype, 200.0, 0.72, 1.4)
    s = capi.Solver(g, params, dtype); s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
    o = O.Oracle(g, params, dtype)
    base = [np.ascontiguousarray(a, dtype) for a in (g.vx, g.vy, g.vz, g.T)]
    cur = grids.perturb(base, seed=1); tmp = grids.perturb(base, seed=2)
    for d in dirs:
        s.upload_layer(capi.LAYER_CUR, cur); s.upload_layer(capi.LAYER_TEMP, tmp)
        s.upload_layer(capi.LAYER_NEXT, [np.zeros_like(c) for c in cur])
        for v in range(4):
            o.set_field(O.L_CUR, v, cur[v]); o.set_field(O.L_TEMP, v, tmp[v]); o.set_field(O.L_NEXT, v, np.zeros_like(cur[v]))
        try:
            s.sweep(d, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
        except capi.Fs3dError as e:
            print(name, "dir", d, "unsupported:", e); continue
        o.sweep(d, DT, O.L_CUR, O.L_TEMP, O.L_NEXT); o.merge(O.L_NEXT, O.L_TEMP)
        a = s.download_layer(capi.LAYER_NEXT); b = o.get_layer_fields(O.L_NEXT)
        at = s.download_layer(capi.LAYER_TEMP); bt = o.get_layer_fields(O.L_TEMP)
        print("%-22s dir %d ran %s  next rel-L2 %s  temp rel-L2 %s  max|d| %.2e" % (
            name, d, s.last_sweep_kernels()["XYZ"[d]], ["%.1e" % rel(x, y) for x, y in zip(a, b)],
            ["%.1e" % rel(x, y) for x, y in zip(at, bt)], max(float(np.abs(x - y).max()) for x, y in zip(a, b))))
    s.close(); o.close()


def steps(g, name, nsteps, kernel):
    dtype = np.float32
    params = capi.fluid_params(dtype, 200.0, 0.72, 1.4)
    s = capi.Solver(g, params, dtype); s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
    o = O.Oracle(g, params, dtype)
    for i in range(nsteps):
        s.UpdateBoundaries(); o.update_boundaries()
        e = s.TimeStep(DT, 4, 2, True); rc, eo = o.time_step(DT, 4, 2, True)
    a = s.download_layer(capi.LAYER_CUR); b = o.get_layer_fields(O.L_CUR)
    print("%-22s %d steps ran %s: rel-L2 u,v,w,T %s  err %.6e vs %.6e" % (name, nsteps, s.last_sweep_kernels(),
          ["%.1e" % rel(x, y) for x, y in zip(a, b)], e, eo))
    s.close(); o.close()


def timing(n, kernel, reps=6):
    dtype = np.float32
    g = grids.box(n, h=1.0 / (n - 1))
    params = capi.fluid_params(dtype, 200.0, 0.72, 1.4)
    s = capi.Solver(g, params, dtype); s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
    for i in range(2):
        s.UpdateBoundaries(); s.TimeStep(DT, 4, 2, False)
    out = {}
    for d in (0, 1, 2):
        try:
            s.sweep(d, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
        except capi.Fs3dError:
            out["XYZ"[d]] = None; continue
        s.enable_timing(True)
        for _ in range(reps):
            s.sweep(d, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
        ms, cnt = s.last_step_timing()
        k = {2: 0, 1: 1, 0: 2}[d]
        out["XYZ"[d]] = round(ms[k] / max(cnt[k], 1), 4)
        s.enable_timing(False)
    import time
    s.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        s.time_step_async(DT, 4, 2)
    s.synchronize(); t = (time.perf_counter() - t0) / 10
    print("timing %d^3 kernel %d: ms per merged sweep %s ; step %.3f ms = %.0f Mcells/s ; ran %s" % (n, kernel, out, t * 1e3, n ** 3 / t / 1e6, s.last_sweep_kernels()))
    s.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    check(grids.box(20, 24, 28, h=0.04), "box 20x24x28")
    check(grids.box_with_obstacle(28, 24, 32, h=0.03), "obstacle 28x24x32")
    check(grids.box_with_obstacle(70, 40, 33, h=0.02), "obstacle 70x40x33")
    check(grids.box(130, 100, 64, h=0.01), "box 130x100x64")
    check(grids.box_with_obstacle(256, 16, 48, h=0.004), "obstacle 256x16x48", dirs=(0,))
    check(grids.box_with_obstacle(12, 256, 40, h=0.004), "obstacle 12x256x40", dirs=(1,))
    steps(grids.box_with_obstacle(28, 24, 32, h=0.03), "obstacle 28x24x32", 3, capi.SWEEP_AUTO)
    steps(grids.box(64, h=1.0 / 63), "box 64^3", 20, capi.SWEEP_AUTO)
    timing(n, capi.SWEEP_EXACT)
    timing(n, capi.SWEEP_AUTO)
