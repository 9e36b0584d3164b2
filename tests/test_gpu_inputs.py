"""The hot path on the geometries of the three input formats, against the CPU oracle ("inputs drop in
unchanged"): the multi-frame Shape2D heart (moving walls: non-zero boundary velocities), the Shape3D sphere (NODE_BOUND skin with
T = 0 against baseT = 1 -- the reference's uninitialised-BC hazard as zero-filled memory) and the white_sea depth map (SeaNetCDF:
a ragged sea bed, in/out streams on two faces).  Bit-exact kernels: equal to the oracle value for value, fp32 and fp64;
partition kernels (the fp32 default): within the stated tolerance and no further from the fp64 solution than the sequential fp32
arithmetic."""
import os

import numpy as np
import pytest

from cmc_fluid_solver_amd import capi, grids, seanetcdf, shape2d, shape3d

pytestmark = pytest.mark.gpu
INP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")


def f32(x):
    return float(np.float32(x))


def _heart():
    nodes, cfg, dt = shape2d.load_case(os.path.join(INP, "heart_us_2D_data.txt"), os.path.join(INP, "heart_us_2D_config.txt"))
    return nodes, cfg, dt


def _sphere():
    from test_shape3d import icosphere
    import tempfile
    v, f = icosphere(11.0, (40.0, 42.0, 45.0), subdiv=1)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "s.txt")
        shape3d.write_mesh(p, [(v, f)])
        nodes, _ = shape3d.load_shape3d(p, f32(0.001), f32(0.001), f32(0.001), align=True)
    cfg = shape2d.Config()
    cfg.Re, cfg.Pr, cfg.lam = f32(200.0), f32(0.72), f32(1.4)
    # dt: diffusion number nu dt / h^2 = 69 (T).  Beyond ~150 on this input -- a skin at T = 0 against 1 inside, h = 1 mm -- fp32 itself
    # is > 1e-6 away from fp64 and the partition solve up to 3x further than the recurrence (tools/stiffness_check.py)
    return nodes, cfg, 0.01


def _sea():
    cfg = shape2d.Config(os.path.join(INP, "white_sea_config.txt"))
    nodes, _ = seanetcdf.load_seanetcdf(os.path.join(INP, "white_sea_data.nc"), cfg.dx, cfg.dy, cfg.dz, cfg.baseT, cfg.bc_inV, cfg.bc_inT, align=True)
    return nodes, cfg, cfg.frame_time / cfg.time_steps


CASES = {"heart_us_multi_frame": _heart, "shape3d_sphere": _sphere, "white_sea": _sea}


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _vrel(A, B):
    num = sum(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2 for a, b in zip(A[:3], B[:3]))
    den = sum(np.linalg.norm(np.asarray(b, np.float64)) ** 2 for b in B[:3])
    return float(np.sqrt(num / max(den, 1e-300)))


@pytest.mark.parametrize("case", list(CASES))
def test_input_geometries_against_the_oracle(built, case):
    from oracle import oracle as O
    nodes, cfg, dt = CASES[case]()
    G, L, steps = 2, 2, 5
    res = {}
    for tag, dtype, kernel in (("exact32", np.float32, capi.SWEEP_EXACT), ("exact64", np.float64, capi.SWEEP_EXACT), ("part", np.float32, capi.SWEEP_AUTO)):
        params = capi.fluid_params(dtype, cfg.Re, cfg.Pr, cfg.lam)
        s = capi.Solver(nodes, params, dtype)
        s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
        o = O.Oracle(nodes, params, dtype) if tag != "part" else None
        if o is not None:
            assert s.num_segments == [o.num_segments(k) for k in range(3)]
        for i in range(steps):
            s.UpdateBoundaries()
            e = s.TimeStep(dtype(dt), G, L, True)
            if o is not None:
                o.update_boundaries()
                rc, eo = o.time_step(dtype(dt), G, L, True)
                assert rc == 0
        res[tag] = s.download_layer(capi.LAYER_CUR)
        if o is not None:
            assert abs(e - eo) <= 1e-10 * abs(eo) + 1e-300, (case, tag, e, eo)
            for a, b in zip(res[tag], o.get_layer_fields(O.L_CUR)):
                assert np.array_equal(a, b), "%s, %s: HIP fields differ from the CPU oracle" % (case, tag)
            o.close()
        else:
            assert set(s.last_sweep_kernels().values()) == {"part"}, s.last_sweep_kernels()
        s.close()
    A, E32, F64 = res["part"], res["exact32"], res["exact64"]
    assert all(np.isfinite(a).all() for a in A)
    ep, er = _vrel(A, F64), _vrel(E32, F64)
    tp, tr = _rel(A[3], F64[3]), _rel(E32[3], F64[3])
    print("%s: %s cells, velocity rel-L2 vs fp64: partition %.2e, sequential fp32 %.2e; T %.2e / %.2e; partition vs fp32 oracle %.2e / %.2e" % (
        case, nodes.shape, ep, er, tp, tr, _vrel(A, E32), _rel(A[3], E32[3])))
    # two fp32 computations against each other (both roundings in the difference); the Shape3D sphere is the stiff one: its skin holds
    # T = 0 against 1 inside
    assert _vrel(A, E32) <= 2e-6 and _rel(A[3], E32[3]) <= 2e-6
    assert ep <= 1.5 * er + 1e-7 and tp <= 1.5 * tr + 1e-7
