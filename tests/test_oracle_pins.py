"""CPU tests of the oracle (no GPU).

PINNED TO THE REFERENCE ITSELF (oracle/_ref = the reference's own Common/Algorithms.h and
Common/Geometry.h compiled as they lie, see oracle/Makefile):
  - SolveTridiagonal (fp32 and fp64), bit for bit;
  - FluidParams (both constructors) and AlignBy32.
PINNED TO REFERENCE OUTPUTS RECORDED IN SURVEY.md (the reference binary run by the survey):
  - see tests/test_grid_loader.py (grid dims, NODE_IN counts, err range of the shipped 64^3 example).
PINNED TO THE REFERENCE'S WHOLE CPU PATH since round 3: tests/test_ref_golden.py holds the oracle bit for bit to fixtures produced
by the reference's own translation units (oracle/Makefile target ref_full).  The hand-computed cases, algebraic properties and the
self-generated golden vectors of BuildMatrix/DissFunc/merge/EvalDivError below remain as unit-level checks of the restatement.
"""
import ctypes as C
import os

import numpy as np
import pytest

from cmc_fluid_solver_amd import capi, grids
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("dtype,ct,name", [(np.float32, C.c_float, "f32"), (np.float64, C.c_double, "f64")])
def test_thomas_is_the_reference_thomas(dtype, ct, name):
    rl = O.ref_lib()
    rng = np.random.default_rng(1)
    for n in (2, 3, 5, 64, 257):
        for rep in range(5):
            a = rng.uniform(-1, 1, n).astype(dtype); c = rng.uniform(-1, 1, n).astype(dtype)
            b = rng.uniform(2.1, 4, n).astype(dtype); d = rng.uniform(-5, 5, n).astype(dtype)
            x = O.tridiag(a, b, c, d)
            aa, bb, cc, dd = [v.copy() for v in (a, b, c, d)]
            xr = np.zeros_like(a)
            getattr(rl, "ref_tridiag_" + name)(*[v.ctypes.data_as(C.POINTER(ct)) for v in (aa, bb, cc, dd, xr)], n)
            assert np.array_equal(x, xr)


@needs_ref
def test_fluid_params_and_align_are_the_reference_ones():
    rl = O.ref_lib()
    pf = C.POINTER(C.c_float)
    for Re, Pr, lam in ((200.0, 0.72, 1.4), (1000.0, 0.7, 1.3), (50.0, 6.9, 1.01)):
        ref = np.zeros(4, np.float32)
        rl.ref_fluid_params_normalized_f32(Re, Pr, lam, ref.ctypes.data_as(pf))
        assert np.array_equal(O.fluid_params(np.float32, Re, Pr, lam), ref)
        assert np.array_equal(np.array(capi.fluid_params(np.float32, Re, Pr, lam), np.float32), ref)
    ref = np.zeros(4, np.float32)
    rl.ref_fluid_params_physical_f32(0.001, 998.0, 287.0, 0.6, 4180.0, ref.ctypes.data_as(pf))
    assert np.array_equal(np.array(capi.fluid_params_physical(np.float32, 0.001, 998.0, 287.0, 0.6, 4180.0), np.float32), ref)
    for n in list(range(0, 200)) + [255, 256, 257, 1023]:
        assert O.lib().fs3d_oracle_align_by_32(n) == rl.ref_align_by_32(n)


def test_thomas_solves_the_system():
    rng = np.random.default_rng(2)
    n = 50
    a = rng.uniform(-1, 1, n); c = rng.uniform(-1, 1, n); b = rng.uniform(3, 4, n); d = rng.uniform(-1, 1, n)
    a[0] = 0
    x = O.tridiag(a, b, c, d)
    c2 = c.copy(); c2[-1] = 0
    res = b * x + np.concatenate([[0], a[1:] * x[:-1]]) + np.concatenate([c2[:-1] * x[1:], [0]]) - d
    assert np.abs(res).max() < 1e-12


def test_segment_generation_semantics():
    """Grid3D::GenerateListSegments (Grid3D.cpp:47-127) on hand-made lines."""
    g = grids.box(8, 7, 9)
    o = O.Oracle(g, capi.fluid_params(np.float64, 200, 0.72, 1.4), np.float64)
    # empty box: one segment per interior line, spanning the whole dim
    assert [o.num_segments(d) for d in range(3)] == [5 * 7, 6 * 7, 6 * 5]
    assert all(s[6] == 8 and s[0] == 0 and s[3] == 7 for s in o.segments(O.X))
    assert all(s[6] == 9 for s in o.segments(O.Z))
    # a wall in the middle of x splits every X line in two; the wall cell closes one and opens the next
    g.type[4, 1:-1, 1:-1] = grids.NODE_BOUND
    o = O.Oracle(g, capi.fluid_params(np.float64, 200, 0.72, 1.4), np.float64)
    sx = o.segments(O.X)
    assert len(sx) == 2 * 5 * 7
    assert {(s[0], s[3]) for s in sx} == {(0, 4), (4, 7)}
    # a run of NODE_IN that reaches the end of the line without a closing cell is dropped
    g2 = grids.box(8, 7, 9)
    g2.type[7, 3, 4] = grids.NODE_IN
    o2 = O.Oracle(g2, capi.fluid_params(np.float64, 200, 0.72, 1.4), np.float64)
    assert o2.num_segments(O.X) == 5 * 7 - 1


def test_interior_row_against_hand_formula():
    """One sweep on a field with a single perturbed cell: the x that comes out must satisfy the rows of
    AdiSolver3D::BuildMatrix (AdiSolver3D.cpp:755-801) evaluated independently in numpy (fp64)."""
    g = grids.box(9, 8, 10, h=0.1)
    params = capi.fluid_params(np.float64, 200.0, 0.72, 1.4)
    o = O.Oracle(g, params, np.float64)
    rng = np.random.default_rng(3)
    cur = [rng.uniform(-0.1, 0.1, g.shape) for _ in range(3)] + [1 + rng.uniform(-0.01, 0.01, g.shape)]
    tmp = [rng.uniform(-0.1, 0.1, g.shape) for _ in range(3)] + [1 + rng.uniform(-0.01, 0.01, g.shape)]
    for v in range(4):
        o.set_field(O.L_CUR, v, cur[v]); o.set_field(O.L_TEMP, v, tmp[v])
    dt = 0.1
    o.sweep(O.Y, dt, O.L_CUR, O.L_TEMP, O.L_NEXT)
    x = o.get_layer_fields(O.L_NEXT)
    v_T, v_vis, t_vis, t_phi = [float(p) for p in params]
    h = g.dx
    i, k = 4, 5
    for j in range(1, g.dimy - 1):          # interior rows of the Y line (i, :, k)
        q = tmp[1][i, j, k] / (2 * h)
        for var, vis in ((0, v_vis), (1, v_vis), (2, v_vis), (3, t_vis)):
            a = -q - vis / h**2; b = 3 / dt + 2 * vis / h**2; c = q - vis / h**2
            d = cur[var][i, j, k] * 3 / dt
            if var == 1:
                d -= v_T * (tmp[3][i, j + 1, k] - tmp[3][i, j - 1, k]) / (2 * h)
            if var == 3:
                dy = lambda f: (f[i, j + 1, k] - f[i, j - 1, k]) / (2 * h)
                u_y, v_y, w_y = dy(tmp[0]), dy(tmp[1]), dy(tmp[2])
                v_x = (tmp[1][i + 1, j, k] - tmp[1][i - 1, j, k]) / (2 * h)
                v_z = (tmp[1][i, j, k + 1] - tmp[1][i, j, k - 1]) / (2 * h)
                d += t_phi * (u_y * u_y + 2 * v_y * v_y + w_y * w_y + u_y * v_x + w_y * v_z)
            lhs = a * x[var][i, j - 1, k] + b * x[var][i, j, k] + c * x[var][i, j + 1, k]
            assert lhs == pytest.approx(d, rel=1e-11, abs=1e-11)
    # boundary rows: NOSLIP velocity = node value, FREE temperature: 2*x0 - x1 = 0 (ApplyBC0, :804-827)
    assert x[0][i, 0, k] == 0.0 and x[3][i, 0, k] == pytest.approx(x[3][i, 1, k] / 2, rel=1e-14)


def test_merge_and_div_error_definitions():
    g = grids.box(7, 8, 9, h=0.1)
    o = O.Oracle(g, capi.fluid_params(np.float64, 200, 0.72, 1.4), np.float64)
    rng = np.random.default_rng(4)
    A = [rng.uniform(-1, 1, g.shape) for _ in range(4)]
    B = [rng.uniform(-1, 1, g.shape) for _ in range(4)]
    for v in range(4):
        o.set_field(O.L_NEXT, v, A[v]); o.set_field(O.L_TEMP, v, B[v])
    o.merge(O.L_NEXT, O.L_TEMP)
    inside = g.type == grids.NODE_IN
    for v, f in enumerate(o.get_layer_fields(O.L_TEMP)):
        assert np.array_equal(f[inside], ((B[v] + A[v]) / 2)[inside]) and np.array_equal(f[~inside], B[v][~inside])
    # EvalDivError (TimeLayer3D.h:595-641) in numpy
    U, V, W = A[:3]
    e, n = o.eval_div_error(O.L_NEXT)
    tot, cnt = 0.0, 0
    for i in range(g.dimx - 1):
        for j in range(g.dimy - 1):
            for k in range(g.dimz - 1):
                if g.type[i, j, k] != grids.NODE_IN:
                    continue
                ex = (U[i, j, k] + U[i, j-1, k] + U[i, j-1, k-1] + U[i, j, k-1] - U[i-1, j, k] - U[i-1, j-1, k] - U[i-1, j-1, k-1] - U[i-1, j, k-1]) * g.dz * g.dy / 4
                ey = (V[i, j, k] + V[i-1, j, k] + V[i-1, j, k-1] + V[i, j, k-1] - V[i, j-1, k] - V[i-1, j-1, k] - V[i-1, j-1, k-1] - V[i, j-1, k-1]) * g.dx * g.dz / 4
                ez = (W[i, j, k] + W[i, j-1, k] + W[i-1, j-1, k] + W[i-1, j, k] - W[i, j, k-1] - W[i, j-1, k-1] - W[i-1, j-1, k-1] - W[i-1, j, k-1]) * g.dx * g.dy / 4
                tot += abs(ex + ey + ez); cnt += 1
    assert n == cnt and e == pytest.approx(tot / cnt, rel=1e-12)


def test_symmetry_of_the_box_flow():
    """The box, its boundary data and the scheme are symmetric under y -> -y: u, w, T even, v odd."""
    g = grids.box(12, 11, 10, h=0.08)
    o = O.Oracle(g, capi.fluid_params(np.float64, 200, 0.72, 1.4), np.float64)
    for _ in range(3):
        o.update_boundaries(); o.time_step(0.1, 4, 2, True)
    u, v, w, T = o.get_layer_fields(O.L_CUR)
    assert np.abs(u - u[:, ::-1, :]).max() < 1e-12 and np.abs(v + v[:, ::-1, :]).max() < 1e-12
    assert np.abs(T - T[:, ::-1, :]).max() < 1e-12


@pytest.mark.parametrize("case", ["box_16x14x18", "obstacle_20x16x18"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_reproduces_committed_golden_vectors(case, dtype):
    z = np.load(os.path.join(GOLD, "%s_%s.npz" % (case, np.dtype(dtype).name)))
    dims = tuple(int(d) for d in z["dims"])
    mk = grids.box if case.startswith("box") else grids.box_with_obstacle
    g = mk(*dims, h=float(z["h"][0]))
    o = O.Oracle(g, capi.fluid_params(dtype, 200.0, 0.72, 1.4), dtype)
    assert [o.num_segments(d) for d in range(3)] == list(z["nseg"])
    G, L = [int(v) for v in z["GL"]]
    for step in range(1, 6):
        o.update_boundaries()
        rc, e = o.time_step(float(z["dt"][0]), G, L, True)
        assert rc == 0 and e == z["err"][step - 1]
        if "u_step%d" % step in z:
            for v, f in zip("uvwT", o.get_layer_fields(O.L_CUR)):
                assert np.array_equal(f, z["%s_step%d" % (v, step)])
    V, T = o.get_layer()
    assert np.array_equal(V, z["getlayer_V"]) and np.array_equal(T, z["getlayer_T"])


def test_time_step_reports_divergence():
    g = grids.box(10, 10, 10)
    o = O.Oracle(g, capi.fluid_params(np.float32, 200, 0.72, 1.4), np.float32)
    base = [np.ascontiguousarray(a, np.float32) for a in (g.vx, g.vy, g.vz, g.T)]
    for v, f in enumerate(grids.perturb(base, vel=50.0)):
        o.set_field(O.L_CUR, v, f)
    rc, e = o.time_step(0.1, 1, 1, True)
    assert rc == 1 and e > 0.01


# ---- the host-side loaders against the reference's own code (oracle/_ref: Common/Config.h + LinuxIO.cpp, Geometry.h) -------------
INPUTS = os.path.join(GOLD, "inputs")


@needs_ref
@pytest.mark.parametrize("name", ["box_pipe_2D_config.txt", "non_uniform_pipe_2D_config.txt", "white_sea_config.txt", "heart_us_2D_config.txt"])
def test_config_parser_is_the_reference_parser(name):
    """Every value the reference's Config::LoadFromFile leaves behind for a config, against the Python parser (shape2d.Config; the
    C++ one, host/Config.h, is held to the Python one in tests/test_grid_loader.py / test_host_driver.py)."""
    from cmc_fluid_solver_amd import shape2d
    path = os.path.join(INPUTS, name)
    ref = O.ref_config(path)
    assert ref is not None and "rejected" not in ref, ref
    c = shape2d.Config(path)
    for k in ("R_specific", "k", "cv", "baseT", "bc_strength", "bc_inT", "viscosity", "density", "Re", "Pr", "lam", "depth_var", "frame_time", "dx", "dy", "dz", "depth"):
        assert getattr(c, k) == ref[k], (k, getattr(c, k), ref[k])
    assert tuple(c.bc_inV) == (ref["bc_inVx"], ref["bc_inVy"], ref["bc_inVz"])
    for k in ("cycles", "time_steps", "out_time_steps", "outdimx", "outdimy", "outdimz", "num_global", "num_local"):
        assert getattr(c, k) == ref[k], k
    assert bool(c.bc_noslip) == bool(ref["bc_noslip"]) and bool(c.useNormalizedParams) == bool(ref["useNormalizedParams"])
    assert c.out_vars == ref["out_vars"]
    assert ["2D", "3D"][ref["problem_dim"]] == c.problem_dim and ["Shape2D", "Shape3D", "SeaNetCDF"][ref["in_fmt"]] == c.in_fmt
    assert ["NetCDF", "MultiVox"][ref["out_fmt"]] == c.out_fmt and ["Explicit", "ADI", "Stable"][ref["solver"]] == c.solver


@needs_ref
def test_reference_parser_rejects_what_ours_rejects(tmp_path):
    """The shipped heart_us configs (old keys, no out_vars) and configs with a missing key: the reference prints its message and
    exits; the Python / C++ parsers raise with the same message."""
    from cmc_fluid_solver_amd import shape2d
    good = open(os.path.join(INPUTS, "box_pipe_2D_config.txt")).read().replace("\r", "")
    cases = {"no_vars": "\n".join(ln for ln in good.split("\n") if not ln.startswith("out_vars")),
             "no_dz": "\n".join(ln for ln in good.split("\n") if not ln.startswith("grid_dz")),
             "no_solver": "\n".join(ln for ln in good.split("\n") if not ln.startswith("solver"))}
    for tag, text in cases.items():
        p = str(tmp_path / (tag + ".txt"))
        open(p, "w").write(text)
        ref = O.ref_config(p)
        assert "rejected" in ref, tag
        with pytest.raises(ValueError) as ei:
            shape2d.Config(p)
        assert str(ei.value) in ref["rejected"], (tag, str(ei.value), ref["rejected"])


@needs_ref
def test_bounding_boxes_are_the_reference_ones(tmp_path):
    """BBox2D::Build / BBox3D::Build (Geometry.h:464-480, 510-529) on the points of the fixture inputs, all frames."""
    from cmc_fluid_solver_amd import shape2d, shape3d
    rl = O.ref_lib()
    pf, pi = C.POINTER(C.c_float), C.POINTER(C.c_int)
    for name in ("box_pipe_2D_data.txt", "non_uniform_pipe_2D_data.txt", "heart_us_2D_data.txt"):
        frames = shape2d.parse_shape2d(open(os.path.join(INPUTS, name)).read())
        pts = [np.array([p for sh in fr["shapes"] for p in sh["points"]], np.float32) for fr in frames]
        npts = np.array([len(p) for p in pts], np.int32)
        xy = np.ascontiguousarray(np.concatenate(pts), np.float32)
        out = np.zeros(4, np.float32)
        rl.ref_bbox2d_build(len(frames), npts.ctypes.data_as(pi), xy.ctypes.data_as(pf), out.ctypes.data_as(pf))
        g2 = shape2d.Grid2D(frames, 0.01, 0.01, 1.0, False)
        assert tuple(np.float32(v) for v in g2.bbox) == tuple(out), name
    rng = np.random.default_rng(5)
    vs = [rng.uniform(-30, 50, (40, 3)), rng.uniform(-35, 45, (40, 3))]
    tri = np.array([[0, 1, 2]])
    path = str(tmp_path / "m.txt")
    shape3d.write_mesh(path, [(v, tri) for v in vs])
    frames = shape3d.parse_shape3d(open(path).read())
    sh = shape3d.Shape3D(frames, 0.001, 0.001, 0.001, False)
    nv = np.array([40, 40], np.int32)
    xyz = np.ascontiguousarray(np.concatenate([fr["v"] for fr in frames]), np.float32)
    out = np.zeros(6, np.float32)
    rl.ref_bbox3d_build(2, nv.ctypes.data_as(pi), xyz.ctypes.data_as(pf), out.ctypes.data_as(pf))
    assert tuple(np.float32(v) for v in sh.bbox) == tuple(out)


@needs_ref
def test_depth_resampling_is_the_reference_one():
    """DepthInfo3D(nx, ny, info) (Geometry.h:429-441): the `d` variable of a SeaNetCDF run and the depth lookup of the loader."""
    from cmc_fluid_solver_amd import seanetcdf
    rl = O.ref_lib()
    pf = C.POINTER(C.c_float)
    rng = np.random.default_rng(6)
    src = rng.uniform(-200, 100, (301, 722)).astype(np.float32)
    for nx, ny in ((150, 100), (160, 96), (301, 722), (7, 1000)):
        out = np.zeros((nx, ny), np.float32)
        rl.ref_depth_resample(nx, ny, 301, 722, src.ctypes.data_as(pf), out.ctypes.data_as(pf))
        assert np.array_equal(out, seanetcdf.resample_depths(src, nx, ny))
