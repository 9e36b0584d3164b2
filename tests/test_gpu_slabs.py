"""The x-slab (multi-GPU) path on ONE card: several slab contexts joined by the in-process transport
(fs3d_comm_init_local), one host thread per slab.  Everything but the wire is the code the RCCL ranks
run -- halo planes, ghost-aware sweeps, the cross-slab X sweep with its line-block pipeline, the
two-scalar reduction -- so the decomposed result must equal the single-context result value for value,
and the oracle's.  (AdiSolver3D.cu:524-640 and TimeLayer3D.h:272-335 are the reference's versions.)
"""
import numpy as np
import pytest

from cmc_fluid_solver_amd import capi, grids
from cmc_fluid_solver_amd.slab import slab_range

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True)
def _exact_kernels(monkeypatch):
    """These tests assert bit-equality with the CPU oracle: new contexts start on the bit-exact kernels
    (FS3D_SWEEP_EXACT).  The partition kernels (the fp32 default) have their own tolerance tests in test_gpu_part.py."""
    monkeypatch.setenv("FS3D_DEFAULT_KERNEL", "4")


DT = 0.1
PARAMS = (200.0, 0.72, 1.4)


def _single(g, dtype, steps, G, L):
    s = capi.Solver(g, capi.fluid_params(dtype, *PARAMS), dtype)
    s.UpdateBoundaries()
    errs = [s.TimeStep(DT, G, L, True) for _ in range(steps)]
    out = s.download_layer(capi.LAYER_CUR)
    s.close()
    return out, errs


def _slabs(g, dtype, nranks, steps, G, L, xblocks=None, monkeypatch=None):
    if xblocks is not None:
        monkeypatch.setenv("FS3D_XBLOCKS", str(xblocks))
    grp = capi.LocalGroup(g, capi.fluid_params(dtype, *PARAMS), nranks, dtype)

    def work(rank, s):
        s.UpdateBoundaries()
        errs = [s.TimeStep(DT, G, L, True) for _ in range(steps)]
        return s.download_layer(capi.LAYER_CUR), errs
    res = grp.run(work)
    grp.close()
    fields = [np.concatenate([res[r][0][v] for r in range(nranks)], axis=0) for v in range(4)]
    return fields, [res[r][1] for r in range(nranks)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_slabs_equal_single_context(dtype, nranks):
    g = grids.box_with_obstacle(30, 24, 32, h=0.03)          # 30 planes: uneven slabs for 4 ranks
    ref, ref_err = _single(g, dtype, 3, 2, 2)
    got, errs = _slabs(g, dtype, nranks, 3, 2, 2)
    for v in range(4):
        assert np.array_equal(ref[v], got[v]), "field %d differs between %d slabs and one context" % (v, nranks)
    for r in range(nranks):                                   # every rank holds the global error
        np.testing.assert_allclose(errs[r], ref_err, rtol=1e-12)


@pytest.mark.parametrize("xblocks", [1, 3, 8])
def test_x_pipeline_block_count_is_invisible(xblocks, monkeypatch):
    g = grids.box_with_obstacle(24, 40, 48, h=0.03)          # plane = 1920 lines = 30 waves
    ref, _ = _single(g, np.float32, 2, 1, 2)
    got, _ = _slabs(g, np.float32, 3, 2, 1, 2, xblocks=xblocks, monkeypatch=monkeypatch)
    for v in range(4):
        assert np.array_equal(ref[v], got[v])


def test_slabs_against_oracle():
    from oracle import oracle as O
    dtype = np.float32
    g = grids.box(26, 20, 24, h=0.04)
    params = capi.fluid_params(dtype, *PARAMS)
    o = O.Oracle(g, params, dtype)
    o.update_boundaries()
    oerr = [o.time_step(DT, 2, 1)[1] for _ in range(2)]
    got, errs = _slabs(g, dtype, 2, 2, 2, 1)
    for v, b in enumerate(o.get_layer_fields(O.L_CUR)):
        assert np.array_equal(got[v], b)
    np.testing.assert_allclose(errs[0], oerr, rtol=1e-12)


def test_slab_kernel_level_calls():
    """fs3d_sweep / fs3d_eval_div_error are collective too (halo exchange inside)."""
    dtype = np.float32
    g = grids.box_with_obstacle(28, 24, 32, h=0.03)
    base = [np.ascontiguousarray(a, dtype) for a in (g.vx, g.vy, g.vz, g.T)]
    cur = grids.perturb(base, seed=7)
    tmp = grids.perturb(base, seed=8)
    s = capi.Solver(g, capi.fluid_params(dtype, *PARAMS), dtype)
    s.upload_layer(capi.LAYER_CUR, cur); s.upload_layer(capi.LAYER_TEMP, tmp)
    ref = {}
    for d in (capi.DIR_X, capi.DIR_Y, capi.DIR_Z):
        s.sweep(d, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT)
        ref[d] = s.download_layer(capi.LAYER_NEXT)
    ref_err = s.eval_div_error(capi.LAYER_TEMP)
    s.close()

    nranks = 3
    grp = capi.LocalGroup(g, capi.fluid_params(dtype, *PARAMS), nranks, dtype)

    def work(rank, sv):
        x0, x1 = slab_range(g.dimx, rank, nranks)
        sv.upload_layer(capi.LAYER_CUR, [a[x0:x1] for a in cur])
        sv.upload_layer(capi.LAYER_TEMP, [a[x0:x1] for a in tmp])
        out = {}
        for d in (capi.DIR_X, capi.DIR_Y, capi.DIR_Z):
            sv.sweep(d, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT)
            out[d] = sv.download_layer(capi.LAYER_NEXT)
        return out, sv.eval_div_error(capi.LAYER_TEMP)
    res = grp.run(work)
    grp.close()
    for d in ref:
        for v in range(4):
            got = np.concatenate([res[r][0][d][v] for r in range(nranks)], axis=0)
            assert np.array_equal(ref[d][v], got), "dir %d field %d" % (d, v)
    for r in range(nranks):
        assert res[r][1][1] == ref_err[1]
        np.testing.assert_allclose(res[r][1][0], ref_err[0], rtol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("nranks,xblocks", [(2, 1), (3, 4), (4, 3)])
def test_pipe_kernel_halves_across_slabs(dtype, nranks, xblocks, monkeypatch):
    """dimz a multiple of 64: the X sweep of each slab runs as the forward / backward half of the pipe kernel
    (rows through the HBM scratch, carries between the ranks) -- forced with FS3D_SWEEP_PIPE so that a silent
    fall-back to the thread-per-line halves would fail.  Lines of 50 cells cut into uneven slabs, an obstacle
    (two segments per line, segment ends inside slabs), tiny values in part of the state (division core + redo)."""
    monkeypatch.setenv("FS3D_XBLOCKS", str(xblocks))
    g = grids.box_with_obstacle(50, 20, 64, h=0.03)
    params = capi.fluid_params(dtype, *PARAMS)
    s = capi.Solver(g, params, dtype)
    s.UpdateBoundaries()
    ref_err = [s.TimeStep(DT, 2, 2, True) for _ in range(3)]
    ref = s.download_layer(capi.LAYER_CUR)
    s.close()

    grp = capi.LocalGroup(g, params, nranks, dtype)
    for sv in grp.solvers:
        sv.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_PIPE)

    def work(rank, sv):
        sv.UpdateBoundaries()
        errs = [sv.TimeStep(DT, 2, 2, True) for _ in range(3)]
        return sv.download_layer(capi.LAYER_CUR), errs
    res = grp.run(work)
    grp.close()
    for v in range(4):
        got = np.concatenate([res[r][0][v] for r in range(nranks)], axis=0)
        assert np.array_equal(ref[v], got), "field %d" % v
    np.testing.assert_allclose(res[0][1], ref_err, rtol=1e-12)


def test_pipe_kernel_halves_long_slabs():
    """Slabs of 100 planes: the 32-cells-per-wave instance of the halves, relay over several waves per slab."""
    dtype = np.float32
    g = grids.box(200, 8, 64, h=0.02)
    params = capi.fluid_params(dtype, *PARAMS)
    base = [np.ascontiguousarray(a, dtype) for a in (g.vx, g.vy, g.vz, g.T)]
    cur, tmp = grids.perturb(base, seed=5), grids.perturb(base, seed=6)
    s = capi.Solver(g, params, dtype)
    s.upload_layer(capi.LAYER_CUR, cur); s.upload_layer(capi.LAYER_TEMP, tmp)
    s.sweep(capi.DIR_X, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
    ref_next, ref_tmp = s.download_layer(capi.LAYER_NEXT), s.download_layer(capi.LAYER_TEMP)
    s.close()
    nranks = 2
    grp = capi.LocalGroup(g, params, nranks, dtype)
    for sv in grp.solvers:
        sv.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_PIPE)

    def work(rank, sv):
        x0, x1 = slab_range(g.dimx, rank, nranks)
        sv.upload_layer(capi.LAYER_CUR, [a[x0:x1] for a in cur])
        sv.upload_layer(capi.LAYER_TEMP, [a[x0:x1] for a in tmp])
        sv.sweep(capi.DIR_X, DT, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
        return sv.download_layer(capi.LAYER_NEXT), sv.download_layer(capi.LAYER_TEMP)
    res = grp.run(work)
    grp.close()
    for v in range(4):
        assert np.array_equal(ref_next[v], np.concatenate([res[r][0][v] for r in range(nranks)], axis=0)), "next %d" % v
        assert np.array_equal(ref_tmp[v], np.concatenate([res[r][1][v] for r in range(nranks)], axis=0)), "temp %d" % v


@pytest.mark.parametrize("nranks", [2, 4])
def test_shipped_example_on_slabs_100_steps(nranks):
    """The reference's shipped 64^3 box_pipe case (file-driven: loader + config), 100 steps as its main loop runs them,
    on 2 and 4 x-slabs: the err trace and the final fields equal the single-context run's (which the parity suite holds
    against the oracle and the reference's recorded err range)."""
    import os
    from cmc_fluid_solver_amd import shape2d
    inp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")
    nodes, cfg, dt = shape2d.load_case(os.path.join(inp, "box_pipe_2D_data.txt"), os.path.join(inp, "box_pipe_2D_config.txt"))
    params = capi.fluid_params(np.float32, cfg.Re, cfg.Pr, cfg.lam)
    steps = 100

    def loop(sv):
        errs = []
        for i in range(steps):
            sv.UpdateBoundaries()
            errs.append(sv.TimeStep(np.float32(dt), cfg.num_global, cfg.num_local, i % 10 == 0))
        return sv.download_layer(capi.LAYER_CUR), errs

    s = capi.Solver(nodes, params, np.float32)
    ref, ref_err = loop(s)
    s.close()
    assert 1.0e-5 < ref_err[0] < 3.0e-5 and 1.0e-5 < ref_err[-1] < 3.0e-5     # err 1.25e-5 ... 2.3e-5
    grp = capi.LocalGroup(nodes, params, nranks, np.float32)
    res = grp.run(lambda rank, sv: loop(sv))
    grp.close()
    for v in range(4):
        assert np.array_equal(ref[v], np.concatenate([res[r][0][v] for r in range(nranks)], axis=0)), "field %d" % v
    np.testing.assert_allclose(res[0][1], ref_err, rtol=1e-12)


@pytest.mark.parametrize("nranks", [2, 3, 5, 8])
@pytest.mark.parametrize("dtype,kernel", [(np.float32, capi.SWEEP_AUTO), (np.float32, capi.SWEEP_EXACT), (np.float64, capi.SWEEP_EXACT)])
def test_distributed_interface_solve_equals_the_all_gather_form(built, nranks, dtype, kernel):
    """(r3) FS3D_XSOLVE 3: every rank solves the R x R interface systems of the lines it owns and hands every rank its two boundary
    values -- two all-to-alls of point-to-point transfers instead of one all-gather (xGMI is point-to-point: (R-1)/R x 26 words per
    line on the wires instead of (R-1) x 18).  Same operations on the same values: the fields after two steps are BIT-IDENTICAL to
    the all-gather form's, for uneven slabs, a plane whose line count is no multiple of the ranks, an obstacle across the cuts;
    and the default picks it from three ranks on."""
    from cmc_fluid_solver_amd import grids
    g = grids.box_with_obstacle(67, 23, 64, h=0.02)              # 23 x 64 = 1472 lines: not a multiple of 3, 5, 8 x 64
    params = capi.fluid_params(dtype, 200.0, 0.72, 1.4)

    def run(xsolve):
        grp = capi.LocalGroup(g, params, nranks, dtype)

        def work(r, sv):
            sv.set_option(capi.OPT_SWEEP_KERNEL, kernel)
            if xsolve is not None:
                sv.set_option(capi.OPT_XSOLVE, xsolve)
            errs = []
            for i in range(2):
                sv.UpdateBoundaries(); errs.append(sv.TimeStep(0.1, 2, 2, True))
            return sv.last_sweep_kernels()["X"], sv.download_layer(capi.LAYER_CUR), errs
        try:
            return grp.run(work)
        finally:
            grp.close()
    ag, a2a = run(capi.XSOLVE_REDUCED), run(capi.XSOLVE_REDUCED_A2A)
    assert all("all-to-all" in r[0] for r in a2a) and not any("all-to-all" in r[0] for r in ag)
    for r in range(nranks):
        for v in range(4):
            assert np.array_equal(ag[r][1][v], a2a[r][1][v]), "slab %d field %d" % (r, v)
        assert ag[r][2] == a2a[r][2]
    if kernel == capi.SWEEP_AUTO:
        dflt = run(None)
        assert all(("all-to-all" in r[0]) == (nranks >= 3) and "reduced-interface" in r[0] for r in dflt), [r[0] for r in dflt]
        for r in range(nranks):
            for v in range(4):
                assert np.array_equal(dflt[r][1][v], ag[r][1][v])
