"""BASELINE configs[3]'s workload on one card: the 512^3 fp32 box (lines of 512 cells: 32 chunks per line in the X / Y
partition kernels, the segmented exact kernel for Z), and the same box as 8 x-slabs of 64 planes through the in-process
slab group (the protocol the RCCL ranks run).  Sizes the CPU oracle does not reach in seconds: the checkers are the
thread-per-line kernel (bit-equal to the oracle, tests/test_gpu_parity.py) and size-independent properties."""
import numpy as np
import pytest

from cmc_fluid_solver_amd import capi, grids

pytestmark = pytest.mark.gpu
PARAMS = (200.0, 0.72, 1.4)
N = 512


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def box512():
    return grids.box(N, h=1.0 / (N - 1))


def _seeded(g):
    base = [np.ascontiguousarray(a, np.float32) for a in (g.vx, g.vy, g.vz, g.T)]
    return grids.perturb(base, seed=11), grids.perturb(base, seed=12)


def test_sweeps_512_against_the_thread_per_line_kernel(built, box512):
    """One merged sweep per direction: production kernels (AUTO) vs the thread-per-line kernel on the same seeded state."""
    g = box512
    params = capi.fluid_params(np.float32, *PARAMS)
    cur, tmp = _seeded(g)
    res = {}
    for kernel in (capi.SWEEP_AUTO, capi.SWEEP_LINE):
        s = capi.Solver(g, params, np.float32)
        s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
        zero = np.zeros(g.shape, np.float32)
        for d in range(3):
            s.upload_layer(capi.LAYER_CUR, cur); s.upload_layer(capi.LAYER_TEMP, tmp)
            s.upload_layer(capi.LAYER_NEXT, [zero] * 4)      # cells off the segments keep what `next` held: the same in both runs
            s.sweep(d, 0.1, capi.LAYER_CUR, capi.LAYER_TEMP, capi.LAYER_NEXT, merge=True)
            res[(kernel, d)] = (s.download_layer(capi.LAYER_NEXT), s.download_layer(capi.LAYER_TEMP), s.last_sweep_kernels()["XYZ"[d]])
        s.close()
    for d in range(3):
        A, B = res[(capi.SWEEP_AUTO, d)], res[(capi.SWEEP_LINE, d)]
        print("512^3 dir %d ran %s: next rel-L2 %s" % (d, A[2], ["%.1e" % rel(a, b) for a, b in zip(A[0], B[0])]))
        for v in range(4):
            if A[2] == "part":
                assert rel(A[0][v], B[0][v]) <= 1e-6 and rel(A[1][v], B[1][v]) <= 1e-6, "dir %d field %d" % (d, v)
            else:                                   # the exact kernels: bit for bit
                if not np.array_equal(A[0][v], B[0][v]):
                    bad = np.argwhere(A[0][v] != B[0][v])
                    print("dir %d field %d: %d cells differ, rel-L2 %.2e, first %s: %r vs %r; k range %d..%d" % (d, v, len(bad), rel(A[0][v], B[0][v]),
                          bad[0], A[0][v][tuple(bad[0])], B[0][v][tuple(bad[0])], bad[:, 2].min(), bad[:, 2].max()))
                assert np.array_equal(A[0][v], B[0][v]) and np.array_equal(A[1][v], B[1][v]), "dir %d field %d" % (d, v)
    assert all(res[(capi.SWEEP_AUTO, d)][2] == "part" for d in range(3))


def _steps(g, dtype, kernel, nsteps=2):
    s = capi.Solver(g, capi.fluid_params(dtype, *PARAMS), dtype)
    s.set_option(capi.OPT_SWEEP_KERNEL, kernel)
    errs = []
    for i in range(nsteps):
        s.UpdateBoundaries()
        errs.append(s.TimeStep(0.1, 4, 2, True))
    out = s.download_layer(capi.LAYER_CUR), errs, s.last_sweep_kernels()
    s.close()
    return out


def _vec_rel(A, B):
    num = sum(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2 for a, b in zip(A[:3], B[:3]))
    den = sum(np.linalg.norm(np.asarray(b, np.float64)) ** 2 for b in B[:3])
    return float(np.sqrt(num / den))


def test_two_steps_512_properties_and_yardstick(built, box512):
    """Two time steps of the 512^3 box from rest with the production kernels: finite fields, walls untouched, inflow imposed,
    y mirror symmetry, and the fp64 yardstick of tests/test_gpu_part.py (h = 1/511: two fp32 evaluation orders differ by
    several 1e-6 here; what counts is the distance to the fp64 solution next to the sequential fp32 recurrence's)."""
    g = box512
    A, errs, ran = _steps(g, np.float32, capi.SWEEP_AUTO)
    print("512^3: ran %s, err %s" % (ran, errs))
    assert ran["X"] == "part" and ran["Y"] == "part"
    for a in A:
        assert np.isfinite(a).all()
    assert 0 < errs[-1] < 1e-3
    wall = g.type == grids.NODE_BOUND
    assert np.array_equal(A[0][wall], np.zeros(int(wall.sum()), np.float32))            # no-slip walls keep u = 0
    assert np.array_equal(A[0][0, 1:-1, 1:-1], np.ones((N - 2, N - 2), np.float32))     # inflow valve u = 1
    u = A[0].astype(np.float64)
    assert np.abs(u - u[:, ::-1, :]).max() <= 5e-5 * np.abs(u).max()
    del u
    B, eb, _ = _steps(g, np.float32, capi.SWEEP_EXACT)
    C, ec, _ = _steps(g, np.float64, capi.SWEEP_EXACT)
    ep, er = _vec_rel(A, C), _vec_rel(B, C)
    tp, tr = rel(A[3], C[3]), rel(B[3], C[3])
    print("512^3 after 2 steps, rel-L2 vs fp64: velocity partition %.2e / sequential fp32 %.2e; T %.2e / %.2e" % (ep, er, tp, tr))
    assert ep <= 1.5 * er + 1e-7 and tp <= 1.5 * tr + 1e-7
    assert errs[-1] == pytest.approx(eb[-1], rel=1e-3)


def test_512_as_8_slabs_equals_one_context(built, box512):
    """The slab protocol at configs[3]'s size: 8 x-slabs of 64 planes (in-process group, one thread per slab on the same card;
    halo planes before every sweep, the cross-slab X halves, the 2-scalar error reduction) against ONE context, both on the
    bit-exact kernels: fields equal value for value, the divergence error to 1e-12."""
    from cmc_fluid_solver_amd.slab import slab_range
    g = box512
    params = capi.fluid_params(np.float32, *PARAMS)
    A, errs, ran = _steps(g, np.float32, capi.SWEEP_EXACT)
    grp = capi.LocalGroup(g, params, 8, np.float32)

    def work(r, sv):
        sv.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_EXACT)
        out = []
        for i in range(2):
            sv.UpdateBoundaries()
            out.append(sv.TimeStep(0.1, 4, 2, True))
        return out, sv.download_layer(capi.LAYER_CUR)
    res = grp.run(work)
    assert [slab_range(N, r, 8) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    for v in range(4):
        full = np.concatenate([res[r][1][v] for r in range(8)], axis=0)
        assert np.array_equal(full, A[v]), "field %d: 8 slabs != one context" % v
    grp.close()
    for r in range(8):
        assert res[r][0][-1] == pytest.approx(errs[-1], rel=1e-12)


def test_512_cubed_equals_the_reference(built):
    """(r3) BASELINE configs[3]'s grid held to the REFERENCE ITSELF: tests/golden/ref_box512_f32.npz is one step of the reference's
    own CPU build on the file-driven 512^3 box (grid_dx = grid_dy 0.0021, grid_dz 0.002: 112,135,625 NODE_IN cells).  The loader's
    node arrays hash to the reference's; the bit-exact kernels -- one context, and the same grid as 8 x-slabs of 64 planes through
    the in-process group with the rank pipeline -- return its fields bit for bit (sha256 of the raw arrays); the production
    kernels stay within the stated distance on the fixture's strided sample."""
    import refgolden as RG
    from cmc_fluid_solver_amd.slab import slab_range
    fx = RG.Fixture("box512", "f32")
    m = fx.meta
    nodes, cfg, dt = fx.loader()
    assert nodes.shape == (512, 512, 512) and nodes.count(grids.NODE_IN) == m["node_in"] == 112135625
    assert RG.sha(nodes.type) == m["nodes_sha"]["type"] and RG.sha(np.asarray(nodes.T, np.float32)) == m["nodes_sha"]["T"]
    params = fx.params()
    step_dt, G, L = fx.step_dt(), cfg.num_global, cfg.num_local
    want = m["step_sha"]["1"]

    s = capi.Solver(nodes, params, np.float32)
    s.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_EXACT)
    s.UpdateBoundaries(); err = s.TimeStep(step_dt, G, L, True)
    one = s.download_layer(capi.LAYER_CUR)
    s.close()
    for v, a in zip("UVWT", one):
        assert RG.sha(a) == want[v], "%s after step 1 differs from the reference at 512^3" % v
    assert "%.8f" % err == "%.8f" % m["err_trace"][0]
    del one

    grp = capi.LocalGroup(nodes, params, 8, np.float32)

    def work(r, sv):
        sv.set_option(capi.OPT_SWEEP_KERNEL, capi.SWEEP_EXACT)
        sv.UpdateBoundaries(); e = sv.TimeStep(step_dt, G, L, True)
        return sv.download_layer(capi.LAYER_CUR), e
    res = grp.run(work)
    grp.close()
    assert [slab_range(512, r, 8) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    for k, v in enumerate("UVWT"):
        assert RG.sha(np.concatenate([res[r][0][k] for r in range(8)], axis=0)) == want[v], "8 slabs: %s differs from the reference" % v
    assert all(r[1] == pytest.approx(m["step_err"]["1"], rel=1e-10) for r in res)
    del res

    s = capi.Solver(nodes, params, np.float32)
    s.UpdateBoundaries(); s.TimeStep(step_dt, G, L, True)
    got = [a[::16, ::16, ::16] for a in s.download_layer(capi.LAYER_CUR)]
    assert set(s.last_sweep_kernels().values()) == {"part"}
    s.close()
    smp = [fx.sample(v, 1) for v in "UVWT"]
    mask = fx.z["node_type"][::16, ::16, ::16] != 1
    vel = RG.rel_l2(np.stack(got[:3]), np.stack(smp[:3]), np.stack([mask] * 3))
    tt = RG.rel_l2(got[3], smp[3], mask)
    print("512^3 step 1: production kernels vs the reference: velocity %.3g  T %.3g" % (vel, tt))
    assert vel <= 5.7e-6 and tt <= 3.6e-6           # measured 3.8e-6 / 2.4e-6 (condition number of the line systems ~ 340 at h = 1/511)
