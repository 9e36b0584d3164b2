"""CPU: the partition (reduced-interface) algebra of cmc_fluid_solver_amd/partition.py -- what csrc/kernels_part.hip
executes inside a workgroup / a wave and what the cross-slab X solve executes across ranks -- against the sequential
Thomas solve of the oracle (pinned bit for bit to the reference's Common/Algorithms.h, tests/test_oracle_pins.py).
Algebraically exact: fp64 agrees to round-off, fp32 to a few ulp of the solution norm."""
import numpy as np
import pytest

from cmc_fluid_solver_amd import partition as pt


def _system(n, nlines, dtype, seed, nrhs=4):
    """rows shaped like the solver's: weakly dominant interior rows b = 3/dt + 2 vis, a/c = -vis -+ q, BC rows at the ends"""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-130, 130, (n, nlines))
    a = (-q - 325.0).astype(dtype); c = (q - 325.0).astype(dtype); b = np.full((n, nlines), 680.0, dtype)
    a[0] = 0; b[0] = 1; c[0] = 0            # NOSLIP start row
    a[-1] = -1; b[-1] = 2; c[-1] = 0        # FREE end row
    d = rng.uniform(-5, 5, (n, nrhs, nlines)).astype(dtype)
    return a, b, c, d


def _thomas_ref(a, b, c, d):
    from oracle import oracle as O
    n, nrhs, nl = d.shape
    out = np.empty((n, nrhs, nl), np.float64)
    for l in range(nl):
        for r in range(nrhs):
            out[:, r, l] = O.tridiag(a[:, l].astype(np.float64), b[:, l].astype(np.float64), c[:, l].astype(np.float64),
                                     np.ascontiguousarray(d[:, r, l], np.float64))
    return out


CASES = [(256, list(range(0, 257, 16))),            # X / Y sweeps: 16 chunks of 16 cells per line
         (256, list(range(0, 257, 4))),             # Z sweep: 64 lanes x 4 cells
         (128, list(range(0, 129, 4))),
         (37, [0, 5, 6, 20, 37]),                   # ragged chunks, one of a single cell
         (256, [0, 32, 64, 96, 128, 160, 192, 224, 256])]     # 8 x-slabs of 32 planes


@pytest.mark.parametrize("reduced", ["thomas", "pcr"])
@pytest.mark.parametrize("n,bounds", CASES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_partition_solve_equals_thomas(built, dtype, n, bounds, reduced):
    a, b, c, d = _system(n, 12, dtype, seed=n + len(bounds))
    x = pt.solve(a, b, c, d, bounds, reduced)
    ref = _thomas_ref(a, b, c, d)
    err = np.linalg.norm(x - ref) / np.linalg.norm(ref)
    assert err <= (5e-15 if dtype == np.float64 else 5e-7), err
    # the solution satisfies the rows (independent of any reference solve)
    x64 = x.astype(np.float64)
    res = b[:, None].astype(np.float64) * x64 - d
    res[1:] += a[1:, None] * x64[:-1]
    res[:-1] += c[:-1, None] * x64[1:]
    assert np.abs(res).max() <= (1e-9 if dtype == np.float64 else 2e-2) * 1.0


def test_identity_rows_decouple_segments():
    """SKIP rows (identity, d = 0) between two segments of a line: each segment's solution is its own Thomas solve."""
    rng = np.random.default_rng(3)
    n, nl = 64, 5
    a, b, c, d = _system(n, nl, np.float64, 9, nrhs=1)
    for s in (20, 21, 22):
        a[s] = 0; b[s] = 1; c[s] = 0; d[s] = 0
    a[23] = 0; c[19] = 0                    # START of the second segment, END of the first
    x = pt.solve(a, b, c, d, list(range(0, 65, 8)))
    ref = _thomas_ref(a, b, c, d)
    assert np.abs(x - ref).max() <= 1e-12
    assert np.abs(x[20:23]).max() == 0
