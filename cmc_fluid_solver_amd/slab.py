"""x-slab decomposition helpers (host logic shared by bench.py, the tests and multi-GPU callers).

The reference splits the grid along x, its slowest axis (GPUplan::splitEven1D, GPUplan.cpp:122-141;
PARAplan::get1D, PARAplan.cpp:71-87): contiguous slabs, one ghost plane per neighbour.
"""
import numpy as np


def slab_range(dimx, rank, nranks):
    """Planes [x0, x1) of `rank`: even split, the remainder goes to the first ranks."""
    q, r = divmod(dimx, nranks)
    x0 = rank * q + min(rank, r)
    return x0, x0 + q + (1 if rank < r else 0)


def thomas_forward_slab(a, b, c, d, carry):
    """Forward elimination (Algorithms.h:23-32) of the rows of ONE slab of a line batch.

    a,b,c,d: [n_local, nlines] rows of this slab (row 0 of the first slab has a == 0);
    carry: (c', d') reached by the previous slab at its last row, or None for the first slab.
    Returns cp[n_local, nlines], dp[n_local, nlines], and the outgoing carry.
    Same operations, in the same order, as the unsplit recurrence -> bit-identical results.
    """
    n = a.shape[0]
    cp = np.empty_like(c)
    dp = np.empty_like(d)
    if carry is None:
        cp[0] = c[0] / b[0]
        dp[0] = d[0] / b[0]
        start = 1
    else:
        pc, pd = carry
        start = 0
    for i in range(start, n):
        if i > 0:
            pc, pd = cp[i - 1], dp[i - 1]
        den = b[i] - a[i] * pc
        cp[i] = c[i] / den
        dp[i] = (d[i] - pd * a[i]) / den
    return cp, dp, (cp[n - 1].copy(), dp[n - 1].copy())


def thomas_backward_slab(cp, dp, xcarry):
    """Back-substitution (Algorithms.h:34-37) of one slab; xcarry = x of the next slab's first row,
    or None for the last slab (x[n-1] = d'[n-1]).  Returns x[n_local, nlines] and x[0]."""
    n = cp.shape[0]
    x = np.empty_like(dp)
    if xcarry is None:
        x[n - 1] = dp[n - 1]
        nxt = x[n - 1]
        start = n - 2
    else:
        nxt = xcarry
        start = n - 1
    for i in range(start, -1, -1):
        x[i] = dp[i] - cp[i] * nxt
        nxt = x[i]
    return x, x[0].copy()
