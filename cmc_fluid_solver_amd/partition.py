"""Partition (reduced-interface) solve of a batch of tridiagonal lines -- host-side statement of the
algebra that `csrc/kernels_part.hip` executes inside a workgroup (thread chunks of a line) and that the
cross-slab X sweep executes across ranks (one chunk per x-slab).  numpy, any float dtype.

The reference solves every line with the sequential Thomas algorithm (Common/Algorithms.h:21-38) and,
across GPUs, pipelines that recurrence rank by rank (FluidSolver3D/AdiSolver3D.cu:524-640).  Here a line
    a[i] x[i-1] + b[i] x[i] + c[i] x[i+1] = d[i],   a[0] = c[n-1] = 0
is cut into P chunks.  The INTERFACE unknown of chunk p is X_p = x at its last cell.  Every chunk eliminates
its other cells at the same time (no chunk waits for another):
    down-sweep over the cells before the interface:  x[last-1] = G  - V  X_{p-1} - W  X_p
    up-sweep over the same cells:                    x[first]  = G' - V' X_{p-1} - W' X_p
The interface cell's own row then couples X_{p-1}, X_p, X_{p+1}: a P x P tridiagonal system per line (one small
exchange), after which every chunk back-substitutes on its own.  Algebraically exact; the rounding differs from
the sequential recurrence (tolerances: DESIGN.md section 5, tests/test_partition_algebra.py).

Arrays are [cells, lines] (or [cells, nrhs, lines] for d): the recurrences run along axis 0, vectorised over the rest.
"""
import numpy as np


def _bc(m, d):
    """broadcast a matrix coefficient [lines] against a right-hand side [lines] or [nrhs, lines]"""
    return m if d.ndim == m.ndim else m[None]


def chunk_eliminate(a, b, c, d):
    """Interface coefficients of ONE chunk.  a,b,c: [m, lines]; d: [m, lines] or [m, nrhs, lines].

    Returns a dict:
      A, Bp, cl [lines], Dp [.., lines]  -- the interface row with x[last-1] eliminated:
                                            A X_{p-1} + Bp X_p + cl x_first(p+1) = Dp
      Vf, Wf [lines], Gf [.., lines]     -- x_first = Gf - Vf X_{p-1} - Wf X_p
    m == 1: the chunk is its interface cell alone.
    """
    m = a.shape[0]
    if m == 1:
        z = np.zeros_like(a[0])
        return dict(A=a[0].copy(), Bp=b[0].copy(), cl=c[0].copy(), Dp=d[0].copy(),
                    Vf=z, Wf=z - 1, Gf=np.zeros_like(d[0]))
    # down-sweep over cells 0 .. m-2, unknown X_{p-1} to the left
    r = 1 / b[0]
    cp, lp, dp = c[0] * r, a[0] * r, d[0] * _bc(r, d[0])
    for i in range(1, m - 1):
        r = 1 / (b[i] - a[i] * cp)
        dp = (d[i] - _bc(a[i], dp) * dp) * _bc(r, dp)
        lp = -a[i] * lp * r
        cp = c[i] * r
    # up-sweep over cells m-2 .. 0, unknown X_p to the right
    r = 1 / b[m - 2]
    ap, up, ep = a[m - 2] * r, c[m - 2] * r, d[m - 2] * _bc(r, d[0])
    for i in range(m - 3, -1, -1):
        r = 1 / (b[i] - c[i] * ap)
        ep = (d[i] - _bc(c[i], ep) * ep) * _bc(r, ep)
        up = -c[i] * up * r
        ap = a[i] * r
    al, bl, cl, dl = a[m - 1], b[m - 1], c[m - 1], d[m - 1]
    return dict(A=-al * lp, Bp=bl - al * cp, cl=cl.copy(), Dp=dl - _bc(al, dp) * dp, Vf=ap, Wf=up, Gf=ep)


def reduced_rows(co):
    """Rows (lo, di, up, rhs) of the P x P interface system from the chunks' coefficients (list, line order)."""
    P = len(co)
    lo, di, up, rhs = [], [], [], []
    for p in range(P):
        k = co[p]
        if p + 1 < P:
            n = co[p + 1]
            di.append(k["Bp"] - k["cl"] * n["Vf"])
            up.append(-k["cl"] * n["Wf"])
            rhs.append(k["Dp"] - _bc(k["cl"], n["Gf"]) * n["Gf"])
        else:                                   # the line ends here: its last row has c = 0
            di.append(k["Bp"].copy()); up.append(np.zeros_like(k["Bp"])); rhs.append(k["Dp"].copy())
        lo.append(k["A"])
    return np.stack(lo), np.stack(di), np.stack(up), np.stack(rhs)


def thomas(lo, di, up, rhs):
    """Sequential solve of the small interface system (what one thread per line does in the kernel)."""
    P = lo.shape[0]
    cp = np.empty_like(up); dp = np.empty_like(rhs)
    r = 1 / di[0]
    cp[0] = up[0] * r; dp[0] = rhs[0] * _bc(r, rhs[0])
    for p in range(1, P):
        r = 1 / (di[p] - lo[p] * cp[p - 1])
        cp[p] = up[p] * r
        dp[p] = (rhs[p] - _bc(lo[p], dp[p]) * dp[p - 1]) * _bc(r, dp[p])
    x = np.empty_like(rhs)
    x[P - 1] = dp[P - 1]
    for p in range(P - 2, -1, -1):
        x[p] = dp[p] - _bc(cp[p], x[p]) * x[p + 1]
    return x


def pcr(lo, di, up, rhs):
    """Parallel cyclic reduction of the interface system (what the lanes of a wave do in the Z sweep):
    log2(P) steps, every equation eliminates its neighbours at distance 1, 2, 4, ... at once."""
    P = lo.shape[0]
    r = 1 / di
    a, c, d = lo * r, up * r, rhs * (r if rhs.ndim == r.ndim else r[:, None])

    def shift(v, s):        # v[p - s], zero rows outside the line
        out = np.zeros_like(v)
        if s > 0:
            out[s:] = v[:-s]
        else:
            out[:s] = v[-s:]
        return out
    s = 1
    while s < P:
        am, cm, dm = shift(a, s), shift(c, s), shift(d, s)
        ap, cpl, dpl = shift(a, -s), shift(c, -s), shift(d, -s)
        r = 1 / (1 - a * cm - c * ap)
        rb = r if d.ndim == r.ndim else r[:, None]
        ab = a if d.ndim == a.ndim else a[:, None]
        cb = c if d.ndim == c.ndim else c[:, None]
        d = (d - ab * dm - cb * dpl) * rb
        a, c = -a * am * r, -c * cpl * r
        s *= 2
    return d


def chunk_backsub(a, b, c, d, x_left, x_own):
    """Cells of one chunk once its neighbours' interface value x_left = X_{p-1} (None: first chunk) and its own
    x_own = X_p are known: Thomas over the cells before the interface with both ends given."""
    m = a.shape[0]
    x = np.empty_like(d)
    x[m - 1] = x_own
    if m == 1:
        return x
    cp = np.empty_like(c[:m - 1]); dp = np.empty_like(d[:m - 1])
    d0 = d[0] if x_left is None else d[0] - _bc(a[0], d[0]) * x_left
    r = 1 / b[0]
    cp[0] = c[0] * r; dp[0] = d0 * _bc(r, d0)
    for i in range(1, m - 1):
        r = 1 / (b[i] - a[i] * cp[i - 1])
        cp[i] = c[i] * r
        dp[i] = (d[i] - _bc(a[i], dp[i]) * dp[i - 1]) * _bc(r, dp[i])
    nxt = x[m - 1]
    for i in range(m - 2, -1, -1):
        x[i] = dp[i] - _bc(cp[i], nxt) * nxt
        nxt = x[i]
    return x


def solve(a, b, c, d, bounds, reduced="thomas"):
    """The whole line batch: chunks [bounds[p], bounds[p+1]) of the cells."""
    P = len(bounds) - 1
    sl = [slice(bounds[p], bounds[p + 1]) for p in range(P)]
    co = [chunk_eliminate(a[s], b[s], c[s], d[s]) for s in sl]
    X = (thomas if reduced == "thomas" else pcr)(*reduced_rows(co))
    return np.concatenate([chunk_backsub(a[s], b[s], c[s], d[s], X[p - 1] if p else None, X[p])
                           for p, s in enumerate(sl)], axis=0)
