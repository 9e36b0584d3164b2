"""The reference's 2D path (CPU only): explicit advection/diffusion step + pressure projection over a Grid2D.

Python twin of cmc_fluid_solver_amd/host/Stable2D.h (same operations in np.float32 / double; the header lists the reference lines --
StableSolver2D.cpp:21-234, TimeLayer2D.h:24-186, Solver2D.cpp:21-84 -- and the three deliberate readings).  Pinned to the reference (r3): equal to the reference's own StableSolver2D + Grid2D bit for bit
on the fixtures tests/golden/ref2d_*.npz (tests/test_stable2d.py).
Plain Python loops for the Gauss-Seidel sweeps (they are sequential by definition): small grids only.
"""
import numpy as np

from .grids import NODE_BOUND, NODE_IN, NODE_OUT, NODE_VALVE

F = np.float32
DIV_ERR_THRESHOLD, POISSON_ERR_THRESHOLD, MAX_GLOBAL_ITERS = 0.1, 1e-2, 100


class Stable2D:
    def __init__(self, grid, v_vis):
        self.g = grid
        self.v_vis = F(v_vis)
        self.nx, self.ny = grid.dimx, grid.dimy
        self.hx, self.hy = F(grid.dx), F(grid.dy)
        z = lambda: [np.zeros((self.nx, self.ny), np.float32) for _ in range(3)]
        self.cur, self.next, self.temp, self.next_w = z(), z(), z(), z()
        self.cur[0][:] = grid.velx; self.cur[1][:] = grid.vely; self.cur[2][:] = grid.T
        self._copy_all(self.cur, self.next); self._copy_all(self.cur, self.temp)
        self.err, self.poisson_sweeps, self.global_iters = 0.0, 0, 0

    def _copy_type(self, a, b, t):
        m = self.g.cell == t
        m[-1, :] = False; m[:, -1] = False                       # the last row and column are left out (TimeLayer2D.h:109-150)
        for x, y in zip(a, b):
            y[m] = x[m]

    def _copy_all(self, a, b):
        for t in (NODE_IN, NODE_OUT, NODE_BOUND, NODE_VALVE):
            self._copy_type(a, b, t)

    def update_boundaries(self):
        g = self.g
        m = (g.cell == NODE_BOUND) | (g.cell == NODE_VALVE)
        self.cur[0][m] = g.velx[m]; self.cur[1][m] = g.vely[m]; self.cur[2][m] = g.T[m]
        self._copy_type(self.cur, self.next, NODE_BOUND); self._copy_type(self.cur, self.next, NODE_VALVE)

    def _div_error(self, l):
        c = self.g.cell == NODE_IN
        m = c[:-1, :-1] & c[1:, :-1] & c[:-1, 1:] & c[1:, 1:]
        u, v = l[0], l[1]
        tx = (self.hy * (u[1:, :-1] - u[:-1, :-1]) + (u[1:, 1:] - u[:-1, 1:]) / F(2)).astype(np.float32)
        ty = (self.hx * (v[:-1, 1:] - v[:-1, :-1]) + (v[1:, 1:] - v[1:, :-1]) / F(2)).astype(np.float32)
        e, count = F(0), 0
        a = np.abs((tx + ty).astype(np.float32))
        for i in range(self.nx - 1):                              # float accumulation in the reference's order
            for j in range(self.ny - 1):
                if m[i, j]:
                    e = F(e + a[i, j]); count += 1
        with np.errstate(all="ignore"):
            return float(e / F(count)) if count else float("nan")

    def _advect(self, dt, fc, ft, fn, inner):
        hx, hy, nu = self.hx, self.hy, self.v_vis
        tu, tv = self.temp[0], self.temp[1]
        for i, j in inner:
            fx = F(F(ft[i + 1, j] - ft[i - 1, j]) / F(F(2) * hx)); fy = F(F(ft[i, j + 1] - ft[i, j - 1]) / F(F(2) * hy))
            lap = F(F(F(F(ft[i + 1, j] - F(F(2) * ft[i, j])) + ft[i - 1, j]) / F(hx * hx)) + F(F(F(ft[i, j + 1] - F(F(2) * ft[i, j])) + ft[i, j - 1]) / F(hy * hy)))
            rhs = F(F(F(F(-tu[i, j]) * fx) - F(tv[i, j] * fy)) + F(nu * lap))
            fn[i, j] = F(fc[i, j] + F(dt * rhs))

    def _project(self, w, proj, inner, bound):
        g = self.g
        hx, hy = self.hx, self.hy
        div = np.zeros((self.nx, self.ny), np.float32)
        for i, j in inner:
            div[i, j] = F(F(F(w[0][i + 1, j] - w[0][i - 1, j]) / F(F(2) * hx)) + F(F(w[1][i, j + 1] - w[1][i, j - 1]) / F(F(2) * hy)))
        dx2, dy2 = g.dx * g.dx, g.dy * g.dy
        rcp = 0.5 / (dx2 + dy2)
        q = np.zeros((self.nx, self.ny), np.float32)
        cell = g.cell
        while True:
            e = 0.0

            def relax(i, j, i0, i1, j0, j1):
                nonlocal e
                qn = rcp * ((i0 + i1) * dy2 + (j0 + j1) * dx2 - float(div[i, j]) * dx2 * dy2)
                d = qn - float(q[i, j])
                ce = abs(d / qn) if qn != 0.0 else (float("nan") if d == 0.0 else float("inf"))
                if ce > e:
                    e = ce
                q[i, j] = F(qn)
            for i, j in bound:
                i0 = float(q[i - 1, j]) if cell[i - 1, j] == NODE_IN else float(q[i + 1, j])
                i1 = float(q[i + 1, j]) if cell[i + 1, j] == NODE_IN else float(q[i - 1, j])
                j0 = float(q[i, j - 1]) if cell[i, j - 1] == NODE_IN else float(q[i, j + 1])
                j1 = float(q[i, j + 1]) if cell[i, j + 1] == NODE_IN else float(q[i, j - 1])
                relax(i, j, i0, i1, j0, j1)
            for i, j in inner:
                relax(i, j, float(q[i - 1, j]), float(q[i + 1, j]), float(q[i, j - 1]), float(q[i, j + 1]))
            self.poisson_sweeps += 1
            if not e >= POISSON_ERR_THRESHOLD:
                break
        for i, j in inner:
            proj[0][i, j] = F(w[0][i, j] - F(F(q[i + 1, j] - q[i - 1, j]) / F(F(2) * hx)))
            proj[1][i, j] = F(w[1][i, j] - F(F(q[i, j + 1] - q[i, j - 1]) / F(F(2) * hy)))

    def time_step(self, dt, num_global, num_local=1):
        g = self.g
        dt = F(dt)
        self._copy_all(self.cur, self.temp)
        inner = [(i, j) for i in range(self.nx) for j in range(self.ny) if g.cell[i, j] == NODE_IN]
        bound = [(i, j) for i in range(self.nx) for j in range(self.ny) if g.cell[i, j] in (NODE_BOUND, NODE_VALVE)]
        for i, j in inner + bound:
            if i in (0, self.nx - 1) or j in (0, self.ny - 1):
                raise ValueError("2D solver: a fluid or boundary cell lies on the edge of the grid")
        self.poisson_sweeps = 0
        err = self._div_error(self.next)
        it = 0
        while it < num_global or err > DIV_ERR_THRESHOLD:
            self._copy_all(self.cur, self.next_w)
            self._advect(dt, self.cur[0], self.temp[0], self.next_w[0], inner)
            self._advect(dt, self.cur[1], self.temp[1], self.next_w[1], inner)
            self._project(self.next_w, self.next, inner, bound)
            err = self._div_error(self.next)
            m = g.cell == NODE_IN
            m[-1, :] = False; m[:, -1] = False
            for a, b in zip(self.next, self.temp):
                b[m] = ((b[m] + a[m]).astype(np.float32) / F(2)).astype(np.float32)
            if it > MAX_GLOBAL_ITERS:
                raise RuntimeError("Exceeded max number of iterations")
            if err > DIV_ERR_THRESHOLD * 10:
                raise RuntimeError("Error is too big!")
            it += 1
        self.global_iters, self.err = it, err
        out = g.cell == NODE_OUT
        self.next[0][out] = 0; self.next[1][out] = 0; self.next[2][out] = F(g.startT)
        self._copy_all(self.next, self.cur)

    def get_layer(self, odx, ody):
        ii = (np.arange(odx) * self.nx) // odx
        jj = (np.arange(ody) * self.ny) // ody
        return [a[np.ix_(ii, jj)].copy() for a in self.next]
