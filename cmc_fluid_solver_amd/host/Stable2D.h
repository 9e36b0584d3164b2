// The reference's 2D path, CPU only (BASELINE configs[0] "plumbing, no GPU"): the explicit
// advection/diffusion step followed by a pressure projection -- the only Poisson solve of the reference -- over a Grid2D.
// Restated from the algorithm of
//   StableSolver2D::{Init, SolveU, SolveV, Project, PrepareIndices, TimeStep}   (FluidSolver2D/StableSolver2D.cpp:21-234)
//   TimeLayer2D (central differences, EvalDivError, Copy*/Merge*)              (FluidSolver2D/TimeLayer2D.h:24-186)
//   Solver2D::{UpdateBoundaries, ClearOutterCells, GetLayer}                    (FluidSolver2D/Solver2D.cpp:21-84)
// in FTYPE = float where the reference computes in FTYPE and in double where it does (the Poisson update).  Deliberate readings of
// under-specified spots (the reference's 2D driver is Windows-only, FluidSolver2D.cpp:18-30):
//   * the convergence test of the Poisson sweep, err = max(|dq / q_new|, err), follows the MSVC `max` macro: a NaN (0/0: q_new = q = 0)
//     does not enter err; q_new = 0 with q != 0 gives inf and keeps the sweep going;
//   * "Exceeded max number of iterations" / "Error is too big!" throw instead of exit(1);
//   * bc_type NoSlip only (the free-slip rasteriser variant, Grid2D.cpp:117-153 with bc_strength, is not restated).
// Pinned to the reference (r3): tests/test_stable2d.py holds this solver (through fs2d_run) bit for bit to U, V, T and the err prints of
// the reference's own Grid2D + StableSolver2D objects (oracle/ref_harness_2d.cpp over the reference's translation units as they lie) on
// the authored cavity at 54^2 and 128^2 and on the 10-frame heart outline with moving walls (tests/golden/ref2d_*.npz) -- the three
// readings above are what the reference's Linux build does.  Python twin: cmc_fluid_solver_amd/stable2d.py.
#pragma once
#include <cmath>
#include <stdexcept>
#include <vector>

#include "Shape2D.h"

namespace fs3d {

struct Layer2D {
    int nx = 0, ny = 0;
    float hx = 0, hy = 0;
    std::vector<float> u, v, t;
    void Resize(int nx_, int ny_, float hx_, float hy_) { nx = nx_; ny = ny_; hx = hx_; hy = hy_; u.assign((size_t)nx * ny, 0); v = u; t = u; }
    size_t at(int i, int j) const { return (size_t)i * ny + j; }
    static float ddx(const std::vector<float> &f, size_t c, size_t s, float h) { return (f[c + s] - f[c - s]) / (2 * h); }
    static float d2(const std::vector<float> &f, size_t c, size_t s, float h) { return (f[c + s] - 2 * f[c] + f[c - s]) / (h * h); }
};

class Stable2D {
public:
    static constexpr double DIV_ERR_THRESHOLD = 0.1, POISSON_ERR_THRESHOLD = 1e-2;      // StableSolver2D.h:23-24
    static constexpr int MAX_GLOBAL_ITERS = 100;                                          // :25
    Grid2D *grid = nullptr;
    Layer2D cur, next, temp, next_w;
    std::vector<float> q, div;
    float v_vis = 0;
    double err = 0;
    long poisson_sweeps = 0;              // Gauss-Seidel sweeps of the last TimeStep (diagnostic)
    int global_iters = 0;

    void Init(Grid2D &g, float v_vis_)
    {
        grid = &g; v_vis = v_vis_;
        const int nx = g.dimx, ny = g.dimy;
        for (Layer2D *l : {&cur, &next, &temp, &next_w}) l->Resize(nx, ny, (float)g.dx, (float)g.dy);
        q.assign((size_t)nx * ny, 0); div = q;
        for (size_t c = 0; c < cur.u.size(); c++) { cur.u[c] = g.velx[c]; cur.v[c] = g.vely[c]; cur.t[c] = g.T[c]; }
        CopyAll(cur, next); CopyAll(cur, temp);
    }
    // Solver2D::UpdateBoundaries: the grid's boundary data into cur, cur's boundary cells into next
    void UpdateBoundaries()
    {
        const Grid2D &g = *grid;
        for (size_t c = 0; c < cur.u.size(); c++)
            if (g.cell[c] == NODE_BOUND || g.cell[c] == NODE_VALVE) { cur.u[c] = g.velx[c]; cur.v[c] = g.vely[c]; cur.t[c] = g.T[c]; }
        CopyType(cur, next, NODE_BOUND); CopyType(cur, next, NODE_VALVE);
    }
    void TimeStep(float dt, int num_global, int /*num_local*/)
    {
        const Grid2D &g = *grid;
        const int nx = g.dimx, ny = g.dimy;
        CopyAll(cur, temp);
        inner.clear(); bound.clear();                                                       // PrepareIndices: i outer, j inner
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++) {
                const uint8_t t = g.cell[cur.at(i, j)];
                const bool edge = i == 0 || j == 0 || i == nx - 1 || j == ny - 1;
                if (t == NODE_IN || t == NODE_BOUND || t == NODE_VALVE) {
                    // the stencils and the mirrored Poisson neighbours reach one cell further (the reference reads outside its arrays there)
                    if (edge) throw std::runtime_error("2D solver: a fluid or boundary cell lies on the edge of the grid");
                    (t == NODE_IN ? inner : bound).push_back(cur.at(i, j));
                }
            }
        poisson_sweeps = 0;
        err = DivError(next);
        int it;
        for (it = 0; it < num_global || err > DIV_ERR_THRESHOLD; it++) {
            CopyAll(cur, next_w);
            Advect(dt, cur.u, temp.u, next_w.u); Advect(dt, cur.v, temp.v, next_w.v);       // SolveU, SolveV
            Project(next_w, next);
            err = DivError(next);
            for (int i = 0; i + 1 < nx; i++)                                                // next->MergeAllto(grid, temp, NODE_IN)
                for (int j = 0; j + 1 < ny; j++) {
                    const size_t c = cur.at(i, j);
                    if (g.cell[c] == NODE_IN) { temp.u[c] = (temp.u[c] + next.u[c]) / 2; temp.v[c] = (temp.v[c] + next.v[c]) / 2; temp.t[c] = (temp.t[c] + next.t[c]) / 2; }
                }
            if (it > MAX_GLOBAL_ITERS) throw std::runtime_error("Exceeded max number of iterations");
            if (err > DIV_ERR_THRESHOLD * 10) throw std::runtime_error("Error is too big!");
        }
        global_iters = it;
        for (size_t c = 0; c < next.u.size(); c++)                                          // ClearOutterCells
            if (g.cell[c] == NODE_OUT) { next.u[c] = 0; next.v[c] = 0; next.t[c] = g.startT; }
        CopyAll(next, cur);
    }
    // Solver2D::GetLayer: nearest-neighbour pick from `next`
    void GetLayer(std::vector<float> &ou, std::vector<float> &ov, std::vector<double> &oT, int odx, int ody) const
    {
        ou.resize((size_t)odx * ody); ov.resize(ou.size()); oT.resize(ou.size());
        for (int i = 0; i < odx; i++)
            for (int j = 0; j < ody; j++) {
                const size_t c = next.at(i * next.nx / odx, j * next.ny / ody), o = (size_t)i * ody + j;
                ou[o] = next.u[c]; ov[o] = next.v[c]; oT[o] = next.t[c];
            }
    }

private:
    std::vector<size_t> inner, bound;
    // TimeLayer2D::Copy*to: cells of one type, the last row and column left out (TimeLayer2D.h:109-150)
    void CopyType(const Layer2D &a, Layer2D &b, uint8_t type) const
    {
        for (int i = 0; i + 1 < a.nx; i++)
            for (int j = 0; j + 1 < a.ny; j++) {
                const size_t c = a.at(i, j);
                if (grid->cell[c] == type) { b.u[c] = a.u[c]; b.v[c] = a.v[c]; b.t[c] = a.t[c]; }
            }
    }
    void CopyAll(const Layer2D &a, Layer2D &b) const { for (uint8_t t : {NODE_IN, NODE_OUT, NODE_BOUND, NODE_VALVE}) CopyType(a, b, t); }
    // SolveU / SolveV: f_new = f_cur + dt (-U f_x - V f_y + nu (f_xx + f_yy)) with the stencils on temp
    void Advect(float dt, const std::vector<float> &fc, const std::vector<float> &ft, std::vector<float> &fn) const
    {
        const size_t sx = (size_t)temp.ny, sy = 1;
        for (size_t c : inner) {
            const float fx = Layer2D::ddx(ft, c, sx, temp.hx), fy = Layer2D::ddx(ft, c, sy, temp.hy);
            const float lap = Layer2D::d2(ft, c, sx, temp.hx) + Layer2D::d2(ft, c, sy, temp.hy);
            fn[c] = fc[c] + dt * (-temp.u[c] * fx - temp.v[c] * fy + v_vis * lap);
        }
    }
    // TimeLayer2D::EvalDivError (TimeLayer2D.h:92-107), the reference's own (lopsided) cell formula
    double DivError(const Layer2D &l) const
    {
        const Grid2D &g = *grid;
        float e = 0;
        int count = 0;
        const size_t sx = (size_t)l.ny;
        for (int i = 0; i + 1 < l.nx; i++)
            for (int j = 0; j + 1 < l.ny; j++) {
                const size_t c = l.at(i, j);
                if (g.cell[c] == NODE_IN && g.cell[c + sx] == NODE_IN && g.cell[c + 1] == NODE_IN && g.cell[c + sx + 1] == NODE_IN) {
                    const float tx = l.hy * (l.u[c + sx] - l.u[c]) + (l.u[c + sx + 1] - l.u[c + 1]) / 2;
                    const float ty = l.hx * (l.v[c + 1] - l.v[c]) + (l.v[c + sx + 1] - l.v[c + sx]) / 2;
                    e += std::fabs(tx + ty);
                    count++;
                }
            }
        return e / count;
    }
    // Project: div = U_x + V_y; Gauss-Seidel on q_xx + q_yy = div (boundary cells first, Neumann by mirroring; then the inner cells),
    // until the largest relative change of a sweep is below the threshold; proj = w - grad q
    void Project(const Layer2D &w, Layer2D &proj)
    {
        const Grid2D &g = *grid;
        const size_t sx = (size_t)w.ny, sy = 1;
        std::fill(div.begin(), div.end(), 0.0f);
        for (size_t c : inner) div[c] = Layer2D::ddx(w.u, c, sx, w.hx) + Layer2D::ddx(w.v, c, sy, w.hy);
        const double dx2 = g.dx * g.dx, dy2 = g.dy * g.dy, rcp = 0.5 / (dx2 + dy2);
        std::fill(q.begin(), q.end(), 0.0f);
        double e;
        do {
            e = 0.0;
            auto relax = [&](size_t c, double i0, double i1, double j0, double j1) {
                const double qn = rcp * ((i0 + i1) * dy2 + (j0 + j1) * dx2 - div[c] * dx2 * dy2);
                const double ce = std::fabs((qn - q[c]) / qn);
                if (ce > e) e = ce;                                                         // a NaN does not enter (see the header)
                q[c] = (float)qn;
            };
            for (size_t c : bound) {
                const double i0 = g.cell[c - sx] == NODE_IN ? q[c - sx] : q[c + sx], i1 = g.cell[c + sx] == NODE_IN ? q[c + sx] : q[c - sx];
                const double j0 = g.cell[c - sy] == NODE_IN ? q[c - sy] : q[c + sy], j1 = g.cell[c + sy] == NODE_IN ? q[c + sy] : q[c - sy];
                relax(c, i0, i1, j0, j1);
            }
            for (size_t c : inner) relax(c, q[c - sx], q[c + sx], q[c - sy], q[c + sy]);
            poisson_sweeps++;
            if (poisson_sweeps > 10000000) throw std::runtime_error("Poisson sweep does not converge");
        } while (e >= POISSON_ERR_THRESHOLD);
        for (size_t c : inner) {
            proj.u[c] = w.u[c] - Layer2D::ddx(q, c, sx, w.hx);
            proj.v[c] = w.v[c] - Layer2D::ddx(q, c, sy, w.hy);
        }
    }
};

}  // namespace fs3d
