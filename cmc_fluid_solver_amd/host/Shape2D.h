// Shape2D input surface in C++: the reference's `in_fmt Shape2D` text geometry extruded in depth into the Node
// array the solver consumes.  Operation for operation (FTYPE = float where the reference computes in FTYPE):
//   Grid2D::LoadFromFile / ComputeBorderVelocities / Init / Build / RasterLine / FloodFill
//                                       (FluidSolver2D/Grid2D.cpp:109-372, 376-396)
//   BBox2D::Build                       (Common/Geometry.h:455-486)
//   Grid3D::LoadFromFile / Prepare2D    (FluidSolver3D/Grid3D.cpp:488-513, 608-668)
// Multi-frame inputs (moving walls, data/3D/large_tests/heart_us): all frames are read, border velocities come from the move
// between consecutive frames, the bounding box covers all frames, Prepare(time) interpolates a sub-frame.  The reference's 3D
// time loop prepares the grid once, at time 0 (`grid->Prepare(t)` is commented out, FluidSolver3D.cpp:237).
// Python twin with the same pins (grid dims, NODE_IN counts of SURVEY.md): cmc_fluid_solver_amd/shape2d.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "AdiSolver3D_hip.h"

namespace fs3d {

inline int AlignBy32(int n) { return (n & 31) == 0 ? n : (((n >> 5) + 1) << 5); }   // Geometry.h:564-568

struct Shape2DFrame {
    struct Shape { std::vector<float> px, py, vx, vy, gx, gy; bool active = false; };
    float duration = 0;
    std::vector<Shape> shapes;
};

// FluidSolver2D::Grid2D (Grid2D.h:42-104)
struct Grid2D {
    static constexpr float GRID_SCALE_FACTOR = 0.001f, BBOX_PADDING = 0.02f, INF = 1e10f;   // Grid2D.h:31, Geometry.h:22-24
    int dimx = 0, dimy = 0;
    double dx = 0, dy = 0;
    float bbox[4] = {0, 0, 0, 0};        // pMin.x, pMin.y, pMax.x, pMax.y
    float duration = 0;
    std::vector<uint8_t> cell;           // [dimx][dimy] NodeType
    std::vector<float> velx, vely, T;

    static float num(std::string tok)    // ReadPoint2D (IO.h:511-540): ',' accepted as decimal mark; (FTYPE)atof
    {
        std::replace(tok.begin(), tok.end(), ',', '.');
        return (float)std::atof(tok.c_str());
    }

    std::vector<Shape2DFrame> frames;    // every frame of the file; gx/gy = grid coordinates (Init)
    float startT = 0;

    void Load(const std::string &path, double dx_, double dy_, double startT_, bool align, double time = 0.0)
    {
        std::ifstream in(path.c_str());
        if (!in) throw std::runtime_error("cannot open Shape2D file " + path);
        std::vector<std::string> t;
        for (std::string w; in >> w;) t.push_back(w);
        size_t i = 0;
        auto next = [&]() -> const std::string & { if (i >= t.size()) throw std::runtime_error("Shape2D file ends early"); return t[i++]; };
        const int num_frames = std::atoi(next().c_str());
        if (num_frames < 1 || (size_t)num_frames > t.size()) throw std::runtime_error("Shape2D file: bad number of frames");
        frames.assign(num_frames, Shape2DFrame());
        for (auto &fr : frames) {                                    // Grid2D.cpp:282-318
            fr.duration = num(next());
            const int nshapes = std::atoi(next().c_str());
            if (nshapes < 0 || (size_t)nshapes > t.size()) throw std::runtime_error("Shape2D file: bad number of shapes");
            for (int s = 0; s < nshapes; s++) {
                Shape2DFrame::Shape sh;
                const int npts = std::atoi(next().c_str());
                if (npts < 0 || (size_t)npts > t.size()) throw std::runtime_error("Shape2D file: bad number of points");
                for (int p = 0; p < npts; p++) {
                    const float x = num(next()), y = num(next());
                    sh.px.push_back(x * GRID_SCALE_FACTOR); sh.py.push_back(y * GRID_SCALE_FACTOR);
                }
                sh.active = next()[0] == 'M';                        // "Motion vx vy" | "Passive"
                float vx = 0, vy = 0;
                if (sh.active) { vx = num(next()); vy = num(next()); }
                sh.vx.assign(npts, vx * GRID_SCALE_FACTOR);
                sh.vy.assign(npts, vy * GRID_SCALE_FACTOR);
                fr.shapes.push_back(sh);
            }
        }
        for (size_t f = 1; f < frames.size(); f++) {
            if (frames[f].shapes.size() != frames[0].shapes.size()) throw std::runtime_error("Shape2D: frames differ in their number of shapes");
            for (size_t s = 0; s < frames[f].shapes.size(); s++)
                if (frames[f].shapes[s].px.size() != frames[0].shapes[s].px.size()) throw std::runtime_error("Shape2D: a shape changes its number of points between frames");
        }
        // ComputeBorderVelocities(j) for every frame, in order (Grid2D.cpp:365-366, 375-396): the velocities of frame j+1 from the
        // move j -> j+1.  One frame: next == frame, every difference is 0: passive shapes at rest, active ones keep theirs.
        for (size_t j = 0; j < frames.size(); j++) {
            Shape2DFrame &a = frames[j], &b = frames[(j + 1) % frames.size()];
            const float m = (float)(1 / (double)a.duration);
            for (size_t s = 0; s < a.shapes.size(); s++) {
                const auto &sa = a.shapes[s];
                auto &sb = b.shapes[s];
                for (size_t p = 0; p < sa.px.size(); p++)
                    if (!sa.active) { sb.vx[p] = (float)(sb.px[p] - sa.px[p]) * m; sb.vy[p] = (float)(sb.py[p] - sa.py[p]) * m; }
                    else { sb.vx[p] = mix(sb.vx[p], 1.0f, sa.px[p] - sb.px[p], m); sb.vy[p] = mix(sb.vy[p], 1.0f, sa.py[p] - sb.py[p], m); }
            }
        }
        dx = dx_; dy = dy_; duration = frames[0].duration; startT = (float)startT_;
        // BBox2D::Build over the points of all frames (Geometry.h:455-486)
        float pminx = INF, pminy = INF, pmaxx = -INF, pmaxy = -INF;
        for (auto &fr : frames)
            for (auto &sh : fr.shapes)
                for (size_t p = 0; p < sh.px.size(); p++) {
                    pminx = std::min(pminx, sh.px[p]); pminy = std::min(pminy, sh.py[p]);
                    pmaxx = std::max(pmaxx, sh.px[p]); pmaxy = std::max(pmaxy, sh.py[p]);
                }
        const float wx = pmaxx - pminx, wy = pmaxy - pminy;
        pminx = pminx - wx * BBOX_PADDING; pminy = pminy - wy * BBOX_PADDING;
        pmaxx = pmaxx + wx * BBOX_PADDING; pmaxy = pmaxy + wy * BBOX_PADDING;
        bbox[0] = pminx; bbox[1] = pminy; bbox[2] = pmaxx; bbox[3] = pmaxy;
        // (a corrupt coordinate or a tiny grid step: refuse before anything of that size is allocated or rasterised)
        if (!(pmaxx >= pminx) || !(pmaxy >= pminy) || !(dx > 0) || !(dy > 0) || (double)(pmaxx - pminx) / dx > 65536.0 || (double)(pmaxy - pminy) / dy > 65536.0)
            throw std::runtime_error("Shape2D: the shapes do not give a grid of a sensible size (more than 65536 cells along an axis, or no points)");
        // Grid2D::Init (Grid2D.cpp:212-246)
        dimx = (int)std::ceil((double)(float)(pmaxx - pminx) / dx) + 1;
        dimy = (int)std::ceil((double)(float)(pmaxy - pminy) / dy) + 1;
        if (align) { dimx = AlignBy32(dimx); dimy = AlignBy32(dimy); }
        if ((double)dimx * dimy > 268435456.0) throw std::runtime_error("Shape2D: more than 2^28 cells in the plane");
        const float fdx = (float)dx, fdy = (float)dy;
        for (auto &fr : frames)
            for (auto &sh : fr.shapes) {
                sh.gx.resize(sh.px.size()); sh.gy.resize(sh.px.size());
                for (size_t p = 0; p < sh.px.size(); p++) { sh.gx[p] = (float)(sh.px[p] - pminx) / fdx; sh.gy[p] = (float)(sh.py[p] - pminy) / fdy; }
            }
        Prepare(time);
    }

    // ---- frames in time (Grid2D.cpp:447-519)
    int GetFramesNum() const { return (int)frames.size(); }
    double GetCycleLenght() const { double r = 0; for (auto &fr : frames) r += fr.duration; return r; }
    int GetFrame(double time) const { double r, a0, a1; return Locate(time, r, a0, a1); }
    float GetLayerTime(double time) const { double r, a0, a1; Locate(time, r, a0, a1); return (float)(a1 - r); }

    // Grid2D::Prepare(time) -> ComputeSubframe(frame, substep) -> Build (Grid2D.cpp:398-461)
    void Prepare(double time)
    {
        double r, a0, a1;
        const int frame = Locate(time, r, a0, a1);
        const double substep = (r - a0) / (a1 - a0), isubstep = 1 - substep;
        const Shape2DFrame &f0 = frames[frame], &f1 = frames[(frame + 1) % frames.size()];
        const float s = (float)substep, is = (float)isubstep;
        Shape2DFrame sub;
        for (size_t q = 0; q < f0.shapes.size(); q++) {
            const auto &sa = f0.shapes[q], &sb = f1.shapes[q];
            Shape2DFrame::Shape sh;
            sh.active = sa.active;
            const size_t n = sa.px.size();
            sh.gx.resize(n); sh.gy.resize(n); sh.vx.resize(n); sh.vy.resize(n);
            for (size_t p = 0; p < n; p++) {
                sh.gx[p] = mix(sa.gx[p], is, sb.gx[p], s); sh.gy[p] = mix(sa.gy[p], is, sb.gy[p], s);
                sh.vx[p] = mix(sa.vx[p], is, sb.vx[p], s); sh.vy[p] = mix(sa.vy[p], is, sb.vy[p], s);
            }
            sub.shapes.push_back(sh);
        }
        Build(sub, startT);
    }

private:
    // a*wa + b*wb with each product and the sum rounded to float (no contraction into an fma)
    static float mix(float a, float wa, float b, float wb) { volatile float x = a * wa, y = b * wb; return x + y; }
    int Locate(double time, double &r, double &a0, double &a1) const
    {
        std::vector<double> a(frames.size() + 1, 0.0);
        for (size_t i = 1; i <= frames.size(); i++) a[i] = a[i - 1] + frames[i - 1].duration;
        r = std::fmod(time, a[frames.size()]);
        int frame = 0;
        for (size_t i = 1; i < frames.size(); i++) if (a[i] < r) frame = (int)i;
        a0 = a[frame]; a1 = a[frame + 1];
        return frame;
    }
    size_t id(int x, int y) const { return (size_t)x * dimy + y; }
    // Grid2D::RasterLine (Grid2D.cpp:117-153), bc_noslip == true (Grid3D.cpp:28)
    void RasterLine(float p1x, float p1y, float p2x, float p2y, float v1x, float v1y, float v2x, float v2y, uint8_t color, float startT)
    {
        const float ox = p2x - p1x, oy = p2y - p1y;
        const int steps = (int)std::max(std::fabs(ox), std::fabs(oy)) + 1;
        const float dpx = ox / (float)steps, dpy = oy / (float)steps;
        const float dvx = (float)(v2x - v1x) / (float)steps, dvy = (float)(v2y - v1y) / (float)steps;
        float px = p1x, py = p1y, vx = v1x, vy = v1y;
        for (int s = 0; s <= steps; s++) {
            const int x = (int)px, y = (int)py;
            if (x < 0 || y < 0 || x >= dimx || y >= dimy) throw std::runtime_error("Shape2D: shape point outside the grid");
            cell[id(x, y)] = color; velx[id(x, y)] = vx; vely[id(x, y)] = vy; T[id(x, y)] = startT;
            px = px + dpx; py = py + dpy; vx = vx + dvx; vy = vy + dvy;
        }
    }
    // Grid2D::Build (Grid2D.cpp:248-285) + FloodFill (:167-210)
    void Build(const Shape2DFrame &fr, float startT)
    {
        const size_t n = (size_t)dimx * dimy;
        cell.assign(n, NODE_IN); velx.assign(n, 0); vely.assign(n, 0); T.assign(n, 0);
        for (int pass = 0; pass < 2; pass++) {
            const bool active = pass == 0;
            const uint8_t color = active ? NODE_VALVE : NODE_BOUND;
            for (const auto &sh : fr.shapes) {
                if (sh.active != active) continue;
                for (size_t p = 0; p + 1 < sh.gx.size(); p++)
                    RasterLine(sh.gx[p], sh.gy[p], sh.gx[p + 1], sh.gy[p + 1], sh.vx[p], sh.vy[p], sh.vx[p + 1], sh.vy[p + 1], color, startT);
            }
        }
        std::vector<std::pair<int, int>> stack{{0, 0}};
        cell[id(0, 0)] = NODE_OUT;
        const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
        while (!stack.empty()) {
            const auto c = stack.back(); stack.pop_back();
            for (int q = 0; q < 4; q++) {
                const int a = c.first + di[q], b = c.second + dj[q];
                if (a >= 0 && a < dimx && b >= 0 && b < dimy && cell[id(a, b)] == NODE_IN) { cell[id(a, b)] = NODE_OUT; stack.push_back({a, b}); }
            }
        }
        for (size_t c = 0; c < n; c++)
            if (cell[c] == NODE_IN || cell[c] == NODE_OUT) { velx[c] = 0; vely[c] = 0; T[c] = startT; }
    }

public:
    uint8_t Cell(int x, int y) const { return cell[id(x, y)]; }
    float VelX(int x, int y) const { return velx[id(x, y)]; }
    float VelY(int x, int y) const { return vely[id(x, y)]; }
    float Temp(int x, int y) const { return T[id(x, y)]; }
};

// Grid3D(dx,dy,dz,depth,depth_var,baseT) + LoadFromFile + Prepare2D(0) (Grid3D.cpp:488-513, 608-668)
template <typename FTYPE>
void LoadShape2D(Grid3D<FTYPE> &g, Grid2D &g2, const std::string &path, double dx, double dy, double dz, double depth, double depth_var, double baseT, bool align)
{
    g2.Load(path, dx, dy, baseT, align);
    const int dimx = g2.dimx, dimy = g2.dimy;
    if (!(dz > 0) || !(depth >= 0) || depth / dz > 65536.0) throw std::runtime_error("Shape2D: depth / grid_dz gives more than 65536 cells");
    const int active_dimz = (int)std::ceil(depth / dz) + 1;                 // Grid3D.cpp:503-505
    const int dimz = align ? AlignBy32(active_dimz) : active_dimz;
    if ((double)dimx * dimy * dimz >= 2147483648.0) throw std::runtime_error("Shape2D: grid of more than 2^31 cells");
    g.Resize(dimx, dimy, dimz);
    g.dx = dx; g.dy = dy; g.dz = dz; g.baseT = baseT;
    // memset(nodes, 0): type NODE_IN, bc NOSLIP, v = 0, T = 0   (Grid3D.cpp:612)
    std::fill(g.type.begin(), g.type.end(), (uint8_t)NODE_IN);
    const int height = std::max(active_dimz - 2 - 2, 0);
    for (int i = 0; i < dimx; i++)
        for (int j = 0; j < dimy; j++) {
            const uint8_t c = g2.Cell(i, j);
            if (c == NODE_OUT) { for (int k = 0; k < dimz; k++) g.type[g.Index(i, j, k)] = NODE_OUT; continue; }
            for (int k = active_dimz - 1; k < dimz; k++) g.type[g.Index(i, j, k)] = NODE_OUT;                 // :626-628
            g.SetBound(i, j, active_dimz - 2, BC_NOSLIP, BC_FREE, 0, 0, 0, (FTYPE)(float)baseT);               // :629
            const double x = -1 + 2 * (double)i / dimx, y = -1 + 2 * (double)j / dimy;
            const double z = 1.0 - (x * x + y * y) * 0.5;
            const int bottom = 1 + (int)(depth_var * z * height);                                              // :632-636
            g.type[g.Index(i, j, 0)] = NODE_OUT;
            for (int k = 1; k <= bottom; k++) g.SetBound(i, j, k, BC_NOSLIP, BC_FREE, 0, 0, 0, (FTYPE)(float)baseT);   // :638-639
            for (int k = bottom + 1; k < active_dimz - 2; k++) {
                if (c == NODE_BOUND)                                                                           // :646-648
                    g.SetBound(i, j, k, BC_NOSLIP, BC_FREE, (FTYPE)g2.VelX(i, j), (FTYPE)g2.VelY(i, j), 0, (FTYPE)g2.Temp(i, j));
                else if (c == NODE_VALVE) {                                                                    // :649-655
                    const bool rest = g2.VelX(i, j) == 0 && g2.VelY(i, j) == 0;
                    g.SetBound(i, j, k, rest ? BC_FREE : BC_NOSLIP, rest ? BC_FREE : BC_NOSLIP, (FTYPE)g2.VelX(i, j), (FTYPE)g2.VelY(i, j), 0,
                               (FTYPE)g2.Temp(i, j), NODE_VALVE);
                } else {                                                                                       // NODE_IN, :656-659
                    g.type[g.Index(i, j, k)] = NODE_IN; g.T[g.Index(i, j, k)] = (FTYPE)(float)baseT;
                }
            }
        }
}

}  // namespace fs3d
