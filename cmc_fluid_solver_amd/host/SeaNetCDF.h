// SeaNetCDF input surface in C++: the reference's `in_fmt SeaNetCDF` depth map (a netCDF-4 file: _lat_subset, _lon_subset, z)
// turned into the Node array the solver consumes.  Operation for operation:
//   Grid3D::LoadNetCDF                (FluidSolver3D/Grid3D.cpp:433-486): latitude -> x, longitude -> y, depth -> z (z < 0: sea)
//   Grid3D::Init                      (Grid3D.cpp:351-371)
//   Grid3D::Prepare3D_NetCDF          (Grid3D.cpp:968-1075): sea cells, a NODE_BOUND skin on both sides of the sea bed, in/out
//                                     streams (NODE_VALVE, +-bc_initv, T = bc_initT / 2 - bc_initT) on the faces y = max and x = max
//   DepthInfo3D                       (Common/Geometry.h:418-447): the depth map and its nearest-neighbour resampling (`d` output)
// The file is read with host/Hdf5Min.h (no libnetcdf here).  Python twin: cmc_fluid_solver_amd/seanetcdf.py.
// Parity unpinned: the survey could not run the reference on this input (it needs a real libnetcdf).
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "AdiSolver3D_hip.h"
#include "Hdf5Min.h"
#include "Shape2D.h"

namespace fs3d {

struct DepthInfo3D {
    int dimx = 0, dimy = 0;
    std::vector<float> depth;                              // [dimx][dimy]
    DepthInfo3D() {}
    DepthInfo3D(int nx, int ny, const DepthInfo3D &info) : dimx(nx), dimy(ny), depth((size_t)nx * ny)     // "simple filter", Geometry.h:429-441
    {
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++) depth[j + (size_t)i * dimy] = info.depth[(size_t)(j * info.dimy / ny) + (size_t)info.dimy * (i * info.dimx / nx)];
    }
};

struct SeaNetCDF {
    DepthInfo3D depths;
    float bbox[6] = {0, 0, 0, 0, 0, 0};                    // pMin.xyz, pMax.xyz
    int dimx = 0, dimy = 0, dimz = 0;

    template <typename FTYPE>
    void Load(Grid3D<FTYPE> &g, const std::string &path, double dx, double dy, double dz, double baseT, const double bcInVel[3], double bcInT, bool align)
    {
        const Hdf5File f(path);
        const std::vector<double> lats = f.Read("_lat_subset"), lons = f.Read("_lon_subset"), z = f.Read("z");
        const int nx = (int)lats.size(), ny = (int)lons.size();
        if ((size_t)nx * ny != z.size()) throw std::runtime_error("SeaNetCDF: z is not [_lat_subset][_lon_subset]");
        depths.dimx = nx; depths.dimy = ny; depths.depth.resize(z.size());
        for (size_t c = 0; c < z.size(); c++) depths.depth[c] = (float)z[c];
        // bbox: the corner coordinates, the deepest point, one cell of slack below it (Grid3D.cpp:466-476)
        const float INF = 1e10f;
        float mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
        const float pts[2][3] = {{(float)lats[0], (float)lons[0], 0.0f}, {(float)lats[nx - 1], (float)lons[ny - 1], 0.0f}};
        for (auto &p : pts) for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
        for (float v : depths.depth) if (v < mn[2]) mn[2] = v;
        mn[2] -= (float)dz;
        for (int a = 0; a < 3; a++) { bbox[a] = mn[a]; bbox[3 + a] = mx[a]; }
        // Grid3D::Init
        dimx = (int)std::ceil((float)(mx[0] - mn[0]) / dx) + 1;
        dimy = (int)std::ceil((float)(mx[1] - mn[1]) / dy) + 1;
        dimz = (int)std::ceil((float)(mx[2] - mn[2]) / dz) + 1;
        for (int a = 0; a < 3; a++) {
            const double h = a == 0 ? dx : (a == 1 ? dy : dz);
            if (!(mx[a] >= mn[a]) || !(h > 0) || (double)(mx[a] - mn[a]) / h > 65536.0) throw std::runtime_error("SeaNetCDF: the depth map does not give a grid of a sensible size");
        }
        if (align) { dimx = AlignBy32(dimx); dimy = AlignBy32(dimy); dimz = AlignBy32(dimz); }
        if ((double)dimx * dimy * dimz >= 2147483648.0) throw std::runtime_error("SeaNetCDF: grid of more than 2^31 cells");
        g.Resize(dimx, dimy, dimz);
        g.dx = dx; g.dy = dy; g.dz = dz; g.baseT = baseT;
        // Grid3D::Prepare3D_NetCDF
        const FTYPE bT = (FTYPE)(float)baseT;
        for (size_t c = 0; c < g.type.size(); c++) { g.type[c] = NODE_OUT; g.bc_vel[c] = BC_NOSLIP; g.bc_temp[c] = BC_NOSLIP; g.vx[c] = g.vy[c] = g.vz[c] = 0; g.T[c] = bT; }
        for (int i = 0; i < dimx; i++)
            for (int j = 0; j < dimy; j++) {
                const int di = i * depths.dimx / dimx, dj = j * depths.dimy / dimy;
                const float zz = depths.depth[dj + (size_t)di * depths.dimy];
                if (zz < 0.0f) {
                    const int bound_k = (int)(dimz * zz / mn[2]);
                    for (int k = 1; k < bound_k; k++) g.type[g.Index(i, j, k)] = NODE_IN;
                }
            }
        auto touches = [&](int i, int j, int k, uint8_t t) {
            return g.type[g.Index(i - 1, j, k)] == t || g.type[g.Index(i + 1, j, k)] == t || g.type[g.Index(i, j - 1, k)] == t ||
                   g.type[g.Index(i, j + 1, k)] == t || g.type[g.Index(i, j, k - 1)] == t || g.type[g.Index(i, j, k + 1)] == t;
        };
        const FTYPE bTf = (FTYPE)(float)baseT;
        for (int i = 1; i < dimx - 1; i++)                       // sea cells next to NODE_OUT become the bound (in sweep order: a cell turned
            for (int j = 1; j < dimy - 1; j++)                   //  NODE_BOUND no longer counts as NODE_OUT for the cells after it -- it never did)
                for (int k = 1; k < dimz - 1; k++)
                    if (g.type[g.Index(i, j, k)] == NODE_IN && touches(i, j, k, NODE_OUT)) g.SetBound(i, j, k, BC_NOSLIP, BC_NOSLIP, 0, 0, 0, bTf);
        std::vector<size_t> idx;
        for (int i = 1; i < dimx - 1; i++)                       // ... and the NODE_OUT cells next to that bound, collected first, set afterwards
            for (int j = 1; j < dimy - 1; j++)
                for (int k = 1; k < dimz - 1; k++)
                    if (g.type[g.Index(i, j, k)] == NODE_OUT && touches(i, j, k, NODE_BOUND)) idx.push_back(g.Index(i, j, k));
        for (size_t c : idx) { g.type[c] = NODE_BOUND; g.bc_vel[c] = BC_NOSLIP; g.bc_temp[c] = BC_NOSLIP; g.vx[c] = g.vy[c] = g.vz[c] = 0; g.T[c] = bTf; }
        // in / out streams on the faces y = dimy-1 and x = dimx-1: the upper half of the water column flows in, the lower half out
        const float vin[3] = {(float)bcInVel[0], (float)bcInVel[1], (float)bcInVel[2]};
        auto stream = [&](int i, int j) {
            int start = -1, end = 0;
            for (int k = 0; k < dimz; k++) if (g.type[g.Index(i, j, k)] == NODE_IN) { if (start < 0) start = k; end = k; }
            for (int k = 0; k < dimz; k++)
                if (g.type[g.Index(i, j, k)] == NODE_IN) {
                    const size_t c = g.Index(i, j, k);
                    const bool in = k < (start + end) / 2;
                    g.type[c] = NODE_VALVE; g.bc_vel[c] = BC_NOSLIP; g.bc_temp[c] = BC_NOSLIP;
                    g.vx[c] = in ? vin[0] : 0.0f - vin[0]; g.vy[c] = in ? vin[1] : 0.0f - vin[1]; g.vz[c] = in ? vin[2] : 0.0f - vin[2];
                    g.T[c] = in ? (FTYPE)(float)bcInT : (FTYPE)(2.0f - (float)bcInT);
                }
        };
        for (int i = 0; i < dimx; i++) stream(i, dimy - 1);
        for (int j = 0; j < dimy; j++) stream(dimx - 1, j);
    }
};

}  // namespace fs3d
