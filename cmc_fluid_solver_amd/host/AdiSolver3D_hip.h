// C++ host side above the C ABI (include/fs3d.h): the reference's solver interface for this path,
// re-shaped around libfs3d_hip.so.  A driver written against the reference's Solver3D
// (Solver3D.h:24-49) / AdiSolver3D (AdiSolver3D.h:61-70) calls the same methods with the same
// argument meaning: Init, UpdateBoundaries, TimeStep, GetLayer; failures surface as
// std::runtime_error exactly where the reference throws (gpuSafeCall, GPUplan.cpp:173-193;
// "Error is too big!", AdiSolver3D.cpp:371-374).
//
// Header-only; link with -lfs3d_hip.  No CPU fallback exists: without the library or a GPU
// every call throws.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fs3d.h"

namespace fs3d {

// Geometry.h:29-43
enum NodeType : uint8_t { NODE_IN = 0, NODE_OUT = 1, NODE_BOUND = 2, NODE_VALVE = 3 };
enum BCtype : uint8_t { BC_NOSLIP = 0, BC_FREE = 1 };

// Geometry.h:538-562
template <typename FTYPE>
struct FluidParams {
    FTYPE v_T, v_vis, t_vis, t_phi;
    FluidParams() : v_T(1), v_vis(0), t_vis(0), t_phi(0) {}
    FluidParams(double Re, double Pr, double lambda)
        : v_T(1.0), v_vis((FTYPE)(1.0 / Re)), t_vis((FTYPE)(1.0 / (Re * Pr))), t_phi((FTYPE)((lambda - 1) / (lambda * Re))) {}
    FluidParams(double vis, double rho, double R, double k, double cv)
        : v_T((FTYPE)R), v_vis((FTYPE)(vis / rho)), t_vis((FTYPE)(k / (rho * cv))), t_phi((FTYPE)(vis / (rho * cv))) {}
};

// The part of Grid3D (Grid3D.h:95-200) the solver consumes: dims, spacing and the Node array
// (type, bc_vel, bc_temp, v, T; Grid3D.h:73-88) as structure-of-arrays.
template <typename FTYPE>
struct Grid3D {
    int dimx = 0, dimy = 0, dimz = 0;
    double dx = 0, dy = 0, dz = 0, baseT = 0;
    std::vector<uint8_t> type, bc_vel, bc_temp;
    std::vector<FTYPE> vx, vy, vz, T;
    void Resize(int nx, int ny, int nz)
    {
        dimx = nx; dimy = ny; dimz = nz;
        const size_t n = (size_t)nx * ny * nz;
        type.assign(n, NODE_OUT); bc_vel.assign(n, BC_NOSLIP); bc_temp.assign(n, BC_NOSLIP);
        vx.assign(n, 0); vy.assign(n, 0); vz.assign(n, 0); T.assign(n, 0);
    }
    size_t Index(int i, int j, int k) const { return ((size_t)i * dimy + j) * dimz + k; }   // TimeLayer3D.h:256-259
    NodeType GetType(int i, int j, int k) const { return (NodeType)type[Index(i, j, k)]; }
    // Node::SetBound, Grid3D.h:80-87
    void SetBound(int i, int j, int k, BCtype bv, BCtype bt, FTYPE ux, FTYPE uy, FTYPE uz, FTYPE t, NodeType nt = NODE_BOUND)
    {
        const size_t id = Index(i, j, k);
        type[id] = nt; bc_vel[id] = bv; bc_temp[id] = bt; vx[id] = ux; vy[id] = uy; vz[id] = uz; T[id] = t;
    }
};

template <typename FTYPE>
class AdiSolver3D {
public:
    AdiSolver3D() = default;
    AdiSolver3D(const AdiSolver3D &) = delete;
    AdiSolver3D &operator=(const AdiSolver3D &) = delete;
    ~AdiSolver3D() { if (ctx_) fs3d_destroy(ctx_); }

    // AdiSolver3D::Init (AdiSolver3D.cpp:166-268) + CreateSegments (:553-562) + the cur layer's
    // CopyFromGrid constructor (TimeLayer3D.h:1076-1090).  x0/x1 select this process's x-slab.
    void Init(int device, const Grid3D<FTYPE> &grid, const FluidParams<FTYPE> &params, int x0 = 0, int x1 = -1)
    {
        if (x1 < 0) x1 = grid.dimx;
        grid_ = &grid;
        const fs3d_precision prec = sizeof(FTYPE) == 4 ? FS3D_F32 : FS3D_F64;
        fs3d_status st = fs3d_create(&ctx_, device, prec, x1 - x0, grid.dimy, grid.dimz, grid.dx, grid.dy, grid.dz, x0, grid.dimx);
        if (st != FS3D_OK) throw std::runtime_error(std::string("fs3d_create: ") + fs3d_last_error(nullptr));
        chk(fs3d_set_params(ctx_, params.v_T, params.v_vis, params.t_vis, params.t_phi));
        chk(fs3d_upload_nodes(ctx_, grid.type.data(), grid.bc_vel.data(), grid.bc_temp.data(), grid.vx.data(), grid.vy.data(),
                              grid.vz.data(), grid.T.data(), numSegs));
        chk(fs3d_init_layers_from_nodes(ctx_));
        dimx = x1 - x0; dimy = grid.dimy; dimz = grid.dimz;
    }
    // multi-GPU: join the RCCL group (one process per GPU); id = 128-byte ncclUniqueId from UniqueId() on rank 0
    static void UniqueId(void *id128) { if (fs3d_comm_unique_id(id128) != FS3D_OK) throw std::runtime_error("fs3d_comm_unique_id failed"); }
    void JoinGroup(const void *id128, int rank, int nranks) { chk(fs3d_comm_init(ctx_, id128, rank, nranks)); }
    // multi-GPU, one process (the reference's GPUplan mode, "GPU n"): one solver per slab, each driven by its own host thread,
    // joined by an in-process group; every collective call (TimeStep, ...) is then made by all the threads
    static void *CreateLocalGroup(int nranks) { void *g = nullptr; if (fs3d_local_group_create(nranks, &g) != FS3D_OK) throw std::runtime_error("fs3d_local_group_create failed"); return g; }
    static void DestroyLocalGroup(void *g) { fs3d_local_group_destroy(g); }
    void JoinLocalGroup(void *group, int rank) { chk(fs3d_comm_init_local(ctx_, group, rank)); }

    void UpdateBoundaries() { chk(fs3d_update_boundaries(ctx_)); }                        // AdiSolver3D.cpp:286-304
    // AdiSolver3D::TimeStep (AdiSolver3D.cpp:306-391); throws where the reference throws
    void TimeStep(FTYPE dt, int num_global, int num_local, bool computeError)
    {
        fs3d_status st = fs3d_time_step(ctx_, dt, num_global, num_local, computeError ? 1 : 0, &diffError);
        if (st != FS3D_OK) throw std::runtime_error(fs3d_last_error(ctx_));
    }
    // Solver3D::GetLayer (Solver3D.cpp:21-25): v = interleaved x,y,z (Vec3D), T = double
    void GetLayer(FTYPE *v, double *T, int outdimx = 0, int outdimy = 0, int outdimz = 0) { chk(fs3d_get_layer(ctx_, v, T, outdimx, outdimy, outdimz)); }
    // cur layer on the host (ScalarField3D(CPU, field) copy constructor, TimeLayer3D.h:358-383)
    void DownloadCur(FTYPE *u, FTYPE *v, FTYPE *w, FTYPE *T) { chk(fs3d_download_layer(ctx_, FS3D_LAYER_CUR, u, v, w, T)); }
    // device time per event class since EnableTiming(true): 0 sweeps Z, 1 sweeps Y, 2 sweeps X, 3 everything else
    void EnableTiming(bool on) { chk(fs3d_enable_timing(ctx_, on ? 1 : 0)); }
    void Timings(float ms[4], int count[4]) { chk(fs3d_last_step_timing(ctx_, ms, count)); }
    // the same under the reference Profiler's event names (Common/Profiler.h; AdiSolver3D.cpp:297-367, 555-680)
    void ProfilerEvents(const char *names[FS3D_N_EVENTS], float ms[FS3D_N_EVENTS], int count[FS3D_N_EVENTS]) { chk(fs3d_profiler_events(ctx_, names, ms, count)); }
    double EvalDivError() { double e = 0; chk(fs3d_eval_div_error(ctx_, FS3D_LAYER_NEXT, &e, nullptr)); return e; }

    fs3d_ctx *ctx() const { return ctx_; }              // for C-ABI calls the class does not wrap

    double diffError = 0.0;
    int numSegs[3] = {0, 0, 0};
    int dimx = 0, dimy = 0, dimz = 0;

private:
    void chk(fs3d_status st) { if (st != FS3D_OK) throw std::runtime_error(fs3d_last_error(ctx_)); }
    fs3d_ctx *ctx_ = nullptr;
    const Grid3D<FTYPE> *grid_ = nullptr;
};

}  // namespace fs3d
