// Test program for the C++ host class: steps a small box with fs3d::AdiSolver3D (HIP) and with the CPU
// oracle, compares the fields value for value.  Built and run by tests/test_host_cpp.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../cmc_fluid_solver_amd/host/AdiSolver3D_hip.h"

extern "C" {
void *fs3d_oracle_create_f32(int, int, int, double, double, double);
void fs3d_oracle_destroy_f32(void *);
void fs3d_oracle_set_params_f32(void *, double, double, double, double);
void fs3d_oracle_set_nodes_f32(void *, const uint8_t *, const uint8_t *, const uint8_t *, const float *, const float *, const float *, const float *);
void fs3d_oracle_create_segments_f32(void *);
void fs3d_oracle_init_layers_f32(void *);
void fs3d_oracle_update_boundaries_f32(void *);
int fs3d_oracle_time_step_f32(void *, double, int, int, int, double *);
void fs3d_oracle_get_field_f32(void *, int, int, float *);
}

int main()
{
    using namespace fs3d;
    const int nx = 20, ny = 18, nz = 24;
    Grid3D<float> g;
    g.Resize(nx, ny, nz);
    g.dx = g.dy = g.dz = 0.04; g.baseT = 1.0;
    for (int i = 0; i < nx; i++) for (int j = 0; j < ny; j++) for (int k = 0; k < nz; k++) {
        const bool shell = i == 0 || j == 0 || k == 0 || i == nx - 1 || j == ny - 1 || k == nz - 1;
        const size_t id = g.Index(i, j, k);
        if (!shell) { g.type[id] = NODE_IN; g.T[id] = 1.0f; continue; }
        const bool face_in = i == 0 && j > 0 && j < ny - 1 && k > 0 && k < nz - 1;
        const bool face_out = i == nx - 1 && j > 0 && j < ny - 1 && k > 0 && k < nz - 1;
        if (face_in) g.SetBound(i, j, k, BC_NOSLIP, BC_NOSLIP, 1.0f, 0, 0, 1.0f, NODE_VALVE);
        else if (face_out) g.SetBound(i, j, k, BC_FREE, BC_FREE, 0, 0, 0, 1.0f, NODE_VALVE);
        else g.SetBound(i, j, k, BC_NOSLIP, BC_FREE, 0, 0, 0, 1.0f);
    }
    FluidParams<float> params(200.0, 0.72, 1.4);
    try {
        AdiSolver3D<float> solver;
        solver.Init(0, g, params);
        void *o = fs3d_oracle_create_f32(nx, ny, nz, g.dx, g.dy, g.dz);
        fs3d_oracle_set_params_f32(o, params.v_T, params.v_vis, params.t_vis, params.t_phi);
        fs3d_oracle_set_nodes_f32(o, g.type.data(), g.bc_vel.data(), g.bc_temp.data(), g.vx.data(), g.vy.data(), g.vz.data(), g.T.data());
        fs3d_oracle_create_segments_f32(o);
        fs3d_oracle_init_layers_f32(o);
        for (int step = 0; step < 3; step++) {
            solver.UpdateBoundaries();
            solver.TimeStep(0.1f, 4, 2, true);
            double eo = 0;
            fs3d_oracle_update_boundaries_f32(o);
            if (fs3d_oracle_time_step_f32(o, 0.1, 4, 2, 1, &eo)) { printf("oracle diverged\n"); return 2; }
            if (std::fabs(solver.diffError - eo) > 1e-12 * std::fabs(eo)) { printf("diffError mismatch %g %g\n", solver.diffError, eo); return 3; }
        }
        const size_t n = (size_t)nx * ny * nz;
        std::vector<float> f[4], r(n);
        for (auto &v : f) v.resize(n);
        solver.DownloadCur(f[0].data(), f[1].data(), f[2].data(), f[3].data());
        for (int v = 0; v < 4; v++) {
            fs3d_oracle_get_field_f32(o, 0, v, r.data());
            for (size_t i = 0; i < n; i++) if (!(f[v][i] == r[i])) { printf("field %d differs at %zu: %g vs %g\n", v, i, f[v][i], r[i]); return 4; }
        }
        // error path: a bad device ordinal must throw, not fall back
        bool threw = false;
        try { AdiSolver3D<float> bad; bad.Init(99, g, params); } catch (const std::runtime_error &) { threw = true; }
        if (!threw) { printf("bad device did not throw\n"); return 5; }
        fs3d_oracle_destroy_f32(o);
    } catch (const std::exception &e) {
        printf("exception: %s\n", e.what());
        return 1;
    }
    printf("HOST_CPP_OK\n");
    return 0;
}
