// Command-line driver: the reference's FluidSolver3D main (FluidSolver3D/FluidSolver3D.cpp:60-330) on top of
// libfs3d_hip.so.   fs3d_run <input data> <output prefix> <config> [align] [GPU [n]] [double] [--steps N] [--same-device] [--grid-only FILE]
//   * reads the config (host/Config.h) and a Shape2D, Shape3D or SeaNetCDF geometry (host/Shape2D.h, Shape3D.h, SeaNetCDF.h), prints the grid summary lines
//     the reference prints ("Grid = X x Y x Z", "NODE_IN points = ..."),
//   * runs the same loop: dt = cycle length / (frames * time_steps), UpdateBoundaries + TimeStep per step with the
//     divergence error every 10th step and on the last one, "err = ..." and the progress line per step,
//   * writes <output prefix>_res.nc through host/NetCDF3.h every out_time_steps steps (GetLayer).
// `transpose`, `decompose`, `blocking n` of the reference are accepted and ignored (backend tuning switches); `CSV`
// switches the closing timing table to the reference's comma-separated form.
// There is no CPU backend here: without a GPU the run stops with the library's error.
// --grid-only FILE: build the grid, dump it (dims, type, bc_vel, bc_temp, vx, vy, vz, T as raw arrays) and exit
//   without touching the GPU -- used by the CPU tests to compare the C++ loader with its Python twin.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "AdiSolver3D_hip.h"
#include "Config.h"
#include "NetCDF3.h"
#include "Shape2D.h"
#include "Shape3D.h"
#include "SeaNetCDF.h"
#include "GridImage.h"

// what the time loop and the result header need from the geometry (Grid3D::GetFramesNum / GetCycleLength / GetFrame / GetBBox)
struct RunGeom {
    int frames = 1;
    double length = 0;
    float bbox[6] = {0, 0, 0, 0, 0, 0};
    const fs3d::Grid2D *g2 = nullptr;                 // Shape2D: frames in time; Shape3D / SeaNetCDF: frame 0 throughout (Grid3D.cpp:322-328)
    const fs3d::DepthInfo3D *depths = nullptr;        // SeaNetCDF: the depth map (`d` output variable), x / y in degrees
    int GetFrame(double t) const { return g2 ? g2->GetFrame(t) : 0; }
};

template <typename FTYPE>
static int run_slabs(const fs3d::Grid3D<FTYPE> &grid, const RunGeom &geo, const std::string &prefix, const fs3d::Config &cfg, int nslabs,
                     bool same_device, long max_steps, bool csv);

template <typename FTYPE>
static int run(const std::string &data, const std::string &prefix, const fs3d::Config &cfg, bool align, int device, long max_steps, const std::string &grid_only, bool csv,
               int nslabs, bool same_device, bool grid_images)
{
    using namespace fs3d;
    Grid3D<FTYPE> grid;
    Grid2D g2;
    Shape3D sh3;
    RunGeom geo;
    SeaNetCDF sea;
    if (cfg.in_fmt == "SeaNetCDF") {
        std::printf("Geometry: depths from NetCDF\n");                                           // FluidSolver3D.cpp:133-138
        sea.Load(grid, data, cfg.dx, cfg.dy, cfg.dz, cfg.baseT, cfg.bc_inV, cfg.bc_inT, align);
        geo.frames = 1; geo.length = cfg.frame_time; geo.depths = &sea.depths;                   // num_frames = 1, Grid3D.cpp:483
        for (int a = 0; a < 6; a++) geo.bbox[a] = sea.bbox[a];
    } else if (cfg.in_fmt == "Shape3D") {
        std::printf("Geometry: 3D polygons\n");                                                  // FluidSolver3D.cpp:121-126
        LoadShape3D(grid, sh3, data, cfg.dx, cfg.dy, cfg.dz, cfg.baseT, align);
        geo.frames = sh3.GetFramesNum(); geo.length = cfg.frame_time;                            // Grid3D.cpp:298-309
        for (int a = 0; a < 6; a++) geo.bbox[a] = sh3.bbox[a];
    } else {
        std::printf("Geometry: extruded 2D shape\n");                                            // :127-132
        LoadShape2D(grid, g2, data, cfg.dx, cfg.dy, cfg.dz, cfg.depth, cfg.depth_var, cfg.baseT, align);
        geo.frames = g2.GetFramesNum(); geo.length = g2.GetCycleLenght(); geo.g2 = &g2;
        const float bb[6] = {g2.bbox[0], g2.bbox[1], 0.0f, g2.bbox[2], g2.bbox[3], (float)cfg.depth};   // BBox3D(bbox2D, depth), :203
        for (int a = 0; a < 6; a++) geo.bbox[a] = bb[a];
    }
    std::printf("Grid = %i x %i x %i\n", grid.dimx, grid.dimy, grid.dimz);                      // FluidSolver3D.cpp:146
    double inside = 0;
    for (uint8_t t : grid.type) inside += t == NODE_IN;
    std::printf("NODE_IN points = %f of total %f, volume = %f\n", inside, (double)grid.dimx * grid.dimy * grid.dimz,
                inside * grid.dx * grid.dy * grid.dz);                                          // :170
    if (grid_images) OutputGridImages(grid, prefix + "_grid_3d");                             // FluidSolver3D.cpp:152-153 (there: always)
    if (!grid_only.empty()) {
        FILE *f = std::fopen(grid_only.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + grid_only);
        const int hdr[4] = {grid.dimx, grid.dimy, grid.dimz, (int)sizeof(FTYPE)};
        std::fwrite(hdr, sizeof hdr, 1, f);
        std::fwrite(grid.type.data(), 1, grid.type.size(), f); std::fwrite(grid.bc_vel.data(), 1, grid.bc_vel.size(), f);
        std::fwrite(grid.bc_temp.data(), 1, grid.bc_temp.size(), f);
        for (const std::vector<FTYPE> *a : {&grid.vx, &grid.vy, &grid.vz, &grid.T}) std::fwrite(a->data(), sizeof(FTYPE), a->size(), f);
        std::fclose(f);
        return 0;
    }
    if (nslabs > 1) return run_slabs<FTYPE>(grid, geo, prefix, cfg, nslabs, same_device, max_steps, csv);
    FluidParams<FTYPE> params = cfg.useNormalizedParams ? FluidParams<FTYPE>(cfg.Re, cfg.Pr, cfg.lambda)
                                                        : FluidParams<FTYPE>(cfg.viscosity, cfg.density, cfg.R_specific, cfg.k, cfg.cv);
    AdiSolver3D<FTYPE> solver;
    solver.Init(device, grid, params);
    std::printf("Segments: %i %i %i (x, y, z)\n", solver.numSegs[0], solver.numSegs[1], solver.numSegs[2]);

    const int frames = geo.frames;                                     // FluidSolver3D.cpp:194-195
    const double length = geo.length;
    const double dt = length / (frames * cfg.time_steps);              // :196
    const double finaltime = length * cfg.cycles;
    const std::string out = prefix + "_res.nc";
    NetCDF3Writer nc;
    const float *bbox = geo.bbox;
    DepthInfo3D out_depths;                                                                  // IO.h:270-273
    if (geo.depths) out_depths = DepthInfo3D(cfg.outdimx, cfg.outdimy, *geo.depths);
    nc.Create(out, bbox, dt * cfg.out_time_steps, finaltime, cfg.outdimx, cfg.outdimy, cfg.outdimz, cfg.out_vars, geo.depths != nullptr,
              geo.depths ? out_depths.depth.data() : nullptr);
    std::vector<FTYPE> resVel((size_t)cfg.outdimx * cfg.outdimy * cfg.outdimz * 3);
    std::vector<double> resT((size_t)cfg.outdimx * cfg.outdimy * cfg.outdimz);

    solver.EnableTiming(true);
    const auto t0 = std::chrono::steady_clock::now();
    double t = dt;
    long steps = 0;
    int lastframe = -1;
    // the geometry is frame 0's for the whole run: the reference prepares the grid once, before the loop (grid->Prepare(0), :226;
    // the per-step grid->Prepare(t) is commented out, :237) -- the frame only restarts the substep counter
    for (int i = 0; t < finaltime && (max_steps < 0 || steps < max_steps); t += dt, i++, steps++) {
        const int currentframe = geo.GetFrame(t);                                                        // :229-236
        if (currentframe != lastframe) { lastframe = currentframe; i = 0; }
        solver.UpdateBoundaries();                                                                       // :244
        solver.TimeStep((FTYPE)dt, cfg.num_global, cfg.num_local, (i % 10 == 0) || (t + dt >= finaltime)); // :245
        std::printf("\rerr = %.8f,", solver.diffError);                                                  // AdiSolver3D.cpp:376
        const float elapsed = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
        const float perres = (float)t * 100 / (float)finaltime;                                          // PrintTimeStepInfo, IO.h:455-478
        if (perres < 2) std::printf(" frame %i\tsubstep %i\t%i%%\t(----- left)", currentframe, i, (int)perres);
        else {
            const float left = elapsed * (100 - perres) / perres;
            std::printf(" frame %i\tsubstep %i\t%i%%\t(%i h %i m %i s left)", currentframe, i, (int)perres, ((int)left) / 3600, (((int)left) / 60) % 60, ((int)left) % 60);
        }
        std::fflush(stdout);
        if ((i % cfg.out_time_steps) == 0) {                                                             // :254-264
            solver.GetLayer(resVel.data(), resT.data(), cfg.outdimx, cfg.outdimy, cfg.outdimz);
            nc.AppendLayer(resVel.data(), resT.data());
        }
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // the reference's Profiler table (Common/Profiler.h:90-131: sorted by total time, events that never ran are absent; event
    // names of AdiSolver3D.cpp:297-367, 555-680).  Device times from HIP events; MergeLayer has no launches of its own while
    // the merge is fused into the sweep kernels; CreateSegments is the host time of the geometry upload.
    const char *names[FS3D_N_EVENTS]; float ms[FS3D_N_EVENTS]; int cnt[FS3D_N_EVENTS];
    solver.ProfilerEvents(names, ms, cnt);
    std::vector<int> order;
    for (int e = 0; e < FS3D_N_EVENTS; e++) if (cnt[e]) order.push_back(e);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return ms[a] > ms[b]; });
    double total = 0;
    if (csv) std::printf("\n%s,%s,%s,%s,\n", "Event Name", "Total (ms)", "Avg (ms)", "Count");
    else std::printf("\nProfiling data node(0):\n%16s%16s%16s%16s\n", "Event Name", "Total (ms)", "Avg (ms)", "Count");
    for (int e : order) {
        if (csv) std::printf("%s,%.2f,%.2f,%i,\n", names[e], ms[e], ms[e] / cnt[e], cnt[e]);
        else std::printf("%16s%16.2f%16.2f%16i\n", names[e], ms[e], ms[e] / cnt[e], cnt[e]);
        total += ms[e];
    }
    if (csv) std::printf("%s,%.2f,sec\n", "Overall", total / 1000);
    else std::printf("%16s%16.2f sec\n", "Overall", total / 1000);
    {
        int kx, ky, kz, sg;
        const char *kn[] = {"none", "line", "pipe", "part"};
        fs3d_last_sweep_kernel(solver.ctx(), 0, &kx, &sg); fs3d_last_sweep_kernel(solver.ctx(), 1, &ky, &sg); fs3d_last_sweep_kernel(solver.ctx(), 2, &kz, &sg);
        std::printf("Sweep kernels: X %s, Y %s, Z %s\n", kn[kx & 3], kn[ky & 3], kn[kz & 3]);
    }
    std::printf("%ld steps in %.3f s: %.1f Mcells/s; %u layers in %s\n", steps, sec,
                (double)grid.dimx * grid.dimy * grid.dimz * steps / sec / 1e6, nc.NumRecords(), out.c_str());
    return 0;
}

// "GPU n" with n > 1: the reference's single-process multi-GPU mode (GPUplan): n x-slabs (GPUplan::splitEven1D,
// GPUplan.cpp:122-141), one solver and one host thread per slab, joined by the library's in-process group.
// Results equal the single-GPU run's value for value.  Slab r runs on device r, or all on device 0 with --same-device.
namespace {
struct Barrier {
    std::mutex m; std::condition_variable cv; int n, waiting = 0; long gen = 0; bool broken = false;
    explicit Barrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const long g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); } else cv.wait(lk, [&] { return gen != g || broken; });
        if (broken) throw std::runtime_error("another slab thread failed");
    }
    void abort() { { std::lock_guard<std::mutex> lk(m); broken = true; } cv.notify_all(); }
};
}

template <typename FTYPE>
static int run_slabs(const fs3d::Grid3D<FTYPE> &grid, const RunGeom &geo, const std::string &prefix, const fs3d::Config &cfg, int nslabs,
                     bool same_device, long max_steps, bool csv)
{
    using namespace fs3d;
    FluidParams<FTYPE> params = cfg.useNormalizedParams ? FluidParams<FTYPE>(cfg.Re, cfg.Pr, cfg.lambda)
                                                        : FluidParams<FTYPE>(cfg.viscosity, cfg.density, cfg.R_specific, cfg.k, cfg.cv);
    const double length = geo.length, dt = length / (geo.frames * cfg.time_steps), finaltime = length * cfg.cycles;
    const std::string out = prefix + "_res.nc";
    NetCDF3Writer nc;
    const float *bbox = geo.bbox;
    DepthInfo3D out_depths;                                                                  // IO.h:270-273
    if (geo.depths) out_depths = DepthInfo3D(cfg.outdimx, cfg.outdimy, *geo.depths);
    nc.Create(out, bbox, dt * cfg.out_time_steps, finaltime, cfg.outdimx, cfg.outdimy, cfg.outdimz, cfg.out_vars, geo.depths != nullptr,
              geo.depths ? out_depths.depth.data() : nullptr);
    const size_t ncell = (size_t)grid.dimx * grid.dimy * grid.dimz, plane = (size_t)grid.dimy * grid.dimz;
    std::vector<FTYPE> fullV(ncell * 3), resVel((size_t)cfg.outdimx * cfg.outdimy * cfg.outdimz * 3);
    std::vector<double> fullT(ncell), resT((size_t)cfg.outdimx * cfg.outdimy * cfg.outdimz);
    void *group = AdiSolver3D<FTYPE>::CreateLocalGroup(nslabs);
    Barrier bar(nslabs);
    std::vector<std::exception_ptr> errs(nslabs);
    std::vector<std::thread> th;
    long steps_done = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < nslabs; r++)
        th.emplace_back([&, r] {
            try {
                const int q = grid.dimx / nslabs, rem = grid.dimx % nslabs;
                const int x0 = r * q + std::min(r, rem), x1 = x0 + q + (r < rem ? 1 : 0);
                AdiSolver3D<FTYPE> solver;
                solver.Init(same_device ? 0 : r, grid, params, x0, x1);
                solver.JoinLocalGroup(group, r);
                if (r == 0) std::printf("Slabs: %d x-slabs of %d..%d planes\n", nslabs, q, q + (rem ? 1 : 0));
                double t = dt;
                long steps = 0;
                int lastframe = -1;
                for (int i = 0; t < finaltime && (max_steps < 0 || steps < max_steps); t += dt, i++, steps++) {
                    const int currentframe = geo.GetFrame(t);
                    if (currentframe != lastframe) { lastframe = currentframe; i = 0; }
                    solver.UpdateBoundaries();
                    solver.TimeStep((FTYPE)dt, cfg.num_global, cfg.num_local, (i % 10 == 0) || (t + dt >= finaltime));
                    if (r == 0) { std::printf("\rerr = %.8f, frame %i\tsubstep %i\t%i%%", solver.diffError, currentframe, i, (int)((float)t * 100 / (float)finaltime)); std::fflush(stdout); }
                    if ((i % cfg.out_time_steps) == 0) {
                        // each slab's part of `next` at full resolution (NODE_OUT stamped 99999), then FilterToArrays on the
                        // assembled layer (TimeLayer3D.h:819-924: nearest-neighbour down-sample)
                        solver.GetLayer(fullV.data() + 3 * (size_t)x0 * plane, fullT.data() + (size_t)x0 * plane, 0, 0, 0);
                        bar.wait();
                        if (r == 0) {
                            for (int a = 0; a < cfg.outdimx; a++)
                                for (int b = 0; b < cfg.outdimy; b++)
                                    for (int c = 0; c < cfg.outdimz; c++) {
                                        const size_t id = grid.Index(a * grid.dimx / cfg.outdimx, b * grid.dimy / cfg.outdimy, c * grid.dimz / cfg.outdimz);
                                        const size_t ind = ((size_t)a * cfg.outdimy + b) * cfg.outdimz + c;
                                        resVel[3 * ind] = fullV[3 * id]; resVel[3 * ind + 1] = fullV[3 * id + 1]; resVel[3 * ind + 2] = fullV[3 * id + 2];
                                        resT[ind] = fullT[id];
                                    }
                            nc.AppendLayer(resVel.data(), resT.data());
                        }
                        bar.wait();
                    }
                }
                if (r == 0) steps_done = steps;
            } catch (...) {
                // release the other slab threads: they wait for this one in the transport (halo planes, carries) or at the
                // output barrier and would never return
                errs[r] = std::current_exception();
                fs3d_local_group_abort(group);
                bar.abort();
            }
        });
    for (auto &t : th) t.join();
    AdiSolver3D<FTYPE>::DestroyLocalGroup(group);
    for (auto &e : errs) if (e) {          // the first failure, not the "another slab thread failed" it caused
        try { std::rethrow_exception(e); } catch (std::exception &x) { if (std::string(x.what()).find("another slab thread") == std::string::npos && std::string(x.what()).find("a peer failed") == std::string::npos) throw; } catch (...) { throw; }
    }
    for (auto &e : errs) if (e) std::rethrow_exception(e);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)csv;
    std::printf("\n%ld steps in %.3f s: %.1f Mcells/s; %u layers in %s\n", steps_done, sec,
                (double)grid.dimx * grid.dimy * grid.dimz * steps_done / sec / 1e6, nc.NumRecords(), out.c_str());
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::printf("Usage: %s <input data> <output prefix> <config file> [align] [GPU [n]] [double] [--steps N] [--grid-only FILE] [--grid-images]\n", argv[0]);
        return 0;
    }
    try {
        fs3d::Config cfg;
        cfg.Load(argv[3]);
        if (cfg.problem_dim != "3D") throw std::runtime_error("only `dimension 3D` runs are supported");
        if (cfg.in_fmt != "Shape2D" && cfg.in_fmt != "Shape3D" && cfg.in_fmt != "SeaNetCDF") throw std::runtime_error("in_fmt " + cfg.in_fmt + ": unknown input format");
        if (cfg.in_fmt != "Shape2D" && !(cfg.frame_time > 0)) throw std::runtime_error("must specify frame time!");   // the cycle length of a Shape3D run (Grid3D.cpp:303-309)
        if (cfg.solver != "ADI") throw std::runtime_error("solver " + cfg.solver + " is not implemented (the reference implements ADI only)");
        bool align = false, dbl = false, csv = false, same_device = false, grid_images = false;
        int nslabs = 1;
        int device = 0;
        long max_steps = -1;
        std::string grid_only;
        for (int a = 4; a < argc; a++) {
            const std::string s = argv[a];
            if (s == "align") align = true;
            else if (s == "GPU") { if (a + 1 < argc && std::atoi(argv[a + 1]) > 0) nslabs = std::atoi(argv[++a]); }   // the reference's "GPU n": n devices of one process
            else if (s == "--same-device") same_device = true;
            else if (s == "double") dbl = true;
            else if (s == "--device" && a + 1 < argc) device = std::atoi(argv[++a]);
            else if (s == "--steps" && a + 1 < argc) max_steps = std::atol(argv[++a]);
            else if (s == "--grid-only" && a + 1 < argc) grid_only = argv[++a];
            else if (s == "blocking") { if (a + 1 < argc) a++; }
            else if (s == "CSV") csv = true;
            else if (s == "--grid-images") grid_images = true;       // <prefix>_grid_3d/<k>.bmp: the node types, one image per z-slice
            // transpose, decompose: accepted, no effect
        }
        return dbl ? run<double>(argv[1], argv[2], cfg, align, device, max_steps, grid_only, csv, nslabs, same_device, grid_images)
                   : run<float>(argv[1], argv[2], cfg, align, device, max_steps, grid_only, csv, nslabs, same_device, grid_images);
    } catch (std::exception &e) {
        std::fprintf(stderr, "\n\nCaught exception:\n%s\n\nTerminating...\n", e.what());     // FluidSolver3D.cpp:313-318
        return -1;
    }
}
