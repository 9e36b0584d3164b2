// fs2d_run: the reference's 2D command line, CPU only (FluidSolver2D/FluidSolver2D.cpp:22-158) for `solver Stable`:
//   fs2d_run <input data> <output file> <config file> [--steps N] [--dump FILE]
// Config (dimension 2D: the input format is Shape2D), Grid2D without alignment, FluidParams from viscosity / density, the time
// loop with the frame bookkeeping of the 3D driver plus grid.Prepare(t) every step (moving walls), results as the reference's
// CDL text (OutputNetCDFHeader2D / OutputNetCDF2D_U, Common/IO.h:278-349, 388-407: dimensions, axes, the u component per layer).
// The Explicit and ADI 2D solvers and the MultiVox text output are not restated.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "Config.h"
#include "Stable2D.h"

int main(int argc, char **argv)
{
    if (argc < 4) { std::printf("Usage: %s <input data> <output file> <config file> [--steps N]\n", argv[0]); return 0; }
    try {
        using namespace fs3d;
        Config cfg;
        cfg.Load(argv[3]);
        if (cfg.problem_dim != "2D") throw std::runtime_error("fs2d_run runs `dimension 2D` configs (3D: fs3d_run)");
        if (cfg.solver != "Stable") throw std::runtime_error("solver " + cfg.solver + ": only the Stable 2D solver is restated");
        if (cfg.out_fmt != "NetCDF") throw std::runtime_error("out_fmt " + cfg.out_fmt + ": only the NetCDF (CDL text) output of the 2D path is restated");
        if (!cfg.bc_noslip) throw std::runtime_error("bc_type Slip: only NoSlip boundaries are restated");
        long max_steps = -1;
        std::string dump;
        for (int a = 4; a < argc; a++) {
            if (std::string(argv[a]) == "--steps" && a + 1 < argc) max_steps = std::atol(argv[++a]);
            else if (std::string(argv[a]) == "--dump" && a + 1 < argc) dump = argv[++a];       // raw u, v, T of the last layer (tests)
        }
        Grid2D grid;
        grid.Load(argv[1], cfg.dx, cfg.dy, cfg.baseT, false);
        std::printf("dx,dy,dimx,dimy,bc_noslip\n%f,%f,%i,%i,%i\n", cfg.dx, cfg.dy, grid.dimx, grid.dimy, (int)cfg.bc_noslip);     // FluidSolver2D.cpp:56-57
        const FluidParams<float> params(cfg.viscosity, cfg.density, cfg.R_specific, cfg.k, cfg.cv);                       // :62
        Stable2D solver;
        solver.Init(grid, params.v_vis);
        const int frames = grid.GetFramesNum();
        const double length = grid.GetCycleLenght(), dt = length / (frames * cfg.time_steps), finaltime = length * cfg.cycles;
        const int ox = cfg.outdimx, oy = cfg.outdimy;
        {   // OutputNetCDFHeader2D
            FILE *f = std::fopen(argv[2], "w");
            if (!f) throw std::runtime_error(std::string("cannot create ") + argv[2]);
            std::fprintf(f, "netcdf 2d_scalar_time_array {\ndimensions:\n\tx = %i ;\n\ty = %i ;\n\ttime = UNLIMITED ;\nvariables:\n", ox, oy);
            std::fprintf(f, "\tfloat x(x) ;\n\t\tx:units = \"metres\" ;\n\t\tx:actual_range = %.2ff, %.2ff ;\n\t\tx:long_name = \"X coordinate\" ;\n", grid.bbox[0], grid.bbox[2]);
            std::fprintf(f, "\tfloat y(y) ;\n\t\ty:units = \"metres\" ;\n\t\ty:actual_range = %.2ff, %.2ff ;\n\t\ty:long_name = \"Y coordinate\" ;\n", grid.bbox[1], grid.bbox[3]);
            std::fprintf(f, "\tdouble time(time) ;\n\t\ttime:units = \"s\" ;\n\t\ttime:actual_range = 0.f, %.2ff ;\n\t\ttime:long_name = \"Time\" ;\n", finaltime);
            std::fprintf(f, "\tdouble u(time, x, y) ;\n\t\tu:units = \"m/s\" ;\n\t\tu:actual_range = 0.f, 1.f ;\n\t\tu:valid_range = 0.f, 1.f ;\n\t\tu:long_name = \"U velocity\" ;\n"
                            "\t\tu:scale_factor =  1.f ;\n\t\tu:var_desc = \"U velocity\",\n\t\t\t\"U\" ; \n");
            std::fprintf(f, "\t// global attributes\n\t:Conventions = \"COARDS\" ;\n\t:title = \"2D Time U velocity data from FluidSolver2D\" ;\n"
                            "\t:history = \"created by using FluidSolver2D library\" ;\n\t:description = \"Test data\" ;\n\t:platform = \"Model\" ;\ndata:\n");
            const float ddx = (float)(grid.bbox[2] - grid.bbox[0]) / ox, ddy = (float)(grid.bbox[3] - grid.bbox[1]) / oy;
            std::fprintf(f, "x = ");
            for (int i = 0; i < ox - 1; i++) std::fprintf(f, "%.2f, ", grid.bbox[0] + ddx * i);
            std::fprintf(f, "%.2f ;\ny = ", grid.bbox[0] + ddx * ox);
            for (int i = 0; i < oy - 1; i++) std::fprintf(f, "%.2f, ", grid.bbox[1] + ddy * i);
            std::fprintf(f, "%.2f ;\ntime = ", grid.bbox[1] + ddy * oy);
            for (float c = 0; c < finaltime; c += (float)(dt * cfg.out_time_steps)) std::fprintf(f, "%.2f, ", c);
            std::fprintf(f, "%.2f ;\nu = \n", finaltime);
            std::fclose(f);
        }
        std::printf("dt = %f\n", dt);
        std::vector<float> ru, rv;
        std::vector<double> rT;
        const auto t0 = std::chrono::steady_clock::now();
        int lastframe = -1, currentcycle = 0;
        long steps = 0, layers = 0, sweeps = 0;
        double t = dt;
        for (int i = 0; t < finaltime && (max_steps < 0 || steps < max_steps); t += dt, i++, steps++) {
            const int currentframe = grid.GetFrame(t);
            if (currentframe != lastframe) { if (currentframe == 0) currentcycle++; lastframe = currentframe; i = 0; }
            grid.Prepare(t);                                                                                              // :118
            solver.UpdateBoundaries();
            solver.TimeStep((float)dt, cfg.num_global, cfg.num_local);
            sweeps += solver.poisson_sweeps;
            std::printf("\rerr = %.4f, frame %i\tsubstep %i\t%i%%", solver.err, currentframe, i, (int)((float)t * 100 / (float)finaltime));
            std::fflush(stdout);
            if ((i % cfg.out_time_steps) == 0) {
                solver.GetLayer(ru, rv, rT, ox, oy);
                const bool finish = (i + cfg.out_time_steps >= cfg.time_steps) && (currentframe == frames - 1) && (currentcycle == cfg.cycles);   // :133-134
                FILE *f = std::fopen(argv[2], "a");
                for (int a = 0; a < ox; a++) {
                    for (int b = 0; b < oy; b++) std::fprintf(f, "%.3f%s", ru[(size_t)a * oy + b], finish && a == ox - 1 && b == oy - 1 ? " ; " : ", ");
                    std::fprintf(f, "\n");
                }
                if (finish) std::fprintf(f, "}");
                std::fclose(f);
                layers++;
            }
        }
        if (!dump.empty()) {
            FILE *f = std::fopen(dump.c_str(), "wb");
            if (!f) throw std::runtime_error("cannot create " + dump);
            const int hdr[2] = {grid.dimx, grid.dimy};
            std::fwrite(hdr, sizeof hdr, 1, f);
            for (const std::vector<float> *a : {&solver.next.u, &solver.next.v, &solver.next.t}) std::fwrite(a->data(), sizeof(float), a->size(), f);
            std::fclose(f);
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("\n%ld steps in %.3f s (%ld Poisson sweeps); %ld layers in %s\n", steps, sec, sweeps, layers, argv[2]);
        return 0;
    } catch (std::exception &e) {
        std::printf("%s\n", e.what());
        return 1;
    }
}
