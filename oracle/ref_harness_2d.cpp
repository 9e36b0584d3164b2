// TEST INFRASTRUCTURE ONLY.
// An own `main` over the reference's OWN 2D classes (Grid2D, StableSolver2D, Solver2D, Config), linked against the reference's
// translation units compiled where they lie (oracle/Makefile, target ref_full; nothing copied, nothing stubbed -- see
// ref_harness_adi.cpp).  The reference's own 2D driver (FluidSolver2D/FluidSolver2D.cpp) does not compile on Linux
// (`#include <string.>`, MAX_PATH, `Config::Config()`); the sequence of calls below is its main's (:48-141) without the output files:
// Config::LoadFromFile, Grid2D(dx, dy, baseT, bc_noslip, bc_strength), LoadFromFile(data, ""), Prepare(0, 0),
// FluidParams(viscosity, density, R, k, cv), StableSolver2D::Init, then per step grid.Prepare(t), UpdateBoundaries, TimeStep,
// SetGridBoundaries with the frame bookkeeping of the loop.
//
// usage: ref_stable2d <data> <config> <dump file> <max steps> <dump steps: comma list>
// Dump: "FS2DREF1" | i32 sizeof(FTYPE) dimx dimy frames | f64 dx dy dt cycle_length v_vis
//       then per dumped step: i32 s | u8 type[n] | FTYPE U[n] V[n] T[n] (the new layer) ; i32 -1
#include "StableSolver2D.h"
#include "../Common/IO.h"
#include "../Common/Config.h"
#include <vector>
#include <string>

using namespace FluidSolver2D;
using namespace Common;

struct Probe : public StableSolver2D {
    TimeLayer2D *Cur() { return cur; }
};

template <class T> static void put(FILE *f, const T *p, size_t n) { if (n && fwrite(p, sizeof(T), n, f) != n) { perror("fwrite"); exit(3); } }
template <class T> static void put1(FILE *f, T v) { put(f, &v, 1); }

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s data config dump maxsteps dumpsteps\n", argv[0]); return 2; }
    try {
        const int max_steps = atoi(argv[4]);
        std::vector<int> dumps;
        { std::string t(argv[5]); size_t p = 0; while (p <= t.size() && t != "-") { size_t q = t.find(',', p); if (q == std::string::npos) q = t.size(); dumps.push_back(atoi(t.substr(p, q - p).c_str())); p = q + 1; } }
        Config();
        Config::LoadFromFile(argv[2]);
        char empty[4] = "";
        Grid2D grid(Config::dx, Config::dy, Config::baseT, Config::bc_noslip, Config::bc_strength);
        if (grid.LoadFromFile(argv[1], empty))
            printf("dx,dy,dimx,dimy,bc_noslip\n%f,%f,%i,%i,%i\n", Config::dx, Config::dy, grid.dimx, grid.dimy, Config::bc_noslip);
        grid.Prepare(0, 0);
        FluidParams params(Config::viscosity, Config::density, Config::R_specific, Config::k, Config::cv);
        Probe *solver = new Probe();
        solver->Init(&grid, params);

        int frames = grid.GetFramesNum();
        double length = grid.GetCycleLenght();
        double dt = length / (frames * Config::time_steps);
        double finaltime = length * Config::cycles;
        printf("dt = %f\n", dt);

        FILE *f = fopen(argv[3], "wb");
        if (!f) { perror(argv[3]); return 3; }
        put(f, "FS2DREF1", 8);
        put1<int>(f, (int)sizeof(FTYPE)); put1<int>(f, grid.dimx); put1<int>(f, grid.dimy); put1<int>(f, frames);
        put1<double>(f, grid.dx); put1<double>(f, grid.dy); put1<double>(f, dt); put1<double>(f, length); put1<double>(f, params.v_vis);
        const size_t n = (size_t)grid.dimx * grid.dimy;
        std::vector<unsigned char> ty(n);
        std::vector<FTYPE> buf(n);

        int lastframe = -1, step = 0;
        double t = dt;
        for (int i = 0; t < finaltime && step < max_steps; t += dt, i++) {
            int currentframe = grid.GetFrame(t);
            if (currentframe != lastframe) { lastframe = currentframe; i = 0; }
            grid.Prepare(t);
            solver->UpdateBoundaries();
            solver->TimeStep((FTYPE)dt, Config::num_global, Config::num_local);
            solver->SetGridBoundaries();
            step++;
            printf(" step %d frame %d i %d\n", step, currentframe, i);
            bool dump = false;
            for (size_t d = 0; d < dumps.size(); d++) dump |= dumps[d] == step;
            if (!dump) continue;
            TimeLayer2D *cur = solver->Cur();
            put1<int>(f, step);
            for (int a = 0; a < grid.dimx; a++) for (int b = 0; b < grid.dimy; b++) ty[(size_t)a * grid.dimy + b] = (unsigned char)grid.GetType(a, b);
            put(f, ty.data(), n);
            for (int v = 0; v < 3; v++) {
                for (int a = 0; a < grid.dimx; a++)
                    for (int b = 0; b < grid.dimy; b++) buf[(size_t)a * grid.dimy + b] = v == 0 ? cur->U(a, b) : v == 1 ? cur->V(a, b) : cur->T(a, b);
                put(f, buf.data(), n);
            }
        }
        put1<int>(f, -1);
        fclose(f);
        printf("\ndone: %d steps\n", step);
    } catch (std::exception &e) {
        fprintf(stderr, "\n\nCaught exception:\n%s\n", e.what());
        return 1;
    }
    return 0;
}
