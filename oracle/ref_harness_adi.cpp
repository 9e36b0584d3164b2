// TEST INFRASTRUCTURE ONLY.
// An own `main` over the reference's OWN classes, linked against the reference's translation units compiled where they
// lie under /root/reference/src (oracle/Makefile, target ref_full).  Nothing is copied and nothing is stubbed: cuda_runtime.h
// is NVIDIA's genuine header that ships in this image's triton package, netcdf.h is the reference's vendored src/NetCDF/netcdf.h,
// and the symbols the link leaves unresolved (cuda*, nc_*, the *_GPU seam functions) are never reached with backend == CPU.
//
// The sequence of calls is the reference driver's (FluidSolver3D/FluidSolver3D.cpp:63-262) minus the NetCDF/BMP output:
// PARAplan::init(CPU), Config::LoadFromFile, Grid3D ctor for the input format, SetFrameTime/SetBoundParams, LoadFromFile,
// Prepare_CPU(0), Split, Init_GPU (returns at once on CPU), AdiSolver3D::Init, CreateSegments, Prepare(0), then the driver's
// loop: UpdateBoundaries + TimeStep(dt, num_global, num_local, (i%10==0) || last) with its frame bookkeeping.
//
// usage: ref_adi <data> <config> <dump file> <max steps> <dump steps: comma list> [align] [gridtimes: comma list of times]
// Dump (little endian, raw):
//   "FS3DREF1" | i32 sizeof(FTYPE) dimx dimy dimz frames | f64 dx dy dz dt cycle_length | f64 v_T v_vis t_vis t_phi
//   u8 type[n] | u8 bc_vel[n] | u8 bc_temp[n] | FTYPE vel[3n] | FTYPE T[n]                         (grid after Prepare(0))
//   i32 ngridtimes, then per time: f64 t | u8 type[n] | FTYPE vel[3n]                              (grid after Prepare_CPU(t))
//   then records, s = 1-based step (after the s-th TimeStep):
//     i32 1 | i32 s | f64 err (cur->EvalDivError(grid) of the new layer) | FTYPE U[n] V[n] W[n] T[n]   (the new layer, `cur` after the swap)
//     i32 2 | i32 s | i32 ox oy oz | FTYPE v[3m] | f64 T[m]       (Solver3D::GetLayer at the config's out dims, at the driver's output steps)
//   i32 -1
#include "FluidSolver3D.h"
#include <typeinfo>

using namespace FluidSolver3D;
using namespace Common;

struct Probe : public AdiSolver3D {           // `cur` is a protected member of Solver3D
    TimeLayer3D *Cur() { return cur; }
};

static std::vector<double> numbers(const char *s)
{
    std::vector<double> r;
    if (!s || !*s || !strcmp(s, "-")) return r;
    std::string t(s);
    size_t p = 0;
    while (p <= t.size()) {
        size_t q = t.find(',', p);
        if (q == std::string::npos) q = t.size();
        r.push_back(atof(t.substr(p, q - p).c_str()));
        p = q + 1;
    }
    return r;
}

template <class T> static void put(FILE *f, const T *p, size_t n) { if (n && fwrite(p, sizeof(T), n, f) != n) { perror("fwrite"); exit(3); } }
template <class T> static void put1(FILE *f, T v) { put(f, &v, 1); }

static void put_grid(FILE *f, Grid3D *g, bool all)
{
    size_t n = (size_t)g->dimx * g->dimy * g->dimz;
    std::vector<unsigned char> ty(n), bv(n), bt(n);
    std::vector<FTYPE> vel(3 * n);
    std::vector<FTYPE> T(n);
    size_t c = 0;
    for (int i = 0; i < g->dimx; i++)
        for (int j = 0; j < g->dimy; j++)
            for (int k = 0; k < g->dimz; k++, c++) {
                ty[c] = (unsigned char)g->GetType(i, j, k);
                bv[c] = (unsigned char)g->GetBC_vel(i, j, k);
                bt[c] = (unsigned char)g->GetBC_temp(i, j, k);
                Vec3D v = g->GetVel(i, j, k);
                vel[3 * c] = v.x; vel[3 * c + 1] = v.y; vel[3 * c + 2] = v.z;
                T[c] = g->GetT(i, j, k);
            }
    put(f, ty.data(), n);
    if (all) { put(f, bv.data(), n); put(f, bt.data(), n); }
    put(f, vel.data(), 3 * n);
    if (all) put(f, T.data(), n);
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s data config dump maxsteps dumpsteps [align] [gridtimes]\n", argv[0]); return 2; }
    try {
        BackendType backend = CPU;
        PARAplan *pplan = PARAplan::Instance();
        pplan->init(backend);
        const int max_steps = atoi(argv[4]);
        std::vector<double> dump_steps = numbers(argv[5]);
        bool align = false;
        std::vector<double> grid_times;
        for (int a = 6; a < argc; a++) {
            if (!strcmp(argv[a], "align")) align = true;
            else grid_times = numbers(argv[a]);
        }
        printf("%s precision computations\n", (typeid(FTYPE) == typeid(float)) ? "Single" : "Double");

        Config();
        Config::LoadFromFile(argv[2]);
        Grid3D *grid = NULL;
        SplitType split_type = EVEN_X;
        if (Config::in_fmt == Shape3D)
            grid = new Grid3D(Config::dx, Config::dy, Config::dz, Config::baseT, backend, false, split_type);
        else if (Config::in_fmt == Shape2D)
            grid = new Grid3D(Config::dx, Config::dy, Config::dz, Config::depth, Config::depth_var, Config::baseT, backend, false, split_type);
        else
            throw std::runtime_error("SeaNetCDF input needs a real libnetcdf");
        grid->SetFrameTime(Config::frame_time);
        grid->SetBoundParams(Config::bc_inV, Config::bc_inT);
        if (grid->LoadFromFile(argv[1], align))
            printf("Grid = %i x %i x %i\n", grid->dimx, grid->dimy, grid->dimz);
        grid->Prepare_CPU(0.0);
        grid->Split();
        grid->Init_GPU();

        double inside = 0;
        for (int i = 0; i < grid->dimx; i++)
            for (int j = 0; j < grid->dimy; j++)
                for (int k = 0; k < grid->dimz; k++)
                    if (grid->GetType(i, j, k) == NODE_IN) inside += 1.0;
        printf("NODE_IN points = %f of total %f\n", inside, double(grid->dimx) * grid->dimy * grid->dimz);

        FluidParams *params;
        if (Config::useNormalizedParams) params = new FluidParams(Config::Re, Config::Pr, Config::lambda);
        else params = new FluidParams(Config::viscosity, Config::density, Config::R_specific, Config::k, Config::cv);

        Probe *solver = new Probe();
        solver->Init(backend, false, grid, *params, false, 1);

        int frames = grid->GetFramesNum();
        double length = grid->GetCycleLength();
        double dt = length / (frames * Config::time_steps);
        double finaltime = length * Config::cycles;

        solver->CreateSegments();
        grid->Prepare(0);

        FILE *f = fopen(argv[3], "wb");
        if (!f) { perror(argv[3]); return 3; }
        put(f, "FS3DREF1", 8);
        put1<int>(f, (int)sizeof(FTYPE)); put1<int>(f, grid->dimx); put1<int>(f, grid->dimy); put1<int>(f, grid->dimz); put1<int>(f, frames);
        put1<double>(f, grid->dx); put1<double>(f, grid->dy); put1<double>(f, grid->dz); put1<double>(f, dt); put1<double>(f, length);
        put1<double>(f, params->v_T); put1<double>(f, params->v_vis); put1<double>(f, params->t_vis); put1<double>(f, params->t_phi);
        put_grid(f, grid, true);

        // geometry at other times (Grid3D::Prepare_CPU(t): the interpolated sub-frame of a multi-frame input).  The driver itself only
        // ever prepares t = 0 (its per-step grid->Prepare(t) is commented out, FluidSolver3D.cpp:237); state is restored afterwards.
        put1<int>(f, (int)grid_times.size());
        for (size_t g = 0; g < grid_times.size(); g++) {
            grid->Prepare_CPU(grid_times[g]);
            put1<double>(f, grid_times[g]);
            put_grid(f, grid, false);
        }
        if (!grid_times.empty()) grid->Prepare_CPU(0.0);

        size_t n = (size_t)grid->dimx * grid->dimy * grid->dimz;
        int ox = Config::outdimx, oy = Config::outdimy, oz = Config::outdimz;
        size_t m = (size_t)ox * oy * oz;
        Vec3D *resVel = new Vec3D[m];
        double *resT = new double[m];
        std::vector<FTYPE> buf(n);

        int lastframe = -1, step = 0;
        double t = dt;
        for (int i = 0; t < finaltime && step < max_steps; t += dt, i++) {
            int currentframe = grid->GetFrame(t);
            if (currentframe != lastframe) { lastframe = currentframe; i = 0; }
            solver->UpdateBoundaries();
            solver->TimeStep((FTYPE)dt, Config::num_global, Config::num_local, (i % 10 == 0) || (t + dt >= finaltime));
            step++;
            printf(" step %d frame %d i %d\n", step, currentframe, i);
            bool dump = false;
            for (size_t d = 0; d < dump_steps.size(); d++) dump |= ((int)dump_steps[d] == step);
            if (dump) {
                TimeLayer3D *cur = solver->Cur();
                put1<int>(f, 1); put1<int>(f, step);
                put1<double>(f, cur->EvalDivError(grid));
                ScalarField3D *fld[4] = {cur->U, cur->V, cur->W, cur->T};
                for (int v = 0; v < 4; v++) {
                    size_t c = 0;
                    for (int a = 0; a < grid->dimx; a++)
                        for (int b = 0; b < grid->dimy; b++)
                            for (int k = 0; k < grid->dimz; k++, c++) buf[c] = fld[v]->elem(a, b, k);
                    put(f, buf.data(), n);
                }
            }
            // the driver's result layers (FluidSolver3D.cpp:249-259): GetLayer is NOT read-only -- it sets the NODE_OUT cells of the
            // older layer to 99999 (Solver3D.cpp:23) -- so it is called exactly where the driver calls it, never elsewhere
            if ((i % Config::out_time_steps) == 0) {
                solver->GetLayer(resVel, resT, ox, oy, oz);
                put1<int>(f, 2); put1<int>(f, step);
                put1<int>(f, ox); put1<int>(f, oy); put1<int>(f, oz);
                for (size_t c = 0; c < m; c++) { FTYPE v3[3] = {resVel[c].x, resVel[c].y, resVel[c].z}; put(f, v3, 3); }
                put(f, resT, m);
            }
        }
        put1<int>(f, -1);
        fclose(f);
        printf("\ndone: %d steps\n", step);
    } catch (std::exception &e) {
        fprintf(stderr, "\n\nCaught exception:\n%s\n", e.what());
        return 1;
    }
    fflush(stdout);
    return 0;
}
