// TEST INFRASTRUCTURE ONLY.
// The reference's OWN config parser, compiled where it lies (Common/Config.h, with Common/LinuxIO.{h,cpp} that its Linux build
// puts in front of it -- the include order of Common/IO.h:44-49).  Nothing is copied and nothing is stubbed.  Config::LoadFromFile
// ends the process with exit(0) on a config it rejects: callers run this in a child process for such files.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "Common/LinuxIO.h"
using namespace LinuxIO;
#include "Common/Geometry.h"
#include "Common/Config.h"

extern "C" {
// out_d: R_specific k cv baseT bc_strength bc_inV.xyz bc_inT viscosity density Re Pr lambda depth_var frame_time dx dy dz depth (21)
// out_i: bc_noslip useNormalizedParams cycles time_steps out_time_steps outdimx outdimy outdimz num_global num_local problem_dim in_fmt
//        out_fmt solverID n_out_vars (15); out_vars: the names, space separated
void ref_config_load(const char *path, double *out_d, int *out_i, char *out_vars, int out_vars_cap)
{
    using Common::Config;
    Config c;                                   // the constructor sets the defaults (FluidSolver3D.cpp:115: `Config();`)
    (void)c;
    char buf[4096];
    strncpy(buf, path, sizeof buf - 1); buf[sizeof buf - 1] = 0;
    Config::LoadFromFile(buf);
    const double d[21] = {Config::R_specific, Config::k, Config::cv, Config::baseT, Config::bc_strength, Config::bc_inV.x, Config::bc_inV.y, Config::bc_inV.z,
                          Config::bc_inT, Config::viscosity, Config::density, Config::Re, Config::Pr, Config::lambda, Config::depth_var, Config::frame_time,
                          Config::dx, Config::dy, Config::dz, Config::depth, 0.0};
    for (int k = 0; k < 21; k++) out_d[k] = d[k];
    const int i[15] = {Config::bc_noslip, Config::useNormalizedParams, Config::cycles, Config::time_steps, Config::out_time_steps, Config::outdimx,
                       Config::outdimy, Config::outdimz, Config::num_global, Config::num_local, (int)Config::problem_dim, (int)Config::in_fmt,
                       (int)Config::out_fmt, (int)Config::solverID, (int)Config::out_vars.size()};
    for (int k = 0; k < 15; k++) out_i[k] = i[k];
    std::string s;
    for (size_t k = 0; k < Config::out_vars.size(); k++) s += (k ? " " : "") + Config::out_vars[k];
    strncpy(out_vars, s.c_str(), out_vars_cap - 1); out_vars[out_vars_cap - 1] = 0;
}
}
