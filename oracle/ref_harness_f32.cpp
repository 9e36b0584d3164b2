// TEST INFRASTRUCTURE ONLY.
// Thin extern "C" harness around the reference's OWN headers, compiled where they
// lie under /root/reference/src (see oracle/Makefile, target _ref).  Nothing is
// copied and nothing is stubbed: Common/Geometry.h and Common/Algorithms.h are
// the only reference files on the hot path that compile in this image without
// cuda_runtime.h / libnetcdf.  FTYPE = float as shipped (Geometry.h:21).
#include "Common/Geometry.h"
#include "Common/Algorithms.h"

extern "C" {
// Common::SolveTridiagonal, Algorithms.h:21-38
void ref_tridiag_f32(float *a, float *b, float *c, float *d, float *x, int n)
{
    Common::SolveTridiagonal(a, b, c, d, x, n);
}
// Common::FluidParams, Geometry.h:538-562
void ref_fluid_params_normalized_f32(double Re, double Pr, double lambda, float *out)
{
    Common::FluidParams p(Re, Pr, lambda);
    out[0] = p.v_T; out[1] = p.v_vis; out[2] = p.t_vis; out[3] = p.t_phi;
}
void ref_fluid_params_physical_f32(double vis, double rho, double R, double k, double cv, float *out)
{
    Common::FluidParams p(vis, rho, R, k, cv);
    out[0] = p.v_T; out[1] = p.v_vis; out[2] = p.t_vis; out[3] = p.t_phi;
}
// Common::AlignBy32, Geometry.h:564-568
int ref_align_by_32(int num) { return Common::AlignBy32(num); }
}
