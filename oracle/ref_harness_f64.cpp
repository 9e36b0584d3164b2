// TEST INFRASTRUCTURE ONLY.
// fp64 instantiation of the reference's Thomas solver: Algorithms.h takes FTYPE
// from its includer, so defining it here (instead of including Geometry.h, whose
// hard "#define FTYPE float" a -D cannot override) gives the reference algorithm
// in double without touching or copying any reference file.
#define FTYPE double
#include "Common/Algorithms.h"

extern "C" void ref_tridiag_f64(double *a, double *b, double *c, double *d, double *x, int n)
{
    Common::SolveTridiagonal(a, b, c, d, x, n);
}
