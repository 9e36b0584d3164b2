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

// Common::BBox2D::Build, Geometry.h:464-480: frames x shapes flattened -- npts[f] points of frame f (one shape per frame is
// enough: Build only walks the points), xy = all points, out = pMin.x pMin.y pMax.x pMax.y
void ref_bbox2d_build(int num_frames, const int *npts, const float *xy, float *out)
{
    Common::FrameInfo2D *frames = new Common::FrameInfo2D[num_frames];
    size_t at = 0;
    for (int f = 0; f < num_frames; f++) {
        frames[f].Init(1);
        frames[f].Shapes[0].Init(npts[f]);
        for (int k = 0; k < npts[f]; k++, at++) frames[f].Shapes[0].Points[k] = Common::Vec2D(xy[2 * at], xy[2 * at + 1]);
    }
    Common::BBox2D bb;
    bb.Build(num_frames, frames);
    out[0] = bb.pMin.x; out[1] = bb.pMin.y; out[2] = bb.pMax.x; out[3] = bb.pMax.y;
    for (int f = 0; f < num_frames; f++) frames[f].Dispose();
    delete[] frames;
}
// Common::BBox3D::Build, Geometry.h:510-529
void ref_bbox3d_build(int num_frames, const int *nverts, const float *xyz, float *out)
{
    Common::FrameInfo3D *frames = new Common::FrameInfo3D[num_frames];
    size_t at = 0;
    for (int f = 0; f < num_frames; f++) {
        frames[f].Init(1);
        frames[f].Shapes[0].InitVerts(nverts[f]);
        for (int k = 0; k < nverts[f]; k++, at++) frames[f].Shapes[0].Vertices[k] = Common::Vec3D(xyz[3 * at], xyz[3 * at + 1], xyz[3 * at + 2]);
    }
    Common::BBox3D bb;
    bb.Build(num_frames, frames);
    out[0] = bb.pMin.x; out[1] = bb.pMin.y; out[2] = bb.pMin.z; out[3] = bb.pMax.x; out[4] = bb.pMax.y; out[5] = bb.pMax.z;
    delete[] frames;
}
// Common::DepthInfo3D resampling constructor, Geometry.h:429-441
void ref_depth_resample(int nx, int ny, int src_nx, int src_ny, const float *src, float *out)
{
    Common::DepthInfo3D info(src_nx, src_ny);
    for (int c = 0; c < src_nx * src_ny; c++) info.depth[c] = src[c];
    Common::DepthInfo3D r(nx, ny, &info);
    for (int c = 0; c < nx * ny; c++) out[c] = r.depth[c];
}
}
