"""Shape3D input surface: the reference's `in_fmt Shape3D` triangle-mesh geometry rasterised into the Node array.

Python twin of cmc_fluid_solver_amd/host/Shape3D.h (same operations in np.float32 where the reference computes in FTYPE = float;
the header lists the reference lines and the two deliberate deviations: NODE_BOUND cells read as zero-filled memory --
BC_NOSLIP, v = 0, T = 0 -- and cells addressed outside the grid are ignored).  Restates
  Grid3D::Load3DShape / Init / Prepare3D_Shape / ComputeSubframeInfo / Build / RasterPolygon / ProjectPointOnPolygon /
  RasterLine / FloodFill                         (FluidSolver3D/Grid3D.cpp:351-431, 676-946)
  BBox3D::Build                                  (Common/Geometry.h:510-529)
Pinned to the reference (r3): tests/test_ref_golden.py holds this loader cell for cell to the node arrays of the reference's own
Grid3D on the shipped box_pipe_3D and tetra meshes and on a two-frame icosphere at five times (tests/golden/ref_*_3D_*.npz,
ref_tetra_f32.npz); tests/test_shape3d.py compares the C++ loader with this twin.
"""
import math

import numpy as np

from .grids import BC_NOSLIP, NODE_BOUND, NODE_IN, NODE_OUT, Nodes
from .shape2d import align_by_32

F = np.float32
GRID_SCALE_FACTOR = F(0.001)
COMP_EPS = 1e-8
BBOX_PADDING = 0.02
INF = 1e10


def parse_shape3d(text):
    """Grid3D::Load3DShape (Grid3D.cpp:373-416): frames of (vertices [n,3] float32, velocities [n,3], triangles [m,3] int)."""
    tok = text.replace("\r", "").split()
    it = iter(tok)
    num = lambda: F(float(next(it).replace(",", ".")))
    frames = []
    for _ in range(int(next(it))):
        nv = int(next(it))
        v = np.zeros((nv, 3), np.float32); w = np.zeros((nv, 3), np.float32)
        for k in range(nv):
            v[k] = [F(num() * GRID_SCALE_FACTOR) for _ in range(3)]
            w[k] = [num() for _ in range(3)]
        nt = int(next(it))
        idx = np.array([int(next(it)) for _ in range(3 * nt)], np.int64).reshape(nt, 3)
        if idx.size and (idx.min() < 0 or idx.max() >= nv):
            raise ValueError("Shape3D: triangle index outside the vertex list")
        frames.append({"v": v, "vel": w, "idx": idx, "duration": 1.0 / 75})
    return frames


class Shape3D:
    def __init__(self, frames, dx, dy, dz, align, time=0.0):
        self.frames = frames
        self.dx, self.dy, self.dz = dx, dy, dz
        allv = np.concatenate([fr["v"] for fr in frames], axis=0)
        mn = np.minimum(allv.min(axis=0), F(INF)).astype(np.float32); mx = np.maximum(allv.max(axis=0), F(-INF)).astype(np.float32)
        w = (mx - mn).astype(np.float32)
        pad = (w * F(BBOX_PADDING)).astype(np.float32)
        mn = (mn - pad).astype(np.float32); mx = (mx + pad).astype(np.float32)
        self.bbox = tuple(mn) + tuple(mx)
        dims = [int(math.ceil(float(F(mx[a] - mn[a])) / d)) + 1 for a, d in enumerate((dx, dy, dz))]
        if align:
            dims = [align_by_32(d) for d in dims]
        self.dimx, self.dimy, self.dimz = dims
        h = np.array([F(dx), F(dy), F(dz)], np.float32)
        for fr in frames:
            fr["g"] = ((fr["v"] - mn).astype(np.float32) / h).astype(np.float32)
        self.prepare(time)

    def prepare(self, time):
        nf = len(self.frames)
        a = [0.0]
        for fr in self.frames:
            a.append(a[-1] + fr["duration"])
        r = math.fmod(time, a[-1])
        frame = 0
        for i in range(1, nf):
            if a[i] < r:
                frame = i
        s = F((r - a[frame]) / (a[frame + 1] - a[frame])); i_s = F(F(1) - s)
        f0, f1 = self.frames[frame], self.frames[(frame + 1) % nf]
        g = ((f0["g"] * i_s).astype(np.float32) + (f1["g"] * s).astype(np.float32)).astype(np.float32)
        self.build(g, f0["idx"])

    # ---- rasteriser -----------------------------------------------------------------------------------------------
    def _set(self, i, j, k, c):
        if 0 <= i < self.dimx and 0 <= j < self.dimy and 0 <= k < self.dimz:
            self.type[i, j, k] = c

    @staticmethod
    def _horizon(p1, p2, p):
        if abs(float(F(p1[1] - p2[1]))) < COMP_EPS:
            return (p[0], p[1])
        return (F(p1[0] + F(F(F(p2[0] - p1[0]) * F(p[1] - p1[1])) / F(p2[1] - p1[1]))), p[1])

    def _project(self, d_, i, j, tp, n, d):
        o = [(1, 2), (0, 2), (0, 1)][d_]
        with np.errstate(all="ignore"):
            kf = F(F(F(-d) - F(F(tp[0] * n[o[0]]) + F(tp[1] * n[o[1]]))) / n[d_])
        if not np.isfinite(kf) or abs(float(kf)) > 2e9:
            return
        k = int(kf)
        lim = (self.dimx, self.dimy, self.dimz)[d_]
        if 0 <= k < lim:
            if d_ == 0:
                self._set(k, i, j, NODE_BOUND)
            elif d_ == 1:
                self._set(i, k, j, NODE_BOUND)
            else:
                self._set(i, j, k, NODE_BOUND)

    def _scan_half(self, p, yend, dp, e1, e2, di, d_, n, d):
        bound = 4 * (self.dimx + self.dimy + self.dimz) + 16
        while p[1] < yend:
            j = int(p[1])
            last_i = int(self._horizon(e1, e2, p)[0])
            i, guard = int(p[0]), 0
            while i != last_i + di:
                guard += 1
                if guard > bound:
                    raise ValueError("Shape3D: a scan line of a polygon never reaches its end cell (the reference loops there)")
                self._project(d_, i, j, (F(i), p[1]), n, d)
                i += di
            p = (F(p[0] + dp[0]), F(p[1] + dp[1]))
        return p

    def _raster_polygon(self, p1, p2, p3):
        eq = lambda a, b: all(abs(float(F(a[q] - b[q]))) < COMP_EPS for q in range(3))
        if eq(p1, p2) and eq(p1, p3):
            return
        a = [F(p2[q] - p1[q]) for q in range(3)]; b = [F(p3[q] - p1[q]) for q in range(3)]
        n = [F(F(a[1] * b[2]) - F(a[2] * b[1])), F(F(a[2] * b[0]) - F(a[0] * b[2])), F(F(a[0] * b[1]) - F(a[1] * b[0]))]
        ln = F(np.sqrt(F(F(F(n[0] * n[0]) + F(n[1] * n[1])) + F(n[2] * n[2]))))
        if not ln > 0:
            return
        t = F(F(1) / ln)
        n = [F(c * t) for c in n]
        d = F(-F(F(F(p1[0] * n[0]) + F(p1[1] * n[1])) + F(p1[2] * n[2])))
        maxv = max(abs(n[0]), abs(n[1]), abs(n[2]))
        d_ = 0
        for q in range(3):
            if abs(float(F(maxv - abs(n[q])))) < COMP_EPS:
                d_ = q
        o = [(1, 2), (0, 2), (0, 1)][d_]
        pp = [(p[o[0]], p[o[1]]) for p in (p1, p2, p3)]
        if pp[2][1] < pp[1][1]: pp[1], pp[2] = pp[2], pp[1]
        if pp[0][1] > pp[1][1]: pp[0], pp[1] = pp[1], pp[0]
        if pp[2][1] < pp[1][1]: pp[1], pp[2] = pp[2], pp[1]
        pp1, pp2, pp3 = pp
        mid = self._horizon(pp1, pp3, pp2)
        dir1 = (F(mid[0] - pp1[0]), F(mid[1] - pp1[1])); dir2 = (F(pp3[0] - mid[0]), F(pp3[1] - mid[1]))
        steps1 = int(max(abs(dir1[0]), abs(dir1[1]))) + 1; steps2 = int(max(abs(dir2[0]), abs(dir2[1]))) + 1
        dp1 = (F(dir1[0] / F(steps1)), F(dir1[1] / F(steps1))); dp2 = (F(dir2[0] / F(steps2)), F(dir2[1] / F(steps2)))
        di = 1 if mid[0] < pp2[0] else -1
        p = self._scan_half(pp1, mid[1], dp1, pp1, pp2, di, d_, n, d)
        self._scan_half(p, pp3[1], dp2, pp2, pp3, di, d_, n, d)

    def _raster_line(self, p1, p2):
        dr = [F(p2[q] - p1[q]) for q in range(3)]
        steps = int(max(abs(dr[0]), abs(dr[1]), abs(dr[2]))) + 1
        dp = [F(c / F(steps)) for c in dr]
        p = list(p1)
        for _ in range(steps + 1):
            self._set(int(p[0]), int(p[1]), int(p[2]), NODE_BOUND)
            p = [F(p[q] + dp[q]) for q in range(3)]

    def build(self, g, idx):
        self.type = np.full((self.dimx, self.dimy, self.dimz), NODE_IN, np.uint8)
        for i1, i2, i3 in idx:
            p1, p2, p3 = (tuple(F(c) for c in g[q]) for q in (i1, i2, i3))
            self._raster_polygon(p1, p2, p3)
            self._raster_line(p1, p2); self._raster_line(p1, p3); self._raster_line(p3, p2)
        # flood fill NODE_OUT from (0,0,0) through NODE_IN cells, 6-neighbourhood (scipy labels the same set)
        from scipy import ndimage
        free = self.type == NODE_IN
        free[0, 0, 0] = True
        lab, _ = ndimage.label(free)
        self.type[lab == lab[0, 0, 0]] = NODE_OUT


def load_shape3d(path_or_text, dx, dy, dz, baseT=1.0, align=True, is_text=False):
    """Grid3D(dx,dy,dz,baseT) + LoadFromFile + Prepare_CPU(0) for a Shape3D input -> (Nodes, Shape3D)."""
    text = path_or_text if is_text else open(path_or_text, "r").read()
    sh = Shape3D(parse_shape3d(text), dx, dy, dz, align)
    return nodes_of(sh, dx, dy, dz, baseT), sh


def nodes_of(sh, dx, dy, dz, baseT=1.0):
    """The Node array of a prepared Shape3D grid: NODE_BOUND cells carry NOSLIP, v = 0, T = 0 (what the reference's run holds
    there: tests/golden/ref_box_pipe_3D_f32.npz), every other cell T = baseT."""
    shape = sh.type.shape
    z8 = np.zeros(shape, np.uint8)
    zero = np.zeros(shape, np.float64)
    T = np.where(sh.type == NODE_BOUND, 0.0, float(F(baseT)))
    return Nodes(sh.dimx, sh.dimy, sh.dimz, dx, dy, dz, sh.type.copy(), z8 + BC_NOSLIP, z8 + BC_NOSLIP, zero, zero.copy(), zero.copy(), T)


def write_mesh(path, frames):
    """A Shape3D file from [(vertices in mm [n,3], triangles [m,3])]: the test inputs are written with this."""
    with open(path, "w") as f:
        f.write("%d\n" % len(frames))
        for v, tri in frames:
            f.write("%d\n" % len(v))
            for p in v:
                f.write("%.6g %.6g %.6g 0 0 0\n" % tuple(p))
            f.write("%d\n" % len(tri))
            for t in tri:
                f.write("%d %d %d\n" % tuple(t))
