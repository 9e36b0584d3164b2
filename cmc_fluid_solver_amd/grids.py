"""Node arrays (the reference's Grid3D `Node` AoS, Grid3D.h:73-88, as SoA) and
synthetic geometry builders used by bench.py and the tests.

A `Nodes` object is what the C-ABI's fs3d_upload_nodes() takes: per cell the node
type, the two boundary-condition kinds and the boundary values (v, T).
Index order is the reference's: id = i*dimy*dimz + j*dimz + k (TimeLayer3D.h:256-259).
"""
from dataclasses import dataclass

import numpy as np

# Geometry.h:29-43
NODE_IN, NODE_OUT, NODE_BOUND, NODE_VALVE = 0, 1, 2, 3
BC_NOSLIP, BC_FREE = 0, 1


@dataclass
class Nodes:
    dimx: int
    dimy: int
    dimz: int
    dx: float
    dy: float
    dz: float
    type: np.ndarray      # uint8 [dimx,dimy,dimz]
    bc_vel: np.ndarray    # uint8
    bc_temp: np.ndarray   # uint8
    vx: np.ndarray        # float64 (cast to the solver precision on upload)
    vy: np.ndarray
    vz: np.ndarray
    T: np.ndarray

    @property
    def shape(self):
        return (self.dimx, self.dimy, self.dimz)

    @property
    def ncells(self):
        return self.dimx * self.dimy * self.dimz

    def count(self, t):
        return int((self.type == t).sum())

    def slab(self, x0, x1):
        """Planes [x0, x1) as an independent Nodes object (x-slab of the reference's
        decomposition, Grid3D.cpp:567-581)."""
        s = slice(x0, x1)
        return Nodes(x1 - x0, self.dimy, self.dimz, self.dx, self.dy, self.dz,
                     *[np.ascontiguousarray(a[s]) for a in
                       (self.type, self.bc_vel, self.bc_temp, self.vx, self.vy, self.vz, self.T)])


def _empty(dimx, dimy, dimz, dx, dy, dz):
    sh = (dimx, dimy, dimz)
    return Nodes(dimx, dimy, dimz, dx, dy, dz,
                 np.full(sh, NODE_OUT, np.uint8), np.zeros(sh, np.uint8), np.zeros(sh, np.uint8),
                 np.zeros(sh), np.zeros(sh), np.zeros(sh), np.zeros(sh))


def _set_bound(n, sel, bc_vel, bc_temp, v, T, ntype=NODE_BOUND):
    """Node::SetBound (Grid3D.h:80-87) on a selection."""
    n.type[sel] = ntype
    n.bc_vel[sel] = bc_vel
    n.bc_temp[sel] = bc_temp
    n.vx[sel], n.vy[sel], n.vz[sel] = v
    n.T[sel] = T


def box(dimx, dimy=None, dimz=None, h=None, baseT=1.0, inflow=1.0):
    """Direct synthetic box: outer shell NODE_BOUND (no-slip
    velocity / free temperature, as Grid3D::Prepare2D sets walls, Grid3D.cpp:626-642),
    the open part of the x=0 face an inflow valve (no-slip, U=inflow, T=baseT), the
    open part of the x=dimx-1 face a free outflow valve (Grid3D.cpp:650-655), the
    interior NODE_IN at rest with T=baseT.  One segment per line, length = dim."""
    dimy = dimy or dimx
    dimz = dimz or dimx
    h = h if h is not None else 1.0 / (max(dimx, dimy, dimz) - 1)
    n = _empty(dimx, dimy, dimz, h, h, h)
    n.type[...] = NODE_IN
    n.T[...] = baseT
    shell = np.zeros(n.shape, bool)
    shell[0], shell[-1] = True, True
    shell[:, 0], shell[:, -1] = True, True
    shell[:, :, 0], shell[:, :, -1] = True, True
    _set_bound(n, shell, BC_NOSLIP, BC_FREE, (0.0, 0.0, 0.0), baseT)
    face = np.zeros(n.shape, bool)
    face[0, 1:-1, 1:-1] = True
    _set_bound(n, face, BC_NOSLIP, BC_NOSLIP, (inflow, 0.0, 0.0), baseT, NODE_VALVE)
    face[...] = False
    face[-1, 1:-1, 1:-1] = True
    _set_bound(n, face, BC_FREE, BC_FREE, (0.0, 0.0, 0.0), baseT, NODE_VALVE)
    return n


def box_with_obstacle(dimx, dimy=None, dimz=None, h=None, baseT=1.0, inflow=1.0, lo=0.4, hi=0.6):
    """box() plus a solid block in the middle: its surface is NODE_BOUND (no-slip),
    its inside NODE_OUT.  Lines through the block carry two segments
    (MAX_SEGS_PER_ROW, Grid3D.h:43) and the masks are non-trivial (BASELINE config 5)."""
    n = box(dimx, dimy, dimz, h, baseT, inflow)
    r = [(max(2, int(lo * d)), min(d - 3, int(hi * d))) for d in n.shape]
    blk = np.zeros(n.shape, bool)
    blk[r[0][0]:r[0][1] + 1, r[1][0]:r[1][1] + 1, r[2][0]:r[2][1] + 1] = True
    inner = np.zeros(n.shape, bool)
    inner[r[0][0] + 1:r[0][1], r[1][0] + 1:r[1][1], r[2][0] + 1:r[2][1]] = True
    _set_bound(n, blk, BC_NOSLIP, BC_FREE, (0.0, 0.0, 0.0), baseT)
    n.type[inner] = NODE_OUT
    n.bc_vel[inner] = BC_NOSLIP
    n.bc_temp[inner] = BC_NOSLIP
    n.T[inner] = 0.0
    return n


def perturb(fields, seed=1234, vel=0.05, temp=0.01, mask=None):
    """Seeded stress state: u,v,w += U(-vel,vel), T += U(-temp,temp)
    on the cells selected by mask (default: everywhere)."""
    rng = np.random.default_rng(seed)
    out = []
    for v, f in enumerate(fields):
        amp = temp if v == 3 else vel
        d = rng.uniform(-amp, amp, size=f.shape).astype(f.dtype)
        if mask is not None:
            d = np.where(mask, d, 0).astype(f.dtype)
        out.append(f + d)
    return out
