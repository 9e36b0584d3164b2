"""SeaNetCDF input surface: the reference's `in_fmt SeaNetCDF` depth map (netCDF-4: _lat_subset, _lon_subset, z) -> Node array.

Python twin of cmc_fluid_solver_amd/host/SeaNetCDF.h (the header cites the reference lines): Grid3D::LoadNetCDF / Init /
Prepare3D_NetCDF (FluidSolver3D/Grid3D.cpp:351-371, 433-486, 968-1075) and DepthInfo3D (Common/Geometry.h:418-447).  The file
is read with hdf5_min.py.  Parity unpinned (the survey could not run the reference on this input).
"""
import math

import numpy as np

from .grids import BC_NOSLIP, NODE_BOUND, NODE_IN, NODE_OUT, NODE_VALVE, Nodes
from .hdf5_min import Hdf5File
from .shape2d import align_by_32

F = np.float32


def resample_depths(depth, nx, ny):
    """DepthInfo3D(nx, ny, info): nearest-neighbour pick (Geometry.h:429-441)."""
    dx_, dy_ = depth.shape
    ii = (np.arange(nx) * dx_) // nx
    jj = (np.arange(ny) * dy_) // ny
    return depth[np.ix_(ii, jj)].astype(np.float32)


def load_seanetcdf(path, dx, dy, dz, baseT=1.0, bc_inV=(0.0, 0.0, 0.0), bc_inT=1.0, align=True):
    """-> (Nodes, info) with info = {"bbox": 6 floats, "depth": float32 [nlat, nlon]}."""
    f = Hdf5File(path)
    lats, lons = f.read("_lat_subset"), f.read("_lon_subset")
    depth = f.read("z").astype(np.float32)
    nx, ny = len(lats), len(lons)
    assert depth.shape == (nx, ny)
    mn = [min(F(lats[0]), F(lats[-1])), min(F(lons[0]), F(lons[-1])), F(min(F(0), depth.min()))]
    mx = [max(F(lats[0]), F(lats[-1])), max(F(lons[0]), F(lons[-1])), F(0)]
    mn[2] = F(mn[2] - F(dz))
    dims = [int(math.ceil(float(F(mx[a] - mn[a])) / h)) + 1 for a, h in enumerate((dx, dy, dz))]
    if align:
        dims = [align_by_32(d) for d in dims]
    dimx, dimy, dimz = dims
    typ = np.full(dims, NODE_OUT, np.uint8)
    bT = float(F(baseT))
    vx = np.zeros(dims); vy = np.zeros(dims); vz = np.zeros(dims); T = np.full(dims, bT)
    zz = resample_depths(depth, dimx, dimy)                                  # depth[dj + di * dimy] with di = i*nx/dimx, dj = j*ny/dimy
    with np.errstate(all="ignore"):
        bound_k = (F(dimz) * zz / mn[2]).astype(np.float32).astype(np.int64)     # (int)(dimz * z / pMin.z), float arithmetic
    k = np.arange(dimz)[None, None, :]
    typ[(zz < 0)[:, :, None] & (k >= 1) & (k < bound_k[:, :, None])] = NODE_IN

    def touches(t):
        m = typ == t
        r = np.zeros(dims, bool)
        r[1:-1, 1:-1, 1:-1] = (m[:-2, 1:-1, 1:-1] | m[2:, 1:-1, 1:-1] | m[1:-1, :-2, 1:-1] | m[1:-1, 2:, 1:-1] | m[1:-1, 1:-1, :-2] | m[1:-1, 1:-1, 2:])
        return r
    typ[(typ == NODE_IN) & touches(NODE_OUT)] = NODE_BOUND                   # T = baseT, v = 0 already
    typ[(typ == NODE_OUT) & touches(NODE_BOUND)] = NODE_BOUND
    vin = [float(F(c)) for c in bc_inV]
    t_in, t_out = float(F(bc_inT)), float(F(F(2.0) - F(bc_inT)))

    def stream(i, j):
        col = np.nonzero(typ[i, j] == NODE_IN)[0]
        if not len(col):
            return
        half = (int(col[0]) + int(col[-1])) // 2
        for kk in col:
            inflow = kk < half
            typ[i, j, kk] = NODE_VALVE
            vx[i, j, kk], vy[i, j, kk], vz[i, j, kk] = (vin if inflow else [float(F(0) - F(c)) for c in vin])
            T[i, j, kk] = t_in if inflow else t_out
    for i in range(dimx):
        stream(i, dimy - 1)
    for j in range(dimy):
        stream(dimx - 1, j)
    z8 = np.zeros(dims, np.uint8)
    nodes = Nodes(dimx, dimy, dimz, dx, dy, dz, typ, z8 + BC_NOSLIP, z8 + BC_NOSLIP, vx, vy, vz, T)
    return nodes, {"bbox": tuple(float(c) for c in mn) + tuple(float(c) for c in mx), "depth": depth}
