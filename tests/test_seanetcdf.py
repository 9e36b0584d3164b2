"""SeaNetCDF input surface (third loader): the reference's white_sea depth map, a netCDF-4 (HDF5) file,
read without libnetcdf (host/Hdf5Min.h + zlib, Python twin hdf5_min.py) and turned into the Node array (host/SeaNetCDF.h, twin
seanetcdf.py).  Parity unpinned: the survey could not run the reference on this input; C++ against the twin cell for cell, both
against properties derived by hand from Grid3D::Prepare3D_NetCDF (Grid3D.cpp:968-1075)."""
import os
import re
import subprocess

import numpy as np
import pytest

from cmc_fluid_solver_amd import build as B
from cmc_fluid_solver_amd import capi, grids, hdf5_min, seanetcdf, shape2d

INP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")
DATA = os.path.join(INP, "white_sea_data.nc")          # the reference's data/3D/large_tests/white_sea input, byte for byte
CONF = os.path.join(INP, "white_sea_config.txt")


def f32(x):
    return float(np.float32(x))


@pytest.fixture(scope="module")
def driver(built):
    return B.build_driver()


def _load(align=True):
    cfg = shape2d.Config(CONF)
    assert cfg.in_fmt == "SeaNetCDF" and cfg.out_vars == ["u", "v", "d"] and cfg.frame_time == 100.0
    return cfg, seanetcdf.load_seanetcdf(DATA, cfg.dx, cfg.dy, cfg.dz, cfg.baseT, cfg.bc_inV, cfg.bc_inT, align=align)


def test_hdf5_reader_on_the_white_sea_file():
    f = hdf5_min.Hdf5File(DATA)
    assert set(f.links) == {"_lat_subset", "_lon_subset", "z"}
    lat, lon, z = f.read("_lat_subset"), f.read("_lon_subset"), f.read("z")
    assert lat.shape == (301,) and lon.shape == (722,) and z.shape == (301, 722) and z.dtype == np.float32
    np.testing.assert_allclose(np.diff(lat), 1.0 / 60, rtol=1e-6)            # one arc minute
    np.testing.assert_allclose(np.diff(lon), 1.0 / 60, rtol=1e-6)
    assert abs(lat[0] - 62.9916667) < 1e-6 and abs(lon[-1] - 43.025) < 1e-6
    assert z.min() == -233.0 and z.max() == 1113.0 and 0.22 < (z < 0).mean() < 0.24      # the White Sea: a quarter of the map
    with pytest.raises(hdf5_min.Hdf5Error):
        f.read("no_such_variable")
    with pytest.raises(hdf5_min.Hdf5Error):
        hdf5_min.Hdf5File(os.path.join(INP, "box_pipe_2D_data.txt"))


def test_grid_from_the_depth_map():
    cfg, (nodes, info) = _load(align=False)
    # dims: ceil(extent / step) + 1 with one extra cell of depth below the deepest point
    assert nodes.shape == (144, 82, 80)
    assert info["bbox"][2] == -236.0 and info["bbox"][5] == 0.0
    t = nodes.type
    # the water column of a sea point: NODE_OUT at k = 0, then sea / skin cells up to dimz * z / zmin
    zz = seanetcdf.resample_depths(info["depth"], 144, 82)
    i, j = np.unravel_index(np.argmin(zz), zz.shape)
    col = t[i, j]
    assert col[0] == grids.NODE_OUT and (col[3:70] == grids.NODE_IN).all() and col[79] == grids.NODE_OUT
    assert (t[zz >= 0] != grids.NODE_IN).all()                                  # land: nothing but NODE_OUT and the skin
    # no sea cell inside the grid touches NODE_OUT: the skin is closed
    inn, out = t == grids.NODE_IN, t == grids.NODE_OUT
    core = (slice(1, -1),) * 3
    for ax in range(3):
        for sh in (-1, 1):
            assert not (inn[core] & np.roll(out, sh, axis=ax)[core]).any()
    # streams: valves only on the faces y = max and x = max; the upper half flows in with bc_initv / bc_initT, the lower half out
    v = t == grids.NODE_VALVE
    assert v.any() and not v[:-1, :-1, :].any()
    vin = nodes.vx[v] == f32(-0.25)
    assert vin.any() and (~vin).any()
    assert (nodes.vy[v][vin] == f32(-0.25)).all() and (nodes.vx[v][~vin] == f32(0.25)).all() and (nodes.vz[v] == 0).all()
    assert (nodes.T[v][vin] == f32(1.01)).all() and np.allclose(nodes.T[v][~vin], 2.0 - 1.01, atol=1e-6)
    assert (nodes.T[~v] == 1.0).all() and not nodes.bc_vel.any() and not nodes.bc_temp.any()


@pytest.mark.parametrize("prec", ["float", "double"])
@pytest.mark.parametrize("align", [True, False])
def test_cpp_loader_equals_python_loader(driver, prec, align, tmp_path):
    dump = str(tmp_path / "grid.bin")
    args = [driver, DATA, str(tmp_path / "out"), CONF] + (["align"] if align else []) + ["--grid-only", dump] + (["double"] if prec == "double" else [])
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
    cfg, (nodes, info) = _load(align)
    assert "Geometry: depths from NetCDF" in out and "Grid = %d x %d x %d" % nodes.shape in out
    assert float(re.search(r"NODE_IN points = ([0-9.]+) of total", out).group(1)) == float((nodes.type == grids.NODE_IN).sum())
    raw = open(dump, "rb").read()
    nx, ny, nz, esz = np.frombuffer(raw[:16], np.int32)
    n, off = nx * ny * nz, 16
    for name in ("type", "bc_vel", "bc_temp"):
        a = np.frombuffer(raw[off:off + n], np.uint8).reshape(nx, ny, nz); off += n
        assert np.array_equal(a, getattr(nodes, name)), name
    dt = np.float32 if esz == 4 else np.float64
    for name in ("vx", "vy", "vz", "T"):
        a = np.frombuffer(raw[off:off + n * esz], dt).reshape(nx, ny, nz); off += n * esz
        assert np.array_equal(a, np.asarray(getattr(nodes, name), dt)), name


@pytest.mark.gpu
def test_driver_runs_the_white_sea_input(driver, tmp_path, monkeypatch):
    """fs3d_run on the shipped white_sea case (its own config, unchanged): dt = frame_time / time_steps, degree units, the depth
    variable `d`; err prints and result layers equal the Python path's through the same library."""
    from scipy.io import netcdf_file
    monkeypatch.setenv("FS3D_DEFAULT_KERNEL", "4")
    prefix = str(tmp_path / "sea")
    nsteps = 12
    out = subprocess.run([driver, DATA, prefix, CONF, "align", "GPU", "--steps", str(nsteps)], check=True, capture_output=True, text=True).stdout
    cfg, (nodes, info) = _load(True)
    assert nodes.shape == (160, 96, 96)
    dt = cfg.frame_time / (1 * cfg.time_steps)
    s = capi.Solver(nodes, capi.fluid_params(np.float32, cfg.Re, cfg.Pr, cfg.lam), np.float32)
    errs, first = [], None
    for i in range(nsteps):
        s.UpdateBoundaries()
        errs.append(s.TimeStep(np.float32(dt), cfg.num_global, cfg.num_local, i % 10 == 0))
        if i == 0:
            first = s.GetLayer((cfg.outdimx, cfg.outdimy, cfg.outdimz))
    printed = [float(x) for x in re.findall(r"err = ([0-9.]+),", out)]
    ref = []
    for i, e in enumerate(errs):
        ref.append(e if i % 10 == 0 else ref[-1])
    np.testing.assert_allclose(printed, [float("%.8f" % e) for e in ref], atol=1e-12)
    f = netcdf_file(prefix + "_res.nc", "r", mmap=False)
    assert set(f.variables) == {"x", "y", "z", "time", "u", "v", "d"}
    assert f.variables["x"].units == b"degree_north" and f.variables["y"].units == b"degree_east"
    np.testing.assert_array_equal(f.variables["d"][:], seanetcdf.resample_depths(info["depth"], cfg.outdimx, cfg.outdimy))
    assert f.variables["u"].shape == (1, cfg.outdimx, cfg.outdimy, cfg.outdimz)
    np.testing.assert_array_equal(f.variables["u"][0], first[0][..., 0].astype(np.float64))
    np.testing.assert_array_equal(f.variables["v"][0], first[0][..., 1].astype(np.float64))
    f.close()
    u = first[0][..., 0]
    assert np.isfinite(u[u < 9e4]).all() and np.abs(u[u < 9e4]).max() > 0.01          # the inflow drives the basin


def test_a_btree_node_that_names_itself_is_refused(driver, tmp_path):
    """the chunk B-tree's level byte and child addresses come from the file.  A copy of the white_sea file whose TREE node
    claims level 1 and lists ITSELF as its first child must be refused by both readers (it used to recurse until the stack ran out)."""
    raw = bytearray(open(DATA, "rb").read())
    p = raw.find(b"TREE")
    assert p > 0 and raw[p + 4] == 1 and raw[p + 5] == 0
    rank = 2
    ks = 8 + 8 * (rank + 1)
    raw[p + 5] = 1                                                   # "internal node"
    raw[p + 24 + ks:p + 24 + ks + 8] = int(p).to_bytes(8, "little")  # first child = this node (file base address is 0)
    bad = tmp_path / "self.nc"
    bad.write_bytes(bytes(raw))
    with pytest.raises(hdf5_min.Hdf5Error):
        hdf5_min.Hdf5File(str(bad)).read("z")
    r = subprocess.run([driver, str(bad), str(tmp_path / "out"), CONF, "--grid-only", str(tmp_path / "g.bin")], capture_output=True, text=True)
    assert r.returncode not in (0, -11, 139) and "HDF5" in (r.stderr + r.stdout), (r.returncode, r.stderr[-300:])


def test_hdf5_reader_equals_the_real_hdf5_library():
    """(r3) The own minimal HDF5 reader against the REAL library: tests/golden/white_sea_hdf5.npz holds every dataset of the reference's
    white_sea_data.nc as `h5dump -b` of libhdf5 1.10 read it in the build container (tests/golden/make_hdf5_golden.py).  Bit for bit.
    (The C++ reader is tied to this one cell for cell through the grids they produce: test_cpp_loader_equals_python_loader.)"""
    z = np.load(os.path.join(os.path.dirname(INP), "white_sea_hdf5.npz"))
    f = hdf5_min.Hdf5File(DATA)
    names = [k for k in z.files if not k.endswith("_sha256")]
    assert "z" in names
    for n in names:
        a = np.ascontiguousarray(f.read(n)).ravel()
        assert a.size == z[n].size and np.array_equal(a.astype(z[n].dtype), z[n]), n
        if a.dtype == z[n].dtype:
            import hashlib
            assert hashlib.sha256(a.tobytes()).hexdigest() == str(z[n + "_sha256"])
