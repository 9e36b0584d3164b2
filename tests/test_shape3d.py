"""Shape3D input surface (second loader): triangle meshes rasterised into the Node array.
The C++ loader (host/Shape3D.h, through fs3d_run --grid-only) is compared with its Python twin (shape3d.py) cell for cell, and both
with hand-derived properties of small closed meshes written by the test; the twin itself is held to the reference's own Grid3D in
tests/test_ref_golden.py (shipped box_pipe_3D / tetra meshes, this file's two-frame icosphere)."""
import os
import re
import subprocess

import numpy as np
import pytest

from cmc_fluid_solver_amd import build as B
from cmc_fluid_solver_amd import capi, grids, shape3d


def icosphere(radius, centre, subdiv=1):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m)); cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v) * radius + np.array(centre), np.array(f)


CONFIG = """dimension 3D
in_fmt Shape3D
Re 200.0
Pr 0.72
lambda 1.4
bc_type NoSlip
grid_dx 0.001
grid_dy 0.001
grid_dz 0.001
frame_time 0.4
cycles 1
time_steps 4
out_time_steps 2
out_gridx 16
out_gridy 16
out_gridz 16
out_fmt NetCDF
out_vars 4 u v w T
solver ADI
num_global 2
num_local 1
"""


@pytest.fixture(scope="module")
def driver(built):
    return B.build_driver()


@pytest.fixture(scope="module")
def sphere_case(tmp_path_factory):
    d = tmp_path_factory.mktemp("shape3d")
    v, f = icosphere(11.0, (40.0, 42.0, 45.0), subdiv=1)            # millimetres; 80 faces, none axis-aligned
    v2 = v + np.array([1.0, 0.0, 0.5])                               # a second frame: the sphere moved
    data, cfg = str(d / "sphere_3D_data.txt"), str(d / "sphere_3D_config.txt")
    shape3d.write_mesh(data, [(v, f), (v2, f)])
    open(cfg, "w").write(CONFIG)
    return data, cfg, d


def _grid_dump(path):
    raw = open(path, "rb").read()
    nx, ny, nz, esz = np.frombuffer(raw[:16], np.int32)
    n = nx * ny * nz
    out, off = {}, 16
    for name in ("type", "bc_vel", "bc_temp"):
        out[name] = np.frombuffer(raw[off:off + n], np.uint8).reshape(nx, ny, nz); off += n
    dt = np.float32 if esz == 4 else np.float64
    for name in ("vx", "vy", "vz", "T"):
        out[name] = np.frombuffer(raw[off:off + n * esz], dt).reshape(nx, ny, nz); off += n * esz
    return out


def test_python_rasteriser_properties(sphere_case):
    """A closed sphere: a NODE_BOUND shell, NODE_IN inside it, NODE_OUT around it; the bounding box covers both frames."""
    data, _, _ = sphere_case
    nodes, sh = shape3d.load_shape3d(data, 0.001, 0.001, 0.001, align=False)
    assert len(sh.frames) == 2
    ext = np.array(sh.bbox[3:]) - np.array(sh.bbox[:3])
    np.testing.assert_allclose(ext, np.array([23.0, 22.0, 22.5]) * 1e-3 * 1.04, rtol=2e-2)     # both frames + 2 % padding per side
    assert nodes.shape == tuple(int(np.ceil(float(np.float32(e)) / 0.001)) + 1 for e in ext)
    t = nodes.type
    n_in, n_b = int((t == grids.NODE_IN).sum()), int((t == grids.NODE_BOUND).sum())
    r = 11.0                                                         # cells: 1 mm
    assert 0.75 * 4 / 3 * np.pi * r ** 3 < n_in < 4 / 3 * np.pi * r ** 3            # the inscribed polyhedron is smaller than the sphere
    assert 0.7 * 4 * np.pi * r ** 2 < n_b < 2.5 * 4 * np.pi * r ** 2
    # every NODE_IN cell is enclosed: no NODE_IN cell touches a NODE_OUT cell (the shell has no holes)
    inn, out = t == grids.NODE_IN, t == grids.NODE_OUT
    for ax in range(3):
        a = [slice(None)] * 3; b = [slice(None)] * 3
        a[ax], b[ax] = slice(0, -1), slice(1, None)
        assert not (inn[tuple(a)] & out[tuple(b)]).any() and not (out[tuple(a)] & inn[tuple(b)]).any()
    c = tuple(int(round((p * 1e-3 - o) / 0.001)) for p, o in zip((40.0, 42.0, 45.0), sh.bbox[:3]))
    assert t[c] == grids.NODE_IN and t[0, 0, 0] == grids.NODE_OUT
    assert (nodes.T[t == grids.NODE_BOUND] == 0).all() and (nodes.T[t != grids.NODE_BOUND] == 1).all()
    # the second frame is another geometry
    first = sh.type.copy()
    sh.prepare(1.0 / 75 + 0.006)
    assert (sh.type != first).any()


@pytest.mark.parametrize("prec", ["float", "double"])
@pytest.mark.parametrize("align", [True, False])
def test_cpp_loader_equals_python_loader(driver, sphere_case, prec, align):
    data, cfg, d = sphere_case
    dump = str(d / ("grid_%s_%d.bin" % (prec, align)))
    args = [driver, data, str(d / "out"), cfg] + (["align"] if align else []) + ["--grid-only", dump] + (["double"] if prec == "double" else [])
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
    nodes, sh = shape3d.load_shape3d(data, float(np.float32(0.001)), float(np.float32(0.001)), float(np.float32(0.001)), align=align)
    assert "Geometry: 3D polygons" in out and "Grid = %d x %d x %d" % nodes.shape in out
    m = re.search(r"NODE_IN points = ([0-9.]+) of total", out)
    assert float(m.group(1)) == float((nodes.type == grids.NODE_IN).sum())
    g = _grid_dump(dump)
    assert np.array_equal(g["type"], nodes.type)
    assert not g["bc_vel"].any() and not g["bc_temp"].any() and not g["vx"].any()
    assert np.array_equal(g["T"], np.asarray(nodes.T, g["T"].dtype))


def test_driver_rejects_a_shape3d_run_without_frame_time(driver, sphere_case):
    data, cfg, d = sphere_case
    bad = str(d / "noframe.txt")
    open(bad, "w").write(CONFIG.replace("frame_time 0.4\n", ""))
    out = subprocess.run([driver, data, str(d / "o"), bad], capture_output=True, text=True)
    assert out.returncode != 0 and "frame time" in (out.stdout + out.stderr)


@pytest.mark.gpu
def test_driver_runs_a_shape3d_input(driver, sphere_case, monkeypatch):
    """fs3d_run on the sphere: dt = frame_time / (frames * time_steps), frame 0 throughout; prints and result layers equal the
    Python path's through the same library."""
    from scipy.io import netcdf_file
    monkeypatch.setenv("FS3D_DEFAULT_KERNEL", "4")
    data, cfg, d = sphere_case
    prefix = str(d / "run")
    out = subprocess.run([driver, data, prefix, cfg, "align", "GPU"], check=True, capture_output=True, text=True).stdout
    nodes, sh = shape3d.load_shape3d(data, float(np.float32(0.001)), float(np.float32(0.001)), float(np.float32(0.001)), align=True)
    from cmc_fluid_solver_amd import shape2d
    c = shape2d.Config(cfg)                                      # frame_time goes through a float, like every real number of a config
    dt = c.frame_time / (2 * c.time_steps)
    final = c.frame_time * c.cycles
    nsteps = len(re.findall(r"substep (\d+)", out))
    assert nsteps in (7, 8) and set(re.findall(r"frame (\d+)\tsubstep", out)) == {"0"}
    s = capi.Solver(nodes, capi.fluid_params(np.float32, c.Re, c.Pr, c.lam), np.float32)
    layers, errs = [], []
    t = dt
    for i in range(nsteps):
        s.UpdateBoundaries()
        errs.append(s.TimeStep(np.float32(dt), 2, 1, i % 10 == 0 or t + dt >= final))
        if i % 2 == 0:
            layers.append(s.GetLayer((16, 16, 16)))
        t += dt
    f = netcdf_file(prefix + "_res.nc", "r", mmap=False)
    assert f.variables["T"].shape == (len(layers), 16, 16, 16)
    for r, (V, T) in enumerate(layers):
        np.testing.assert_array_equal(f.variables["T"][r], T)
        np.testing.assert_array_equal(f.variables["u"][r], V[..., 0].astype(np.float64))
    np.testing.assert_allclose(f.variables["x"].actual_range, [sh.bbox[0], sh.bbox[3]], rtol=1e-6)
    np.testing.assert_allclose(f.variables["z"].actual_range, [sh.bbox[2], sh.bbox[5]], rtol=1e-6)
    f.close()
    assert np.isfinite(layers[-1][1][layers[-1][1] < 9e4]).all()
