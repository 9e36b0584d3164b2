"""The C++ host side above the C ABI that a user of the reference runs (cmc_fluid_solver_amd/host/):
Config parser, Shape2D loader, netCDF result writer, command-line driver.  CPU part: the loader against its
Python twin (itself pinned to the reference's recorded grid dims / NODE_IN counts), the writer read back with
scipy.  GPU part: the driver end to end against the Python path through the same library."""
import os
import re
import subprocess

import numpy as np
import pytest

from cmc_fluid_solver_amd import build as B
from cmc_fluid_solver_amd import capi, shape2d


@pytest.fixture(autouse=True)
def _exact_kernels(monkeypatch):
    """These tests assert bit-equality with the CPU oracle: new contexts start on the bit-exact kernels
    (FS3D_SWEEP_EXACT).  The partition kernels (the fp32 default) have their own tolerance tests in test_gpu_part.py."""
    monkeypatch.setenv("FS3D_DEFAULT_KERNEL", "4")


HERE = os.path.dirname(os.path.abspath(__file__))
INPUTS = os.path.join(HERE, "golden", "inputs")
CASES = {"box_pipe": ("box_pipe_2D_data.txt", "box_pipe_2D_config.txt"),
         "non_uniform_pipe": ("non_uniform_pipe_2D_data.txt", "non_uniform_pipe_2D_config.txt"),
         "heart_us": ("heart_us_2D_data.txt", "heart_us_2D_config.txt")}           # 10 frames (moving wall + valve)


@pytest.fixture(scope="module")
def driver(built):
    return B.build_driver()


def _grid_dump(path):
    raw = open(path, "rb").read()
    nx, ny, nz, esz = np.frombuffer(raw[:16], np.int32)
    n = nx * ny * nz
    off = 16
    out = {}
    for name in ("type", "bc_vel", "bc_temp"):
        out[name] = np.frombuffer(raw[off:off + n], np.uint8).reshape(nx, ny, nz); off += n
    dt = np.float32 if esz == 4 else np.float64
    for name in ("vx", "vy", "vz", "T"):
        out[name] = np.frombuffer(raw[off:off + n * esz], dt).reshape(nx, ny, nz); off += n * esz
    assert off == len(raw)
    return out


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("prec", ["float", "double"])
def test_cpp_loader_equals_python_loader(driver, case, prec, tmp_path):
    data, cfgf = (os.path.join(INPUTS, f) for f in CASES[case])
    dump = str(tmp_path / "grid.bin")
    args = [driver, data, str(tmp_path / "out"), cfgf, "align", "--grid-only", dump] + (["double"] if prec == "double" else [])
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
    nodes, cfg, _ = shape2d.load_case(data, cfgf, align=True)
    assert "Grid = %d x %d x %d" % nodes.shape in out
    m = re.search(r"NODE_IN points = ([0-9.]+) of total", out)
    assert float(m.group(1)) == float((nodes.type == 0).sum())
    g = _grid_dump(dump)
    assert np.array_equal(g["type"], nodes.type) and np.array_equal(g["bc_vel"], nodes.bc_vel) and np.array_equal(g["bc_temp"], nodes.bc_temp)
    dt = np.float32 if prec == "float" else np.float64
    for name in ("vx", "vy", "vz", "T"):
        assert np.array_equal(g[name], np.asarray(getattr(nodes, name), dt)), name


def test_grid_images(driver, tmp_path):
    """--grid-images: Grid3D::OutputImage, one 24-bit BMP of the node types per z-slice (Grid3D.cpp:1112-1173)."""
    data, cfgf = (os.path.join(INPUTS, f) for f in CASES["non_uniform_pipe"])
    prefix = str(tmp_path / "img")
    subprocess.run([driver, data, prefix, cfgf, "align", "--grid-images", "--grid-only", str(tmp_path / "g.bin")], check=True, capture_output=True)
    nodes, cfg, _ = shape2d.load_case(data, cfgf, align=True)
    nx, ny, nz = nodes.shape
    assert sorted(os.listdir(prefix + "_grid_3d"), key=lambda s: int(s[:-4])) == ["%d.bmp" % k for k in range(nz)]
    colour = {0: (245, 73, 69), 1: (0, 0, 0), 2: (255, 255, 255), 3: (241, 41, 212)}
    for k in (0, 1, nz // 2, nz - 1):
        raw = open(os.path.join(prefix + "_grid_3d", "%d.bmp" % k), "rb").read()
        assert raw[:2] == b"BM" and int.from_bytes(raw[10:14], "little") == 54 and int.from_bytes(raw[28:30], "little") == 24
        assert int.from_bytes(raw[18:22], "little") == ny and int.from_bytes(raw[22:26], "little") == nx
        img = np.frombuffer(raw[54:], np.uint8).reshape(nx, -1)[:, :3 * ny].reshape(nx, ny, 3)[::-1]      # stored bottom-up: x runs down
        want = np.array([colour[t] for t in nodes.type[:, :, k].ravel()], np.uint8).reshape(nx, ny, 3)
        assert np.array_equal(img, want), k


def test_box_pipe_grid_matches_the_reference_printout(driver, tmp_path):
    """the reference prints `Grid = 64 x 64 x 64` and 115248 NODE_IN points for the shipped example."""
    data, cfgf = (os.path.join(INPUTS, f) for f in CASES["box_pipe"])
    out = subprocess.run([driver, data, str(tmp_path / "o"), cfgf, "align", "--grid-only", str(tmp_path / "g.bin")],
                         check=True, capture_output=True, text=True).stdout
    assert "Grid = 64 x 64 x 64" in out and "NODE_IN points = 115248.000000 of total 262144.000000" in out


def test_config_errors_are_reported_like_the_reference(driver, tmp_path):
    bad = tmp_path / "cfg.txt"
    bad.write_text("dimension 3D\r\nin_fmt Shape2D\r\ngrid_dx 0.02\r\ngrid_dy 0.02\r\ngrid_dz 0.02\r\nout_fmt NetCDF\r\nsolver ADI\r\nout_vars 1 u\r\n")
    r = subprocess.run([driver, os.path.join(INPUTS, CASES["box_pipe"][0]), str(tmp_path / "o"), str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "cannot find depth!" in r.stderr


def test_netcdf_writer_roundtrip(tmp_path):
    from scipy.io import netcdf_file
    exe = os.path.join(HERE, "netcdf_writer_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(HERE, "netcdf_writer_test.cpp"), "-o", exe])
    path = str(tmp_path / "t_res.nc")
    assert subprocess.run([exe, path], check=True, capture_output=True, text=True).stdout.strip() == "3"
    f = netcdf_file(path, "r", mmap=False)
    assert {k: v for k, v in f.dimensions.items()} == {"x": 3, "y": 4, "z": 5, "t": None}
    assert f.Conventions == b"COARDS" and f.title == b"cmc-fluid-solver results"
    assert set(f.variables) == {"x", "y", "z", "time", "u", "w", "T"}
    np.testing.assert_allclose(f.variables["x"][:], np.float32(-0.5) + np.float32(2.0 / 3) * np.arange(3, dtype=np.float32), rtol=1e-6)
    assert f.variables["x"].units == b"metres" and list(f.variables["z"].actual_range) == [0.0, 1.0]
    np.testing.assert_array_equal(f.variables["time"][:], [0.0, 0.5, 1.0])
    assert f.variables["u"].shape == (3, 3, 4, 5) and f.variables["u"].dimensions == ("t", "x", "y", "z")
    assert f.variables["T"].units == b"tmp" and f.variables["u"].units == b"m/s" and f.variables["w"].var_desc == b"w"
    assert float(f.variables["T"].missing_value) == 99999.0
    c = np.arange(60).reshape(3, 4, 5)
    for layer in range(3):
        np.testing.assert_array_equal(f.variables["u"][layer], 100.0 * layer + c)
        np.testing.assert_array_equal(f.variables["w"][layer], 0.5 * c - layer)
        T = 1.0 + 0.001 * c + layer
        T.flat[7] = 99999.0
        np.testing.assert_array_equal(f.variables["T"][layer], T)
    f.close()


def test_netcdf_writer_depth_variable_and_degree_units(tmp_path):
    """SeaNetCDF runs: x / y in degrees and the fixed-size variable `d` (x, y) beside the records (IO.h:168-171, 188-196, 270-273)."""
    from scipy.io import netcdf_file
    exe = os.path.join(HERE, "netcdf_writer_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(HERE, "netcdf_writer_test.cpp"), "-o", exe])
    path = str(tmp_path / "sea_res.nc")
    assert subprocess.run([exe, path, "sea"], check=True, capture_output=True, text=True).stdout.strip() == "3"
    f = netcdf_file(path, "r", mmap=False)
    assert set(f.variables) == {"x", "y", "z", "time", "u", "d"}
    assert f.variables["x"].units == b"degree_north" and f.variables["y"].units == b"degree_east" and f.variables["z"].units == b"metres"
    d = f.variables["d"]
    assert d.dimensions == ("x", "y") and d.units == b"m" and d.long_name == b"depth" and float(d.missing_value) == 99999.0
    np.testing.assert_array_equal(d[:], (-10.0 * np.arange(12) + 0.5).reshape(3, 4).astype(np.float32))
    assert f.variables["u"].shape == (3, 3, 4, 5)
    np.testing.assert_array_equal(f.variables["u"][1], 100.0 + np.arange(60).reshape(3, 4, 5))
    np.testing.assert_array_equal(f.variables["time"][:], [0.0, 0.5, 1.0])
    f.close()


@pytest.mark.gpu
def test_driver_runs_the_shipped_example(driver, tmp_path):
    """fs3d_run on data/3D box_pipe (the fixture copy): err values and the result layers equal the Python path's."""
    from scipy.io import netcdf_file
    data, cfgf = (os.path.join(INPUTS, f) for f in CASES["box_pipe"])
    prefix = str(tmp_path / "box")
    nsteps = 21
    out = subprocess.run([driver, data, prefix, cfgf, "align", "GPU", "--steps", str(nsteps)], check=True, capture_output=True, text=True).stdout
    errs = [float(x) for x in re.findall(r"err = ([0-9.]+),", out)]
    assert len(errs) == nsteps
    nodes, cfg, dt = shape2d.load_case(data, cfgf, align=True)
    s = capi.Solver(nodes, capi.fluid_params(np.float32, cfg.Re, cfg.Pr, cfg.lam), np.float32)
    layers, ref_err = [], []
    for i in range(nsteps):
        s.UpdateBoundaries()
        e = s.TimeStep(np.float32(dt), cfg.num_global, cfg.num_local, i % 10 == 0)
        ref_err.append(e)
        if i % cfg.out_time_steps == 0:
            layers.append(s.GetLayer((cfg.outdimx, cfg.outdimy, cfg.outdimz)))
    np.testing.assert_allclose(errs, [float("%.8f" % e) for e in ref_err], atol=1e-12)
    f = netcdf_file(prefix + "_res.nc", "r", mmap=False)
    assert f.variables["u"].shape == (len(layers), cfg.outdimx, cfg.outdimy, cfg.outdimz)
    for r, (V, T) in enumerate(layers):
        for c, name in enumerate("uvw"):
            np.testing.assert_array_equal(f.variables[name][r], V[..., c].astype(np.float64))
        np.testing.assert_array_equal(f.variables["T"][r], T)
    np.testing.assert_allclose(f.variables["time"][:], np.arange(len(layers)) * dt * cfg.out_time_steps)
    f.close()


@pytest.mark.gpu
def test_driver_result_file_equals_the_reference(driver, tmp_path, monkeypatch):
    """f2 held to the REFERENCE, not to this library: fs3d_run (bit-exact kernels) on the shipped box_pipe case, all 100 steps; every
    record of u, v, w, T in `_res.nc` equals the result layer the reference's own Solver3D::GetLayer handed to its NetCDF writer at
    that step (tests/golden/ref_box_pipe_f32.npz: sha256 of all 10 layers, the first and the last one in full), and every printed
    `err` equals the reference's print."""
    from scipy.io import netcdf_file
    import refgolden as RG
    fx = RG.Fixture("box_pipe", "f32")
    m = fx.meta
    data = os.path.join(INPUTS, m["data"])
    cfgf = tmp_path / "config.txt"
    cfgf.write_text(m["config_text"])
    prefix = str(tmp_path / "box")
    monkeypatch.setenv("FS3D_DEFAULT_KERNEL", "4")
    out = subprocess.run([driver, data, prefix, str(cfgf), "align", "GPU"], check=True, capture_output=True, text=True).stdout
    assert "Sweep kernels:" in out and "part" not in out.split("Sweep kernels:")[1].splitlines()[0]
    errs = re.findall(r"err = ([0-9.]+),", out)
    assert errs == ["%.8f" % e for e in m["err_trace"]]
    f = netcdf_file(prefix + "_res.nc", "r", mmap=False)
    steps = m["layer_steps"]
    assert f.variables["u"].shape[0] == len(steps) == 10
    for r, st in enumerate(steps):
        V = np.stack([f.variables[nm][r] for nm in "uvw"], axis=-1).astype(np.float32)
        T = np.ascontiguousarray(f.variables["T"][r], np.float64)
        assert RG.sha(V) == m["layer_sha"][str(st)]["outV"] and RG.sha(T) == m["layer_sha"][str(st)]["outT"], "record %d (step %d)" % (r, st)
        if "outV_step%d" % st in fx.z:
            np.testing.assert_array_equal(V, fx.z["outV_step%d" % st])
            np.testing.assert_array_equal(T, fx.z["outT_step%d" % st])
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [["GPU"], ["GPU", "2", "--same-device"]])
def test_driver_runs_a_multi_frame_input(driver, mode, tmp_path):
    """The 10-frame heart_us input: dt = cycle length / (frames * time_steps), the substep counter restarts at every frame
    (which moves the error evaluation and the output cadence), the geometry stays frame 0's (FluidSolver3D.cpp:193-266).
    The driver's prints and result layers equal the Python path's, which runs shape2d.time_loop over the same library."""
    from scipy.io import netcdf_file
    data, cfgf = (os.path.join(INPUTS, f) for f in CASES["heart_us"])
    prefix = str(tmp_path / "heart")
    out = subprocess.run([driver, data, prefix, cfgf, "align"] + mode, check=True, capture_output=True, text=True).stdout
    nodes, cfg, dt = shape2d.load_case(data, cfgf, align=True)
    loop = list(shape2d.time_loop(cfg.grid2d, cfg))
    assert len(loop) in (29, 30) and {fr for _, _, fr, _, _ in loop} == set(range(10))
    assert [(int(a), int(b)) for a, b in re.findall(r"frame (\d+)\tsubstep (\d+)", out)] == [(fr, i) for _, i, fr, _, _ in loop]
    s = capi.Solver(nodes, capi.fluid_params(np.float32, cfg.Re, cfg.Pr, cfg.lam), np.float32)
    layers, ref_err = [], []
    for t, i, fr, with_err, output in loop:
        s.UpdateBoundaries()
        e = s.TimeStep(np.float32(dt), cfg.num_global, cfg.num_local, with_err)
        ref_err.append(e if with_err else ref_err[-1])
        if output:
            layers.append(s.GetLayer((cfg.outdimx, cfg.outdimy, cfg.outdimz)))
    errs = [float(x) for x in re.findall(r"err = ([0-9.]+),", out)]
    np.testing.assert_allclose(errs, [float("%.8f" % e) for e in ref_err], atol=1e-12)
    f = netcdf_file(prefix + "_res.nc", "r", mmap=False)
    assert f.variables["u"].shape == (len(layers), cfg.outdimx, cfg.outdimy, cfg.outdimz)
    for r, (V, T) in enumerate(layers):
        for c, name in enumerate("uvw"):
            np.testing.assert_array_equal(f.variables[name][r], V[..., c].astype(np.float64))
        np.testing.assert_array_equal(f.variables["T"][r], T)
    f.close()
    assert np.abs(layers[-1][0][layers[-1][0] < 9e4]).max() > 1e-3           # the moving wall drives a flow


@pytest.mark.gpu
@pytest.mark.parametrize("nslabs", [2, 3])
def test_driver_gpu_n_mode_equals_single_gpu(driver, nslabs, tmp_path):
    """`GPU n` (the reference's one-process multi-GPU mode): n x-slabs, one host thread each, in-process transport; all
    slabs on device 0 here (--same-device).  err prints and result file equal the single-GPU run's."""
    from scipy.io import netcdf_file
    data, cfgf = (os.path.join(INPUTS, f) for f in CASES["non_uniform_pipe"])
    outs = {}
    for tag, extra in (("one", ["GPU"]), ("slabs", ["GPU", str(nslabs), "--same-device"])):
        prefix = str(tmp_path / tag)
        out = subprocess.run([driver, data, prefix, cfgf, "align"] + extra + ["--steps", "21"], check=True, capture_output=True, text=True).stdout
        f = netcdf_file(prefix + "_res.nc", "r", mmap=False)
        outs[tag] = ([float(x) for x in re.findall(r"err = ([0-9.]+),", out)],
                     {v: np.array(f.variables[v][:]) for v in ("u", "v", "w", "T", "time")})
        f.close()
    assert len(outs["one"][0]) == 21 and outs["one"][0] == outs["slabs"][0]
    for v in ("u", "v", "w", "T", "time"):
        assert outs["one"][1][v].shape[0] == 3
        np.testing.assert_array_equal(outs["one"][1][v], outs["slabs"][1][v])


@pytest.mark.gpu
def test_driver_prints_the_reference_profiler_vocabulary(driver, tmp_path):
    """The closing table of fs3d_run carries the reference Profiler's event names (Common/Profiler.h:90-133; StartEvent/StopEvent
    sites AdiSolver3D.cpp:297-367, 555-680), sorted by total time, and names the sweep kernels that ran."""
    data, conf = [os.path.join(INPUTS, f) for f in CASES["box_pipe"]]
    out = subprocess.run([driver, data, str(tmp_path / "pp"), conf, "align", "GPU", "--steps", "12"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    tab = out.stdout[out.stdout.index("Profiling data node(0):"):]
    for name in ("SolveSegments_X", "SolveSegments_Y", "SolveSegments_Z", "CopyLayer", "EvalDivError", "UpdateBoundaries", "CreateSegments", "Overall"):
        assert name in tab, tab
    rows = [l.split() for l in tab.splitlines()[2:] if l.strip() and not l.strip().startswith(("Overall", "Sweep", "12 steps"))]
    totals = [float(r[1]) for r in rows if len(r) == 4]
    assert totals == sorted(totals, reverse=True)
    assert "Sweep kernels:" in out.stdout
