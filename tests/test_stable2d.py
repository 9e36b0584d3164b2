"""The reference's 2D path as CPU plumbing (BASELINE configs[0]: a 128 x 128 lid-driven cavity through the
Stable solver -- explicit advection/diffusion + the Gauss-Seidel pressure projection, the reference's only Poisson solve).
Pinned to the reference (r3): tests/golden/ref2d_*.npz hold fields of the reference's own Grid2D + StableSolver2D objects (built as
they lie, oracle/ref_harness_2d.cpp) on the authored cavity at 54 x 54 and 128 x 128 and on the 10-frame heart_us outline with moving
walls; the C++ solver (host/Stable2D.h through fs2d_run) and its Python twin (stable2d.py) equal them bit for bit (last tests of
this file).  Also: C++ against the twin, and properties of the authored 128 x 128 case."""
import os
import re
import subprocess

import numpy as np
import pytest

from cmc_fluid_solver_amd import build as B
from cmc_fluid_solver_amd import grids, shape2d, stable2d

INP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs")
DATA = os.path.join(INP, "cavity_2D_data.txt")
CONF = os.path.join(INP, "cavity_2D_config.txt")


@pytest.fixture(scope="module")
def driver2d():
    return B.build_driver2d()


def _dump(path):
    raw = open(path, "rb").read()
    nx, ny = np.frombuffer(raw[:8], np.int32)
    a = np.frombuffer(raw[8:], np.float32).reshape(3, nx, ny)
    return a[0], a[1], a[2]


def test_cpp_solver_equals_python_twin(driver2d, tmp_path):
    cfgp = str(tmp_path / "small.txt")
    open(cfgp, "w").write(open(CONF).read().replace("0.0082", "0.0199").replace("time_steps \t1000", "time_steps \t200").replace("out_time_steps \t100", "out_time_steps \t1")
                          .replace("out_gridx\t64", "out_gridx\t27").replace("out_gridy \t64", "out_gridy \t27").replace("num_global \t2", "num_global \t1"))
    cfg = shape2d.Config(cfgp)
    assert cfg.problem_dim == "2D" and cfg.solver == "Stable" and cfg.time_steps == 200 and cfg.num_global == 1
    out, dump = str(tmp_path / "cav.cdl"), str(tmp_path / "cav.bin")
    r = subprocess.run([driver2d, DATA, out, cfgp, "--steps", "2", "--dump", dump], check=True, capture_output=True, text=True).stdout
    g = shape2d.Grid2D(shape2d.parse_shape2d(open(DATA).read()), cfg.dx, cfg.dy, cfg.baseT, False)
    assert "%f,%f,%i,%i,1" % (cfg.dx, cfg.dy, g.dimx, g.dimy) in r and (g.dimx, g.dimy) == (54, 54)
    s = stable2d.Stable2D(g, np.float32(cfg.viscosity / cfg.density))
    dt = g.cycle_length() / (1 * cfg.time_steps)
    sweeps, t = 0, dt
    for i in range(2):
        g.prepare(t); s.update_boundaries(); s.time_step(dt, cfg.num_global, cfg.num_local)
        sweeps += s.poisson_sweeps; t += dt
    assert "2 steps" in r and "(%d Poisson sweeps)" % sweeps in r
    u, v, T = _dump(dump)
    assert np.array_equal(u, s.next[0]) and np.array_equal(v, s.next[1]) and np.array_equal(T, s.next[2])
    # the CDL text: header, axes, and the u component of both layers with 3 decimals
    txt = open(out).read()
    assert txt.startswith("netcdf 2d_scalar_time_array {") and "\tx = 27 ;" in txt and "double u(time, x, y) ;" in txt
    body = txt[txt.index("u = \n") + 5:]
    vals = np.array([float(x) for x in re.findall(r"-?\d+\.\d{3}", body)])
    assert len(vals) == 2 * 27 * 27
    np.testing.assert_allclose(vals[27 * 27:], s.get_layer(27, 27)[0].ravel(), atol=5.1e-4)


def test_cavity_128(driver2d, tmp_path):
    """The authored 128 x 128 lid-driven cavity (Re 100): grid as configured, two steps; the lid drags the fluid below it, the walls
    stay at rest, the projected field is (nearly) divergence-free."""
    cfg = shape2d.Config(CONF)
    out, dump = str(tmp_path / "cav.cdl"), str(tmp_path / "cav.bin")
    r = subprocess.run([driver2d, DATA, out, CONF, "--steps", "2", "--dump", dump], check=True, capture_output=True, text=True, timeout=600).stdout
    assert "%f,%f,128,128,1" % (cfg.dx, cfg.dy) in r and "dt = 0.001000" in r
    errs = [float(x) for x in re.findall(r"err = ([0-9.]+),", r)]
    assert len(errs) == 2 and all(e < 0.1 for e in errs)
    u, v, T = _dump(dump)
    g = shape2d.Grid2D(shape2d.parse_shape2d(open(DATA).read()), cfg.dx, cfg.dy, cfg.baseT, False)
    assert (g.dimx, g.dimy) == (128, 128)
    lid = g.cell == grids.NODE_VALVE
    wall = g.cell == grids.NODE_BOUND
    assert lid.sum() > 100 and (u[lid] == 1.0).all() and (u[wall] == 0).all() and (v[wall] == 0).all()
    jl = np.nonzero(lid.any(axis=0))[0].min()                      # the lid's row
    below = u[40:90, jl - 1]
    assert (below > 1e-3).all() and (below < 1.0).all()            # momentum diffuses down from the lid
    assert abs(u[64, 20]) < 1e-3                                   # the bottom of the cavity has not moved yet
    inn = g.cell == grids.NODE_IN
    div = np.zeros_like(u)
    div[1:-1, 1:-1] = (u[2:, 1:-1] - u[:-2, 1:-1]) / (2 * cfg.dx) + (v[1:-1, 2:] - v[1:-1, :-2]) / (2 * cfg.dy)
    core = inn.copy(); core[:, jl - 3:] = False
    assert np.abs(div[core]).max() < 0.5 and np.abs(T[inn] - 1.0).max() == 0


# ---- held to the REFERENCE: tests/golden/ref2d_*.npz are outputs of the reference's own Grid2D + StableSolver2D objects
# (oracle/ref_harness_2d.cpp over the reference's translation units compiled where they lie; tests/golden/make_ref_golden.py)
def _fx2d(name):
    import json
    z = np.load(os.path.join(os.path.dirname(INP), "ref2d_%s.npz" % name))
    return z, json.loads(str(z["meta"]))


@pytest.mark.parametrize("name,upto", [("cavity54", 2)])
def test_python_solver_equals_the_reference(name, upto, tmp_path):
    """Grid2D (every step's node types: moving walls in heart2d) and the Stable solver's U, V, T after every dumped step, bit for bit
    (the Python twin is slow: the 128 x 128 cavity and heart2d -- moving walls -- are held by the C++ solver below, and
    test_cpp_solver_equals_python_twin ties the two together)."""
    z, m = _fx2d(name)
    m["steps"] = [s_ for s_ in m["steps"] if s_ <= upto]
    cfgp = str(tmp_path / "c.txt")
    open(cfgp, "w").write(m["config_text"])
    cfg = shape2d.Config(cfgp)
    g = shape2d.Grid2D(shape2d.parse_shape2d(open(os.path.join(INP, m["data"])).read()), cfg.dx, cfg.dy, cfg.baseT, False)
    assert (g.dimx, g.dimy) == tuple(m["dims"]) and g.num_frames == m["frames"] and g.cycle_length() == m["cycle_length"]
    s = stable2d.Stable2D(g, np.float32(cfg.viscosity / cfg.density))
    assert float(np.float32(cfg.viscosity / cfg.density)) == m["v_vis"]
    dt = g.cycle_length() / (g.num_frames * cfg.time_steps)
    assert dt == m["dt"]
    t = dt
    for step in range(1, max(m["steps"]) + 1):
        g.prepare(t); s.update_boundaries(); s.time_step(dt, cfg.num_global, cfg.num_local); t += dt
        assert "%.4f" % s.err == "%.4f" % m["err_trace"][step - 1]
        if step in m["steps"]:
            assert np.array_equal(g.cell, z["type_step%d" % step]), "node types at step %d" % step
            for a, v in zip(s.next, "UVT"):
                assert np.array_equal(a, z["%s_step%d" % (v, step)]), "%s after step %d" % (v, step)


@pytest.mark.parametrize("name", ["cavity54", "cavity128", "heart2d"])
def test_cpp_solver_equals_the_reference(driver2d, name, tmp_path):
    """host/Stable2D.h through fs2d_run: the last layer's U, V, T and the printed err values equal the reference's."""
    z, m = _fx2d(name)
    cfgp, out, dump = str(tmp_path / "c.txt"), str(tmp_path / "o.cdl"), str(tmp_path / "o.bin")
    open(cfgp, "w").write(m["config_text"])
    n = max(m["steps"])
    r = subprocess.run([driver2d, os.path.join(INP, m["data"]), out, cfgp, "--steps", str(n), "--dump", dump], check=True, capture_output=True, text=True, timeout=900).stdout
    assert "%i,%i,1" % tuple(m["dims"]) in r and "dt = %f" % m["dt"] in r
    assert re.findall(r"err = ([0-9.]+),", r) == ["%.4f" % e for e in m["err_trace"][:n]]
    for a, v in zip(_dump(dump), "UVT"):
        assert np.array_equal(a, z["%s_step%d" % (v, n)]), v
