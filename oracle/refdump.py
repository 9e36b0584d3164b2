"""TEST INFRASTRUCTURE ONLY -- runs the reference's own CPU path (oracle/_ref/ref_adi_f32|f64, built by `make ref_full`
from the reference's translation units where they lie; see oracle/ref_harness_adi.cpp) and reads its dump.

Only tests/ and tests/golden/make_ref_golden.py use this, and only in the build container: the GPU box has neither
/root/reference nor a need for it -- the committed fixtures under tests/golden/ref_*.npz carry the outputs.
"""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def binary(dtype):
    p = os.path.join(_HERE, "_ref", "ref_adi_f32" if np.dtype(dtype) == np.float32 else "ref_adi_f64")
    return p if os.path.exists(p) else None


def available():
    return binary(np.float32) is not None and binary(np.float64) is not None


def strip_cr(src, dst):
    """The reference's Linux build reads CRLF files wrongly (NODE_IN = 0); its own run script strips the \\r first
    (bin/Release/run_examples_CPU.sh:12-15)."""
    with open(src, "rb") as f:
        b = f.read()
    with open(dst, "wb") as f:
        f.write(b.replace(b"\r", b""))


def run(data_path, config_text, dtype, max_steps, dump_steps, align=True, grid_times=(), threads=None, timeout=1800):
    """-> dict(dims, frames, dx, dy, dz, dt, cycle_length, params, nodes{type,bc_vel,bc_temp,vel,T}, grid_at{t: {type, vel}},
    steps{s: {err, U, V, W, T}}, layers{s: {outV, outT}} (GetLayer at the driver's output steps), stdout, err_trace (the `err = %.8f` prints, one per step))"""
    exe = binary(dtype)
    if exe is None:
        raise RuntimeError("oracle/_ref/ref_adi_* not built (make -C oracle ref_full needs /root/reference)")
    ft = np.dtype(dtype)
    tmp = tempfile.mkdtemp(prefix="refrun_")
    try:
        d = os.path.join(tmp, "data.txt")
        c = os.path.join(tmp, "config.txt")
        strip_cr(data_path, d)
        with open(c, "w") as f:
            f.write(config_text.replace("\r", ""))
        out = os.path.join(tmp, "dump.bin")
        cmd = [exe, d, c, out, str(int(max_steps)), ",".join(str(int(s)) for s in dump_steps) or "-"]
        if align:
            cmd.append("align")
        if len(grid_times):
            cmd.append(",".join(repr(float(t)) for t in grid_times))
        env = dict(os.environ)
        if threads:
            env["OMP_NUM_THREADS"] = str(threads)
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=tmp)
        if p.returncode != 0 or not os.path.exists(out):
            raise RuntimeError("reference run failed (rc %d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-2000:]))
        res = read_dump(out, ft)
        res["stdout"] = p.stdout
        res["err_trace"] = [float(m) for m in re.findall(r"err = ([0-9.]+),", p.stdout)]
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def read_dump(path, ft):
    with open(path, "rb") as f:
        b = f.read()
    assert b[:8] == b"FS3DREF1"
    o = 8
    fsz, dimx, dimy, dimz, frames = struct.unpack_from("<5i", b, o); o += 20
    assert fsz == ft.itemsize, "dump is %d-byte FTYPE" % fsz
    dx, dy, dz, dt, length, vT, vvis, tvis, tphi = struct.unpack_from("<9d", b, o); o += 72
    dims = (dimx, dimy, dimz)
    n = dimx * dimy * dimz

    def take(dtype, count, shape):
        nonlocal o
        a = np.frombuffer(b, dtype=dtype, count=count, offset=o).reshape(shape).copy()
        o += a.nbytes
        return a

    nodes = {"type": take(np.uint8, n, dims), "bc_vel": take(np.uint8, n, dims), "bc_temp": take(np.uint8, n, dims),
             "vel": take(ft, 3 * n, dims + (3,)), "T": take(ft, n, dims)}
    (ng,) = struct.unpack_from("<i", b, o); o += 4
    grid_at = {}
    for _ in range(ng):
        (t,) = struct.unpack_from("<d", b, o); o += 8
        grid_at[t] = {"type": take(np.uint8, n, dims), "vel": take(ft, 3 * n, dims + (3,))}
    steps, layers = {}, {}
    while True:
        (kind,) = struct.unpack_from("<i", b, o); o += 4
        if kind < 0:
            break
        (s,) = struct.unpack_from("<i", b, o); o += 4
        if kind == 1:
            (err,) = struct.unpack_from("<d", b, o); o += 8
            st = {"err": err}
            for v in "UVWT":
                st[v] = take(ft, n, dims)
            steps[s] = st
        else:
            ox, oy, oz = struct.unpack_from("<3i", b, o); o += 12
            layers[s] = {"outV": take(ft, 3 * ox * oy * oz, (ox, oy, oz, 3)), "outT": take(np.float64, ox * oy * oz, (ox, oy, oz))}
    assert o == len(b)
    return {"dims": dims, "frames": frames, "dx": dx, "dy": dy, "dz": dz, "dt": dt, "cycle_length": length,
            "params": (vT, vvis, tvis, tphi), "nodes": nodes, "grid_at": grid_at, "steps": steps, "layers": layers}


def binary2d():
    p = os.path.join(_HERE, "_ref", "ref_stable2d")
    return p if os.path.exists(p) else None


def run2d(data_path, config_text, max_steps, dump_steps, timeout=1800):
    """The reference's Grid2D + StableSolver2D (oracle/ref_harness_2d.cpp) -> dict(dims, frames, dx, dy, dt, cycle_length, v_vis,
    steps{s: {type, U, V, T}}, stdout)."""
    exe = binary2d()
    if exe is None:
        raise RuntimeError("oracle/_ref/ref_stable2d not built (make -C oracle ref_full needs /root/reference)")
    tmp = tempfile.mkdtemp(prefix="refrun2d_")
    try:
        d, c, out = (os.path.join(tmp, n) for n in ("data.txt", "config.txt", "dump.bin"))
        strip_cr(data_path, d)
        with open(c, "w") as f:
            f.write(config_text.replace("\r", ""))
        p = subprocess.run([exe, d, c, out, str(int(max_steps)), ",".join(str(int(s)) for s in dump_steps) or "-"],
                           capture_output=True, text=True, timeout=timeout, cwd=tmp)
        if p.returncode != 0 or not os.path.exists(out):
            raise RuntimeError("reference 2D run failed (rc %d)\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-2000:]))
        with open(out, "rb") as f:
            b = f.read()
        assert b[:8] == b"FS2DREF1"
        fsz, dimx, dimy, frames = struct.unpack_from("<4i", b, 8)
        dx, dy, dt, length, v_vis = struct.unpack_from("<5d", b, 24)
        assert fsz == 4
        o, n, steps = 64, dimx * dimy, {}
        while True:
            (s,) = struct.unpack_from("<i", b, o); o += 4
            if s < 0:
                break
            st = {"type": np.frombuffer(b, np.uint8, n, o).reshape(dimx, dimy).copy()}; o += n
            for v in "UVT":
                st[v] = np.frombuffer(b, np.float32, n, o).reshape(dimx, dimy).copy(); o += 4 * n
            steps[s] = st
        assert o == len(b)
        return {"dims": (dimx, dimy), "frames": frames, "dx": dx, "dy": dy, "dt": dt, "cycle_length": length, "v_vis": v_vis,
                "steps": steps, "stdout": p.stdout, "err_trace": [float(m) for m in re.findall(r"err = ([0-9.]+),", p.stdout)]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
