#!/usr/bin/env python3
"""Generates tests/golden/ref_*.npz -- OUTPUTS OF THE REFERENCE ITSELF.

Every array in these fixtures was produced by the reference's own CPU path: its translation units compiled where they lie
under /root/reference/src (`make -C oracle ref_full`: genuine cuda_runtime.h from the image's triton package, the reference's
vendored netcdf.h, no stand-ins), driven by oracle/ref_harness_adi.cpp in the order of FluidSolver3D/FluidSolver3D.cpp:63-262.
`*_f32` = the reference as shipped (FTYPE float, Common/Geometry.h:21); `*_f64` = the same sources with that one #define
switched to double in a temporary copy ("FTYPE switched").  Inputs: the data/config files under tests/golden/inputs (the
reference's shipped examples byte for byte, \\r stripped at run time as its own run script does; `u_bend` is authored here), with
the config keys listed per case replaced.

A fixture holds data only: the inputs' names and the exact config text, grid dims, dt, FluidParams, the node arrays
(type/bc as uint8 volumes; boundary values compact over the BOUND/VALVE cells), the printed `err` trace, and per dumped step
the four fields of the new layer -- in full for small cases, as sha256 + a strided sample for large ones -- plus the
result layers `Solver3D::GetLayer` hands to the NetCDF writer at the driver's output steps.

Run in the build container from the repo root:   python tests/golden/make_ref_golden.py [case ...]
"""
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refdump  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
INP = os.path.join(HERE, "inputs")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def set_keys(text, **kv):
    """Replace `key value` lines of a reference config (or append the key)."""
    text = text.replace("\r", "")
    for k, v in kv.items():
        pat = re.compile(r"^%s[ \t]+.*$" % re.escape(k), re.M)
        line = "%s\t\t%s" % (k, v)
        text = pat.sub(line, text) if pat.search(text) else text.rstrip("\n") + "\n" + line + "\n"
    return text


# name: data file, config file, config keys replaced, align, precisions, steps run, steps with full fields, steps with sha + sample,
#       sample stride, geometry times (Grid3D::Prepare_CPU(t) volumes)
CASES = {
    # authored U-bend (two arms, a fin, a closed box: up to 3 segments per row), ragged dims (no align), sloped bottom
    "u_bend": dict(data="u_bend_2D_data.txt", config="u_bend_2D_config.txt", keys={}, align=False, prec=("f32", "f64"),
                   steps=10, full={"f32": (1, 2, 10), "f64": (10,)}, hashed=(1, 2, 5, 10), stride=0),
    # the shipped 64^3 example, its own config, all 100 steps
    "box_pipe": dict(data="box_pipe_2D_data.txt", config="box_pipe_2D_config.txt", keys={}, align=True, prec=("f32", "f64"),
                     steps=100, full={}, hashed=(1, 2, 10, 50, 100), stride=2),
    # other iteration counts (num_global / num_local) on the shipped example: the merge cadence of TimeStep / SolveDirection
    "box_pipe_g1l3": dict(data="box_pipe_2D_data.txt", config="box_pipe_2D_config.txt", keys=dict(num_global="1", num_local="3", out_gridx="16", out_gridy="16", out_gridz="16"),
                          align=True, prec=("f32", "f64"), steps=5, full={}, hashed=(1, 2, 5), stride=4),
    "box_pipe_g3l1": dict(data="box_pipe_2D_data.txt", config="box_pipe_2D_config.txt", keys=dict(num_global="3", num_local="1", out_gridx="16", out_gridy="16", out_gridz="16"),
                          align=True, prec=("f32", "f64"), steps=5, full={}, hashed=(1, 2, 5), stride=4),
    # the shipped masked-bottom example (depth_var 0.2)
    "non_uniform_pipe": dict(data="non_uniform_pipe_2D_data.txt", config="non_uniform_pipe_2D_config.txt", keys={}, align=True,
                             prec=("f32", "f64"), steps=20, full={}, hashed=(1, 2, 10, 20), stride=2),
    # BASELINE configs[1]: 128^3 file-driven box
    "box128": dict(data="box_pipe_2D_data.txt", config="box_pipe_2D_config.txt",
                   keys=dict(grid_dx="0.0085", grid_dy="0.0085", grid_dz="0.0085", out_gridx="32", out_gridy="32", out_gridz="32"),
                   align=True, prec=("f32", "f64"), steps=10, full={}, hashed=(1, 2, 10), stride=4),
    # BASELINE configs[2]: 256^3 fp32 file-driven box (the headline size)
    "box256": dict(data="box_pipe_2D_data.txt", config="box_pipe_2D_config.txt",
                   keys=dict(grid_dx="0.0042", grid_dy="0.0042", grid_dz="0.0042", out_gridx="32", out_gridy="32", out_gridz="32"),
                   align=True, prec=("f32",), steps=3, full={}, hashed=(1, 2, 3), stride=8),
    # BASELINE configs[4]: 256^3 with masked geometry from data/3D
    "non_uniform256": dict(data="non_uniform_pipe_2D_data.txt", config="non_uniform_pipe_2D_config.txt",
                           keys=dict(grid_dx="0.0042", grid_dy="0.0042", grid_dz="0.0042", out_gridx="32", out_gridy="32", out_gridz="32"),
                           align=True, prec=("f32",), steps=2, full={}, hashed=(1, 2), stride=8),
    # BASELINE configs[3]: 512^3 fp32 (the grid the 8-GPU run cuts into 64-plane slabs): one step, hashes + a 32^3 sample
    "box512": dict(data="box_pipe_2D_data.txt", config="box_pipe_2D_config.txt",
                   keys=dict(grid_dx="0.0021", grid_dy="0.0021", grid_dz="0.002", out_gridx="16", out_gridy="16", out_gridz="16"),
                   align=True, prec=("f32",), steps=1, full={}, hashed=(1,), stride=16, big=True),
    # multi-frame Shape2D (moving walls + valve): adapted config (the shipped one is rejected by the reference's parser)
    "heart_us": dict(data="heart_us_2D_data.txt", config="heart_us_2D_config.txt", keys={}, align=True, prec=("f32",),
                     steps=8, full={}, hashed=(1, 2, 3, 4, 8), stride=2, grid_times="frames"),
    # Shape3D inputs (triangle meshes): the shipped box_pipe_3D and tetra examples -- their shipped config names no out_vars and no
    # frame_time, which the reference's parser / time loop need: both keys appended (tests/golden/inputs/box_pipe_3D_config.txt);
    # tetra diverges in the reference itself at step 9 ("Error is too big!"): 5 steps
    "box_pipe_3D": dict(data="box_pipe_3D_data.txt", config="box_pipe_3D_config.txt", keys=dict(out_gridx="32", out_gridy="32", out_gridz="32"),
                        align=True, prec=("f32",), steps=10, full={}, hashed=(1, 2, 10), stride=4),
    "tetra": dict(data="tetra_data.txt", config="box_pipe_3D_config.txt", keys=dict(out_gridx="32", out_gridy="16", out_gridz="16"),
                  align=True, prec=("f32",), steps=5, full={}, hashed=(1, 2, 5), stride=2),
    # authored two-frame icosphere (tests/test_shape3d.py: 80 faces, none axis-aligned, the second frame shifted): moving mesh
    "sphere_3D": dict(data="sphere_3D_data.txt", config="sphere_3D_config.txt", keys={}, align=True, prec=("f32", "f64"), steps=7,
                      full={"f32": (1, 7)}, hashed=(1, 2, 7), stride=0, grid_times=(0.0, 0.1, 0.2, 0.3, 0.5)),
}


def compact_nodes(nd, ft):
    ty = nd["type"]
    sel = ty >= 2
    out = {"node_type": ty, "node_bc_vel": nd["bc_vel"], "node_bc_temp": nd["bc_temp"],
           "bnd_vel": nd["vel"][sel], "bnd_T": nd["T"][sel]}
    # values of the cells that are not boundary cells: constants per type (asserted, stored)
    consts = {}
    for t, nm in ((0, "in"), (1, "out")):
        m = ty == t
        if m.any():
            for key, arr in (("T", nd["T"][m]), ("vx", nd["vel"][..., 0][m]), ("vy", nd["vel"][..., 1][m]), ("vz", nd["vel"][..., 2][m])):
                u = np.unique(arr)
                assert len(u) == 1, "%s cells carry %d different %s values" % (nm, len(u), key)
                consts["%s_%s" % (nm, key)] = float(u[0])
    meta = {"node_consts": consts, "nodes_sha": {k: sha(nd[k]) for k in ("type", "bc_vel", "bc_temp", "vel", "T")},
            "node_in": int((ty == 0).sum())}
    return out, meta


def make(name, case):
    cfg_text = set_keys(open(os.path.join(INP, case["config"])).read(), **case["keys"])
    for prec in case["prec"]:
        ft = np.float32 if prec == "f32" else np.float64
        full = tuple(case["full"].get(prec, ()))
        dumps = sorted(set(full) | set(case["hashed"]))
        gt = ()
        if case.get("grid_times") == "frames":
            # multi-frame input: geometry at the start, inside and at the end of frames (probe run for the cycle length first)
            probe = refdump.run(os.path.join(INP, case["data"]), cfg_text, ft, 0, [], case["align"])
            fr, ln = probe["frames"], probe["cycle_length"]
            gt = [0.0, 0.25 * ln / fr, 0.5 * ln / fr, 1.0 * ln / fr, 1.5 * ln / fr, (fr - 0.5) * ln / fr]
        elif case.get("grid_times"):
            gt = list(case["grid_times"])
        t0 = time.time()
        # the geometry at other times comes from a run of its own (no steps): Prepare_CPU(t) leaves traces in the Node array
        # (Shape3D: cells that were NODE_BOUND at time t keep T = 0 when they are fluid again), the stepping run must not see them
        r = refdump.run(os.path.join(INP, case["data"]), cfg_text, ft, case["steps"], dumps, case["align"])
        if len(gt):
            r["grid_at"] = refdump.run(os.path.join(INP, case["data"]), cfg_text, ft, 0, [], case["align"], gt)["grid_at"]
        out, meta = compact_nodes(r["nodes"], ft)
        meta.update(case=name, data=case["data"], config_text=cfg_text, align=case["align"], prec=prec, dims=list(r["dims"]),
                    frames=r["frames"], dx=r["dx"], dy=r["dy"], dz=r["dz"], dt=r["dt"], cycle_length=r["cycle_length"],
                    params=list(r["params"]), steps_run=case["steps"], err_trace=r["err_trace"], stride=case["stride"],
                    full_steps=list(full), hashed_steps=list(case["hashed"]), layer_steps=sorted(r["layers"]),
                    step_err={}, step_sha={}, layer_sha={}, grid_times=list(gt),
                    stdout_head=[ln for ln in r["stdout"].split("\n") if ln.startswith(("Grid =", "NODE_IN", "WARNING", "Single", "Double"))],
                    generator="tests/golden/make_ref_golden.py; oracle/_ref/ref_adi_%s = reference TUs as they lie%s"
                              % (prec, "" if prec == "f32" else " (temporary copy, FTYPE switched to double)"))
        s = case["stride"]
        sampled = (2, max(case["hashed"]))          # strided samples (for the tolerance checks of the fp32 production kernels)
        meta["sampled_steps"] = list(sampled) if s else []
        for st, rec in r["steps"].items():
            meta["step_err"][str(st)] = rec["err"]
            meta["step_sha"][str(st)] = {v: sha(rec[v]) for v in "UVWT"}
            for v in "UVWT":
                if st in full:
                    out["%s_step%d" % (v, st)] = rec[v]
                elif s and st in sampled:
                    out["%s_sample%d" % (v, st)] = rec[v][::s, ::s, ::s]
        lsteps = sorted(r["layers"])
        for st in lsteps:
            meta["layer_sha"][str(st)] = {"outV": sha(r["layers"][st]["outV"]), "outT": sha(r["layers"][st]["outT"])}
        for st in lsteps[:1] + lsteps[-1:]:       # first and last result layer in full (they are small: the config's out grid)
            out["outV_step%d" % st] = r["layers"][st]["outV"]
            out["outT_step%d" % st] = r["layers"][st]["outT"]
        for i, t in enumerate(gt):
            g = r["grid_at"][t]
            out["grid%d_type" % i] = g["type"]
            out["grid%d_bnd_vel" % i] = g["vel"][g["type"] >= 2]
        out["meta"] = np.array(json.dumps(meta))
        fn = os.path.join(HERE, "ref_%s_%s.npz" % (name, prec))
        np.savez_compressed(fn, **out)
        print("%-40s %8.1f KB  %5.1f s  dims %s  NODE_IN %d  err %s" % (os.path.basename(fn), os.path.getsize(fn) / 1e3, time.time() - t0,
                                                                      r["dims"], meta["node_in"], r["err_trace"][-1:]), flush=True)


# ---- the 2D path (f4): the reference's own Grid2D + StableSolver2D (oracle/ref_harness_2d.cpp) ------------------------------------
HEART2D = """dimension\t2D
viscosity \t0.004
density \t1.0
bc_type\t\tNoSlip
grid_dx\t\t0.0007
grid_dy\t\t0.0007
cycles \t\t1
time_steps \t400
out_time_steps \t2
out_gridx\t32
out_gridy \t32
out_fmt\t\tNetCDF
solver\t\tStable
num_global \t2
num_local \t1
"""
CASES2D = {
    # the authored lid-driven cavity (BASELINE configs[0]) as configured: 128 x 128
    "cavity128": dict(data="cavity_2D_data.txt", config="cavity_2D_config.txt", keys={}, steps=(1, 2)),
    # the same cavity at 54 x 54, one global iteration
    "cavity54": dict(data="cavity_2D_data.txt", config="cavity_2D_config.txt",
                     keys=dict(grid_dx="0.0199", grid_dy="0.0199", time_steps="200", out_time_steps="1", out_gridx="27", out_gridy="27", num_global="1"),
                     steps=(1, 2, 3)),
    # moving walls: the 10-frame heart_us outline as a 2D problem (grid.Prepare(t) every step); with fewer than ~400 steps per frame
    # the reference's Stable solver stops with "Error is too big!" at the first step
    "heart2d": dict(data="heart_us_2D_data.txt", config_text=HEART2D, keys={}, steps=(1, 2, 4)),
}


def make2d(name, case):
    cfg_text = case.get("config_text") or set_keys(open(os.path.join(INP, case["config"])).read(), **case["keys"])
    t0 = time.time()
    r = refdump.run2d(os.path.join(INP, case["data"]), cfg_text, max(case["steps"]), case["steps"])
    out = {}
    for st, rec in r["steps"].items():
        for k in ("type", "U", "V", "T"):
            out["%s_step%d" % (k, st)] = rec[k]
    meta = dict(case=name, data=case["data"], config_text=cfg_text, dims=list(r["dims"]), frames=r["frames"], dx=r["dx"], dy=r["dy"], dt=r["dt"],
                cycle_length=r["cycle_length"], v_vis=r["v_vis"], err_trace=r["err_trace"], steps=list(case["steps"]),
                generator="tests/golden/make_ref_golden.py; oracle/_ref/ref_stable2d = the reference's Grid2D + StableSolver2D as they lie")
    out["meta"] = np.array(json.dumps(meta))
    fn = os.path.join(HERE, "ref2d_%s.npz" % name)
    np.savez_compressed(fn, **out)
    print("%-40s %8.1f KB  %5.1f s  dims %s  err %s" % (os.path.basename(fn), os.path.getsize(fn) / 1e3, time.time() - t0, r["dims"], r["err_trace"]), flush=True)


if __name__ == "__main__":
    if not refdump.available():
        sys.exit("oracle/_ref/ref_adi_f32|f64 missing: run `make -C oracle ref_full` (needs /root/reference)")
    names = sys.argv[1:] or list(CASES) + list(CASES2D)
    for nm in names:
        if nm in CASES2D:
            make2d(nm, CASES2D[nm])
        else:
            make(nm, CASES[nm])
