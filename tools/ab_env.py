"""A/B of environment / library configurations on one box, interleaved (GPU box).
   python tools/ab_env.py name[:lib][:ENV=VAL,ENV=VAL] ...     e.g.  v64::FS3D_PART_VARIANT=64  o0:exp:FS3D_PART_ORDER=0"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "cmc_fluid_solver_amd")
code = ("import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi\n"
        "P.timing(256, capi.SWEEP_AUTO, reps=12); P.timing(256, capi.SWEEP_AUTO, reps=16)") % HERE
cfgs = [("default", None, {})]
for a in sys.argv[1:]:
    parts = a.split(":")
    name, lib = parts[0], (parts[1] if len(parts) > 1 and parts[1] else None)
    env = dict(kv.split("=") for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
    cfgs.append((name, lib, env))
acc = {c[0]: [] for c in cfgs}
for r in range(3):
    for name, lib, envx in cfgs:
        env = dict(os.environ); env.update(envx)
        if lib:
            env["FS3D_LIB_PATH"] = os.path.join(PKG, "libfs3d_hip_%s.so" % lib)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
        m = re.search(r"'X': ([0-9.]+), 'Y': ([0-9.]+), 'Z': ([0-9.]+)\} ; step ([0-9.]+) ms", out)
        acc[name].append([float(x) for x in m.groups()])
for name, _, _ in cfgs:
    a = list(zip(*acc[name]))
    print("%-10s X %.4f  Y %.4f  Z %.4f  step %.3f ms (mean of 3)" % (name, *[sum(v) / len(v) for v in a]), flush=True)
