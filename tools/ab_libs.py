"""A/B of library variants (build.build_variant) on one box, interleaved.  python tools/ab_libs.py name1 name2 ...   (GPU box)"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "cmc_fluid_solver_amd")
code = ("import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi\n"
        "P.timing(256, capi.SWEEP_AUTO, reps=12); P.timing(256, capi.SWEEP_AUTO, reps=16)") % HERE
names = ["default"] + sys.argv[1:]
acc = {n: [] for n in names}
for r in range(3):
    for n in names:
        env = dict(os.environ)
        if n != "default":
            env["FS3D_LIB_PATH"] = os.path.join(PKG, "libfs3d_hip_%s.so" % n)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
        m = re.search(r"'X': ([0-9.]+), 'Y': ([0-9.]+), 'Z': ([0-9.]+)\} ; step ([0-9.]+) ms", out)
        acc[n].append([float(x) for x in m.groups()])
for n in names:
    a = list(zip(*acc[n]))
    print("%-10s X %.4f  Y %.4f  Z %.4f  step %.3f ms (mean of 3)" % (n, *[sum(v) / len(v) for v in a]))
