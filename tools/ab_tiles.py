"""A/B of the X/Y tile configurations on ONE box, interleaved (boxes and runs differ by several %): 64-line tiles (default) against
two 32-line tiles per CU with the second workgroup of every CU started late.  python tools/ab_tiles.py [rounds]   (GPU box)"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = ("import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi\n"
        "P.timing(256, capi.SWEEP_AUTO, reps=12); P.timing(256, capi.SWEEP_AUTO, reps=16)") % HERE
cfgs = [("64 lines", 64, 0), ("32 lines", 0, 0), ("32 lines, 2nd wg +8us", 0, 0x0240), ("32 lines, 2nd wg +16us (default)", 0, None), ("32 lines, 2nd wg +24us", 0, 0x0640)]
if os.environ.get("AB_ORDER1"):
    cfgs = [("32 lines, +16us (default)", 0, None), ("same, lane tiles fastest", 0, 0x0441), ("+12us", 0, 0x0340), ("+20us", 0, 0x0540)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
acc = {c[0]: [] for c in cfgs}
for r in range(rounds):
    for name, var, order in cfgs:
        env = dict(os.environ, FS3D_PART_VARIANT=str(var))
        if order is not None:
            env["FS3D_PART_ORDER"] = str(order)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
        m = re.search(r"'X': ([0-9.]+), 'Y': ([0-9.]+), 'Z': ([0-9.]+)\} ; step ([0-9.]+) ms", out)
        acc[name].append([float(x) for x in m.groups()])
        print(r, name, acc[name][-1], flush=True)
for name in acc:
    a = list(zip(*acc[name]))
    print("%-34s X %.4f  Y %.4f  Z %.4f  step %.3f ms (mean of %d); X+Y min %.4f" % (name, *[sum(v) / len(v) for v in a], rounds, min(x + y for x, y in zip(a[0], a[1]))))
