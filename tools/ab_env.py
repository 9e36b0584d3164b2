"""Interleaved A/B of environment settings on one box: python tools/ab_env.py [rounds] name:K=V,K=V ...   ("default:" = no change)."""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = ("import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi\n"
        "P.timing(256, capi.SWEEP_AUTO, reps=12); P.timing(256, capi.SWEEP_AUTO, reps=16)") % HERE
rounds = int(sys.argv[1])
cfgs = []
for a in sys.argv[2:]:
    name, _, kv = a.partition(":")
    cfgs.append((name, dict(x.split("=") for x in kv.split(",") if x)))
acc = {c[0]: [] for c in cfgs}
for r in range(rounds):
    for name, env in cfgs:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300).stdout.strip().splitlines()[-1]
        m = re.search(r"'X': ([0-9.]+), 'Y': ([0-9.]+), 'Z': ([0-9.]+)\} ; step ([0-9.]+) ms", out)
        acc[name].append([float(x) for x in m.groups()])
for name in acc:
    a = list(zip(*acc[name]))
    print("%-24s X %.4f  Y %.4f  Z %.4f  step %.3f ms (mean of %d)" % (name, *[sum(v) / len(v) for v in a], rounds))
