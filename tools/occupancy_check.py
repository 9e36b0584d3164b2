"""32-line tiles (FS3D_PART_VARIANT=7): two workgroups per CU against one (FS3D_PART_LDSPAD holds the second one out).
python tools/occupancy_check.py   (GPU box)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi; P.timing(256, capi.SWEEP_AUTO, reps=12)" % HERE
for var, order, pad in ((0, 0, 0), (7, 1, 0), (7, 1, 50000), (7, 0, 0), (7, 0, 50000)):
    env = dict(os.environ, FS3D_PART_VARIANT=str(var), FS3D_PART_ORDER=str(order), FS3D_PART_LDSPAD=str(pad))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=200)
    print("variant %d order %d lds pad %6d: %s" % (var, order, pad, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
