"""Field padding / layer skew experiment (FS3D_FIELD_PAD elements, FS3D_LAYER_SKEW bytes; read at context creation).
python tools/pad_sweep.py   (GPU box)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi; P.timing(256, capi.SWEEP_AUTO, reps=12); P.timing(256, capi.SWEEP_AUTO, reps=12)" % HERE
import itertools
cfgs = [(0, 0)] + list(itertools.product((320, 576, 1088, 2112, 3136, 8256), (1280, 4864, 9472, 20736)))
for pad, skew in cfgs:
    env = dict(os.environ, FS3D_FIELD_PAD=str(pad), FS3D_LAYER_SKEW=str(skew))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("field pad %6d elements, layer skew %6d bytes: %s" % (pad, skew, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
