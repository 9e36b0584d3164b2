"""Start-stagger experiment of the X/Y partition kernel: FS3D_PART_ORDER = mode | (delay << 8) is read once per process,
so every setting runs in its own child process.  python tools/stagger_sweep.py   (GPU box)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); import part_check as P; from cmc_fluid_solver_amd import capi; P.timing(256, capi.SWEEP_AUTO, reps=12)" % HERE
settings = ([0, 1] + [1 | 0x40 | (d << 8) for d in (2, 4, 6, 8, 12)]) if os.environ.get("FS3D_PART_VARIANT") == "7" else [0] + [m | (d << 8) for m in (0x10, 0x20) for d in (3, 6, 9, 12)]
for o in (settings if len(sys.argv) < 2 else [int(x, 0) for x in sys.argv[1:]]):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FS3D_PART_ORDER=str(o)), capture_output=True, text=True, timeout=200)
    print("order %#06x (mode %#x, delay %d x 3.9 us): %s" % (o, o & 0x30, o >> 8, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
