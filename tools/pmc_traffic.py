#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs into HBM bytes per launch per kernel.
Usage: python tools/pmc_traffic.py <fetch_csv> <write_csv> [cells]
Units (MI355X_MICROARCH.md, HBM): FETCH_SIZE/WRITE_SIZE are in KiB... reported raw and with the guide's
gfx950 correction for wide coalesced reads (FETCH_SIZE counts 128-byte requests as 64 bytes: x2)."""
import csv, sys, collections
def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
cells = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 256.0**3
print("%-52s %6s %14s %14s %12s %12s" % ("kernel", "calls", "fetch KiB/launch", "write KiB/launch", "B/cell raw", "B/cell corr"))
for k in sorted(f):
    fk = sum(f[k]) / len(f[k]); wk = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    print("%-52s %6d %14.0f %14.0f %12.1f %12.1f" % (k[:52], len(f[k]), fk, wk, (fk + wk) * 1024 / cells, (2 * fk + wk) * 1024 / cells))

# --json <path>: per-launch HBM bytes of the three sweep classes for bench.py's `roofline.traffic` (profiles/pmc_traffic.json), stamped
# with the hash of the kernel sources they were measured with (bench.py drops the figure when the sources have changed since).
if "--json" in sys.argv:
    import hashlib, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = b"".join(open(os.path.join(root, "cmc_fluid_solver_amd", "csrc", n), "rb").read() for n in ("kernels_part.hip", "fs3d_common.h"))
    out = {"grid": [256, 256, 256], "dtype": "f32", "round": 3, "kernel_source_sha16": hashlib.sha256(src).hexdigest()[:16],
           "note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline, averaged over the launches of a time step (half of them do not store `next`). X/Y partition "
                   "kernels load 4 bytes per lane: FETCH_SIZE raw (calibration: profiles/r1_pmc_traffic.txt); the Z kernel loads 16 bytes per "
                   "lane: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM: 128-byte requests tallied as 64)."}
    for k in f:
        fk = sum(f[k]) / len(f[k]); wk = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
        if "k_sweep_part_z" in k: out["sweep_Z"] = int((2 * fk + wk) * 1024)
        elif "k_sweep_part<float, 0" in k: out["sweep_X"] = int((fk + wk) * 1024)
        elif "k_sweep_part<float, 1" in k: out["sweep_Y"] = int((fk + wk) * 1024)
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
