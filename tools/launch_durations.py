#!/usr/bin/env python3
"""Per-launch durations of the sweep kernels from a rocprofv3 --kernel-trace CSV, in launch order (one time step).
Usage: python tools/launch_durations.py <kernel_trace.csv> [first_launch_index]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "k_sweep_pipe" in n and "true" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("<")[1].split(">")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) - 24
for i, (t, n, d) in enumerate(rows[k0:k0 + 24]):
    print("%3d  dir %s  %8.1f us" % (i, n.split(",")[1].strip(), d))
