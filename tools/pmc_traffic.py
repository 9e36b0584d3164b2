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
cells = float(sys.argv[3]) if len(sys.argv) > 3 else 256.0**3
print("%-52s %6s %14s %14s %12s %12s" % ("kernel", "calls", "fetch KiB/launch", "write KiB/launch", "B/cell raw", "B/cell corr"))
for k in sorted(f):
    fk = sum(f[k]) / len(f[k]); wk = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    print("%-52s %6d %14.0f %14.0f %12.1f %12.1f" % (k[:52], len(f[k]), fk, wk, (fk + wk) * 1024 / cells, (2 * fk + wk) * 1024 / cells))
