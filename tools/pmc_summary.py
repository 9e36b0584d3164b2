"""Average every PMC counter per kernel over the launches of a rocprofv3 --pmc run.
python tools/pmc_summary.py <dir with *_counter_collection.csv> [cells]"""
import csv, glob, sys, collections
cells = float(sys.argv[2]) if len(sys.argv) > 2 else 256.0 ** 3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "sweep" not in k and "copyBuffer" not in k: continue
    print(k[:70])
    for cn, v in sorted(agg[k].items()):
        m = sum(v) / len(v)
        print("    %-28s calls %3d  mean %16.1f   per cell %10.3f" % (cn, len(v), m, m / cells))
