"""Summarise a rocprofv3 --pmc pass: per kernel symbol (largest grid), mean of each counter per launch.
usage: python scratch/pmc_summary.py <dir> [name filter ...]"""
import collections, csv, glob, sys
d = sys.argv[1]
filt = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if filt and not any(x in k for x in filt):
            continue
        acc[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(acc.items()):
    print(k[:90], "grid", g)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
