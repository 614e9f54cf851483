"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv (one line per kernel and counter)."""
import csv
import glob
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: [0.0, set()])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0][:60]
                a = acc[(name, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1].add(row["Dispatch_Id"])
        for (name, ctr), (v, d) in sorted(acc.items()):
            if v > 0:
                print(f"{name:60s} {ctr:28s} dispatches={len(d):4d} sum={v:.6g} per_dispatch={v / len(d):.6g}")
