"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name and counter, the mean
over dispatches.  usage: python tools/pmc_summary.py <dir-with-csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

for root in sys.argv[1:]:
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0][-60:]
            if "dbgsom" not in name:
                continue
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        print("#", path)
        for (name, c), v in sorted(acc.items()):
            print(f"{name:62s} {c:32s} n={len(v):2d} mean={sum(v) / len(v):.6g}")
