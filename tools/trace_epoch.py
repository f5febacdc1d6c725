"""Summarise a rocprofv3 --kernel-trace CSV: the kernels of ONE steady-state epoch in launch order
(name, duration, gap to the previous kernel's end), and the per-kernel averages.
usage: python tools/trace_epoch.py <dir> [first-kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "row_sqnorms"
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "dbgsom" in r["Kernel_Name"]]
    starts = [k for k, r in enumerate(rows) if first in r["Kernel_Name"]]
    if len(starts) < 3:
        print("too few epochs in", path)
        continue
    a, b = starts[-2], starts[-1]          # the last complete epoch
    print(f"# {path}: epoch of {b - a} launches")
    t_prev = None
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("dbgsom::", "").replace("void ", "")[:58]
        gap = 0 if t_prev is None else s - t_prev
        print(f"{(s - t0) / 1e3:9.1f} us  {name:58s} {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  "
              f"grid {r.get('Grid_Size', '?'):>9s} wg {r.get('Workgroup_Size', '?'):>4s} vgpr {r.get('VGPR_Count', '?')}")
        t_prev = max(e, t_prev or 0)
    print(f"# epoch span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy "
          f"{sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[a:b]) / 1e3:.1f} us")
    acc = defaultdict(list)
    for r in rows:
        acc[r["Kernel_Name"].split("(")[0].replace("dbgsom::", "")[:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("# averages")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:72s} n={len(v):4d} avg {sum(v) / len(v) / 1e3:9.2f} us")
