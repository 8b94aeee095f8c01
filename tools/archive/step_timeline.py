#!/usr/bin/env python3
"""Timeline of the last kernels of a rocprofv3 --kernel-trace run (csv): start / end / duration in us relative to the
first listed kernel, queue, short kernel name.  Usage: python tools/step_timeline.py <dir> [n_last=12]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
print(f"{'start us':>10} {'end us':>10} {'dur us':>8}  queue  kernel")
for s, e, q, name in rows:
    short = name.split("(")[0].replace("void sdfv::(anonymous namespace)::", "")[:70]
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{q:<4} {short}")
