#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid size).  `--stats` averages over every dispatch of a
kernel NAME; a bench run with the texture-placement probe on launches the fill kernel over the first slices only as well, so
the whole-grid launches are separated from those by their grid size here.
usage: python tools/kernel_trace_avg.py <dir with *_kernel_trace.csv> [min dispatches]"""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
min_n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void ", "").replace("sdfv::(anonymous namespace)::", ""))
        grid = tuple(int(row.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        acc[(name, grid)].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
print(f'{"kernel":70s} {"grid (threads)":>24s} {"calls":>6s} {"avg us":>9s} {"median":>9s} {"min":>9s} {"max":>9s}')
for (name, grid), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(d) < min_n:
        continue
    d.sort()
    print(f"{name[:70]:70s} {str(grid):>24s} {len(d):6d} {sum(d) / len(d) / 1e3:9.2f} {d[len(d) // 2] / 1e3:9.2f} {d[0] / 1e3:9.2f} {d[-1] / 1e3:9.2f}")
