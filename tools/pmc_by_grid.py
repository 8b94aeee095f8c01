#!/usr/bin/env python3
"""rocprofv3 counter-collection CSVs -> per (kernel, grid size, counter) averages.  usage: pmc_by_grid.py <dir with pmc_*/>"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void ", "").replace("sdfv::(anonymous namespace)::", ""))
        grid = row.get("Grid_Size") or "x".join(str(row.get(k, "")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        k = (name[:64], grid, row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    print("==", os.path.relpath(f, root))
    for (kern, grid, ctr), (tot, n) in sorted(acc.items()):
        print(f"{kern:64s} grid {grid:>12s} {ctr:18s} dispatches={n:4d} avg={tot / n:.5g}")
