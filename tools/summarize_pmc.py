#!/usr/bin/env python3
"""Collapses rocprofv3 counter-collection CSVs (one row per dispatch x counter) to per-kernel averages."""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        m = re.search(r"(fill_dense_kernel|fill_pass_kernel|raymarch_kernel|grid_init_kernel|FillFunctor|\w+_kernel)", row["Kernel_Name"])
        k = (m.group(1) if m else row["Kernel_Name"][:40], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    print("==", os.path.relpath(f, root))
    for (kern, ctr), (tot, n) in sorted(acc.items()):
        if True:
            print(f"{kern:62s} {ctr:22s} dispatches={n:4d} avg={tot / n:.4g}")
