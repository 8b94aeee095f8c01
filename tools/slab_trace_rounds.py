#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of tools/sharded_march_bench.py and splits the LAST replay of the sharded march into its
rounds: per round the duration of every rank's raymarch_slab_kernel launch, their sum (what the one-GPU loopback pays) and
their maximum (what eight GPUs running side by side would pay, before the xGMI exchange).
python tools/slab_trace_rounds.py <kernel_trace.csv> <world> [out.json]"""
import json, sys
import pandas as pd
t = pd.read_csv(sys.argv[1])
world = int(sys.argv[2])
t = t[t.Kernel_Name.str.contains("raymarch_slab_kernel")].sort_values("Start_Timestamp")
dur = ((t.End_Timestamp - t.Start_Timestamp) / 1e3).tolist()
threads = t.Grid_Size_X.tolist()
first = max(threads)  # round one launches one thread per pixel; later rounds one per slot of the two ray lists
starts = [k for k in range(len(threads) - world + 1) if all(threads[k + r] == first for r in range(world)) and (k == 0 or threads[k - 1] != first)]
k = starts[-1]
rounds = []
while k + world <= len(dur) and (not rounds or threads[k] != first):
    d = [round(x, 1) for x in dur[k:k + world]]
    rounds.append({"threads_per_rank": int(threads[k]), "us_per_rank": d, "sum_us": round(sum(d), 1), "max_us": max(d)})
    k += world
res = {"world": world, "rounds": rounds, "sum_of_sums_us": round(sum(r["sum_us"] for r in rounds), 1),
       "sum_of_maxima_us": round(sum(r["max_us"] for r in rounds), 1),
       "note": "last replay in the trace; sum_of_maxima_us = the kernels on the critical path when every rank has its own GPU"}
print(json.dumps(res))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
