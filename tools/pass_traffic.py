#!/usr/bin/env python3
"""tools/gpu_profile_pass.sh's per-group rocprofv3 output -> one JSON: for every case of bench.py's `progressive` block the
kernels it runs, their average duration under the profiler and their HBM bytes per launch from the PMC passes (WRITE_SIZE +
2 x FETCH_SIZE KiB: MI355X_MICROARCH.md's gfx950 correction).  bench.py attaches `traffic` and `rocprof_ms` per case from it.
usage: pass_traffic.py <gpurun_out/prof_tag> <side>"""
import collections, csv, glob, json, os, re, sys

root, side = sys.argv[1], int(sys.argv[2])
n = side ** 3
rows2 = side * ((side + 1) // 2) ** 2   # threads of the whole-rows kernel at step 2
rows4 = side * ((side + 3) // 4) ** 2


def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", "").replace("sdfv::(anonymous namespace)::", "").replace("sdfv::", ""))


def trace(group):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, group, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), int(r.get("Grid_Size_X", 0) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1)))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc


def pmc(group, which):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, group, which, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = r.get("Grid_Size")
            grid = int(grid) if grid else int(r.get("Grid_Size_X", 0) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
            k = (short(r["Kernel_Name"]), grid)
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


# case of bench.py's progressive block -> (group it was profiled in, [(kernel name prefix, threads launched)])
# (name prefix, name suffix): the configuration policy in between is spelled DefaultCfgT<false> since round 4
DENSE, ROWS_T, ROWS_F, PASS, QUAD = (("fill_dense_kernel<256, true", ""), ("fill_pass_rows_kernel<", ", true>"), ("fill_pass_rows_kernel<", ", false>"),
                                     ("fill_pass_kernel<", ""), ("fill_pass_quad_kernel<", ""))
ADAPT = ("fill_pass_rows_adaptive_kernel<", "")  # round 5: an unflagged, unboxed pass with step >= 2 (a lane per 4 voxels of the visited rows)
CASES = {
    "virgin_load_2_passes": ("load_virgin", [(ROWS_T, rows2), (DENSE, n)]),
    "fresh_load_2_passes": ("load_virgin", [(ROWS_T, rows2), (DENSE, n)]),
    "fresh_pass_step_2_flagged": ("load_virgin", [(ROWS_T, rows2)]),
    "fresh_load_2_passes_unflagged": ("load_unflagged", [(ADAPT, rows2 // 4), (QUAD, n // 4)]),
    "fresh_pass_step_2": ("load_unflagged", [(ADAPT, rows2 // 4)]),
    "fresh_pass_step_1_after_step_2": ("load_unflagged", [(QUAD, n // 4)]),
    "fresh_pass_step_1": ("fresh_step1", [(QUAD, n // 4)]),
    "edit_full_box_3_passes": ("edit_full", [(ROWS_F, rows4), (ROWS_F, rows2), (DENSE, n)]),
    "edit_eighth_box_3_passes": ("edit_eighth", [(PASS, n // 64), (PASS, n // 8), (QUAD, n // 4)]),
    "noop_pass_step_1": ("noop", [(QUAD, n // 4)]),
}
try:  # which build / box the traces describe (this runs on the box, right after them)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import source_hash
    from bench_common import running_build_id
    STAMP = {"build_id": running_build_id(), "box": source_hash.box_uuid()}
except Exception:
    STAMP = {"build_id": None, "box": None}
out = {"side": side, **STAMP, "source": "tools/gpu_profile_pass.sh (rocprofv3 --kernel-trace / --pmc WRITE_SIZE / --pmc FETCH_SIZE, one group of cases "
                               "per process, warm, 100 repetitions); hbm_bytes = (WRITE_SIZE + 2 x FETCH_SIZE) KiB x 1024", "cases": {}}
cache = {}
for case, (group, kernels) in CASES.items():
    if group not in cache:
        cache[group] = (trace(group), pmc(group, "pmc_wr"), pmc(group, "pmc_rd"))
    tr, wr, rd = cache[group]
    entry = {"group": group, "kernels": [], "rocprof_ms": 0.0, "hbm_bytes": 0}

    def find(table, prefix, threads):
        hits = [k for k in table if k[0].startswith(prefix[0]) and k[0].rstrip().endswith(prefix[1])]
        exact = [k for k in hits if k[1] == threads]
        pool = exact or sorted(hits, key=lambda k: abs(k[1] - threads))[:1]  # (a kernel's grid is rounded up to whole workgroups)
        return pool[0] if pool and abs(pool[0][1] - threads) <= 1024 else None
    ok = True
    for prefix, threads in kernels:
        kt, kw, kr = find(tr, prefix, threads), find(wr, prefix, threads), find(rd, prefix, threads)
        if not (kt and kw and kr):
            ok = False
            entry["kernels"].append({"kernel": prefix[0] + "..." + prefix[1], "threads": threads, "error": "not found in the profile"})
            continue
        d = sorted(tr[kt])
        us = sum(d) / len(d) / 1e3
        b = int((wr[kw] + 2.0 * rd[kr]) * 1024)
        entry["kernels"].append({"kernel": kt[0], "threads": kt[1], "calls": len(d), "avg_us": round(us, 2), "median_us": round(d[len(d) // 2] / 1e3, 2),
                                 "WRITE_SIZE_KiB": round(wr[kw], 1), "FETCH_SIZE_KiB": round(rd[kr], 1), "hbm_bytes": b})
        entry["rocprof_ms"] += us / 1e3
        entry["hbm_bytes"] += b
    entry["rocprof_ms"] = round(entry["rocprof_ms"], 5)
    entry["complete"] = ok
    out["cases"][case] = entry
print(json.dumps(out, indent=1))
