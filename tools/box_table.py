#!/usr/bin/env python3
"""One row per bench.py line found in the given files / directories: which box (rocm-smi unique id, clocks as stamped by the
line itself) gave which numbers -- so that a reader can tell a regression from a slow box.
python tools/box_table.py <file-or-dir>... [--out profiles/r04_box_table.json]"""
import glob, json, os, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
if out in args:
    args.remove(out)
files = []
for a in args:
    files += sorted(glob.glob(os.path.join(a, "**", "*.json"), recursive=True)) if os.path.isdir(a) else [a]
rows = []
for f in files:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        continue
    if not isinstance(d, dict) or "box" not in d or "metric" not in d:
        continue
    smi = d["box"].get("rocm_smi", {})
    if not isinstance(smi, dict):
        smi = {}
    if "uuid" in d["box"] and "Unique ID" not in smi:  # round 5's contract line: box = {uuid, arch, sclk}
        smi = {"Unique ID": d["box"].get("uuid"), "sclk clock speed:": d["box"].get("sclk")}
    host = d.get("host_load") or {}
    t512 = (d.get("roofline") or {}).get("target_512") or {}
    rows.append({"file": f, "box": smi.get("Unique ID"), "sclk": smi.get("sclk clock speed:"), "workload": d["config"].get("workload"),
                 "host_512_update_ms": (host.get("512") or {}).get("update_ms") if "update_ms" in (host.get("512") or {}) else ((host.get("512") or {}).get("dense") or {}).get("update_ms"),
                 "n_gpus": d["n_gpus"], "pipeline": d.get("pipeline"), "fill_ms": d.get("ms_per_step_fill"),
                 "march_ms": d.get("ms_per_step_raymarch"), "value_Mvoxels_s": d["value"], "frac": (d.get("roofline") or {}).get("frac"),
                 "frac_bus": (d.get("roofline") or {}).get("frac_bus"),
                 "plain_512_ms": (t512.get("plain") or {}).get("ms"), "fused_512_ms": (t512.get("fused") or {}).get("ms"),
                 "batch_ms": (d.get("batch_raymarch") or {}).get("ms_per_batch")})
rows.sort(key=lambda r: (str(r["box"]), r["file"]))
for r in rows:
    print(json.dumps(r))
if out:
    json.dump({"what": "bench.py lines of the round by box (tools/box_table.py)", "rows": rows}, open(out, "w"), indent=1)
