#!/usr/bin/env python3
"""The bench line(s) a profiled bench.py process printed, out of its combined stdout + stderr log: the full record (stderr,
"[bench full record] {...}") reduced to what a profile summary's header needs, so that the rocprof averages below it and the
HIP-event averages of the SAME process stand side by side.  usage: bench_line_of.py <log>"""
import json
import sys

full = None
for line in open(sys.argv[1], errors="replace"):
    if line.startswith("[bench full record] "):
        full = json.loads(line[len("[bench full record] "):])
if full is None:
    print("# (no bench record in the log)")
    sys.exit(0)
smi = ((full.get("box") or {}).get("rocm_smi") or {}) if isinstance((full.get("box") or {}).get("rocm_smi"), dict) else {}
print(f"# build_id: {full.get('build_id')}  box: {smi.get('Unique ID') or (full.get('box') or {}).get('uuid')}")
keep = ("value", "value_rays", "ms_per_step", "ms_per_step_fill", "ms_per_step_raymarch", "pipeline", "texture_placement", "raymarch_kernel_ms")
print("# bench line of THIS process:", json.dumps({k: full.get(k) for k in keep}))
r = full.get("roofline") or {}
print("# roofline (HIP events over the K timed launches):", json.dumps({k: r.get(k) for k in ("kernel", "avg_launch_ms", "achieved", "frac", "frac_bus", "traffic")}))
for name in ("pipeline_plain", "pipeline_fused", "pipeline_fused_ilv"):
    p = full.get(name)
    if isinstance(p, dict) and p.get("ms_fill") not in (None, float("inf")):
        print(f"# {name}:", json.dumps({k: p.get(k) for k in ("ms_fill", "ms_raymarch", "fill_frac_8d")}))
print("# box:", json.dumps((full.get("box") or {}).get("rocm_smi")))
