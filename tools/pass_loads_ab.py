#!/usr/bin/env python3
"""SDFV_OPT_PASS_LOADS A/B on ONE box in ONE process: bench.py's `progressive` cases (bench_extras.progressive_block) with the
update_required scan reading the volume through cached loads (1) and through nontemporal loads (2), interleaved twice.
usage: python tools/pass_loads_ab.py [side ...]   -> JSON {side: {case: [cached ms, nt ms, nt / cached]}}"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sdf-viewer_amd")
from bench_extras import progressive_block  # noqa: E402

sides = [int(a) for a in sys.argv[1:]] or [256, 512]
prm = pkg.default_params()
runs = {1: [], 2: []}
for rnd in range(2):
    for mode in (1, 2):
        with pkg.options({pkg._capi.OPT_PASS_LOADS: mode}):
            runs[mode].append(progressive_block(pkg, torch, prm, sides))
out = {"build_id": pkg.lib.sdfv_build_id().decode()}
for side in sides:
    res = {}
    for case, v in runs[1][0][str(side)].items():
        if not isinstance(v, dict) or "ms" not in v:
            continue
        a = min(r[str(side)][case]["ms"] for r in runs[1])
        b = min(r[str(side)][case]["ms"] for r in runs[2])
        res[case] = [a, b, round(b / a, 3)]
    out[str(side)] = res
print(json.dumps(out, indent=1))
