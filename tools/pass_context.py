#!/usr/bin/env python3
"""Why does the whole-rows pass (fill_pass_rows_kernel<.., true>, step 2 over 256^3: 151 MB of stores) take 28 us in bench.py's
`fresh_pass_step_2_flagged` and 42 us under rocprofv3?  The same launch timed with HIP events after
different predecessors on the stream.   python tools/pass_context.py [side]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
placed = "--separate" not in sys.argv
t0, t1 = pkg.alloc_textures_placed(g) if placed else pkg.alloc_textures(g)
dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
other = torch.empty(64 << 20, dtype=torch.float32, device="cuda")  # 256 MB elsewhere
VS = K.PASS_VIRGIN_GRID | K.PASS_SAME_LOAD
rows = lambda: pkg.fill_grid_pass(prm, g, 2, t0, t1, dist=dist, flags=VS)
pred = {"sdfv_grid_init + dist.fill_ (bench.py's setup)": lambda: (pkg.grid_init(g, t0, t1), dist.fill_(pkg.AIR_DIST)),
        "the fused dense fill (the 2-pass load's previous repetition)": lambda: pkg.fill_grid(prm, g, t0, t1, dist=dist),
        "the plain dense fill": lambda: pkg.fill_grid(prm, g, t0, t1),
        "itself": rows,
        "a 256 MB fill_ of other memory": lambda: other.fill_(1.0),
        "nothing (stream idle, synchronised)": lambda: torch.cuda.synchronize()}
out = {"side": side, "kernel": "fill_pass_rows_kernel<DefaultCfg, FRESH> step 2",
       "placement": "one block, tex1 at pkg.default_texture_skew() after tex0" if placed else "two separate allocations",
       "tex1_minus_tex0_mod_16KiB": (t1.data_ptr() - t0.data_ptr()) % 16384, "after": {}}
for _ in range(30): pkg.fill_grid(prm, g, t0, t1, dist=dist)
for name, p in pred.items():
    ts = []
    for _ in range(30):
        p()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rows(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    out["after"][name] = {"median_us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1), "max_us": round(ts[-1], 1)}
print(json.dumps(out, indent=1))
