#!/usr/bin/env python3
"""The tex0 <-> tex1 distance within a 2 MiB page decides 6-8 % of the fill's rate, reproducibly per grid size (tools/
place3_sweep.py).  Is there a rule in the row width?  Fused fill, one block per shape, skews 0 .. 60 KiB in 4 KiB steps.
python tools/place_width_sweep.py"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdf-viewer_amd")
prm = pkg.default_params()
def ms(fn, reps):
    for _ in range(max(5, reps // 4)): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
shapes = [(128, 128, 128), (128, 512, 256), (192, 192, 192), (256, 256, 256), (256, 512, 128), (320, 320, 320), (384, 384, 384), (512, 256, 128),
          (512, 512, 512), (640, 640, 160), (768, 768, 96), (1024, 1024, 64), (1024, 256, 64), (250, 250, 250), (300, 300, 300)]
skews = [k * 4096 for k in range(0, 16)]
out = {}
for dims in shapes:
    g = pkg.make_grid(dims)
    nv = dims[0] * dims[1] * dims[2]; n = nv * 4
    big = torch.empty(2 * n + (1 << 20) // 4, dtype=torch.float32, device="cuda")
    dvol = torch.empty((dims[2], dims[1], dims[0]), dtype=torch.float32, device="cuda")
    pad = (-big.data_ptr()) % (2 << 20)
    reps = max(20, min(400, int(3e9 / (nv * 36))))
    r = {}
    for rnd in range(2):
        for s in skews:
            o0 = pad // 4; o1 = o0 + n + s // 4
            if o1 + n > big.numel(): continue
            t0 = big[o0:o0 + n].view(dims[2], dims[1], dims[0], 4); t1 = big[o1:o1 + n].view(dims[2], dims[1], dims[0], 4)
            r.setdefault(s, []).append(ms(lambda: pkg.fill_grid(prm, g, t0, t1, dist=dvol), reps))
    avg = {s: sum(v) / len(v) for s, v in r.items()}
    best = min(avg, key=avg.get)
    out["x".join(map(str, dims))] = {"row_bytes": dims[0] * 16, "tex_bytes_mod_2MiB": (n * 4) % (2 << 20), "best_skew": best,
                                     "rel": {str(s): round(avg[s] / avg[best], 3) for s in skews if s in avg}, "best_ms": round(avg[best], 4)}
    del big, dvol
print(json.dumps(out))
