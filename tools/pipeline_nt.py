#!/usr/bin/env python3
"""Interleaved fill -> march pipeline (the real per-frame sequence) with plain vs non-temporal texture stores in the dense
fill: does keeping the write-once textures out of L2 help the march that follows (it re-reads only the distance volume and
a few texels under the hits)?  python tools/pipeline_nt.py [side=256]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = (1920, 1080) if side <= 256 else (3840, 2160)
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
rp = pkg.default_render_params(g); cam = pkg.camera_look_at(aspect=W / H)
rgba = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
def run(fn, n=30, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
res = {}
for rnd in range(3):
    for nt in (0, 1):
        pkg.set_option(K.OPT_FILL_NONTEMPORAL, 1 if nt else 2)
        def fused():
            pkg.fill_grid(prm, g, t0, t1, dist=dist); pkg.raymarch(rp, t0, t1, cam, W, H, out=rgba, dist=dist)
        def plain():
            pkg.fill_grid(prm, g, t0, t1); pkg.raymarch(rp, t0, t1, cam, W, H, out=rgba)
        res.setdefault(f"fused_nt{nt}", []).append(run(fused))
        res.setdefault(f"plain_nt{nt}", []).append(run(plain))
        res.setdefault(f"fill_fused_nt{nt}", []).append(run(lambda: pkg.fill_grid(prm, g, t0, t1, dist=dist)))
        res.setdefault(f"fill_plain_nt{nt}", []).append(run(lambda: pkg.fill_grid(prm, g, t0, t1)))
pkg.set_option(K.OPT_FILL_NONTEMPORAL, 0)
res["march_dist"] = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=rgba, dist=dist))
res["march_tex0"] = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=rgba))
print(json.dumps({"side": side, "ms": res}))
