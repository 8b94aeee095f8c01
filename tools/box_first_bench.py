#!/usr/bin/env python3
"""Raymarch, single frame: groups of tiles under the projected bounding box launched first (SDFV_OPT_RAYMARCH_BOX_FIRST)
vs plain group order.  python tools/box_first_bench.py [side=256]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = (1920, 1080) if side <= 256 else (3840, 2160)
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pkg.fill_grid(prm, g, t0, t1, dist=dist)
rp = pkg.default_render_params(g)
def run(fn, n=40, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
res = {}
out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
views = {"default": pkg.camera_look_at(aspect=W / H)}
for k, c in enumerate(pkg.orbit_cameras(8, aspect=W / H)[1:]):
    views[f"orbit{k + 1}"] = c
views["close"] = pkg.camera_look_at(eye=(1.2, 1.5, 2.4), aspect=W / H)
views["far"] = pkg.camera_look_at(eye=(5.0, 6.0, 10.0), aspect=W / H)
views["axis"] = pkg.camera_look_at(eye=(0.0, 0.0, 5.0), aspect=W / H)
for name, cam in views.items():
    for group in (0, 1, 2, 3):
        for first in (0, 1):
            if group == 1 and first or group == 0 and not first:
                continue
            pkg.set_option(K.OPT_RAYMARCH_TILE_GROUP, group); pkg.set_option(K.OPT_RAYMARCH_BOX_FIRST, first)
            res.setdefault(f"dist_group{group}_first{first}", {})[name] = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist), n=20, warm=0.05)
            res.setdefault(f"tex0_group{group}_first{first}", {})[name] = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out), n=20, warm=0.05)
for k in res:
    res[k]["sum"] = round(sum(res[k].values()), 4)
print(json.dumps({"side": side, "res": res}))
