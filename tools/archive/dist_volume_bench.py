#!/usr/bin/env python3
"""Raymarch with and without the compact distance volume (sdfv_commit_distance) at both bench workloads."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("sdf-viewer_amd")
def timed(fn, reps=10):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for side, W, H in ((256, 1920, 1080), (512, 3840, 2160), (512, 1920, 1080)):
    prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
    t0, t1 = pkg.alloc_textures(g); pkg.fill_grid(prm, g, t0, t1)
    rp = pkg.default_render_params(g)
    out = torch.empty((1, H, W, 4), device="cuda")
    ms_commit = timed(lambda: pkg.commit_distance(g, t0))
    dist = pkg.commit_distance(g, t0)
    for ncam in (1, 16):
        cams = pkg.orbit_cameras(ncam, aspect=W / H)
        o = torch.empty((ncam, H, W, 4), device="cuda")
        a = timed(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=o))
        b = timed(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=o, dist=dist))
        print(f"{side}^3 {W}x{H} x{ncam:2d}: tex0.r {a:.3f} ms ({ncam * W * H / a / 1e3:.0f} Mrays/s)   distance volume {b:.3f} ms "
              f"({ncam * W * H / b / 1e3:.0f} Mrays/s)   commit {ms_commit:.3f} ms")
    del t0, t1, dist
