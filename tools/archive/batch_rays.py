#!/usr/bin/env python3
"""Throughput of the batched raymarch (BASELINE.json config 5 shape on ONE GPU): n cameras x 1080p per call."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("sdf-viewer_amd")
side, W, H = 256, 1920, 1080
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); pkg.fill_grid(prm, g, t0, t1)
rp = pkg.default_render_params(g)
for n in (1, 2, 4, 8, 16, 64):
    cams = pkg.orbit_cameras(n, aspect=W / H)
    out = torch.empty((n, H, W, 4), dtype=torch.float32, device="cuda")
    for _ in range(2): pkg.raymarch(rp, t0, t1, cams, W, H, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps): pkg.raymarch(rp, t0, t1, cams, W, H, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{n:3d} cameras: {ms:.3f} ms/batch  {n * W * H / ms / 1e3:.0f} Mrays/s  hit fraction {float((out[..., 3] > 0).float().mean()):.3f}")
