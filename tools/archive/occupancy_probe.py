#!/usr/bin/env python3
"""The raymarch kernel's resident waves per SIMD capped by unused dynamic LDS (SDFV_OPT_RAYMARCH_WAVES_PER_SIMD), single
frames and batches, one binary.  160 KB of LDS per CU, workgroups of 4 waves (one per SIMD): a w-th of it per workgroup
allows w workgroups = w waves per SIMD (the register file allows 7).  python tools/occupancy_probe.py [side=256]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = (1920, 1080) if side <= 256 else (3840, 2160)
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pkg.fill_grid(prm, g, t0, t1, dist=dist)
rp = pkg.default_render_params(g)
def run(fn, n, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
res = {}
caps = {7: 0, 6: 6, 5: 5, 4: 4, 3: 3}
for ncam in (1, 16, 64):
    cams = pkg.orbit_cameras(ncam, aspect=W / H)
    out = torch.empty((ncam, H, W, 4), dtype=torch.float32, device="cuda")
    for rnd in range(2):
        for waves, lds in caps.items():
            pkg.set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, lds)
            res.setdefault(f"cams{ncam}_waves{waves}", []).append(run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=out, dist=dist), 40 if ncam == 1 else 5))
pkg.set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, 0)
print(json.dumps({"side": side, "ms": res}))
