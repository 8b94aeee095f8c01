#!/usr/bin/env python3
"""Data for the launcher's occupancy rule: single frames, eleven views x {1080p/256^3, 1440p/512^3,
4K/512^3}, every cap on the resident waves per SIMD (SDFV_OPT_RAYMARCH_WAVES_PER_SIMD; 0 = the register file's 7 -- or,
with the rule in, the launcher's own choice), plus what the launcher can know about the view: the screen rectangle of the
projected bounding box in 16 x 16 tiles.
python tools/occupancy_rule_bench.py [--auto-only]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
auto_only = "--auto-only" in sys.argv
CONFIGS = [(256, 1920, 1080), (512, 2560, 1440), (512, 3840, 2160)]
def run(fn, n=30, warm=0.05):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 4)
out_all = {}
prm = pkg.default_params()
for side, W, H in CONFIGS:
    g = pkg.make_grid((side,) * 3)
    t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    rp = pkg.default_render_params(g)
    out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
    views = {"default": pkg.camera_look_at(aspect=W / H)}
    for k, c in enumerate(pkg.orbit_cameras(8, aspect=W / H)[1:]):
        views[f"orbit{k + 1}"] = c
    views["close"] = pkg.camera_look_at(eye=(1.2, 1.5, 2.4), aspect=W / H)
    views["closer"] = pkg.camera_look_at(eye=(0.9, 1.1, 1.8), aspect=W / H)
    views["far"] = pkg.camera_look_at(eye=(5.0, 6.0, 10.0), aspect=W / H)
    views["axis"] = pkg.camera_look_at(eye=(0.0, 0.0, 5.0), aspect=W / H)
    views["inside"] = pkg.camera_look_at(eye=(0.2, 0.1, 0.3), target=(1.0, 0.5, -1.0), aspect=W / H)
    res = {}
    for name, cam in views.items():
        pkg.set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, 0)
        pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)
        torch.cuda.synchronize()
        hit = out[0, ..., 3] > 0
        tiles = hit.view(H // 8, 8, W // 8, 8).any(dim=3).any(dim=1) if H % 8 == 0 and W % 8 == 0 else None
        ys, xs = torch.nonzero(hit, as_tuple=True)
        info = {"hit_pixels": int(hit.sum()), "hit_waves": int(tiles.sum()) if tiles is not None else None,
                "rect_px": [int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1] if len(xs) else None}
        ms = {}
        for rnd in range(2):
            for cap in ([0] if auto_only else [0, 6, 5, 4, 3, 2]):
                pkg.set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, cap)
                v = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist))
                ms[str(cap)] = min(ms.get(str(cap), 1e9), v)
        pkg.set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, 0)
        info["ms_by_cap"] = ms
        res[name] = info
        print(f"{side} {W}x{H} {name:8s} {info['hit_pixels']:8d} px {info['hit_waves']} waves  " + "  ".join(f"{c}:{v:.4f}" for c, v in ms.items()), file=sys.stderr, flush=True)
    out_all[f"{side}_{W}x{H}"] = res
    del t0, t1, dist, out
print(json.dumps(out_all))
