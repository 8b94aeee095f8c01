#!/usr/bin/env python3
"""The (y, z)-quad volume (16 B/voxel: a cell = 32 contiguous bytes) against the distance, y-pair and y-interleaved volumes:
single frames (1080p/256^3, 4K/512^3, a few views) and the 64-camera batch, alternating rounds in one process, bits compared.
python tools/quads_bench.py [torch]   (torch: build the volume with torch indexing instead of sdfv_commit_quads)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd")
def run(fn, n=40, warm=0.05):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def quads_by_torch(d):
    """[D, H, W] -> [D, H, W, 4]: (d[z][y], d[z+1][y], d[z][y+1], d[z+1][y+1]) at x, the +1 clamped at the last row / slice."""
    D, H, W = d.shape
    zi = torch.clamp(torch.arange(D, device=d.device) + 1, max=D - 1)
    yi = torch.clamp(torch.arange(H, device=d.device) + 1, max=H - 1)
    dz, dy = d[zi], d[:, yi]
    return torch.stack([d, dz, dy, dz[:, yi]], dim=-1).contiguous()
use_torch = len(sys.argv) > 1 and sys.argv[1] == "torch"
res = {}
prm = pkg.default_params()
for side, W, H in ((256, 1920, 1080), (512, 3840, 2160)):
    g = pkg.make_grid((side,) * 3)
    t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    pairs = pkg.commit_pairs(g, dist)
    ilv = pkg.commit_interleaved(g, dist)
    if use_torch or not hasattr(pkg, "commit_quads"):
        quads = quads_by_torch(dist)
    else:
        quads = pkg.commit_quads(g, dist)
        res[f"{side}_commit_quads_ms"] = round(run(lambda: pkg.commit_quads(g, dist, quads=quads), n=20), 4)
        res[f"{side}_commit_pairs_ms"] = round(run(lambda: pkg.commit_pairs(g, dist, pairs=pairs), n=20), 4)
        assert torch.equal(quads, quads_by_torch(dist))
    rp = pkg.default_render_params(g)
    outs = {k: torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda") for k in ("dist", "pairs", "ilv", "quads")}
    views = {"default": pkg.camera_look_at(aspect=W / H)}
    for k, c in enumerate(pkg.orbit_cameras(8, aspect=W / H)[1::2]): views[f"orbit{2 * k + 1}"] = c
    views["close"] = pkg.camera_look_at(eye=(1.2, 1.5, 2.4), aspect=W / H)
    views["far"] = pkg.camera_look_at(eye=(5.0, 6.0, 10.0), aspect=W / H)
    views["inside"] = pkg.camera_look_at(eye=(0.2, 0.1, 0.3), target=(1.0, 0.5, -1.0), aspect=W / H)
    kws = {"dist": {}, "pairs": {"pairs": pairs}, "ilv": {"ilv": ilv}, "quads": {"quads": quads}}
    for name, cam in views.items():
        ms = {k: [] for k in kws}
        for rnd in range(3):
            for k, kw in kws.items():
                ms[k].append(run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=outs[k], dist=dist, **kw)))
        same = all(bool(torch.equal(outs[k].view(torch.int32), outs["dist"].view(torch.int32))) for k in kws)
        res[f"{side}_{W}x{H}_{name}"] = {**{f"{k}_ms": round(min(v), 4) for k, v in ms.items()}, "same_bits": same}
        print(side, name, res[f"{side}_{W}x{H}_{name}"], file=sys.stderr, flush=True)
    if side == 256:
        cams = pkg.upload_cameras(pkg.orbit_cameras(64, aspect=W / H))
        big = torch.empty((64, H, W, 4), dtype=torch.float32, device="cuda"); big2 = torch.empty_like(big)
        rec = {}
        for k, kw in kws.items():
            o = big if k == "dist" else big2
            rec[f"{k}_ms"] = round(min(run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=o, dist=dist, **kw), n=5) for _ in range(3)), 4)
            rec[f"Mrays_s_{k}"] = round(64 * W * H / rec[f"{k}_ms"] / 1e3, 1)
            if k != "dist":
                rec["same_bits"] = rec.get("same_bits", True) and bool(torch.equal(big.view(torch.int32), big2.view(torch.int32)))
        res["256_batch64"] = rec
        print("batch64", rec, file=sys.stderr, flush=True)
        del big, big2
    del t0, t1, dist, pairs, ilv, quads
print(json.dumps(res))
