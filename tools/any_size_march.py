#!/usr/bin/env python3
"""The hand-written march loop on grids of any size (rows addressed by 24-bit multiplies, not shifts): 1080p frames over
cubic / non-cubic, power-of-two / other grids -- the compiler's loop against the hand-written one over tex0.r, the distance,
pair and interleaved volumes, bits compared.  python tools/any_size_march.py > profiles/r03_any_size_march.json"""
import json
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdf-viewer_amd")
K = pkg._capi
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
W, H = 1920, 1080
prm = pkg.default_params()
report = {"image": [W, H], "note": "default camera; a non-cubic grid handed dist + pairs / ilv marches over dist (the launcher's choice)"}
for dims in ((256,) * 3, (250,) * 3, (255,) * 3, (300,) * 3, (384,) * 3, (200, 300, 150), (256, 256, 128), (512,) * 3, (500,) * 3):
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g); dist = torch.empty(dims[::-1], dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    pairs = pkg.commit_pairs(g, dist); ilv = pkg.commit_interleaved(g, dist) if dims[1] % 2 == 0 else None
    rp = pkg.default_render_params(g); cam = pkg.camera_look_at(aspect=W / H)
    out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda"); ref = torch.empty_like(out)
    with pkg.options({K.OPT_RAYMARCH_DISABLE: K.RM_NO_ASM_LOOP}):
        tmc = timeit(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=ref, dist=dist))
    res = {}
    for name, kw in (("dist", dict(dist=dist)), ("tex0", dict()), ("pairs", dict(dist=dist, pairs=pairs)), ("ilv", dict(dist=dist, ilv=ilv))):
        if name == "ilv" and ilv is None: continue
        res[name] = (round(timeit(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, **kw)), 4), bool(torch.equal(out.view(torch.int32), ref.view(torch.int32))))
    print(f"{dims}: compiler-loop dist {tmc:.4f}  asm {res}", file=sys.stderr, flush=True)
    report["x".join(map(str, dims))] = {"compiler_loop_dist_ms": round(tmc, 4), **{k + "_ms": v[0] for k, v in res.items()},
                                        "same_bits": all(v[1] for v in res.values())}
    del t0, t1, dist, pairs, ilv
print(json.dumps(report))
