#!/usr/bin/env python3
"""Tuning build: what would longest-first tile scheduling buy?  A first frame yields every tile's longest ray (aux step
counts); following frames render the tiles in orders built from them (SDFV_OPT_TUNING_TILE_ORDER): longest first, longest
first with the XCDs dealt round-robin (position L -> XCD L % 8 anyway), and the reverse as a control.
python tools/tile_order_probe.py [side=256]   (256: 1080p, 512: 4K)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SDFGRID_LIBRARY"] = os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid_tuning.so")
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = (1920, 1080) if side <= 256 else (3840, 2160)
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pkg.fill_grid(prm, g, t0, t1, dist=dist)
rp = pkg.default_render_params(g)
out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
def run(fn, n=30, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
ty, tx = (H + 15) // 16, (W + 15) // 16
def tile_cost(cam):
    aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True, dist=dist)[1]
    steps = aux[0, :, :, 1].to(torch.int32)
    pad = torch.zeros((ty * 16, tx * 16), dtype=torch.int32, device="cuda"); pad[:H, :W] = steps
    return pad.view(ty, 16, tx, 16).amax(dim=(1, 3)).flatten()
views = {"default": pkg.camera_look_at(aspect=W / H)}
cams = pkg.orbit_cameras(64, aspect=W / H)
res = {}
for name, cam, prev in (("default", views["default"], views["default"]), ("orbit_step", cams[1], cams[0]), ("orbit_far", cams[8], cams[0])):
    cost = tile_cost(prev)                      # the PREVIOUS frame's costs (same frame for "default")
    pkg.set_option(K.OPT_TUNING_TILE_ORDER, 0)
    ref = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist).clone()
    r = {"product_order": [], "longest_first": [], "shortest_first": []}
    lf = torch.argsort(cost, descending=True, stable=True).to(torch.int32).contiguous()
    sf = torch.argsort(cost, descending=False, stable=True).to(torch.int32).contiguous()
    for rnd in range(3):
        pkg.set_option(K.OPT_TUNING_TILE_ORDER, 0)
        r["product_order"].append(run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)))
        for key, order in (("longest_first", lf), ("shortest_first", sf)):
            pkg.set_option(K.OPT_TUNING_TILE_ORDER, order.data_ptr())
            r[key].append(run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)))
            assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    res[name] = r
pkg.set_option(K.OPT_TUNING_TILE_ORDER, 0)
print(json.dumps({"side": side, "image": [W, H], "ms": res}))
