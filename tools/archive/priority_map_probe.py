#!/usr/bin/env python3
"""Tuning build: would the frame be shorter if the waves that turn out long had issue priority from the start?  A first
frame yields every tile's longest ray (aux step counts); tiles at or above a threshold get s_setprio 3 at wave start in the
following frames (SDFV_OPT_TUNING_PRIORITY_MAP).  An oracle for the decision, not a product feature.
python tools/priority_map_probe.py [side=256]"""
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
rp = pkg.default_render_params(g); cam = pkg.camera_look_at(aspect=W / H)
out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
def run(fn, n=40, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True, dist=dist)[1]
steps = aux[0, :, :, 1].to(torch.int32)
ty, tx = (H + 15) // 16, (W + 15) // 16
pad = torch.zeros((ty * 16, tx * 16), dtype=torch.int32, device="cuda"); pad[:H, :W] = steps
tile_max = pad.view(ty, 16, tx, 16).amax(dim=(1, 3))
ref = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist).clone()
res = {"tiles": int(tx * ty), "no_map": []}
for rnd in range(3):
    pkg.set_option(K.OPT_TUNING_PRIORITY_MAP, 0)
    res["no_map"].append(run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)))
    for thr in (1, 16, 32, 48, 64, 96, 128, 192):
        m = (tile_max >= thr).to(torch.uint8).contiguous()
        pkg.set_option(K.OPT_TUNING_PRIORITY_MAP, m.data_ptr())
        ms = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist))
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
        res.setdefault(f"prio_tiles_with_max_steps_ge_{thr}", []).append([ms, int(m.sum())])
pkg.set_option(K.OPT_TUNING_PRIORITY_MAP, 0)
print(json.dumps(res))
