#!/usr/bin/env python3
"""What the part of the image that misses the box costs: the default frame, a frame with nothing covered (camera looking
away), a plain memset of the image, and the rows that hold the box alone.  python tools/cull_cost.py"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("sdf-viewer_amd")
side, W, H = 256, 1920, 1080
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pkg.fill_grid(prm, g, t0, t1, dist=dist)
rp = pkg.default_render_params(g)
def run(fn, n=50, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
cam = pkg.camera_look_at(aspect=W / H)
away = pkg.camera_look_at(eye=(2.5, 3.0, 5.0), target=(5.0, 6.0, 10.0), aspect=W / H)
out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")
res = {"full": run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)),
       "looking_away_all_culled": run(lambda: pkg.raymarch(rp, t0, t1, away, W, H, out=out, dist=dist)),
       "memset_image": run(lambda: out.zero_())}
aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True, dist=dist)[1]
st = aux[0, :, :, 0]
rows = torch.nonzero((st != 0).any(dim=1)).flatten(); cols = torch.nonzero((st != 0).any(dim=0)).flatten()
y0, y1 = int(rows.min()) // 16 * 16, min(H, (int(rows.max()) // 16 + 1) * 16)
res["covered_rows"] = [int(rows.min()), int(rows.max())]; res["covered_cols"] = [int(cols.min()), int(cols.max())]
band = torch.empty((1, y1 - y0, W, 4), dtype=torch.float32, device="cuda")
res["rows_with_box_only"] = run(lambda: pkg.raymarch(rp, t0, t1, cam, W, H, y0=y0, y1=y1, out=band, dist=dist))
# batches: 64 cameras on the orbit, and 64 cameras looking away (every wave culled): what the background costs a batch
cams = pkg.orbit_cameras(64, aspect=W / H)
outb = torch.empty((64, H, W, 4), dtype=torch.float32, device="cuda")
res["batch64_orbit"] = run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=outb, dist=dist), n=5)
res["batch64_all_culled"] = run(lambda: pkg.raymarch(rp, t0, t1, [away] * 64, W, H, out=outb, dist=dist), n=5)
res["batch64_memset"] = run(lambda: outb.zero_(), n=5)
print(json.dumps(res))
