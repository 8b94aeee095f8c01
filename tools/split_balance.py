#!/usr/bin/env python3
"""Config 5 on ONE GPU, rank by rank: what each of `world` ranks would do under the three splits of the 64-camera batch --
contiguous row bands, whole cameras, interleaved 16-row tile bands -- timed one share after the other.  The slowest share
is what a barrier-to-barrier measurement over `world` GPUs sees; mean / max = the split's balance.
python tools/split_balance.py [world]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd")
par = importlib.import_module("sdf-viewer_amd.parallel")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
side, W, H, n = 256, 1920, 1080, 64
g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pkg.fill_grid(pkg.default_params(), g, t0, t1, dist=dist); pairs = pkg.commit_pairs(g, dist)
rp = pkg.default_render_params(g)
cams = pkg.orbit_cameras(n, aspect=W / H)
def run(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
res = {"world": world, "whole_batch_ms": round(run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, dist=dist, pairs=pairs)), 4)}
shares = {"rows": [], "cameras": [], "tiles": [], "tiles8": []}
for r in range(world):
    y0, y1 = par.split_rows(H, r, world)
    shares["rows"].append(run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, y0=y0, y1=y1, dist=dist, pairs=pairs)))
    mine = [cams[i] for i in par.split_cameras(n, r, world)]
    shares["cameras"].append(run(lambda: pkg.raymarch(rp, t0, t1, mine, W, H, dist=dist, pairs=pairs)))
    if hasattr(par, "split_bands"):
        first, step, _auto = par.split_bands(H, r, world)
        shares["tiles"].append(run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, bands=(first, step), dist=dist, pairs=pairs)))
        shares["tiles8"].append(run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, bands=(first, step, 8), dist=dist, pairs=pairs)))
K = pkg._capi
with pkg.options({K.OPT_RAYMARCH_BATCH_STREAMS: 0}):  # the tile shares' four launches one after the other
    shares["tiles_one_stream"] = [run(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, bands=(r, world, 16), dist=dist, pairs=pairs))
                                  for r in range(world)]
# a batch of small views (multi-view style): 64 cameras x 256 x 256, side streams on / off
small = pkg.orbit_cameras(n, aspect=1.0)
res["small_views_64x256x256_ms"] = {"side_streams": round(run(lambda: pkg.raymarch(rp, t0, t1, small, 256, 256, dist=dist, pairs=pairs)), 4)}
with pkg.options({K.OPT_RAYMARCH_BATCH_STREAMS: 0}):
    res["small_views_64x256x256_ms"]["one_stream"] = round(run(lambda: pkg.raymarch(rp, t0, t1, small, 256, 256, dist=dist, pairs=pairs)), 4)
res["band_height_split_bands_picks"] = par.split_bands(H, 0, world)[2]
for k, v in shares.items():
    if v:
        res[k] = {"ms_per_rank": [round(x, 4) for x in v], "max_ms": round(max(v), 4), "mean_ms": round(sum(v) / len(v), 4),
                  "balance": round(sum(v) / len(v) / max(v), 3), "speedup_over_one_gpu": round(res["whole_batch_ms"] / max(v), 2)}
print(json.dumps(res))
