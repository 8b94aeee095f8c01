#!/usr/bin/env python3
"""Resident waves over time in ONE 64-camera batch launch (tuning build: per-wave start / end stamps of the device-wide
100 MHz counter, 16 bits each, unwrapped along the dispatch order).  Prints the average number of resident waves per SIMD,
of MARCHING waves among them, and a coarse timeline.  python tools/batch_timeline.py [n_cameras=64] [--json out]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SDFGRID_LIBRARY"] = os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid_tuning.so")
import numpy as np
import torch
pkg = importlib.import_module("sdf-viewer_amd")
K = pkg._capi
n_cam = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
side, W, H = 256, 1920, 1080
g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pkg.fill_grid(pkg.default_params(), g, t0, t1, dist=dist)
pairs = pkg.commit_pairs(g, dist)
rp = pkg.default_render_params(g)
cams = pkg.upload_cameras(pkg.orbit_cameras(max(n_cam, 17), aspect=W / H)) if n_cam > 16 else pkg.orbit_cameras(n_cam, aspect=W / H)
out = torch.empty((max(n_cam, 17) if n_cam > 16 else n_cam, H, W, 4), dtype=torch.float32, device="cuda")
tiles = ((W + 15) // 16) * ((H + 15) // 16)
n_waves = tiles * 4 * out.shape[0]
buf = torch.zeros((n_waves, 4), dtype=torch.int64, device="cuda")
if "--plain-order" in sys.argv:
    pkg.set_option(K.OPT_RAYMARCH_TILE_GROUP, 1)
for _ in range(3):
    pkg.raymarch(rp, t0, t1, cams, W, H, out=out, dist=dist, pairs=pairs)
pkg.set_option(K.OPT_TUNING_WAVE_TIMING, buf.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
pkg.raymarch(rp, t0, t1, cams, W, H, out=out, dist=dist, pairs=pairs)
b.record(); torch.cuda.synchronize()
pkg.set_option(K.OPT_TUNING_WAVE_TIMING, 0)
launch_us = a.elapsed_time(b) * 1e3
d = buf.cpu().numpy()
it = d[:, 2] & 0xffff
rt0 = ((d[:, 2] >> 32) & 0xffff).astype(np.int64)
rt1 = ((d[:, 2] >> 48) & 0xffff).astype(np.int64)
dur = (rt1 - rt0) & 0xffff  # ticks of 10 ns; a wave lives far less than 655 us
# unwrap the starts along the dispatch order (camera-major, tiles in launch order): consecutive starts are close
step = np.diff(rt0, prepend=rt0[0])
step = ((step + 0x8000) & 0xffff) - 0x8000
start = np.cumsum(step) + 0
start -= start.min()
end = start + dur
span = end.max()
marching = it > 0
res = {"cameras": int(out.shape[0]), "launch_us_events": round(launch_us, 1), "span_us_stamps": round(span * 0.01, 1), "waves": int(n_waves),
       "marching_waves": int(marching.sum()), "mean_wave_us": round(float(dur.mean()) * 0.01, 3),
       "mean_marching_wave_us": round(float(dur[marching].mean()) * 0.01, 3), "simds": 1024}
res["avg_resident_waves_per_simd"] = round(float(dur.sum()) / span / 1024, 2)
res["avg_resident_marching_waves_per_simd"] = round(float(dur[marching].sum()) / span / 1024, 2)
# timeline: resident waves at 40 instants
ts = np.linspace(0, span, 42)[1:-1]
order_s, order_e = np.sort(start), np.sort(end)
res["timeline_resident_per_simd"] = [round(float(np.searchsorted(order_s, t, "right") - np.searchsorted(order_e, t, "right")) / 1024, 2) for t in ts]
ms, me = np.sort(start[marching]), np.sort(end[marching])
res["timeline_marching_per_simd"] = [round(float(np.searchsorted(ms, t, "right") - np.searchsorted(me, t, "right")) / 1024, 2) for t in ts]
# per XCD (workgroup L of the launch runs on XCD L % 8): when does each finish, how full is it while it runs
wg = np.arange(n_waves) // 4  # stamps are indexed by TILE: (camera, tile row, tile column)
tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
col, row, cam_of = wg % tiles_x, (wg // tiles_x) % tiles_y, wg // (tiles_x * tiles_y)
if "--plain-order" in sys.argv:  # SDFV_OPT_RAYMARCH_TILE_GROUP 1: workgroup x renders column x
    xcd = (col + row * tiles_x + cam_of * tiles_x * tiles_y) % 8
else:  # the product's order: workgroup (x, y, z) renders column (x + y + z) mod tiles_x
    xcd = (((col - row - cam_of) % tiles_x) + row * tiles_x + cam_of * tiles_x * tiles_y) % 8
res["per_xcd"] = []
for k in range(8):
    m = xcd == k
    s_k, e_k = start[m], end[m]
    res["per_xcd"].append({"first_start_us": round(float(s_k.min()) * 0.01, 1), "last_end_us": round(float(e_k.max()) * 0.01, 1),
                           "last_start_us": round(float(s_k.max()) * 0.01, 1),
                           "avg_resident_per_simd": round(float(dur[m].sum()) / float(e_k.max() - s_k.min()) / 128, 2),
                           "marching_wave_us_total": round(float(dur[m & marching].sum()) * 0.01, 0)})
# how long after the LAST dispatch does the launch go on (the tail), and the longest waves
res["last_dispatch_us"] = round(float(start.max()) * 0.01, 1)
res["longest_waves_us"] = [round(float(x) * 0.01, 1) for x in np.sort(dur)[-5:]]
print(json.dumps(res))
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
