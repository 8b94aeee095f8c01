#!/usr/bin/env python3
"""What handing over a batch of cameras costs (round 4: up to 16 ride in the kernel arguments, more are read from device memory):
a 64-camera batch of tiny images (the march itself is a few microseconds), cameras as a host array copied by the launcher
(SDFV_OPT_RAYMARCH_CAMERA_STAGING 1), as launches of 16 (0), and as a device array read in place -- host time per call and
stream time per call.  python tools/camera_staging_cost.py [n_cameras=64] [side=32] [width=16] [height=16] [reps=200]
(64 256 1920 1080 10 = BASELINE config 5's batch)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd")
K = pkg._capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
side = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g)
pkg.fill_grid(pkg.default_params(), g, t0, t1)
rp = pkg.default_render_params(g)
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16, 16)
REPS = int(sys.argv[5]) if len(sys.argv) > 5 else 200
cams = pkg.orbit_cameras(n, aspect=W / H)
dist_vol = pkg.commit_distance(g, t0)
pairs = pkg.commit_pairs(g, dist_vol)
dev = pkg.upload_cameras(cams)
out = torch.empty((n, H, W, 4), dtype=torch.float32, device="cuda")
def timed(fn, reps=REPS):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); host = (time.perf_counter() - t) / reps * 1e6
    torch.cuda.synchronize()
    return {"host_us_per_call": round(host, 2), "stream_us_per_call": round(e0.elapsed_time(e1) * 1e3 / reps, 2)}
res = {"cameras": n, "image": [W, H], "grid": side, "march_over": "y-pair volume"}
kw = {"dist": dist_vol, "pairs": pairs}
res["host_array_staged"] = timed(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=out, **kw))
with pkg.options({K.OPT_RAYMARCH_CAMERA_STAGING: 0}):
    res["host_array_launches_of_16"] = timed(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=out, **kw))
res["device_array"] = timed(lambda: pkg.raymarch(rp, t0, t1, dev, W, H, out=out, **kw))
res["one_camera"] = timed(lambda: pkg.raymarch(rp, t0, t1, cams[0], W, H, out=out[:1], **kw))
res["host_array_staged_again"] = timed(lambda: pkg.raymarch(rp, t0, t1, cams, W, H, out=out, **kw))
res["device_array_again"] = timed(lambda: pkg.raymarch(rp, t0, t1, dev, W, H, out=out, **kw))
print(json.dumps(res))
