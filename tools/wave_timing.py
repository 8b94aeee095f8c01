#!/usr/bin/env python3
"""Tuning: per-wave start/end cycle stamps of the raymarch kernel (SDFV_RAYMARCH_WAVE_TIMING)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("sdf-viewer_amd")
side, W, H = 256, 1920, 1080
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); pkg.fill_grid(prm, g, t0, t1)
rp = pkg.default_render_params(g); cam = pkg.camera_look_at(aspect=W / H)
n_waves = ((W + 15) // 16) * ((H + 15) // 16) * 4
buf = torch.zeros((n_waves, 4), dtype=torch.int64, device="cuda")
dist = pkg.commit_distance(g, t0) if "--dist" in sys.argv else None
for _ in range(3): pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist)
os.environ["SDFV_RAYMARCH_WAVE_TIMING"] = hex(buf.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist); b.record(); torch.cuda.synchronize()
d = buf.cpu().numpy()
start, end, it = d[:, 0], d[:, 1], d[:, 2]
t00 = start.min(); dur = end - start
print("kernel ms", a.elapsed_time(b), "span cycles", end.max() - t00, "waves", n_waves)
print("iterations: max", it.max(), "mean(active)", it[it > 0].mean(), "active waves", (it > 0).sum())
print("wave duration cycles: idle-waves median", np.median(dur[it == 0]), "active median", np.median(dur[it > 0]), "max", dur.max())
act = it > 0
cpi = dur[act] / it[act]
print("cycles per iteration (active waves): median", np.median(cpi), "p10", np.percentile(cpi, 10), "p90", np.percentile(cpi, 90))
order = np.argsort(-dur)[:10]
for i in order: print("wave", i, "iters", it[i], "dur", dur[i], "cyc/iter", dur[i] / max(it[i], 1), "start", start[i] - t00, "end", end[i] - t00)
print("last start", (start - t00).max(), "pct of waves started by 25/50/75% of span", [np.mean((start - t00) < f * (end.max() - t00)) for f in (.25, .5, .75)])
