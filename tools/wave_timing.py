#!/usr/bin/env python3
"""Tuning: per-wave start/end cycle stamps of the raymarch kernel (option SDFV_OPT_TUNING_WAVE_TIMING of the TUNING
build, `make -C sdf-viewer_amd/csrc tuning`; the product library has no such code).  Prints a summary and, with
--json PATH, writes the histogram of wave durations / iterations of the default 1080p frame over the 256^3 grid.
Usage: python tools/wave_timing.py [--dist] [--json profiles/r02/wave_timing_1080p.json]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SDFGRID_LIBRARY"] = os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid_tuning.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

pkg = importlib.import_module("sdf-viewer_amd")
K = pkg._capi
side, W, H = 256, 1920, 1080
if "--side" in sys.argv:
    side = int(sys.argv[sys.argv.index("--side") + 1])
if "--4k" in sys.argv:
    W, H = 3840, 2160
prm = pkg.default_params()
g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g)
pkg.fill_grid(prm, g, t0, t1)
rp = pkg.default_render_params(g)
cam = pkg.camera_look_at(aspect=W / H)
n_waves = ((W + 15) // 16) * ((H + 15) // 16) * 4
buf = torch.zeros((n_waves, 4), dtype=torch.int64, device="cuda")
dist = pkg.commit_distance(g, t0) if ("--dist" in sys.argv or "--pairs" in sys.argv) else None
pairs = pkg.commit_pairs(g, dist) if "--pairs" in sys.argv else None  # the y-pair volume (round 3)
_march = pkg.raymarch
pkg.raymarch = lambda *a_, **k_: _march(*a_, pairs=pairs, **k_)
for _ in range(200):  # clock ramp
    pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist)
if "--launch-order" in sys.argv:  # default: the product's order (the stamps are indexed by tile, not by workgroup)
    pkg.set_option(K.OPT_RAYMARCH_TILE_GROUP, 1)
pkg.set_option(K.OPT_TUNING_WAVE_TIMING, buf.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist)
b.record()
torch.cuda.synchronize()
pkg.set_option(K.OPT_TUNING_WAVE_TIMING, 0)
d = buf.cpu().numpy()
start, end, it = d[:, 0], d[:, 1], d[:, 2] & 0xffff
fetch_it = (d[:, 2] >> 16) & 0xffff  # hand-written loop, tuning build: iterations in which some lane entered a new cell
setup_cyc, loop_cyc = d[:, 3] & 0xffffffff, (d[:, 3] >> 32) & 0xffffffff  # active waves only (culled ones store 0)
dur = end - start
# the cycle counters are per XCD: where a wave lies inside the launch comes from the device-wide 100 MHz counter
# (16 bits of it at the wave's start and end, 10 ns ticks), relative to the earliest start
rt0, rt1 = (d[:, 2] >> 32) & 0xffff, (d[:, 2] >> 48) & 0xffff
rel = lambda t: ((t - rt0[0] + 0x8000) & 0xffff) - 0x8000  # signed distance to wave 0's start
first = rel(rt0).min()
start_us, end_us = (rel(rt0) - first) * 0.01, (rel(rt1) - first) * 0.01
t00 = start.min()
act = it > 0
cpi = dur[act] / it[act]
order = np.argsort(-dur)[:10]
summary = {
    "kernel_ms": a.elapsed_time(b), "waves": int(n_waves),
    "active_waves": int(act.sum()), "iterations_max": int(it.max()), "iterations_mean_active": float(it[act].mean()),
    "wave_cycles_idle_median": float(np.median(dur[~act])), "wave_cycles_active_median": float(np.median(dur[act])),
    "wave_cycles_max": int(dur.max()),
    "cycles_per_iteration_active": {"p10": float(np.percentile(cpi, 10)), "median": float(np.median(cpi)),
                                    "p90": float(np.percentile(cpi, 90))},
    "phases_of_waves_with_64_or_more_iterations": {
        "setup_cycles_median": float(np.median(setup_cyc[act & (it >= 64)])),
        "loop_cycles_median": float(np.median(loop_cyc[act & (it >= 64)])),
        "after_loop_cycles_median": float(np.median((dur - setup_cyc - loop_cyc)[act & (it >= 64)]))},
    "phases_of_all_active_waves": {
        "setup_cycles_median": float(np.median(setup_cyc[act])), "loop_cycles_median": float(np.median(loop_cyc[act])),
        "after_loop_cycles_median": float(np.median((dur - setup_cyc - loop_cyc)[act]))},
    "fetch_block_iterations_fraction_active": float(fetch_it[act].sum() / max(it[act].sum(), 1)),
    "longest_waves": [{"wave": int(i), "iterations": int(it[i]), "fetch_block_iterations": int(fetch_it[i]), "cycles": int(dur[i]),
                       "cycles_per_iteration": float(dur[i] / max(it[i], 1)), "setup": int(setup_cyc[i]), "loop": int(loop_cyc[i]),
                       "after_loop": int(dur[i] - setup_cyc[i] - loop_cyc[i]), "start_us": float(start_us[i]),
                       "end_us": float(end_us[i])} for i in order],
    "hist_iterations": {"edges": [0, 1, 8, 16, 32, 64, 128, 192, 255, 256],
                        "counts": np.histogram(it, bins=[0, 1, 8, 16, 32, 64, 128, 192, 255, 256])[0].tolist()},
    "hist_wave_kcycles": {"edges": [0, 2, 5, 10, 20, 50, 100, 150, 200, 300, 1000],
                          "counts": np.histogram(dur / 1e3, bins=[0, 2, 5, 10, 20, 50, 100, 150, 200, 300, 1000])[0].tolist()},
    "launch_us": {"span_first_start_to_last_end": float(end_us.max()), "last_wave_start": float(start_us.max()),
                  "last_active_wave_start": float(start_us[act].max()),
                  "active_wave_start_percentiles_10_50_90": [float(v) for v in np.percentile(start_us[act], [10, 50, 90])]},
    "config": {"grid": side, "image": [W, H], "dist_volume": dist is not None,
               "tile_order": "launch order" if "--launch-order" in sys.argv else "default (box-first, 2 x 2 tile groups)"},
}
# least-squares model of an active wave: cycles = K + C * iterations + F * (iterations that ran the fetch block)
if fetch_it[act].sum() > 0:
    A = np.stack([np.ones(int(act.sum())), it[act].astype(float), fetch_it[act].astype(float)], axis=1)
    coef, *_ = np.linalg.lstsq(A, dur[act].astype(float), rcond=None)
    long = act & (it >= 64)
    A2 = np.stack([np.ones(int(long.sum())), it[long].astype(float), fetch_it[long].astype(float)], axis=1)
    coef2, *_ = np.linalg.lstsq(A2, dur[long].astype(float), rcond=None)
    summary["wave_cycle_model"] = {"form": "cycles = K + C * iterations + F * fetch_block_iterations",
                                   "all_active_waves": {"K": float(coef[0]), "C": float(coef[1]), "F": float(coef[2]), "n": int(act.sum())},
                                   "waves_with_64_or_more_iterations": {"K": float(coef2[0]), "C": float(coef2[1]), "F": float(coef2[2]), "n": int(long.sum())}}
if "--raw" in sys.argv:  # start, end (XCD-relative), iterations, fetch iterations, set-up and loop cycles per wave
    np.save(sys.argv[sys.argv.index("--raw") + 1], np.stack([start_us, end_us, dur, it, fetch_it, setup_cyc, loop_cyc], axis=1))
print(json.dumps(summary, indent=1))
if "--json" in sys.argv:
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
        json.dump(summary, f, indent=1)
