#!/usr/bin/env python3
"""where do the ~27 us between bench.py's `noop_pass_step_1` (HIP events around ONE pass enqueued right
behind the fill that re-creates the state) and the same kernel's duration under rocprofv3 go?  Times the step-1 no-op pass over
a loaded grid (a pure 4 B/voxel read of the distance volume) four ways:
  behind_fill   fill; event; pass; event             (the bench's case: the fill's 4.8 GB of stores are still draining)
  behind_idle   fill; sync; 2 ms idle; event; pass; event   (the queue is empty: host launch latency is inside the events)
  chain_8       fill; sync; pass x 9; events around the last 8  -> per pass (what rocprofv3's 100 warm repetitions see)
  behind_small  fill; a 1-block kernel; event; pass; event   (drain still under way, no fill directly before)
usage: python tools/noop_gap.py [side=512]"""
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdf-viewer_amd")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures_placed(g)
dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
prm = pkg.default_params()
tiny = torch.zeros(64, device="cuda")


def fill():
    pkg.fill_grid(prm, g, t0, t1, dist=dist)


def nop():
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist)


def ev():
    return torch.cuda.Event(enable_timing=True)


def med(xs):
    return sorted(xs)[len(xs) // 2]


out = {"side": side, "bytes_read": side ** 3 * 4, "build_id": pkg.lib.sdfv_build_id().decode()}
for _ in range(5):
    fill(); nop()
torch.cuda.synchronize()
ts = []
for _ in range(15):
    fill(); a, b = ev(), ev(); a.record(); nop(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
out["behind_fill_ms"] = round(med(ts), 4)
ts = []
for _ in range(15):
    fill(); torch.cuda.synchronize(); time.sleep(0.002); a, b = ev(), ev(); a.record(); nop(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
out["behind_idle_ms"] = round(med(ts), 4)
ts = []
for _ in range(15):
    fill(); torch.cuda.synchronize(); nop(); a, b = ev(), ev(); a.record()
    for _ in range(8):
        nop()
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 8)
out["chain_8_ms_per_pass"] = round(med(ts), 4)
ts = []
for _ in range(15):
    fill(); tiny.add_(1.0); a, b = ev(), ev(); a.record(); nop(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
out["behind_small_ms"] = round(med(ts), 4)
# the drain alone: fill; event; NOTHING; event
ts = []
for _ in range(15):
    fill(); a, b = ev(), ev(); a.record(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
out["event_pair_behind_fill_ms"] = round(med(ts), 4)
for k in ("behind_fill_ms", "behind_idle_ms", "chain_8_ms_per_pass", "behind_small_ms"):
    out[k.replace("_ms", "").replace("_per_pass", "") + "_frac_of_8TBs"] = round(side ** 3 * 4 / (out[k] * 1e-3) / 8e12, 3)
print(json.dumps(out, indent=1))
