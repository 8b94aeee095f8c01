#!/usr/bin/env python3
"""Does the fill rate depend on how long the GPU has been busy?  2000 back-to-back 256^3 fills after an idle period,
time per launch in windows of 100 (HIP events), to separate clock ramp / power capping from kernel behaviour."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdf-viewer_amd")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prm = pkg.default_params()
g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g)
pkg.fill_grid(prm, g, t0, t1)
torch.cuda.synchronize()
time.sleep(3.0)  # let the device go idle
n_win, per = 20, 100 if side <= 256 else 20
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
evs[0].record()
for w in range(n_win):
    for _ in range(per):
        pkg.fill_grid(prm, g, t0, t1)
    evs[w + 1].record()
torch.cuda.synchronize()
ms = [evs[w].elapsed_time(evs[w + 1]) / per for w in range(n_win)]
print(f"{side}^3, ms per launch in consecutive windows of {per} launches after 3 s idle:")
print(" ".join(f"{m:.4f}" for m in ms))
