#!/usr/bin/env python3
"""Load = fill + commit.  Separate passes (sdfv_fill_grid, then sdfv_commit_distance re-reading tex0) vs the fused
pass (sdfv_fill_grid_commit: the fill also stores the 4-byte distance volume)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sdf-viewer_amd")


def ms(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for side in (256, 512):
    prm = pkg.default_params()
    g = pkg.make_grid((side, side, side))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((side, side, side), dtype=torch.float32, device="cuda")

    def separate():
        pkg.fill_grid(prm, g, t0, t1)
        pkg.commit_distance(g, t0, dist=dist)

    a = ms(separate)
    b = ms(lambda: pkg.fill_grid(prm, g, t0, t1, dist=dist))
    c = ms(lambda: pkg.fill_grid(prm, g, t0, t1))
    print(f"{side}^3: fill {c:.4f} ms | fill + commit {a:.4f} ms | fused {b:.4f} ms "
          f"({side ** 3 * 36 / b / 1e6:.0f} GB/s of stores)")
