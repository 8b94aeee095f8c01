#!/usr/bin/env python3
"""The dense fill streams two store bursts in lockstep (tex0 and tex1).  How does its rate depend on the distance
between the two textures in the address space?  One big allocation, tex0 at its start, tex1 at size + skew."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdf-viewer_amd")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prm = pkg.default_params()
g = pkg.make_grid((side,) * 3)
n = side ** 3 * 4  # floats per texture
size = n * 4
big = torch.empty(2 * n + (64 << 20) // 4, dtype=torch.float32, device="cuda")
base = big.data_ptr()
pad = (-base) % (2 << 20)  # start on a 2 MiB boundary


def ms(fn, reps=300):
    for _ in range(300):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{side}^3: texture size {size >> 20} MiB; skew = gap between the end of tex0 and the start of tex1")
for skew in (0, 256, 1024, 4096, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20,
             (8 << 20) + 4096, 16 << 20, 17 << 20, 32 << 20, 33 << 20, 48 << 20):
    o0 = pad // 4
    o1 = o0 + n + skew // 4
    t0 = big[o0:o0 + n].view(side, side, side, 4)
    t1 = big[o1:o1 + n].view(side, side, side, 4)
    t = ms(lambda: pkg.fill_grid(prm, g, t0, t1), 300 if side <= 256 else 40)
    print(f"  skew {skew:>10d} B ({skew / (1 << 20):7.3f} MiB): {t:.4f} ms  {2 * size / t / 1e6:6.0f} GB/s")
