#!/usr/bin/env python3
"""tools/place3_sweep.py found the distance volume's position irrelevant and the tex0 <-> tex1 distance decisive within +-53 KiB;
bench.py's probe sometimes finds two SEPARATE allocations faster than any of those.  How does the fused fill depend on the
distance over a wide range?   python tools/place_far_sweep.py [side=512]"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdf-viewer_amd")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
n = side ** 3 * 4; size = n * 4; nd = side ** 3
reps = 200 if side <= 256 else 40
def ms(fn):
    for _ in range(max(5, reps // 4)): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
MAXS = 3 << 30
big = torch.empty((2 * size + MAXS + (8 << 20)) // 4, dtype=torch.float32, device="cuda")
dvol = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
pad = (-big.data_ptr()) % (1 << 30)  # tex0 on a 1 GiB boundary when the block allows
if pad + 2 * size + MAXS > big.numel() * 4: pad = (-big.data_ptr()) % (2 << 20)
skews = [0, 4096, 12288, 65536, 256 << 10, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 10 << 20, 14 << 20, 16 << 20, 18 << 20,
         32 << 20, 34 << 20, 64 << 20, 66 << 20, 96 << 20, 128 << 20, 130 << 20, 192 << 20, 256 << 20, 258 << 20, 384 << 20, 512 << 20,
         514 << 20, 768 << 20, 1 << 30, (1 << 30) + (2 << 20), (1 << 30) + (512 << 20), 2 << 30, (2 << 30) + (2 << 20), 3 << 30]
res = {"side": side, "tex0_mod_1GiB": (big.data_ptr() + pad) % (1 << 30), "fused": {}, "plain": {}}
for rnd in range(2):
    for s in skews:
        o0 = pad // 4; o1 = o0 + n + s // 4
        t0 = big[o0:o0 + n].view(side, side, side, 4); t1 = big[o1:o1 + n].view(side, side, side, 4)
        res["fused"].setdefault(str(s), []).append(round(ms(lambda: pkg.fill_grid(prm, g, t0, t1, dist=dvol)), 4))
        res["plain"].setdefault(str(s), []).append(round(ms(lambda: pkg.fill_grid(prm, g, t0, t1)), 4))
print(json.dumps(res))
