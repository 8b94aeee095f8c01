#!/usr/bin/env python3
"""The FUSED fill streams three store bursts (tex0, tex1, dist).  One block: tex0 at its start, tex1 at size + s1, dist at
2 * size + s1 + sd.  Which (s1, sd) does the fused fill like, how stable is it across fresh allocations of the block, and does
the plain-fill probe (sdfv_tune_texture_placement) predict it?   python tools/place3_sweep.py [side=512] [blocks=3]"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sdf-viewer_amd")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
S1 = (0, 4096, 12288, 28672, 53248)
SD = (0, 4096, 12288, 20480, 36864, 65536, 1 << 20)
out = {"side": side, "reps": reps, "blocks": []}
keep = []
for b in range(blocks):
    big = torch.empty(2 * n + nd + (8 << 20) // 4, dtype=torch.float32, device="cuda")
    keep.append(torch.empty(37 << 20, dtype=torch.uint8, device="cuda"))  # shift where the next block lands
    pad = (-big.data_ptr()) % (2 << 20)
    res = {"base_mod_1GiB": (big.data_ptr() + pad) % (1 << 30), "fused": {}, "plain": {}}
    for rnd in range(2):
        for s1 in S1:
            o0 = pad // 4; o1 = o0 + n + s1 // 4
            t0 = big[o0:o0 + n].view(side, side, side, 4); t1 = big[o1:o1 + n].view(side, side, side, 4)
            res["plain"].setdefault(str(s1), []).append(round(ms(lambda: pkg.fill_grid(prm, g, t0, t1)), 4))
            for sd in SD:
                od = o1 + n + sd // 4
                d = big[od:od + nd].view(side, side, side)
                res["fused"].setdefault(f"{s1},{sd}", []).append(round(ms(lambda: pkg.fill_grid(prm, g, t0, t1, dist=d)), 4))
    sep = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
    res["fused_dist_separate_alloc_s1_0"] = round(ms(lambda: pkg.fill_grid(prm, g, big[pad // 4:pad // 4 + n].view(side, side, side, 4), big[pad // 4 + n:pad // 4 + 2 * n].view(side, side, side, 4), dist=sep)), 4)
    best = min(res["fused"], key=lambda k: sum(res["fused"][k])); worst = max(res["fused"], key=lambda k: sum(res["fused"][k]))
    res["best"] = [best, res["fused"][best]]; res["worst"] = [worst, res["fused"][worst]]
    out["blocks"].append(res)
    del big, sep
print(json.dumps(out))
