#!/usr/bin/env python3
"""The dense fused fill that writes the y-interleaved volume, two forms in one process: a thread per x of BOTH rows of a pair
(default where W is a multiple of 256) against the row-chunk form with the pair meeting in LDS (SDFV_OPT_FILL_FORM = 1), next
to the fused fill with the plain volume.  Alternating rounds, bits compared.  python tools/ilv_fill_ab.py [sides...]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd")
K = pkg._capi
def run(fn, n, warm=0.2):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
res = {}
prm = pkg.default_params()
for side in [int(s) for s in sys.argv[1:]] or [256, 512]:
    g = pkg.make_grid((side,) * 3)
    t0, t1 = pkg.alloc_textures_placed(g)
    vol = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
    flags = K.PASS_VIRGIN_GRID | K.PASS_VOLUME_INTERLEAVED
    n = 40 if side <= 256 else 12
    def pair():  # R4.10: a thread per x of both rows of a pair
        with pkg.options({K.OPT_FILL_FORM: 4}):
            pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=flags)
    plain = lambda: pkg.fill_grid(prm, g, t0, t1, dist=vol)
    def chunk():
        with pkg.options({K.OPT_FILL_FORM: 1}):
            pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=flags)
    def paired():  # R5.3: one row per workgroup, the rows of a pair on one XCD (volume halves merge in its L2)
        with pkg.options({K.OPT_FILL_FORM: 3}):
            pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=flags)
    just_plain = lambda: pkg.fill_grid(prm, g, t0, t1)
    ms = {"pair_rows": [], "row_chunk_lds": [], "xcd_paired": [], "plain_volume": [], "no_volume": []}
    for rnd in range(4):
        ms["pair_rows"].append(run(pair, n))
        ms["row_chunk_lds"].append(run(chunk, n))
        ms["xcd_paired"].append(run(paired, n))
        ms["plain_volume"].append(run(plain, n))
        ms["no_volume"].append(run(just_plain, n))
    pair(); torch.cuda.synchronize(); a = (t0.clone(), t1.clone(), vol.clone())
    same = True
    for other in (chunk, paired):
        t0.fill_(-7.0); vol.fill_(-7.0)
        other(); torch.cuda.synchronize()
        same = same and all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(a, (t0, t1, vol)))
    res[str(side)] = {**{k: round(min(v), 4) for k, v in ms.items()}, "rounds": {k: [round(x, 4) for x in v] for k, v in ms.items()}, "same_bits": same}
    del t0, t1, vol
print(json.dumps(res))
