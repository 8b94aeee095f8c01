#!/usr/bin/env python3
"""Tuning sweep for the dense fill kernel (run on the GPU box): rows per thread x store policy x
grid size, interleaved rounds in ONE process, median/min of HIP-event times.  Also times a plain
torch fill_ (memset-class, store-only) of the same bytes as the known-good store-bandwidth reference."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdf-viewer_amd")

def time_fn(fn, iters=10):
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)

def main():
    sides = [int(s) for s in (sys.argv[1:] or ["256", "512"])]
    prm = pkg.default_params()
    res = []
    for side in sides:
        g = pkg.make_grid((side,) * 3)
        t0, t1 = pkg.alloc_textures(g)
        nbytes = side ** 3 * 32
        variants = [(1, nt) for nt in (1, 0)]
        def memset():
            t0.fill_(1.0); t1.fill_(1.0)
        best = {}
        for rnd in range(3):
            for b, nt in variants:
                pkg.set_option(pkg._capi.OPT_FILL_NONTEMPORAL, 1 if nt else 2)
                ts = time_fn(lambda: pkg.fill_grid(prm, g, t0, t1), 8)
                best.setdefault((b, nt), []).extend(ts)
            best.setdefault("memset", []).extend(time_fn(memset, 8))
        for k, ts in best.items():
            ts = sorted(ts); med = ts[len(ts) // 2]
            res.append(dict(side=side, variant=str(k), med_ms=round(med, 4), min_ms=round(ts[0], 4),
                            med_GBs=round(nbytes / med / 1e6, 1), best_GBs=round(nbytes / ts[0] / 1e6, 1)))
            print(res[-1], flush=True)
    json.dump(res, open("gpurun_out/fill_sweep.json", "w"), indent=1)

if __name__ == "__main__":
    main()
