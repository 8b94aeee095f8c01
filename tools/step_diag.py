#!/usr/bin/env python3
"""Tuning-build diagnostics of the default fill step (loopback): what each stream dependency and the concurrent exchange
cost.  python tools/step_diag.py <side> ; env NCCL_MAX_P2P_NCHANNELS etc. are RCCL's own."""
import importlib, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SDFGRID_LIBRARY"] = os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid_tuning.so")
import torch
pkg = importlib.import_module("sdf-viewer_amd"); par = importlib.import_module("sdf-viewer_amd.parallel"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
comm = par.SlabComm(pkg, 0, 1, periodic=True)
prm = pkg.default_params(); dims = (side,) * 3; grid = pkg.make_grid(dims)
slab = par.alloc_slab(dims, 0, 1, "cuda", periodic=True, pkg=pkg)
def run(fn, n=40, warm=0.15):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)
res = {}
for rnd in range(2):
    # bits: 1 no start dependency, 2 no final wait, 4 no boundary launch, 8 no RCCL group, 16 no ghost copy
    for name, diag in (("step", 0), ("no_start_event", 1), ("no_final_wait", 2), ("neither", 3), ("no_boundary_launch", 4),
                       ("no_rccl", 8), ("no_ghost_copy", 16), ("no_rccl_no_copy", 24), ("only_the_waits", 28),
                       ("nothing_but_the_fill", 31)):
        pkg.set_option(K.OPT_TUNING_WAVE_TIMING, diag)
        res.setdefault(name, []).append(run(lambda: comm.fill_step(prm, grid, slab)))
    pkg.set_option(K.OPT_TUNING_WAVE_TIMING, 0)
    res.setdefault("plain", []).append(run(lambda: pkg.fill_grid(prm, grid, slab.owned0, slab.owned1)))
    res.setdefault("exchange_alone", []).append(run(lambda: comm.halo_exchange(grid, slab)))
print(json.dumps({"side": side, "env": {k: v for k, v in os.environ.items() if k.startswith("NCCL_")}, "ms": res}))
comm.close()
