#!/usr/bin/env python3
"""The progressive / changed_box passes as a profiling workload (rocprofv3 around it: tools/gpu_profile_pass.sh): every case of
bench.py's `progressive` block REPS times at one grid size, the state re-created before every run.  python tools/pass_workload.py [side]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = 10
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
def fresh():
    pkg.grid_init(g, t0, t1); dist.fill_(pkg.AIR_DIST)
def loaded():
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
whole, eighth = (-1, -1, -1, 1, 1, 1), (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5)
for _ in range(REPS):
    fresh()   # the reference's default load, flagged as SDFViewer::update flags it: rows kernel (fresh), then the dense kernel
    pkg.fill_grid_pass(prm, g, 2, t0, t1, dist=dist, flags=K.PASS_FRESH_GRID | K.PASS_SAME_LOAD)
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist, flags=K.PASS_SAME_LOAD)
    fresh()   # the same load unflagged: fill_pass_kernel (step 2), fill_pass_quad_kernel (step 1)
    pkg.fill_grid_pass(prm, g, 2, t0, t1, dist=dist)
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist)
    fresh()   # a single step-1 pass over a fresh grid: fill_pass_quad_kernel with every voxel to update
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist)
    loaded()  # edit whose box is the whole bounding box: rows kernel (copy-through) x 2, dense kernel
    for st in (4, 2, 1):
        pkg.fill_grid_pass(prm, g, st, t0, t1, changed_box=whole, dist=dist)
    for st in (4, 2, 1):  # edit of 1/8 of the volume: fill_pass_kernel x 2, fill_pass_quad_kernel
        pkg.fill_grid_pass(prm, g, st, t0, t1, changed_box=eighth, dist=dist)
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist)  # no-op pass
torch.cuda.synchronize()
