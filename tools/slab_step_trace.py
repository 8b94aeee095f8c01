#!/usr/bin/env python3
"""Runs a few sdfv_slab_fill_step calls (periodic world of 1 = loopback) for a kernel-trace timeline:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/steptrace -- python tools/slab_step_trace.py 256"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sdf-viewer_amd")
par = importlib.import_module("sdf-viewer_amd.parallel")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
comm = par.SlabComm(pkg, 0, 1, periodic=True)
slab = par.alloc_slab((side, side, side), 0, 1, "cuda", periodic=True)
g = pkg.make_grid((side, side, side))
prm = pkg.default_params()
for _ in range(steps):
    comm.fill_step(prm, g, slab)
torch.cuda.synchronize()
comm.close()
