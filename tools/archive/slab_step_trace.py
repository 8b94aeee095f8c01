#!/usr/bin/env python3
"""Runs a few sdfv_slab_fill_step calls (periodic world of 1 = loopback) for a kernel-trace timeline:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/steptrace -o t -- python tools/slab_step_trace.py 256 30 side_boundary
   python tools/step_timeline.py gpurun_out/steptrace
forms: auto | side_boundary | side_boundary_event | side_boundary_unpacked"""
import importlib
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sdf-viewer_amd")
par = importlib.import_module("sdf-viewer_amd.parallel")
K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
form = sys.argv[3] if len(sys.argv) > 3 else "auto"
FORMS = {"auto": 0, "side_boundary": K.STEP_SIDE_BOUNDARY, "side_boundary_event": K.STEP_SIDE_BOUNDARY | K.STEP_START_EVENT,
         "side_boundary_unpacked": K.STEP_SIDE_BOUNDARY | K.STEP_UNPACKED}
comm = par.SlabComm(pkg, 0, 1, periodic=True)
slab = par.alloc_slab((side, side, side), 0, 1, "cuda", periodic=True)
g = pkg.make_grid((side, side, side))
prm = pkg.default_params()
pkg.set_option(K.OPT_SLAB_STEP_FORM, FORMS[form])
for _ in range(steps):
    comm.fill_step(prm, g, slab)
torch.cuda.synchronize()
comm.close()
