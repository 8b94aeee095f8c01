#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc WRITE_SIZE` (or FETCH_SIZE): interleaves the dense fill kernel with a plain
store-only kernel of KNOWN byte count (torch fill_ over the same two textures), so WRITE_SIZE can be
calibrated in this access pattern as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("sdf-viewer_amd")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g)
both = torch.empty((2,) + tuple(t0.shape), dtype=torch.float32, device="cuda")  # one fill_ launch = both textures' bytes
for _ in range(10):
    pkg.fill_grid(prm, g, t0, t1)
    both.fill_(1.0)
torch.cuda.synchronize()
print("known bytes per launch:", side ** 3 * 32)
