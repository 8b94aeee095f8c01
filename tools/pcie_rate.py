#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point sdfv_fill_grid_host (hipMalloc + kernel + two pageable D2H
copies), for DESIGN.md.  Never reported as bench.py's `value`."""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("sdf-viewer_amd")
for side in (64, 256):
    prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
    t0 = np.zeros((side, side, side, 4), np.float32); t1 = np.zeros_like(t0)
    pkg.check(pkg.lib.sdfv_fill_grid_host(C.byref(prm), 0, C.byref(g), t0.ctypes.data, t1.ctypes.data))
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        pkg.check(pkg.lib.sdfv_fill_grid_host(C.byref(prm), 0, C.byref(g), t0.ctypes.data, t1.ctypes.data))
        ts.append(time.perf_counter() - t)
    dt = min(ts)
    print(f"sdfv_fill_grid_host {side}^3: {dt * 1e3:.2f} ms, {side ** 3 / dt / 1e6:.1f} Mvoxels/s PCIe-inclusive, "
          f"{side ** 3 * 32 / dt / 1e9:.2f} GB/s over the link (pageable host memory)")
