#!/usr/bin/env python3
"""Latency of ONE SDFSurface::sample through the host-buffer convenience the per-point provider ABI uses
(sdfv_sample_points_host with n = 1): what a caller of the reference's ffi.rs ABI pays per call."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sdf-viewer_amd")
prm = pkg.default_params()
for n in (1, 64, 1024, 4096):
    pts = np.random.default_rng(0).uniform(-1, 1, size=(n, 3)).astype(np.float32)
    out = np.zeros((n, 7), np.float32)
    call = lambda: pkg.lib.sdfv_sample_points_host(C.byref(prm), 0, pts.ctypes.data, n, 0, out.ctypes.data)
    for _ in range(50):
        call()
    t = time.perf_counter()
    reps = 2000
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t) / reps
    print(f"n = {n:5d}: {dt * 1e6:8.1f} us per call  ({n / dt / 1e6:7.2f} Mpoints/s)")
