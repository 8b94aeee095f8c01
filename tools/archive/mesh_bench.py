#!/usr/bin/env python3
"""Mesher front end on one MI355X: sdfv_mesh_extract + sdfv_mesh_postproc wall time per lattice size, and the
streaming rates of the two batched sources (unit-cube points in, distances / normals out)."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sdf-viewer_amd")


def wall(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def main():
    prm = pkg.default_params()
    print("sdfv_mesh_extract (demo SDF, marching cubes) + sdfv_mesh_postproc")
    for n in (64, 128, 256, 512, 1024):
        dt, (v, i) = wall(lambda: pkg.mesh_extract(prm, n), 3)
        dtp, _ = wall(lambda: pkg.mesh_postproc(prm, v), 5)
        cells = n ** 3
        print(f"  -v {n:5d}: {v.shape[0]:9d} vertices {i.shape[0] // 3:9d} triangles  extract {dt * 1e3:8.2f} ms "
              f"({cells / dt / 1e9:6.1f} Gcells/s)  postproc {dtp * 1e3:7.3f} ms ({v.shape[0] * 84 / dtp / 1e9:6.0f} GB/s)")
    n = 1 << 26
    pts = torch.rand((n, 3), device="cuda")
    dt, _ = wall(lambda: pkg.source_sample_scalar(prm, pts), 10)
    print(f"ScalarSource  {n} points: {dt * 1e3:.3f} ms  {n * 16 / dt / 1e9:.0f} GB/s (12 B in + 4 B out per point)")
    dt, _ = wall(lambda: pkg.source_sample_normal(prm, pts), 10)
    print(f"HermiteSource {n} points: {dt * 1e3:.3f} ms  {n * 24 / dt / 1e9:.0f} GB/s (12 B in + 12 B out per point)")


if __name__ == "__main__":
    main()
