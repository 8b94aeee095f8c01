#!/usr/bin/env python3
"""Whole-workload parity at BASELINE.json's configs[1] and configs[2] (not a unit test: the oracle needs the host's
cores for tens of seconds).  Every voxel of both textures against the oracle's dense fill, bit for bit, in z-chunks;
then every pixel of the frame: the pre-shading march record bit for bit, RGBA within 1e-4.
Usage (GPU box): python tools/full_parity.py [256 512]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sdf-viewer_amd")
import oracle_binding as oracle  # noqa: E402  (checker only)

WORKLOADS = {256: (1920, 1080), 512: (3840, 2160)}


def cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def check(side, threads=None, log=print, eye=(2.5, 3.0, 5.0), **param_overrides):
    """-> (differing texture words, {aux field: differing words}, max |RGBA - oracle|)"""
    threads = threads or cores()
    if True:
        W, H = WORKLOADS[side]
        dims = (side, side, side)
        prm = pkg.default_params(**param_overrides)
        oprm = oracle.params_from(prm)
        g = pkg.make_grid(dims)
        t0, t1 = pkg.alloc_textures(g)
        pkg.fill_grid(prm, g, t0, t1)
        torch.cuda.synchronize()
        h0, h1 = t0.cpu().numpy(), t1.cpu().numpy()
        t = time.time()
        bad = 0
        chunk = 32
        for z in range(0, side, chunk):
            r0, r1 = oracle.fill_dense(oprm, dims, z0=z, z1=z + chunk, threads=threads)
            bad += int((h0[z:z + chunk].view(np.uint32) != r0.view(np.uint32)).sum())
            bad += int((h1[z:z + chunk].view(np.uint32) != r1.view(np.uint32)).sum())
        log(f"{side}^3 fill: {side ** 3} voxels x 8 words compared with the oracle in {time.time() - t:.1f} s "
            f"({threads} threads): {bad} differing words")
        rp = pkg.default_render_params(g)
        cam = pkg.camera_look_at(eye=eye, aspect=W / H)
        dist = pkg.commit_distance(g, t0)
        rgba, aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True, dist=dist)
        torch.cuda.synchronize()
        t = time.time()
        want_rgba, want_aux = oracle.raymarch(oracle.copy_struct(oracle.RenderParams, rp), h0, h1,
                                              oracle.copy_struct(oracle.Camera, cam), W, H, threads=threads)
        got_aux = aux[0].cpu().numpy().view(oracle.AUX_DTYPE).reshape(H, W)
        fields = ["status", "steps", "hit_pos", "t", "raw0", "raw1", "normal", "depth"]
        diff = {f: int((got_aux[f].view(np.uint32) != want_aux[f].view(np.uint32)).sum()) for f in fields}
        err = float(np.abs(rgba[0].cpu().numpy() - want_rgba).max())
        hits = int((want_aux["status"] == 1).sum())
        log(f"{W}x{H} march over {side}^3: {W * H} pixels ({hits} hits, {int(want_aux['steps'].sum())} march steps) "
            f"compared in {time.time() - t:.1f} s: differing words per aux field {diff}; max |RGBA - oracle| = {err:.3g}")
        return bad, diff, err


def check_config4(world=8, side=1024, threads=None, log=print):
    """BASELINE.json config 4's grid (1024^3 as 8 z-slabs) rehearsed on ONE GPU: every rank's slab filled by the slab
    path (z_begin/z_end) into its place, all 68.7 GB compared with the oracle on the device, chunk by chunk.
    -> number of differing 32-slice chunks."""
    threads = threads or cores()
    dims = (side, side, side)
    prm = pkg.default_params()
    oprm = oracle.params_from(prm)
    t0 = torch.empty((side, side, side, 4), dtype=torch.float32, device="cuda")
    t1 = torch.empty_like(t0)
    t = time.time()
    for r in range(world):
        z0, z1 = side * r // world, side * (r + 1) // world
        pkg.fill_grid(prm, pkg.make_grid(dims, z_begin=z0, z_end=z1), t0[z0:z1], t1[z0:z1])
    torch.cuda.synchronize()
    fill_s = time.time() - t
    t = time.time()
    bad, chunk = 0, 32
    for z in range(0, side, chunk):
        r0, r1 = oracle.fill_dense(oprm, dims, z0=z, z1=z + chunk, threads=threads)
        same = torch.equal(t0[z:z + chunk].view(torch.int32), torch.from_numpy(r0).cuda().view(torch.int32)) and \
            torch.equal(t1[z:z + chunk].view(torch.int32), torch.from_numpy(r1).cuda().view(torch.int32))
        bad += 0 if same else 1
    log(f"{side}^3 as {world} z-slabs: {side ** 3} voxels filled in {fill_s * 1e3:.1f} ms ({world} launches), compared with "
        f"the oracle in {time.time() - t:.1f} s ({threads} threads): {bad} of {side // chunk} chunks differ")
    return bad


def random_sweep(n, seed=1, log=print, side=256):
    """n random parameter sets and cameras at configs[1]'s (side 256) or configs[2]'s (512) full size; -> number of runs
    with any difference."""
    rng = np.random.default_rng(seed)
    failures = 0
    for k in range(n):
        kw = dict(cube_half_side=float(rng.integers(20, 101)) / 100.0, sphere_radius=float(np.float32(rng.uniform(0.2, 1.25))),
                  max_distance_custom_material=float(np.float32(rng.uniform(0.0, 0.25))),
                  cube_material=int(rng.integers(0, 2)), sphere_material=int(rng.integers(0, 2)),
                  disable_sphere=int(rng.integers(0, 5) == 0))
        v = rng.normal(size=3)
        eye = tuple(float(x) for x in v / np.linalg.norm(v) * rng.uniform(0.3, 6.0))
        bad, diff, err = check(side, log=lambda m: None, eye=eye, **kw)
        ok = bad == 0 and not any(diff.values()) and err <= 1e-4
        failures += 0 if ok else 1
        log(f"run {k}: {kw} eye {tuple(round(e, 2) for e in eye)} -> texture words {bad}, aux {sum(diff.values())}, rgba {err:.2g}"
            + ("" if ok else "   <-- MISMATCH"))
    return failures


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
        side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
        bad = random_sweep(n, seed=1 if side == 256 else 2, log=lambda m: print(m, flush=True), side=side)
        print(f"sweep: {n} random full-size runs, {bad} with differences", flush=True)
        return
    for side in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
        if side == 1024:
            check_config4(log=lambda m: print(m, flush=True))
        else:
            check(side, log=lambda m: print(m, flush=True))


if __name__ == "__main__":
    main()
