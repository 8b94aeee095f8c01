#!/usr/bin/env python3
"""Whole-workload parity at BASELINE.json's configs[1] and configs[2] (not a unit test: the oracle needs the host's
cores for tens of seconds).  Every voxel of both textures against the oracle's dense fill, bit for bit, in z-chunks;
then every pixel of the frame: the pre-shading march record bit for bit, RGBA within 1e-4.
Both of bench.py's pipelines (plain / fused) with the kernel variants the bench times (default options, no aux record).
Usage (GPU box): python tools/full_parity.py [256 512 1024] | sweep N [side] [plain|fused]"""
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


AUX_FIELDS = ["status", "steps", "hit_pos", "t", "raw0", "raw1", "normal", "depth"]
PIPELINES = ("plain", "fused", "fused_ilv")


def check(side, threads=None, log=print, eye=(2.5, 3.0, 5.0), pipeline="plain", **param_overrides):
    """One BASELINE configuration through ONE of bench.py's two pipelines, under the library's default options -- the
    kernel instantiations the driver times, not stand-ins for them:
      plain: sdfv_fill_grid (plain stores)                    -> the no-aux march over tex0.r
      fused: sdfv_fill_grid_commit (nt texture stores + the compact distance volume in the same launch)
                                                              -> the no-aux march over THAT volume
    Compared: both textures (and, fused, the distance volume) against the oracle's dense fill word for word; the no-aux
    RGBA against the oracle on every pixel and, bit for bit, against the RGBA of the aux kernel over the same buffers;
    the aux kernel's pre-shading record against the oracle word for word.
    -> (differing texture [+ volume] words, {aux field: differing words, "rgba_noaux_vs_aux": differing words},
        max |no-aux RGBA - oracle|)"""
    assert pipeline in PIPELINES, pipeline
    threads = threads or cores()
    W, H = WORKLOADS[side]
    dims = (side, side, side)
    prm = pkg.default_params(**param_overrides)
    oprm = oracle.params_from(prm)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    dist = ilv_vol = None
    if pipeline == "fused":
        dist = torch.full((side, side, side), -3.0, dtype=torch.float32, device=t0.device)
        pkg.fill_grid(prm, g, t0, t1, dist=dist)   # sdfv_fill_grid_commit, default options (auto-nt)
    elif pipeline == "fused_ilv":
        # round 4's third pipeline: the dense fill writes the march's y-interleaved volume itself (a step-1 virgin pass with
        # SDFV_PASS_VOLUME_INTERLEAVED); compared below as the de-interleaved distance volume
        ilv_vol = torch.full((side, side, side), -3.0, dtype=torch.float32, device=t0.device)
        pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=ilv_vol, flags=pkg._capi.PASS_VIRGIN_GRID | pkg._capi.PASS_VOLUME_INTERLEAVED)
    else:
        pkg.fill_grid(prm, g, t0, t1)              # sdfv_fill_grid
    torch.cuda.synchronize()
    h0, h1 = t0.cpu().numpy(), t1.cpu().numpy()
    hd = None if dist is None else dist.cpu().numpy()
    if ilv_vol is not None:  # rows 2p, 2p + 1 as one row of pairs -> [D, H, W]
        v = ilv_vol.view(side, side // 2, side, 2)
        hd = torch.stack([v[..., 0], v[..., 1]], dim=2).reshape(side, side, side).cpu().numpy()
    t = time.time()
    bad = 0
    chunk = 32
    for z in range(0, side, chunk):
        r0, r1 = oracle.fill_dense(oprm, dims, z0=z, z1=z + chunk, threads=threads)
        bad += int((h0[z:z + chunk].view(np.uint32) != r0.view(np.uint32)).sum())
        bad += int((h1[z:z + chunk].view(np.uint32) != r1.view(np.uint32)).sum())
        if hd is not None:
            bad += int((hd[z:z + chunk].view(np.uint32) != r0[..., 0].view(np.uint32)).sum())
    words = 8 + (1 if hd is not None else 0)
    log(f"[{pipeline}] {side}^3 fill: {side ** 3} voxels x {words} words compared with the oracle in {time.time() - t:.1f} s "
        f"({threads} threads): {bad} differing words")
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at(eye=eye, aspect=W / H)
    # the kernel bench.py times: no aux record, default options (hand-written loop, interior fetch, box-first order)
    rgba = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, ilv=ilv_vol)
    rgba_aux, aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True, dist=dist, ilv=ilv_vol)
    pairs_diff = None
    if dist is not None:  # the y-pair volume (what a host that renders many frames per load marches over): same bits
        pairs = pkg.commit_pairs(g, dist)
        rgba_pairs = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, pairs=pairs)
        pairs_diff = int((rgba_pairs.view(torch.int32) != rgba.view(torch.int32)).sum().item())
        del pairs, rgba_pairs
        ilv = pkg.commit_interleaved(g, dist)
        rgba_ilv = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, ilv=ilv)
        ilv_diff = int((rgba_ilv.view(torch.int32) != rgba.view(torch.int32)).sum().item())
        del ilv, rgba_ilv
    torch.cuda.synchronize()
    t = time.time()
    want_rgba, want_aux = oracle.raymarch(oracle.copy_struct(oracle.RenderParams, rp), h0, h1,
                                          oracle.copy_struct(oracle.Camera, cam), W, H, threads=threads)
    got_aux = aux[0].cpu().numpy().view(oracle.AUX_DTYPE).reshape(H, W)
    diff = {f: int((got_aux[f].view(np.uint32) != want_aux[f].view(np.uint32)).sum()) for f in AUX_FIELDS}
    diff["rgba_noaux_vs_aux"] = int((rgba.view(torch.int32) != rgba_aux.view(torch.int32)).sum().item())
    if pairs_diff is not None:
        diff["rgba_pairs_vs_dist"] = pairs_diff
        diff["rgba_ilv_vs_dist"] = ilv_diff
    err = float(np.abs(rgba[0].cpu().numpy() - want_rgba).max())
    hits = int((want_aux["status"] == 1).sum())
    log(f"[{pipeline}] {W}x{H} no-aux march over {'the distance volume' if dist is not None else ('the y-interleaved volume the fill wrote' if ilv_vol is not None else 'tex0.r')} of {side}^3: "
        f"{W * H} pixels ({hits} hits, {int(want_aux['steps'].sum())} march steps) compared in {time.time() - t:.1f} s: "
        f"differing words per aux field / no-aux RGBA vs aux RGBA {diff}; max |RGBA - oracle| = {err:.3g}")
    return bad, diff, err


def check_config4_fused(world=8, side=1024, threads=None, log=print):
    """BASELINE.json config 4 through the step bench.py --gpus N times per rank: each of the 8 ranks' z-slabs of the 1024^3
    grid filled by sdfv_slab_fill_step_commit (textures + distance volume + the RCCL halo exchange) on the library's
    communicator in loopback (periodic world of 1: the ghosts receive the slab's own last / first slice), default
    options.  Owned slices of tex0 / tex1 / the volume against the oracle word for word, ghosts against the wrap.
    -> number of differing 32-slice chunks + differing ghost slices."""
    import importlib
    par = importlib.import_module("sdf-viewer_amd.parallel")
    threads = threads or cores()
    dims = (side, side, side)
    prm = pkg.default_params()
    oprm = oracle.params_from(prm)
    comm = par.SlabComm(pkg, 0, 1, periodic=True)
    bad, chunk = 0, 32
    t = time.time()
    try:
        for r in range(world):
            z0, z1 = par.slab_range(side, r, world)
            slab = par.alloc_slab((side, side, z1 - z0), 0, 1, "cuda", periodic=True)
            slab.z_begin, slab.z_end = z0, z1
            grid = pkg.make_grid(dims, z_begin=z0, z_end=z1)
            dist = torch.full(tuple(slab.tex0.shape[:3]), -3.0, dtype=torch.float32, device="cuda")
            for _ in range(2):
                comm.fill_step(prm, grid, slab, dist=dist)
            torch.cuda.synchronize()
            own_d = dist[slab.ghost_lo:slab.ghost_lo + (z1 - z0)]
            for z in range(z0, z1, chunk):
                r0, r1 = oracle.fill_dense(oprm, dims, z0=z, z1=min(z + chunk, z1), threads=threads)
                a, b = z - z0, min(z + chunk, z1) - z0
                d0 = torch.from_numpy(r0).cuda()
                same = torch.equal(slab.owned0[a:b].view(torch.int32), d0.view(torch.int32)) and \
                    torch.equal(slab.owned1[a:b].view(torch.int32), torch.from_numpy(r1).cuda().view(torch.int32)) and \
                    torch.equal(own_d[a:b].view(torch.int32), d0[..., 0].contiguous().view(torch.int32))
                bad += 0 if same else 1
            for tex in (slab.tex0, slab.tex1):
                bad += 0 if torch.equal(tex[0].view(torch.int32), tex[-2].view(torch.int32)) else 1   # ghost_lo = last owned
                bad += 0 if torch.equal(tex[-1].view(torch.int32), tex[1].view(torch.int32)) else 1   # ghost_hi = first owned
            bad += 0 if torch.equal(dist[0].view(torch.int32), slab.tex0[0, ..., 0].contiguous().view(torch.int32)) else 1
            bad += 0 if torch.equal(dist[-1].view(torch.int32), slab.tex0[-1, ..., 0].contiguous().view(torch.int32)) else 1
            del slab, dist, own_d
    finally:
        comm.close()
    log(f"[fused] {side}^3 as {world} z-slabs through sdfv_slab_fill_step_commit (RCCL loopback): textures, distance volume "
        f"and ghosts compared with the oracle in {time.time() - t:.1f} s ({threads} threads): {bad} chunks / ghost slices differ")
    return bad


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


def random_sweep(n, seed=1, log=print, side=256, pipeline="plain"):
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
        bad, diff, err = check(side, log=lambda m: None, eye=eye, pipeline=pipeline, **kw)
        ok = bad == 0 and not any(diff.values()) and err <= 1e-4
        failures += 0 if ok else 1
        log(f"run {k}: {kw} eye {tuple(round(e, 2) for e in eye)} -> texture words {bad}, aux {sum(diff.values())}, rgba {err:.2g}"
            + ("" if ok else "   <-- MISMATCH"))
    return failures


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sweep":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
        side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
        pipeline = sys.argv[4] if len(sys.argv) > 4 else "fused"
        bad = random_sweep(n, seed=1 if side == 256 else 2, log=lambda m: print(m, flush=True), side=side, pipeline=pipeline)
        print(f"sweep [{pipeline}]: {n} random full-size runs, {bad} with differences", flush=True)
        return
    for side in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
        if side == 1024:
            check_config4(log=lambda m: print(m, flush=True))
            check_config4_fused(log=lambda m: print(m, flush=True))
        else:
            for pipeline in PIPELINES:
                check(side, log=lambda m: print(m, flush=True), pipeline=pipeline)


if __name__ == "__main__":
    main()
