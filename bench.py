#!/usr/bin/env python3
"""bench.py -- Mvoxels/s grid fill + Mrays/s sphere-trace, demo SDF (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: fill the voxel grid from the
demo SDF, then sphere-trace it.  The metric is a PAIR (voxels and rays are different units), so the two
halves are timed in two separately bracketed regions of exactly K steps each:
    value       = Mvoxels/s of the fill  (the half BASELINE.json's target is stated on)
    value_rays  = Mrays/s of the raymarch (all W*H pixels counted, misses included)
    ms_per_step = fill ms + raymarch ms.
Both halves belong to ONE pipeline over the same buffers: at N = 1 two are timed -- `pipeline_plain` (sdfv_fill_grid,
32 B/voxel -> sdfv_raymarch over tex0.r) and `pipeline_fused` (sdfv_fill_grid_commit, 36 B/voxel -> sdfv_raymarch_accel
over the compact distance volume that fill wrote) -- and value / value_rays / ms_per_step / roofline come from the one
that is faster end to end (`pipeline`).  The N = 1 line also carries `target_512` (the 512^3 fill the north-star target
is stated on), `roofline_raymarch` with SURVEY 8(d)'s modelled bytes, and `halo_loopback`; N > 1 lines carry `config4`
(cube geometry at 512^3 voxels per rank: 8 ranks = BASELINE config 4's 1024^3) next to the default slab geometry.
Inputs are deterministic (integer lattice + fixed cameras) and already resident in HBM; outputs stay in HBM.

N = 1 workload: BASELINE.json configs[1] (256^3 grid + 1920x1080).  --workload 512 selects configs[2]
(512^3 + 3840x2160).  N > 1 (weak scaling): the grid grows to side^3 voxels PER RANK, sharded by z-slab
(--weak-geometry slab: along z only, each rank fills the N = 1 slab; cube: towards config 4's 1024^3),
each fill step includes the one-voxel RCCL halo exchange (overlapped with the interior fill); the raymarch renders one camera per rank
(orbit) over a replica of the N = 1 grid.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The multi-GPU fill step uses
# the caller's stream, the communicator's high-priority stream and RCCL's own; with 4 queues two of them can end up on
# one queue and the step runs 2.5x slower (265 us instead of 106 at 256^3, depending on creation order); with 8 every
# order measured is fast.  Read by the HIP runtime at start-up, hence set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the host driver of this pool only supports dmabuf IPC: without this RCCL's peer mapping fails (hipIpcGetMemHandle: invalid
# argument).  Already exported on the GPU boxes; kept here for any environment the driver builds itself.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FILL_BYTES_PER_VOXEL = 32  # tex0 16 B + tex1 16 B, store-only

WORKLOADS = {
    "256": dict(side=256, width=1920, height=1080, name="demo_sdf 256^3 grid + 1920x1080 sphere-trace (configs[1])"),
    "512": dict(side=512, width=3840, height=2160, name="demo_sdf 512^3 grid + 3840x2160 sphere-trace (configs[2])"),
    "64": dict(side=64, width=512, height=512, name="demo_sdf 64^3 grid + 512x512 sphere-trace (configs[0])"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="256")
    ap.add_argument("--weak-geometry", choices=["slab", "cube"], default="slab",
                    help="N>1: grow the grid along z only (every rank fills the N=1 slab; default) or towards a cube "
                         "(8 ranks x --workload 512 = BASELINE.json config 4, 1024^3)")
    ap.add_argument("--halo-transport", choices=["auto", "rccl", "torch"], default="auto",
                    help="N>1: rccl = sdfv_slab_fill_step over the library's own RCCL communicator (default under "
                         "nccl), torch = torch.distributed P2P ops")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed busy period before the warm-up steps of each timed region (device clock ramp)")
    ap.add_argument("--no-tuned-placement", action="store_true",
                    help="allocate tex0 and tex1 separately instead of letting sdfv_tune_texture_placement choose the "
                         "distance between them inside one block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the extra 64-camera batch (profiling runs)")
    ap.add_argument("--batch-split", choices=["both", "all", "tiles", "cameras", "rows"], default="all",
                    help="N>1, 64-camera batch: tiles = every rank renders the 16-row tile bands r, r + N, ... of every camera "
                         "(BASELINE.json config 5's image-tile split, balanced), rows = one contiguous range of rows per rank "
                         "(the same split unbalanced: the outer ranks see background only), cameras = whole cameras dealt to the "
                         "ranks; default: all three are timed, `batch_raymarch` is the tile split and carries the others beside it")
    ap.add_argument("--per-step-samples", type=int, default=50,
                    help="launches timed one by one with HIP events for ms_per_step_median / p95 (outside the K-step regions)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-overlapped", action="store_true",
                    help="N=1: skip the double-buffered (fill of step k+1 beside the march of step k) measurement")
    ap.add_argument("--pipeline", choices=["both", "plain", "fused"], default="both",
                    help="N=1: time both pipelines and report the faster one (default), or only one (profiling runs: "
                         "per-kernel rocprof averages then belong to one kernel variant)")
    ap.add_argument("--no-progressive", action="store_true",
                    help="N=1: skip the progressive / changed_box block (LoadingManager passes, SURVEY 8(f)1)")
    ap.add_argument("--no-target-512", action="store_true", help="N=1: skip the 512^3 fill block (north-star target config)")
    ap.add_argument("--no-config4", action="store_true", help="N>1: skip the cube-geometry block (BASELINE config 4)")
    ap.add_argument("--config4-side", type=int, default=512,
                    help="N>1: voxels per rank of the cube-geometry block are side^3 (8 x 512^3 = config 4's 1024^3)")
    return ap.parse_args()


PREWARM_S = 0.25  # --prewarm-ms


def STAGE(name):
    """Registers the collective stage about to block (parallel.enter_stage) for the watchdog's report."""
    par = sys.modules.get("sdf-viewer_amd.parallel")
    if par is not None:
        par.enter_stage("bench.py " + name)


def prewarm(fn, torch, dist=None, world=1, device=None, seconds=None):
    """Untimed: keep the device busy with `fn` for ~0.25 s so that the timed steps run at the clocks a busy GPU runs at.
    The MI355X idles at a few hundred MHz and needs ~10 ms of load to ramp (tools/clock_ramp.py: the first 8 ms of
    256^3 fills after an idle period are 9 % slower than the steady state, and on some boxes the rate keeps drifting
    for about a second: tools/tex_skew_sweep.py); a few warm-up steps of 0.1 ms each do not get it there.  At N > 1 the steps contain exchanges, so every rank must make the SAME number of calls: the
    count is agreed on (MAX over ranks) before the loop."""
    seconds = PREWARM_S if seconds is None else seconds
    if seconds <= 0:
        return
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    n = min(2000, int(seconds / one) + 1)
    if world > 1:
        STAGE("prewarm: all_reduce(MAX) of the call count")
        t = torch.tensor([n], dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = int(t.item())
    STAGE("prewarm: running")
    for _ in range(n):
        fn()
    torch.cuda.synchronize()


def timed_region(fn, steps, torch, dist, world, device):
    """EXACTLY `steps` calls of fn bracketed by barrier + synchronize on both sides; MAX over ranks.
    Also returns the HIP-event time of the region on the launch stream (kernel time incl. launch gaps)."""
    if world > 1:
        STAGE("timed_region: barrier before")
        dist.barrier()
    torch.cuda.synchronize()
    STAGE("timed_region: K steps + synchronize")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        STAGE("timed_region: barrier after")
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        STAGE("timed_region: all_reduce(MAX) of the times")
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])
    return dt, ev_ms


def placement_note(args, slab):
    """How tex0 and tex1 were placed (the dense fill's two store streams run up to ~10 % faster or slower with it)."""
    if args.no_tuned_placement:
        return "two separate allocations, not probed"
    gap = slab.tex1.data_ptr() - slab.tex0.data_ptr() - slab.tex0.numel() * 4
    if 0 <= gap <= (64 << 10):
        return f"placement probe kept: one block, tex1 {gap} B after tex0's end (sdfv_tune_texture_placement)"
    return "placement probe kept: two separate allocations (faster here than the block candidates)"


def effective_cores():
    """Host cores this process can actually use: the affinity mask, capped by the cgroup CPU quota (a container
    may see every core of the node but be throttled to a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, (q + p // 2) // p))
        except Exception:
            pass
    return n


def cpu_baseline(workload, budget_s):
    """The oracle (a port: the Rust reference cannot be built here) timed on this box's host cores, on a bounded
    sample of the same workload (SURVEY.md 8d / BASELINE.md 2):
      value           1 thread, the reference's own loop shape: SDFViewer::update in LoadingManager order with
                      the default 2 passes over the same grid (scene/sdf/mod.rs:173-215), whole loads repeated
                      until the time budget is used;
      all_cores       same arithmetic, dense pass, OpenMP over z on every host core (generous baseline);
      value_rays      material.frag restatement, 1 thread, 8-row bands of the same image over the same grid."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as oracle
    side, W, H = workload["side"], workload["width"], workload["height"]
    prm = oracle.default_params()
    dims = (side, side, side)
    cores = effective_cores()
    n_vox = side ** 3
    # (i) faithful single-thread loop
    t0, t1 = oracle.grid_init(dims)
    loads, fill_dt, partial = 0, 0.0, 0
    while fill_dt < budget_s * 0.4:
        if loads:
            t0.fill(oracle.AIR_DIST)
            t1.fill(oracle.AIR_DIST)
        lm = oracle.lm_new(dims, 2)
        t = time.perf_counter()
        # cap one load at the remaining budget: iterations of pass 2 over computed voxels are the cheap skips
        done = oracle.viewer_update(prm, dims, lm, t0, t1, max_iterations=2 ** 62)
        fill_dt += time.perf_counter() - t
        loads += 1
        partial += done
        if fill_dt > budget_s:  # one load of a big grid may exceed the budget on its own: stop there
            break
    fill_mvox = loads * n_vox / fill_dt / 1e6
    # (ii) all host cores, dense.  Fresh buffers whose pages are first touched by the OpenMP team itself (one
    # untimed warm-up fill), so that on a multi-socket host every thread writes to its own NUMA node.
    threads = min(cores, side)
    a0 = np.empty((side, side, side, 4), np.float32)
    a1 = np.empty_like(a0)
    dense_args = (oracle.C.byref(prm), 0, oracle.u3(dims), oracle.f3((-1, -1, -1)), oracle.f3((1, 1, 1)), 0, side,
                  a0.ctypes.data, a1.ctypes.data, threads)
    oracle.L.or_fill_dense(*dense_args)
    n_all, all_dt = 0, 0.0
    while all_dt < budget_s * 0.2:
        t = time.perf_counter()
        oracle.L.or_fill_dense(*dense_args)
        all_dt += time.perf_counter() - t
        n_all += 1
    all_mvox = n_all * n_vox / all_dt / 1e6
    del a0, a1
    # raymarch: 8-row bands of the same image over the grid just filled, middle of the image outwards
    rp = oracle.default_render_params(dims)
    cam = oracle.camera_look_at(aspect=W / H)
    band, k, t = 8, 0, time.perf_counter()
    n_bands = H // band
    while time.perf_counter() - t < budget_s * 0.4 and k < n_bands:
        b = n_bands // 2 + ((k + 1) // 2) * (1 if k % 2 else -1)
        oracle.raymarch(rp, t0, t1, cam, W, H, y0=b * band, y1=(b + 1) * band, threads=1, want_aux=False)
        k += 1
    rows_done = k * band
    rays = rows_done * W / (time.perf_counter() - t) / 1e6
    # the same image on every usable core (OpenMP over rows), whole frames until a small budget is used
    frames, all_rays_dt = 0, 0.0
    while all_rays_dt < budget_s * 0.1:
        t = time.perf_counter()
        oracle.raymarch(rp, t0, t1, cam, W, H, threads=threads, want_aux=False)
        all_rays_dt += time.perf_counter() - t
        frames += 1
    all_rays = frames * W * H / all_rays_dt / 1e6
    return {"value": round(fill_mvox, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": f"{loads} complete load(s) of the {side}^3 grid in LoadingManager order, 2 passes "
                      f"({loads * n_vox} voxels, {fill_dt:.1f} s), oracle/grid_fill.c gcc -O3 -ffp-contract=off, 1 thread",
            "all_cores": {"value": round(all_mvox, 3), "unit": "Mvoxels/s", "cores": threads,
                          "sample": f"{n_all} dense fill(s) of the {side}^3 grid, OpenMP over z ({all_dt:.1f} s); "
                                    f"{threads} = usable cores (affinity {len(os.sched_getaffinity(0))}, cgroup quota applied)"},
            "value_rays": round(rays, 3), "unit_rays": "Mrays/s",
            "sample_rays": f"{rows_done} central rows of the {W}x{H} image, 1 thread",
            "all_cores_rays": {"value": round(all_rays, 3), "unit": "Mrays/s", "cores": threads,
                               "sample": f"{frames} whole {W}x{H} frame(s), OpenMP over rows ({all_rays_dt:.1f} s)"}}


def load_traffic(workload_key, name="fill_pmc_traffic.json"):
    """HBM bytes per launch from the committed PMC pass (profiles/*_pmc_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(path))
        return d.get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def raymarch_traffic_report(workload_key, launch_ms, path_key="product_path", traffic_key=None):
    """SURVEY.md 8(d)'s raymarch byte model next to the measured figures.  compulsory / nominal bytes come from the
    oracle's deterministic counts (tools/raymarch_bytes.py -> profiles/raymarch_model_bytes.json, committed); `traffic`
    is the HBM bytes of one launch from the committed PMC pass.  The kernel is bound by dependent-gather latency /
    instruction issue, not by HBM: `frac` says how far from the HBM roofline the COMPULSORY bytes are."""
    rep = {"kernel": "raymarch_kernel", "bound": "latency (<=255 dependent gathers per ray), not hbm",
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": round(launch_ms, 5)}
    traffic = load_traffic(traffic_key or workload_key, "raymarch_pmc_traffic.json") if workload_key else None
    model = None
    try:
        model = json.load(open(os.path.join(ROOT, "profiles", "raymarch_model_bytes.json"))).get(workload_key)
    except Exception:
        pass
    sec = launch_ms * 1e-3
    rep["traffic"] = traffic
    rep["traffic_GBs"] = None if traffic is None else round(traffic / sec / 1e9, 1)
    if model:
        comp, nominal = model["compulsory_bytes"], model["nominal_gather_bytes"]
        pp = model[path_key]
        rep.update({
            "compulsory_bytes": comp, "nominal_gather_bytes": nominal,
            "achieved": round(comp / sec / 1e9, 1), "frac": round(comp / sec / 1e9 / HBM_PEAK_GBS, 4),
            "nominal_gather_GBs": round(nominal / sec / 1e9, 1),
            "traffic_over_compulsory": None if traffic is None else round(traffic / comp, 3),
            "path_run": {"which": path_key, "compulsory_bytes": pp["compulsory_bytes"],
                         "compulsory_line_bytes": pp["compulsory_line_bytes"],
                         "nominal_gather_bytes": pp["nominal_gather_bytes"],
                         "nominal_gather_GBs": round(pp["nominal_gather_bytes"] / sec / 1e9, 1),
                         "traffic_over_compulsory_lines": None if traffic is None else round(traffic / pp["compulsory_line_bytes"], 3)},
            "counts": model["counts"], "unique_texels": model["unique_texels"],
            "note": "SURVEY 8(d): compulsory = 16 B x (unique tex0 + tex1 texels touched) + 16 B x W*H (achieved/frac "
                    "are computed from it); nominal = 128 B x (sum steps + 5 x hits) + 16 B x W*H (cache-level gather "
                    "rate); path_run = the same two figures for the kernel variant timed here; traffic = PMC HBM bytes "
                    "per launch of this configuration (committed pass)"})
    else:
        rep.update({"achieved": rep["traffic_GBs"], "frac": None if traffic is None else round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "no byte model committed for this configuration (tools/raymarch_bytes.py)"})
    return rep


def progressive_block(pkg, torch, prm, sides, reps=7):
    """SURVEY 8(f)1 under the bench's measurement discipline: the LoadingManager passes (loading.rs:50-76) with
    update_required (scene/sdf/mod.rs:184-190) on the device, over textures that travel with their distance volume
    (sdfv_fill_grid_pass_dist).  Per case: median ms over `reps` runs (HIP events; the state is re-created, untimed, before
    every run), visited voxels, updated voxels, and the fraction of the HBM roofline on SURVEY 8(d)'s incremental figure,
    36 B per UPDATED voxel (4 B read + 32 B written) + 4 B per voxel that is visited only."""
    import ctypes as C
    AIR = pkg.AIR_DIST
    out = {}
    for side in sides:
        g = pkg.make_grid((side,) * 3)
        t0, t1 = pkg.alloc_textures(g)
        dist = torch.empty((side,) * 3, dtype=torch.float32, device=t0.device)
        n = side ** 3

        def fresh():
            pkg.grid_init(g, t0, t1)
            dist.fill_(AIR)

        def loaded():
            pkg.fill_grid(prm, g, t0, t1, dist=dist)

        def timed(fn, setup):
            ts = []
            for _ in range(reps):
                setup()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            return sorted(ts)[len(ts) // 2]

        FRESH, SAME = pkg._capi.PASS_FRESH_GRID, pkg._capi.PASS_SAME_LOAD

        def passes(steps, box=None, flagged=False):
            # flagged: what host/sdf_viewer.cpp's LoadingManager tells the library (sdfv_fill_grid_pass_ex): the first pass
            # of a load sees a fresh grid, the later ones revisit what the same load wrote -- nothing is read
            return lambda: [pkg.fill_grid_pass(prm, g, st, t0, t1, changed_box=box, dist=dist,
                                               flags=((FRESH | SAME) if k == 0 else SAME) if flagged else 0)
                            for k, st in enumerate(steps)]

        def visited(steps):
            return sum((-(-side // st)) ** 3 for st in steps)

        def case(ms, vis, upd, what):
            bytes_ = 36 * upd + 4 * (vis - upd)
            return {"ms": round(ms, 4), "visited_voxels": vis, "updated_voxels": upd,
                    "Mvoxels_s_updated": round(upd / ms / 1e3, 1) if upd else 0.0,
                    "algorithmic_bytes": bytes_, "frac": round(bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "what": what}

        whole = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)
        eighth = (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5)
        # voxels of the 1/8 box: coordinates idx/(N-1)*2-1 in [-0.5, 0.5] per axis
        in_box_axis = sum(1 for i in range(side) if -0.5 <= (i / (side - 1)) * 2.0 - 1.0 <= 0.5)
        res = {}
        res["fresh_load_2_passes"] = case(timed(passes((2, 1), flagged=True), fresh), visited((2, 1)), n,
                                          "the reference's DEFAULT load (cli/mod.rs:13-18): step 2 then step 1 over a fresh grid, as "
                                          "SDFViewer::update enqueues it (sdfv_fill_grid_pass_ex with the LoadingManager's knowledge: "
                                          "store-only; the intermediate LOD-2 state is produced)")
        res["fresh_load_2_passes"]["algorithmic_bytes"] = 36 * (n + visited((2,)))  # no reads; the step-2 lattice is written twice
        res["fresh_load_2_passes"]["frac"] = round(36 * (n + visited((2,))) / (res["fresh_load_2_passes"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res["fresh_pass_step_2_flagged"] = case(timed(passes((2,), flagged=True), fresh), visited((2,)), visited((2,)),
                                                "first pass of that load alone (whole visited rows written: 1/4 of the textures)")
        res["fresh_load_2_passes_unflagged"] = case(timed(passes((2, 1)), fresh), visited((2, 1)), n,
                                                    "the same load through sdfv_fill_grid_pass_dist (update_required read from the volume)")
        res["fresh_pass_step_2"] = case(timed(passes((2,)), fresh), visited((2,)), visited((2,)), "first pass, unflagged")

        def after_step2():
            fresh()
            pkg.fill_grid_pass(prm, g, 2, t0, t1, dist=dist)
        res["fresh_pass_step_1_after_step_2"] = case(timed(passes((1,)), after_step2), n, n - visited((2,)), "second pass of that load alone")
        res["fresh_pass_step_1"] = case(timed(passes((1,)), fresh), n, n, "a single step-1 pass over a fresh grid")
        res["edit_full_box_3_passes"] = case(timed(passes((4, 2, 1), whole), loaded), visited((4, 2, 1)), visited((4, 2, 1)),
                                             "parameter edit whose changed_box is the whole bounding box (what the demo reports): "
                                             "steps 4, 2, 1 rewrite everything they visit")
        res["edit_eighth_box_3_passes"] = case(timed(passes((4, 2, 1), eighth), loaded), visited((4, 2, 1)),
                                               sum((-(-in_box_axis // st)) ** 3 for st in (4, 2, 1)),
                                               "changed_box = [-0.5, 0.5]^3 (1/8 of the volume); updated count approximate for step > 1")
        res["noop_pass_step_1"] = case(timed(passes((1,)), loaded), n, 0, "step-1 pass over a loaded grid, no box: reads the volume, writes nothing")
        res["dense_fused_fill_ms"] = round(timed(lambda: pkg.fill_grid(prm, g, t0, t1, dist=dist), lambda: None), 4)
        out[str(side)] = res
        del t0, t1, dist
    out["note"] = ("sdfv_fill_grid_pass_dist; frac = (36 B x updated + 4 B x visited-only voxels) / ms / 8 TB/s; every intermediate "
                   "state is bit-identical to the oracle's LoadingManager loop (tests/test_gpu_fill.py)")
    return out


def batch_valu_roofline(workload_key, world, ms_per_batch):
    """The 64-camera batch fills the machine with short waves and is bound by VALU ISSUE, not by HBM: one wave64 VALU
    instruction occupies its SIMD's issue port for 4 cycles (16 lanes per cycle), so a SIMD retires at most clock / 4 wave
    instructions per second.  frac = (VALU wave-instructions of one batch, PMC SQ_INSTS_VALU, committed pass) /
    (1024 SIMDs x clock / 4 x batch time)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "raymarch_batch_valu.json")))[workload_key]
    except Exception:  # noqa: BLE001
        return None
    simds, clock = 1024 * world, float(d.get("clock_GHz", 2.4))
    peak = simds * clock * 1e9 / 4.0
    rate = d["valu_wave_instructions_per_batch"] / (ms_per_batch * 1e-3)
    return {"bound": "valu issue", "valu_wave_instructions_per_batch": d["valu_wave_instructions_per_batch"],
            "achieved": round(rate / 1e12, 3), "peak": round(peak / 1e12, 3), "unit": "T wave-instructions/s",
            "frac": round(rate / peak, 4), "simds": simds, "clock_GHz": clock, "source": d.get("source"),
            "note": "wave64 VALU instruction = 4 issue cycles on its SIMD; peak = SIMDs x clock / 4"}


def raymarch_rank_cameras_report(workload_key, world, launch_ms):
    """N > 1: every rank marches ONE camera of the `world`-camera orbit (camera r = camera r * 64 / world of the 64-camera
    orbit when world divides 64) over its replica.  SURVEY 8(d)'s compulsory / nominal bytes of exactly those cameras come
    from the per-camera entries of profiles/raymarch_model_bytes.json's batch model; achieved = their sum / the (max over
    ranks) launch time, against `world` x 8 TB/s."""
    rep = {"kernel": "raymarch_kernel", "bound": "latency (<=255 dependent gathers per ray), not hbm",
           "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "avg_launch_ms": round(launch_ms, 5), "traffic": None}
    try:
        model = json.load(open(os.path.join(ROOT, "profiles", "raymarch_model_bytes.json")))
        per_cam = model[f"{workload_key}_batch64"]["per_camera"]
        if 64 % world:
            raise KeyError(f"{world} does not divide the 64-camera orbit the model was made for")
        mine = [per_cam[r * 64 // world] for r in range(world)]
    except Exception as e:  # noqa: BLE001
        rep.update({"achieved": None, "frac": None, "note": f"no per-camera byte model for this configuration: {type(e).__name__}: {e}"})
        return rep
    sec = launch_ms * 1e-3
    comp = sum(c["compulsory_bytes"] for c in mine)
    nominal = sum(c["nominal_gather_bytes"] for c in mine)
    rep.update({"compulsory_bytes": comp, "nominal_gather_bytes": nominal,
                "product_path_compulsory_bytes": sum(c["product_path_compulsory_bytes"] for c in mine),
                "achieved": round(comp / sec / 1e9, 1), "frac": round(comp / sec / 1e9 / (HBM_PEAK_GBS * world), 4),
                "nominal_gather_GBs": round(nominal / sec / 1e9, 1),
                "cameras": [{"orbit_index": r * 64 // world, "hits": c["hits"], "sum_steps": c["sum_steps"]} for r, c in enumerate(mine)],
                "note": "SURVEY 8(d) over the ranks' cameras (one each): compulsory = 16 B x (unique tex0 + tex1 texels of that "
                        "camera's frame) + 16 B x W*H, summed over ranks; peak = n_gpus x 8 TB/s; no PMC pass exists for N > 1"})
    return rep


class NativeStdoutToStderr:
    """RCCL prints a version banner to the C-level stdout when its first communicator comes up (buffered, so it lands
    after anything Python has printed).  The contract is ONE JSON line on stdout: while the benchmark runs, file
    descriptor 1 points at stderr; it is restored (after flushing the C streams) just before the JSON line is printed."""

    def __enter__(self):
        import ctypes
        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def restore(self):
        if self.saved is not None:
            sys.stdout.flush()
            self.libc.fflush(None)
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None

    def __exit__(self, *exc):
        self.restore()
        return False


# The line as far as it is known: set once the two timed regions of the contract are in (value, value_rays, ms_per_step,
# roofline), then completed extra by extra.  Should an EXTRA (self-check, config 4 block, batch, ...) hang on first contact with
# real xGMI, the watchdog prints what is here instead of losing the measurement with it.
PARTIAL = {"line": None, "since": None, "redirect": None}


def main():
    # A collective that never completes (a rank lost, P2P unavailable) must not hang the job for ever.  A watchdog thread
    # fires after SDFV_BENCH_WATCHDOG_S seconds in total (default 900), or SDFV_BENCH_EXTRAS_S (default 420) after the
    # contract's measurement was complete: it names the collective stage this rank is stuck in (parallel.enter_stage: every
    # blocking stage of the multi-GPU path registers itself first), dumps every thread's stack, and -- when the measurement
    # is already in -- rank 0 prints the line with what it has ("watchdog": the stage) and every rank exits 0; otherwise 3.
    import faulthandler
    import threading
    watchdog = float(os.environ.get("SDFV_BENCH_WATCHDOG_S", "900"))
    extras = float(os.environ.get("SDFV_BENCH_EXTRAS_S", "420"))
    done = threading.Event()
    t_start = time.monotonic()

    def bark():
        while not done.wait(1.0):
            now = time.monotonic()
            late = now - t_start > watchdog
            stuck_in_extras = PARTIAL["since"] is not None and now - PARTIAL["since"] > extras
            if not (late or stuck_in_extras):
                continue
            code = 3
            try:
                par = sys.modules.get("sdf-viewer_amd.parallel")
                name, age, count = par.current_stage() if par else ("(parallel not imported)", 0.0, 0)
                rank = os.environ.get("RANK", "0")
                print(f"[bench rank {rank}/{os.environ.get('WORLD_SIZE', '1')}] WATCHDOG after {now - t_start:.0f} s: "
                      f"stuck in stage #{count} '{name}' for {age:.1f} s", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                if PARTIAL["line"] is not None:
                    code = 0
                    if rank == "0":
                        line = dict(PARTIAL["line"])
                        line["watchdog"] = f"an extra did not finish: stuck in stage '{name}' for {age:.0f} s; the line carries what was measured before it"
                        if PARTIAL["redirect"] is not None:
                            PARTIAL["redirect"].restore()
                        print(json.dumps(line), flush=True)
            finally:
                os._exit(code)

    if watchdog > 0:
        threading.Thread(target=bark, daemon=True).start()
    with NativeStdoutToStderr() as redirect:
        PARTIAL["redirect"] = redirect
        run(redirect)
    done.set()


def region(fn, steps, warmup, torch, dist, world, device):
    """prewarm + `warmup` untimed calls + EXACTLY `steps` timed calls of fn -> (wall ms per call, HIP-event ms per call)."""
    prewarm(fn, torch, dist, world, device)
    for _ in range(warmup):
        fn()
    dt, ev_ms = timed_region(fn, steps, torch, dist, world, device)
    return dt / steps * 1e3, ev_ms / steps


def per_step_stats(fn, n, torch, warm=5):
    """"hipEvent around the kernel, >= 20 iterations after warm-up, median".  n calls of fn, each bracketed
    by its own pair of HIP events on the launch stream (n + 1 events, one between consecutive calls) -> the distribution
    the K-step mean hides (box / placement / clock spread).  Outside the timed regions; never `value`."""
    if n <= 0:
        return None
    for _ in range(warm):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
    q = lambda f: ms[min(n - 1, int(f * n))]  # noqa: E731
    return {"samples": n, "median": round(ms[n // 2] if n % 2 else 0.5 * (ms[n // 2 - 1] + ms[n // 2]), 5),
            "p95": round(q(0.95), 5), "min": round(ms[0], 5), "max": round(ms[-1], 5),
            "note": "one HIP-event pair per launch (includes the event's own packet: ~1-2 us more than back-to-back launches)"}


FILL_8D_PEAK_MVOX = HBM_PEAK_GBS * 1e9 / FILL_BYTES_PER_VOXEL / 1e6  # 250 000 Mvoxels/s = 8 TB/s at SURVEY 8(d)'s 32 B/voxel


def fill_roofline(kern_ms, voxels, bytes_per_voxel, traffic):
    """the fill's algorithmic bytes are 32 B/voxel (tex0 + tex1), whatever else the launch stores.  achieved /
    frac are on that figure (frac = Mvoxels/s / 250 000, the scale the north-star target is worded on); the fused launch
    also stores the compact distance volume (36 B/voxel on the bus): achieved_bus / frac_bus say how busy the bus is."""
    gbs = FILL_BYTES_PER_VOXEL * voxels / (kern_ms * 1e-3) / 1e9
    bus = bytes_per_voxel * voxels / (kern_ms * 1e-3) / 1e9
    return {"kernel": "fill_dense_kernel", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_8d": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_voxel": FILL_BYTES_PER_VOXEL,
            "algorithmic_bytes_per_launch": FILL_BYTES_PER_VOXEL * voxels,
            "bus_bytes_per_voxel": bytes_per_voxel, "achieved_bus": round(bus, 1), "frac_bus": round(bus / HBM_PEAK_GBS, 4),
            "frac_note": "frac = frac_8d = 32 B/voxel x voxels / launch time / 8 TB/s (SURVEY 8d; = Mvoxels/s / 250 000); "
                         "frac_bus counts every byte the launch stores (36 B/voxel when it also writes the distance volume)",
            "avg_launch_ms": round(kern_ms, 5),
            "avg_launch_note": "HIP events around K back-to-back launches / K: includes the ~6 us gap between "
                               "launches, which rocprofv3's kernel-only average leaves out (8 % at 256^3, under 1 % at 512^3)"}


def run(redirect):
    global PREWARM_S
    args = parse_args()
    PREWARM_S = args.prewarm_ms / 1e3
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("sdf-viewer_amd")
    par = importlib.import_module("sdf-viewer_amd.parallel")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsdfgrid has no CPU path")
    # SDFV_BENCH_BACKEND=gloo lets the N>1 code path be exercised by several ranks sharing one GPU (test boxes);
    # the real thing is nccl = RCCL, one GPU per rank.
    backend = os.environ.get("SDFV_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # SDFV_BENCH_FORCE_MULTI=1 with WORLD_SIZE 1 (one-GPU test boxes): the N > 1 code path under the REAL backend -- torch's
    # RCCL process group and the library's own RCCL communicator in one process, the rank its own z-neighbour (periodic
    # world of 1), every collective of the self-checks on device tensors.  Reported with "loopback": true; never a result.
    loopback = world == 1 and os.environ.get("SDFV_BENCH_FORCE_MULTI") == "1"
    multi = world > 1 or loopback
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        STAGE(f"init_process_group({backend}, rank {rank} of {world}) + first all_reduce")
        if backend == "nccl":
            # device_id = eager communicator creation, and one collective on an uninitialised buffer to be sure: the
            # process's first RCCL communicator has to exist BEFORE the fill stream launches its first kernel (a
            # stream whose hardware queue predates it runs sdfv_slab_fill_step 2.5x slower).
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            dist.all_reduce(torch.empty(1, device=device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = device if (multi and backend == "nccl") else "cpu"  # where small agreement tensors live

    wl = WORKLOADS[args.workload]
    side, W, H = wl["side"], wl["width"], wl["height"]
    prm = pkg.default_params()
    K, Wm = args.steps, args.warmup
    aux_steps = max(2, min(K, 10))

    def make_filler(gdims, slab, slab_dist=None):
        """N>1: halo exchange overlapped with the fill.  Under nccl the library's own RCCL communicator carries it (one C
        call per step); should any rank fail to create or use one, EVERY rank falls back to torch.distributed P2P."""
        transport = args.halo_transport
        if transport == "auto":
            transport = "rccl" if (multi and backend == "nccl") else "torch"
        filler = None
        if transport == "rccl":
            try:
                STAGE("make_filler: SlabFiller(transport=rccl) -> SlabComm.__init__")
                filler = par.SlabFiller(pkg, prm, gdims, slab, rank, world, transport="rccl", dist=slab_dist, periodic=loopback)
                STAGE("make_filler: first sdfv_slab_fill_step_commit over the library communicator + synchronize")
                filler.step()  # a first step, so that a communicator that cannot exchange shows up here, not mid-run
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001 -- reported, then decided collectively below
                filler = None
                print(f"[bench rank {rank}] library communicator unavailable: {e}", file=sys.stderr, flush=True)
            if multi:
                STAGE("make_filler: all_reduce(MIN) agreeing on the library communicator")
                ok = torch.tensor([1 if filler is not None else 0], device=cdev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if filler is not None and filler.comm is not None:
                        filler.comm.close()
                    filler = None
        if filler is None:
            transport = "torch"
            filler = par.SlabFiller(pkg, prm, gdims, slab, rank, world, transport="torch", dist=slab_dist, periodic=loopback)
        return filler, transport

    # ---------------- the grid: side^3 voxels per rank, z-slab of the weak-scaled global grid ----------------
    gdims = par.weak_scaling_dims(side, world, args.weak_geometry)
    slab = par.alloc_slab(gdims, rank, world, device, pkg=None if args.no_tuned_placement else pkg, periodic=loopback)
    grid = pkg.make_grid(gdims, z_begin=slab.z_begin, z_end=slab.z_end)
    owned0, owned1 = slab.owned0, slab.owned1
    voxels_per_rank = pkg.slab_voxels(grid)
    total_voxels = voxels_per_rank * world  # identical per rank by construction

    out = {}
    if not multi:
        # ---------------- N = 1: two CONSISTENT pipelines over the SAME buffers ----------------
        #   plain: sdfv_fill_grid (32 B/voxel)         -> sdfv_raymarch over tex0.r in place
        #   fused: sdfv_fill_grid_commit (36 B/voxel)  -> sdfv_raymarch_accel over the compact distance volume it wrote
        # Each half is timed in its own region of exactly K steps (voxels and rays are different units); the textures
        # the march reads are the ones the timed fill wrote.  value / value_rays / ms_per_step all come from ONE
        # pipeline: the one that is faster end to end.
        rp = pkg.default_render_params(grid)
        cam0 = pkg.camera_look_at(aspect=W / H)  # the reference default camera (scene/mod.rs:82-95)
        rgba = torch.empty((1, H, W, 4), dtype=torch.float32, device=device)
        dist_vol = torch.empty((side, side, side), dtype=torch.float32, device=device)

        def both(fill_kw, march_kw):
            def step():
                pkg.fill_grid(prm, grid, owned0, owned1, **fill_kw)
                pkg.raymarch(rp, owned0, owned1, cam0, W, H, out=rgba, **march_kw)
            return region(step, K, Wm, torch, dist, 1, device)[0]

        INF = float("inf")
        fill_plain_ms = fill_plain_ev = march_tex0_ms = march_tex0_ev = inter_plain_ms = INF
        fill_fused_ms = fill_fused_ev = march_dist_ms = march_dist_ev = inter_fused_ms = INF
        if args.pipeline in ("both", "plain"):
            fill_plain_ms, fill_plain_ev = region(lambda: pkg.fill_grid(prm, grid, owned0, owned1), K, Wm, torch, dist, 1, device)
            march_tex0_ms, march_tex0_ev = region(lambda: pkg.raymarch(rp, owned0, owned1, cam0, W, H, out=rgba), K, Wm,
                                                  torch, dist, 1, device)
            inter_plain_ms = both({}, {})
        if args.pipeline in ("both", "fused"):
            fill_fused_ms, fill_fused_ev = region(lambda: pkg.fill_grid(prm, grid, owned0, owned1, dist=dist_vol), K, Wm,
                                                  torch, dist, 1, device)
            march_dist_ms, march_dist_ev = region(lambda: pkg.raymarch(rp, owned0, owned1, cam0, W, H, out=rgba, dist=dist_vol),
                                                  K, Wm, torch, dist, 1, device)
            inter_fused_ms = both({"dist": dist_vol}, {"dist": dist_vol})
        # The fused pipeline software-pipelined over TWO sets of buffers and two streams: the march of step k (latency-bound on
        # a few long waves, most of the machine idle) runs beside the fill of step k + 1 (HBM-store-bound) -- what the
        # reference's own frame loop does on one thread in turns (scene/mod.rs:166-200: fill passes and the render of the
        # state so far alternate within a frame).  Every step still fills a whole grid and marches a whole frame over the grid
        # ITS fill wrote; nothing is skipped.  Reported beside the sequential pipelines, never as `value`.
        overlapped = None
        if args.pipeline in ("both", "fused") and not args.no_overlapped:
            b0, b1 = torch.empty_like(owned0), torch.empty_like(owned1)
            bd = torch.empty_like(dist_vol)
            sets = [(owned0, owned1, dist_vol), (b0, b1, bd)]
            s_fill, s_march = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device, priority=-1)
            filled = [torch.cuda.Event(), torch.cuda.Event()]
            marched = [torch.cuda.Event(), torch.cuda.Event()]
            state = {"k": 0}

            def overlapped_step():
                k = state["k"]
                state["k"] = k + 1
                t0b, t1b, db = sets[k & 1]
                if k >= 2:
                    s_fill.wait_event(marched[k & 1])  # the march of step k - 2 has finished reading this set
                pkg.fill_grid(prm, grid, t0b, t1b, dist=db, stream=s_fill)
                filled[k & 1].record(s_fill)
                s_march.wait_event(filled[k & 1])
                pkg.raymarch(rp, t0b, t1b, cam0, W, H, out=rgba, dist=db, stream=s_march)
                marched[k & 1].record(s_march)

            ov_ms = region(overlapped_step, K, Wm, torch, dist, 1, device)[0]
            torch.cuda.synchronize()
            same = bool(torch.equal(b0.view(torch.int32), owned0.view(torch.int32)) and torch.equal(bd.view(torch.int32), dist_vol.view(torch.int32)))
            overlapped = {"ms_per_step": round(ov_ms, 4), "Mvoxels_s": round(voxels_per_rank / ov_ms / 1e3, 1),
                          "Mrays_s": round(W * H / ov_ms / 1e3, 1), "buffer_sets": 2, "both_sets_identical": same,
                          "what": "fused pipeline, double-buffered: march of step k on one stream beside the fill of step k + 1 "
                                  "on another; K full fills + K full frames per K steps; not used for value / ms_per_step"}
            del b0, b1, bd
        # the separate device-side commit (what a caller pays who filled without the volume and wants it afterwards)
        commit_ms = INF
        if args.pipeline == "both":
            commit_ms = region(lambda: pkg.commit_distance(grid, owned0, dist=dist_vol), aux_steps, 1, torch, dist, 1, device)[0]
        pipes = {
            "plain": {"fill": "sdfv_fill_grid (32 B/voxel)", "march": "sdfv_raymarch over tex0.r",
                      "ms_fill": round(fill_plain_ms, 4), "ms_raymarch": round(march_tex0_ms, 4),
                      "ms_per_step": round(fill_plain_ms + march_tex0_ms, 4),
                      "ms_per_step_interleaved": round(inter_plain_ms, 4),
                      "Mvoxels_s": round(voxels_per_rank / fill_plain_ms / 1e3, 1), "Mrays_s": round(W * H / march_tex0_ms / 1e3, 1),
                      "fill_frac_of_hbm_peak": round(32 * voxels_per_rank / (fill_plain_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "fill_frac_8d": round(32 * voxels_per_rank / (fill_plain_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "fused": {"fill": "sdfv_fill_grid_commit (36 B/voxel: textures + compact distance volume)",
                      "march": "sdfv_raymarch_accel over the distance volume",
                      "ms_fill": round(fill_fused_ms, 4), "ms_raymarch": round(march_dist_ms, 4),
                      "ms_per_step": round(fill_fused_ms + march_dist_ms, 4),
                      "ms_per_step_interleaved": round(inter_fused_ms, 4),
                      "Mvoxels_s": round(voxels_per_rank / fill_fused_ms / 1e3, 1), "Mrays_s": round(W * H / march_dist_ms / 1e3, 1),
                      "fill_frac_of_hbm_peak": round(36 * voxels_per_rank / (fill_fused_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "fill_frac_8d": round(32 * voxels_per_rank / (fill_fused_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        chosen = "fused" if fill_fused_ms + march_dist_ms < fill_plain_ms + march_tex0_ms else "plain"
        if chosen == "fused":
            fill_ms, march_ms, kern_ms, march_ev, bpv = fill_fused_ms, march_dist_ms, fill_fused_ev, march_dist_ev, 36
        else:
            fill_ms, march_ms, kern_ms, march_ev, bpv = fill_plain_ms, march_tex0_ms, fill_plain_ev, march_tex0_ev, 32
        fill_mvox = voxels_per_rank / fill_ms / 1e3
        march_mrays = W * H / march_ms / 1e3
        # the distribution behind the two means (SURVEY 8d asks for the median): launches timed one by one
        if chosen == "fused":
            st_fill = per_step_stats(lambda: pkg.fill_grid(prm, grid, owned0, owned1, dist=dist_vol), args.per_step_samples, torch)
            st_march = per_step_stats(lambda: pkg.raymarch(rp, owned0, owned1, cam0, W, H, out=rgba, dist=dist_vol),
                                      args.per_step_samples, torch)
        else:
            st_fill = per_step_stats(lambda: pkg.fill_grid(prm, grid, owned0, owned1), args.per_step_samples, torch)
            st_march = per_step_stats(lambda: pkg.raymarch(rp, owned0, owned1, cam0, W, H, out=rgba), args.per_step_samples, torch)
        if st_fill and st_march:
            out["ms_per_step_median"] = round(st_fill["median"] + st_march["median"], 4)
            out["ms_per_step_p95"] = round(st_fill["p95"] + st_march["p95"], 4)
            out["per_step"] = {"fill": st_fill, "raymarch": st_march,
                               "Mvoxels_s_median": round(voxels_per_rank / st_fill["median"] / 1e3, 1),
                               "Mrays_s_median": round(W * H / st_march["median"] / 1e3, 1),
                               "frac_8d_median": round(voxels_per_rank / st_fill["median"] / 1e3 / FILL_8D_PEAK_MVOX, 4),
                               "note": "ms_per_step_median / _p95 = fill + raymarch of `pipeline`, each launch between its own "
                                       "pair of HIP events; value / ms_per_step stay the K-step means of the contract"}
        out["pipeline"] = chosen
        out["pipeline_plain"] = pipes["plain"] if args.pipeline in ("both", "plain") else None
        out["pipeline_fused"] = pipes["fused"] if args.pipeline in ("both", "fused") else None
        out["pipeline_note"] = ("two consistent pipelines over the same buffers; value, value_rays, ms_per_step and "
                                "roofline all come from `pipeline` (the faster one end to end); *_interleaved = K steps "
                                "of fill immediately followed by its march in one timed region")
        out["pipeline_overlapped"] = overlapped
        out["commit_ms"] = round(commit_ms, 4) if commit_ms != INF else None
        out["commit_note"] = ("sdfv_commit_distance as a pass of its own (device-side SDFViewer::commit for a grid filled "
                              "without the volume); not part of either pipeline")
        traffic = load_traffic(args.workload + ("_fused" if bpv == 36 else ""))
        out["roofline"] = fill_roofline(kern_ms, voxels_per_rank, bpv, traffic)
        out["roofline_raymarch"] = raymarch_traffic_report(args.workload, march_ev,
                                                           "product_path" if chosen == "fused" else "tex0_path",
                                                           args.workload + ("" if chosen == "fused" else "_tex0"))
        transport, filler = None, None
        my_cams, r0, r1 = [cam0], owned0, owned1
    else:
        # ---------------- N > 1: z-slab fill step with the halo exchange; raymarch over a replica ----------------
        # The FUSED pipeline per rank (what the N = 1 line reports): every rank's step is sdfv_slab_fill_step_commit --
        # its slab's textures AND compact distance volume in one pass (36 B/voxel) + the halo exchange -- so that `value`
        # at N ranks is N times the same work as `pipeline_fused` at N = 1; the march reads the volume its fill wrote.
        slab_dist = torch.empty(tuple(slab.tex0.shape[:3]), dtype=torch.float32, device=device)
        own_dist = slab_dist[slab.ghost_lo:slab.ghost_lo + (slab.z_end - slab.z_begin)]
        filler, transport = make_filler(gdims, slab, slab_dist)
        fill_ms, _ = region(filler.step, K, Wm, torch, dist, world, device)
        # dominant kernel alone (a step = boundary work + exchange + fill), HIP events on the launch stream
        _, kern_ms = region(lambda: pkg.fill_grid(prm, grid, owned0, owned1, dist=own_dist), K, 1, torch, dist, world, device)
        fill_mvox = total_voxels / fill_ms / 1e3
        out["roofline"] = fill_roofline(kern_ms, voxels_per_rank, 36, load_traffic(args.workload + "_fused"))
        out["fill_step_fraction_of_plain_fill"] = round(kern_ms / fill_ms, 3)
        # raymarch: one camera per rank over a replica of the N = 1 grid, filled by the same fused fill
        rgrid = pkg.make_grid((side, side, side))
        r0, r1 = pkg.alloc_textures(rgrid, device=device)
        dist_vol = torch.empty((side, side, side), dtype=torch.float32, device=device)
        pkg.fill_grid(prm, rgrid, r0, r1, dist=dist_vol)
        rp = pkg.default_render_params(rgrid)
        cams = pkg.orbit_cameras(world, aspect=W / H)  # camera 0 = the reference default (scene/mod.rs:82-95)
        my_cams = [cams[i] for i in par.split_cameras(world, rank, world)]
        rgba = torch.empty((len(my_cams), H, W, 4), dtype=torch.float32, device=device)
        march_ms, march_ev = region(lambda: pkg.raymarch(rp, r0, r1, my_cams, W, H, out=rgba, dist=dist_vol), K, Wm,
                                    torch, dist, world, device)
        march_mrays = W * H * world / march_ms / 1e3
        # the distribution behind the two means, like the N = 1 line: every rank times its launches one by one (the step
        # contains the exchange, so all ranks make the same calls), MAX over ranks of median and p95
        STAGE("per-step statistics of the fill step and the march")
        st_fill = per_step_stats(filler.step, min(args.per_step_samples, 20), torch, warm=2)
        st_march = per_step_stats(lambda: pkg.raymarch(rp, r0, r1, my_cams, W, H, out=rgba, dist=dist_vol), min(args.per_step_samples, 20), torch, warm=2)
        if st_fill and st_march:
            t = torch.tensor([st_fill["median"], st_fill["p95"], st_march["median"], st_march["p95"]], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fm, fp, mm, mp = (float(v) for v in t.tolist())
            out["ms_per_step_median"] = round(fm + mm, 4)
            out["ms_per_step_p95"] = round(fp + mp, 4)
            out["per_step"] = {"fill": {"median": round(fm, 5), "p95": round(fp, 5), "samples": st_fill["samples"]},
                               "raymarch": {"median": round(mm, 5), "p95": round(mp, 5), "samples": st_march["samples"]},
                               "note": "MAX over ranks of each rank's median / p95; one HIP-event pair per step"}
        out["pipeline"] = "fused"
        out["pipeline_note"] = ("N > 1: value = z-slab fill step = the fused fill per rank (textures + distance volume, "
                                "36 B/voxel) incl. the RCCL halo exchange: N x the work of pipeline_fused at N = 1; value_rays "
                                "= one camera per rank over a replica, marched over the distance volume that fill wrote")
        out["roofline_raymarch"] = raymarch_rank_cameras_report(args.workload, world, march_ev)
        # what RCCL itself says about the library communicator the step ran on (None under the torch transport)
        comm = getattr(filler, "comm", None)
        out["rccl_ranks"] = None if comm is None else comm.rccl_ranks[1]
        out["rccl_rank_of_rank0"] = None if comm is None else comm.rccl_ranks[0]
        out["torch_world_size"] = dist.get_world_size()

    # ---------------- the contract's measurement is complete: the line exists from here on ----------------
    line = {
        "metric": "Mvoxels/s grid fill + Mrays/s sphere-trace @1080p, demo SDF",
        "metric_note": "the metric is a pair: value = Mvoxels/s of the grid fill, value_rays = Mrays/s of the sphere-trace",
        "value": round(fill_mvox, 1),
        "unit": "Mvoxels/s",
        "value_rays": round(march_mrays, 1),
        "unit_rays": "Mrays/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "prewarm_ms": args.prewarm_ms,
        "texture_placement": placement_note(args, slab),
        "ms_per_step": round(fill_ms + march_ms, 4),
        "ms_per_step_fill": round(fill_ms, 4),
        "ms_per_step_raymarch": round(march_ms, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (demo SDF defaults on the integer lattice, fixed cameras; no RNG)",
        "sharded_fill_verified": None,
        "sharded_march": None,
        "loopback": True if loopback else None,
        "backend": None if not multi else ("rccl" if backend == "nccl" else backend + " (test only)"),
        "halo_transport": None if not multi else {"rccl": "sdfv_slab_fill_step (library RCCL communicator)",
                                                   "torch": "torch.distributed batch_isend_irecv"}[transport],
        "config": {"workload": wl["name"], "grid_global": list(gdims), "voxels_per_gpu": voxels_per_rank,
                   "image": [W, H], "cameras_per_gpu": len(my_cams),
                   "weak_geometry": None if not multi else args.weak_geometry,
                   "parallelism": "single GPU" if world == 1 else f"z-slab x{world} + 1-voxel RCCL halo; 1 camera/GPU"},
        "keys_note": "N = 1 and N > 1 lines carry the same contract keys; N > 1 adds rccl_ranks, torch_world_size, config4",
    }
    line.update(out)
    line["raymarch_kernel_ms"] = round(march_ev, 4)
    for key in ("batch_raymarch", "target_512", "progressive", "halo_loopback", "config4"):
        line[key] = None
    PARTIAL["line"], PARTIAL["since"] = line, time.monotonic()

    # ---------------- N = 1 extras ----------------
    target_512 = None
    halo_loopback = None
    if not multi and not args.no_batch:
        # the north-star target configuration (>= 70 % of the HBM roofline on the 512^3 fill), whatever --workload is
        if not args.no_target_512:
            try:
                if side == 512:
                    t_slab, t_grid = slab, grid
                else:
                    t_slab = par.alloc_slab((512, 512, 512), 0, 1, device, pkg=None if args.no_tuned_placement else pkg)
                    t_grid = pkg.make_grid((512, 512, 512))
                t_dist = torch.empty((512, 512, 512), dtype=torch.float32, device=device)
                ts = max(5, min(K, 20))
                t_ms, t_ev = region(lambda: pkg.fill_grid(prm, t_grid, t_slab.owned0, t_slab.owned1), ts, 2, torch, dist, 1, device)
                f_ms, f_ev = region(lambda: pkg.fill_grid(prm, t_grid, t_slab.owned0, t_slab.owned1, dist=t_dist), ts, 2,
                                    torch, dist, 1, device)
                n512 = 512 ** 3
                target_512 = {"grid": [512, 512, 512], "steps": ts, "ms_fill": round(t_ms, 4),
                              "Mvoxels_s": round(n512 / t_ms / 1e3, 1), "avg_launch_ms": round(t_ev, 5),
                              "achieved_GBs": round(32 * n512 / (t_ev * 1e-3) / 1e9, 1),
                              "frac": round(32 * n512 / (t_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "frac_8d": round(32 * n512 / (t_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "per_launch": per_step_stats(lambda: pkg.fill_grid(prm, t_grid, t_slab.owned0, t_slab.owned1),
                                                           min(args.per_step_samples, 30), torch, warm=2),
                              "fused_commit": {"ms_fill": round(f_ms, 4), "Mvoxels_s": round(n512 / f_ms / 1e3, 1),
                                               "achieved_GBs": round(36 * n512 / (f_ev * 1e-3) / 1e9, 1),
                                               "frac": round(36 * n512 / (f_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "frac_8d": round(32 * n512 / (f_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "frac_note": "frac: 36 B/voxel on the bus; frac_8d: SURVEY 8(d)'s 32 B/voxel"},
                              "texture_placement": placement_note(args, t_slab), "target_frac": 0.70,
                              "note": "north_star: >= 70 % HBM-roofline Mvoxels/s on the demo SDF 512^3 grid fill at 1 GPU; "
                                      "32 B/voxel algorithmic, HIP events over the timed launches"}
                del t_dist
                if side != 512:
                    del t_slab
            except Exception as e:  # noqa: BLE001 -- an extra, never fatal
                target_512 = {"error": f"{type(e).__name__}: {e}"}
        # the multi-GPU fill step in loopback, measured by tools/slab_step_probe.py in a process of its own (so that
        # this process never brings up an RCCL communicator at N = 1)
        try:
            import subprocess
            torch.cuda.synchronize()
            loop = {}
            for s_side in sorted({side, 512}):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slab_step_probe.py"), str(s_side), str(K)],
                                   capture_output=True, text=True, timeout=300)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                loop[str(s_side)] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else \
                    {"error": (r.stderr or r.stdout)[-300:]}
            halo_loopback = loop[str(side)]
            halo_loopback["by_side"] = {k: {kk: v.get(kk) for kk in ("ms_per_step", "plain_fill_ms", "fraction_of_plain_fill_rate",
                                                                     "ghosts_verified", "deferred_join", "error") if kk in v}
                                        for k, v in loop.items()}
        except Exception as e:  # noqa: BLE001 -- an extra, never fatal
            halo_loopback = {"error": f"{type(e).__name__}: {e}"}

    line["target_512"], line["halo_loopback"] = target_512, halo_loopback  # (the watchdog prints the line as it stands)
    progressive = None
    if not multi and not args.no_batch and not args.no_progressive:
        try:
            progressive = progressive_block(pkg, torch, prm, sorted({side, 512}))
        except Exception as e:  # noqa: BLE001 -- an extra, never fatal
            progressive = {"error": f"{type(e).__name__}: {e}"}

    line["progressive"] = progressive
    # ---------------- config 5 shape: a batch of 64 cameras, split over the ranks (extra, not `value`) ----------------
    n_batch = 64
    batch_report = None
    if not args.no_batch:
        if not multi:
            pkg.fill_grid(prm, grid, owned0, owned1, dist=dist_vol)  # the volume of the grid being marched
        rgrid_whole = pkg.make_grid((side, side, side))  # the grid the batch marches (N > 1: every rank's replica)
        batch_cams = pkg.orbit_cameras(n_batch, aspect=W / H)
        batch_steps = max(2, min(K, 5))

        # A host that renders MANY frames per load marches over the y-pair volume (sdfv_commit_pairs: 8 B/voxel, built once
        # from the distance volume; two 16-byte gathers per cell instead of four 8-byte ones, bit-identical).  The batch is
        # exactly that case: 64 frames over one grid.  Its one-off cost is reported and folded into value_incl_commit.
        # Beyond the last-level cache (512^3) the library advises the y-interleaved volume instead (4 B/voxel, rows paired).
        volume_kind = pkg.march_volume_advice(rgrid_whole)
        if volume_kind is None:  # (a grid that is not cubic: the distance volume is what marches fastest)
            accel_kw, commit_pairs_ms = {}, 0.0
        elif volume_kind == "interleaved":
            accel_vol = pkg.commit_interleaved(rgrid_whole, dist_vol)
            commit_pairs_ms = region(lambda: pkg.commit_interleaved(rgrid_whole, dist_vol, ilv=accel_vol), 3, 1, torch, dist, world, device)[0]
            accel_kw = {"ilv": accel_vol}
        else:
            accel_vol = pkg.commit_pairs(rgrid_whole, dist_vol)
            commit_pairs_ms = region(lambda: pkg.commit_pairs(rgrid_whole, dist_vol, pairs=accel_vol), 3, 1, torch, dist, world, device)[0]
            accel_kw = {"pairs": accel_vol}

        def time_split(split, use_pairs=True):
            """tiles = BASELINE config 5 as named (image-tile split), balanced: the 16-row tile bands r, r + N, ... of EVERY
            camera per rank (sdfv_raymarch_bands); rows = one contiguous range of rows of every camera per rank; cameras =
            whole cameras dealt to the ranks.  At N = 1 all three are the same call."""
            where = {}
            if split == "tiles" and world > 1:
                mine = batch_cams
                where = {"bands": par.split_bands(H, rank, world)}
                n_rows = len(par.band_rows(H, *where["bands"]))
            elif split == "cameras":
                mine = [batch_cams[i] for i in par.split_cameras(n_batch, rank, world)]
                n_rows = H
            else:
                mine = batch_cams
                by0, by1 = par.split_rows(H, rank, world)
                where, n_rows = {"y0": by0, "y1": by1}, by1 - by0
            batch_out = torch.empty((len(mine), n_rows, W, 4), dtype=torch.float32, device=device)

            def batch_step():
                pkg.raymarch(rp, r0, r1, mine, W, H, out=batch_out, dist=dist_vol, **where, **(accel_kw if use_pairs else {}))

            batch_step()
            batch_dt, _ = timed_region(batch_step, batch_steps, torch, dist, world, device)
            ms = batch_dt / batch_steps * 1e3
            return {"split": split if world > 1 else None, "cameras_per_gpu": len(mine), "rows_per_gpu": n_rows,
                    "value": round(n_batch * W * H / ms / 1e3, 1), "unit": "Mrays/s", "ms_per_batch": round(ms, 4),
                    "march_over": (f"y-{volume_kind} volume" if volume_kind == "interleaved" else "y-pair volume") if use_pairs
                                  else "distance volume"}

        splits = ["tiles"] if world == 1 else (["tiles", "rows", "cameras"] if args.batch_split in ("all", "both") else [args.batch_split])
        reports = {sp: time_split(sp) for sp in splits}
        batch_report = {"cameras": n_batch, "image": [W, H]}
        batch_report.update(reports[splits[0]])
        if "cameras" in splits[1:]:
            batch_report["camera_split"] = reports["cameras"]
        if "rows" in splits[1:]:
            batch_report["contiguous_rows_split"] = reports["rows"]
        over_dist = time_split(splits[0], use_pairs=False)
        batch_report["over_distance_volume"] = {k: over_dist[k] for k in ("value", "ms_per_batch")}
        batch_report["commit_pairs_ms"] = round(commit_pairs_ms, 4)
        batch_report["value_incl_commit"] = round(n_batch * W * H / (batch_report["ms_per_batch"] + commit_pairs_ms) / 1e3, 1)
        batch_report["note"] = ("BASELINE.json configs[4] shape (64-camera orbit) over the same grid, distance-volume march; "
                                "top level = the image-tile split config 5 names, balanced (tile bands r, r + N, ... per rank; "
                                "contiguous_rows_split = one range of rows per rank, tools/split_balance.py), camera_split = whole cameras per rank; the march "
                                "reads the volume sdfv_march_volume_advice names (march_over), built once per load (commit_pairs_ms; "
                                "value_incl_commit folds it in), "
                                "over_distance_volume = the same batch over the 4 B/voxel volume")
        # the viewer's steady state: ONE camera, frame after frame over the loaded grid (the reference repaints per event) --
        # the same frame as `value_rays`, over the advised volume instead of the distance volume the fill wrote
        if world == 1:
            frame_out = torch.empty((1, H, W, 4), dtype=torch.float32, device=device)
            cam0 = pkg.camera_look_at(aspect=W / H)
            f_acc = region(lambda: pkg.raymarch(rp, r0, r1, cam0, W, H, out=frame_out, dist=dist_vol, **accel_kw), 20, 3, torch, dist, world, device)[0]
            f_dist = region(lambda: pkg.raymarch(rp, r0, r1, cam0, W, H, out=frame_out, dist=dist_vol), 20, 3, torch, dist, world, device)[0]
            batch_report["steady_state_frame"] = {"march_over": batch_report["march_over"], "ms_per_frame": round(f_acc, 4),
                                                  "value": round(W * H / f_acc / 1e3, 1), "unit": "Mrays/s",
                                                  "over_distance_volume_ms": round(f_dist, 4)}
        # the batch is issue-bound, not HBM-bound: VALU wave-instructions per batch / (SIMDs x clock / 4 cycles per wave64
        # VALU instruction) -- the roofline that actually bounds it (counts: profiles/raymarch_batch_valu.json)
        batch_report["roofline_raymarch_batch"] = batch_valu_roofline(args.workload, world, batch_report["ms_per_batch"])

    line["batch_raymarch"] = batch_report
    # ---------------- N > 1 extras: BASELINE config 4's geometry, and the self-checks ----------------
    config4 = None
    verified = None
    sharded_march = None
    if multi:
        if not args.no_config4:
            # cube geometry at --config4-side^3 voxels per rank: 8 ranks x 512^3 = config 4's 1024^3 (1024 x 1024 slices,
            # 33.5 MB per halo message pair and direction against a 128-slice slab)
            try:
                cside = args.config4_side
                cdims = par.weak_scaling_dims(cside, world, "cube")
                cslab = par.alloc_slab(cdims, rank, world, device, pkg=None if args.no_tuned_placement else pkg, periodic=loopback)
                cgrid = pkg.make_grid(cdims, z_begin=cslab.z_begin, z_end=cslab.z_end)
                cdist = torch.empty(tuple(cslab.tex0.shape[:3]), dtype=torch.float32, device=device)
                c_own = cdist[cslab.ghost_lo:cslab.ghost_lo + (cslab.z_end - cslab.z_begin)]
                cfiller = par.SlabFiller(pkg, prm, cdims, cslab, rank, world, transport=transport,
                                         comm=filler.comm, dist=cdist, periodic=loopback)  # the same communicator serves this slab too
                cs = max(3, min(K, 10))
                c_ms, _ = region(cfiller.step, cs, 2, torch, dist, world, device)
                _, c_kern = region(lambda: pkg.fill_grid(prm, cgrid, cslab.owned0, cslab.owned1, dist=c_own), cs, 1, torch,
                                   dist, world, device)
                cvox = pkg.slab_voxels(cgrid)
                config4 = {"grid_global": list(cdims), "voxels_per_gpu": cvox, "steps": cs,
                           "value": round(cvox * world / c_ms / 1e3, 1), "unit": "Mvoxels/s",
                           "ms_per_step_fill": round(c_ms, 4), "plain_fill_ms": round(c_kern, 4),
                           "fill_step_fraction_of_plain_fill": round(c_kern / c_ms, 3),
                           "frac_of_hbm_peak_per_gpu": round(36 * cvox / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "bytes_per_voxel": 36, "halo_bytes_per_direction": int(cdims[0]) * int(cdims[1]) * 32,
                           "note": "BASELINE.json configs[3] geometry (cube; 8 x 512^3 = 1024^3), weak scaling like `value`"}
                del cslab, cdist
            except Exception as e:  # noqa: BLE001 -- an extra, never fatal
                config4 = {"error": f"{type(e).__name__}: {e}"}
        line["config4"] = config4
        # outside the timed regions: the gathered slabs must equal a dense local fill of the global grid, and the
        # ghost slices must equal what the neighbour computed (= a local recompute: the SDF is analytic)
        if gdims[0] * gdims[1] * gdims[2] * 32 <= 8 << 30:
            try:
                filler.step()
                torch.cuda.synchronize()
                full0, full1 = par.gather_replica(slab, gdims, world)
                chk0, chk1 = pkg.alloc_textures(pkg.make_grid(gdims), device=device)
                pkg.fill_grid(prm, pkg.make_grid(gdims), chk0, chk1)
                torch.cuda.synchronize()
                ok = torch.equal(full0, chk0) and torch.equal(full1, chk1)
                if loopback:  # periodic world of 1: the ghosts hold the grid's last and first slice
                    want0, want1 = torch.cat([chk0[-1:], chk0, chk0[:1]]), torch.cat([chk1[-1:], chk1, chk1[:1]])
                else:
                    lo, hi = slab.z_begin - slab.ghost_lo, slab.z_end + slab.ghost_hi
                    want0, want1 = chk0[lo:hi], chk1[lo:hi]
                ok = ok and torch.equal(slab.tex0, want0) and torch.equal(slab.tex1, want1)
                ok = ok and torch.equal(slab_dist, want0[..., 0])  # the distance volume the step wrote, ghosts included
                del want0, want1
                flag = torch.tensor([1.0 if ok else 0.0], device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                verified = bool(flag.item() == 1.0)
                del full0, full1
                # the consumer of the halo: raymarch the grid where it lies (sharded, rays handed between ranks)
                # and compare with a march over the whole grid, bit for bit
                try:
                    if loopback:
                        raise RuntimeError("skipped: the loopback slab is periodic, the sharded march is not")
                    ggrid = pkg.make_grid(gdims)
                    grp = pkg.default_render_params(ggrid)
                    sw, sh = 320, 180
                    scam = pkg.camera_look_at(eye=(1.5, 2.0, 3.5), aspect=sw / sh)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    # under RCCL the whole march runs inside the library (sdfv_slab_march: `world` rounds + the ray exchange
                    # enqueued in one call, no host round trip); over gloo (tests) torch.distributed carries the rays
                    lib_comm = getattr(filler, "comm", None) if transport == "rccl" else None
                    STAGE("sharded march self-check (" + ("sdfv_slab_march over the library communicator" if lib_comm else "torch.distributed rounds") + ")")
                    got = par.raymarch_sharded(pkg, grp, grid, slab, scam, sw, sh, rank, world, comm=lib_comm)
                    torch.cuda.synchronize()
                    sharded_march_ms = (time.perf_counter() - t0) * 1e3
                    want = pkg.raymarch(grp, chk0, chk1, scam, sw, sh)[0]
                    same = torch.equal(got.view(torch.int32), want.view(torch.int32)) and bool((want[..., 3] > 0).any())
                    flag = torch.tensor([1.0 if same else 0.0], device=cdev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    sharded_march = {"verified": bool(flag.item() == 1.0), "image": [sw, sh], "rounds": world,
                                     "ms": round(sharded_march_ms, 3),
                                     "transport": "sdfv_slab_march (library RCCL communicator, no host round trip per round)" if lib_comm
                                                  else "torch.distributed (counter read-back + count exchange per round)",
                                     "note": "sdfv_raymarch_slab over the sharded grid vs sdfv_raymarch over the whole "
                                             "grid, bit for bit; not part of the timed regions"}
                except Exception as e:  # noqa: BLE001
                    sharded_march = {"verified": f"error: {type(e).__name__}: {e}"}
                del chk0, chk1
            except Exception as e:  # never lose the measurement over the self-check
                verified = f"error: {type(e).__name__}: {e}"
        else:
            verified = "skipped (global grid > 8 GiB)"

    line.update({"sharded_fill_verified": verified, "sharded_march": sharded_march, "batch_raymarch": batch_report,
                 "target_512": target_512, "progressive": progressive, "halo_loopback": halo_loopback, "config4": config4})
    PARTIAL["since"] = None  # the extras are done (the CPU baseline below only uses this process's host cores)
    if rank == 0:
        if not args.no_cpu_baseline and not multi:  # rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_baseline_seconds)
        redirect.restore()
        print(json.dumps(line), flush=True)
    if multi:
        if getattr(filler, "comm", None) is not None:
            torch.cuda.synchronize()
            filler.comm.close()  # the library's RCCL communicator, while every peer is still alive
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
