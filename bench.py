#!/usr/bin/env python3
"""bench.py -- Mvoxels/s grid fill + Mrays/s sphere-trace, demo SDF (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: fill the voxel grid from the
demo SDF, then sphere-trace it.  The metric is a PAIR (voxels and rays are different units), so the two
halves are timed in two separately bracketed regions of exactly K steps each:
    value       = Mvoxels/s of the fill  (the half BASELINE.json's target is stated on)
    value_rays  = Mrays/s of the raymarch (all W*H pixels counted, misses included)
    ms_per_step = fill ms + raymarch ms.
Both halves belong to ONE pipeline over the same buffers: at N = 1 two are timed -- `pipeline_plain` (sdfv_fill_grid,
32 B/voxel -> sdfv_raymarch_ex over tex0.r) and `pipeline_fused` (sdfv_fill_grid_commit, 36 B/voxel -> sdfv_raymarch_ex with
desc.dist = the compact distance volume that fill wrote) -- and value / value_rays / ms_per_step / roofline come from the one
that is faster end to end (`pipeline`).  The N = 1 line also carries `target_512` (the 512^3 fill the north-star target
is stated on), `roofline_raymarch` with SURVEY 8(d)'s modelled bytes, and `halo_loopback`; N > 1 lines carry `config4`
(cube geometry at 512^3 voxels per rank: 8 ranks = BASELINE config 4's 1024^3) next to the default slab geometry.
Inputs are deterministic (integer lattice + fixed cameras) and already resident in HBM; outputs stay in HBM.

N = 1 workload: BASELINE.json configs[1] (256^3 grid + 1920x1080).  --workload 512 selects configs[2]
(512^3 + 3840x2160).  N > 1 (weak scaling): the grid grows to side^3 voxels PER RANK, sharded by z-slab
(--weak-geometry slab: along z only, each rank fills the N = 1 slab; cube: towards config 4's 1024^3),
each fill step includes the one-voxel RCCL halo exchange (overlapped with the interior fill); the raymarch renders one camera per rank
(orbit) over a replica of the N = 1 grid.

This file is the CONTRACT: arguments, the two timed regions, the JSON line, the CPU baseline, the watchdog.  Everything the
line carries beside that (target_512, progressive, host_load, batch_raymarch, halo_loopback, config4, self-checks) is
bench_extras.py; the timing helpers both share are bench_common.py.

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

from types import SimpleNamespace

import bench_common
from bench_common import (FILL_8D_PEAK_MVOX, FILL_BYTES_PER_VOXEL, HBM_PEAK_GBS, STAGE, fill_roofline, load_traffic,  # noqa: F401
                          running_build_id, traffic_stale,
                          per_step_stats, placement_note, region)

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
    ap.add_argument("--separate-textures", "--no-tuned-placement", dest="separate_textures", action="store_true",
                    help="allocate tex0 and tex1 separately instead of in one block at SDFViewer::new_voxels' fixed distance "
                         "(pkg.alloc_textures_placed)")
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
    ap.add_argument("--pipeline", choices=["both", "plain", "fused", "fused_ilv"], default="both",
                    help="N=1: time all pipelines and report the fastest one (default \"both\", a name from when there were two), or only one (profiling runs: "
                         "per-kernel rocprof averages then belong to one kernel variant)")
    ap.add_argument("--no-progressive", action="store_true",
                    help="N=1: skip the progressive / changed_box block (LoadingManager passes, SURVEY 8(f)1)")
    ap.add_argument("--no-host-load", action="store_true", help="N=1: skip the SDFViewer load through libsdfviewer_host.so")
    ap.add_argument("--no-target-512", action="store_true", help="N=1: skip the 512^3 fill block (north-star target config)")
    ap.add_argument("--no-config4", action="store_true", help="N>1: skip the cube-geometry block (BASELINE config 4)")
    ap.add_argument("--config4-side", type=int, default=512,
                    help="N>1: voxels per rank of the cube-geometry block are side^3 (8 x 512^3 = config 4's 1024^3)")
    return ap.parse_args()



def box_stamp(torch, device):
    """Which box, at which clocks: the same binary differs by +-8 % from box to box, so every line says where
    it was measured.  Best effort (sysfs / rocm-smi may be unavailable); never fatal."""
    import socket
    import subprocess
    st = {"host": socket.gethostname()}
    try:
        p = torch.cuda.get_device_properties(device)
        st.update({"name": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count,
                   "hbm_GiB": round(p.total_memory / 2 ** 30, 1)})
        st["uuid"] = str(getattr(p, "uuid", "")) or None
    except Exception as e:  # noqa: BLE001
        st["props_error"] = f"{type(e).__name__}: {e}"
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showuniqueid", "--showperflevel", "--json"], capture_output=True,
                           text=True, timeout=20)
        card = next(iter(json.loads(r.stdout[r.stdout.index("{"):]).values()))
        st["rocm_smi"] = {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "fclk", "unique", "performance"))}
    except Exception as e:  # noqa: BLE001
        st["rocm_smi"] = f"unavailable: {type(e).__name__}"
    st["note"] = "clocks as rocm-smi reports them right after the timed regions (the device is still warm)"
    return st


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
            "sample_short": f"{loads} load(s) of {side}^3 in LoadingManager order ({fill_dt:.1f} s); {rows_done} rows of {W}x{H}",
            "sample": f"{loads} complete load(s) of the {side}^3 grid in LoadingManager order, 2 passes "
                      f"({loads * n_vox} voxels, {fill_dt:.1f} s), oracle/grid_fill.c gcc -O3 -ffp-contract=off, 1 thread",
            "all_cores": {"value": round(all_mvox, 3), "unit": "Mvoxels/s", "cores": threads,
                          "sample": f"{n_all} dense fill(s) of the {side}^3 grid, OpenMP over z ({all_dt:.1f} s); "
                                    f"{threads} = usable cores (affinity {len(os.sched_getaffinity(0))}, cgroup quota applied)"},
            "value_rays": round(rays, 3), "unit_rays": "Mrays/s",
            "sample_rays": f"{rows_done} central rows of the {W}x{H} image, 1 thread",
            "all_cores_rays": {"value": round(all_rays, 3), "unit": "Mrays/s", "cores": threads,
                               "sample": f"{frames} whole {W}x{H} frame(s), OpenMP over rows ({all_rays_dt:.1f} s)"}}


CONTRACT_LIMIT = 4096  # bytes: the driver's capture parsed round 3's 15 KB line and lost round 4's 22 KB one


def _finite(v):
    """json.dumps would write NaN / Infinity (not JSON); the contract line carries null instead."""
    if isinstance(v, float):
        return v if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _finite(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_finite(x) for x in v]
    return v


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def contract_line(line, full_path):
    """The ONE short JSON line of the contract (the reference's analogue is one timing log line, scene/mod.rs:180-191): metric,
    value, ms_per_step, config, dtype, roofline, cpu_baseline and a handful of numbers beside them.  Everything else the run
    measured -- progressive, host_load, batch, loopback, per-step distributions, the prose -- is `line`, written to
    `full_path` and to stderr."""
    c = _pick(line, "metric", "value", "unit", "value_rays", "unit_rays", "n_gpus", "steps", "warmup", "ms_per_step",
              "ms_per_step_fill", "ms_per_step_raymarch", "ms_per_step_median", "ms_per_step_p95", "higher_is_better", "scaling",
              "vs_baseline", "dtype")
    c["data"] = "synthetic"
    cfg = line["config"]
    c["config"] = {"workload": cfg["workload"], "grid": cfg["grid_global"], "image": cfg["image"],
                   "voxels_per_gpu": cfg["voxels_per_gpu"], "cameras_per_gpu": cfg["cameras_per_gpu"],
                   "parallelism": cfg["parallelism"]}
    c["pipeline"] = line.get("pipeline")
    r = line.get("roofline") or {}
    c["roofline"] = _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "algorithmic_bytes_per_launch",
                          "avg_launch_ms", "frac_bus", "rccl_ranks")
    c["build_id"] = line.get("build_id")
    t = r.get("target_512")
    if isinstance(t, dict):
        c["roofline"]["target_512"] = {k: _pick(v, "ms", "frac_8d") for k, v in t.items() if isinstance(v, dict)}
    rr = line.get("roofline_raymarch") or {}
    c["roofline_raymarch"] = _pick(rr, "bound", "achieved", "frac", "traffic", "traffic_stale", "compulsory_bytes", "avg_launch_ms")
    if "bound" in c["roofline_raymarch"]:
        c["roofline_raymarch"]["bound"] = "latency"
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "value_rays", "unit_rays")
        c["cpu_baseline"]["sample"] = cb.get("sample_short", "")
        for k in ("all_cores", "all_cores_rays"):
            if isinstance(cb.get(k), dict):
                c["cpu_baseline"][k] = _pick(cb[k], "value", "cores")
    b = line.get("batch_raymarch")
    if isinstance(b, dict):
        c["batch_raymarch"] = _pick(b, "cameras", "value", "unit", "ms_per_batch", "split", "gather_ms", "value_incl_gather")
        if isinstance(b.get("rgba8_only"), dict):
            c["batch_raymarch"]["rgba8_ms"] = b["rgba8_only"].get("ms_per_batch")
        for k in ("camera_split", "contiguous_rows_split"):
            if isinstance(b.get(k), dict):
                c["batch_raymarch"][k] = _pick(b[k], "value", "ms_per_batch", "gather_ms")
    h = line.get("host_load")
    if isinstance(h, dict):
        c["host_load"] = {k: {"update_ms": v["dense"].get("update_ms"), "load_ms": v["dense"].get("load_ms"),
                              "progressive_load_ms": (v.get("progressive") or {}).get("load_ms")}
                          for k, v in h.items() if isinstance(v, dict) and isinstance(v.get("dense"), dict)}
    ing = line.get("ingest")
    if isinstance(ing, dict) and isinstance(ing.get("whole_load"), dict):
        c["ingest"] = {"Mvoxels_per_s": ing["whole_load"].get("Mvoxels_per_s"), "threads": ing.get("threads"),
                       "one_thread": (ing.get("whole_load_1_thread") or {}).get("Mvoxels_per_s"),
                       "worst_30ms_call": (ing.get("frame_loop_30ms") or {}).get("worst_call_ms"),
                       "batched": ((ing.get("batched") or {}).get("whole_load") or {}).get("Mvoxels_per_s")}
    p = line.get("progressive")
    if isinstance(p, dict):
        c["progressive"] = {side: {name: [case.get("ms"), case.get("frac")] for name, case in cases.items()
                                   if isinstance(case, dict) and "ms" in case}
                            for side, cases in p.items() if isinstance(cases, dict)}
    hl = line.get("halo_loopback")
    if isinstance(hl, dict) and isinstance(hl.get("by_side"), dict):
        c["halo_loopback"] = {k: _pick(v, "ms_per_step", "plain_fill_ms", "fraction_of_plain_fill_rate", "ghosts_verified")
                              for k, v in hl["by_side"].items()}
    if line.get("n_gpus", 1) > 1 or line.get("loopback"):
        c.update(_pick(line, "loopback", "backend", "rccl_ranks", "torch_world_size", "sharded_fill_verified",
                       "fill_step_fraction_of_plain_fill"))
        c["halo_transport"] = (line.get("halo_transport") or "").split(" ")[0] or None
        if isinstance(line.get("config4"), dict):
            c["config4"] = _pick(line["config4"], "grid_global", "value", "ms_per_step_fill", "plain_fill_ms", "error")
        if isinstance(line.get("sharded_march"), dict):
            c["sharded_march"] = _pick(line["sharded_march"], "verified", "ms")
    c["incomplete"] = line.get("incomplete")
    if "watchdog" in line:
        c["watchdog"] = str(line["watchdog"])[:160]
    box = line.get("box") or {}
    smi = box.get("rocm_smi") if isinstance(box.get("rocm_smi"), dict) else {}
    c["box"] = {"uuid": smi.get("Unique ID") or box.get("uuid"), "arch": box.get("arch"),
                "sclk": (smi.get("sclk clock speed:") or "").strip("()") or None}
    c["full"] = full_path
    c = _finite(c)
    # the limit is part of the contract: drop the optional blocks, least important first, should the line ever outgrow it
    for drop in ("progressive", "halo_loopback", "ingest", "host_load", "batch_raymarch", "roofline_raymarch", "box"):
        if len(json.dumps(c, separators=(",", ":"), allow_nan=False)) < CONTRACT_LIMIT:
            break
        c.pop(drop, None)
    return c


def emit(line, redirect):
    """Full record -> $SDFV_BENCH_FULL_JSON (default gpurun_out/bench_full_n<N>.json) and stderr; the contract line -> the LAST
    line of stdout."""
    full_path = os.environ.get("SDFV_BENCH_FULL_JSON") or os.path.join("gpurun_out", f"bench_full_n{line.get('n_gpus', 1)}.json")
    full = json.dumps(_finite(line), allow_nan=False)
    try:
        absolute = full_path if os.path.isabs(full_path) else os.path.join(ROOT, full_path)
        os.makedirs(os.path.dirname(absolute), exist_ok=True)
        with open(absolute, "w") as f:
            f.write(full + "\n")
    except OSError as e:
        full_path = f"(not written: {e})"
    print("[bench full record] " + full, file=sys.stderr, flush=True)
    text = json.dumps(contract_line(line, full_path), separators=(",", ":"), allow_nan=False)
    if redirect is not None:
        redirect.restore()
    print(text, flush=True)


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
                        line["incomplete"] = True  # a hang in an extra must not look like a clean run
                        line["watchdog"] = f"an extra did not finish: stuck in stage '{name}' for {age:.0f} s; the line carries what was measured before it"
                        emit(line, PARTIAL["redirect"])
            finally:
                os._exit(code)

    if watchdog > 0:
        threading.Thread(target=bark, daemon=True).start()
    with NativeStdoutToStderr() as redirect:
        PARTIAL["redirect"] = redirect
        run(redirect)
    done.set()


def run(redirect):
    args = parse_args()
    bench_common.PREWARM_S = args.prewarm_ms / 1e3
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("sdf-viewer_amd")
    par = importlib.import_module("sdf-viewer_amd.parallel")
    from bench_extras import raymarch_rank_cameras_report, raymarch_traffic_report  # the march's byte model beside `roofline`

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
    slab = par.alloc_slab(gdims, rank, world, device, pkg=None if args.separate_textures else pkg, periodic=loopback)
    grid = pkg.make_grid(gdims, z_begin=slab.z_begin, z_end=slab.z_end)
    owned0, owned1 = slab.owned0, slab.owned1
    voxels_per_rank = pkg.slab_voxels(grid)
    total_voxels = voxels_per_rank * world  # identical per rank by construction

    out = {}
    if not multi:
        # ---------------- N = 1: two CONSISTENT pipelines over the SAME buffers ----------------
        #   plain: sdfv_fill_grid (32 B/voxel)         -> sdfv_raymarch over tex0.r in place
        #   fused: sdfv_fill_grid_commit (36 B/voxel)  -> sdfv_raymarch_ex (desc.dist) over the compact distance volume it wrote
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
        # third pipeline (round 4): the fused fill writes the march's y-INTERLEAVED volume itself (SDFV_PASS_VOLUME_INTERLEAVED: a
        # step-1 virgin pass = the dense kernel, textures + that volume, 36 B/voxel) and the march gathers from it -- what
        # SDFViewer does for grids beyond the last-level cache; no commit pass between the two
        fill_ilv_ms = fill_ilv_ev = march_ilv_ms = march_ilv_ev = inter_ilv_ms = INF
        ILV_FLAGS = pkg._capi.PASS_VIRGIN_GRID | pkg._capi.PASS_VOLUME_INTERLEAVED
        if args.pipeline in ("both", "fused_ilv") and side % 2 == 0:
            ilv_vol = torch.empty((side, side, side), dtype=torch.float32, device=device)

            def fill_ilv():
                pkg.fill_grid_pass(prm, grid, 1, owned0, owned1, dist=ilv_vol, flags=ILV_FLAGS)

            def march_ilv():
                pkg.raymarch(rp, owned0, owned1, cam0, W, H, out=rgba, ilv=ilv_vol)

            fill_ilv_ms, fill_ilv_ev = region(fill_ilv, K, Wm, torch, dist, 1, device)
            march_ilv_ms, march_ilv_ev = region(march_ilv, K, Wm, torch, dist, 1, device)
            inter_ilv_ms = region(lambda: (fill_ilv(), march_ilv()), K, Wm, torch, dist, 1, device)[0]
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
                      "march": "sdfv_raymarch_ex over the distance volume (desc.dist)",
                      "ms_fill": round(fill_fused_ms, 4), "ms_raymarch": round(march_dist_ms, 4),
                      "ms_per_step": round(fill_fused_ms + march_dist_ms, 4),
                      "ms_per_step_interleaved": round(inter_fused_ms, 4),
                      "Mvoxels_s": round(voxels_per_rank / fill_fused_ms / 1e3, 1), "Mrays_s": round(W * H / march_dist_ms / 1e3, 1),
                      "fill_frac_of_hbm_peak": round(36 * voxels_per_rank / (fill_fused_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "fill_frac_8d": round(32 * voxels_per_rank / (fill_fused_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        pipes["fused_ilv"] = {"fill": "sdfv_fill_grid_pass_ex(step 1, VIRGIN_GRID | VOLUME_INTERLEAVED) (36 B/voxel: textures + the y-interleaved volume)",
                              "march": "sdfv_raymarch_ex over that volume (desc.ilv)",
                              "ms_fill": round(fill_ilv_ms, 4), "ms_raymarch": round(march_ilv_ms, 4),
                              "ms_per_step": round(fill_ilv_ms + march_ilv_ms, 4), "ms_per_step_interleaved": round(inter_ilv_ms, 4),
                              "Mvoxels_s": round(voxels_per_rank / fill_ilv_ms / 1e3, 1), "Mrays_s": round(W * H / march_ilv_ms / 1e3, 1),
                              "fill_frac_of_hbm_peak": round(36 * voxels_per_rank / (fill_ilv_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "fill_frac_8d": round(32 * voxels_per_rank / (fill_ilv_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        totals = {"plain": fill_plain_ms + march_tex0_ms, "fused": fill_fused_ms + march_dist_ms, "fused_ilv": fill_ilv_ms + march_ilv_ms}
        chosen = min(totals, key=totals.get)
        if chosen == "fused":
            fill_ms, march_ms, kern_ms, march_ev, bpv = fill_fused_ms, march_dist_ms, fill_fused_ev, march_dist_ev, 36
        elif chosen == "fused_ilv":
            fill_ms, march_ms, kern_ms, march_ev, bpv = fill_ilv_ms, march_ilv_ms, fill_ilv_ev, march_ilv_ev, 36
        else:
            fill_ms, march_ms, kern_ms, march_ev, bpv = fill_plain_ms, march_tex0_ms, fill_plain_ev, march_tex0_ev, 32
        fill_mvox = voxels_per_rank / fill_ms / 1e3
        march_mrays = W * H / march_ms / 1e3
        # the distribution behind the two means (SURVEY 8d asks for the median): launches timed one by one
        if chosen == "fused_ilv":
            st_fill = per_step_stats(fill_ilv, args.per_step_samples, torch)
            st_march = per_step_stats(march_ilv, args.per_step_samples, torch)
        elif chosen == "fused":
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
        out["pipeline_fused_ilv"] = pipes["fused_ilv"] if (args.pipeline in ("both", "fused_ilv") and side % 2 == 0) else None
        out["pipeline_note"] = ("three consistent pipelines over the same buffers (plain; fused = + the distance volume; fused_ilv = + the "
                                "y-interleaved volume instead); value, value_rays, ms_per_step and "
                                "roofline all come from `pipeline` (the faster one end to end); *_interleaved = K steps "
                                "of fill immediately followed by its march in one timed region")
        out["pipeline_overlapped"] = overlapped
        out["commit_ms"] = round(commit_ms, 4) if commit_ms != INF else None
        out["commit_note"] = ("sdfv_commit_distance as a pass of its own (device-side SDFViewer::commit for a grid filled "
                              "without the volume); not part of either pipeline")
        traffic, stale = None, None
        for key in {"plain": ("",), "fused": ("_fused",), "fused_ilv": ("_fused_ilv", "_fused")}[chosen]:  # (ilv: the same bytes)
            if traffic is None:
                traffic, stale = load_traffic(args.workload + key), traffic_stale(args.workload + key)
        out["roofline"] = fill_roofline(kern_ms, voxels_per_rank, bpv, traffic)
        out["roofline"]["traffic_stale"] = stale  # True: the PMC pass traced another build than the one timed here
        if chosen == "fused_ilv" and side % 256 == 0 and side >= 512:  # (csrc/fill_kernels.hip launch_fill_dense: rows two workgroups wide)
            out["roofline"]["kernel"] = "fill_dense_ilv_paired_kernel"
        out["roofline_raymarch"] = raymarch_traffic_report(args.workload, march_ev,
                                                           "tex0_path" if chosen == "plain" else "product_path",
                                                           args.workload + {"plain": "_tex0", "fused": "", "fused_ilv": "_ilv"}[chosen])
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
        out["roofline"]["traffic_stale"] = traffic_stale(args.workload + "_fused")
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
        out["roofline"]["rccl_ranks"] = out["rccl_ranks"]  # (the driver's record keeps nested objects whole)
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
    line["incomplete"] = None  # True only when the watchdog had to print the line for an extra that hung
    line["box"] = box_stamp(torch, device)
    line["build_id"] = running_build_id()  # sdfv_build_id(): which kernel sources the timed library was built from
    for key in ("batch_raymarch", "target_512", "progressive", "host_load", "ingest", "halo_loopback", "config4"):
        line[key] = None
    PARTIAL["line"], PARTIAL["since"] = line, time.monotonic()

    # ---------------- everything beside the contract (bench_extras.py): never `value`, never fatal ----------------
    import bench_extras
    ctx = SimpleNamespace(args=args, pkg=pkg, par=par, torch=torch, dist=dist, world=world, rank=rank, device=device, cdev=cdev,
                          multi=multi, loopback=loopback, backend=backend, transport=transport, filler=filler, slab=slab,
                          grid=grid, gdims=gdims, side=side, W=W, H=H, K=K, prm=prm, rp=rp, r0=r0, r1=r1, owned0=owned0,
                          owned1=owned1, dist_vol=dist_vol, slab_dist=slab_dist if multi else None, line=line)
    bench_extras.run_extras(ctx)
    PARTIAL["since"] = None  # the extras are done (the CPU baseline below only uses this process's host cores)
    if rank == 0:
        if not args.no_cpu_baseline and not multi:  # rank 0, N = 1 only
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_baseline_seconds)
        emit(line, redirect)
    if multi:
        if getattr(filler, "comm", None) is not None:
            torch.cuda.synchronize()
            filler.comm.close()  # the library's RCCL communicator, while every peer is still alive
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

