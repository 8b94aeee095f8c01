"""bench_common.py -- the timing discipline bench.py (the contract line) and bench_extras.py (everything beside it) share:
untimed pre-warm, K steps bracketed by barrier + synchronize with MAX over ranks, per-launch HIP-event statistics, and the
fill's roofline on SURVEY 8(d)'s bytes.  No measurement lives here."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FILL_BYTES_PER_VOXEL = 32  # tex0 16 B + tex1 16 B, store-only

PREWARM_S = 0.25  # --prewarm-ms (bench.py sets it)


def STAGE(name):
    """Registers the collective stage about to block (parallel.enter_stage) for the watchdog's report."""
    par = sys.modules.get("sdf-viewer_amd.parallel")
    if par is not None:
        par.enter_stage("bench.py " + name)


def prewarm(fn, torch, dist=None, world=1, device=None, seconds=None):
    """Untimed: keep the device busy with `fn` for ~0.25 s so that the timed steps run at the clocks a busy GPU runs at.
    The MI355X idles at a few hundred MHz and needs ~10 ms of load to ramp (EXPERIMENTS, round 1 clock-ramp probe: the first 8 ms of
    256^3 fills after an idle period are 9 % slower than the steady state, and on some boxes the rate keeps drifting
    for about a second: round 2 texture-skew sweep); a few warm-up steps of 0.1 ms each do not get it there.  At N > 1 the steps contain exchanges, so every rank must make the SAME number of calls: the
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
    if args.separate_textures:
        return "two separate allocations"
    gap = slab.tex1.data_ptr() - slab.tex0.data_ptr() - slab.tex0.numel() * 4
    return f"one block, tex1 {gap} B after tex0's end: SDFViewer::new_voxels' fixed placement for this texture size (no probe)"


def _traffic_entry(workload_key, name):
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", name))).get(workload_key)
        return e if isinstance(e, dict) else None
    except Exception:
        return None


def load_traffic(workload_key, name="fill_pmc_traffic.json"):
    """HBM bytes per launch from the committed PMC pass (profiles/*_pmc_traffic.json), or None."""
    e = _traffic_entry(workload_key, name)
    return None if e is None else e.get("hbm_bytes_per_launch")


def traffic_stale(workload_key, name="fill_pmc_traffic.json"):
    """Does the committed PMC pass describe ANOTHER build than the library that is running?  The pass is stamped with the build
    id of the library it traced (tools/pmc_to_traffic.py, from the summary's header); True when that differs from
    sdfv_build_id() of the running library (or the entry predates the stamp), None when there is no entry."""
    e = _traffic_entry(workload_key, name)
    if e is None:
        return None
    return e.get("build_id") != running_build_id()


def running_build_id():
    """sdfv_build_id() of the library the bench runs: sha256[:12] over the kernel sources it was built from."""
    try:
        import importlib
        return importlib.import_module("sdf-viewer_amd").lib.sdfv_build_id().decode()
    except Exception:
        return None


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
            "frac_note": "frac = frac_8d = 32 B/voxel x voxels / avg_launch_ms / 8 TB/s (SURVEY 8d), on the HIP-EVENT time of the K "
                         "launches; `value` is the same K launches on the WALL clock between barrier + synchronize (the contract), so "
                         "value / 250 000 is 1-2 % below frac (the first launch's latency and the final synchronize); "
                         "frac_bus counts every byte the launch stores (36 B/voxel when it also writes the distance volume)",
            "avg_launch_ms": round(kern_ms, 5),
            "avg_launch_note": "HIP events around K back-to-back launches / K: includes the ~6 us gap between "
                               "launches, which rocprofv3's kernel-only average leaves out (8 % at 256^3, under 1 % at 512^3)"}
