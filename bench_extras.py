"""bench_extras.py -- what bench.py's JSON line carries BESIDE the contract's measurement: the 512^3 target block, the
multi-GPU fill step in loopback, the progressive / changed_box cases, the drop-in host's load (libsdfviewer_host.so), the
64-camera batch with its splits and gathers, BASELINE config 4's geometry and the N > 1 self-checks.  None of it is `value`;
every block is wrapped so that it can fail (or hang: bench.py's watchdog) without losing the contract's numbers.

run_extras(c) receives bench.py's state as one namespace and completes c.line in place."""
import json
import os
import subprocess
import sys
import time

from bench_common import (HBM_PEAK_GBS, ROOT, STAGE, load_traffic, per_step_stats, placement_note, region, timed_region, traffic_stale)


def raymarch_traffic_report(workload_key, launch_ms, path_key="product_path", traffic_key=None):
    """SURVEY.md 8(d)'s raymarch byte model next to the measured figures.  compulsory / nominal bytes come from the
    oracle's deterministic counts (tools/raymarch_bytes.py -> profiles/raymarch_model_bytes.json, committed); `traffic`
    is the HBM bytes of one launch from the committed PMC pass.  The kernel is bound by dependent-gather latency /
    instruction issue, not by HBM: `frac` says how far from the HBM roofline the COMPULSORY bytes are."""
    rep = {"kernel": "raymarch_kernel", "bound": "latency (<=255 dependent gathers per ray), not hbm",
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": round(launch_ms, 5)}
    traffic = load_traffic(traffic_key or workload_key, "raymarch_pmc_traffic.json") if workload_key else None
    stale = traffic_stale(traffic_key or workload_key, "raymarch_pmc_traffic.json") if workload_key else None
    if traffic is None and traffic_key and traffic_key.endswith("_ilv"):  # no PMC pass of the interleaved march yet: the distance-volume one
        traffic, stale = load_traffic(workload_key, "raymarch_pmc_traffic.json"), traffic_stale(workload_key, "raymarch_pmc_traffic.json")
    rep["traffic_stale"] = stale
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


def host_load_block(sides):
    """VERDICT r03 missing 5: what a host that drops libsdfviewer_host.so in pays for a load -- SDFViewer::from_bb -> update
    -> commit (reference: src/app/scene/mod.rs:139-156, src/app/scene/sdf/mod.rs:46-101,128-239) -- measured by the C++
    program sdf-viewer_amd/sdf-viewer-host-bench (host/host_load_bench.cpp), in a process of its own, through the product
    library, not through the kernels' entry points."""
    exe = os.path.join(ROOT, "sdf-viewer_amd", "sdf-viewer-host-bench")
    out = {}
    for side in sides:
        key = f"{side}"
        try:
            r = subprocess.run([exe, "--side", str(side), "--reps", "20"], capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[key] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:  # noqa: BLE001 -- an extra, never fatal
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    out["note"] = ("C++ SDFViewer (libsdfviewer_host.so): load_ms = update() to completion + commit() on the GPU (HIP events); dense = "
                   "one update() with the reference's 30 ms budget (the fused fill; new_voxels writes nothing), progressive = one "
                   "LoadingManager pass per update() (2 passes: whole visited rows, then the dense kernel); commit_ms = the march "
                   "volume the following frames read")
    return out


def ingest_block(side=384):
    """SDFViewer::update for an SDF only the HOST can sample (any `impl SDFSurface`, scene/sdf/mod.rs:128):
    tests/c/gyroid_provider.c (a library behind include/sdf_provider.h's per-point ABI) built here with gcc, loaded through
    ProviderSDF, loaded into a side^3-bounded grid by sdf-viewer-host-bench --ingest.  CPU-bound by construction (one malloc'ing
    FFI call per voxel, like the reference's wasm provider): the figure is host sampling throughput, the device's share (H2D of 32 B
    per voxel + sdfv_pack_samples) hides behind it.  `batched`: the same SDF from a library that also exports the optional
    `sample_batch` (the trait's "Batched sampling" TODO): one call per gathered block, no allocation per point."""
    import tempfile
    exe = os.path.join(ROOT, "sdf-viewer_amd", "sdf-viewer-host-bench")

    def run(batched):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                lib = os.path.join(tmp, "libgyroid_provider.so")
                subprocess.run(["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-I",
                                os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "gyroid_provider.c"), "-o", lib, "-lm"] +
                               (["-DGYROID_BATCH"] if batched else []), check=True, capture_output=True, timeout=120)
                r = subprocess.run([exe, "--ingest", lib, "--side", str(side), "--passes", "2"], capture_output=True, text=True, timeout=300)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            return json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:  # noqa: BLE001 -- an extra, never fatal
            return {"error": f"{type(e).__name__}: {e}"}

    out = run(False)
    out["batched"] = run(True)
    out["note"] = ("host-sampled SDF (gyroid behind the per-point ABI) -> pinned H2D -> sdfv_pack_samples; whole_load = one update() with "
                   "an unlimited budget on `threads` host threads, frame_loop_30ms = the reference's frame loop (30 ms per call; "
                   "worst_call_ms = the longest call), setup_ms = the first call (transfer buffers, host mirror, workers; not in load_ms), "
                   "whole_load_1_thread = the reference's single-threaded loop; batched = the provider also exports sample_batch; CPU-bound")
    return out


def progressive_block(pkg, torch, prm, sides, reps=7):
    """SURVEY 8(f)1 under the bench's measurement discipline: the LoadingManager passes (loading.rs:50-76) with
    update_required (scene/sdf/mod.rs:184-190) on the device, over textures that travel with their distance volume
    (sdfv_fill_grid_pass_ex with its `dist` argument).  Per case: median ms over `reps` runs (HIP events; the state is re-created, untimed, before
    every run), visited voxels, updated voxels, and the fraction of the HBM roofline on SURVEY 8(d)'s incremental figure,
    36 B per UPDATED voxel (4 B read + 32 B written) + 4 B per voxel that is visited only."""
    AIR = pkg.AIR_DIST
    out = {}
    for side in sides:
        g = pkg.make_grid((side,) * 3)
        # one block, tex1 at the viewer's default distance from tex0 (host/sdf_viewer.cpp, pkg.default_texture_skew): the product
        # host's placement is what is timed.  (The whole-rows pass stays bimodal from PROCESS to process all the same -- 28 or 44 us
        # at 256^3 on one box, whichever way the textures are placed virtually: EXPERIMENTS R4.4.)
        t0, t1 = pkg.alloc_textures_placed(g)
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

        FRESH, SAME, VIRGIN = pkg._capi.PASS_FRESH_GRID, pkg._capi.PASS_SAME_LOAD, pkg._capi.PASS_VIRGIN_GRID

        def virgin_passes(steps):
            # what host/sdf_viewer.cpp enqueues since round 4: new_voxels writes nothing, every pass of the load says so
            return lambda: [pkg.fill_grid_pass(prm, g, st, t0, t1, dist=dist, flags=VIRGIN | SAME) for st in steps]

        NOOP = pkg._capi.PASS_EXPECT_NOOP

        def passes(steps, box=None, flagged=False, hint=0):
            # flagged: what host/sdf_viewer.cpp's LoadingManager tells the library (sdfv_fill_grid_pass_ex): the first pass
            # of a load sees a fresh grid, the later ones revisit what the same load wrote -- nothing is read
            return lambda: [pkg.fill_grid_pass(prm, g, st, t0, t1, changed_box=box, dist=dist,
                                               flags=((FRESH | SAME) if k == 0 else SAME) if flagged else hint)
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
        res["virgin_load_2_passes"] = case(timed(virgin_passes((2, 1)), lambda: None), visited((2, 1)), n,
                                           "the same load over a VIRGIN grid (SDFV_PASS_VIRGIN_GRID: new_voxels' [AIR_DIST; 4] is never "
                                           "written; step 2 writes its rows whole, step 1 is the dense kernel) -- what SDFViewer::update "
                                           "enqueues; no sdfv_grid_init / sdfv_commit_distance precedes it")
        res["virgin_load_2_passes"]["algorithmic_bytes"] = 36 * (n + side * (-(-side // 2)) ** 2)  # every voxel once + the step-2 rows
        res["virgin_load_2_passes"]["frac"] = round(res["virgin_load_2_passes"]["algorithmic_bytes"] / (res["virgin_load_2_passes"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res["fresh_pass_step_2_flagged"] = case(timed(passes((2,), flagged=True), fresh), visited((2,)), visited((2,)),
                                                "first pass of that load alone (whole visited rows written: 1/4 of the textures)")
        res["fresh_load_2_passes_unflagged"] = case(timed(passes((2, 1)), fresh), visited((2, 1)), n,
                                                    "the same load, the caller saying nothing (sdfv_fill_grid_pass_ex, flags 0; update_required read from the volume)")
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
        res["noop_pass_step_1"] = case(timed(passes((1,), hint=NOOP), loaded), n, 0,
                                       "step-1 pass over a loaded grid, no box, as SDFViewer::update enqueues it once an edit has been worked off "
                                       "(SDFV_PASS_EXPECT_NOOP: beyond the last-level cache the scan streams the volume with nontemporal loads): "
                                       "reads the volume, writes nothing")
        res["noop_pass_step_1_unhinted"] = case(timed(passes((1,)), loaded), n, 0, "the same pass from a caller that says nothing (cached loads)")
        res["dense_fused_fill_ms"] = round(timed(lambda: pkg.fill_grid(prm, g, t0, t1, dist=dist), lambda: None), 4)
        # HBM bytes per case from the committed PMC passes, and the same kernels' durations under rocprofv3 (warm, 100
        # repetitions: tools/gpu_profile_pass.sh -> profiles/pass_traffic.json, regenerated per round) next to the times measured here
        try:
            whole = json.load(open(os.path.join(ROOT, "profiles", "pass_traffic.json")))[str(side)]
            prof = whole["cases"]
            from bench_common import running_build_id
            res["traffic_stale"] = whole.get("build_id") != running_build_id()  # the PMC passes traced another build
        except Exception:  # noqa: BLE001
            prof = {}
        for name, c in res.items():
            p = prof.get(name) if isinstance(c, dict) else None
            if p and p.get("complete"):
                c["traffic"] = p["hbm_bytes"]
                c["traffic_over_algorithmic"] = round(p["hbm_bytes"] / c["algorithmic_bytes"], 3)
                c["rocprof_ms"] = p["rocprof_ms"]
        out[str(side)] = res
        del t0, t1, dist
    out["note"] = ("sdfv_fill_grid_pass_ex; traffic = HBM bytes of the case's kernels from the committed PMC passes (WRITE_SIZE + 2 x "
                   "FETCH_SIZE), rocprof_ms = their kernel-only durations under rocprofv3 (no launch gaps: a few us per pass below "
                   "`ms`); frac = (36 B x updated + 4 B x visited-only voxels) / ms / 8 TB/s; every intermediate "
                   "state is bit-identical to the oracle's LoadingManager loop (tests/test_gpu_fill.py)")
    return out


def batch_valu_roofline(workload_key, world, ms_per_batch):
    """The 64-camera batch fills the machine with short waves; is it bound by VALU ISSUE?  CDNA4's SIMDs are 32 lanes wide
    (MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues over 2 cycles; 157.3 TFLOP/s fp32 = 64 FLOP/clk/SIMD),
    so a SIMD retires at most clock / 2 wave instructions per second.  frac = (VALU wave-instructions of one batch, PMC
    SQ_INSTS_VALU, committed pass) / (1024 SIMDs x clock / 2 x batch time).  Rounds 2-3 priced this with clock / 4 (the 16-lane
    SIMD of earlier CDNA): their 0.67-0.72 is 0.34-0.36 on this scale -- the batch is not VALU-issue bound either way (removing a
    sixth of its instructions moved nothing, EXPERIMENTS R3.9); its gathers are what it waits for."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "raymarch_batch_valu.json")))[workload_key]
    except Exception:  # noqa: BLE001
        return None
    simds, clock = 1024 * world, float(d.get("clock_GHz", 2.4))
    peak = simds * clock * 1e9 / 2.0
    rate = d["valu_wave_instructions_per_batch"] / (ms_per_batch * 1e-3)
    return {"bound": "valu issue", "valu_wave_instructions_per_batch": d["valu_wave_instructions_per_batch"],
            "achieved": round(rate / 1e12, 3), "peak": round(peak / 1e12, 3), "unit": "T wave-instructions/s",
            "frac": round(rate / peak, 4), "simds": simds, "clock_GHz": clock, "source": d.get("source"),
            "note": "wave64 VALU instruction = 2 issue cycles on a CDNA4 SIMD-32; peak = SIMDs x clock / 2 (rounds 2-3 used / 4: twice this frac)"}


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


def run_extras(c):
    """c: bench.py's state after the contract's two timed regions (a SimpleNamespace).  Completes c.line in place."""
    args, pkg, par, torch, dist = c.args, c.pkg, c.par, c.torch, c.dist
    world, rank, device, cdev = c.world, c.rank, c.device, c.cdev
    multi, loopback, backend, transport, filler = c.multi, c.loopback, c.backend, c.transport, c.filler
    slab, grid, gdims, side, W, H, K, prm, rp = c.slab, c.grid, c.gdims, c.side, c.W, c.H, c.K, c.prm, c.rp
    r0, r1, owned0, owned1, dist_vol, slab_dist, line = c.r0, c.r1, c.owned0, c.owned1, c.dist_vol, c.slab_dist, c.line
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
                    t_slab = par.alloc_slab((512, 512, 512), 0, 1, device, pkg=None if args.separate_textures else pkg)
                    t_grid = pkg.make_grid((512, 512, 512))
                t_dist = torch.empty((512, 512, 512), dtype=torch.float32, device=device)
                ts = max(5, min(K, 20))
                t_ms, t_ev = region(lambda: pkg.fill_grid(prm, t_grid, t_slab.owned0, t_slab.owned1), ts, 2, torch, dist, 1, device)
                f_ms, f_ev = region(lambda: pkg.fill_grid(prm, t_grid, t_slab.owned0, t_slab.owned1, dist=t_dist), ts, 2,
                                    torch, dist, 1, device)
                ilv_flags = pkg._capi.PASS_VIRGIN_GRID | pkg._capi.PASS_VOLUME_INTERLEAVED
                i_ms, i_ev = region(lambda: pkg.fill_grid_pass(prm, t_grid, 1, t_slab.owned0, t_slab.owned1, dist=t_dist, flags=ilv_flags),
                                    ts, 2, torch, dist, 1, device)
                n512 = 512 ** 3
                target_512 = {"grid": [512, 512, 512], "steps": ts, "ms_fill": round(t_ms, 4),
                              "Mvoxels_s": round(n512 / t_ms / 1e3, 1), "avg_launch_ms": round(t_ev, 5),
                              "achieved_GBs": round(32 * n512 / (t_ev * 1e-3) / 1e9, 1),
                              "frac": round(32 * n512 / (t_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "frac_8d": round(32 * n512 / (t_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "per_launch": per_step_stats(lambda: pkg.fill_grid(prm, t_grid, t_slab.owned0, t_slab.owned1),
                                                           min(args.per_step_samples, 30), torch, warm=2),
                              "fused_commit": {"ms_fill": round(f_ms, 4), "Mvoxels_s": round(n512 / f_ms / 1e3, 1),
                                               "avg_launch_ms": round(f_ev, 5),
                                               "achieved_GBs": round(36 * n512 / (f_ev * 1e-3) / 1e9, 1),
                                               "frac": round(36 * n512 / (f_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "frac_8d": round(32 * n512 / (f_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                               "frac_note": "frac: 36 B/voxel on the bus; frac_8d: SURVEY 8(d)'s 32 B/voxel"},
                              "fused_ilv": {"ms_fill": round(i_ms, 4), "Mvoxels_s": round(n512 / i_ms / 1e3, 1), "avg_launch_ms": round(i_ev, 5),
                                            "frac": round(36 * n512 / (i_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "frac_8d": round(32 * n512 / (i_ev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "what": "the fill that writes the march's y-interleaved volume (what SDFViewer runs at this size)"},
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
    if isinstance(target_512, dict) and "frac_8d" in target_512:
        # the configuration the north-star target is worded on, INSIDE the nested object the driver's record keeps whole
        line["roofline"]["target_512"] = {"plain": {"ms": target_512["avg_launch_ms"], "frac_8d": target_512["frac_8d"]},
                                          "fused": {"ms": round(target_512["fused_commit"]["avg_launch_ms"], 5),
                                                    "frac_8d": target_512["fused_commit"]["frac_8d"],
                                                    "frac_bus": target_512["fused_commit"]["frac"]},
                                          "fused_ilv": {"ms": target_512["fused_ilv"]["avg_launch_ms"], "frac_8d": target_512["fused_ilv"]["frac_8d"]},
                                          "target_frac": 0.70, "note": "512^3 dense fill, HIP events; full block: top-level target_512"}
    host_load = None
    if not multi and not args.no_batch and not args.no_host_load:
        host_load = host_load_block(sorted({side, 512}))
    line["host_load"] = host_load
    line["ingest"] = ingest_block() if (not multi and not args.no_batch and not args.no_host_load) else None
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
            camera per rank (sdfv_march_desc.band_first / band_step); rows = one contiguous range of rows of every camera per rank; cameras =
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
            # the cameras are a HOST array here, as a host that moves them every batch hands them over: beyond 16 the launcher
            # copies them to stream-ordered device memory inside the call (timed above).  Resident cameras, for comparison:
            ms_resident = None
            if len(mine) > 16:
                host_cams, mine = mine, pkg.upload_cameras(mine, device=device)
                batch_step()
                ms_resident = timed_region(batch_step, batch_steps, torch, dist, world, device)[0] / batch_steps * 1e3
                mine = host_cams
            # the same batch into the 8-bit UNORM plane only (sdfv_march_desc.rgba8: the reference's framebuffer format): 4 B per
            # pixel stored instead of 16 -- what a host that displays, encodes or gathers the images wants
            batch_out8 = torch.empty((len(mine), n_rows, W, 4), dtype=torch.uint8, device=device)

            def batch_step8():
                pkg.raymarch(rp, r0, r1, mine, W, H, dist=dist_vol, rgba8=batch_out8, f32=False, **where, **(accel_kw if use_pairs else {}))

            ms_rgba8 = None
            if use_pairs:
                batch_step8()
                ms_rgba8 = timed_region(batch_step8, batch_steps, torch, dist, world, device)[0] / batch_steps * 1e3
            rep = {"split": split if world > 1 else None, "band_height": where["bands"][2] if "bands" in where else None,
                   "cameras_per_gpu": len(mine), "rows_per_gpu": n_rows,
                   "value": round(n_batch * W * H / ms / 1e3, 1), "unit": "Mrays/s", "ms_per_batch": round(ms, 4),
                   "march_over": (f"y-{volume_kind} volume" if volume_kind == "interleaved" else "y-pair volume") if use_pairs
                                 else "distance volume"}
            if ms_resident is not None:
                rep["ms_per_batch_cameras_in_device_memory"] = round(ms_resident, 4)
            if ms_rgba8 is not None:
                rep["rgba8_only"] = {"ms_per_batch": round(ms_rgba8, 4), "value": round(n_batch * W * H / ms_rgba8 / 1e3, 1),
                                     "note": "same batch, outColor stored as 8-bit UNORM only (sdfv_march_desc.rgba8, rgba = NULL)"}
            if multi and use_pairs:
                # SURVEY 8(e)'s collective of config 5: the images assembled on rank 0.  Over the library communicator when the
                # step runs on it (sdfv_comm_gather_bands / _gather_cameras: one message per peer, all links into one rank),
                # torch.distributed otherwise (gloo tests).  n_batch x W x H x 16 B into ONE rank dwarfs a rank's share of the
                # march: said here with its own number rather than left out of the line.
                lib_comm = getattr(filler, "comm", None) if transport == "rccl" else None

                def gather_step():
                    if split == "tiles" and world > 1:
                        return par.gather_bands(batch_out, H, rank, world, comm=lib_comm, band_height=where["bands"][2])
                    if split == "cameras":
                        return par.gather_images(batch_out, n_batch, rank, world, comm=lib_comm)
                    return par.gather_rows(batch_out, H, rank, world) if world > 1 else batch_out

                try:
                    STAGE(f"batch gather ({split}) to rank 0")
                    gather_step()
                    g_dt, _ = timed_region(gather_step, 2, torch, dist, world, device)
                    g_ms = g_dt / 2 * 1e3
                    rep.update({"gather_ms": round(g_ms, 4), "gather_bytes_to_rank0": n_batch * W * H * 16 * (world - 1) // max(world, 1),
                                "gather_transport": "library RCCL communicator (sdfv_comm_gather_*)" if lib_comm is not None
                                                    else "torch.distributed gather",
                                "value_incl_gather": round(n_batch * W * H / (ms + g_ms) / 1e3, 1)})
                except Exception as e:  # noqa: BLE001 -- an extra, never fatal
                    rep["gather_ms"] = f"error: {type(e).__name__}: {e}"
                # the same gather of the 8-bit UNORM plane (what the reference's framebuffer holds): a quarter of the bytes
                try:
                    def gather_step8():
                        if split == "tiles" and world > 1:
                            return par.gather_bands(batch_out8, H, rank, world, comm=lib_comm, band_height=where["bands"][2])
                        if split == "cameras":
                            return par.gather_images(batch_out8, n_batch, rank, world, comm=lib_comm)
                        return par.gather_rows(batch_out8, H, rank, world) if world > 1 else batch_out8
                    STAGE(f"batch gather of the rgba8 plane ({split}) to rank 0")
                    gather_step8()
                    g8 = timed_region(gather_step8, 2, torch, dist, world, device)[0] / 2 * 1e3
                    rep["rgba8_only"].update({"gather_ms": round(g8, 4), "gather_bytes_to_rank0": n_batch * W * H * 4 * (world - 1) // max(world, 1),
                                              "value_incl_gather": round(n_batch * W * H / (ms_rgba8 + g8) / 1e3, 1)})
                except Exception as e:  # noqa: BLE001 -- an extra, never fatal
                    rep["rgba8_only"]["gather_ms"] = f"error: {type(e).__name__}: {e}"
                rep["headline"] = ("`value`: every rank keeps the bands it rendered (SURVEY 8(e): the gather is optional); value_incl_gather / "
                                   "rgba8_only.value_incl_gather = with the images assembled on rank 0 as fp32 / as the 8-bit framebuffer format")
            return rep

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
        # `value` above is the march of one batch over a volume built ONCE per load; a reader who counts the volume
        # against this batch alone reads value_incl_commit -- said here rather than left to the note
        batch_report["value_excludes"] = "the one-off commit of the march volume (commit_pairs_ms, once per LOAD; value_incl_commit charges it to this one batch)"
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
        # how far the batch is from the VALU ISSUE roofline: VALU wave-instructions per batch / (SIMDs x clock / 2 cycles per wave64
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
                cslab = par.alloc_slab(cdims, rank, world, device, pkg=None if args.separate_textures else pkg, periodic=loopback)
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
                full0, full1 = par.gather_replica(slab, gdims, world, comm=getattr(filler, "comm", None) if transport == "rccl" else None)
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
