#!/usr/bin/env python3
"""One rank's multi-GPU fill step, measured on ONE GPU in a process of its own (bench.py runs this at N = 1 and
attaches the result as "halo_loopback"): a periodic communicator of world size 1 makes the rank its own z-neighbour, so
the boundary-first fill, the signal / event hand-off, the RCCL group on its own stream, the ghost copy and both stream
dependencies all run; only the xGMI transfer is missing.  The communicator is created before the first kernel, as the
process group is at N > 1 (a stream whose hardware queue predates the process's first RCCL communicator runs the
step's fill / record / wait pattern 2.5x slower -- DESIGN.md 6).  Prints one JSON line.
Usage: python tools/slab_step_probe.py <side> <steps> [--forms]     (--forms: every form of the step, interleaved)"""
import importlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # see bench.py: keeps the step's streams on separate hardware queues

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    side = int(args[0]) if len(args) > 0 else 256
    steps = int(args[1]) if len(args) > 1 else 30
    pkg = importlib.import_module("sdf-viewer_amd")
    par = importlib.import_module("sdf-viewer_amd.parallel")
    K = pkg._capi
    torch.cuda.set_device(0)
    comm = par.SlabComm(pkg, 0, 1, periodic=True)  # first: see the docstring
    prm = pkg.default_params()
    dims = (side, side, side)
    grid = pkg.make_grid(dims)
    slab = par.alloc_slab(dims, 0, 1, "cuda", periodic=True, pkg=pkg)

    def run(fn, n, warm_s=0.25, after=None):
        t_end = time.perf_counter() + warm_s  # device clock ramp, like bench.py's pre-warm
        while time.perf_counter() < t_end:
            fn()
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        if after is not None:
            after()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def ghosts_ok():
        return bool(torch.equal(slab.tex0[0], slab.owned0[-1]) and torch.equal(slab.tex1[-1], slab.owned1[0]) and
                    torch.equal(slab.tex1[0], slab.owned1[-1]) and torch.equal(slab.tex0[-1], slab.owned0[0]))

    dist = torch.empty(tuple(slab.tex0.shape[:3]), dtype=torch.float32, device="cuda")
    own_dist = dist[slab.ghost_lo:slab.ghost_lo + side]

    def step():
        comm.fill_step(prm, grid, slab)

    def plain():
        pkg.fill_grid(prm, grid, slab.owned0, slab.owned1)

    def step_fused():  # what bench.py --gpus N runs per rank: the fused fill (textures + distance volume) + the exchange
        comm.fill_step(prm, grid, slab, dist=dist)

    def plain_fused():
        pkg.fill_grid(prm, grid, slab.owned0, slab.owned1, dist=own_dist)

    out = {"steps": steps, "wait_value_capable": comm.wait_value_capable}
    if "--graph" in sys.argv and "--graph-child" not in sys.argv:
        # the capture crashes the process on this image (hipStreamEndCapture): run it in a child and report
        import subprocess
        comm.close()
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__)] + sys.argv[1:] + ["--graph-child"],
                           capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out = json.loads(lines[-1]) if lines else {}
        out["graph_child_returncode"] = r.returncode  # -11: segmentation fault inside the capture
        if r.returncode != 0:
            out["graph_child_stderr_tail"] = r.stderr[-600:]
        print(json.dumps(out), flush=True)
        return
    if "--graph" in sys.argv:
        # the event form of the step (fork / join by events: capturable) recorded into a HIP graph and
        # replayed, against the same form enqueued call by call
        form = K.STEP_SIDE_BOUNDARY | K.STEP_START_EVENT
        s = torch.cuda.Stream()
        res = {}
        with pkg.options({K.OPT_SLAB_STEP_FORM: form}):
            with torch.cuda.stream(s):
                res["eager_event_form_ms"] = round(run(lambda: comm.fill_step(prm, grid, slab, dist=dist, stream=s), steps), 4)
                slab.tex0[0].fill_(-1.0)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                        comm.fill_step(prm, grid, slab, dist=dist, stream=s)
                    res["graph_ms"] = round(run(g.replay, steps), 4)
                    res["graph_ghosts_ok"] = ghosts_ok() and bool(torch.equal(dist, slab.tex0[..., 0]))
                except Exception as e:  # noqa: BLE001
                    res["graph_error"] = f"{type(e).__name__}: {e}"[:400]
        res["eager_default_form_ms"] = round(run(step_fused, steps), 4)
        res["fused_fill_alone_ms"] = round(run(plain_fused, steps), 4)
        out["graph"] = res
        comm.close()
        print(json.dumps(out), flush=True)
        return
    if "--forms" in sys.argv:
        forms = {"side_boundary": K.STEP_SIDE_BOUNDARY, "side_boundary_event": K.STEP_SIDE_BOUNDARY | K.STEP_START_EVENT,
                 "side_boundary_unpacked": K.STEP_SIDE_BOUNDARY | K.STEP_UNPACKED}
        res = {k: [] for k in forms}
        res["plain_fill"] = []
        for rnd in range(3):
            for name, form in forms.items():
                slab.tex0[0].fill_(-1.0)
                slab.tex1[-1].fill_(-1.0)
                with pkg.options({K.OPT_SLAB_STEP_FORM: form}):
                    ms = run(step, steps, warm_s=0.1)
                res[name].append(round(ms, 4))
                if not ghosts_ok():
                    res[name].append("GHOSTS WRONG")
            res["plain_fill"].append(round(run(plain, steps, warm_s=0.1), 4))
        out["forms_ms"] = res
        best_plain = min(res["plain_fill"])
        out["fraction_of_plain_fill_rate"] = {k: round(best_plain / min(x for x in v if not isinstance(x, str)), 3)
                                              for k, v in res.items() if k != "plain_fill"}
    else:
        step_ms = run(step_fused, steps)
        same = ghosts_ok() and bool(torch.equal(dist, slab.tex0[..., 0]))
        fill_ms = run(plain_fused, steps)
        p_step_ms = run(step, steps)
        p_fill_ms = run(plain, steps)
        # the same fused steps with SDFV_STEP_DEFER_JOIN: no per-step wait of the caller's stream for the exchange, one
        # sdfv_slab_comm_join behind the K steps (a streaming caller joins where it reads the ghosts, not after every fill)
        slab.tex0[0].fill_(-1.0)
        with pkg.options({K.OPT_SLAB_STEP_FORM: K.STEP_DEFER_JOIN}):
            d_step_ms = run(step_fused, steps, after=comm.join)
        d_same = ghosts_ok() and bool(torch.equal(dist, slab.tex0[..., 0]))
        out.update({"ms_per_step": round(step_ms, 4), "plain_fill_ms": round(fill_ms, 4),
                    "fraction_of_plain_fill_rate": round(fill_ms / step_ms, 3), "ghosts_verified": same,
                    "what": "fused step (sdfv_slab_fill_step_commit, 36 B/voxel) against the fused fill alone",
                    "deferred_join": {"ms_per_step": round(d_step_ms, 4), "fraction_of_plain_fill_rate": round(fill_ms / d_step_ms, 3),
                                      "ghosts_verified": d_same,
                                      "what": "SDFV_STEP_DEFER_JOIN: K steps, ONE sdfv_slab_comm_join behind them"},
                    "unfused": {"ms_per_step": round(p_step_ms, 4), "plain_fill_ms": round(p_fill_ms, 4),
                                "fraction_of_plain_fill_rate": round(p_fill_ms / p_step_ms, 3)},
                    "form": "plain dense fill on the caller's stream; boundary slices -> packed buffers, RCCL, ghost copy "
                            "on the communicator's stream",
                    "note": "sdfv_slab_fill_step with the rank as its own neighbour (periodic world of 1), in its own "
                            "process: the step bench.py --gpus N times per rank, minus the xGMI transfer"})
    comm.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
