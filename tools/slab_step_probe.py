#!/usr/bin/env python3
"""One rank's multi-GPU fill step, measured on ONE GPU in a process of its own (bench.py runs this at N = 1 and
attaches the result as "halo_loopback"): a periodic communicator of world size 1 makes the rank its own z-neighbour, so
boundary fills, the RCCL group (4 sends + 4 receives of one slice) on its own stream, the interior fill and both stream
dependencies all run; only the xGMI transfer is missing.  The communicator is created before the first kernel, as the
process group is at N > 1 (a stream whose hardware queue predates the process's first RCCL communicator runs the
step's fill / record / wait pattern 2.5x slower -- DESIGN.md 6).  Prints one JSON line.
Usage: python tools/slab_step_probe.py <side> <steps>"""
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
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    pkg = importlib.import_module("sdf-viewer_amd")
    par = importlib.import_module("sdf-viewer_amd.parallel")
    torch.cuda.set_device(0)
    comm = par.SlabComm(pkg, 0, 1, periodic=True)  # first: see the docstring
    prm = pkg.default_params()
    dims = (side, side, side)
    grid = pkg.make_grid(dims)
    slab = par.alloc_slab(dims, 0, 1, "cuda", periodic=True, pkg=pkg)

    def run(fn, n):
        t_end = time.perf_counter() + 0.25  # device clock ramp, like bench.py's pre-warm
        while time.perf_counter() < t_end:
            fn()
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    step_ms = run(lambda: comm.fill_step(prm, grid, slab), steps)
    same = torch.equal(slab.tex0[0], slab.owned0[-1]) and torch.equal(slab.tex1[-1], slab.owned1[0])
    fill_ms = run(lambda: pkg.fill_grid(prm, grid, slab.owned0, slab.owned1), steps)
    comm.close()
    print(json.dumps({"ms_per_step": round(step_ms, 4), "plain_fill_ms": round(fill_ms, 4),
                      "fraction_of_plain_fill_rate": round(fill_ms / step_ms, 3), "ghosts_verified": bool(same),
                      "steps": steps,
                      "note": "sdfv_slab_fill_step with the rank as its own neighbour (periodic world of 1), in its own "
                              "process: the step bench.py --gpus N times per rank, minus the xGMI transfer"}), flush=True)


if __name__ == "__main__":
    main()
