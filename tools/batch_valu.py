#!/usr/bin/env python3
"""VALU wave-instructions of ONE 64-camera batch (BASELINE config 5's shape on one GPU) -> profiles/raymarch_batch_valu.json,
which bench.py turns into `batch_raymarch.roofline_raymarch_batch` (the batch is issue-bound: DESIGN.md 3.3).

  run     python tools/batch_valu.py run [side]           the workload: fused fill, then REPS batches (under rocprofv3 --pmc)
  reduce  python tools/batch_valu.py reduce <dir> [side]  sums SQ_INSTS_VALU / SQ_WAVES / SQ_BUSY_CYCLES over the raymarch
                                                          dispatches of the counter CSV under <dir>, / REPS

GPU box:
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \\
      -d gpurun_out/batch_valu -o pmc --output-format csv -- python tools/batch_valu.py run
  python tools/batch_valu.py reduce gpurun_out/batch_valu"""
import csv
import glob
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 3
IMAGES = {256: (1920, 1080), 512: (3840, 2160)}


def run(side):
    sys.path.insert(0, ROOT)
    import torch
    pkg = importlib.import_module("sdf-viewer_amd")
    W, H = IMAGES[side]
    prm = pkg.default_params()
    g = pkg.make_grid((side,) * 3)
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    rp = pkg.default_render_params(g)
    cams = pkg.orbit_cameras(64, aspect=W / H)
    out = torch.empty((64, H, W, 4), dtype=torch.float32, device="cuda")
    pairs = pkg.commit_pairs(g, dist)  # what bench.py's batch marches over
    for _ in range(REPS):
        pkg.raymarch(rp, t0, t1, cams, W, H, out=out, dist=dist, pairs=pairs)
    torch.cuda.synchronize()


def reduce(root, side):
    tot, n = {}, 0
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "raymarch_kernel" not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            n += row["Counter_Name"] == "SQ_WAVES"
    if not tot:
        raise SystemExit(f"no raymarch_kernel rows under {root}")
    per = {k: v / REPS for k, v in tot.items()}
    path = os.path.join(ROOT, "profiles", "raymarch_batch_valu.json")
    try:
        out = json.load(open(path))
    except Exception:  # noqa: BLE001
        out = {}
    out[str(side)] = {"valu_wave_instructions_per_batch": int(per["SQ_INSTS_VALU"]), "waves_per_batch": int(per.get("SQ_WAVES", 0)),
                      "salu_wave_instructions_per_batch": int(per.get("SQ_INSTS_SALU", 0)),
                      "vmem_read_wave_instructions_per_batch": int(per.get("SQ_INSTS_VMEM_RD", 0)),
                      "launches_per_batch": n // REPS, "cameras": 64, "image": list(IMAGES[side]), "clock_GHz": 2.4,
                      "source": f"rocprofv3 --pmc SQ_INSTS_VALU ... -- python tools/batch_valu.py run {side} ({REPS} batches over the y-pair volume averaged)",
                      "note": "SQ_INSTS_VALU counts wave-level instructions; clock = MI355X peak engine clock (MI355X_MICROARCH.md)"}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out[str(side)]))


if __name__ == "__main__":
    side = int(sys.argv[3 if sys.argv[1] == "reduce" else 2]) if len(sys.argv) > (3 if sys.argv[1] == "reduce" else 2) else 256
    run(side) if sys.argv[1] == "run" else reduce(sys.argv[2], side)
