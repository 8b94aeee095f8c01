#!/usr/bin/env python3
"""SURVEY.md 8(d) "Algorithmic bytes -- raymarch": the two modelled figures, from the oracle's deterministic counts.

    compulsory HBM bytes  = 16 B x (unique tex0 texels touched + unique tex1 texels touched) + 16 B x W*H output
    nominal gather bytes  = 128 B x (sum of march steps + 5 x hits) + 16 B x W*H
                            (8 texels x 16 B per trilinear fetch; 1 tex1 + 4 normal fetches per hit)

plus the same two figures for the path the product actually runs (march over the compact 4-byte distance volume,
sdfNormal not evaluated under the scene's ambient-only lighting: DESIGN.md 3.3) and the compulsory bytes at the
granularity HBM is read at (unique 128-byte lines).  CPU only (oracle); writes profiles/raymarch_model_bytes.json,
which bench.py attaches to `roofline_raymarch`.

Usage: python tools/raymarch_bytes.py [--skip-512] [--skip-batch]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as oracle  # noqa: E402

LINE = 128  # bytes per L2 / HBM request line on gfx950


def lines_touched(mask, bytes_per_texel):
    """Unique LINE-byte lines of a [D, H, W] array of `bytes_per_texel` elements covered by the touched texels."""
    per_line = LINE // bytes_per_texel
    flat = mask.reshape(-1)
    pad = (-flat.size) % per_line
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, np.uint8)])
    return int(flat.reshape(-1, per_line).max(axis=1).sum())


def orbit_eyes(n, eye0=(2.5, 3.0, 5.0)):
    r, a0 = np.hypot(eye0[0], eye0[2]), np.arctan2(eye0[2], eye0[0])
    return [(float(r * np.cos(a0 + 2 * np.pi * k / n)), eye0[1], float(r * np.sin(a0 + 2 * np.pi * k / n))) for k in range(n)]


def model(side, W, H, eyes, threads):
    dims = (side, side, side)
    prm = oracle.default_params()
    t0, t1 = oracle.fill_dense(prm, dims, threads=threads)
    rp = oracle.default_render_params(dims)
    maps, tot = None, None
    per_camera = []
    for eye in eyes:
        cam = oracle.camera_look_at(eye=eye, aspect=W / H)
        maps, c = oracle.raymarch_touch(rp, t0, t1, cam, W, H, threads=threads, maps=maps)
        if len(eyes) > 1:
            # the same camera on its own (bench.py --gpus N marches ONE orbit camera per rank: N > 1 lines take their
            # byte model from these entries instead of reporting none)
            own, _ = oracle.raymarch_touch(rp, t0, t1, cam, W, H, threads=threads, maps=None)
            m, h0, h1 = int(own["march0"].sum()), int(own["hit0"].sum()), int(own["hit1"].sum())
            all0 = int((own["march0"] | own["hit0"] | own["normal0"]).sum())
            per_camera.append({"eye": [round(e, 6) for e in eye], "hits": c["hits"], "sum_steps": c["sum_steps"],
                               "covered": c["covered"], "max_steps": c["max_steps"],
                               "compulsory_bytes": 16 * (all0 + h1) + 16 * W * H,
                               "nominal_gather_bytes": 128 * (c["sum_steps"] + 5 * c["hits"]) + 16 * W * H,
                               "product_path_compulsory_bytes": 4 * m + 16 * (h0 + h1) + 16 * W * H})
        if tot is None:
            tot = c
        else:
            tot = {k: (max(tot[k], c[k]) if k == "max_steps" else tot[k] + c[k]) for k in c}
    n_cam = len(eyes)
    out_bytes = 16 * W * H * n_cam
    u_march = int(maps["march0"].sum())
    u_hit0, u_hit1, u_norm = int(maps["hit0"].sum()), int(maps["hit1"].sum()), int(maps["normal0"].sum())
    u_tex0_all = int((maps["march0"] | maps["hit0"] | maps["normal0"]).sum())          # the shader as written
    u_tex0_no_normal = int((maps["march0"] | maps["hit0"]).sum())
    res = {
        "grid": side, "image": [W, H], "cameras": n_cam, "counts": tot,
        "unique_texels": {"tex0_march": u_march, "tex0_hit": u_hit0, "tex1_hit": u_hit1, "tex0_normal_taps": u_norm,
                          "tex0_all": u_tex0_all, "of_grid": side ** 3},
        # SURVEY 8(d), as defined there (shader as written: normal taps included, 16-byte texels everywhere)
        "compulsory_bytes": 16 * (u_tex0_all + u_hit1) + out_bytes,
        "nominal_gather_bytes": 128 * (tot["sum_steps"] + 5 * tot["hits"]) + out_bytes,
        # the product's path: march reads the 4-byte distance volume, full texels only under the hits, no normal
        "product_path": {
            "compulsory_bytes": 4 * u_march + 16 * (u_hit0 + u_hit1) + out_bytes,
            "compulsory_line_bytes": LINE * (lines_touched(maps["march0"], 4) + lines_touched(maps["hit0"], 16)
                                             + lines_touched(maps["hit1"], 16)) + out_bytes,
            "nominal_gather_bytes": 32 * tot["sum_steps"] + 2 * 128 * tot["hits"] + out_bytes,
            "note": "march over the compact distance volume (8 x 4 B per fetch), 8 x 16 B of tex0 and of tex1 per hit, "
                    "sdfNormal dead under ambient-only lighting; *_line_bytes counts unique 128-byte lines",
        },
        "tex0_path": {  # sdfv_raymarch without the distance volume: march gathers tex0.r out of 16-byte texels
            "compulsory_bytes": 16 * (u_tex0_no_normal + u_hit1) + out_bytes,
            "compulsory_line_bytes": LINE * (lines_touched(maps["march0"] | maps["hit0"], 16)
                                             + lines_touched(maps["hit1"], 16)) + out_bytes,
            "nominal_gather_bytes": 128 * (tot["sum_steps"] + 1 * tot["hits"]) + out_bytes,
        },
        "output_bytes": out_bytes,
    }
    if per_camera:
        res["per_camera"] = per_camera
    return res


def main():
    threads = max(1, min(os.cpu_count() or 1, 16))
    out = {"source": "oracle/liboracle.so (or_raymarch_touch), tools/raymarch_bytes.py", "line_bytes": LINE,
           "definition": "SURVEY.md 8(d): compulsory = 16 B x (unique tex0 + unique tex1 texels) + 16 B x W*H; "
                         "nominal = 128 B x (sum steps + 5 x hits) + 16 B x W*H"}
    t = time.time()
    out["256"] = model(256, 1920, 1080, [(2.5, 3.0, 5.0)], threads)
    print("256/1080p", round(time.time() - t, 1), "s", json.dumps(out["256"]["counts"]), flush=True)
    if "--skip-batch" not in sys.argv:
        t = time.time()
        out["256_batch64"] = model(256, 1920, 1080, orbit_eyes(64), threads)
        print("256/64x1080p", round(time.time() - t, 1), "s", flush=True)
    if "--skip-512" not in sys.argv:
        t = time.time()
        out["512"] = model(512, 3840, 2160, [(2.5, 3.0, 5.0)], threads)
        print("512/4K", round(time.time() - t, 1), "s", json.dumps(out["512"]["counts"]), flush=True)
    path = os.path.join(ROOT, "profiles", "raymarch_model_bytes.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
