#!/usr/bin/env python3
"""Writes the three files of tools/ref_golden/golden_gen.rs FROM THE ORACLE, in the generator's exact schema.

NOT reference output and never to be placed in tests/golden/ (tests/ref_golden.py refuses files whose generator says
"emulated"): it exists so that the consumer code -- the tests that will read the real files -- is itself exercised on CPU
(tests/test_ref_golden.py writes into a temporary directory and reads back), and as an executable description of the schema.

    python tools/ref_golden/emulate.py <out_dir>
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GENERATOR = "emulated from oracle/ by tools/ref_golden/emulate.py -- NOT reference output"

CONFIGS = [  # golden_gen.rs CONFIGS
    dict(flags=[], params=dict(cube_half_side=0.95, cube_material=0, sphere_radius=1.05, sphere_material=1,
                               max_distance_custom_material=0.05, disable_sphere=0), seeded_points=4096),
    dict(flags=["--cube-material", "normal", "--sphere-material", "brick"],
         params=dict(cube_half_side=0.95, cube_material=1, sphere_radius=1.05, sphere_material=0,
                     max_distance_custom_material=0.05, disable_sphere=0), seeded_points=1024),
    dict(flags=["--disable-sphere", "true"],
         params=dict(cube_half_side=0.95, cube_material=0, sphere_radius=1.05, sphere_material=1,
                     max_distance_custom_material=0.05, disable_sphere=1), seeded_points=512),
    dict(flags=["--cube-half-side", "0.8", "--sphere-radius", "0.3", "--max-distance-custom-material", "0.25"],
         params=dict(cube_half_side=0.8, cube_material=0, sphere_radius=0.3, sphere_material=1,
                     max_distance_custom_material=0.25, disable_sphere=0), seeded_points=512),
]
FIXED_POINTS = [[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.9, 0.6, 0.0], [0.96, 0.3, -0.2], [-1.0, 0.96825397, 0.015873075],
                [-0.015873015, 0.015873075, 1.0], [0.95, 0.95, 0.95], [-0.95, 0.2, 0.95], [0.0, 1.05, 0.0], [0.0, -1.0, 0.0],
                [-0.0, 0.0, -0.0], [0.97, -0.97, 0.5]]


def hx(v):
    return "%08x" % int(np.float32(v).view(np.uint32))


def hx_list(vs):
    return [hx(v) for v in vs]


def points(n):
    """golden_gen.rs points(): the fixed points, then a xorshift32 stream over [-1.25, 1.25)^3 (float32 arithmetic)."""
    p = [list(map(np.float32, q)) for q in FIXED_POINTS]
    s = 0x9E3779B9

    def nxt():
        nonlocal s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        return np.float32(np.float32(np.float32(s >> 8) * np.float32(1.0 / 16777216.0)) * np.float32(2.5)) - np.float32(1.25)
    for _ in range(n):
        p.append([nxt(), nxt(), nxt()])
    return np.array(p, np.float32)


def write_all(out_dir, oracle):
    os.makedirs(out_dir, exist_ok=True)
    samples = {"generator": GENERATOR, "crate_version": "1.6.6", "configs": []}
    for cfg in CONFIGS:
        prm = oracle.default_params(**cfg["params"])
        pts = points(cfg["seeded_points"])
        params = {k: (v if isinstance(v, int) else hx(v)) for k, v in cfg["params"].items()}
        ids = {}
        for sdf_id in (0, 1, 2):
            ids[str(sdf_id)] = {
                "bounding_box": hx_list([-1, -1, -1, 1, 1, 1]),
                "sample": [hx_list(r) for r in oracle.sample_many(prm, pts, False, sdf_id)],
                "sample_distance_only": [hx_list(r) for r in oracle.sample_many(prm, pts, True, sdf_id)],
                "normal": [hx_list(r) for r in oracle.normal_many(prm, pts, 0.0, sdf_id)],
            }
        samples["configs"].append({"flags": cfg["flags"], "params": params, "points": [hx_list(p) for p in pts], "ids": ids})
    with open(os.path.join(out_dir, "ref_samples.json"), "w") as f:
        json.dump(samples, f, separators=(",", ":"))

    cases = []
    for k in range(256):
        c = np.float32(k) / np.float32(255.0)
        cases += [c, (c.view(np.uint32) + np.uint32(1)).view(np.float32)]
        if k > 0:
            cases.append((c.view(np.uint32) - np.uint32(1)).view(np.float32))
        cases += [np.float32(np.float32(k) + np.float32(0.4)) / np.float32(255.0),
                  np.float32(np.float32(k) + np.float32(0.6)) / np.float32(255.0)]
    cases += [np.float32(x) for x in (-0.0, -0.25, 1.5, 256.0, np.inf, -np.inf, np.nan, 0.5, 0.6, 0.7)]
    srgb = {"generator": GENERATOR, "cases": []}
    for c in cases:
        u8 = int(oracle.L.or_srgb_quantize(float(c)))
        srgb["cases"].append({"c": hx(c), "u8": u8, "linear": hx(oracle.L.or_srgb_u8_to_linear(u8))})
    with open(os.path.join(out_dir, "ref_srgb.json"), "w") as f:
        json.dump(srgb, f, separators=(",", ":"))

    dims = (9, 7, 5)
    grid = {"generator": GENERATOR, "dims": list(dims), "loading_passes": 2, "air_dist": hx(oracle.AIR_DIST), "grids": []}
    for ci in (0, 1):
        prm = oracle.default_params(**CONFIGS[ci]["params"])
        t0, t1 = oracle.grid_init(dims)
        lm = oracle.lm_new(dims, 2)
        it = oracle.viewer_update(prm, dims, lm, t0, t1)
        entry = {"config": ci, "iterations": int(it), "passes_left": 0,
                 "tex0": [hx_list(t) for t in t0.reshape(-1, 4)], "tex1": [hx_list(t) for t in t1.reshape(-1, 4)]}
        if ci == 0:
            # SDFViewer::update after set_parameter: changed() = the bounding box -> a 3-pass manager with the box; the call
            # after it starts one more 3-pass manager without a box, which finds nothing to do (scene/sdf/mod.rs:131-156)
            edited = oracle.default_params(**dict(CONFIGS[0]["params"], max_distance_custom_material=float(np.float32(0.1))))
            lm = oracle.lm_new(dims, 3)
            it = oracle.viewer_update(edited, dims, lm, t0, t1, changed_box=(-1, -1, -1, 1, 1, 1))
            lm = oracle.lm_new(dims, 3)
            it += oracle.viewer_update(edited, dims, lm, t0, t1)
            entry["edit"] = {"max_distance_custom_material": hx(0.1), "iterations": int(it),
                             "tex0": [hx_list(t) for t in t0.reshape(-1, 4)], "tex1": [hx_list(t) for t in t1.reshape(-1, 4)]}
        grid["grids"].append(entry)
    with open(os.path.join(out_dir, "ref_grid_9x7x5.json"), "w") as f:
        json.dump(grid, f, separators=(",", ":"))


if __name__ == "__main__":
    import oracle_binding
    write_all(sys.argv[1], oracle_binding)
    print("wrote emulated ref_*.json to", sys.argv[1], "(NOT reference output)")
