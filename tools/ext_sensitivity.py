#!/usr/bin/env python3
"""How far do the outputs move if one of the [EXT] assumptions is wrong?  (VERDICT r01, weak #1 / next #3.)

Four pieces of arithmetic on the path live in crates / a GL driver that are not under /root/reference and are restated
from their published behaviour (oracle/sdf_oracle.h).  This tool re-runs the ORACLE at configs[1] (256^3 grid, 1920x1080,
default camera) with each assumption swapped for its plausible alternative, ONE AT A TIME, and records against the
baseline restatement: words changed in tex0 / tex1, hit/miss flips, step-count changes, max |dRGBA| -- i.e. how far the
1e-4 RGBA gate and the bit-exact texture gate are from each assumption.  CPU only.
Writes profiles/ext_sensitivity.json.   Usage: python tools/ext_sensitivity.py [side=256]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as oracle  # noqa: E402

WHAT = {
    "srgb_pow_ulp_up": "Srgba::to_linear_srgb: powf result one ulp UP in every table entry of the pow branch (libm not correctly rounded) -- scene/sdf/mod.rs:201 [three-d-asset 0.9.2]",
    "srgb_pow_ulp_down": "Srgba::to_linear_srgb: powf result one ulp DOWN",
    "srgb_double_pow": "Srgba::to_linear_srgb evaluated in f64 and rounded once",
    "srgb_quant_round": "Srgba::from(Vec3): round(c*255) instead of the truncating `as u8`",
    "cgmath_normalize_div": "cgmath normalize as v / |v| instead of v * (1/|v|) -- sphere.rs:123, defaults.rs:55",
    "glsl_mix_lerp": "GLSL mix(a,b,t) as a + t*(b-a) -- trilinear filter and colour mapping, material.frag:19-23,163-168",
    "glsl_normalize_rsq": "GLSL normalize as v * (1/length) -- material.frag:79,134",
    "trilinear_weighted_sum": "GL LINEAR filter as the spec's 8-term weighted sum instead of nested mix x,y,z (driver-defined)",
}


def run(side, W, H, threads):
    dims = (side, side, side)
    prm = oracle.default_params()
    t0, t1 = oracle.fill_dense(prm, dims, threads=threads)
    rp = oracle.default_render_params(dims)
    cam = oracle.camera_look_at(aspect=W / H)
    rgba, aux = oracle.raymarch(rp, t0, t1, cam, W, H, threads=threads)
    return t0, t1, rgba, aux


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    W, H = 1920, 1080
    threads = max(1, min(os.cpu_count() or 1, 16))
    oracle.L.or_set_ext_variant(0)
    b0, b1, brgba, baux = run(side, W, H, threads)
    out = {"config": {"grid": side, "image": [W, H], "camera": "reference default (scene/mod.rs:82-95)"},
           "gates": {"textures": "bit-exact", "rgba": 1e-4},
           "baseline": {"hits": int((baux["status"] == 1).sum()), "covered": int((baux["status"] != 0).sum()),
                        "texture_words": int(b0.size + b1.size)},
           "variants": {}}
    for name, flag in oracle.EXT_VARIANTS.items():
        t = time.time()
        oracle.L.or_set_ext_variant(flag)
        try:
            v0, v1, vrgba, vaux = run(side, W, H, threads)
        finally:
            oracle.L.or_set_ext_variant(0)
        d0 = v0.view(np.uint32) != b0.view(np.uint32)
        d1 = v1.view(np.uint32) != b1.view(np.uint32)
        both_hit = (vaux["status"] == 1) & (baux["status"] == 1)
        drgba = np.abs(vrgba - brgba)
        rec = {
            "what": WHAT[name],
            "tex0_words_changed": int(d0.sum()), "tex1_words_changed": int(d1.sum()),
            "tex0_max_abs_diff": float(np.abs(v0 - b0).max()), "tex0_voxels_changed": int(d0.any(axis=-1).sum()),
            "tex0_distance_channel_changed": int(d0[..., 0].sum()),
            "hit_miss_flips": int(((vaux["status"] == 1) != (baux["status"] == 1)).sum()),
            "status_changes": int((vaux["status"] != baux["status"]).sum()),
            "step_count_changes": int((vaux["steps"] != baux["steps"]).sum()),
            "hit_pos_words_changed": int((vaux["hit_pos"].view(np.uint32) != baux["hit_pos"].view(np.uint32)).any(axis=-1).sum()),
            "max_abs_drgba": float(drgba.max()),
            "max_abs_drgba_where_both_hit": float(drgba[both_hit].max()) if both_hit.any() else 0.0,
            "pixels_over_1e-4": int((drgba.max(axis=-1) > 1e-4).sum()),
            "seconds": round(time.time() - t, 1),
        }
        rec["verdict"] = ("textures identical; " if rec["tex0_words_changed"] + rec["tex1_words_changed"] == 0 else
                          f"{rec['tex0_words_changed'] + rec['tex1_words_changed']} texture words differ; ") + \
                         ("RGBA within the 1e-4 gate" if rec["pixels_over_1e-4"] == 0 else
                          f"{rec['pixels_over_1e-4']} pixels beyond the 1e-4 gate (max {rec['max_abs_drgba']:.3g})")
        out["variants"][name] = rec
        print(name, json.dumps({k: v for k, v in rec.items() if k != "what"}), flush=True)
    path = os.path.join(ROOT, "profiles", "ext_sensitivity.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
