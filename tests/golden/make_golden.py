#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference (Rust + GLSL) cannot be built or imported in this image and its own tests hold no numeric
vectors for this path, so the fixtures come from two sources, both independent of the C
oracle they are used to pin:

  1. demo_sdf_kat.json   -- the known-answer vectors of SURVEY.md 8(c), derived by hand from the reference
                            source (src/sdf/demo/*.rs, src/app/scene/sdf/mod.rs:179-208), default params.
  2. grid_9x7x5.npz,     -- an op-by-op numpy-float32 restatement of the same source written below (every
     points_512.npz         operation on np.float32 scalars, so each one rounds to f32 like Rust's), run on a
                            small non-cubic grid and on 512 seeded points for several parameter sets.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))

AIR = F(1e-1) + F(0.001234)  # scene/sdf/mod.rs:42


def srgb_lut():
    out = np.zeros(256, np.float32)
    for i in range(256):
        c = F(i) / F(255.0)
        if c < F(0.04045):
            out[i] = c / F(12.92)
        else:
            b = (c + F(0.055)) / F(1.055)
            out[i] = F(np.power(np.float64(b), np.float64(F(2.4))))  # correctly rounded f32 powf
    return out


LUT = srgb_lut()


def brick(u, v):  # cube.rs:189-202
    row = v / F(0.25)
    off = F(np.floor(row)) / F(4.0)
    bx = F(np.fmod(abs(u + off), F(0.5)))
    by = F(np.fmod(abs(v), F(0.25)))
    m = F(0.2) / F(2.0) * F(0.25)
    if bx < m or bx > F(0.5) - m or by < m or by > F(0.25) - m:
        return (F(56.) / F(255.), F(70.) / F(255.), F(60.) / F(255.), F(0.4), F(0.5), F(1.0))
    return (F(150.) / F(255.), F(24.) / F(255.), F(10.) / F(255.), F(0.2), F(0.8), F(0.0))


def render(material, d, p, n):  # cube.rs:51-58, 205-220
    if material == 0:
        ax, ay, az = abs(n[0]), abs(n[1]), abs(n[2])
        if ax > ay:
            uv = (p[2], p[1]) if ax > az else (p[0], p[1])
        elif ay > az:
            uv = (p[2], p[0])
        else:
            uv = (p[0], p[1])
        return (d,) + brick(*uv)
    return (d, abs(n[0]), abs(n[1]), abs(n[2]), F(0), F(0), F(0))


def cube(prm, p, distance_only):  # cube.rs:79-89, 164-177
    h = F(prm["cube_half_side"])
    d = max(max(abs(p[0]), abs(p[1])), abs(p[2])) - h
    if distance_only or d > F(0.1):
        return (d, F(0), F(0), F(0), F(0), F(0), F(0))
    n = [F(0), F(0), F(0)]
    for i in range(3):
        if abs(p[i]) > h:
            n[i] = F(-1.0) if np.signbit(p[i]) else F(1.0)
    return render(prm["cube_material"], d, p, n)


def sphere(prm, p, distance_only):  # sphere.rs:37-47, 122-124
    with np.errstate(all="ignore"):
        ln = np.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
        d = ln - F(prm["sphere_radius"])
        if distance_only or d > F(0.1):
            return (d, F(0), F(0), F(0), F(0), F(0), F(0))
        inv = F(1.0) / ln
        n = (p[0] * inv, p[1] * inv, p[2] * inv)
    return render(prm["sphere_material"], d, p, n)


def demo(prm, p, distance_only=False):  # demo/mod.rs:51-75
    sb = cube(prm, p, distance_only)
    if prm["disable_sphere"]:
        return sb
    ss = sphere(prm, p, distance_only)
    dist = max(sb[0], -ss[0])
    inter = abs(sb[0]) - abs(ss[0])
    s = list(sb if inter < F(0) else ss)
    if abs(inter) <= F(prm["max_distance_custom_material"]):
        s[1:] = [F(0.5), F(0.6), F(0.7), F(0.5), F(0.0), F(0.0)]
    s[0] = dist
    return tuple(s)


def sample(prm, sdf_id, p, distance_only=False):
    return (demo, cube, sphere)[sdf_id](prm, p, distance_only)


def quant(c):  # (c * 255.0) as u8
    v = c * F(255.0)
    if not v > 0:
        return 0
    return 255 if v >= 255 else int(v)


def pack(s):  # scene/sdf/mod.rs:196-208
    d = F(1e-1) + s[0]
    d = F(0) if d < 0 else (F(1) if d > 1 else d)
    col = s[1:4]
    if col[0] == 0 and col[1] == 0 and col[2] == 0:
        col = (F(0.5), F(0.5), F(0.5))
    t0 = [d] + [LUT[quant(c)] for c in col]
    t1 = [s[4], s[5], F(1.0) if s[6] <= 0 else s[6], AIR]
    return t0, t1


def coord(i, n, lo, hi):  # scene/sdf/mod.rs:178-182
    size = F(hi) - F(lo)
    return F(i) / (F(n) - F(1.0)) * size + F(lo)


DEFAULT = dict(cube_half_side=0.95, cube_material=0, sphere_radius=1.05, sphere_material=1,
               max_distance_custom_material=0.05, disable_sphere=0)
PARAM_SETS = [
    DEFAULT,
    dict(DEFAULT, cube_material=1, sphere_material=0),
    dict(DEFAULT, disable_sphere=1),
    dict(DEFAULT, cube_half_side=0.5, sphere_radius=0.6, max_distance_custom_material=0.0),
    dict(DEFAULT, cube_half_side=0.8, sphere_radius=0.3, max_distance_custom_material=0.25),
]

# hand-derived from the reference source; default params.
SURVEY_KATS = [
    dict(p=[0, 0, 0], sample=[1.05, 56 / 255, 70 / 255, 60 / 255, 0.4, 0.5, 1.0],
         tex0=[1.0, 0.0395462, 0.0612461, 0.0451862], tex1=[0.4, 0.5, 1.0], u8=[56, 70, 60]),
    dict(p=[1, 1, 1], sample=[0.050000012, 56 / 255, 70 / 255, 60 / 255, 0.4, 0.5, 1.0],
         tex0=[0.15, 0.0395462, 0.0612461, 0.0451862], tex1=[0.4, 0.5, 1.0]),
    dict(p=[0.9, 0.6, 0], sample=[-0.031665444, 0.5, 0.6, 0.7, 0.5, 0.0, 0.0],
         tex0=[0.06833456, 0.2122307, 0.3185468, 0.4452012], tex1=[0.5, 0.0, 1.0], u8=[127, 153, 178]),
    dict(p=[0.96, 0.3, -0.2], sample=[0.02452445, 0.5, 0.6, 0.7, 0.5, 0.0, 0.0],
         tex0=[0.12452445, 0.2122307, 0.3185468, 0.4452012], tex1=[0.5, 0.0, 1.0]),
    dict(p=[-1, 0.96825397, 0.015873075], sample=[0.050000012, 56 / 255, 70 / 255, 60 / 255, 0.4, 0.5, 1.0],
         tex0=[0.15, 0.0395462, 0.0612461, 0.0451862], tex1=[0.4, 0.5, 1.0]),
]
SURVEY_COORDS_64 = {"0": -1.0, "1": -0.96825397, "31": -0.015873015, "32": 0.015873075, "63": 1.0}


def main():
    with open(os.path.join(HERE, "demo_sdf_kat.json"), "w") as f:
        json.dump(dict(source="SURVEY.md 8(c), hand-derived from the reference source; default demo params",
                       air_dist_bits="0x3DCF53C6", kats=SURVEY_KATS, coords_n64_bb_m1_1=SURVEY_COORDS_64), f, indent=1)

    dims = (9, 7, 5)
    bb_min, bb_max = (-1.0, -0.5, -1.25), (1.0, 1.0, 1.0)
    grids = {}
    for k, prm in enumerate(PARAM_SETS):
        t0 = np.zeros((dims[2], dims[1], dims[0], 4), np.float32)
        t1 = np.zeros_like(t0)
        for z in range(dims[2]):
            for y in range(dims[1]):
                for x in range(dims[0]):
                    p = (coord(x, dims[0], bb_min[0], bb_max[0]), coord(y, dims[1], bb_min[1], bb_max[1]),
                         coord(z, dims[2], bb_min[2], bb_max[2]))
                    a, b = pack(sample(prm, 0, p))
                    t0[z, y, x] = a
                    t1[z, y, x] = b
        grids[f"tex0_{k}"] = t0
        grids[f"tex1_{k}"] = t1
    np.savez_compressed(os.path.join(HERE, "grid_9x7x5.npz"), dims=np.array(dims), bb_min=np.array(bb_min, np.float32),
                        bb_max=np.array(bb_max, np.float32),
                        params=np.array([[p[k] for k in DEFAULT] for p in PARAM_SETS], np.float64), **grids)

    rng = np.random.default_rng(20250404)
    pts = rng.uniform(-1.3, 1.3, size=(512, 3)).astype(np.float32)
    pts[:16] = rng.uniform(-1e-3, 1e-3, size=(16, 3)).astype(np.float32)       # near the origin
    pts[16:48] = (pts[16:48] / np.linalg.norm(pts[16:48], axis=1, keepdims=True) * 1.05).astype(np.float32)  # sphere shell
    pts[48:80, 0] = np.float32(0.95)                                              # on a cube face
    pts[80:96] = rng.uniform(-40.0, 40.0, size=(16, 3)).astype(np.float32)        # far outside
    out = {}
    for k, prm in enumerate(PARAM_SETS):
        for sdf_id in (0, 1, 2):
            for do in (0, 1):
                out[f"s_{k}_{sdf_id}_{do}"] = np.array(
                    [sample(prm, sdf_id, tuple(F(c) for c in p), bool(do)) for p in pts], np.float32)
    np.savez_compressed(os.path.join(HERE, "points_512.npz"), points=pts,
                        params=np.array([[p[k] for k in DEFAULT] for p in PARAM_SETS], np.float64), **out)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
