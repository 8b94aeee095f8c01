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
     mesh_front_200.npz  -- the same restatement behind the mesher front end (src/sdf/meshers/isosurface.rs:78-99,
                            mesh.rs:22-33): ScalarSource, HermiteSource and Mesh::postproc on 200 seeded points.
  3. raymarch_12cube_40x30.npz -- the same kind of restatement of material.frag (sphere tracing, texel-centre
                            trilinear, ambient shading, ACES, sRGB) for two cameras over a 12^3 grid.

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


def normal(prm, sdf_id, p):  # SDFSurface::normal(p, None): demo/mod.rs:147-156, cube.rs:164-177, sphere.rs:122-124
    def cube_n():
        h = F(prm["cube_half_side"])
        return tuple((F(-1.0) if np.signbit(c) else F(1.0)) if abs(c) > h else F(0) for c in p)

    def sphere_n():
        with np.errstate(all="ignore"):
            inv = F(1.0) / np.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
            return (p[0] * inv, p[1] * inv, p[2] * inv)
    if sdf_id == 1:
        return cube_n()
    if sdf_id == 2:
        return sphere_n()
    if abs(cube(prm, p, True)[0]) < abs(sphere(prm, p, True)[0]):
        return cube_n()
    return tuple(-c for c in sphere_n())


def vert_pos_to(bb_min, bb_max, p):  # meshers/isosurface.rs:95-99
    return tuple(F(p[i]) * (F(bb_max[i]) - F(bb_min[i])) + F(bb_min[i]) for i in range(3))


def postproc(prm, sdf_id, v):  # Mesh::postproc, meshers/mesh.rs:22-33; v = 12 floats
    v = [F(c) for c in v]
    pos = tuple(v[0:3])
    s = sample(prm, sdf_id, pos, False)
    if v[3] * v[3] + v[4] * v[4] + v[5] * v[5] < F(0.0001):
        v[3:6] = normal(prm, sdf_id, pos)
    v[6:12] = s[1:7]
    return v


def make_mesh_front_fixture():
    rng = np.random.default_rng(20250405)
    unit = rng.uniform(0.0, 1.0, size=(200, 3)).astype(np.float32)
    unit[:4] = [(0, 0, 0), (1, 1, 1), (0.5, 0.5, 0.5), (0.25, 1.0, 0.0)]
    bb_min, bb_max = (-1.0, -0.75, -1.25), (1.0, 1.0, 0.5)
    verts = np.zeros((200, 12), np.float32)
    verts[:, 0:3] = rng.uniform(-1.1, 1.1, size=(200, 3)).astype(np.float32)
    verts[:70, 3:6] = rng.normal(size=(70, 3)).astype(np.float32)
    verts[140:, 3] = np.float32(0.01) * (1 + rng.uniform(-1e-3, 1e-3, size=60)).astype(np.float32)
    out = {}
    with np.errstate(all="ignore"):
        for k, prm in enumerate(PARAM_SETS):
            for sdf_id in (0, 1, 2):
                world = [vert_pos_to(bb_min, bb_max, p) for p in unit]
                out[f"scalar_{k}_{sdf_id}"] = np.array([sample(prm, sdf_id, w, True)[0] for w in world], np.float32)
                out[f"normal_{k}_{sdf_id}"] = np.array([normal(prm, sdf_id, w) for w in world], np.float32)
                out[f"postproc_{k}_{sdf_id}"] = np.array([postproc(prm, sdf_id, v) for v in verts], np.float32)
    np.savez_compressed(os.path.join(HERE, "mesh_front_200.npz"), unit_points=unit, bb_min=np.array(bb_min, np.float32),
                        bb_max=np.array(bb_max, np.float32), vertices=verts,
                        params=np.array([[p[k] for k in DEFAULT] for p in PARAM_SETS], np.float64), **out)


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


# ---------------------------------------------------------------------------------------------------------
# Independent numpy-float32 restatement of the fragment shader (src/app/scene/sdf/material.frag) + the
# three-d pieces it calls, for a second opinion on oracle/raymarch.c.  Restatement choices are the ones
# documented in oracle/raymarch.c (slab-test fragment, texel-centre trilinear with MirroredRepeat, mix() along
# x then y then z, normalize = v / length, ambient-only lighting, ACES, linear->sRGB).
# ---------------------------------------------------------------------------------------------------------
def v3(*a):
    return [F(x) for x in a]


def vlen(a):
    return np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def vnorm(a):
    l = vlen(a)
    return [a[0] / l, a[1] / l, a[2] / l]


def mixf(a, b, t):
    return a * (F(1.0) - t) + b * t


def look_at(eye, target, up, fovy_deg, aspect, near, far):
    e, t, u0 = v3(*eye), v3(*target), v3(*up)
    f = [t[i] - e[i] for i in range(3)]
    fl = F(1.0) / vlen(f)
    f = [c * fl for c in f]
    s = [f[1] * u0[2] - f[2] * u0[1], f[2] * u0[0] - f[0] * u0[2], f[0] * u0[1] - f[1] * u0[0]]
    sl = F(1.0) / vlen(s)
    s = [c * sl for c in s]
    u = [s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]]
    half = F(fovy_deg) * (F(3.14159265358979323846) / F(180.0)) / F(2.0)
    tan_half = F(np.tan(half))
    # bvp is not consumed by the fixture (depth is not compared); zeros keep the POD layout
    return dict(eye=e, right=s, up=u, forward=f, tan_half_fovy=tan_half, aspect=F(aspect), bvp=[F(0)] * 16)


def mirror(i, n):
    m = i % (2 * n)
    return m if m < n else 2 * n - 1 - m


def tex_linear(tex, p01):
    d, h, w = tex.shape[:3]
    u = p01[0] * F(w) - F(0.5)
    v = p01[1] * F(h) - F(0.5)
    ww = p01[2] * F(d) - F(0.5)
    fu, fv, fw = np.floor(u), np.floor(v), np.floor(ww)
    ax, ay, az = u - fu, v - fv, ww - fw
    i0, j0, k0 = int(fu), int(fv), int(fw)
    t = lambda i, j, k: tex[mirror(k, d), mirror(j, h), mirror(i, w)]
    out = []
    for c in range(4):
        c00 = mixf(t(i0, j0, k0)[c], t(i0 + 1, j0, k0)[c], ax)
        c10 = mixf(t(i0, j0 + 1, k0)[c], t(i0 + 1, j0 + 1, k0)[c], ax)
        c01 = mixf(t(i0, j0, k0 + 1)[c], t(i0 + 1, j0, k0 + 1)[c], ax)
        c11 = mixf(t(i0, j0 + 1, k0 + 1)[c], t(i0 + 1, j0 + 1, k0 + 1)[c], ax)
        out.append(mixf(mixf(c00, c10, ay), mixf(c01, c11, ay), az))
    return out


def oob(p, bmin, bmax):
    o = [max(bmin[i] - p[i], p[i] - bmax[i]) for i in range(3)]
    return max(o[0], max(o[1], o[2]))


def shade_px(raw0, raw1):
    out = []
    for c in range(3):
        albedo = raw0[1 + c] * F(1.0)
        lit = raw1[2] * F(1.0) * mixf(albedo, F(0.0), raw1[0])
        lit = (lit * (F(2.51) * lit + F(0.03))) / (lit * (F(2.43) * lit + F(0.59)) + F(0.14))  # ACES
        lit = min(max(lit, F(0.0)), F(1.0))
        sel = F(1.0) if lit >= F(0.0031308) else F(0.0)
        lo = lit * F(12.92)
        hi = F(1.055) * F(np.power(np.float64(lit), np.float64(F(1.0) / F(2.4)))) - F(0.055)
        out.append(mixf(lo, hi, sel))
    return out + [F(1.0)]


def march_px(tex0, tex1, cam, bmin, bmax, W, H, px, py):
    rec = dict(status=0, steps=0, hit_pos=v3(0, 0, 0), rgba=v3(0, 0, 0) + [F(0)])
    ndc_x = ((F(px) + F(0.5)) / F(W)) * F(2.0) - F(1.0)
    ndc_y = F(1.0) - ((F(py) + F(0.5)) / F(H)) * F(2.0)
    sx = ndc_x * cam["aspect"] * cam["tan_half_fovy"]
    sy = ndc_y * cam["tan_half_fovy"]
    eye = cam["eye"]
    d0 = vnorm([cam["forward"][i] + cam["right"][i] * sx + cam["up"][i] * sy for i in range(3)])
    with np.errstate(all="ignore"):
        t1 = [(bmin[i] - eye[i]) / d0[i] for i in range(3)]
        t2 = [(bmax[i] - eye[i]) / d0[i] for i in range(3)]
    lo = [np.fmin(a, b) for a, b in zip(t1, t2)]
    hi = [np.fmax(a, b) for a, b in zip(t1, t2)]
    tnear = np.fmax(np.fmax(lo[0], lo[1]), lo[2])
    tfar = np.fmin(np.fmin(hi[0], hi[1]), hi[2])
    if not (tfar >= tnear and tfar > 0):
        return rec
    tfrag = tnear if tnear > 0 else tfar
    pos = [eye[i] + d0[i] * tfrag for i in range(3)]
    rd = vnorm([pos[i] - eye[i] for i in range(3)])
    ro = pos
    if oob([ro[i] + rd[i] * F(0.2) for i in range(3)], bmin, bmax) > 0:
        ro = [eye[i] + rd[i] * F(0.2) for i in range(3)]
    size = [bmax[i] - bmin[i] for i in range(3)]
    p = ro
    status, steps, raw0 = -1, 0, None
    for i in range(256):
        if i >= 255:
            status = -1
            break
        if oob(p, bmin, bmax) > F(1e-4):
            status = -2
            break
        s = tex_linear(tex0, [(p[c] - bmin[c]) / size[c] for c in range(3)])
        steps += 1
        d = s[0] - F(1e-1)
        if d < F(1e-5):
            status, raw0 = 1, s
            break
        p = [p[c] + rd[c] * d for c in range(3)]
    rec.update(status=status, steps=steps, hit_pos=p)
    if status == 1:
        raw1 = tex_linear(tex1, [(p[c] - bmin[c]) / size[c] for c in range(3)])
        rec["rgba"] = shade_px(raw0, raw1)
    return rec


def make_raymarch_fixture():
    n = 12
    bmin, bmax = v3(-1, -1, -1), v3(1, 1, 1)
    tex0 = np.zeros((n, n, n, 4), np.float32)
    tex1 = np.zeros_like(tex0)
    for z in range(n):
        for y in range(n):
            for x in range(n):
                p = (coord(x, n, -1, 1), coord(y, n, -1, 1), coord(z, n, -1, 1))
                a, b = pack(sample(DEFAULT, 0, p))
                tex0[z, y, x] = a
                tex1[z, y, x] = b
    W, H = 40, 30
    out = {}
    for k, eye in enumerate([(2.5, 3.0, 5.0), (0.2, 0.1, -0.3)]):  # the default camera, and one inside the box
        cam = look_at(eye, (0, 0, 0) if k == 0 else (1.0, 0.8, 0.9), (0, 1, 0), 45.0, W / H, 0.1, 1000.0)
        status = np.zeros((H, W), np.int32)
        steps = np.zeros((H, W), np.int32)
        hit_pos = np.zeros((H, W, 3), np.float32)
        rgba = np.zeros((H, W, 4), np.float32)
        for py in range(H):
            for px in range(W):
                r = march_px(tex0, tex1, cam, bmin, bmax, W, H, px, py)
                status[py, px], steps[py, px] = r["status"], r["steps"]
                hit_pos[py, px] = r["hit_pos"] if r["status"] != 0 else 0
                rgba[py, px] = r["rgba"]
        pod = np.array(cam["eye"] + cam["right"] + cam["up"] + cam["forward"] + [cam["tan_half_fovy"], cam["aspect"]] +
                       cam["bvp"], np.float32)
        out.update({f"cam_{k}": pod, f"status_{k}": status, f"steps_{k}": steps, f"hit_pos_{k}": hit_pos, f"rgba_{k}": rgba})
    np.savez_compressed(os.path.join(HERE, "raymarch_12cube_40x30.npz"), tex0=tex0, tex1=tex1, width=W, height=H, **out)


def main():
    make_raymarch_fixture()
    make_mesh_front_fixture()
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
