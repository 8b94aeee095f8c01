"""CPU tests pinning the oracle (oracle/*.c) to everything available for this path:
  - the reference's own unit tests (LoadingManager, src/app/scene/sdf/loading.rs:117-171), restated;
  - the hand-derived known-answer vectors of SURVEY.md 8(c) (tests/golden/demo_sdf_kat.json);
  - an independent numpy-float32 restatement (tests/golden/make_golden.py -> *.npz).
The reference itself cannot run here, so parity with it stays "unpinned" beyond these."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_KEYS = ["cube_half_side", "cube_material", "sphere_radius", "sphere_material",
              "max_distance_custom_material", "disable_sphere"]


def params_from_row(oracle, row):
    ints = {"cube_material", "sphere_material", "disable_sphere"}
    kw = {k: (int(v) if k in ints else float(v)) for k, v in zip(PARAM_KEYS, row)}
    return oracle.default_params(**kw)


def test_air_dist_bits(oracle):
    assert np.float32(oracle.AIR_DIST).view(np.uint32) == 0x3DCF53C6


def test_survey_kats(oracle):
    kat = json.load(open(os.path.join(GOLD, "demo_sdf_kat.json")))
    prm = oracle.default_params()
    for k in kat["kats"]:
        s = oracle.sample(prm, k["p"])
        np.testing.assert_allclose(s, np.array(k["sample"], np.float32), rtol=0, atol=1e-7)  # the vectors carry 7-8 significant digits
        t0, t1 = oracle.pack(s)
        np.testing.assert_allclose(t0, np.array(k["tex0"], np.float32), rtol=0, atol=1e-7)
        np.testing.assert_array_equal(t1[:3], np.array(k["tex1"], np.float32))
        assert t1[3] == np.float32(oracle.AIR_DIST)  # tex1.a keeps new_voxels' AIR_DIST
        if "u8" in k:
            colour = s[1:4] if any(s[1:4] != 0) else np.full(3, 0.5, np.float32)
            assert [oracle.L.or_srgb_quantize(float(c)) for c in colour] == k["u8"]
    for idx, want in kat["coords_n64_bb_m1_1"].items():
        assert np.float32(oracle.L.or_voxel_coord(int(idx), 64, -1.0, 1.0)) == np.float32(want)


def test_numpy_restatement_grid(oracle):
    g = np.load(os.path.join(GOLD, "grid_9x7x5.npz"))
    dims = tuple(int(d) for d in g["dims"])
    for k, row in enumerate(g["params"]):
        t0, t1 = oracle.fill_dense(params_from_row(oracle, row), dims, g["bb_min"], g["bb_max"], threads=2)
        np.testing.assert_array_equal(t0.view(np.uint32), g[f"tex0_{k}"].view(np.uint32))
        np.testing.assert_array_equal(t1.view(np.uint32), g[f"tex1_{k}"].view(np.uint32))


def test_numpy_restatement_points(oracle):
    g = np.load(os.path.join(GOLD, "points_512.npz"))
    pts = g["points"]
    for k, row in enumerate(g["params"]):
        prm = params_from_row(oracle, row)
        for sdf_id in (0, 1, 2):
            for do in (0, 1):
                got = oracle.sample_many(prm, pts, bool(do), sdf_id)
                want = g[f"s_{k}_{sdf_id}_{do}"]
                # bit-exact, NaN-aware (the sphere's colour is NaN at |p| = 0 only)
                np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_srgb_lut_is_correctly_rounded(oracle):
    c = np.arange(256, dtype=np.float32) / np.float32(255)
    hi = np.power(((c + np.float32(0.055)) / np.float32(1.055)).astype(np.float64), float(np.float32(2.4)))
    want = np.where(c < np.float32(0.04045), (c / np.float32(12.92)).astype(np.float64), hi).astype(np.float32)
    got = np.array([oracle.L.or_srgb_u8_to_linear(i) for i in range(256)], np.float32)
    np.testing.assert_array_equal(got, want)


def test_srgb_quantize_saturates(oracle):
    q = oracle.L.or_srgb_quantize
    assert q(0.0) == 0 and q(-1.0) == 0 and q(float("nan")) == 0
    assert q(1.0) == 255 and q(7.5) == 255 and q(0.999) == 254 and q(0.5) == 127


# ---- the reference's own tests: loading.rs:117-171 -------------------------------------------------
def loading_impl(oracle, limits):
    hits = np.zeros(limits[0] * limits[1] * limits[2], np.int64)
    num_passes = 3
    m = oracle.lm_new(limits, num_passes)
    remaining = oracle.L.or_lm_len(oracle.C.byref(m))
    iterations = 0
    total = iterations + remaining
    while True:
        v = oracle.lm_next(m)
        if v is None:
            break
        flat = v[0] + v[1] * limits[0] + v[2] * limits[0] * limits[1]
        hits[flat] += 1
        assert hits[flat] <= num_passes
        iterations += 1
        remaining = oracle.L.or_lm_len(oracle.C.byref(m))
        assert total == iterations + remaining
    assert (hits >= 1).all(), "developer error: voxel was not hit"


@pytest.mark.parametrize("limits", [(2, 2, 2), (8, 8, 8), (11, 11, 11), (8, 11, 17), (32, 32, 32)])
def test_interlacing(oracle, limits):
    loading_impl(oracle, limits)


def test_prev_power_of_2_and_passes_left(oracle):
    assert [oracle.L.or_prev_power_of_2(x) for x in (0, 1, 2, 3, 4, 7, 8, 1023)] == [0, 1, 2, 2, 4, 4, 8, 512]
    m = oracle.lm_new((4, 4, 4), 3)
    assert m.step_size == 4 and oracle.L.or_lm_passes_left(oracle.C.byref(m)) == 3
    while oracle.lm_next(m) is not None:
        pass
    assert oracle.L.or_lm_passes_left(oracle.C.byref(m)) == 0


def test_progressive_update_converges_to_dense(oracle):
    """G9: the LoadingManager-ordered update loop ends in exactly the dense fill."""
    prm = oracle.default_params()
    dims = (12, 9, 10)
    t0, t1 = oracle.grid_init(dims)
    lm = oracle.lm_new(dims, 2)
    steps = 0
    while True:
        n = oracle.viewer_update(prm, dims, lm, t0, t1, max_iterations=97)
        steps += n
        if n == 0:
            break
    d0, d1 = oracle.fill_dense(prm, dims, threads=2)
    np.testing.assert_array_equal(t0.view(np.uint32), d0.view(np.uint32))
    np.testing.assert_array_equal(t1.view(np.uint32), d1.view(np.uint32))
    assert steps == 12 * 9 * 10 + 6 * 5 * 5


def test_grid_dims_from_bb(oracle):
    import ctypes as C
    dims = (C.c_uint32 * 3)()
    oracle.L.or_grid_dims_from_bb(oracle.f3((-1, -1, -1)), oracle.f3((1, 1, 1)), 64, dims)
    assert list(dims) == [64, 64, 64]
    oracle.L.or_grid_dims_from_bb(oracle.f3((0, 0, 0)), oracle.f3((2, 1, 0.5)), 64, dims)
    assert list(dims) == [64, 32, 16]


def test_raymarch_oracle_sanity(oracle):
    """Default camera on a 32^3 grid: the box is hit, corners of the image are not, hits are opaque."""
    prm = oracle.default_params()
    dims = (32, 32, 32)
    t0, t1 = oracle.fill_dense(prm, dims, threads=4)
    rp = oracle.default_render_params(dims)
    cam = oracle.camera_look_at(aspect=1.0)
    rgba, aux = oracle.raymarch(rp, t0, t1, cam, 48, 48, threads=4)
    assert aux["status"][0, 0] == 0 and (rgba[0, 0] == 0).all()
    hit = aux["status"] == 1
    assert hit.sum() > 200
    assert (rgba[hit][:, 3] == 1.0).all() and (rgba[~hit] == 0).all()
    assert (aux["steps"][hit] >= 1).all() and aux["steps"].max() <= 255
    assert np.isfinite(rgba).all() and rgba.min() >= 0.0 and rgba.max() <= 1.0


def test_numpy_restatement_raymarch(oracle):
    """oracle/raymarch.c against the independent numpy-float32 restatement of material.frag (two cameras, one of
    them inside the volume): hit flags, step counts and hit positions bit for bit, shaded RGBA to pow() rounding."""
    import ctypes as C
    g = np.load(os.path.join(GOLD, "raymarch_12cube_40x30.npz"))
    t0, t1 = np.ascontiguousarray(g["tex0"]), np.ascontiguousarray(g["tex1"])
    W, H = int(g["width"]), int(g["height"])
    # the fixture's grid is itself a restatement output: it must equal the oracle's fill
    d0, d1 = oracle.fill_dense(oracle.default_params(), (12, 12, 12), threads=1)
    np.testing.assert_array_equal(t0.view(np.uint32), d0.view(np.uint32))
    np.testing.assert_array_equal(t1.view(np.uint32), d1.view(np.uint32))
    rp = oracle.default_render_params((12, 12, 12))
    for k in (0, 1):
        cam = oracle.Camera()
        C.memmove(C.byref(cam), g[f"cam_{k}"].ctypes.data, C.sizeof(cam))
        rgba, aux = oracle.raymarch(rp, t0, t1, cam, W, H, threads=1)
        np.testing.assert_array_equal(aux["status"], g[f"status_{k}"])
        np.testing.assert_array_equal(aux["steps"], g[f"steps_{k}"])
        covered = aux["status"] != 0
        np.testing.assert_array_equal(aux["hit_pos"][covered].view(np.uint32), g[f"hit_pos_{k}"][covered].view(np.uint32))
        assert np.abs(rgba - g[f"rgba_{k}"]).max() <= 2e-7
        assert (aux["status"] == 1).sum() > 50
    # the oracle's own camera builder agrees with the restated one (bvp excluded: the fixture leaves it zero)
    cam0 = oracle.camera_look_at(aspect=W / H)
    np.testing.assert_array_equal(np.frombuffer(bytes(cam0), np.float32)[:14].view(np.uint32),
                                  g["cam_0"][:14].view(np.uint32))


def test_numpy_restatement_mesh_front(oracle):
    """ScalarSource / HermiteSource / Mesh::postproc (src/sdf/meshers) against the numpy restatement's fixture."""
    g = np.load(os.path.join(GOLD, "mesh_front_200.npz"))
    unit, verts = g["unit_points"], g["vertices"]
    for k, row in enumerate(g["params"]):
        prm = params_from_row(oracle, row)
        for sdf_id in (0, 1, 2):
            got = oracle.source_scalar_many(prm, unit, g["bb_min"], g["bb_max"], sdf_id)
            np.testing.assert_array_equal(got.view(np.uint32), g[f"scalar_{k}_{sdf_id}"].view(np.uint32))
            got = oracle.source_normal_many(prm, unit, g["bb_min"], g["bb_max"], sdf_id)
            np.testing.assert_array_equal(got.view(np.uint32), g[f"normal_{k}_{sdf_id}"].view(np.uint32))
            got = oracle.mesh_postproc(prm, verts, sdf_id)
            np.testing.assert_array_equal(got.view(np.uint32), g[f"postproc_{k}_{sdf_id}"].view(np.uint32))


def test_ply_colour_quantisation(oracle):
    """(c * 255.9999) as u8, meshers/mesh.rs:106-108: truncating and saturating."""
    q = oracle.L.or_ply_color_u8
    assert [q(0.0), q(1.0), q(0.5), q(2.0), q(-0.3), q(float("nan"))] == [0, 255, 127, 255, 0, 0]
    assert q(1.0 / 255.9999 * 3) in (2, 3) and q(0.999) == 255 and q(0.99) == 253


def test_texel_touch_recording_agrees_with_the_plain_march(oracle):
    """or_raymarch_touch (SURVEY 8d byte model) is the same march with bookkeeping: its counts equal what the aux records
    of or_raymarch say, every texel under a hit was also read by the march's last fetch, tex1's footprint equals tex0's."""
    dims = (24, 20, 16)
    t0, t1 = oracle.fill_dense(oracle.default_params(), dims, threads=2)
    rp = oracle.default_render_params(dims)
    cam = oracle.camera_look_at(aspect=64 / 48)
    _, aux = oracle.raymarch(rp, t0, t1, cam, 64, 48, threads=2)
    maps, c = oracle.raymarch_touch(rp, t0, t1, cam, 64, 48, threads=2)
    assert c["pixels"] == 64 * 48 and c["covered"] == int((aux["status"] != 0).sum())
    assert c["hits"] == int((aux["status"] == 1).sum()) > 0
    assert c["sum_steps"] == int(aux["steps"].sum()) and c["max_steps"] == int(aux["steps"].max())
    assert maps["march0"].any() and (maps["hit0"] <= maps["march0"]).all()
    np.testing.assert_array_equal(maps["hit0"], maps["hit1"])
    assert maps["normal0"].sum() >= maps["hit0"].sum() > 0
    # accumulating a second camera only adds texels
    cam2 = oracle.camera_look_at(eye=(-2.5, 3.0, 5.0), aspect=64 / 48)
    before = maps["march0"].copy()
    maps, c2 = oracle.raymarch_touch(rp, t0, t1, cam2, 64, 48, threads=2, maps=maps)
    assert (before <= maps["march0"]).all() and maps["march0"].sum() > before.sum()


def test_ext_variant_switches_default_off_and_change_what_they_name(oracle):
    """The [EXT] sensitivity switches (tools/ext_sensitivity.py) are off in every parity test and each moves only its piece."""
    assert oracle.L.or_get_ext_variant() == 0
    base = np.array([oracle.L.or_srgb_u8_to_linear(i) for i in range(256)], np.float32)
    try:
        oracle.L.or_set_ext_variant(oracle.EXT_VARIANTS["srgb_pow_ulp_up"])
        up = np.array([oracle.L.or_srgb_u8_to_linear(i) for i in range(256)], np.float32)
        pow_branch = np.arange(256) / 255.0 >= 0.04045
        np.testing.assert_array_equal(up[~pow_branch], base[~pow_branch])
        np.testing.assert_array_equal(up[pow_branch].view(np.uint32), base[pow_branch].view(np.uint32) + 1)
        oracle.L.or_set_ext_variant(oracle.EXT_VARIANTS["srgb_quant_round"])
        assert oracle.L.or_srgb_quantize(0.5) == 128 and oracle.L.or_srgb_quantize(0.999) == 255
    finally:
        oracle.L.or_set_ext_variant(0)
    assert oracle.L.or_srgb_quantize(0.5) == 127 and oracle.L.or_srgb_quantize(0.999) == 254   # truncating `as u8`
    np.testing.assert_array_equal(np.array([oracle.L.or_srgb_u8_to_linear(i) for i in range(256)], np.float32), base)


def test_committed_raymarch_byte_model_matches_its_definition():
    """profiles/raymarch_model_bytes.json (what bench.py attaches to roofline_raymarch): the two SURVEY 8(d) figures follow
    from the recorded counts by the survey's own formulas."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "raymarch_model_bytes.json")
    d = json.load(open(path))
    for key in ("256", "512", "256_batch64"):
        m = d[key]
        out = 16 * m["image"][0] * m["image"][1] * m["cameras"]
        assert m["output_bytes"] == out
        assert m["nominal_gather_bytes"] == 128 * (m["counts"]["sum_steps"] + 5 * m["counts"]["hits"]) + out
        u = m["unique_texels"]
        assert m["compulsory_bytes"] == 16 * (u["tex0_all"] + u["tex1_hit"]) + out
        assert u["tex0_march"] <= u["tex0_all"] <= u["of_grid"] and m["counts"]["max_steps"] <= 255
    assert d["256"]["counts"] == {"pixels": 2073600, "covered": 283681, "hits": 245501, "sum_steps": 4367031, "max_steps": 255}
