"""The ORACLE against the reference-side fixtures (tests/golden/ref_*.json, written by tools/ref_golden/golden_gen.rs run
inside a checkout of the reference).  Absent files skip; the consumer code itself is exercised on files the emulator
writes from the oracle into a temporary directory (plumbing only -- that proves nothing about parity and says so)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import ref_golden as rg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emulator():
    spec = importlib.util.spec_from_file_location("ref_golden_emulate", os.path.join(ROOT, "tools", "ref_golden", "emulate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def check_srgb(data, oracle):
    """Truncate or round?  The answer must be the product's default; the linear words must be the oracle's."""
    policy = rg.srgb_policy(data)
    assert policy == "truncate", (
        "the reference's Srgba::from ROUNDS: make rounding the default -- g_options.ext_srgb_quant = 1 in "
        "sdf-viewer_amd/csrc/api_internal.h and OR_EXT_SRGB_QUANT_ROUND in oracle/demo_sdf.c's or_ext_variant_flags; both "
        "policies are compiled in and tested (tests/test_gpu_ext_srgb.py)")
    u8 = [case["u8"] for case in data["cases"]]
    want = rg.f32([case["linear"] for case in data["cases"]])
    got = np.array([oracle.L.or_srgb_u8_to_linear(k) for k in u8], np.float32)
    rg.assert_same_words(got, want, "to_linear_srgb")


def check_samples(data, oracle):
    for cfg in data["configs"]:
        prm = oracle.default_params(**rg.params_kw(cfg))
        pts = rg.f32(cfg["points"])
        for sdf_id in (0, 1, 2):
            rec = cfg["ids"][str(sdf_id)]
            what = f"flags {cfg['flags']} id {sdf_id}"
            rg.assert_same_words(oracle.sample_many(prm, pts, False, sdf_id), rg.f32(rec["sample"]), what + " sample")
            rg.assert_same_words(oracle.sample_many(prm, pts, True, sdf_id), rg.f32(rec["sample_distance_only"]),
                                 what + " sample(distance_only)")
            rg.assert_same_words(oracle.normal_many(prm, pts, 0.0, sdf_id), rg.f32(rec["normal"]), what + " normal")
            assert rg.f32(rec["bounding_box"]).tolist() == [-1, -1, -1, 1, 1, 1]


def check_grid(data, oracle, configs):
    dims = tuple(data["dims"])
    assert rg.f32(data["air_dist"]).view(np.uint32) == np.float32(oracle.AIR_DIST).view(np.uint32) == 0x3DCF53C6
    shape = (dims[2], dims[1], dims[0], 4)
    for entry in data["grids"]:
        kw = rg.params_kw(configs[entry["config"]])
        prm = oracle.default_params(**kw)
        t0, t1 = oracle.grid_init(dims)
        lm = oracle.lm_new(dims, data["loading_passes"])
        assert oracle.viewer_update(prm, dims, lm, t0, t1) == entry["iterations"]
        assert oracle.L.or_lm_passes_left(C.byref(lm)) == entry["passes_left"] == 0
        rg.assert_same_words(t0, rg.f32(entry["tex0"]).reshape(shape), f"grid config {entry['config']} tex0")
        rg.assert_same_words(t1, rg.f32(entry["tex1"]).reshape(shape), f"grid config {entry['config']} tex1")
        if "edit" in entry:
            e = entry["edit"]
            edited = oracle.default_params(**dict(kw, max_distance_custom_material=float(rg.f32(e["max_distance_custom_material"]))))
            lm = oracle.lm_new(dims, 3)
            it = oracle.viewer_update(edited, dims, lm, t0, t1, changed_box=(-1, -1, -1, 1, 1, 1))
            lm = oracle.lm_new(dims, 3)
            it += oracle.viewer_update(edited, dims, lm, t0, t1)
            assert it == e["iterations"]
            rg.assert_same_words(t0, rg.f32(e["tex0"]).reshape(shape), "grid after the edit tex0")
            rg.assert_same_words(t1, rg.f32(e["tex1"]).reshape(shape), "grid after the edit tex1")


def test_ref_golden_srgb_pins_the_quantisation(oracle):
    check_srgb(rg.load("ref_srgb.json"), oracle)


def test_ref_golden_samples_pin_the_demo_sdf(oracle):
    check_samples(rg.load("ref_samples.json"), oracle)


def test_ref_golden_grid_pins_update_and_packing(oracle):
    check_grid(rg.load("ref_grid_9x7x5.json"), oracle, rg.load("ref_samples.json")["configs"])


def test_ref_golden_consumer_plumbing_on_emulated_files(tmp_path, oracle):
    """Not a parity statement: the files are the oracle's own output in the generator's schema; this keeps the loader, the hex
    codec, the policy detector and the three checks above from rotting while the real files do not exist."""
    emu = _emulator()
    emu.write_all(str(tmp_path), oracle)
    samples = rg.load("ref_samples.json", str(tmp_path), allow_emulated=True)
    assert [len(c["points"]) for c in samples["configs"]] == [4096 + 12, 1024 + 12, 512 + 12, 512 + 12]
    check_samples(samples, oracle)
    check_srgb(rg.load("ref_srgb.json", str(tmp_path), allow_emulated=True), oracle)
    check_grid(rg.load("ref_grid_9x7x5.json", str(tmp_path), allow_emulated=True), oracle, samples["configs"])
    # the detector tells the two conversions apart, and a corrupted word is reported
    flag = oracle.EXT_VARIANTS["srgb_quant_round"]
    oracle.L.or_set_ext_variant(flag)
    try:
        emu.write_all(str(tmp_path / "round"), oracle)
    finally:
        oracle.L.or_set_ext_variant(0)
    assert rg.srgb_policy(rg.load("ref_srgb.json", str(tmp_path / "round"), allow_emulated=True)) == "round"
    samples["configs"][0]["ids"]["0"]["sample"][5][0] = "3f800000"
    with pytest.raises(AssertionError, match="differ from the reference"):
        check_samples(samples, oracle)


def test_ref_golden_refuses_emulated_files_as_fixtures(tmp_path, oracle):
    _emulator().write_all(str(tmp_path), oracle)
    with pytest.raises(pytest.fail.Exception, match="emulate.py"):
        rg.load("ref_srgb.json", str(tmp_path))


def test_ref_golden_seeded_points_match_the_generator_arithmetic():
    """The xorshift32 -> float32 arithmetic of golden_gen.rs's points(), restated: first seeded point, and the range."""
    pts = _emulator().points(64)
    assert pts.shape == (76, 3) and pts.dtype == np.float32
    assert float(np.abs(pts[12:]).max()) < 1.25
    s = 0x9E3779B9
    s ^= (s << 13) & 0xFFFFFFFF
    s ^= s >> 17
    s ^= (s << 5) & 0xFFFFFFFF
    assert pts[12, 0] == np.float32(np.float32(np.float32(s >> 8) * np.float32(2.0 ** -24)) * np.float32(2.5)) - np.float32(1.25)
