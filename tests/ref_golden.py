"""Loader of the reference-side fixtures tools/ref_golden/golden_gen.rs writes (tests/golden/ref_*.json).

The files do not exist until someone with a Rust toolchain runs the generator inside a checkout of the reference
(tools/ref_golden/README.md); every consumer skips with that reason.  Floats travel as the 8 hex digits of f32::to_bits."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ("ref_samples.json", "ref_srgb.json", "ref_grid_9x7x5.json")
SKIP_REASON = "tests/golden/{} not present: run tools/ref_golden/golden_gen.rs inside the reference (tools/ref_golden/README.md)"
PARAM_INTS = ("cube_material", "sphere_material", "disable_sphere")
PARAM_FLOATS = ("cube_half_side", "sphere_radius", "max_distance_custom_material")


def f32(words):
    """Nested lists of hex strings -> float32 array of the same shape."""
    a = np.asarray(words)
    flat = np.fromiter((int(w, 16) for w in a.ravel()), dtype=np.uint32, count=a.size)
    return flat.view(np.float32).reshape(a.shape)


def load(name, directory=None, allow_emulated=False):
    """The parsed file, or pytest.skip when it is absent.  A file the emulator wrote from the ORACLE is refused unless the
    caller is the consumer's own plumbing test: the oracle cannot pin itself."""
    path = os.path.join(directory or GOLD, name)
    if not os.path.exists(path):
        pytest.skip(SKIP_REASON.format(name))
    with open(path) as f:
        data = json.load(f)
    emulated = "emulated" in data.get("generator", "")
    if emulated and not allow_emulated:
        pytest.fail(f"{path} was written by tools/ref_golden/emulate.py (from the oracle), not by the reference: remove it")
    return data


def params_kw(cfg):
    """A config's "params" object -> keyword arguments of default_params() (product or oracle)."""
    kw = {k: int(cfg["params"][k]) for k in PARAM_INTS}
    kw.update({k: float(f32(cfg["params"][k])) for k in PARAM_FLOATS})
    return kw


def same_words(got, want):
    """Bit equality of two float32 arrays, except that a NaN matches a NaN of any payload (x86, gcc and gfx950 agree on the
    default NaN in practice, but only NaN-ness is part of the reference's semantics)."""
    got = np.ascontiguousarray(got, np.float32)
    want = np.ascontiguousarray(want, np.float32)
    assert got.shape == want.shape, (got.shape, want.shape)
    return (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))


def assert_same_words(got, want, what):
    ok = same_words(got, want)
    if not ok.all():
        bad = np.argwhere(~ok)
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {ok.size} words differ from the reference; first at {first}: "
                             f"got {np.asarray(got, np.float32)[first]!r} want {np.asarray(want, np.float32)[first]!r}")


def srgb_policy(data):
    """Which Srgba::from conversion explains every recorded u8: "truncate", "round", or a failure listing both miss counts."""
    c = f32([case["c"] for case in data["cases"]])
    u8 = np.array([case["u8"] for case in data["cases"]], np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        v = c * np.float32(255.0)
        def as_u8(x):  # Rust `as u8`: truncating, saturating, NaN -> 0
            x = np.where(np.isnan(x), np.float32(0), x)
            return np.clip(np.trunc(x), 0, 255).astype(np.int64)
        trunc = as_u8(v)
        rnd = as_u8(v + np.float32(0.5))
    miss_t, miss_r = int((trunc != u8).sum()), int((rnd != u8).sum())
    if miss_t == 0 and miss_r != 0:
        return "truncate"
    if miss_r == 0 and miss_t != 0:
        return "round"
    raise AssertionError(f"neither conversion explains ref_srgb.json: truncate misses {miss_t}, round misses {miss_r} of {len(u8)} "
                         "cases -- three-d-asset's Srgba::from does something else; restate it in oracle/grid_fill.c first")
