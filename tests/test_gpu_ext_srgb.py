"""SDFV_OPT_EXT_SRGB_QUANT: the one restated piece of un-vendored arithmetic whose alternative moves visible output
(three-d-asset's Srgba::from(Vector3<f32>), call site /root/reference/src/app/scene/sdf/mod.rs:201) is a switch of the
PRODUCT, not only of the oracle.  Both policies -- 0: (c * 255.0) as u8, 1: (c * 255.0 + 0.5) as u8 -- are compiled into every
fill and pass kernel; each is bit-exact against the oracle evaluating the same policy (OR_EXT_SRGB_QUANT_ROUND), from the
smallest grids to every voxel of BASELINE configs[1]."""
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def policy(pkg, oracle, rounding):
    """The library's option and the oracle's variant flag, set together and restored together."""
    before = oracle.L.or_get_ext_variant()
    flag = oracle.EXT_VARIANTS["srgb_quant_round"]
    oracle.L.or_set_ext_variant((before | flag) if rounding else (before & ~flag))
    try:
        with pkg.options({pkg._capi.OPT_EXT_SRGB_QUANT: 1 if rounding else 0}):
            yield
    finally:
        oracle.L.or_set_ext_variant(before)


def bits(t):
    return t.cpu().numpy().view(np.uint32)


def fill(pkg, prm, dims, sdf_id=0, dist=False, bb=((-1, -1, -1), (1, 1, 1))):
    g = pkg.make_grid(dims, *bb)
    t0, t1 = pkg.alloc_textures(g)
    t0.fill_(-7.0)
    t1.fill_(-7.0)
    d = torch.full(tuple(t0.shape[:-1]), -7.0, dtype=torch.float32, device="cuda") if dist else None
    pkg.fill_grid(prm, g, t0, t1, sdf_id=sdf_id, dist=d)
    torch.cuda.synchronize()
    return t0, t1, d


def test_option_is_validated_and_defaults_to_truncation(pkg):
    K = pkg._capi
    assert pkg.get_option(K.OPT_EXT_SRGB_QUANT) == 0
    with pytest.raises(pkg.SdfvError):
        pkg.set_option(K.OPT_EXT_SRGB_QUANT, 2)
    assert pkg.get_option(K.OPT_EXT_SRGB_QUANT) == 0


@pytest.mark.parametrize("rounding", [False, True])
@pytest.mark.parametrize("dims", [(64, 64, 64), (33, 5, 70), (130, 3, 9), (9, 7, 5)])
def test_dense_fill_both_policies(pkg, oracle, rounding, dims):
    """Default configuration (compile-time packed constants of that policy) and the run-time configurations (every sub-tree,
    both materials), plain and fused fill, row-chunk and flat index forms."""
    cases = [(pkg.default_params(), 0), (pkg.default_params(cube_material=1, sphere_material=0), 0),
             (pkg.default_params(disable_sphere=1), 0), (pkg.default_params(), 1), (pkg.default_params(), 2),
             (pkg.default_params(sphere_material=0), 2)]
    with policy(pkg, oracle, rounding):
        for prm, sdf_id in cases:
            r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, sdf_id=sdf_id)
            for fused in (False, True):
                t0, t1, d = fill(pkg, prm, dims, sdf_id=sdf_id, dist=fused)
                np.testing.assert_array_equal(bits(t0), r0.view(np.uint32))
                np.testing.assert_array_equal(bits(t1), r1.view(np.uint32))
                assert d is None or torch.equal(d, t0[..., 0])
            with pkg.options({pkg._capi.OPT_FILL_FORM: pkg._capi.FILL_FORM["flat"]}):
                t0, t1, _ = fill(pkg, prm, dims, sdf_id=sdf_id)
            np.testing.assert_array_equal(bits(t0), r0.view(np.uint32))


@pytest.mark.parametrize("rounding", [False, True])
def test_interleaved_volume_fills_both_policies(pkg, oracle, rounding):
    """The fills that write the y-interleaved volume -- the row-chunk form and, from widths of 512, the kernel with a thread per
    x of both rows of a pair -- under either policy, default and run-time configurations, against the oracle."""
    K = pkg._capi
    for dims in ((512, 4, 3), (128, 6, 5)):
        for prm in (pkg.default_params(), pkg.default_params(cube_material=1, sphere_material=0, sphere_radius=0.9)):
            with policy(pkg, oracle, rounding):
                g = pkg.make_grid(dims)
                t0, t1 = pkg.alloc_textures(g)
                vol = torch.full(tuple(t0.shape[:-1]), -7.0, dtype=torch.float32, device="cuda")
                pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | K.PASS_VOLUME_INTERLEAVED)
                torch.cuda.synchronize()
                r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims)
            np.testing.assert_array_equal(bits(t0), r0.view(np.uint32))
            np.testing.assert_array_equal(bits(t1), r1.view(np.uint32))
            d = t0[..., 0]
            assert torch.equal(vol, torch.stack([d[:, 0::2], d[:, 1::2]], dim=-1).reshape(vol.shape))


def test_the_two_policies_differ_where_the_sensitivity_study_says(pkg, oracle):
    """The switch is not a no-op: at 64^3 the default demo's custom material (0.5, 0.6, 0.7 -> 127/153/178 truncated,
    128/153/179 rounded) and the sphere's |n| colours move; distances and tex1 do not."""
    prm = pkg.default_params()
    with policy(pkg, oracle, False):
        a0, a1, _ = fill(pkg, prm, (64, 64, 64))
    with policy(pkg, oracle, True):
        b0, b1, _ = fill(pkg, prm, (64, 64, 64))
    assert torch.equal(a1, b1) and torch.equal(a0[..., 0], b0[..., 0])
    moved = int((a0[..., 1:] != b0[..., 1:]).sum())
    assert moved > 1000, moved
    custom = oracle.pack(oracle.sample(oracle.params_from(prm), (0.9, 0.6, 0.0)))[0]
    assert custom[1:].tolist() == pytest.approx([0.2122307, 0.3185468, 0.4452012], abs=1e-7)  # SURVEY 8(c) KAT: truncation


@pytest.mark.parametrize("rounding", [False, True])
def test_progressive_passes_both_policies(pkg, oracle, rounding):
    """Every pass kernel (general, quad, whole rows fresh / loaded, flagged and unflagged) under either policy: the states at
    pass boundaries are the oracle's LoadingManager loop's, then a boxed edit."""
    K = pkg._capi
    dims = (24, 20, 12)
    prm, edited = pkg.default_params(), pkg.default_params(sphere_radius=0.8, cube_material=1)
    with policy(pkg, oracle, rounding):
        for use_dist in (False, True):
            for flagged in (False, True):
                g = pkg.make_grid(dims)
                t0, t1 = pkg.alloc_textures(g)
                pkg.grid_init(g, t0, t1)
                dist = pkg.commit_distance(g, t0) if use_dist else None
                r0, r1 = oracle.grid_init(dims)
                lm = oracle.lm_new(dims, 3)
                for k, step in enumerate((4, 2, 1)):
                    flags = ((K.PASS_FRESH_GRID if k == 0 else 0) | K.PASS_SAME_LOAD) if flagged else 0
                    pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist, flags=flags)
                    n = -(-dims[0] // step) * -(-dims[1] // step) * -(-dims[2] // step)
                    assert oracle.viewer_update(oracle.params_from(prm), dims, lm, r0, r1, max_iterations=n) == n
                    torch.cuda.synchronize()
                    np.testing.assert_array_equal(bits(t0), r0.view(np.uint32))
                    np.testing.assert_array_equal(bits(t1), r1.view(np.uint32))
                box = (-0.5, -1.0, -0.25, 0.5, 0.1, 1.0)
                lm = oracle.lm_new(dims, 3)
                for step in (4, 2, 1):
                    pkg.fill_grid_pass(edited, g, step, t0, t1, changed_box=box, dist=dist)
                oracle.viewer_update(oracle.params_from(edited), dims, lm, r0, r1, changed_box=box)
                torch.cuda.synchronize()
                np.testing.assert_array_equal(bits(t0), r0.view(np.uint32))
                np.testing.assert_array_equal(bits(t1), r1.view(np.uint32))
                assert dist is None or torch.equal(dist, t0[..., 0])


@pytest.mark.parametrize("rounding", [False, True])
def test_every_voxel_of_configs1_both_policies(pkg, oracle, rounding):
    """BASELINE configs[1]'s 256^3 grid, every word of both textures and of the fused volume, under either policy -- with the
    kernel instantiations bench.py times (plain stores / nt + volume)."""
    dims = (256, 256, 256)
    prm = pkg.default_params()
    with policy(pkg, oracle, rounding):
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, threads=16)
        for fused in (False, True):
            t0, t1, d = fill(pkg, prm, dims, dist=fused)
            assert torch.equal(t0.cpu().view(torch.int32), torch.from_numpy(r0).view(torch.int32))
            assert torch.equal(t1.cpu().view(torch.int32), torch.from_numpy(r1).view(torch.int32))
            assert d is None or torch.equal(d, t0[..., 0])
