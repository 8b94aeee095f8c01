"""GPU parity tests of the grid fill (through the C ABI): bit-exact against the oracle and the committed
golden fixtures at small sizes; size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_KEYS = ["cube_half_side", "cube_material", "sphere_radius", "sphere_material",
              "max_distance_custom_material", "disable_sphere"]
INTS = {"cube_material", "sphere_material", "disable_sphere"}


def kw_from_row(row):
    return {k: (int(v) if k in INTS else float(v)) for k, v in zip(PARAM_KEYS, row)}


def gpu_fill(pkg, prm, dims, bb_min=(-1, -1, -1), bb_max=(1, 1, 1), z0=0, z1=None, sdf_id=0):
    g = pkg.make_grid(dims, bb_min, bb_max, z0, z1)
    t0, t1 = pkg.alloc_textures(g)
    t0.fill_(-7.0)
    t1.fill_(-7.0)
    pkg.fill_grid(prm, g, t0, t1, sdf_id=sdf_id)
    torch.cuda.synchronize()
    return t0, t1


def assert_bits_equal(gpu_tensor, ref):
    got = gpu_tensor.cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_golden_grid_fixture(pkg):
    g = np.load(os.path.join(GOLD, "grid_9x7x5.npz"))
    dims = tuple(int(d) for d in g["dims"])
    for k, row in enumerate(g["params"]):
        t0, t1 = gpu_fill(pkg, pkg.default_params(**kw_from_row(row)), dims, g["bb_min"], g["bb_max"])
        assert_bits_equal(t0, g[f"tex0_{k}"])
        assert_bits_equal(t1, g[f"tex1_{k}"])


@pytest.mark.parametrize("dims", [(64, 64, 64), (8, 11, 17), (2, 2, 2), (33, 5, 70), (130, 3, 9), (300, 4, 3),
                                  (1, 5, 7), (5, 1, 1)])
def test_dense_fill_matches_oracle(pkg, oracle, dims):
    prm = pkg.default_params()
    t0, t1 = gpu_fill(pkg, prm, dims)
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims)
    assert_bits_equal(t0, r0)
    assert_bits_equal(t1, r1)


@pytest.mark.parametrize("form", ["flat", "rows"])
@pytest.mark.parametrize("dims", [(64, 16, 8), (7, 13, 300), (257, 3, 5), (1, 9, 4), (1000, 6, 2)])
def test_both_index_forms_of_the_dense_fill(pkg, oracle, form, dims):
    """The row-chunk and the flat form of the dense kernel produce the same texels on every width, whichever the
    launcher would have picked (SDFV_OPT_FILL_FORM overrides the choice; slab offsets included)."""
    for prm in (pkg.default_params(), pkg.default_params(cube_material=1, disable_sphere=1)):
        z0 = dims[2] // 3
        # both store policies of each form (2 = plain stores, 1 = nt; auto picks by whether a distance volume is written)
        with pkg.options({pkg._capi.OPT_FILL_FORM: pkg._capi.FILL_FORM[form], pkg._capi.OPT_FILL_NONTEMPORAL: 1}):
            n0, n1 = gpu_fill(pkg, prm, dims, z0=z0, z1=dims[2])
        with pkg.options({pkg._capi.OPT_FILL_FORM: pkg._capi.FILL_FORM[form], pkg._capi.OPT_FILL_NONTEMPORAL: 2}):
            t0, t1 = gpu_fill(pkg, prm, dims, z0=z0, z1=dims[2])
        assert torch.equal(n0, t0) and torch.equal(n1, t1)
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, z0=z0, z1=dims[2])
        assert_bits_equal(t0, r0)
        assert_bits_equal(t1, r1)


@pytest.mark.parametrize("kw", [dict(cube_material=1, sphere_material=0), dict(disable_sphere=1),
                                dict(cube_half_side=0.5, sphere_radius=0.6, max_distance_custom_material=0.0),
                                dict(cube_half_side=0.8, sphere_radius=0.3, max_distance_custom_material=0.25),
                                dict(sphere_radius=1.25, cube_half_side=1.0)])
@pytest.mark.parametrize("sdf_id", [0, 1, 2])
def test_params_and_subtrees(pkg, oracle, kw, sdf_id):
    """Every child SDF can be rendered on its own (app/mod.rs:200-208); parameters as the clap flags."""
    prm = pkg.default_params(**kw)
    dims = (40, 37, 21)
    bb = ((-1.0, -0.75, -1.0), (1.0, 1.0, 0.5))
    t0, t1 = gpu_fill(pkg, prm, dims, *bb, sdf_id=sdf_id)
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, *bb, sdf_id=sdf_id)
    assert_bits_equal(t0, r0)
    assert_bits_equal(t1, r1)


def test_slab_fill_equals_full_fill(pkg, oracle):
    """z-slab sharding: filling slabs independently reproduces the dense grid bit for bit."""
    prm = pkg.default_params()
    dims = (48, 40, 37)
    full0, full1 = gpu_fill(pkg, prm, dims)
    bounds = [0, 9, 10, 25, 37]
    parts0, parts1 = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        s0, s1 = gpu_fill(pkg, prm, dims, z0=a, z1=b)
        parts0.append(s0)
        parts1.append(s1)
    assert torch.equal(torch.cat(parts0), full0) and torch.equal(torch.cat(parts1), full1)
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, z0=9, z1=10)
    assert_bits_equal(parts0[1], r0)


def test_progressive_passes_converge_to_dense(pkg, oracle):
    """LoadingManager passes (step 4, 2, 1) with update_required end in the dense state; every
    intermediate state equals the oracle's LoadingManager-ordered loop stopped at the pass boundary."""
    prm = pkg.default_params()
    oprm = oracle.params_from(prm)
    dims = (21, 18, 13)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    torch.cuda.synchronize()
    assert (t0 == pkg.AIR_DIST).all() and (t1 == pkg.AIR_DIST).all()
    r0, r1 = oracle.grid_init(dims)
    lm = oracle.lm_new(dims, 3)
    for step in (4, 2, 1):
        pkg.fill_grid_pass(prm, g, step, t0, t1)
        n = -(-dims[0] // step) * -(-dims[1] // step) * -(-dims[2] // step)
        assert oracle.viewer_update(oprm, dims, lm, r0, r1, max_iterations=n) == n
        torch.cuda.synchronize()
        assert_bits_equal(t0, r0)
        assert_bits_equal(t1, r1)
    d0, d1 = gpu_fill(pkg, prm, dims)
    assert torch.equal(t0, d0) and torch.equal(t1, d1)


def test_changed_box_refill(pkg, oracle):
    """Parameter edit -> changed() box -> only voxels inside the box (or still AIR) are re-sampled
    (scene/sdf/mod.rs:131-154,184-190)."""
    dims = (24, 24, 24)
    prm = pkg.default_params()
    t0, t1 = gpu_fill(pkg, prm, dims)
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims)
    prm2 = pkg.default_params(sphere_radius=0.8, cube_material=1)
    box = (-0.5, -1.0, -0.25, 0.5, 0.1, 1.0)
    g = pkg.make_grid(dims)
    lm = oracle.lm_new(dims, 3)
    for step in (4, 2, 1):
        pkg.fill_grid_pass(prm2, g, step, t0, t1, changed_box=box)
    oracle.viewer_update(oracle.params_from(prm2), dims, lm, r0, r1, changed_box=box)
    torch.cuda.synchronize()
    assert_bits_equal(t0, r0)
    assert_bits_equal(t1, r1)
    assert not torch.equal(t0, gpu_fill(pkg, prm, dims)[0])


@pytest.mark.parametrize("side", [256, 512])
def test_full_size_properties(pkg, oracle, side):
    """BASELINE.json sizes: properties that do not need a full CPU fill."""
    prm = pkg.default_params()
    dims = (side, side, side)
    t0, t1 = gpu_fill(pkg, prm, dims)
    # (1) range / untouched-channel invariants of the packing
    assert float(t0[..., 0].min()) >= 0.0 and float(t0[..., 0].max()) <= 1.0
    assert bool((t1[..., 3] == pkg.AIR_DIST).all())
    assert bool(torch.isfinite(t0).all()) and bool(torch.isfinite(t1).all())
    assert not bool((t0 == -7.0).any()) and not bool((t1 == -7.0).any())  # every voxel written
    # (2) 4096 seeded voxels + the 8 corners against the oracle, bit for bit
    rng = np.random.default_rng(side)
    idx = rng.integers(0, side, size=(4096, 3))
    idx[:8] = [[x, y, z] for x in (0, side - 1) for y in (0, side - 1) for z in (0, side - 1)]
    ti = torch.from_numpy(idx).cuda()
    got0 = t0[ti[:, 2], ti[:, 1], ti[:, 0]].cpu().numpy()
    got1 = t1[ti[:, 2], ti[:, 1], ti[:, 0]].cpu().numpy()
    pts = np.array([[oracle.L.or_voxel_coord(int(i), side, -1.0, 1.0) for i in v] for v in idx], np.float32)
    samples = oracle.sample_many(oracle.params_from(prm), pts)
    for k in range(len(idx)):
        w0, w1 = oracle.pack(samples[k])
        assert (got0[k].view(np.uint32) == w0.view(np.uint32)).all(), (idx[k], got0[k], w0)
        assert (got1[k].view(np.uint32) == w1.view(np.uint32)).all(), (idx[k], got1[k], w1)
    # (3) three whole slices against the oracle (first, middle, last)
    for z in (0, side // 2, side - 1):
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, z0=z, z1=z + 1)
        assert_bits_equal(t0[z:z + 1], r0)
        assert_bits_equal(t1[z:z + 1], r1)
    # (4) idempotence: a LoadingManager pass over the finished grid changes nothing
    before = (t0.view(torch.int32).sum(dtype=torch.int64).item(), t1.view(torch.int32).sum(dtype=torch.int64).item())
    pkg.fill_grid_pass(prm, pkg.make_grid(dims), 1, t0, t1)
    torch.cuda.synchronize()
    after = (t0.view(torch.int32).sum(dtype=torch.int64).item(), t1.view(torch.int32).sum(dtype=torch.int64).item())
    assert before == after
    # (5) slab sharding at full size: an 8-way slab equals the same slices of the dense fill
    z0, z1 = 3 * side // 8, 4 * side // 8
    s0, s1 = gpu_fill(pkg, prm, dims, z0=z0, z1=z1)
    assert torch.equal(s0, t0[z0:z1]) and torch.equal(s1, t1[z0:z1])


def test_host_buffer_entry_point(pkg, oracle):
    import ctypes as C
    prm = pkg.default_params()
    dims = (16, 12, 10)
    g = pkg.make_grid(dims)
    t0 = np.zeros((10, 12, 16, 4), np.float32)
    t1 = np.zeros_like(t0)
    pkg.check(pkg.lib.sdfv_fill_grid_host(C.byref(prm), 0, C.byref(g), t0.ctypes.data, t1.ctypes.data))
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims)
    np.testing.assert_array_equal(t0.view(np.uint32), r0.view(np.uint32))
    np.testing.assert_array_equal(t1.view(np.uint32), r1.view(np.uint32))


def test_randomised_parameters_and_grids(pkg, oracle):
    """Seeded sweep over the demo's whole parameter space (the GUI sliders' ranges, demo/mod.rs:94-118,
    cube.rs:117-127, sphere.rs:75-85), odd grid shapes and off-centre boxes: dense fill and one progressive pass."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SDFV_SOAK_SEED", 20250404)))  # tools/soak.sh varies the seed
    for trial in range(int(os.environ.get("SDFV_SOAK_TRIALS", 24))):
        kw = dict(cube_half_side=float(rng.integers(0, 101)) / 100.0,             # Int 0..=100 mapped to [0, 1]
                  sphere_radius=float(np.float32(rng.uniform(0.0, 1.25))),
                  max_distance_custom_material=float(np.float32(rng.uniform(0.0, 0.25))),
                  cube_material=int(rng.integers(0, 2)), sphere_material=int(rng.integers(0, 2)),
                  disable_sphere=int(rng.integers(0, 4) == 0))
        prm = pkg.default_params(**kw)
        dims = tuple(int(d) for d in rng.integers(2, 40, size=3))
        lo = rng.uniform(-1.5, -0.2, size=3)
        hi = lo + rng.uniform(0.3, 2.5, size=3)
        sdf_id = int(rng.integers(0, 3))
        t0, t1 = gpu_fill(pkg, prm, dims, lo, hi, sdf_id=sdf_id)
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, lo, hi, sdf_id=sdf_id, threads=2)
        assert_bits_equal(t0, r0)
        assert_bits_equal(t1, r1)
        g = pkg.make_grid(dims, lo, hi)
        p0, p1 = pkg.alloc_textures(g)
        pkg.grid_init(g, p0, p1)
        pkg.fill_grid_pass(prm, g, 1, p0, p1, sdf_id=sdf_id)
        torch.cuda.synchronize()
        assert torch.equal(p0, t0) and torch.equal(p1, t1), (trial, kw, dims)
        # the same load through the distance-volume passes (quad kernel when the width allows), then a boxed refill
        pkg.grid_init(g, p0, p1)
        dvol = pkg.commit_distance(g, p0)
        for step in (2, 1):
            pkg.fill_grid_pass(prm, g, step, p0, p1, sdf_id=sdf_id, dist=dvol)
        torch.cuda.synchronize()
        assert torch.equal(p0, t0) and torch.equal(p1, t1) and torch.equal(dvol, t0[..., 0]), (trial, kw, dims)
        # the same load as its LoadingManager would flag it (first pass: fresh grid; later ones: same load), random steps
        K = pkg._capi
        steps = [int(s_) for s_ in sorted({int(2 ** rng.integers(0, 4)) for _ in range(3)} | {1}, reverse=True)]
        pkg.grid_init(g, p0, p1)
        pkg.commit_distance(g, p0, dist=dvol)
        for k, step in enumerate(steps):
            pkg.fill_grid_pass(prm, g, step, p0, p1, sdf_id=sdf_id, dist=dvol if trial % 2 else None,
                               flags=(K.PASS_FRESH_GRID if k == 0 else 0) | K.PASS_SAME_LOAD)
        torch.cuda.synchronize()
        assert torch.equal(p0, t0) and torch.equal(p1, t1), (trial, kw, dims, steps)
        if trial % 2:
            assert torch.equal(dvol, t0[..., 0]), (trial, kw, dims, steps)
        else:
            pkg.commit_distance(g, p0, dist=dvol)
        other = pkg.default_params(**dict(kw, cube_half_side=min(1.0, kw["cube_half_side"] + 0.07)))
        box = tuple(float(v) for v in np.concatenate([lo + (hi - lo) * rng.uniform(0.0, 0.5, 3), lo + (hi - lo) * rng.uniform(0.5, 1.0, 3)]))
        q0, q1 = t0.clone(), t1.clone()
        pkg.fill_grid_pass(other, g, 1, q0, q1, changed_box=box, sdf_id=sdf_id)
        pkg.fill_grid_pass(other, g, 1, p0, p1, changed_box=box, sdf_id=sdf_id, dist=dvol)
        torch.cuda.synchronize()
        assert torch.equal(p0, q0) and torch.equal(p1, q1) and torch.equal(dvol, p0[..., 0]), (trial, kw, dims, box)
        # an edit whose box holds every voxel (the store-only path the library picks itself), strided then step 1, against
        # the dense fill with the new parameters
        whole = tuple(float(v) for v in np.concatenate([np.minimum(lo, hi) - 0.5, np.maximum(lo, hi) + 0.5]))
        for step in (2, 1):
            pkg.fill_grid_pass(other, g, step, p0, p1, changed_box=whole, sdf_id=sdf_id, dist=dvol)
        torch.cuda.synchronize()
        o0, o1 = gpu_fill(pkg, other, dims, lo, hi, sdf_id=sdf_id)
        assert torch.equal(p0, o0) and torch.equal(p1, o1) and torch.equal(dvol, o0[..., 0]), (trial, kw, dims, "whole box")
        # round 4's forms: the same load over a VIRGIN grid (undefined bytes, nothing initialised), plain or y-interleaved
        # volume, random steps, its lazily initialised intermediate states against the plain path's; then the other
        # Srgba::from policy against the oracle evaluating the same one
        ilv = K.PASS_VOLUME_INTERLEAVED if trial % 3 == 0 and dims[1] % 2 == 0 else 0  # (the layout pairs rows: even heights)
        p0.fill_(float("nan"))
        p1.fill_(-3.0)
        dvol.fill_(float("nan"))
        a0, a1 = pkg.alloc_textures(g)
        pkg.grid_init(g, a0, a1)
        for step in steps:
            pkg.fill_grid_pass(prm, g, step, p0, p1, sdf_id=sdf_id, dist=dvol if trial % 4 else None,
                               flags=K.PASS_VIRGIN_GRID | K.PASS_SAME_LOAD | (ilv if trial % 4 else 0))
            pkg.fill_grid_pass(prm, g, step, a0, a1, sdf_id=sdf_id)
            if step > 1:
                c0, c1, cv = p0.clone(), p1.clone(), dvol.clone()
                pkg.grid_init_unvisited(g, step, c0, c1, dist=cv if trial % 4 else None, flags=ilv if trial % 4 else 0)
                torch.cuda.synchronize()
                assert torch.equal(c0, a0) and torch.equal(c1, a1), (trial, kw, dims, steps, step, "virgin state")
                if trial % 4:
                    assert torch.equal(cv, interleave_rows(a0[..., 0]) if ilv else a0[..., 0]), (trial, dims, steps, step, "virgin volume")
        torch.cuda.synchronize()
        assert torch.equal(p0, t0) and torch.equal(p1, t1), (trial, kw, dims, steps, "virgin load")
        if trial % 4:
            assert torch.equal(dvol, interleave_rows(t0[..., 0]) if ilv else t0[..., 0]), (trial, dims, steps, "virgin load's volume")
        # round 5: a load the caller says NOTHING about, over a volume in either layout, on a width the whole-rows kernel of
        # step >= 2 takes (W % 4 == 0: four random dimensions are widened to it) -- every intermediate state against the
        # plain path's (the per-voxel kernels over tex0.r)
        dims5 = (dims[0] + (-dims[0]) % 4, dims[1] + (dims[1] % 2 if trial % 3 == 0 else 0), dims[2])
        ilv5 = K.PASS_VOLUME_INTERLEAVED if trial % 3 == 0 else 0
        g5 = pkg.make_grid(dims5, lo, hi)
        u0, u1 = pkg.alloc_textures(g5)
        w0, w1 = pkg.alloc_textures(g5)
        pkg.grid_init(g5, u0, u1)
        pkg.grid_init(g5, w0, w1)
        v5 = torch.full(tuple(u0.shape[:-1]), pkg.AIR_DIST, dtype=torch.float32, device="cuda")
        steps5 = [int(2 ** rng.integers(0, 4)) for _ in range(3)] + [1]  # any order, coarse after fine included
        for step in steps5:
            pkg.fill_grid_pass(prm, g5, step, u0, u1, sdf_id=sdf_id, dist=v5, flags=ilv5)
            pkg.fill_grid_pass(prm, g5, step, w0, w1, sdf_id=sdf_id)
            torch.cuda.synchronize()
            assert torch.equal(u0.view(torch.int32), w0.view(torch.int32)) and torch.equal(u1.view(torch.int32), w1.view(torch.int32)), (trial, kw, dims5, steps5, step, "unflagged")
            assert torch.equal(v5, interleave_rows(w0[..., 0]) if ilv5 else w0[..., 0]), (trial, dims5, steps5, step, "unflagged volume")
        del u0, u1, w0, w1, v5
        if trial % 2 == 0:
            flag = oracle.EXT_VARIANTS["srgb_quant_round"]
            before = oracle.L.or_get_ext_variant()
            oracle.L.or_set_ext_variant(before | flag)
            try:
                with pkg.options({K.OPT_EXT_SRGB_QUANT: 1}):
                    e0, e1 = gpu_fill(pkg, prm, dims, lo, hi, sdf_id=sdf_id)
                    for step in steps:
                        pkg.fill_grid_pass(prm, g, step, p0, p1, sdf_id=sdf_id, flags=K.PASS_VIRGIN_GRID | K.PASS_SAME_LOAD)
                    torch.cuda.synchronize()
                x0, x1 = oracle.fill_dense(oracle.params_from(prm), dims, lo, hi, sdf_id=sdf_id, threads=2)
            finally:
                oracle.L.or_set_ext_variant(before)
            assert_bits_equal(e0, x0)
            assert_bits_equal(e1, x1)
            assert torch.equal(p0, e0) and torch.equal(p1, e1), (trial, kw, dims, steps, "rounding policy, virgin load")


@pytest.mark.parametrize("dims", [(64, 64, 64), (33, 5, 70), (130, 3, 9), (1, 5, 7), (256, 8, 4)])
def test_fused_fill_and_commit_writes_the_same_distance_volume(pkg, oracle, dims):
    """sdfv_fill_grid_commit = sdfv_fill_grid + sdfv_commit_distance in one pass (both index forms, slab offsets)."""
    prm = pkg.default_params()
    z0 = dims[2] // 3
    g = pkg.make_grid(dims, z_begin=z0, z_end=dims[2])
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.full(tuple(t0.shape[:-1]), -7.0, dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    torch.cuda.synchronize()
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, z0=z0, z1=dims[2])
    assert_bits_equal(t0, r0)
    assert_bits_equal(t1, r1)
    assert torch.equal(dist, t0[..., 0]) and torch.equal(dist, pkg.commit_distance(g, t0))


@pytest.mark.parametrize("dims,z_range", [((37, 20, 29), (0, 29)), ((40, 12, 29), (0, 29)), ((40, 12, 29), (7, 22)),
                                          ((8, 3, 5), (1, 4))])
def test_progressive_passes_over_the_distance_volume(pkg, oracle, dims, z_range):
    """sdfv_fill_grid_pass_dist: same texels as the plain pass after every pass (fresh load, then a changed_box refill
    with other parameters), the volume stays equal to tex0.r, tex1.a stays AIR_DIST.  Widths that take the quad
    kernel (W % 4 == 0) and widths that do not, whole grids and z-slabs."""
    g = pkg.make_grid(dims, (-1.0, -0.75, -1.0), (1.0, 1.0, 0.5), *z_range)
    prm = pkg.default_params()
    a0, a1 = pkg.alloc_textures(g)
    b0, b1 = pkg.alloc_textures(g)
    pkg.grid_init(g, a0, a1)
    pkg.grid_init(g, b0, b1)
    dist = pkg.commit_distance(g, b0)  # the volume of a freshly initialised grid: AIR_DIST everywhere
    assert bool((dist == pkg.AIR_DIST).all())
    for step in (4, 2, 1):
        pkg.fill_grid_pass(prm, g, step, a0, a1)
        pkg.fill_grid_pass(prm, g, step, b0, b1, dist=dist)
        torch.cuda.synchronize()
        assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(dist, b0[..., 0])
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, (-1.0, -0.75, -1.0), (1.0, 1.0, 0.5), *z_range)
    assert_bits_equal(b0, r0)
    assert_bits_equal(b1, r1)
    edited = pkg.default_params(sphere_radius=0.8, cube_material=1)
    box = (-0.6, -0.5, -0.7, 0.3, 0.9, 0.2)
    for step in (4, 2, 1):
        pkg.fill_grid_pass(edited, g, step, a0, a1, changed_box=box)
        pkg.fill_grid_pass(edited, g, step, b0, b1, changed_box=box, dist=dist)
        torch.cuda.synchronize()
        assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(dist, b0[..., 0])
    assert not torch.equal(b0, torch.from_numpy(r0).cuda())  # the edit did change voxels inside the box
    assert bool((b1[..., 3] == pkg.AIR_DIST).all())


def test_changed_box_whose_faces_lie_exactly_on_voxel_coordinates(pkg, oracle):
    """The pass kernel leaves early for voxels that are clearly outside the box (a cheap coordinate estimate with a
    proven margin) and decides the rest with the exact coordinates: a box bounded by exact voxel coordinates, where
    `>=` / `<=` (scene/sdf/mod.rs:186-188) flip from one voxel to the next, must refill exactly the oracle's voxels."""
    prm, edited = pkg.default_params(), pkg.default_params(cube_half_side=0.5, sphere_radius=0.6)
    bb = ((-1.0, -0.9, -1.3), (0.7, 1.0, 0.4))
    # (24 / 256 wide: the step-1 pass over a volume is the quad kernel, whose waves leave on ONE estimate per lane -- its quad's
    # ends against the box -- since round 5: boxes that hold a single voxel INSIDE a quad, or no voxel at all, must still match)
    for dims in ((23, 31, 19), (24, 31, 19), (256, 9, 6)):
        def coord(i, axis):
            return float(oracle.L.or_voxel_coord(i, dims[axis], bb[0][axis], bb[1][axis]))
        mid = lambda i, axis: 0.5 * (coord(i, axis) + coord(i + 1, axis))  # noqa: E731
        boxes = [(coord(5, 0), coord(7, 1), coord(3, 2), coord(17, 0), coord(min(22, dims[1] - 1), 1), coord(min(11, dims[2] - 1), 2)),
                 (mid(5, 0), coord(2, 1), coord(1, 2), mid(6, 0), coord(4, 1), coord(3, 2)),      # x = 6 alone: inside the quad 4..7
                 (mid(9, 0), coord(2, 1), coord(1, 2), 0.5 * (mid(9, 0) + coord(10, 0)), coord(4, 1), coord(3, 2)),  # between voxels: nothing
                 (coord(0, 0), coord(0, 1), coord(2, 2), coord(dims[0] - 1, 0), coord(0, 1), coord(2, 2))]  # one whole row
        for box in boxes:
            check_boxed_edit(pkg, oracle, dims, bb, prm, edited, box, expect_change=box is not boxes[2])


def check_boxed_edit(pkg, oracle, dims, bb, prm, edited, box, expect_change):
    g = pkg.make_grid(dims, *bb)
    for use_dist in (False, True):
        t0, t1 = pkg.alloc_textures(g)
        dist = torch.empty(tuple(t0.shape[:-1]), dtype=torch.float32, device="cuda") if use_dist else None
        pkg.fill_grid(prm, g, t0, t1, dist=dist)
        pkg.fill_grid_pass(edited, g, 1, t0, t1, changed_box=box, dist=dist)
        torch.cuda.synchronize()
        # oracle: dense fill with the old parameters, then one viewer update pass with the new ones and the box
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, *bb)
        lm = oracle.lm_new(dims, 1)
        oracle.viewer_update(oracle.params_from(edited), dims, lm, r0, r1, changed_box=box, bb_min=bb[0], bb_max=bb[1])
        assert_bits_equal(t0, r0)
        assert_bits_equal(t1, r1)
        n_changed = int((t0.cpu().numpy() != oracle.fill_dense(oracle.params_from(prm), dims, *bb)[0]).any(axis=-1).sum())
        assert (n_changed > 0) == expect_change, (dims, box, n_changed)
        if use_dist:
            assert torch.equal(dist, t0[..., 0])


def test_placed_textures(pkg, oracle):
    """alloc_textures_placed = what SDFViewer::new_voxels allocates: one block, tex1 at the fixed distance for that byte size
    (16-byte aligned, no overlap); a fill into the pair gives the usual texels.  No probe entry point is exported any more."""
    dims = (64, 48, 40)
    g = pkg.make_grid(dims)
    n_bytes = dims[0] * dims[1] * dims[2] * 16
    t0, t1 = pkg.alloc_textures_placed(g)
    assert t1.data_ptr() - t0.data_ptr() - n_bytes == pkg.default_texture_skew(n_bytes) == 0
    assert t0.shape == t1.shape == (dims[2], dims[1], dims[0], 4) and t0.data_ptr() % 16 == 0 and t1.data_ptr() % 16 == 0
    assert pkg.default_texture_skew(1 << 28) == 12288 and pkg.default_texture_skew(1 << 30) == 20480
    prm = pkg.default_params(cube_material=1)
    pkg.fill_grid(prm, g, t0, t1)
    torch.cuda.synchronize()
    r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims)
    assert_bits_equal(t0, r0)
    assert_bits_equal(t1, r1)
    assert not hasattr(pkg.lib, "sdfv_tune_texture_placement")


@pytest.mark.parametrize("dims,z_range", [((64, 32, 16), (0, 16)), ((37, 20, 29), (0, 29)), ((40, 12, 29), (7, 22)),
                                          ((8, 3, 5), (1, 4)), ((256, 8, 6), (0, 6)), ((130, 5, 9), (2, 9))])
@pytest.mark.parametrize("use_dist", [False, True])
def test_fresh_load_with_the_callers_knowledge_reads_nothing(pkg, oracle, dims, z_range, use_dist):
    """sdfv_fill_grid_pass_ex: the first pass of a load flagged SDFV_PASS_FRESH_GRID (writes the visited rows whole: the
    samples and, between them, the AIR texels the fresh grid holds), the later ones SDFV_PASS_SAME_LOAD (store-only; step 1
    = the dense fill).  Every intermediate state -- both textures and the volume -- equals the unflagged passes' and the
    oracle's LoadingManager loop stopped at the pass boundary; non-default parameters too (RuntimeCfg kernels)."""
    K = pkg._capi
    bb = ((-1.0, -0.75, -1.0), (1.0, 1.0, 0.5))
    g = pkg.make_grid(dims, *bb, *z_range)
    for prm in (pkg.default_params(), pkg.default_params(cube_material=1, sphere_radius=0.8, disable_sphere=0)):
        for steps in ((4, 2, 1), (2, 1), (8, 1), (1,)):
            a0, a1 = pkg.alloc_textures(g)
            b0, b1 = pkg.alloc_textures(g)
            pkg.grid_init(g, a0, a1)
            pkg.grid_init(g, b0, b1)
            da = pkg.commit_distance(g, a0) if use_dist else None
            db = pkg.commit_distance(g, b0) if use_dist else None
            for k, step in enumerate(steps):
                pkg.fill_grid_pass(prm, g, step, a0, a1, dist=da)
                pkg.fill_grid_pass(prm, g, step, b0, b1, dist=db,
                                   flags=(K.PASS_FRESH_GRID | K.PASS_SAME_LOAD) if k == 0 else K.PASS_SAME_LOAD)
                torch.cuda.synchronize()
                assert torch.equal(a0.view(torch.int32), b0.view(torch.int32)), (steps, step)
                assert torch.equal(a1.view(torch.int32), b1.view(torch.int32)), (steps, step)
                if use_dist:
                    assert torch.equal(db, b0[..., 0]) and torch.equal(da, db), (steps, step)
            r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, *bb, *z_range)
            assert_bits_equal(b0, r0)
            assert_bits_equal(b1, r1)


def test_fresh_flagged_load_against_the_oracles_loading_manager(pkg, oracle):
    """The flagged passes against the ORACLE's loop (not just against the unflagged kernels) at every pass boundary."""
    K = pkg._capi
    prm = pkg.default_params()
    oprm = oracle.params_from(prm)
    dims = (24, 18, 13)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    r0, r1 = oracle.grid_init(dims)
    lm = oracle.lm_new(dims, 3)
    for k, step in enumerate((4, 2, 1)):
        pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist, flags=(K.PASS_FRESH_GRID if k == 0 else 0) | K.PASS_SAME_LOAD)
        n = -(-dims[0] // step) * -(-dims[1] // step) * -(-dims[2] // step)
        assert oracle.viewer_update(oprm, dims, lm, r0, r1, max_iterations=n) == n
        torch.cuda.synchronize()
        assert_bits_equal(t0, r0)
        assert_bits_equal(t1, r1)
        assert torch.equal(dist, t0[..., 0])


@pytest.mark.parametrize("use_dist", [False, True])
def test_changed_box_that_contains_every_voxel_takes_the_store_only_path(pkg, oracle, use_dist):
    """A parameter edit whose changed box is the whole bounding box (what the demo reports, demo/mod.rs:135-144): the library
    sees that the box contains every voxel coordinate and reads nothing.  Every pass boundary against the oracle's loop with
    the same box; then a box that misses the LAST voxel of each axis by one ulp (general path), same check."""
    dims = (32, 20, 24)
    bb = ((-1.0, -0.9, -1.3), (0.7, 1.0, 0.4))
    g = pkg.make_grid(dims, *bb)
    prm, edited = pkg.default_params(), pkg.default_params(cube_half_side=0.5, sphere_radius=0.6)
    whole = bb[0] + bb[1]
    short = bb[0] + tuple(float(np.nextafter(np.float32(v), np.float32(-10.0))) for v in bb[1])
    for box in (whole, short):
        t0, t1 = pkg.alloc_textures(g)
        dist = torch.empty(tuple(t0.shape[:-1]), dtype=torch.float32, device="cuda") if use_dist else None
        pkg.fill_grid(prm, g, t0, t1, dist=dist)
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, *bb)
        lm = oracle.lm_new(dims, 3)
        for step in (4, 2, 1):
            pkg.fill_grid_pass(edited, g, step, t0, t1, changed_box=box, dist=dist)
            n = -(-dims[0] // step) * -(-dims[1] // step) * -(-dims[2] // step)
            oracle.viewer_update(oracle.params_from(edited), dims, lm, r0, r1, changed_box=box, max_iterations=n,
                                 bb_min=bb[0], bb_max=bb[1])
            torch.cuda.synchronize()
            assert_bits_equal(t0, r0)
            assert_bits_equal(t1, r1)
            if use_dist:
                assert torch.equal(dist, t0[..., 0])
    # the short box really left the last voxels alone: they still hold the first parameters' texels
    first = oracle.fill_dense(oracle.params_from(prm), dims, *bb)[0]
    assert (t0.cpu().numpy()[-1, -1, -1].view(np.uint32) == first[-1, -1, -1].view(np.uint32)).all()


def test_full_size_progressive_load_states(pkg, oracle):
    """256^3: the reference's default 2-pass load through the flagged passes (what SDFViewer::update enqueues), the
    intermediate state against the oracle's loop, the final one against the dense fill."""
    K = pkg._capi
    side = 256
    dims = (side,) * 3
    prm = pkg.default_params()
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    pkg.fill_grid_pass(prm, g, 2, t0, t1, dist=dist, flags=K.PASS_FRESH_GRID | K.PASS_SAME_LOAD)
    torch.cuda.synchronize()
    r0, r1 = oracle.grid_init(dims)
    lm = oracle.lm_new(dims, 2)
    oracle.viewer_update(oracle.params_from(prm), dims, lm, r0, r1, max_iterations=(side // 2) ** 3)
    assert_bits_equal(t0, r0)
    assert_bits_equal(t1, r1)
    assert torch.equal(dist, t0[..., 0])
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist, flags=K.PASS_SAME_LOAD)
    torch.cuda.synchronize()
    d0, d1 = gpu_fill(pkg, prm, dims)
    assert torch.equal(t0, d0) and torch.equal(t1, d1) and torch.equal(dist, d0[..., 0])
    # and the unflagged passes (update_required read from the volume) from the same start
    pkg.grid_init(g, t0, t1)
    pkg.commit_distance(g, t0, dist=dist)
    for step in (2, 1):
        pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist)
    torch.cuda.synchronize()
    assert torch.equal(t0, d0) and torch.equal(t1, d1) and torch.equal(dist, d0[..., 0])


def test_store_only_passes_leave_a_foreign_tex1_alpha_alone(pkg, oracle):
    """update() never writes tex1.a (scene/sdf/mod.rs:205-208).  Grids of this library hold AIR_DIST there, but a pass WITHOUT
    the distance volume makes no such assumption: also on the store-only paths (changed box containing the grid, same-load
    flag) a value the caller put into tex1.a survives, at every step."""
    K = pkg._capi
    dims = (32, 16, 12)
    g = pkg.make_grid(dims)
    prm, edited = pkg.default_params(), pkg.default_params(sphere_radius=0.7)
    whole = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)
    for kwargs in (dict(changed_box=whole), dict(flags=K.PASS_SAME_LOAD)):
        t0, t1 = gpu_fill(pkg, prm, dims)
        t1[..., 3] = 42.0
        for step in (4, 2, 1):
            pkg.fill_grid_pass(edited, g, step, t0, t1, **kwargs)
        torch.cuda.synchronize()
        r0, r1 = oracle.fill_dense(oracle.params_from(edited), dims)
        assert_bits_equal(t0, r0)
        assert bool((t1[..., 3] == 42.0).all())
        np.testing.assert_array_equal(t1.cpu().numpy()[..., :3].view(np.uint32), r1[..., :3].view(np.uint32))


@pytest.mark.parametrize("dims,z_range,steps", [((21, 18, 13), (0, 13), (4, 2, 1)), ((64, 64, 64), (0, 64), (2, 1)),
                                                ((40, 12, 29), (7, 22), (16, 4, 2, 1)), ((130, 9, 6), (1, 6), (2, 1)),
                                                ((256, 16, 8), (0, 8), (8, 1))])
@pytest.mark.parametrize("use_dist", [False, True])
def test_virgin_grid_passes(pkg, oracle, dims, z_range, steps, use_dist):
    """SDFV_PASS_VIRGIN_GRID: a load over a fresh ALLOCATION (new_voxels' [AIR_DIST; 4] never written; the buffers hold
    garbage).  After every pass the rows it visited are the reference's, sdfv_grid_init_unvisited(step) completes a copy to
    the oracle's whole state at that pass boundary, and the step-1 pass leaves the dense grid with nothing undefined."""
    K = pkg._capi
    bb = ((-1.0, -0.75, -1.0), (1.0, 1.0, 0.5))
    g = pkg.make_grid(dims, *bb, *z_range)
    prm = pkg.default_params()
    t0, t1 = pkg.alloc_textures(g)
    t0.fill_(-7.0)
    t1.fill_(float("nan"))
    dist = torch.full(tuple(t0.shape[:-1]), 123.0, dtype=torch.float32, device="cuda") if use_dist else None
    # the oracle: the LoadingManager loop over the WHOLE grid, cut to the slab afterwards
    r0, r1 = oracle.grid_init(dims)
    z0, z1 = z_range
    # before any pass: step 0 = plain new_voxels
    c0, c1 = t0.clone(), t1.clone()
    cd = dist.clone() if use_dist else None
    pkg.grid_init_unvisited(g, 0, c0, c1, dist=cd)
    assert bool((c0 == pkg.AIR_DIST).all()) and bool((c1 == pkg.AIR_DIST).all()) and (cd is None or bool((cd == pkg.AIR_DIST).all()))
    for step in steps:
        pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist, flags=K.PASS_VIRGIN_GRID | K.PASS_SAME_LOAD)
        lm = oracle.lm_new(dims, 1)
        lm.step_size = step  # one pass with this step over the oracle's grid
        oracle.viewer_update(oracle.params_from(prm), dims, lm, r0, r1, bb_min=bb[0], bb_max=bb[1],
                             max_iterations=-(-dims[0] // step) * -(-dims[1] // step) * -(-dims[2] // step))
        torch.cuda.synchronize()
        ys = torch.arange(0, dims[1], step, device="cuda")
        zs = [z - z0 for z in range(z0, z1) if z % step == 0]
        if zs:
            rows0 = t0[zs][:, ys].cpu().numpy()
            rows1 = t1[zs][:, ys].cpu().numpy()
            np.testing.assert_array_equal(rows0.view(np.uint32), r0[z0:z1][zs][:, ys.cpu().numpy()].view(np.uint32))
            np.testing.assert_array_equal(rows1.view(np.uint32), r1[z0:z1][zs][:, ys.cpu().numpy()].view(np.uint32))
        c0, c1 = t0.clone(), t1.clone()
        cd = dist.clone() if use_dist else None
        pkg.grid_init_unvisited(g, step, c0, c1, dist=cd)
        torch.cuda.synchronize()
        assert_bits_equal(c0, r0[z0:z1])
        assert_bits_equal(c1, r1[z0:z1])
        assert cd is None or torch.equal(cd, c0[..., 0])
    d0, d1 = gpu_fill(pkg, prm, dims, *bb, *z_range)
    assert torch.equal(t0, d0) and torch.equal(t1, d1) and (dist is None or torch.equal(dist, t0[..., 0]))
    with pytest.raises(pkg.SdfvError):  # a box test reads the grid
        pkg.fill_grid_pass(prm, g, 2, t0, t1, changed_box=(-1, -1, -1, 1, 1, 1), flags=K.PASS_VIRGIN_GRID)


def test_passes_over_slabs_beyond_32_bit_indices_run_in_pieces(pkg, oracle):
    """a pass over a slab of >= 2^32 voxels used to fail.  It now runs as several launches over pieces of whole
    slices; SDFV_OPT_PASS_INDEX_LIMIT lowers the threshold so that a small grid takes that path: unflagged, flagged, with a
    changed box, with and without the distance volume, whole grids and slabs -- same texels as in one piece."""
    K = pkg._capi
    dims = (24, 20, 23)
    prm, edited = pkg.default_params(), pkg.default_params(sphere_radius=0.8, cube_material=1)
    box = (-0.5, -1.0, -0.25, 0.5, 0.1, 1.0)
    for z_range in ((0, 23), (5, 20)):
        g = pkg.make_grid(dims, z_begin=z_range[0], z_end=z_range[1])
        for use_dist in (False, True):
            for flagged in (False, True):
                results = []
                for limit in (0, 24 * 20 * 3 + 1, 24 * 20 + 1):  # one piece, pieces of 3 slices, pieces of 1 slice
                    with pkg.options({K.OPT_PASS_INDEX_LIMIT: limit}):
                        t0, t1 = pkg.alloc_textures(g)
                        pkg.grid_init(g, t0, t1)
                        dist = pkg.commit_distance(g, t0) if use_dist else None
                        for k, step in enumerate((4, 2, 1)):
                            flags = ((K.PASS_FRESH_GRID if k == 0 else 0) | K.PASS_SAME_LOAD) if flagged else 0
                            pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist, flags=flags)
                        a0, a1 = t0.clone(), t1.clone()
                        for step in (4, 2, 1):
                            pkg.fill_grid_pass(edited, g, step, t0, t1, changed_box=box, dist=dist)
                        torch.cuda.synchronize()
                        assert dist is None or torch.equal(dist, t0[..., 0])
                        results.append((a0, a1, t0, t1))
                for r in results[1:]:
                    assert all(torch.equal(x, y) for x, y in zip(results[0], r)), (z_range, use_dist, flagged)
        r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, z0=z_range[0], z1=z_range[1])
        assert_bits_equal(results[0][0], r0)
    with pytest.raises(pkg.SdfvError):
        pkg.set_option(K.OPT_PASS_INDEX_LIMIT, 1)


def interleave_rows(d):
    """[D, H, W] -> the y-interleaved volume's layout as a [D, H, W]-shaped tensor (rows 2p, 2p + 1 as one row of pairs)."""
    D, H, W = d.shape
    return torch.stack([d[:, 0::2], d[:, 1::2]], dim=-1).reshape(D, H, W).contiguous()


@pytest.mark.parametrize("dims,z_range", [((64, 64, 64), (0, 64)), ((128, 6, 5), (0, 5)), ((256, 4, 7), (2, 7)), ((300, 10, 3), (0, 3)),
                                          ((40, 12, 29), (7, 22)), ((20, 18, 12), (0, 12)), ((512, 2, 2), (0, 2)),
                                          ((768, 4, 3), (1, 3)), ((512, 6, 4), (0, 4))])
def test_fill_and_passes_maintain_the_interleaved_volume(pkg, oracle, dims, z_range):
    """SDFV_PASS_VOLUME_INTERLEAVED: the volume every fill and pass writes / reads is laid out as the march's y-interleaved
    volume.  Dense fused fill (= a step-1 virgin pass), the virgin chain with its lazy initialisation, unflagged passes over an
    initialised grid (general + quad kernels), a boxed edit, a whole-box edit (copy-through rows), the piecewise path: the
    textures equal the plain path's bit for bit and the volume is interleave(tex0.r) after every step."""
    K = pkg._capi
    ILV = K.PASS_VOLUME_INTERLEAVED
    g = pkg.make_grid(dims, z_begin=z_range[0], z_end=z_range[1])
    prm, edited = pkg.default_params(), pkg.default_params(sphere_radius=0.8, cube_material=1)
    want0, want1 = gpu_fill(pkg, prm, dims, z0=z_range[0], z1=z_range[1])

    def check(t0, t1, vol, ref0=None, ref1=None):
        torch.cuda.synchronize()
        if ref0 is not None:
            assert torch.equal(t0, ref0) and torch.equal(t1, ref1)
        assert torch.equal(vol, interleave_rows(t0[..., 0]))

    t0, t1 = pkg.alloc_textures(g)
    t0.fill_(-7.0)
    t1.fill_(-7.0)
    vol = torch.full(tuple(t0.shape[:-1]), -7.0, dtype=torch.float32, device="cuda")
    # (1) the dense fused fill that writes the interleaved volume itself
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | ILV)
    check(t0, t1, vol, want0, want1)
    assert torch.equal(vol, pkg.commit_interleaved(pkg.make_grid((dims[0], dims[1], z_range[1] - z_range[0])), t0[..., 0].contiguous()))
    # widths of two or more workgroups take the form with one row per workgroup and the rows of a pair on one XCD (round 5; a
    # thread per x of BOTH rows of a pair without the eight-XCD placement); SDFV_OPT_FILL_FORM pins each form (1 = the pair meets
    # in LDS), both store forms, the run-time configuration too: the same bits
    for nt in (1, 2):
        for p_ in (prm, pkg.default_params(cube_material=1, sphere_material=0)):
            ref0, ref1 = gpu_fill(pkg, p_, dims, z0=z_range[0], z1=z_range[1])
            for form in (0, 1, 3, 4):  # auto | LDS row-chunk form | one row per workgroup, pairs on one XCD | a thread per pair
                t0.fill_(-7.0)
                vol.fill_(-7.0)
                with pkg.options({K.OPT_FILL_FORM: form, K.OPT_FILL_NONTEMPORAL: nt}):
                    pkg.fill_grid_pass(p_, g, 1, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | ILV)
                check(t0, t1, vol, ref0, ref1)
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | ILV)
    # (2) the virgin chain + lazy initialisation
    t0.fill_(-7.0)
    t1.fill_(float("nan"))
    vol.fill_(123.0)
    a0, a1 = pkg.alloc_textures(g)
    pkg.grid_init(g, a0, a1)
    for step in (4, 2, 1):
        pkg.fill_grid_pass(prm, g, step, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | K.PASS_SAME_LOAD | ILV)
        pkg.fill_grid_pass(prm, g, step, a0, a1)  # the plain path over an initialised grid: the reference state at this boundary
        c0, c1, cv = t0.clone(), t1.clone(), vol.clone()
        pkg.grid_init_unvisited(g, step, c0, c1, dist=cv, flags=ILV)
        check(c0, c1, cv, a0, a1)
    check(t0, t1, vol, want0, want1)
    # (3) unflagged passes over an initialised grid, both fresh-flagged and not, then edits
    for flags in (0, None):
        pkg.grid_init(g, t0, t1)
        vol.fill_(pkg.AIR_DIST)
        for k, step in enumerate((4, 2, 1)):
            f = ILV | (0 if flags == 0 else ((K.PASS_FRESH_GRID if k == 0 else 0) | K.PASS_SAME_LOAD))
            pkg.fill_grid_pass(prm, g, step, t0, t1, dist=vol, flags=f)
            check(t0, t1, vol)
        check(t0, t1, vol, want0, want1)
    box = (-0.5, -1.0, -0.25, 0.5, 0.1, 1.0)
    b0, b1 = want0.clone(), want1.clone()
    for step in (4, 2, 1):
        pkg.fill_grid_pass(edited, g, step, t0, t1, changed_box=box, dist=vol, flags=ILV)
        pkg.fill_grid_pass(edited, g, step, b0, b1, changed_box=box)
        check(t0, t1, vol, b0, b1)
    whole = (-2, -2, -2, 2, 2, 2)
    with pkg.options({K.OPT_PASS_INDEX_LIMIT: dims[0] * dims[1] * 2 + 1}):  # ... in pieces of two slices
        for step in (4, 2, 1):
            pkg.fill_grid_pass(prm, g, step, t0, t1, changed_box=whole, dist=vol, flags=ILV)
            pkg.fill_grid_pass(prm, g, step, b0, b1, changed_box=whole)
            check(t0, t1, vol, b0, b1)
    check(t0, t1, vol, want0, want1)


@pytest.mark.parametrize("dims,z_range", [((64, 64, 64), (0, 64)), ((256, 6, 10), (0, 10)), ((40, 12, 29), (7, 22)), ((300, 10, 6), (0, 6)),
                                          ((1024, 4, 4), (0, 4)), ((8, 6, 5), (0, 5)), ((132, 9, 7), (1, 6))])
@pytest.mark.parametrize("ilv", [False, True])
def test_unflagged_strided_passes_decide_per_wave(pkg, oracle, dims, z_range, ilv):
    """a step >= 2 pass the caller says nothing about (no flags, no box) over a grid with its volume takes
    fill_pass_rows_adaptive_kernel -- whole visited rows, every wave deciding on the volume it reads.  Against the per-voxel
    kernels (no volume; SDFV_OPT_PASS_FORM 1) after EVERY pass, textures and volume bit for bit: a fresh grid (every round
    written whole), the descending chain (rows a coarser pass visited hold samples: mixed rounds), a grid loaded in its lower
    half only, a loaded grid (nothing to do), other parameters over a loaded grid (still nothing: no box); then the oracle."""
    K = pkg._capi
    if ilv and dims[1] % 2:
        pytest.skip("the interleaved volume pairs rows: H must be even")
    flags = K.PASS_VOLUME_INTERLEAVED if ilv else 0
    lo, hi = (-1.0, -0.75, -1.0), (1.0, 1.0, 0.5)
    g = pkg.make_grid(dims, lo, hi, *z_range)
    prm, other = pkg.default_params(), pkg.default_params(sphere_radius=0.8, cube_material=1)
    layout = interleave_rows if ilv else (lambda d: d)

    def fresh():
        t0, t1 = pkg.alloc_textures(g)
        pkg.grid_init(g, t0, t1)
        return t0, t1, torch.full(tuple(t0.shape[:-1]), pkg.AIR_DIST, dtype=torch.float32, device="cuda")

    def same(a, b, what):
        torch.cuda.synchronize()
        assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)), what
        assert torch.equal(a[2].view(torch.int32), layout(a[0][..., 0]).view(torch.int32)), (what, "volume")
        assert torch.equal(b[2].view(torch.int32), layout(b[0][..., 0]).view(torch.int32)), (what, "reference volume")

    for steps in ((8, 4, 2, 1), (2, 1), (4,), (2, 4, 2)):
        a, b = fresh(), fresh()
        for k, step in enumerate(steps):
            pkg.fill_grid_pass(prm, g, step, a[0], a[1], dist=a[2], flags=flags)
            with pkg.options({K.OPT_PASS_FORM: 1}):
                pkg.fill_grid_pass(prm, g, step, b[0], b[1], dist=b[2], flags=flags)
            same(a, b, (steps, k))
        if steps[-1] == 1:
            r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, lo, hi, *z_range)
            assert_bits_equal(a[0], r0)
            assert_bits_equal(a[1], r1)
    # the lower half loaded (a dense fill of those slices), the upper half fresh
    a, b = fresh(), fresh()
    half = max(1, (z_range[1] - z_range[0]) // 2)
    gh = pkg.make_grid(dims, lo, hi, z_range[0], z_range[0] + half)
    for t in (a, b):
        v = torch.empty(tuple(t[0][:half].shape[:-1]), dtype=torch.float32, device="cuda")
        pkg.fill_grid(other, gh, t[0][:half], t[1][:half], dist=v)
        t[2][:half] = layout(v) if (not ilv) else interleave_rows(v)
    for step in (4, 2):
        pkg.fill_grid_pass(prm, g, step, a[0], a[1], dist=a[2], flags=flags)
        with pkg.options({K.OPT_PASS_FORM: 1}):
            pkg.fill_grid_pass(prm, g, step, b[0], b[1], dist=b[2], flags=flags)
        same(a, b, ("half loaded", step))
    # a loaded grid: nothing is AIR, nothing happens -- whatever the parameters
    pkg.fill_grid_pass(prm, g, 1, a[0], a[1], dist=a[2], flags=flags)
    before = (a[0].clone(), a[1].clone(), a[2].clone())
    for step in (8, 2):
        pkg.fill_grid_pass(other, g, step, a[0], a[1], dist=a[2], flags=flags)
    same(a, before, "loaded grid")


def test_interleaved_volume_argument_checks_and_the_march_over_it(pkg, oracle):
    K = pkg._capi
    prm = pkg.default_params()
    odd = pkg.make_grid((16, 7, 4))
    t0, t1 = pkg.alloc_textures(odd)
    vol = torch.empty(tuple(t0.shape[:-1]), dtype=torch.float32, device="cuda")
    with pytest.raises(pkg.SdfvError):  # rows are paired: H must be even
        pkg.fill_grid_pass(prm, odd, 1, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | K.PASS_VOLUME_INTERLEAVED)
    with pytest.raises(pkg.SdfvError):  # the layout bit needs a volume
        pkg.fill_grid_pass(prm, odd, 1, t0, t1, flags=K.PASS_VOLUME_INTERLEAVED)
    # the volume the fill wrote IS the march's ilv volume: same image as over the distance volume, bit for bit
    g = pkg.make_grid((64, 64, 64))
    t0, t1 = pkg.alloc_textures(g)
    vol = torch.empty((64, 64, 64), dtype=torch.float32, device="cuda")
    pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=vol, flags=K.PASS_VIRGIN_GRID | K.PASS_VOLUME_INTERLEAVED)
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at(aspect=1.5)
    dist = pkg.commit_distance(g, t0)
    a = pkg.raymarch(rp, t0, t1, cam, 192, 128, ilv=vol)
    b = pkg.raymarch(rp, t0, t1, cam, 192, 128, dist=dist)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and bool((a[..., 3] > 0).any())


@pytest.mark.parametrize("dims,ilv", [((64, 12, 9), False), ((40, 12, 29), True), ((37, 20, 29), False), ((256, 8, 6), True)])
def test_scan_loads_and_the_noop_hint_change_nothing_but_speed(pkg, oracle, dims, ilv):
    """SDFV_PASS_EXPECT_NOOP is a HINT and SDFV_OPT_PASS_LOADS an A/B switch: however update_required reads the volume (cached or
    nontemporal loads; per-voxel, quad and whole-rows kernels; with and without a volume, either layout), a fresh load, a boxed
    edit and the no-op passes that follow leave the same texels."""
    K = pkg._capi
    g = pkg.make_grid(dims, (-1.0, -0.75, -1.0), (1.0, 1.0, 0.5))
    prm, edited = pkg.default_params(), pkg.default_params(sphere_radius=0.8, cube_material=1)
    box = (-0.6, -0.5, -0.7, 0.3, 0.9, 0.2)
    layout = K.PASS_VOLUME_INTERLEAVED if ilv else 0

    def run(loads, hint, use_dist):
        t0, t1 = pkg.alloc_textures(g)
        pkg.grid_init(g, t0, t1)
        dist = None
        if use_dist:
            dist = torch.full((dims[2], dims[1], dims[0]), pkg.AIR_DIST, dtype=torch.float32, device="cuda")
        fl = (layout if use_dist else 0) | hint
        with pkg.options({K.OPT_PASS_LOADS: loads}):
            for step in (4, 2, 1):
                pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist, flags=fl)
            for step in (4, 2, 1):
                pkg.fill_grid_pass(edited, g, step, t0, t1, changed_box=box, dist=dist, flags=fl)
            for step in (4, 2, 1):  # nothing left to do: the passes the hint is meant for
                pkg.fill_grid_pass(edited, g, step, t0, t1, dist=dist, flags=fl)
        torch.cuda.synchronize()
        return t0, t1, dist

    want = run(1, 0, False)
    r0, r1 = oracle.grid_init(dims)
    bb = ((-1.0, -0.75, -1.0), (1.0, 1.0, 0.5))
    oracle.viewer_update(oracle.params_from(prm), dims, oracle.lm_new(dims, 3), r0, r1, bb_min=bb[0], bb_max=bb[1])
    oracle.viewer_update(oracle.params_from(edited), dims, oracle.lm_new(dims, 3), r0, r1, changed_box=box, bb_min=bb[0], bb_max=bb[1])
    assert_bits_equal(want[0], r0)
    assert_bits_equal(want[1], r1)
    for loads in (0, 1, 2):
        for hint in (0, K.PASS_EXPECT_NOOP):
            for use_dist in (False, True):
                t0, t1, dist = run(loads, hint, use_dist)
                assert torch.equal(t0, want[0]) and torch.equal(t1, want[1]), (loads, hint, use_dist)
                if dist is not None:
                    d = dist.reshape(-1, dims[0], 2).transpose(1, 2).reshape(dims[2], dims[1], dims[0]) if ilv else dist
                    assert torch.equal(d, t0[..., 0])
    with pytest.raises(pkg.SdfvError):
        pkg.set_option(K.OPT_PASS_LOADS, 3)
