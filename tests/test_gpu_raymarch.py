"""GPU parity of the sphere tracer vs the oracle's restatement of material.frag.

Everything fully determined by in-tree reference source (hit flag, step count, hit position, raw texture
samples, normal, depth) must be BIT-EXACT.  The shaded RGBA goes through pow() (ACES -> sRGB), where the
device's libm differs from the host's in the last ulps: tolerance 1e-4 (BASELINE.json north_star), stated
against the oracle's full-fp32 restatement."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RGBA_TOL = 1e-4


def setup_grid(pkg, oracle, dims, bb_min=(-1, -1, -1), bb_max=(1, 1, 1), **kw):
    prm = pkg.default_params(**kw)
    g = pkg.make_grid(dims, bb_min, bb_max)
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(prm, g, t0, t1)
    torch.cuda.synchronize()
    return g, t0, t1, t0.cpu().numpy(), t1.cpu().numpy()


def aux_to_np(oracle, aux):
    return aux.cpu().numpy().view(oracle.AUX_DTYPE).reshape(aux.shape[:-1])


def compare(pkg, oracle, g, t0, t1, h0, h1, cam_kw, width, height, rp_edit=None, y0=0, y1=None):
    rp = pkg.default_render_params(g)
    if rp_edit:
        rp_edit(rp)
    cam = pkg.camera_look_at(aspect=width / height, **cam_kw)
    orp = oracle.copy_struct(oracle.RenderParams, rp)
    ocam = oracle.copy_struct(oracle.Camera, cam)
    want_rgba, want_aux = oracle.raymarch(orp, h0, h1, ocam, width, height, y0=y0, y1=y1)
    dist = pkg.commit_distance(g, t0)
    assert torch.equal(dist, t0[..., 0])
    whole = pkg.make_grid(tuple(int(d) for d in g.dims), tuple(g.bb_min), tuple(g.bb_max))
    pairs = pkg.commit_pairs(whole, dist)  # (d[y], d[min(y + 1, H - 1)]) per texel
    assert torch.equal(pairs[..., 0], dist) and torch.equal(pairs[:, :-1, :, 1], dist[:, 1:]) and torch.equal(pairs[:, -1, :, 1], dist[:, -1])
    ilv = None
    if int(g.dims[1]) % 2 == 0:  # the y-interleaved volume pairs rows 2p, 2p + 1
        ilv = pkg.commit_interleaved(whole, dist)
        v = ilv.view(dist.shape[0], dist.shape[1] // 2, dist.shape[2], 2)
        assert torch.equal(v[..., 0], dist[:, 0::2]) and torch.equal(v[..., 1], dist[:, 1::2])
    # every march kernel family must reproduce the oracle: the fast march (symmetric-box / fused-scale /
    # reciprocal / divide variants) over tex0.r, the same over the compact distance volume, and the general
    # kernel (full MirroredRepeat, the shader's nested loop)
    K = pkg._capi
    # "fast" / "dist" take the hand-written gfx950 march loop where its specialisation applies (power-of-two grid,
    # symmetric box); "*_c" force the compiler's loop on the same kernels
    # "*_b": the hand-written loop with its interior fetch path switched off (every cell through the clamping fetch)
    disabled = {"fast": 0, "dist": 0, "fast_c": K.RM_NO_ASM_LOOP, "dist_c": K.RM_NO_ASM_LOOP, "general": K.RM_NO_FAST_INDEX,
                "fast_b": K.RM_NO_INTERIOR_FETCH, "dist_b": K.RM_NO_INTERIOR_FETCH,
                # "pairs*": the y-pair volume (two 16-byte gathers per cell), its border fetch for every cell, and a launch
                # whose specialisation does not apply (falls back to the distance volume)
                "pairs": 0, "pairs_b": K.RM_NO_INTERIOR_FETCH, "pairs_c": K.RM_NO_ASM_LOOP,
                # "ilv*": the y-interleaved volume, likewise
                "ilv": 0, "ilv_b": K.RM_NO_INTERIOR_FETCH, "ilv_c": K.RM_NO_ASM_LOOP,
                # without the distance volume beside them (given it, a non-cubic grid marches over that instead)
                "pairs_only": 0, "ilv_only": 0,
                "fast_plain": K.RM_NO_SYMMETRIC | K.RM_NO_POW2_SIZE,
                "fast_div": K.RM_NO_SYMMETRIC | K.RM_NO_POW2_EXTENT}
    for variant, mask in disabled.items():
        if variant.startswith("ilv") and ilv is None:
            continue
        with pkg.options({K.OPT_RAYMARCH_DISABLE: mask}):
            use_dist = dist if variant.startswith(("dist", "pairs", "ilv")) and not variant.endswith("_only") else None
            use_pairs = pairs if variant.startswith("pairs") else None
            use_ilv = ilv if variant.startswith("ilv") else None
            rgba, depth, aux = pkg.raymarch(rp, t0, t1, cam, width, height, y0=y0, y1=y1, want_aux=True,
                                            want_depth=True, dist=use_dist, pairs=use_pairs, ilv=use_ilv)
            # the depth plane without the 72-byte record must be the same plane
            rgba_plain, depth_only = pkg.raymarch(rp, t0, t1, cam, width, height, y0=y0, y1=y1, want_depth=True,
                                                  dist=use_dist, pairs=use_pairs, ilv=use_ilv)
            torch.cuda.synchronize()
        assert torch.equal(depth.view(torch.int32), depth_only.view(torch.int32)), variant
        # the kernel WITHOUT the aux record (the one bench.py times) writes the same RGBA as the one with it
        assert torch.equal(rgba.view(torch.int32), rgba_plain.view(torch.int32)), f"{variant}: aux / no-aux RGBA differ"
        assert torch.equal(depth.view(torch.int32), aux[..., -1]), f"{variant}: depth plane != aux.depth"
        got_rgba = rgba[0].cpu().numpy()
        got_aux = aux_to_np(oracle, aux)[0]
        for field in ("status", "steps"):
            np.testing.assert_array_equal(got_aux[field], want_aux[field], err_msg=f"{variant}:{field}")
        for field in ("hit_pos", "t", "raw0", "raw1", "normal", "depth"):
            np.testing.assert_array_equal(got_aux[field].view(np.uint32), want_aux[field].view(np.uint32),
                                          err_msg=f"{variant}:{field}")
        assert np.abs(got_rgba - want_rgba).max() <= RGBA_TOL, variant
        np.testing.assert_array_equal(got_rgba[..., 3], want_rgba[..., 3])
    return got_rgba, got_aux


def test_default_camera_64(pkg, oracle):
    """configs[0] as BASELINE.json words it: 64^3 grid, 512 x 512 sphere-trace, the reference's default camera
    (scene/mod.rs:82-95) -- every pixel, every march variant, against the oracle (this ran at 160 x 120)."""
    env = setup_grid(pkg, oracle, (64, 64, 64))
    rgba, aux = compare(pkg, oracle, *env, cam_kw={}, width=512, height=512)
    assert (aux["status"] == 1).sum() > 20000 and (aux["status"] == 0).sum() > 20000
    rgba, aux = compare(pkg, oracle, *env, cam_kw={}, width=160, height=120)  # (an image that is not a multiple of the tile)
    assert (aux["status"] == 1).sum() > 1000 and (aux["status"] == 0).sum() > 1000


def test_non_square_odd_image_and_row_range(pkg, oracle):
    env = setup_grid(pkg, oracle, (40, 33, 27), (-1, -0.5, -1), (1, 1, 0.75))
    compare(pkg, oracle, *env, cam_kw=dict(eye=(1.5, 2.0, -3.0)), width=77, height=51)
    compare(pkg, oracle, *env, cam_kw=dict(eye=(1.5, 2.0, -3.0)), width=77, height=51, y0=13, y1=30)


def test_camera_inside_the_volume(pkg, oracle):
    """Back-face fragments: the march starts at cameraPosition + 0.2 * dir (material.frag:136-139)."""
    env = setup_grid(pkg, oracle, (48, 48, 48))
    rgba, aux = compare(pkg, oracle, *env, cam_kw=dict(eye=(0.1, 0.2, 0.3), target=(1.0, 0.9, 0.8)), width=64, height=64)
    assert (aux["status"] != 0).all()


def test_axis_aligned_view_and_grazing_rays(pkg, oracle):
    env = setup_grid(pkg, oracle, (32, 32, 32))
    compare(pkg, oracle, *env, cam_kw=dict(eye=(0.0, 0.0, 4.0)), width=64, height=64)
    compare(pkg, oracle, *env, cam_kw=dict(eye=(4.0, 0.999, 0.0), target=(0.0, 0.999, 0.0)), width=96, height=32)


def test_other_sdf_params_and_shading_options(pkg, oracle):
    env = setup_grid(pkg, oracle, (36, 36, 36), cube_material=1, sphere_material=0, sphere_radius=0.9)

    def edit(rp):
        rp.gamma = 2.2
        rp.tint[0], rp.tint[1], rp.tint[2], rp.tint[3] = 0.9, 0.5, 0.25, 0.75
        rp.tone_mapping = 1

    compare(pkg, oracle, *env, cam_kw={}, width=80, height=60, rp_edit=edit)

    def edit2(rp):
        rp.tone_mapping = 3
        rp.color_mapping = 0

    compare(pkg, oracle, *env, cam_kw={}, width=80, height=60, rp_edit=edit2)

    def two_more_ambient_lights(rp):  # the light list beyond the scene's one AmbientLight (scene/mod.rs:106-112)
        rp.ambient[0], rp.ambient[1], rp.ambient[2] = 0.25, 0.25, 0.25
        rp.n_lights = 2
        rp.lights[0].kind, rp.lights[0].intensity = pkg._capi.LIGHT_AMBIENT, 0.5
        rp.lights[0].color[0], rp.lights[0].color[1], rp.lights[0].color[2] = 1.0, 0.5, 0.25
        rp.lights[1].kind, rp.lights[1].intensity = pkg._capi.LIGHT_AMBIENT, 0.125
        rp.lights[1].color[0], rp.lights[1].color[1], rp.lights[1].color[2] = 0.0, 1.0, 1.0

    rgba, _ = compare(pkg, oracle, *env, cam_kw={}, width=80, height=60, rp_edit=two_more_ambient_lights)
    base, _ = compare(pkg, oracle, *env, cam_kw={}, width=80, height=60)
    assert np.abs(rgba[..., :3] - base[..., :3]).max() > 0.05  # the lights really change the picture


def test_tiny_bbox_disables_fast_index(pkg, oracle):
    """A box so small that 1e-4 (the shader's absolute OOB epsilon) spans several texels: rays march outside
    [0,1] in texture space and MirroredRepeat really mirrors; the launcher must fall back to the general kernel."""
    bb = ((-1e-3, -1e-3, -1e-3), (1e-3, 1e-3, 1e-3))
    prm = pkg.default_params(cube_half_side=0.95e-3, sphere_radius=1.05e-3, max_distance_custom_material=0.05e-3)
    g = pkg.make_grid((24, 24, 24), *bb)
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(prm, g, t0, t1)
    torch.cuda.synchronize()
    compare(pkg, oracle, g, t0, t1, t0.cpu().numpy(), t1.cpu().numpy(),
            cam_kw=dict(eye=(2.5e-3, 3e-3, 5e-3), z_near=1e-5), width=64, height=64)


def test_loading_lod_nearest_path(pkg, oracle):
    """While loading (sdfLODDistBetweenSamples > 1) the shader snaps to the coarse lattice with NEAREST
    (material.frag:27-36,46-51) over a partially filled (AIR_DIST) grid."""
    prm = pkg.default_params()
    dims = (32, 32, 32)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    pkg.fill_grid_pass(prm, g, 4, t0, t1)
    torch.cuda.synchronize()
    h0, h1 = t0.cpu().numpy(), t1.cpu().numpy()

    def edit(rp):
        rp.lod_dist_between_samples = 4.0

    compare(pkg, oracle, g, t0, t1, h0, h1, cam_kw={}, width=96, height=96, rp_edit=edit)


def test_multi_camera_batch_equals_single(pkg, oracle):
    """configs[4] shape: a batch of cameras (more than one launch's worth) equals per-camera calls."""
    g, t0, t1, h0, h1 = setup_grid(pkg, oracle, (32, 32, 32))
    rp = pkg.default_render_params(g)
    cams = pkg.orbit_cameras(19, aspect=4 / 3)
    batch = pkg.raymarch(rp, t0, t1, cams, 64, 48)
    torch.cuda.synchronize()
    for k in (0, 7, 16, 18):
        single = pkg.raymarch(rp, t0, t1, cams[k], 64, 48)
        assert torch.equal(batch[k], single[0])
    orp = oracle.copy_struct(oracle.RenderParams, rp)
    want, _ = oracle.raymarch(orp, h0, h1, oracle.copy_struct(oracle.Camera, cams[18]), 64, 48, want_aux=False)
    assert np.abs(batch[18].cpu().numpy() - want).max() <= RGBA_TOL


def test_1080p_properties_on_256_grid(pkg, oracle):
    """configs[1] at full size: rows split across launches are seamless, the image is deterministic,
    and a band of rows matches the oracle."""
    g, t0, t1, h0, h1 = setup_grid(pkg, oracle, (256, 256, 256))
    rp = pkg.default_render_params(g)
    W, H = 1920, 1080
    cam = pkg.camera_look_at(aspect=W / H)
    full = pkg.raymarch(rp, t0, t1, cam, W, H)
    again = pkg.raymarch(rp, t0, t1, cam, W, H)
    parts = [pkg.raymarch(rp, t0, t1, cam, W, H, y0=a, y1=b) for a, b in ((0, 135), (135, 700), (700, 1080))]
    torch.cuda.synchronize()
    assert torch.equal(full, again)
    assert torch.equal(torch.cat(parts, dim=1), full)
    assert bool(torch.isfinite(full).all()) and float(full.min()) >= 0.0 and float(full.max()) <= 1.0
    alpha = full[0, ..., 3]
    assert bool(((alpha == 0) | (alpha == 1)).all()) and 0.05 < float(alpha.mean()) < 0.8
    orp = oracle.copy_struct(oracle.RenderParams, rp)
    ocam = oracle.copy_struct(oracle.Camera, cam)
    want, _ = oracle.raymarch(orp, h0, h1, ocam, W, H, y0=536, y1=544, want_aux=False)
    assert np.abs(full[0, 536:544].cpu().numpy() - want).max() <= RGBA_TOL


def test_golden_raymarch_fixture(pkg, oracle):
    """The committed numpy-restatement fixture (tests/golden/raymarch_12cube_40x30.npz) straight against the GPU."""
    import ctypes as C
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "raymarch_12cube_40x30.npz"))
    W, H = int(g["width"]), int(g["height"])
    grid = pkg.make_grid((12, 12, 12))
    t0, t1 = pkg.alloc_textures(grid)
    pkg.fill_grid(pkg.default_params(), grid, t0, t1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(t0.cpu().numpy().view(np.uint32), g["tex0"].view(np.uint32))
    np.testing.assert_array_equal(t1.cpu().numpy().view(np.uint32), g["tex1"].view(np.uint32))
    rp = pkg.default_render_params(grid)
    for k in (0, 1):
        cam = pkg.Camera()
        C.memmove(C.byref(cam), g[f"cam_{k}"].ctypes.data, C.sizeof(cam))
        rgba, aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True)
        torch.cuda.synchronize()
        a = aux_to_np(oracle, aux)[0]
        np.testing.assert_array_equal(a["status"], g[f"status_{k}"])
        np.testing.assert_array_equal(a["steps"], g[f"steps_{k}"])
        covered = a["status"] != 0
        np.testing.assert_array_equal(a["hit_pos"][covered].view(np.uint32), g[f"hit_pos_{k}"][covered].view(np.uint32))
        assert np.abs(rgba[0].cpu().numpy() - g[f"rgba_{k}"]).max() <= RGBA_TOL


def test_randomised_cameras_grids_and_boxes(pkg, oracle):
    """Seeded sweep: cameras outside / inside / on the box's surface, fields of view from 10 to 120 degrees, grids
    with power-of-two and odd sizes, symmetric / shifted / non-power-of-two boxes -- every kernel specialisation is
    reached through the launcher's own selection, and each must match the oracle bit for bit before shading."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SDFV_SOAK_SEED", 77)))  # tools/soak.sh varies the seed
    boxes = [((-1, -1, -1), (1, 1, 1)), ((0, 0, 0), (2, 2, 2)), ((-0.75, -1, -0.5), (0.75, 1, 0.5)),
             ((-1, -1, -1), (1.5, 0.25, 3.0)), ((-2, -2, -2), (2, 2, 2))]
    for trial in range(int(os.environ.get("SDFV_SOAK_TRIALS", 14))):
        bb_min, bb_max = boxes[trial % len(boxes)]
        dims = tuple(int(d) for d in (rng.choice([16, 32, 64], size=3) if trial % 2 == 0 else rng.integers(5, 50, size=3)))
        scale = float(np.max(np.abs(np.array(bb_max))))
        prm = pkg.default_params(cube_half_side=0.95 * scale * 0.5, sphere_radius=1.05 * scale * 0.5,
                                 max_distance_custom_material=0.05 * scale,
                                 cube_material=int(rng.integers(0, 2)), sphere_material=int(rng.integers(0, 2)))
        g = pkg.make_grid(dims, bb_min, bb_max)
        t0, t1 = pkg.alloc_textures(g)
        pkg.fill_grid(prm, g, t0, t1)
        torch.cuda.synchronize()
        centre = (np.array(bb_min) + np.array(bb_max)) / 2
        half = (np.array(bb_max) - np.array(bb_min)) / 2
        kind = trial % 3
        if kind == 0:      # outside, looking roughly at the box
            eye = centre + rng.normal(size=3) / np.linalg.norm(rng.normal(size=3) + 1e-3) * half.max() * rng.uniform(2.0, 5.0)
            eye = centre + (eye - centre) / np.linalg.norm(eye - centre) * half.max() * rng.uniform(2.0, 5.0)
            target = centre + rng.uniform(-0.3, 0.3, size=3) * half
        elif kind == 1:    # inside the volume
            eye = centre + rng.uniform(-0.8, 0.8, size=3) * half
            target = centre + rng.uniform(-1.0, 1.0, size=3) * half * 1.5
        else:              # exactly on a face of the box
            eye = centre + rng.uniform(-0.9, 0.9, size=3) * half
            eye[trial % 3] = bb_max[trial % 3]
            target = centre
        w, h = int(rng.integers(17, 90)), int(rng.integers(17, 70))
        compare(pkg, oracle, g, t0, t1, t0.cpu().numpy(), t1.cpu().numpy(),
                cam_kw=dict(eye=tuple(float(x) for x in eye), target=tuple(float(x) for x in target),
                            fovy_degrees=float(rng.uniform(10.0, 120.0))), width=w, height=h)


def test_randomised_sweep_of_the_hand_written_march_loop(pkg, oracle):
    """Seeded sweep aimed at the gfx950 assembly loop's specialisation: grids of ANY size (powers of two or not, cubic or not,
    down to 2 texels per axis; a 1-texel axis has NaN coordinates, 0/0 in scene/sdf/mod.rs:179 -- the loop addresses rows by
    24-bit multiplies), symmetric boxes with power-of-two extents (cubic or not), cameras outside /
    inside / on a face.  compare() runs the hand-written loop and the compiler's on the same inputs, over tex0.r and over
    the distance volume, with the aux record (distance and step counters ride along in the loop) -- all bit for bit."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SDFV_SOAK_SEED", 99)) + 5000)
    sizes = [2, 4, 8, 16, 32, 64, 128]
    odd_sizes = [3, 5, 6, 7, 10, 12, 20, 24, 33, 36, 50, 63, 65, 100]
    for trial in range(int(os.environ.get("SDFV_SOAK_TRIALS", 16))):
        pool = sizes if trial % 4 < 2 else sizes + odd_sizes * 2  # half of the trials: sizes that are not powers of two
        dims = tuple(int(rng.choice(pool)) for _ in range(3))
        if dims[0] * dims[1] * dims[2] > 2 ** 19:
            dims = (dims[0], dims[1], max(2, 2 ** 19 // (dims[0] * dims[1])))
        if trial % 2 == 0:  # cubic volumes: the loop's interior fetch path (one compare for all three cell indices)
            dims = (int(rng.choice(sizes[:6] if trial % 4 < 2 else odd_sizes[:-3] + [63, 65])),) * 3
        half = np.array([2.0 ** int(rng.integers(-2, 3)) for _ in range(3)]) if trial % 2 else np.full(3, 2.0 ** int(rng.integers(-2, 3)))
        bb_min, bb_max = tuple(-half), tuple(half)
        scale = float(half.min())
        prm = pkg.default_params(cube_half_side=0.95 * scale, sphere_radius=1.05 * scale,
                                 max_distance_custom_material=0.05 * scale,
                                 cube_material=int(rng.integers(0, 2)), sphere_material=int(rng.integers(0, 2)),
                                 disable_sphere=int(trial % 7 == 3))
        g = pkg.make_grid(dims, bb_min, bb_max)
        t0, t1 = pkg.alloc_textures(g)
        pkg.fill_grid(prm, g, t0, t1)
        torch.cuda.synchronize()
        kind = trial % 3
        if kind == 0:
            d = rng.normal(size=3)
            eye = d / np.linalg.norm(d) * half.max() * rng.uniform(2.0, 6.0)
            target = rng.uniform(-0.3, 0.3, size=3) * half
        elif kind == 1:
            eye = rng.uniform(-0.8, 0.8, size=3) * half
            target = rng.uniform(-1.5, 1.5, size=3) * half
        else:
            eye = rng.uniform(-0.9, 0.9, size=3) * half
            eye[trial % 3] = half[trial % 3] * (1 if trial % 2 else -1)
            target = np.zeros(3)
        w, h = int(rng.integers(17, 100)), int(rng.integers(17, 80))
        compare(pkg, oracle, g, t0, t1, t0.cpu().numpy(), t1.cpu().numpy(),
                cam_kw=dict(eye=tuple(float(x) for x in eye), target=tuple(float(x) for x in target),
                            fovy_degrees=float(rng.uniform(10.0, 120.0)), z_near=float(0.05 * scale)), width=w, height=h)


@pytest.mark.parametrize("size", [(160, 120), (333, 217), (17, 500), (1000, 30)])
def test_every_tile_order_renders_the_same_image(pkg, size):
    """SDFV_OPT_RAYMARCH_TILE_GROUP only changes WHICH workgroup renders which 16 x 16 tile (XCD-aware groups of tiles, with
    padding groups at the image's edges that must write nothing): every order leaves the same bits, aux and depth too."""
    W, H = size
    g = pkg.make_grid((32, 32, 32))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at(aspect=W / H)
    K = pkg._capi
    ref = None
    for order in (1, 0, 2, 3, 4, 5):  # launch order first; 0 = auto (grouped for one camera)
        with pkg.options({K.OPT_RAYMARCH_TILE_GROUP: order}):
            rgba, depth, aux = pkg.raymarch(rp, t0, t1, cam, W, H, want_aux=True, want_depth=True)
            band = pkg.raymarch(rp, t0, t1, cam, W, H, y0=H // 3, y1=2 * H // 3)
        torch.cuda.synchronize()
        got = (rgba.view(torch.int32).clone(), depth.view(torch.int32).clone(), aux.clone(), band.view(torch.int32).clone())
        if ref is None:
            ref = got
            assert bool((rgba[..., 3] > 0).any())
        for a, b in zip(got, ref):
            assert torch.equal(a, b), order


@pytest.mark.parametrize("size", [(160, 120), (333, 217), (640, 480), (70, 900), (1920, 1080)])
def test_box_first_order_covers_every_tile_once(pkg, size):
    """SDFV_OPT_RAYMARCH_BOX_FIRST reorders the groups of tiles (projected bounding box first): for cameras that see the
    box anywhere -- centred, cut by each image edge, tiny, filling the image, behind the camera, from inside -- the image
    written over a sentinel equals the launch-order image bit for bit (every tile rendered, none twice with other
    results, padding groups silent)."""
    W, H = size
    g = pkg.make_grid((32, 32, 32))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    rp = pkg.default_render_params(g)
    K = pkg._capi
    views = [dict(), dict(eye=(2.5, 3.0, 5.0), target=(3.0, 0.0, 0.0)), dict(eye=(2.5, 3.0, 5.0), target=(-3.0, 0.0, 0.0)),
             dict(eye=(2.5, 3.0, 5.0), target=(0.0, 3.0, 0.0)), dict(eye=(2.5, 3.0, 5.0), target=(0.0, -3.5, 0.0)),
             dict(eye=(25.0, 30.0, 50.0)), dict(eye=(0.9, 1.2, 1.7)), dict(eye=(0.2, 0.1, 0.3), target=(1.0, 0.0, 0.0)),
             dict(eye=(2.5, 3.0, 5.0), target=(5.0, 6.0, 10.0)), dict(eye=(0.0, 0.0, 4.0), fovy_degrees=100.0),
             dict(eye=(4.0, 0.0, 0.1), target=(0.0, 0.0, 3.0), fovy_degrees=20.0)]
    for kw in views:
        cam = pkg.camera_look_at(aspect=W / H, **kw)
        with pkg.options({K.OPT_RAYMARCH_TILE_GROUP: 1}):
            ref = pkg.raymarch(rp, t0, t1, cam, W, H)
        for group in (0, 2, 3, 4):
            for rows in ((0, H), (H // 3, 2 * H // 3 + 1)):
                out = torch.full((1, rows[1] - rows[0], W, 4), float("nan"), dtype=torch.float32, device="cuda")
                with pkg.options({K.OPT_RAYMARCH_TILE_GROUP: group, K.OPT_RAYMARCH_BOX_FIRST: 1}):
                    pkg.raymarch(rp, t0, t1, cam, W, H, y0=rows[0], y1=rows[1], out=out)
                assert torch.equal(out.view(torch.int32), ref[:, rows[0]:rows[1]].view(torch.int32)), (kw, group, rows)
        out = torch.full((1, H, W, 4), float("nan"), dtype=torch.float32, device="cuda")
        with pkg.options({K.OPT_RAYMARCH_BOX_FIRST: 0}):
            pkg.raymarch(rp, t0, t1, cam, W, H, out=out)
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), kw


def test_randomised_tile_orders(pkg):
    """Seeded sweep over image sizes, row bands, cameras (outside, inside, looking past the box) and tile-order options:
    every order writes, over a sentinel, the bits the launch order writes."""
    import os
    rng = np.random.default_rng(int(os.environ.get("SDFV_SOAK_SEED", 99)) + 9000)
    K = pkg._capi
    g = pkg.make_grid((32, 32, 32))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    rp = pkg.default_render_params(g)
    for trial in range(int(os.environ.get("SDFV_SOAK_TRIALS", 12))):
        W, H = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        kind = trial % 4
        if kind == 0:
            d = rng.normal(size=3)
            eye, target = d / np.linalg.norm(d) * rng.uniform(1.8, 30.0), rng.uniform(-0.5, 0.5, size=3)
        elif kind == 1:
            eye, target = rng.uniform(-0.9, 0.9, size=3), rng.uniform(-2.0, 2.0, size=3)
        elif kind == 2:
            d = rng.normal(size=3)
            eye = d / np.linalg.norm(d) * rng.uniform(2.0, 8.0)
            target = eye + rng.normal(size=3)  # anywhere: the box may be off screen or behind the camera
        else:
            d = rng.normal(size=3)
            eye, target = d / np.linalg.norm(d) * rng.uniform(2.0, 6.0), rng.uniform(-3.0, 3.0, size=3)
        cam = pkg.camera_look_at(eye=tuple(float(x) for x in eye), target=tuple(float(x) for x in target),
                                 fovy_degrees=float(rng.uniform(5.0, 140.0)), aspect=W / H)
        y0 = int(rng.integers(0, H))
        y1 = int(rng.integers(y0 + 1, H + 1)) if trial % 3 else H
        y0 = y0 if trial % 3 else 0
        use_dist = dist if trial % 2 else None
        with pkg.options({K.OPT_RAYMARCH_TILE_GROUP: 1}):
            ref = pkg.raymarch(rp, t0, t1, cam, W, H, y0=y0, y1=y1, dist=use_dist)
        for group in (0, int(rng.integers(2, 6))):
            for first in (1, 0):
                out = torch.full((1, y1 - y0, W, 4), float("nan"), dtype=torch.float32, device="cuda")
                with pkg.options({K.OPT_RAYMARCH_TILE_GROUP: group, K.OPT_RAYMARCH_BOX_FIRST: first}):
                    pkg.raymarch(rp, t0, t1, cam, W, H, y0=y0, y1=y1, out=out, dist=use_dist)
                assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (trial, W, H, y0, y1, group, first)
        # launches that keep launch order -- batches of cameras, tile bands -- rotate the tile columns by row and camera
        # (the XCDs' static shares balance); SDFV_OPT_RAYMARCH_TILE_GROUP 1 is the plain order
        cams = [cam] + pkg.orbit_cameras(int(rng.integers(1, 5)), aspect=W / H)
        bands = (int(rng.integers(0, 3)), int(rng.integers(1, 4)), int(rng.choice([8, 16]))) if trial % 2 else None
        kw = {"bands": bands} if bands else {"y0": y0, "y1": y1}
        with pkg.options({K.OPT_RAYMARCH_TILE_GROUP: 1}):
            ref = pkg.raymarch(rp, t0, t1, cams, W, H, dist=use_dist, **kw)
        if ref.numel():
            out = torch.full(tuple(ref.shape), float("nan"), dtype=torch.float32, device="cuda")
            pkg.raymarch(rp, t0, t1, cams, W, H, out=out, dist=use_dist, **kw)
            assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (trial, W, H, kw, len(cams), "rotated columns")


def test_occupancy_cap_changes_nothing_but_speed(pkg):
    """SDFV_OPT_RAYMARCH_WAVES_PER_SIMD asks for unused dynamic LDS per workgroup: every value renders the same bits, with
    and without the aux record, single frames and batches, over tex0.r and over the distance volume."""
    K = pkg._capi
    g = pkg.make_grid((64, 64, 64))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    rp = pkg.default_render_params(g)
    W, H = 400, 300
    cams = pkg.orbit_cameras(3, aspect=W / H)
    ref = pkg.raymarch(rp, t0, t1, cams, W, H, want_aux=True, want_depth=True, dist=dist)
    ref0 = pkg.raymarch(rp, t0, t1, cams[0], W, H)
    for waves in (0, 2, 3, 4, 5, 6, 7):  # 0 = the launcher's own rule
        with pkg.options({K.OPT_RAYMARCH_WAVES_PER_SIMD: waves}):
            got = pkg.raymarch(rp, t0, t1, cams, W, H, want_aux=True, want_depth=True, dist=dist)
            got0 = pkg.raymarch(rp, t0, t1, cams[0], W, H)
        torch.cuda.synchronize()
        assert torch.equal(got[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(got[2], ref[2]), waves
        assert torch.equal(got[1].view(torch.int32), ref[1].view(torch.int32)), waves
        assert torch.equal(got0.view(torch.int32), ref0.view(torch.int32)), waves


def test_launcher_occupancy_rule_at_the_sizes_it_fires(pkg):
    """The launcher's rule caps the resident waves (4 per SIMD) for a single frame over a volume larger than the Infinity
    Cache whose projected box holds 1x .. 3.5x the machine's wave slots: 1440p over 512^3 with the default camera is such a
    launch, its close-up and its 64^3 sibling are not.  Whatever it decides, the image is the uncapped one bit for bit."""
    K = pkg._capi
    prm = pkg.default_params()
    g = pkg.make_grid((512, 512, 512))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((512, 512, 512), dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    rp = pkg.default_render_params(g)
    for (W, H), eye in (((2560, 1440), (2.5, 3.0, 5.0)), ((3840, 2160), (2.5, 3.0, 5.0)), ((2560, 1440), (1.2, 1.5, 2.4))):
        cam = pkg.camera_look_at(eye=eye, aspect=W / H)
        for use_dist in (dist, None):
            auto = pkg.raymarch(rp, t0, t1, cam, W, H, dist=use_dist)
            with pkg.options({K.OPT_RAYMARCH_WAVES_PER_SIMD: 7}):
                free = pkg.raymarch(rp, t0, t1, cam, W, H, dist=use_dist)
            assert torch.equal(auto.view(torch.int32), free.view(torch.int32)), (W, H, eye)


def test_march_volume_advice_and_the_launchers_choice_at_512(pkg):
    """sdfv_march_volume_advice: the pair volume while its 8 B/voxel fit the last-level cache, the interleaved volume beyond
    (and whenever H is odd: pairs).  Handed both, the launcher applies the same rule; whichever volume a frame marches over
    -- distance, pairs, interleaved, both -- the image and the depth plane are the same bits, here at 4K over 512^3 where the
    rule picks the interleaved volume, with the camera outside, close and inside."""
    assert pkg.march_volume_advice(pkg.make_grid((256, 256, 256))) == "pairs"
    assert pkg.march_volume_advice(pkg.make_grid((512, 512, 512))) == "interleaved"
    assert pkg.march_volume_advice(pkg.make_grid((511, 511, 511))) == "pairs"       # odd H: no interleaved volume
    assert pkg.march_volume_advice(pkg.make_grid((512, 256, 512))) is None           # not cubic: the distance volume
    prm = pkg.default_params()
    g = pkg.make_grid((512, 512, 512))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((512, 512, 512), dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    pairs, ilv = pkg.commit_pairs(g, dist), pkg.commit_interleaved(g, dist)
    v = ilv.view(512, 256, 512, 2)
    assert torch.equal(v[..., 0], dist[:, 0::2]) and torch.equal(v[..., 1], dist[:, 1::2])
    rp = pkg.default_render_params(g)
    W, H = 3840, 2160
    for kw in (dict(), dict(eye=(1.2, 1.5, 2.4)), dict(eye=(0.2, 0.1, 0.3), target=(1.0, 0.5, -1.0))):
        cam = pkg.camera_look_at(aspect=W / H, **kw)
        ref, ref_depth = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, want_depth=True)
        for vols in (dict(pairs=pairs), dict(ilv=ilv), dict(pairs=pairs, ilv=ilv)):
            got, depth = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, want_depth=True, **vols)
            assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (kw, list(vols))
            assert torch.equal(depth.view(torch.int32), ref_depth.view(torch.int32)), (kw, list(vols))


def test_hand_written_loops_past_2_28_texels(pkg):
    """1024 x 1024 x 512 = 2^29 voxels: the distance and interleaved volumes' byte offsets pass 2^31 (the loops' 32-bit
    offsets are unsigned and reach 2^30 voxels; the pair volume's 8 B/texel stop at 2^28 and fall back to the distance
    volume).  The compiler's loop over the same volume is the reference, bit for bit, image and depth."""
    if torch.cuda.mem_get_info()[0] < 40 << 30:
        pytest.skip("needs 40 GB of free HBM")
    K = pkg._capi
    dims = (1024, 1024, 512)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((512, 1024, 1024), dtype=torch.float32, device="cuda")
    pkg.fill_grid(pkg.default_params(), g, t0, t1, dist=dist)
    ilv = pkg.commit_interleaved(g, dist)
    pairs = pkg.commit_pairs(g, dist)
    rp = pkg.default_render_params(g)
    W, H = 1920, 1080
    for kw in (dict(), dict(eye=(-1.4, -1.1, -2.2)), dict(eye=(0.3, -0.2, 0.9), target=(-1.0, 0.4, -1.0))):
        cam = pkg.camera_look_at(aspect=W / H, **kw)
        with pkg.options({K.OPT_RAYMARCH_DISABLE: K.RM_NO_ASM_LOOP}):
            ref, ref_depth = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, want_depth=True)
        for vols in (dict(), dict(ilv=ilv), dict(pairs=pairs)):
            got, depth = pkg.raymarch(rp, t0, t1, cam, W, H, dist=dist, want_depth=True, **vols)
            assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (kw, list(vols))
            assert torch.equal(depth.view(torch.int32), ref_depth.view(torch.int32)), (kw, list(vols))
    assert (ref_depth < 1.0).sum() > 1000  # the last camera looks at the far corner from inside: hits at high z


def test_row_bands_of_the_image_tile_split_tile_the_frame(pkg):
    """parallel.split_rows (config 5's image-tile split): the bands rendered by the "ranks" concatenate to the frame."""
    import importlib
    par = importlib.import_module("sdf-viewer_amd.parallel")
    dims, W, H = (32, 32, 32), 120, 100
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    rp = pkg.default_render_params(g)
    cams = pkg.orbit_cameras(3, aspect=W / H)
    whole = pkg.raymarch(rp, t0, t1, cams, W, H)
    for world in (2, 3, 7):
        bands = [pkg.raymarch(rp, t0, t1, cams, W, H, *par.split_rows(H, r, world)) for r in range(world)
                 if par.split_rows(H, r, world)[1] > par.split_rows(H, r, world)[0]]
        assert torch.equal(torch.cat(bands, dim=1).view(torch.int32), whole.view(torch.int32))


def test_interleaved_tile_bands_assemble_to_the_frame(pkg):
    """sdfv_raymarch_bands (config 5's balanced image-tile split): rank r of N renders the 16-row bands r, r + N, ... of every
    camera's image into a compact buffer; the N buffers assemble to the frame bit for bit -- RGBA, depth plane and aux
    record, over tex0.r / the distance volume / the pair volume, with the general (NEAREST, lod 2) kernel too, for image
    heights that are and are not multiples of 16 -- and nothing is written past sdfv_band_rows rows."""
    import importlib
    par = importlib.import_module("sdf-viewer_amd.parallel")
    dims = (32, 32, 32)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    pairs = pkg.commit_pairs(g, dist)
    for (W, H), lod in (((120, 100), 1.0), ((96, 64), 1.0), ((70, 33), 2.0), ((50, 7), 1.0)):
        rp = pkg.default_render_params(g)
        rp.lod_dist_between_samples = lod
        cams = pkg.orbit_cameras(3, aspect=W / H)
        for vols in (dict(), dict(dist=dist), dict(dist=dist, pairs=pairs)):
            whole, whole_depth, whole_aux = pkg.raymarch(rp, t0, t1, cams, W, H, want_depth=True, want_aux=True, **vols)
            for world in (1, 2, 3, 8):
                for bh in (16, 8):  # a workgroup's rows per band, or a wave's (what split_bands picks when bands are few)
                    parts = [pkg.raymarch(rp, t0, t1, cams, W, H, bands=par.split_bands(H, r, world, bh), want_depth=True, want_aux=True, **vols)
                             for r in range(world)]
                    for r, (rgba, depth, aux) in enumerate(parts):
                        assert rgba.shape[1] == len(par.band_rows(H, r, world, bh)) == int(pkg.lib.sdfv_band_rows_ex(H, r, world, bh))
                    for k, ref in enumerate((whole, whole_depth, whole_aux)):
                        got = par.assemble_bands([p[k] for p in parts], H, bh)
                        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (W, H, lod, list(vols), world, bh, k)
    # a compact buffer with a canary behind it; a band set beyond the image renders nothing
    rp = pkg.default_render_params(g)
    W, H = 64, 40
    cam = pkg.camera_look_at(aspect=W / H)
    rows = len(par.band_rows(H, 1, 2))  # band 1 only: 16 rows
    buf = torch.full((1, rows + 3, W, 4), -7.0, device="cuda")
    pkg.raymarch(rp, t0, t1, cam, W, H, bands=(1, 2), out=buf)
    torch.cuda.synchronize()
    assert rows == 16 and bool((buf[:, rows:] == -7.0).all()) and not bool((buf[:, :rows] == -7.0).any())
    assert pkg.raymarch(rp, t0, t1, cam, W, H, bands=(3, 4)).shape[1] == 0
    with pytest.raises(pkg.SdfvError):
        pkg.raymarch(rp, t0, t1, cam, W, H, bands=(0, 0))
    with pytest.raises(pkg.SdfvError):
        pkg.raymarch(rp, t0, t1, cam, W, H, bands=(0, 2, 12))  # bands are a workgroup's 16 rows or a wave's 8
    assert par.band_height_for(1080, 8) == 8 and par.band_height_for(1080, 4) == 16 and par.band_height_for(2160, 8) == 16


@pytest.mark.parametrize("n", [17, 64, 70])
def test_batches_beyond_the_cameras_a_launch_carries_in_its_arguments(pkg, n):
    """Up to 16 cameras ride in a launch's kernel-argument block (the block is back under 4 KB); a larger batch is read
    from device memory -- the caller's own device array in place, or the launcher's stream-ordered copy of a host array -- so
    that 64 cameras stay ONE launch; with the copy switched off, or under capture, it goes out as launches of 16.  The same
    pixels, depth and aux records every way, equal to one call per camera; the host array is free again when the call returns."""
    K = pkg._capi
    g = pkg.make_grid((32, 32, 32))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    rp = pkg.default_render_params(g)
    W, H = 72, 40
    cams = pkg.orbit_cameras(n, aspect=W / H)
    singles = [pkg.raymarch(rp, t0, t1, c, W, H, want_aux=True, want_depth=True, dist=dist) for c in cams]
    ref = tuple(torch.cat([s[k] for s in singles]) for k in range(3))
    torch.cuda.synchronize()

    def same(got, what):
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (n, what)

    host = [type(c).from_buffer_copy(bytes(c)) for c in cams]
    got = pkg.raymarch(rp, t0, t1, host, W, H, want_aux=True, want_depth=True, dist=dist)
    for c in host:  # (the launcher took its copy before returning)
        c.eye[0] = float("nan")
    same(got, "host array, staged")
    with pkg.options({K.OPT_RAYMARCH_CAMERA_STAGING: 0}):
        same(pkg.raymarch(rp, t0, t1, cams, W, H, want_aux=True, want_depth=True, dist=dist), "host array, launches of 16")
    dev = pkg.upload_cameras(cams)
    assert dev.numel() == n * 120
    same(pkg.raymarch(rp, t0, t1, dev, W, H, want_aux=True, want_depth=True, dist=dist), "device array")
    same(pkg.raymarch(rp, t0, t1, dev, W, H, want_aux=True, want_depth=True), "device array, tex0.r")
    # bands of a batch, and a stream of the caller's
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        parts = [pkg.raymarch(rp, t0, t1, cams, W, H, bands=(r, 2, 8), dist=dist, stream=side) for r in range(2)]
        side.synchronize()
    rows = [[y for y in range(H) if (y // 8) % 2 == r] for r in range(2)]
    for r in range(2):
        assert torch.equal(parts[r].view(torch.int32), ref[0][:, rows[r]].view(torch.int32)), (n, "bands", r)
    # captured: a host array goes out as launches of 16 (no allocation inside a capture), a device array in place
    for what, arg in (("host", cams), ("device", dev)):
        out = torch.zeros_like(ref[0])
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                pkg.raymarch(rp, t0, t1, arg, W, H, out=out, dist=dist, stream=side)
            out.zero_()
            graph.replay()
            side.synchronize()
        assert torch.equal(out.view(torch.int32), ref[0].view(torch.int32)), (n, "captured", what)
    # a DEVICE array of 16 or fewer cameras (a rank's share of a resident batch) is read in place too -- it used to
    # be dereferenced on the host
    for k in (1, 8, 16):
        part = pkg.raymarch(rp, t0, t1, pkg.upload_cameras(cams[:k]), W, H, dist=dist)
        torch.cuda.synchronize()
        assert torch.equal(part.view(torch.int32), ref[0][:k].view(torch.int32)), (n, "device array of", k)


def test_camera_ring_slots_are_reused_in_order_across_streams(pkg):
    """The ring that carries a host array of more than 16 cameras to the device has 16 slots: the 17th launch rewrites the
    first one's.  Sixty batches with sixty different camera sets, dealt to three streams that run ahead of one another with
    no synchronisation in between -- every image must be the one of ITS cameras (a slot rewritten before its reader has
    finished, or read before its writer, shows as a frame from another set)."""
    g = pkg.make_grid((64, 64, 64))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    dist = pkg.commit_distance(g, t0)
    rp = pkg.default_render_params(g)
    W, H, n = 96, 64, 20
    sets = [pkg.orbit_cameras(n, aspect=W / H, eye0=(2.5 + 0.03 * k, 3.0 - 0.02 * k, 5.0)) for k in range(60)]
    want = []
    for cams in sets:  # one call per set, the device array in place: no ring involved
        want.append(pkg.raymarch(rp, t0, t1, pkg.upload_cameras(cams), W, H, dist=dist))
    torch.cuda.synchronize()
    assert not torch.equal(want[0], want[1])
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [torch.zeros_like(want[0]) for _ in sets]
    for k, cams in enumerate(sets):
        s = streams[k % 3]
        with torch.cuda.stream(s):
            pkg.raymarch(rp, t0, t1, cams, W, H, out=outs[k], dist=dist, stream=s)
    torch.cuda.synchronize()
    for k in range(len(sets)):
        assert torch.equal(outs[k].view(torch.int32), want[k].view(torch.int32)), k
    assert pkg.lib.sdfv_mesh_trim() == 0  # releases the ring; the next batch allocates a new one
    again = pkg.raymarch(rp, t0, t1, sets[7], W, H, dist=dist)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), want[7].view(torch.int32))


def test_small_launches_of_a_batch_overlap_on_side_streams_and_change_nothing(pkg):
    """A batch of more than 64 cameras is several launches; small ones (low-resolution views, band shares) are forked onto the
    library's side streams and joined back into the caller's stream.  Same images as one launch after the other; ordered
    with the caller's stream on both sides (a fill before, a read after, no synchronisation in between); capturable."""
    K = pkg._capi
    dims = (64, 64, 64)
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    prm = pkg.default_params()
    rp = pkg.default_render_params(g)
    W, H = 200, 120
    cams = pkg.orbit_cameras(150, aspect=W / H)  # 3 launches: 64 + 64 + 22
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for radius in (1.05, 0.8):  # the second fill overwrites what the first batch read: the batch must wait for it, and it for the batch
            prm.sphere_radius = radius
            pkg.fill_grid(prm, g, t0, t1, stream=side)
            got, got_depth = pkg.raymarch(rp, t0, t1, cams, W, H, want_depth=True, stream=side)
            summed = got.sum(dim=(1, 2, 3))  # a consumer on the caller's stream
        side.synchronize()
        with pkg.options({K.OPT_RAYMARCH_BATCH_STREAMS: 0}):
            ref, ref_depth = pkg.raymarch(rp, t0, t1, cams, W, H, want_depth=True, stream=side)
        side.synchronize()
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)) and torch.equal(got_depth.view(torch.int32), ref_depth.view(torch.int32))
        assert torch.equal(summed, ref.sum(dim=(1, 2, 3)))
        # captured into a graph: the fork and the joins are events, the side streams join the capture
        out = torch.zeros_like(ref)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pkg.raymarch(rp, t0, t1, cams, W, H, out=out, stream=side)
        out.zero_()
        graph.replay()
        side.synchronize()
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    assert pkg.lib.sdfv_mesh_trim() == 0  # releases the side streams; the next batch makes new ones
    again = pkg.raymarch(rp, t0, t1, cams, W, H)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), ref.view(torch.int32))


def unorm8_of(rgba):
    """The 8-bit UNORM image of an fp32 one: rint(clamp(c, 0, 1) * 255), NaN -> 0 (sdfv_march_desc.rgba8)."""
    return torch.nan_to_num(rgba, nan=0.0).clamp(0.0, 1.0).mul(255.0).round().to(torch.uint8)


def test_rgba8_output_is_the_quantised_fp32_output(pkg):
    """sdfv_march_desc.rgba8 = outColor as the reference's 8-bit framebuffer holds it (frameinput.rs:19-24):
    equal to rint(clip(fp32) * 255) on EVERY pixel of the 64-camera 1080p batch (config 5), whether written beside the fp32 plane
    or instead of it; fp32 stays bit-identical with or without it; single frames, row ranges and tile bands alike."""
    W, H, n = 1920, 1080, 64
    prm = pkg.default_params()
    g = pkg.make_grid((256, 256, 256))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((256, 256, 256), dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
    rp = pkg.default_render_params(g)
    cams = pkg.orbit_cameras(n, aspect=W / H)
    ref = pkg.raymarch(rp, t0, t1, cams, W, H, dist=dist)
    both, img8 = pkg.raymarch(rp, t0, t1, cams, W, H, dist=dist, rgba8="both")
    only8 = pkg.raymarch(rp, t0, t1, cams, W, H, dist=dist, rgba8="only")
    torch.cuda.synchronize()
    assert only8.dtype == torch.uint8 and only8.shape == (n, H, W, 4)
    assert torch.equal(both.view(torch.int32), ref.view(torch.int32))
    want = unorm8_of(ref)
    assert torch.equal(img8, want) and torch.equal(only8, want)
    assert int((want[..., 3] == 255).sum()) > 0 and int((want[..., 3] == 0).sum()) > 0 and len(torch.unique(want)) > 200
    # one camera (the box-first tile order), a row range, a band set; the aux kernel's colour path as well
    one = pkg.raymarch(rp, t0, t1, cams[5], W, H, dist=dist, rgba8="only")
    assert torch.equal(one[0], want[5])
    rows = pkg.raymarch(rp, t0, t1, cams[5], W, H, y0=301, y1=777, rgba8="only")
    assert torch.equal(rows[0], want[5, 301:777])
    bands = pkg.raymarch(rp, t0, t1, cams[:2], W, H, dist=dist, bands=(1, 4, 8), rgba8="only")
    rows_of = np.concatenate([np.arange(b * 8, min(b * 8 + 8, H)) for b in range(1, (H + 7) // 8, 4)])
    assert torch.equal(bands, want[:2][:, torch.from_numpy(rows_of).cuda()])
    _, aux, aux8 = pkg.raymarch(rp, t0, t1, cams[5], 640, 360, want_aux=True, rgba8="both")
    plain = pkg.raymarch(rp, t0, t1, cams[5], 640, 360)
    assert torch.equal(aux8, unorm8_of(plain))
    # quantisation edges through the shading options: no tone / colour mapping leaves values beyond [0, 1] to clamp
    rp2 = pkg.default_render_params(g)
    rp2.tone_mapping, rp2.color_mapping = 0, 0
    rp2.ambient[0], rp2.ambient[1], rp2.ambient[2] = 40.0, 0.5, 0.01
    rp2.tint[3] = 1.75                                   # surfaceColorTint.a passes through unclamped: beyond 1 in fp32, 255 in UNORM
    f32, u8 = pkg.raymarch(rp2, t0, t1, cams[0], 800, 450, rgba8="both")
    assert float(f32.max()) == 1.75 and int(u8.max()) == 255 and torch.equal(u8, unorm8_of(f32))
    rp2.tint[3] = float("nan")
    f32, u8 = pkg.raymarch(rp2, t0, t1, cams[0], 800, 450, rgba8="both")
    assert bool(torch.isnan(f32).any()) and torch.equal(u8, unorm8_of(f32))
    with pytest.raises(pkg.SdfvError, match="no colour output"):
        d = pkg._capi.MarchDesc()
        d.size = C.sizeof(d)
        d.rp = C.pointer(rp)
        d.tex0, d.tex1 = t0.data_ptr(), t1.data_ptr()
        cam = (pkg.Camera * 1)(cams[0])
        d.cameras, d.n_cameras, d.width, d.height, d.y0, d.y1 = cam, 1, 64, 64, 0, 64
        pkg.check(pkg.lib.sdfv_raymarch_ex(C.byref(d), None))


def test_a_binder_built_against_the_120_byte_descriptor_still_renders(pkg):
    """The descriptor is size-prefixed: rgba8 (ABI 5) lies beyond what an ABI 4 binder hands over and reads as NULL."""
    g = pkg.make_grid((32, 32, 32))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, t0, t1)
    rp = pkg.default_render_params(g)
    cam = (pkg.Camera * 1)(pkg.camera_look_at(aspect=1.0))
    out = torch.zeros((48, 48, 4), device="cuda")
    d = pkg._capi.MarchDesc()
    d.size = 120
    d.rp = C.pointer(rp)
    d.tex0, d.tex1, d.rgba = t0.data_ptr(), t1.data_ptr(), out.data_ptr()
    d.cameras, d.n_cameras, d.width, d.height, d.y0, d.y1 = cam, 1, 48, 48, 0, 48
    d.rgba8 = 0xdead0000  # beyond `size`: must not be read
    pkg.check(pkg.lib.sdfv_raymarch_ex(C.byref(d), None))
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), pkg.raymarch(rp, t0, t1, cam[0], 48, 48)[0].view(torch.int32))
