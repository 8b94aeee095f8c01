"""Edge cases of the C ABI on the GPU: empty and degenerate inputs, maximum-size guards, argument validation."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_grid_and_empty_slab_are_no_ops(pkg):
    prm = pkg.default_params()
    for dims, z in [((0, 4, 4), (0, 4)), ((4, 0, 4), (0, 4)), ((4, 4, 0), (0, 0)), ((4, 4, 4), (2, 2))]:
        g = pkg.make_grid(dims, z_begin=z[0], z_end=z[1])
        t0 = torch.full((max(z[1] - z[0], 1), max(dims[1], 1), max(dims[0], 1), 4), 3.0, device="cuda")
        t1 = t0.clone()
        pkg.check(pkg.lib.sdfv_fill_grid(C.byref(prm), 0, C.byref(g), C.c_void_p(t0.data_ptr()), C.c_void_p(t1.data_ptr()), None))
        pkg.check(pkg.lib.sdfv_fill_grid_pass_ex(C.byref(prm), 0, C.byref(g), 2, None, C.c_void_p(t0.data_ptr()),
                                                 C.c_void_p(t1.data_ptr()), None, 0, None))
        pkg.check(pkg.lib.sdfv_grid_init(C.byref(g), C.c_void_p(t0.data_ptr()), C.c_void_p(t1.data_ptr()), None))
        torch.cuda.synchronize()
        assert bool((t0 == 3.0).all()) and bool((t1 == 3.0).all())


def test_single_voxel_axes_reproduce_the_reference_quirk(pkg, oracle):
    """dim == 1 divides 0 by 0 in the position formula (scene/sdf/mod.rs:180): NaN coordinates, same packing."""
    prm = pkg.default_params()
    for dims in [(1, 1, 1), (1, 6, 1), (4, 1, 3)]:
        for sdf_id in (0, 1, 2):  # the sphere alone turns ONE NaN coordinate into a NaN distance (f32::clamp keeps it)
            g = pkg.make_grid(dims)
            t0, t1 = pkg.alloc_textures(g)
            pkg.fill_grid(prm, g, t0, t1, sdf_id=sdf_id)
            pkg.grid_init(g, *(p0 := pkg.alloc_textures(g)))
            pkg.fill_grid_pass(prm, g, 1, p0[0], p0[1], sdf_id=sdf_id)
            torch.cuda.synchronize()
            r0, r1 = oracle.fill_dense(oracle.params_from(prm), dims, threads=1, sdf_id=sdf_id)
            np.testing.assert_array_equal(t0.cpu().numpy().view(np.uint32), r0.view(np.uint32))
            np.testing.assert_array_equal(t1.cpu().numpy().view(np.uint32), r1.view(np.uint32))
            np.testing.assert_array_equal(p0[0].cpu().numpy().view(np.uint32), r0.view(np.uint32))
            np.testing.assert_array_equal(p0[1].cpu().numpy().view(np.uint32), r1.view(np.uint32))


def test_argument_validation_on_device(pkg):
    prm = pkg.default_params()
    g = pkg.make_grid((8, 8, 8))
    t0, t1 = pkg.alloc_textures(g)
    lib = pkg.lib
    misaligned = C.c_void_p(t0.data_ptr() + 4)
    assert lib.sdfv_fill_grid(C.byref(prm), 0, C.byref(g), misaligned, C.c_void_p(t1.data_ptr()), None) == -1
    assert b"16-byte aligned" in lib.sdfv_last_error()
    bad = pkg.default_params(cube_material=5)
    assert lib.sdfv_fill_grid(C.byref(bad), 0, C.byref(g), C.c_void_p(t0.data_ptr()), C.c_void_p(t1.data_ptr()), None) == -1
    assert b"Invalid cube material" in lib.sdfv_last_error()
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at()
    out = torch.zeros((4, 4, 4), device="cuda")
    march = lambda r, y0, y1: pkg._capi.raymarch_rc(r, t0.data_ptr(), t1.data_ptr(), cam, 1, 4, 4, y0, y1, out.data_ptr())  # noqa: E731
    assert march(rp, 3, 2) == -1      # y0 > y1
    assert march(rp, 0, 5) == -1      # y1 > height
    rp.lod_dist_between_samples = 0.5
    assert march(rp, 0, 4) == -1
    huge = pkg.default_render_params(pkg.make_grid((2048, 2048, 2048)))
    assert march(huge, 0, 4) == -1
    assert b"32-bit texel indexing" in lib.sdfv_last_error()


def test_degenerate_images(pkg, oracle):
    prm = pkg.default_params()
    g = pkg.make_grid((16, 16, 16))
    t0, t1 = pkg.alloc_textures(g)
    pkg.fill_grid(prm, g, t0, t1)
    rp = pkg.default_render_params(g)
    # no cameras, no rows: nothing is launched, nothing is touched
    out = torch.full((1, 3, 5, 4), 9.0, device="cuda")
    pkg.check(pkg._capi.raymarch_rc(rp, t0.data_ptr(), t1.data_ptr(), None, 0, 5, 3, 0, 3, out.data_ptr()))
    cam = pkg.camera_look_at(aspect=5 / 3)
    pkg.raymarch(rp, t0, t1, cam, 5, 3, y0=2, y1=2, out=out)
    torch.cuda.synchronize()
    assert bool((out == 9.0).all())
    # a 1x1 image and a one-pixel-wide strip against the oracle
    h0, h1 = t0.cpu().numpy(), t1.cpu().numpy()
    for (w, h) in [(1, 1), (1, 37), (29, 1)]:
        cam = pkg.camera_look_at(aspect=w / h, eye=(0.5, 0.4, 3.0))
        got = pkg.raymarch(rp, t0, t1, cam, w, h)[0].cpu().numpy()
        want, _ = oracle.raymarch(oracle.copy_struct(oracle.RenderParams, rp), h0, h1,
                                  oracle.copy_struct(oracle.Camera, cam), w, h, want_aux=False)
        assert np.abs(got - want).max() <= 1e-4


def test_grid_larger_than_infinity_cache_round_trips_unchanged_elsewhere(pkg):
    """A slab fill must touch exactly its own slices: canaries before and after stay intact (1024x1024 slices)."""
    prm = pkg.default_params()
    dims = (1024, 1024, 1024)            # config 4's global grid; rank 3 of 8 holds slices [384, 512)
    depth = 6                             # a thin slab keeps the test small: slices [509, 515) straddle nothing special
    g = pkg.make_grid(dims, z_begin=509, z_end=509 + depth)
    buf0 = torch.full((depth + 2, 1024, 1024, 4), -5.0, device="cuda")
    buf1 = torch.full((depth + 2, 1024, 1024, 4), -5.0, device="cuda")
    pkg.fill_grid(prm, g, buf0[1:-1], buf1[1:-1])
    torch.cuda.synchronize()
    for b in (buf0, buf1):
        assert bool((b[0] == -5.0).all()) and bool((b[-1] == -5.0).all()) and not bool((b[1:-1] == -5.0).any())
    # slices of the big grid equal the same slices filled one by one
    g1 = pkg.make_grid(dims, z_begin=511, z_end=512)
    s0, s1 = pkg.alloc_textures(g1)
    pkg.fill_grid(prm, g1, s0, s1)
    assert torch.equal(s0[0], buf0[3]) and torch.equal(s1[0], buf1[3])


def test_textures_larger_than_4_gib_use_64_bit_addressing(pkg, oracle):
    """1024 x 1024 x 288 voxels: 4.5 GiB per texture, so byte offsets pass 2^32 (the 512^3 case stays below it)."""
    dims = (1024, 1024, 288)
    prm = pkg.default_params()
    g = pkg.make_grid(dims)
    t0, t1 = pkg.alloc_textures(g)
    assert t0.numel() * 4 > 2 ** 32
    pkg.fill_grid(prm, g, t0, t1)
    torch.cuda.synchronize()
    oprm = oracle.params_from(prm)
    for z in (0, 255, 256, 257, 287):      # slice 256 starts exactly at byte 2^32
        r0, r1 = oracle.fill_dense(oprm, dims, z0=z, z1=z + 1)
        np.testing.assert_array_equal(t0[z:z + 1].cpu().numpy().view(np.uint32), r0.view(np.uint32))
        np.testing.assert_array_equal(t1[z:z + 1].cpu().numpy().view(np.uint32), r1.view(np.uint32))
    # the progressive pass and the compact distance copy address the same range
    pkg.fill_grid_pass(pkg.default_params(sphere_radius=0.5), g, 4, t0, t1, changed_box=(-1, -1, 0.9, 1, 1, 1))
    dist = pkg.commit_distance(g, t0)
    torch.cuda.synchronize()
    assert torch.equal(dist, t0[..., 0])
    assert not torch.equal(t0[284], torch.from_numpy(oracle.fill_dense(oprm, dims, z0=284, z1=285)[0][0]).cuda())
    pkg.fill_grid(prm, g, t0, t1)
    # a small image marched through the far end of the volume (camera behind the last slices) vs the oracle
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at(eye=(0.4, 0.3, 4.0), aspect=1.0)
    got, aux = pkg.raymarch(rp, t0, t1, cam, 48, 48, want_aux=True, dist=pkg.commit_distance(g, t0))
    torch.cuda.synchronize()
    h0, h1 = t0.cpu().numpy(), t1.cpu().numpy()
    want, waux = oracle.raymarch(oracle.copy_struct(oracle.RenderParams, rp), h0, h1,
                                 oracle.copy_struct(oracle.Camera, cam), 48, 48, threads=32)
    a = aux.cpu().numpy().view(oracle.AUX_DTYPE).reshape(48, 48)
    np.testing.assert_array_equal(a["status"], waux["status"])
    np.testing.assert_array_equal(a["steps"], waux["steps"])
    np.testing.assert_array_equal(a["hit_pos"].view(np.uint32), waux["hit_pos"].view(np.uint32))
    assert np.abs(got[0].cpu().numpy() - want).max() <= 1e-4 and (waux["status"] == 1).sum() > 100


def test_enqueue_calls_are_graph_capture_safe(pkg):
    """fill + commit + raymarch captured into a hipGraph (no allocation, no synchronisation inside the library's
    enqueue calls) and replayed: same bits as the eager frame."""
    side, W, H = 32, 96, 64
    prm = pkg.default_params()
    g = pkg.make_grid((side, side, side))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((side, side, side), dtype=torch.float32, device="cuda")
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at(aspect=W / H)
    out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")

    def frame():
        pkg.fill_grid(prm, g, t0, t1)
        pkg.commit_distance(g, t0, dist=dist)
        pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)

    frame()
    torch.cuda.synchronize()
    ref = out.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        frame()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        frame()
    for t in (t0, t1, dist, out):
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32)) and bool((ref[..., 3] > 0).any())


def test_two_host_threads_on_their_own_streams(pkg, oracle):
    """The library keeps no shared mutable state besides thread-local error text and per-thread scratch: two host
    threads driving different grids on different streams at the same time get the results they would get alone."""
    import threading
    results, errors = {}, []

    def work(tag, dims, kw, sdf_id):
        try:
            stream = torch.cuda.Stream()
            prm = pkg.default_params(**kw)
            g = pkg.make_grid(dims)
            with torch.cuda.stream(stream):
                t0, t1 = pkg.alloc_textures(g)
                for _ in range(20):
                    pkg.fill_grid(prm, g, t0, t1, sdf_id=sdf_id, stream=stream)
                    rgba = pkg.raymarch(pkg.default_render_params(g), t0, t1, pkg.camera_look_at(aspect=1.5), 96, 64,
                                        stream=stream)
                    v, i = pkg.mesh_extract(prm, 12, sdf_id=sdf_id, stream=stream)
                    bad = pkg.lib.sdfv_fill_grid(None, 0, None, None, None, None)  # sets THIS thread's error text
                    assert bad == -1
            stream.synchronize()
            results[tag] = (t0.cpu().numpy(), rgba.cpu().numpy(), v.cpu().numpy(), i.cpu().numpy(), prm, dims, sdf_id)
        except Exception as e:  # noqa: BLE001
            errors.append((tag, repr(e)))

    threads = [threading.Thread(target=work, args=("a", (40, 36, 28), dict(), 0)),
               threading.Thread(target=work, args=("b", (33, 17, 45), dict(cube_material=1, sphere_radius=0.8), 2))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for tag, (t0, rgba, v, i, prm, dims, sdf_id) in results.items():
        r0, _ = oracle.fill_dense(oracle.params_from(prm), dims, sdf_id=sdf_id)
        np.testing.assert_array_equal(t0.view(np.uint32), r0.view(np.uint32))
        g = pkg.make_grid(dims)
        s0, s1 = pkg.alloc_textures(g)
        pkg.fill_grid(prm, g, s0, s1, sdf_id=sdf_id)
        alone = pkg.raymarch(pkg.default_render_params(g), s0, s1, pkg.camera_look_at(aspect=1.5), 96, 64)
        np.testing.assert_array_equal(rgba.view(np.uint32), alone.cpu().numpy().view(np.uint32))
        va, ia = pkg.mesh_extract(prm, 12, sdf_id=sdf_id)
        np.testing.assert_array_equal(v.view(np.uint32), va.cpu().numpy().view(np.uint32))
        np.testing.assert_array_equal(i, ia.cpu().numpy())
