"""The ingest path: SDFViewer::update for ANY SDFSurface (reference src/app/scene/sdf/mod.rs:128 `sdf: impl SDFSurface`), not
only the SDF the device can evaluate.  The host samples (sample() is arbitrary user code: a wasm / FFI provider,
src/sdf/wasm/native.rs:188-217), the device packs (sdfv_pack_samples = scene/sdf/mod.rs:196-208).  Checked bit for bit against
the oracle's update loop fed the SAME samples (oracle/grid_fill.c or_viewer_update_fn)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def oracle_pack(oracle, samples):
    """scene/sdf/mod.rs:196-208 per record -> (tex0 [n, 4], tex1 rgb [n, 3])."""
    n = len(samples)
    t0, t1 = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    for i in range(n):
        oracle.L.or_pack_sample(samples[i].ctypes.data, t0[i].ctypes.data, t1[i].ctypes.data)
    return t0, t1[:, :3]


def special_samples(rng, n):
    s = rng.uniform(-0.6, 1.6, size=(n, 7)).astype(np.float32)
    s[:, 0] = rng.uniform(-0.4, 1.2, size=n)                      # both clamps of 0.1 + d
    s[0, 1:4] = 0.0                                               # all-zero colour -> 0.5 grey
    s[1, 1:4] = [-0.0, 0.0, -0.0]                                 # -0 == 0 as well
    s[2, 0] = np.float32(np.nan)                                  # f32::clamp keeps a NaN
    s[3, 0] = np.inf
    s[4, 0] = -np.inf
    s[5, 6] = 0.0                                                 # occlusion <= 0 -> 1
    s[6, 6] = -3.0
    s[7, 1:4] = [np.nan, 2.0, 0.999999]                           # `as u8`: NaN -> 0, saturation
    s[8, 0] = np.float32(0.001234)                                # 0.1 + d == AIR_DIST exactly
    s[9, 1:4] = [1.0 / 255.0 * 0.999, 254.9999 / 255.0, 0.5]      # truncation boundaries
    return s


@pytest.mark.parametrize("layout", ["none", "plain", "interleaved"])
def test_pack_samples_is_updates_packing(pkg, oracle, layout):
    dims = (12, 10, 6)
    g = pkg.make_grid(dims)
    n_vox = dims[0] * dims[1] * dims[2]
    rng = np.random.default_rng(7)
    n = 333                                                       # one whole workgroup + a partial one
    idx = rng.permutation(n_vox)[:n].astype(np.uint32)
    smp = special_samples(rng, n)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    dist = None
    if layout != "none":
        dist = torch.full((n_vox,), pkg.AIR_DIST, dtype=torch.float32, device="cuda")
    flags = pkg._capi.PASS_VOLUME_INTERLEAVED if layout == "interleaved" else 0
    pkg.pack_samples(g, torch.from_numpy(smp).cuda(), t0, t1, indices=torch.from_numpy(idx.view(np.int32)).cuda(), dist=dist,
                     flags=flags)
    torch.cuda.synchronize()
    e0, e1 = oracle.grid_init(dims)
    p0, p1 = oracle_pack(oracle, smp)
    e0.reshape(-1, 4)[idx] = p0
    e1.reshape(-1, 4)[idx, :3] = p1                               # .a keeps AIR_DIST: never written (scene/sdf/mod.rs:205-208)
    assert np.array_equal(bits(t0.cpu().numpy()), bits(e0))
    assert np.array_equal(bits(t1.cpu().numpy()), bits(e1))
    if dist is not None:
        d = dist.cpu().numpy()
        want = e0.reshape(-1, 4)[:, 0]
        if layout == "interleaved":                               # entry ((row >> 1) * W + x) * 2 + (row & 1)
            W = dims[0]
            d = d.reshape(-1, W, 2).transpose(0, 2, 1).reshape(-1)
        assert np.array_equal(bits(d), bits(want))


def test_pack_samples_runs_slabs_and_strays(pkg, oracle):
    """indices == NULL: a contiguous run from index_base; the flat index is relative to the SLAB the pointers address; a record
    addressed beyond the slab is skipped; n == 0 is fine; bad arguments are refused."""
    dims = (9, 4, 7)
    g = pkg.make_grid(dims, z_begin=2, z_end=5)
    slab_vox = 9 * 4 * 3
    rng = np.random.default_rng(3)
    smp = special_samples(rng, 40)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    pkg.pack_samples(g, torch.from_numpy(smp).cuda(), t0, t1, index_base=slab_vox - 25)  # records 25.. fall off the slab
    idx = np.array([5, 0xFFFFFFFF, 17, slab_vox, 3], np.uint32)   # two strays
    pkg.pack_samples(g, torch.from_numpy(smp[:5].copy()).cuda(), t0, t1, indices=torch.from_numpy(idx.view(np.int32)).cuda())
    pkg.pack_samples(g, torch.zeros((0, 7), device="cuda"), t0, t1)
    torch.cuda.synchronize()
    p0, p1 = oracle_pack(oracle, smp)
    e0 = np.full((slab_vox, 4), oracle.AIR_DIST, np.float32)
    e1 = e0.copy()
    e0[slab_vox - 25:] = p0[:25]
    e1[slab_vox - 25:, :3] = p1[:25]
    for k, i in enumerate(idx):
        if i < slab_vox:
            e0[i] = p0[k]
            e1[i, :3] = p1[k]
    assert np.array_equal(bits(t0.cpu().numpy()).reshape(-1, 4), bits(e0))
    assert np.array_equal(bits(t1.cpu().numpy()).reshape(-1, 4), bits(e1))
    with pytest.raises(pkg.SdfvError, match="INTERLEAVED without a volume"):
        pkg.pack_samples(g, torch.from_numpy(smp).cuda(), t0, t1, flags=pkg._capi.PASS_VOLUME_INTERLEAVED)
    with pytest.raises(pkg.SdfvError, match="unknown flags"):
        pkg.pack_samples(g, torch.from_numpy(smp).cuda(), t0, t1, flags=1)


def test_pack_samples_honours_the_srgb_policy(pkg, oracle):
    """SDFV_OPT_EXT_SRGB_QUANT (the one [EXT] piece with visible consequences) applies to ingested samples like to fills."""
    dims = (8, 8, 4)
    g = pkg.make_grid(dims)
    rng = np.random.default_rng(11)
    smp = special_samples(rng, 256)
    t0, t1 = pkg.alloc_textures(g)
    pkg.grid_init(g, t0, t1)
    with pkg.options({pkg._capi.OPT_EXT_SRGB_QUANT: 1}):
        pkg.pack_samples(g, torch.from_numpy(smp).cuda(), t0, t1)
    torch.cuda.synchronize()
    oracle.L.or_set_ext_variant(4)                                # OR_EXT_SRGB_QUANT_ROUND
    try:
        p0, _ = oracle_pack(oracle, smp)
    finally:
        oracle.L.or_set_ext_variant(0)
    q0, _ = oracle_pack(oracle, smp)
    assert np.array_equal(bits(t0.cpu().numpy()).reshape(-1, 4), bits(p0)) and not np.array_equal(bits(p0), bits(q0))


class RefViewer:
    """SDFViewer::update's bookkeeping (scene/sdf/mod.rs:130-156) around the oracle's loop, fed by a C sample function."""

    def __init__(self, oracle, dims, bb, passes, sample_fn, t0=None, t1=None):
        self.o, self.dims, self.bb, self.fn = oracle, dims, bb, sample_fn
        self.t0, self.t1 = oracle.grid_init(dims) if t0 is None else (t0, t1)
        self.lm = oracle.lm_new(dims, passes)
        self.box, self.box_while_loading = None, False

    def update(self, new_box, iterations):
        o = self.o
        just = False
        if new_box is not None:
            nb = np.asarray(new_box, np.float32)
            self.box = nb if self.box is None else np.concatenate([np.minimum(self.box[:3], nb[:3]), np.maximum(self.box[3:], nb[3:])])
            self.box_while_loading = o.L.or_lm_len(C.byref(self.lm)) > 0 or self.box_while_loading
            just = True
        if self.box is not None and o.L.or_lm_len(C.byref(self.lm)) == 0:
            self.lm = o.lm_new(self.dims, 3)
            if not just:
                if not self.box_while_loading:
                    self.box = None
                self.box_while_loading = False
        return o.viewer_update_fn(self.fn, self.dims, self.lm, self.t0, self.t1, changed_box=self.box, max_iterations=iterations,
                                  bb_min=self.bb[:3], bb_max=self.bb[3:])


def assert_viewer_equals(v, ref, what):
    t0, t1 = v.download()
    assert np.array_equal(bits(t0), bits(ref.t0)), what
    assert np.array_equal(bits(t1), bits(ref.t1)), what


@pytest.mark.parametrize("threads,capacity,layout", [(1, 64, "plain"), (3, 1000, "interleaved"), (0, 0, "auto")])
def test_host_sdf_loads_progressively_like_the_reference(host, oracle, gyroid_provider, threads, capacity, layout):
    """a host-only SDF (the gyroid behind the per-point ABI) loads through SDFViewer::update under
    assorted time budgets; after EVERY call the textures equal the oracle's loop advanced by the iterations the call reported --
    all three passes, then a parameter edit whose changed box covers part of the grid (3-pass re-sampling, mod.rs:146-156)."""
    sdf = host.SDF.provider(gyroid_provider)
    raw = C.CDLL(gyroid_provider)
    bb = sdf.bounding_box()
    dims = (24, 12, 18)
    v = host.Viewer.new_voxels(dims, bb, 3, layout=layout)
    assert v.dims() == dims
    v.set_ingest(threads, capacity)
    ref = RefViewer(oracle, dims, bb, 3, raw.gyroid_sample_raw)
    rng = np.random.default_rng(threads)
    budgets = [0.0, 1e-6, 5e-6, 2e-5]
    calls = 0
    while v.remaining():
        n = v.update(sdf, budgets[rng.integers(len(budgets))])
        assert n > 0 and v.last_error() == ""                     # "performs at least one update", mod.rs:126-127
        assert ref.update(None, n) == n
        calls += 1
        if calls % 7 == 0 or not v.remaining():
            assert_viewer_equals(v, ref, f"call {calls}")
        assert v.lod() == 2.0 ** oracle.L.or_lm_passes_left(C.byref(ref.lm))
    assert calls > 3 and v.lod() == 1.0
    # the loaded grid renders like the oracle's march over the same textures
    v.commit()
    img = v.render(96, 64)
    rp = oracle.default_render_params(dims, bb[:3], bb[3:])
    want, _ = oracle.raymarch(rp, ref.t0, ref.t1, oracle.camera_look_at(aspect=96 / 64), 96, 64, want_aux=False)
    assert np.abs(img - want).max() <= 1e-4
    # an edit: the provider reports a box that covers PART of the grid, once
    assert sdf.set_parameter(0, 0.3) is None
    box = np.float32([bb[0], -0.3, bb[2], 0.1, bb[4], bb[5]])
    first = True
    for _ in range(10000):
        n = v.update(sdf, budgets[rng.integers(len(budgets))])
        assert ref.update(box if first else None, n) == n
        first = False
        calls += 1
        if calls % 5 == 0:
            assert_viewer_equals(v, ref, f"edit call {calls}")
        if n == 0 and not v.has_changed_box():
            break
    assert not v.has_changed_box() and ref.box is None
    assert_viewer_equals(v, ref, "after the edit")
    before = oracle.grid_init(dims)[0]
    assert (bits(ref.t0) != bits(before)).any()
    assert sdf.set_parameter(0, 0.15) is None and sdf.changed() is not None   # restore the fixture's default


def test_demo_through_the_per_point_abi_equals_the_device_fill(host, pkg, oracle):
    """the demo SDF consumed as an ORDINARY provider (libsdfdemo_provider.so loaded through ProviderSDF:
    no device form, every sample a call through include/sdf_provider.h) fills the same textures as sdfv_fill_grid."""
    sdf = host.SDF.provider(host.PROVIDER_PATH)
    dims = (52, 52, 52)   # (the provider exports `sample_batch`: a gathered block of voxels is ONE device batch)
    v = host.Viewer.new_voxels(dims, sdf.bounding_box(), 2)
    while v.update(sdf, 0.05):
        pass
    assert v.last_error() == "" and v.lod() == 1.0
    t0, t1 = v.download()
    g = pkg.make_grid(dims)
    d0, d1 = pkg.alloc_textures(g)
    pkg.fill_grid(pkg.default_params(), g, d0, d1)
    torch.cuda.synchronize()
    assert np.array_equal(bits(t0), bits(d0.cpu().numpy())) and np.array_equal(bits(t1), bits(d1.cpu().numpy()))
    r0, r1 = oracle.fill_dense(oracle.default_params(), dims)
    assert np.array_equal(bits(t0), bits(r0)) and np.array_equal(bits(t1), bits(r1))


@pytest.mark.parametrize("layout", ["plain", "interleaved"])
def test_ingest_after_the_device_path_rebuilds_the_host_mirror(host, oracle, gyroid_provider, layout):
    """One viewer, two kinds of SDF: the demo loads on the device, then a host-only SDF reports a change.  update_required for the
    host pass is decided on a mirror of tex0.r that must be fetched from the device's distance volume (either layout) first."""
    gy = host.SDF.provider(gyroid_provider)
    raw = C.CDLL(gyroid_provider)
    bb = gy.bounding_box()
    dims = (16, 8, 12)
    v = host.Viewer.new_voxels(dims, bb, 2, layout=layout)
    demo = host.SDF.demo()
    while v.update(demo, 1.0):
        pass
    r0, r1 = oracle.fill_dense(oracle.default_params(), dims, bb[:3], bb[3:])
    assert np.array_equal(bits(v.download()[0]), bits(r0))
    ref = RefViewer(oracle, dims, bb, 2, raw.gyroid_sample_raw, r0.copy(), r1.copy())
    while o_next := oracle.lm_next(ref.lm):                        # (the reference manager is exhausted like the viewer's)
        pass
    assert gy.set_parameter(0, 0.2) is None
    box = np.float32([bb[0], -0.3, bb[2], 0.1, bb[4], bb[5]])
    first = True
    for _ in range(1000):
        n = v.update(gy, 1e-3)
        assert v.last_error() == ""
        assert ref.update(box if first else None, n) == n
        first = False
        if n == 0 and not v.has_changed_box():
            break
    assert_viewer_equals(v, ref, "gyroid box over the demo grid")
    assert (bits(ref.t0) != bits(r0)).any() and (bits(ref.t0) == bits(r0)).any()   # part resampled, part kept
    # ... and back: the device path after the ingest path (the demo reports its whole box on an edit)
    assert demo.children()[1].set_parameter(1, 0.7) is None
    while v.update(demo, 1.0) or v.has_changed_box():
        pass
    e0, e1 = oracle.fill_dense(oracle.default_params(sphere_radius=0.7), dims, bb[:3], bb[3:])
    t0, t1 = v.download()
    assert np.array_equal(bits(t0), bits(e0)) and np.array_equal(bits(t1), bits(e1))
    assert gy.set_parameter(0, 0.15) is None and gy.changed() is not None


def test_provider_with_batched_sampling_loads_the_same_grid(host, oracle, gyroid_provider_batch):
    """A provider that exports `sample_batch` is sampled through it (one call per gathered block, no allocation per point): the
    same textures, call by call, as the oracle's per-voxel loop."""
    sdf = host.SDF.provider(gyroid_provider_batch)
    raw = C.CDLL(gyroid_provider_batch)
    bb = sdf.bounding_box()
    dims = (48, 24, 36)
    for threads, budget in ((1, 2e-4), (0, 1e-3)):
        v = host.Viewer.new_voxels(dims, bb, 2)
        v.set_ingest(threads, 0)
        ref = RefViewer(oracle, dims, bb, 2, raw.gyroid_sample_raw)
        calls = 0
        while v.remaining():
            n = v.update(sdf, budget)
            assert n > 0 and v.last_error() == "" and ref.update(None, n) == n
            calls += 1
            if calls % 5 == 0:
                assert_viewer_equals(v, ref, f"call {calls}")
        assert_viewer_equals(v, ref, "loaded")


def test_meshing_a_host_only_sdf_stays_refused(host, gyroid_provider):
    """the mesher front end has no CPU path (host/mesh.hpp; meshers are out of scope)."""
    with pytest.raises(RuntimeError):
        host.Mesh.from_sdf(host.SDF.provider(gyroid_provider))


def test_cli_url_provider_loads_through_the_ingest_path(oracle, gyroid_provider, tmp_path):
    """`sdf-viewer-gpu app --max-voxels-side 40 --loading-passes 3 url file://libgyroid_provider.so`: the reference's second
    provider (CliSDFProvider::Url, app/cli/mod.rs:41-46) with a native library where the wasm file stands -- the textures it loads
    equal the oracle's loop over the same sample function, the frame it writes the oracle's march over them."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdf-viewer_amd", "sdf-viewer-gpu")
    out, dump = tmp_path / "frame.ppm", tmp_path / "grid"
    r = subprocess.run([exe, "app", "--max-voxels-side", "40", "--loading-passes", "3", "url", "file://" + gyroid_provider,
                        "--width", "160", "--height", "90", "--out", str(out), "--dump-textures", str(dump)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Using 40x20x30 voxels" in r.stderr and "Loaded last SDF chunk (lod 1)" in r.stderr
    raw = C.CDLL(gyroid_provider)
    bb = np.float32([-1, -0.5, -0.75, 1, 0.5, 0.75])
    dims = (40, 20, 30)
    ref = RefViewer(oracle, dims, bb, 3, raw.gyroid_sample_raw)
    ref.update(None, 2 ** 40)
    t0 = np.fromfile(str(dump) + ".tex0.f32", np.float32).reshape(ref.t0.shape)
    t1 = np.fromfile(str(dump) + ".tex1.f32", np.float32).reshape(ref.t1.shape)
    assert np.array_equal(bits(t0), bits(ref.t0)) and np.array_equal(bits(t1), bits(ref.t1))
    want, _ = oracle.raymarch(oracle.default_render_params(dims, bb[:3], bb[3:]), ref.t0, ref.t1, oracle.camera_look_at(aspect=160 / 90),
                              160, 90, want_aux=False)
    data = open(out, "rb").read()
    header = b"P6\n160 90\n255\n"
    img = np.frombuffer(data[len(header):], np.uint8).reshape(90, 160, 3).astype(np.int32)
    want8 = np.rint(np.clip(np.nan_to_num(want[..., :3] * want[..., 3:4]), 0, 1) * 255).astype(np.int32)
    assert np.abs(img - want8).max() <= 1


def test_a_provider_whose_sample_fails_loads_as_the_default_sample(host, oracle, failing_provider):
    """wasm/native.rs:172,203: a failed bounding_box() is [0, 1]^3, a failed sample() is SDFSample::new(1.0, 0) -- and the viewer
    loads THAT, like the reference would: distance 1.0 everywhere (tex0.r clamps to 1), the all-zero colour replaced by grey."""
    sdf = host.SDF.provider(failing_provider)
    raw = C.CDLL(failing_provider)
    raw.set_fail_mode(0)
    bb = sdf.bounding_box()
    assert np.array_equal(bb, np.float32([0, 0, 0, 1, 1, 1]))
    dims = (10, 6, 8)
    v = host.Viewer.new_voxels(dims, bb, 2)
    ref = RefViewer(oracle, dims, bb, 2, raw.failing_sample_raw)
    while v.remaining():
        n = v.update(sdf, 1.0)
        assert n > 0 and v.last_error() == "" and ref.update(None, n) == n
    assert_viewer_equals(v, ref, "default samples")
    t0, t1 = v.download()
    assert (t0[..., 0] == 1.0).all() and (t1[..., 2] == 1.0).all()   # clamp(0.1 + 1.0); occlusion 0 -> 1
    assert np.unique(t0[..., 1:]).size == 1                           # one grey


@pytest.mark.parametrize("dims,passes", [((2, 2, 2), 1), ((4, 4, 4), 5), ((1, 5, 7), 2), ((9, 1, 1), 3), ((33, 2, 3), 2)])
def test_ingest_over_degenerate_grids(host, oracle, gyroid_provider, dims, passes):
    """Grids the reference's loop handles without noticing: a single voxel along an axis (its coordinate is 0 / 0 = NaN,
    scene/sdf/mod.rs:179-182 -- the sample function is handed the NaN, the box test fails on it), more passes than the grid has
    levels (the coarse passes visit voxel (0, 0, 0) only), rows shorter than a wave.  Whole loads under a zero budget and a generous
    one, one thread and several."""
    sdf = host.SDF.provider(gyroid_provider)
    raw = C.CDLL(gyroid_provider)
    bb = sdf.bounding_box()
    for threads, budget in ((1, 0.0), (4, 1.0)):
        v = host.Viewer.new_voxels(dims, bb, passes)
        v.set_ingest(threads, 0)
        ref = RefViewer(oracle, dims, bb, passes, raw.gyroid_sample_raw)
        total = 0
        while v.remaining():
            n = v.update(sdf, budget)
            assert n > 0 and v.last_error() == ""
            assert ref.update(None, n) == n
            total += n
        assert total == sum(-(-dims[0] // s) * -(-dims[1] // s) * -(-dims[2] // s) for s in (2 ** k for k in range(passes)))
        assert_viewer_equals(v, ref, (dims, passes, threads))
        assert v.update(sdf, budget) == 0 and v.lod() == 1.0


def test_randomised_ingest_loads_and_edits(host, oracle, gyroid_provider, gyroid_provider_batch):
    """Seeded sweep (tools/soak.sh varies the seed): random grid shapes, pass counts, worker counts, buffer sizes, budgets and
    volume layouts; a load, then zero to two parameter edits, each worked off to the end -- every state compared with the oracle's
    loop, after random calls and at the end of every phase."""
    import os
    seed = int(os.environ.get("SDFV_SOAK_SEED", 11))
    trials = int(os.environ.get("SDFV_SOAK_TRIALS", 6))
    rng = np.random.default_rng(seed)
    for trial in range(max(2, trials // 4)):
        lib = gyroid_provider_batch if rng.integers(2) else gyroid_provider
        sdf, raw = host.SDF.provider(lib), C.CDLL(lib)
        bb = sdf.bounding_box()
        dims = tuple(int(v) for v in rng.integers(1, 41, size=3))
        layout = "plain"
        if dims[1] % 2 == 0 and rng.integers(2):
            layout = "interleaved"
        passes = int(rng.integers(1, 5))
        threads, capacity = int(rng.choice([1, 2, 5, 0])), int(rng.choice([0, 7, 64, 1000]))
        budgets = [0.0, 1e-6, 2e-5, 3e-4]
        what = (seed, trial, dims, layout, passes, threads, capacity)
        v = host.Viewer.new_voxels(dims, bb, passes, layout=layout)
        v.set_ingest(threads, capacity)
        ref = RefViewer(oracle, dims, bb, passes, raw.gyroid_sample_raw)
        while v.remaining():
            n = v.update(sdf, budgets[rng.integers(4)])
            assert n > 0 and v.last_error() == "", what
            assert ref.update(None, n) == n, what
            if rng.integers(8) == 0:
                assert_viewer_equals(v, ref, what)
        assert_viewer_equals(v, ref, what)
        box = np.float32([bb[0], -0.3, bb[2], 0.1, bb[4], bb[5]])
        for edit in range(int(rng.integers(0, 3))):
            assert sdf.set_parameter(0, float(rng.uniform(0.05, 0.4))) is None
            first = True
            for _ in range(100000):
                n = v.update(sdf, budgets[rng.integers(4)])
                assert ref.update(box if first else None, n) == n, what
                first = False
                if n == 0 and not v.has_changed_box():
                    break
            assert not v.has_changed_box() and ref.box is None, what
            assert_viewer_equals(v, ref, what + (edit,))
        assert sdf.set_parameter(0, 0.15) is None and sdf.changed() is not None   # the fixture's default for the next trial
