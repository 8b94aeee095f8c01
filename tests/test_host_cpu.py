"""CPU tests of the C++ host mirror: LoadingManager (the reference's own unit tests, loading.rs:117-171, run
against the product's class), SDFDemo hierarchy / parameters / changed() semantics, provider exports."""
import ctypes as C

import numpy as np
import pytest


# ---- loading.rs:117-171 restated against sdfviewer::LoadingManager ----
def loading_impl(host, limits):
    hits = np.zeros(limits[0] * limits[1] * limits[2], np.int64)
    num_passes = 3
    manager = host.LoadingManager(limits, num_passes)
    remaining = manager.len()
    iterations = 0
    total = iterations + remaining
    while True:
        v = manager.next()
        if v is None:
            break
        flat = v[0] + v[1] * limits[0] + v[2] * limits[0] * limits[1]
        hits[flat] += 1
        assert hits[flat] <= num_passes
        iterations += 1
        remaining = manager.len()
        assert total == iterations + remaining
    assert (hits >= 1).all(), "developer error: voxel was not hit"


def test_interlacing_cube_2(host):
    loading_impl(host, (2, 2, 2))


def test_interlacing_cube_8(host):
    loading_impl(host, (8, 8, 8))


def test_interlacing_cube_64(host):
    loading_impl(host, (64, 64, 64))


def test_interlacing_cube_11(host):
    loading_impl(host, (11, 11, 11))


def test_interlacing_non_cube(host):
    loading_impl(host, (8, 11, 17))


def test_loading_manager_matches_oracle_sequence(host, oracle):
    for limits, passes in [((5, 3, 4), 3), ((8, 8, 8), 2), ((7, 1, 2), 4), ((3, 3, 3), 0), ((3, 3, 3), 1)]:
        m = host.LoadingManager(limits, passes)
        o = oracle.lm_new(limits, passes)
        while True:
            assert m.len() == oracle.L.or_lm_len(C.byref(o))
            assert m.passes_left() == oracle.L.or_lm_passes_left(C.byref(o))
            a, b = m.next(), oracle.lm_next(o)
            assert a == b
            if a is None:
                break
        assert m.total_iterations() == o.total_iterations


def test_loading_manager_walks_a_grid_without_voxels_like_the_reference(host, oracle):
    """An axis of 0 voxels: the reference hands out index 0 on it anyway (it returns the index before it tests the limit,
    loading.rs:57-75); the pass / cursor form keeps that walk.  (len() underflows there in the reference: not compared.)"""
    for limits, passes in [((0, 3, 2), 2), ((2, 0, 0), 3), ((0, 0, 0), 2)]:
        m, o = host.LoadingManager(limits, passes), oracle.lm_new(limits, passes)
        for _ in range(64):
            assert m.passes_left() == oracle.L.or_lm_passes_left(C.byref(o))
            a, b = m.next(), oracle.lm_next(o)
            assert a == b, (limits, a, b)
            if a is None:
                break
        assert a is None and m.total_iterations() == o.total_iterations


def test_finish_pass_equals_stepping(host):
    a, b = host.LoadingManager((9, 7, 5), 3), host.LoadingManager((9, 7, 5), 3)
    for _ in range(4):  # partially consume the first pass of a
        a.next()
        b.next()
    while a.step_size():
        step = a.step_size()
        n = a.finish_pass()
        for _ in range(n):
            b.next()
        assert (a.step_size(), a.len(), a.total_iterations(), a.passes_left()) == \
               (b.step_size(), b.len(), b.total_iterations(), b.passes_left()), step
    assert a.next() is None and a.finish_pass() == 0
    assert [host.H.sdfvh_prev_power_of_2(x) for x in (0, 1, 3, 8, 9)] == [0, 1, 2, 8, 8]


# ---- SDFDemo hierarchy and parameters: demo/mod.rs:78-144, cube.rs:91-161, sphere.rs:49-119 ----
def test_demo_hierarchy(host, pkg):
    d = host.SDF.demo()
    assert (d.id(), d.name()) == (0, "Demo")
    ch = d.children()
    assert [(c.id(), c.name()) for c in ch] == [(1, "DemoCube"), (2, "DemoSphere")]
    assert list(d.bounding_box()) == [-1, -1, -1, 1, 1, 1]
    prm, sid = d.device_params(pkg.DemoParams)
    assert bytes(prm) == bytes(pkg.default_params()) and sid == 0
    assert ch[1].device_params(pkg.DemoParams)[1] == 2


def test_demo_parameters_listing(host):
    d = host.SDF.demo()
    ps = d.parameters()
    assert [(p[0], p[1], p[2], p[3]) for p in ps] == [("0", "max_distance_custom_material", "2", "Float(0.05)"),
                                                       ("1", "disable_sphere", "0", "Boolean(false)")]
    cube, sphere = d.children()
    assert [(p[1], p[2], p[3]) for p in cube.parameters()] == [("material", "3", 'String("Brick")'),
                                                               ("half_side", "1", "Int(95)")]  # (0.95f32 * 100.) as i32: the f32 product rounds to 95.0
    assert [(p[1], p[2], p[3]) for p in sphere.parameters()] == [("material", "3", 'String("Normal")'),
                                                                 ("sphere_radius", "2", "Float(1.05)")]


def test_set_parameter_shares_state_and_reports_changed(host, pkg):
    d = host.SDF.demo()
    cube, sphere = d.children()
    assert d.changed() is None
    assert sphere.set_parameter(1, 0.8) is None          # ID_RADIUS
    assert cube.set_parameter(0, "normal") is None       # ID_MATERIAL, case-insensitive FromStr
    assert cube.set_parameter(1, 50) is None             # ID_HALF_SIDE: value as f32 / 100
    prm, _ = d.device_params(pkg.DemoParams)             # the root sees its children's modifications
    assert np.float32(prm.sphere_radius) == np.float32(0.8) and prm.cube_material == 1
    assert np.float32(prm.cube_half_side) == np.float32(0.5)
    # changed_default_impl: one child per call, then the own flag, then None
    assert list(d.changed()) == [-1, -1, -1, 1, 1, 1]    # cube
    assert list(d.changed()) == [-1, -1, -1, 1, 1, 1]    # sphere
    assert d.changed() is None
    assert d.set_parameter(1, True) is None
    assert d.changed() is not None and d.changed() is None
    assert d.device_params(pkg.DemoParams)[0].disable_sphere == 1


def test_set_parameter_errors(host):
    d = host.SDF.demo()
    assert d.set_parameter(7, 1.0) == "Unknown parameter 7 with value Float(1.0)"
    assert d.set_parameter(0, True) == "Unknown parameter 0 with value Boolean(true)"   # wrong kind
    cube = d.children()[0]
    assert cube.set_parameter(1, 0.5) == "Unknown parameter 1 with value Float(0.5)"    # half_side wants Int
    assert cube.set_parameter(0, "marble") == "Invalid cube material"
    assert d.changed() is None


def test_demo_from_cli_flags(host, pkg):
    d = host.SDF.demo("-t", "normal", "--sphere-radius=0.9", "-m", "0.1", "-d", "true", "-c", "0.7", "-l", "BRICK")
    prm, _ = d.device_params(pkg.DemoParams)
    assert (prm.cube_material, prm.sphere_material, prm.disable_sphere) == (1, 0, 1)
    assert [np.float32(x) for x in (prm.cube_half_side, prm.sphere_radius, prm.max_distance_custom_material)] == \
           [np.float32(0.7), np.float32(0.9), np.float32(0.1)]
    with pytest.raises(ValueError, match="Invalid cube material"):
        host.SDF.demo("-t", "marble")
    with pytest.raises(ValueError, match="wasn't expected"):
        host.SDF.demo("--nope")


def test_provider_exports_reference_symbols(host):
    import os
    import re
    raw = C.CDLL(host.PROVIDER_PATH)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sdf_provider.h")).read()
    body = hdr[hdr.index("void init(void);"):]
    declared = set(re.findall(r"\b([a-z_]+)\s*\(", re.sub(r"/\*.*?\*/", "", body, flags=re.S)))
    assert set(host.PROVIDER_SYMBOLS) <= declared and "init_with_args" in declared
    for name in declared:            # every function include/sdf_provider.h declares is exported ...
        getattr(raw, name)
    for name in host.PROVIDER_SYMBOLS:  # ... and they are the reference's own names (ffi.rs:42-337)
        getattr(raw, name)
    assert C.sizeof(host.ParamC) == 88 and C.sizeof(host.ParamValueC) == 24 and C.sizeof(host.ParamKindC) == 24


def test_provider_metadata_calls_without_gpu(host):
    """children / name / parameters / set_parameter / changed need no device."""
    P = host.load_provider()
    P.init()
    pl = P.children(0)
    assert np.frombuffer(host.pl_bytes(pl.contents), np.uint32).tolist() == [1, 2]
    P.children_free(pl)
    for sid, want in ((0, b"Demo"), (1, b"DemoCube"), (2, b"DemoSphere")):
        pl = P.name(sid)
        assert host.pl_bytes(pl.contents) == want
        P.name_free(pl)
    pl = P.name(9)  # unknown id: stderr + null payload (ffi.rs:137-140)
    assert pl.contents.ptr is None and pl.contents.len_bytes == 0
    P.name_free(pl)
    pl = P.parameters(1)
    n = pl.contents.len_bytes // C.sizeof(host.ParamC)
    params = C.cast(pl.contents.ptr, C.POINTER(host.ParamC * n)).contents
    assert n == 2 and host.pl_bytes(params[0].name) == b"material" and params[0].kind.tag == 3
    choices = params[0].kind.v.choices
    items = C.cast(choices.ptr, C.POINTER(host.PointerLength * 2)).contents
    assert [host.pl_bytes(i) for i in items] == [b"Brick", b"Normal"]
    assert host.pl_bytes(params[0].value.v.string_) == b"Brick"
    assert params[1].kind.tag == 1 and (params[1].kind.v.int_.range_start, params[1].kind.v.int_.range_end) == (0, 100)
    assert params[1].value.v.int_ == 95
    P.parameters_free(pl)
    v = host.ParamValueC()
    v.tag = 2
    v.v.float_ = 0.75
    r = P.set_parameter(2, 1, v)
    assert r.contents.tag == 0
    P.set_parameter_free(r)
    r = P.set_parameter(2, 5, v)
    assert r.contents.tag == 1 and host.pl_bytes(r.contents.error) == b"Unknown parameter 5 with value Float(0.75)"
    P.set_parameter_free(r)
    c = P.changed(0)
    assert c.contents.tag == 1 and list(c.contents.bounds) == [-1, -1, -1, 1, 1, 1]
    P.changed_free(c)
    c = P.changed(0)
    assert c.contents.tag == 0
    P.changed_free(c)
    bb = P.bounding_box(2)
    assert list(bb.contents) == [-1, -1, -1, 1, 1, 1]
    P.bounding_box_free(bb)
    bb = P.bounding_box(42)
    assert list(bb.contents) == [0] * 6
    P.bounding_box_free(bb)


def test_cli_flag_surface_and_no_cpu_path():
    """`sdf-viewer-gpu app ... demo ...` takes the reference's flag names (app/cli/mod.rs:10-22 + demo flags)."""
    import os
    import subprocess
    import torch
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdf-viewer_amd", "sdf-viewer-gpu")
    r = subprocess.run([exe, "app", "demo", "-t", "marble"], capture_output=True, text=True)
    assert r.returncode == 2 and "Invalid cube material" in r.stderr
    r = subprocess.run([exe, "app", "--bogus", "demo"], capture_output=True, text=True)
    assert r.returncode == 2 and "wasn't expected" in r.stderr
    r = subprocess.run([exe, "server"], capture_output=True, text=True)
    assert r.returncode == 2
    # `url` (CliSDFProvider::Url, app/cli/mod.rs:41-46): a native provider library stands where the wasm file stands
    r = subprocess.run([exe, "app", "url", "/nonexistent/libsdf.so"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot load SDF provider" in r.stderr
    r = subprocess.run([exe, "app", "url", "https://example.org/sdf.wasm"], capture_output=True, text=True)
    assert r.returncode == 2 and "only local provider libraries" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "app", "--max-voxels-side", "8", "demo"], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr


# ---- Mesh serialisation (meshers/mesh.rs:37-129): host-only, no GPU ----
def test_f32_display_like_rust(host):
    """ply-rs prints floats with `{}`: shortest digits that round-trip, never an exponent, "1" for 1.0."""
    cases = [(1.0, "1"), (0.5, "0.5"), (-2.0, "-2"), (0.1, "0.1"), (1e-7, "0.0000001"), (16777216.0, "16777216"),
             (0.95, "0.95"), (3.4028235e38, "340282350000000000000000000000000000000"), (float("inf"), "inf"),
             (float("-inf"), "-inf"), (0.0, "0"), (-0.0, "-0"), (123.456, "123.456"), (1.5e10, "15000000000")]
    for v, want in cases:
        assert host.format_f32(v) == want, v
    assert host.format_f32(float("nan")) == "NaN"
    rng = np.random.default_rng(3)
    for v in rng.normal(size=300).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 7, size=300).astype(np.float32):
        txt = host.format_f32(float(v))
        assert np.float32(txt) == v and "e" not in txt.lower()
        assert txt == np.format_float_positional(v, unique=True, trim="-")  # numpy's shortest repr agrees


def test_ply_colour_quantisation_matches_oracle(host, oracle):
    for c in [0.0, 1.0, 0.5, 0.999, 0.99, 2.0, -1.0, 1e-3, 0.2196, float("nan")]:
        assert host.H.sdfvh_ply_color_u8(c) == oracle.L.or_ply_color_u8(c)


def test_serialize_ply_structure(host):
    """Header element/property list and order are the reference's (mesh.rs:47-96); one line per vertex and face."""
    v = np.zeros((3, 12), np.float32)
    v[:, 0:3] = [(0, 0, 0), (1, 0, 0), (0, 1.5, 0)]
    v[:, 3:6] = (0, 0, 1)
    v[:, 6:9] = [(1, 0, 0), (0.5, 0.5, 0.5), (0.2, 0.4, 0.6)]
    v[:, 9:12] = (0.25, 0.5, 1.0)
    ply = host.Mesh.from_arrays(v, [0, 1, 2]).serialize_ply("sdf-viewer x.y (test)")
    lines = ply.split("\n")
    assert lines[:4] == ["ply", "format ascii 1.0", "comment Created with sdf-viewer x.y (test)", "element vertex 3"]
    props = [ln for ln in lines if ln.startswith("property")]
    assert props == [f"property float {n}" for n in ("x", "y", "z", "nx", "ny", "nz")] + \
        [f"property uchar {n}" for n in ("red", "green", "blue")] + \
        [f"property float {n}" for n in ("metallic", "roughness", "occlusion")] + \
        ["property list uchar int vertex_index"]
    assert "element face 1" in lines and ply.endswith("\n")
    body = lines[lines.index("end_header") + 1:]
    assert body[0] == "0 0 0 0 0 1 255 0 0 0.25 0.5 1"
    assert body[1] == "1 0 0 0 0 1 127 127 127 0.25 0.5 1"
    assert body[2] == "0 1.5 0 0 0 1 51 102 153 0.25 0.5 1"
    assert body[3] == "3 0 1 2" and body[4] == ""
    # an empty mesh is still a valid file; a trailing partial triangle is dropped (chunks_exact(3), mesh.rs:116)
    empty = host.Mesh.from_arrays(np.zeros((0, 12), np.float32), []).serialize_ply()
    assert "element vertex 0" in empty and "element face 0" in empty
    assert "element face 1" in host.Mesh.from_arrays(v, [0, 1, 2, 1]).serialize_ply()


# ---- host sampling: runs of a pass, and an SDF behind the per-point ABI (the ingest path's host half; no GPU) ----
def test_advance_equals_stepping(host):
    """LoadingManager::advance(n) = n calls of next(): same cursor, same pass changes, same totals; pass_point(k) is the
    k-th point next() yields in the current pass (what SDFViewer::update hands its worker threads)."""
    for limits, passes in (((8, 11, 17), 3), ((5, 1, 3), 2), ((4, 4, 4), 1), ((9, 7, 5), 4)):
        a, b = host.LoadingManager(limits, passes), host.LoadingManager(limits, passes)
        chunk = 1
        while b.step_size():
            n = min(chunk, b.pass_remaining())
            want = [a.next() for _ in range(n)]
            got = [b.pass_point(b.cursor() + k) for k in range(n)]
            assert got == want, (limits, passes, chunk)
            b.advance(n)
            assert (a.len(), a.total_iterations(), a.step_size(), a.passes_left()) == \
                   (b.len(), b.total_iterations(), b.step_size(), b.passes_left())
            chunk = chunk * 3 % 101 + 1
        assert a.next() is None and b.pass_remaining() == 0


def test_provider_sdf_consumes_the_per_point_abi(host, gyroid_provider):
    """ProviderSDF (host/provider_sdf.cpp) = WasmerSDF's role (wasm/native.rs:163-521) for a native library: required exports
    called and freed, optional exports used when present, the trait's defaults when absent."""
    import ctypes as C
    sdf = host.SDF.provider(gyroid_provider)
    raw = C.CDLL(gyroid_provider)
    assert sdf.id() == 0                               # the root SDF (native.rs:78)
    assert np.array_equal(sdf.bounding_box(), np.float32([-1, -0.5, -0.75, 1, 0.5, 0.75]))
    assert sdf.name() == "Object"                      # no `name` export: name_default_impl (defaults.rs:19-21)
    assert sdf.children() == []                        # no `children` export
    assert np.array_equal(sdf.normal([0.1, 0.2, 0.3], 0.001), np.zeros(3, np.float32))  # no `normal` export: zero (native.rs:498)
    assert sdf.sample_concurrency() == 64              # the optional extension export
    assert sdf.device_params.__self__ is sdf           # (binding sanity)
    want = np.zeros(7, np.float32)
    for p in ([0.3, -0.2, 0.5], [-1, -0.5, -0.75], [0.9, 0.4, -0.7]):
        q = np.float32(p)
        raw.gyroid_sample_raw(None, q.ctypes.data_as(C.c_void_p), 0, want.ctypes.data_as(C.c_void_p))
        assert np.array_equal(sdf.sample(p).view(np.uint32), want.view(np.uint32))
        raw.gyroid_sample_raw(None, q.ctypes.data_as(C.c_void_p), 1, want.ctypes.data_as(C.c_void_p))
        assert np.array_equal(sdf.sample(p, True).view(np.uint32), want.view(np.uint32))
    assert np.isnan(sdf.sample([-1, -0.5, -0.75])[0])
    prm = sdf.parameters()
    assert len(prm) == 1 and prm[0][:3] == ["0", "thickness", "2"] and prm[0][4] == "shell thickness"
    assert sdf.changed() is None
    assert sdf.set_parameter(7, 0.5) == "unknown parameter"
    assert sdf.set_parameter(0, 0.25) is None
    assert sdf.parameters()[0][3] == "Float(0.25)"
    box = sdf.changed()
    assert box is not None and np.array_equal(box, np.float32([-1, -0.3, -0.75, 0.1, 0.5, 0.75]))
    assert sdf.changed() is None                       # reported once
    assert sdf.set_parameter(0, 0.15) is None and sdf.changed() is not None  # (back to the default for later tests)


def test_provider_sdf_maps_failed_calls_to_the_defaults(host, failing_provider, capfd):
    """A provider call that fails is logged and answered with the trait's default, never a crash (wasm/native.rs:164-521:
    bounding box [0, 1]^3 :172, sample = SDFSample::new(1.0, 0) :203, no children :226, name_default_impl :266, no parameters
    :290, set_parameter's default error :413, not changed :468, zero normal :506); malformed blocks are read as far as they
    are whole records: unknown enum tags drop the parameter (:333,:368), a child list naming the SDF itself skips that entry
    (:241-244)."""
    import ctypes as C
    raw = C.CDLL(failing_provider)
    sdf = host.SDF.provider(failing_provider)
    default_error = "no parameters implemented by default, overwrite this method"
    raw.set_fail_mode(0)                                          # every call returns NULL
    assert np.array_equal(sdf.bounding_box(), np.float32([0, 0, 0, 1, 1, 1]))
    assert np.array_equal(sdf.sample([0.1, 0.2, 0.3]), np.float32([1, 0, 0, 0, 0, 0, 0]))
    got = sdf.sample_batch(np.zeros((5, 3), np.float32))           # the default loop over sample()
    assert got.shape == (5, 7) and np.array_equal(got, np.tile(np.float32([1, 0, 0, 0, 0, 0, 0]), (5, 1)))
    assert sdf.children() == [] and sdf.name() == "Object" and sdf.parameters() == []
    assert sdf.set_parameter(0, 1.5) == default_error and sdf.changed() is None
    assert np.array_equal(sdf.normal([0.1, 0.2, 0.3], 0.001), np.zeros(3, np.float32))
    assert sdf.sample_concurrency() == 1                           # an export that answers 0
    assert "Failed to get bounding box" in capfd.readouterr().err
    raw.set_fail_mode(1)                                          # well-formed blocks, values a consumer must refuse
    kids = sdf.children()
    assert [k.id() for k in kids] == [5]                           # itself skipped, the cut-off third id never read
    assert sdf.name() == ""
    prm = sdf.parameters()
    assert len(prm) == 1 and prm[0][:4] == ["12", "ok", "1", "Int(2)"]
    assert sdf.set_parameter(12, 1) == "Unknown SDF set parameter result kind enum type"
    assert sdf.changed() is None
    err = capfd.readouterr().err
    assert "include itself" in err and "Unknown SDF param kind enum type 9" in err and "Unknown SDF param value enum type 77" in err
    assert "Unknown SDF changed result kind enum type 3" in err
    raw.set_fail_mode(2)                                          # lengths without data
    assert sdf.children() == [] and sdf.name() == "" and sdf.parameters() == []
    assert sdf.set_parameter(12, 1) == ""                          # an Err without a message is still an Err


def test_provider_sdf_load_errors(host, tmp_path):
    import subprocess
    with pytest.raises(OSError, match="cannot load SDF provider"):
        host.SDF.provider(tmp_path / "missing.so")
    src = tmp_path / "half.c"
    src.write_text("void *bounding_box(unsigned id) { (void)id; return 0; }\n")
    lib = tmp_path / "libhalf.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(lib)])
    with pytest.raises(OSError, match="does not export sample"):  # native.rs:62: `sample` is required
        host.SDF.provider(lib)


def test_demo_provider_is_an_ordinary_provider(host):
    """libsdfdemo_provider.so loads through ProviderSDF like any other library: hierarchy, names, parameters (metadata only
    here: its sample() runs on the GPU)."""
    sdf = host.SDF.provider(host.PROVIDER_PATH)
    assert sdf.name() == "Demo" and sdf.sample_concurrency() == 1   # thread-local registry
    assert sorted((c.id(), c.name()) for c in sdf.children()) == [(1, "DemoCube"), (2, "DemoSphere")]
    assert np.array_equal(sdf.bounding_box(), np.float32([-1, -1, -1, 1, 1, 1]))
    assert [p[1] for p in sdf.parameters()] == [p[1] for p in host.SDF.demo().parameters()]


def test_worker_pool_sessions_never_lose_a_run(tmp_path):
    """The ingest path's WorkerPool (host/worker_pool.hpp): sessions of varying size, threads created mid-life, runs back to
    back -- every worker of every run executes exactly once, also with more workers than CPUs (bounded spin, then yield)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "worker_pool_stress"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Werror", "-I", os.path.join(root, "sdf-viewer_amd", "host"),
                           os.path.join(root, "tests", "c", "worker_pool_stress.cpp"), "-o", str(exe)])
    for workers in (5, 3 * (os.cpu_count() or 4)):
        r = subprocess.run([str(exe), str(workers)], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0 and r.stdout.startswith("ok"), (workers, r.stdout, r.stderr)


def test_batched_sampling_equals_the_per_point_loop(host, gyroid_provider, gyroid_provider_batch):
    """SDFSurface::sample_batch (the trait's "Batched sampling" TODO, src/sdf/mod.rs:39): the default is the loop of sample()
    calls; a provider that exports `sample_batch` answers through it -- same samples, bit for bit, either way."""
    import ctypes as C
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, size=(777, 3)).astype(np.float32)
    pts[0] = [-1, -0.5, -0.75]                                    # the fixture's NaN voxel
    plain, batch = host.SDF.provider(gyroid_provider), host.SDF.provider(gyroid_provider_batch)
    assert not hasattr(C.CDLL(gyroid_provider), "sample_batch") and hasattr(C.CDLL(gyroid_provider_batch), "sample_batch")
    want = np.stack([plain.sample(p) for p in pts])
    for sdf in (plain, batch):
        for distance_only in (False, True):
            ref = want if not distance_only else np.stack([plain.sample(p, True) for p in pts])
            got = sdf.sample_batch(pts, distance_only)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert batch.sample_batch(np.zeros((0, 3), np.float32)).shape == (0, 7)
