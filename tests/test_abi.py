"""CPU tests of the drop-in boundary: libsdfgrid.so loads, exports every symbol include/sdfgrid.h declares,
struct layouts match the reference's repr(C) types, host-side helpers agree with the oracle, and the compute
entry points refuse to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    """(exported, header_only): function names include/sdfgrid.h declares, and the static inline wrappers it defines."""
    hdr = open(os.path.join(ROOT, "include", "sdfgrid.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prose mentions functions too
    inline = set(re.findall(r"SDFV_INLINE\s+int\s+(sdfv_[a-z0-9_]+)\s*\(", hdr))
    declared = set(re.findall(r"^[a-z][a-z0-9_ \*]*?\b(sdfv_[a-z0-9_]+)\s*\(", hdr, flags=re.M)) - inline
    return declared, inline


def test_exports_every_declared_symbol(pkg):
    declared, inline = header_functions()
    assert len(declared) >= 17
    assert declared == set(pkg._capi.PROTOTYPES), (sorted(declared ^ set(pkg._capi.PROTOTYPES)), "binding and header disagree")
    raw = C.CDLL(pkg._capi.LIB_PATH)
    for name in declared:
        getattr(raw, name)
    # ABI v4: ONE march entry point and ONE pass entry point are exported; the positional forms of v3
    # are header-only wrappers and not symbols of the library
    assert inline == {"sdfv_raymarch", "sdfv_raymarch_accel", "sdfv_raymarch_depth", "sdfv_raymarch_pairs", "sdfv_raymarch_volumes",
                      "sdfv_raymarch_bands", "sdfv_fill_grid_pass", "sdfv_fill_grid_pass_dist"}
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", pkg._capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (sdfv_[a-z0-9_]+)", syms))
    assert exported == declared, sorted(exported ^ declared)
    assert sorted(n for n in exported if n.startswith("sdfv_raymarch")) == ["sdfv_raymarch_ex", "sdfv_raymarch_host",
                                                                            "sdfv_raymarch_slab", "sdfv_raymarch_slab_round"]
    assert sorted(n for n in exported if "fill_grid_pass" in n) == ["sdfv_fill_grid_pass_ex"]


def test_struct_layouts(pkg):
    assert C.sizeof(pkg.Sample) == 28          # src/sdf/mod.rs:104-118, 7 LE f32 (wasm/native.rs:204-216)
    assert C.sizeof(pkg.DemoParams) == 24
    assert C.sizeof(pkg.Grid) == 44
    assert C.sizeof(pkg.Camera) == 120
    assert C.sizeof(pkg.Light) == 32
    assert C.sizeof(pkg.RenderParams) == 80 + 4 + 4 * 32
    assert C.sizeof(pkg.MarchAux) == 72
    assert pkg.lib.sdfv_abi_version() == 5
    assert C.sizeof(pkg._capi.MarchDesc) == 128  # the layout a binder mirrors (120 bytes before rgba8: size-prefixed, both work)


def test_options_are_explicit_and_the_library_reads_no_environment(pkg):
    """Kernel-variant knobs travel through sdfv_set_option (thread-local), never through getenv."""
    K = pkg._capi
    for opt in (K.OPT_FILL_NONTEMPORAL, K.OPT_FILL_FORM, K.OPT_RAYMARCH_DISABLE, K.OPT_RAYMARCH_KEEP_NORMAL,
                K.OPT_SLAB_STEP_FORM):
        assert pkg.get_option(opt) == 0                     # product defaults
    with pkg.options({K.OPT_FILL_FORM: 2, K.OPT_RAYMARCH_DISABLE: K.RM_NO_SYMMETRIC | K.RM_NO_POW2_SIZE}):
        assert pkg.get_option(K.OPT_FILL_FORM) == 2 and pkg.get_option(K.OPT_RAYMARCH_DISABLE) == 12
        import threading
        seen = []
        t = threading.Thread(target=lambda: seen.append(pkg.get_option(K.OPT_FILL_FORM)))
        t.start()
        t.join()
        assert seen == [0]                                  # another thread keeps the defaults
    assert pkg.get_option(K.OPT_FILL_FORM) == 0
    assert pkg.get_option(K.OPT_RAYMARCH_TILE_GROUP) == 0 and pkg.get_option(K.OPT_RAYMARCH_BOX_FIRST) == 1
    assert pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_BOX_FIRST, 2) == -1 and pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_TILE_GROUP, 6) == -1
    assert pkg.get_option(K.OPT_RAYMARCH_WAVES_PER_SIMD) == 0
    assert pkg.get_option(K.OPT_RAYMARCH_BATCH_STREAMS) == 1 and pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_BATCH_STREAMS, 2) == -1
    assert pkg.get_option(K.OPT_RAYMARCH_CAMERA_STAGING) == 1 and pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_CAMERA_STAGING, 2) == -1
    assert pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, 1) == -1 and pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_WAVES_PER_SIMD, 8) == -1
    assert pkg.lib.sdfv_set_option(K.OPT_FILL_FORM, 5) == -1  # (3, 4: the two pinned forms of the interleaved-volume fill, round 5)
    assert pkg.get_option(K.OPT_PASS_FORM) == 0 and pkg.lib.sdfv_set_option(K.OPT_PASS_FORM, 2) == -1
    assert pkg.lib.sdfv_set_option(K.OPT_RAYMARCH_DISABLE, 64) == -1
    assert pkg.lib.sdfv_set_option(77, 0) == -1 and b"unknown option" in pkg.lib.sdfv_last_error()
    # the wave-timing stamps exist only in the tuning build
    assert pkg.lib.sdfv_set_option(K.OPT_TUNING_WAVE_TIMING, 4096) == -1
    assert b"tuning build" in pkg.lib.sdfv_last_error()
    for src in os.listdir(os.path.join(ROOT, "sdf-viewer_amd", "csrc")):
        if src.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "sdf-viewer_amd", "csrc", src)).read(), src


def test_light_list_rejects_what_the_reference_does_not_pin(pkg):
    """Only ambient lights are rendered; a directional entry is an argument error (three-d 0.18.2's shader text is not
    in the reference tree), reported before any device work -- so this runs without a GPU."""
    g = pkg.make_grid((4, 4, 4))
    rp = pkg.default_render_params(g)
    assert rp.n_lights == 0
    cam = pkg.camera_look_at()
    march = lambda tex0=16, **kw: pkg._capi.raymarch_rc(rp, tex0, 16, cam, 1, 8, 8, 0, 8, 16, **kw)  # noqa: E731
    rp.n_lights = 1
    rp.lights[0].kind = pkg._capi.LIGHT_DIRECTIONAL
    assert march() == -1
    assert b"three-d 0.18.2 shader source not available" in pkg.lib.sdfv_last_error()
    rp.lights[0].kind = 9
    assert march() == -1 and b"unknown kind" in pkg.lib.sdfv_last_error()
    rp.n_lights = 5
    assert march() == -1
    # misaligned texel buffers are argument errors too (they are read and written as 16-byte texels)
    rp.n_lights = 0
    assert march(tex0=20) == -1 and b"16-byte aligned" in pkg.lib.sdfv_last_error()
    # the descriptor is size-prefixed: a size below its first version is refused, reserved words must be 0
    assert march(size=24) == -1 and b"sdfv_march_desc.size" in pkg.lib.sdfv_last_error()
    assert pkg.lib.sdfv_raymarch_ex(None, None) == -1
    # a band height without a band step is refused BY THE LIBRARY (the header's sdfv_raymarch_bands wrapper hands
    # band_step == 0 over in that shape, so sdfv_last_error names it); a newer caller's longer descriptor is read as far as this
    # library knows it -- provided the rest is zero
    assert march(band_step=0, band_height=16) == -1 and b"band_step 0" in pkg.lib.sdfv_last_error()

    class Longer(C.Structure):
        _fields_ = [("d", pkg._capi.MarchDesc), ("future", C.c_uint64 * 2)]
    lg = Longer()
    lg.d.size = C.sizeof(lg)
    lg.d.rp = C.pointer(rp)
    lg.d.tex0, lg.d.tex1, lg.d.rgba = 16, 16, 16
    lg.d.cameras = C.pointer(cam)
    lg.d.n_cameras, lg.d.width, lg.d.height, lg.d.y0, lg.d.y1 = 1, 8, 8, 0, 8
    lg.future[1] = 7
    assert pkg.lib.sdfv_raymarch_ex(C.cast(C.pointer(lg), C.POINTER(pkg._capi.MarchDesc)), None) == -1
    assert b"beyond them is not 0" in pkg.lib.sdfv_last_error()
    lg.future[1] = 0
    rc = pkg.lib.sdfv_raymarch_ex(C.cast(C.pointer(lg), C.POINTER(pkg._capi.MarchDesc)), None)
    assert b"beyond them" not in pkg.lib.sdfv_last_error() and rc != 0  # (accepted as a descriptor; no device here / bogus pointers)
    p = pkg.default_params()
    assert pkg.lib.sdfv_grid_init(C.byref(g), C.c_void_p(24), C.c_void_p(16), None) == -1
    assert pkg.lib.sdfv_fill_grid_pass_ex(C.byref(p), 0, C.byref(g), 1, None, C.c_void_p(16), C.c_void_p(8), None, 0, None) == -1
    assert pkg.lib.sdfv_commit_distance(C.byref(g), C.c_void_p(8), C.c_void_p(16), None) == -1


def test_defaults_match_reference_flags(pkg):
    p = pkg.default_params()
    assert (np.float32(p.cube_half_side), p.cube_material) == (np.float32(0.95), pkg.MATERIAL_BRICK)   # cube.rs:15-18
    assert (np.float32(p.sphere_radius), p.sphere_material) == (np.float32(1.05), pkg.MATERIAL_NORMAL)  # sphere.rs:11-14
    assert np.float32(p.max_distance_custom_material) == np.float32(0.05) and p.disable_sphere == 0   # demo/mod.rs:26-29
    assert np.float32(pkg.AIR_DIST).view(np.uint32) == 0x3DCF53C6


def test_grid_from_bb(pkg, oracle):
    for bb_min, bb_max, n in [((-1, -1, -1), (1, 1, 1), 64), ((0, 0, 0), (2, 1, 0.5), 64), ((0, 0, 0), (1, 3, 2), 100),
                              ((-1, 0, 0), (1, 2, 0.3), 17)]:
        g = pkg.grid_from_bb(bb_min, bb_max, n)
        want = (C.c_uint32 * 3)()
        oracle.L.or_grid_dims_from_bb(oracle.f3(bb_min), oracle.f3(bb_max), n, want)
        assert list(g.dims) == list(want)
        assert (g.z_begin, g.z_end) == (0, g.dims[2])


def test_camera_matches_oracle(pkg, oracle):
    for eye, aspect in [((2.5, 3.0, 5.0), 1.0), ((2.5, 3.0, 5.0), 1920 / 1080), ((-0.3, 0.2, 0.1), 1.5)]:
        a = pkg.camera_look_at(eye=eye, aspect=aspect)
        b = oracle.camera_look_at(eye=eye, aspect=aspect)
        assert bytes(a) == bytes(b)


def test_render_params_default(pkg, oracle):
    g = pkg.make_grid((64, 32, 16), (-1, 0, 0), (1, 1, 0.5))
    a = pkg.default_render_params(g)
    b = oracle.default_render_params((64, 32, 16), (-1, 0, 0), (1, 1, 0.5))
    assert bytes(a) == bytes(b)


def test_srgb_table_in_library_source_matches_oracle(oracle):
    txt = open(os.path.join(ROOT, "sdf-viewer_amd", "csrc", "srgb_lut.inc")).read().split("*/")[1]
    vals = np.array([float.fromhex(x) for x in re.findall(r"-?0x[0-9a-fp.+-]+", txt)], np.float32)
    want = np.array([oracle.L.or_srgb_u8_to_linear(i) for i in range(256)], np.float32)
    np.testing.assert_array_equal(vals, want)


def test_invalid_arguments_are_reported_not_fatal(pkg):
    lib = pkg.lib
    p = pkg.default_params()
    g = pkg.make_grid((4, 4, 4))
    assert lib.sdfv_fill_grid(None, 0, C.byref(g), None, None, None) == -1
    assert lib.sdfv_fill_grid(C.byref(p), 7, C.byref(g), C.c_void_p(16), C.c_void_p(16), None) == -2  # unknown id
    assert b"Failed to find SDF with ID 7" in lib.sdfv_last_error()                                   # ffi.rs:47
    bad = pkg.make_grid((4, 4, 4), z_begin=3, z_end=9)
    assert lib.sdfv_fill_grid(C.byref(p), 0, C.byref(bad), C.c_void_p(16), C.c_void_p(16), None) == -1
    assert lib.sdfv_fill_grid_pass_ex(C.byref(p), 0, C.byref(g), 3, None, C.c_void_p(16), C.c_void_p(16), None, 0, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback(pkg):
    """Without a HIP device the product fails loudly instead of computing anything on the CPU."""
    p = pkg.default_params()
    g = pkg.make_grid((4, 4, 4))
    t0 = np.zeros((4, 4, 4, 4), np.float32)
    t1 = np.zeros_like(t0)
    rc = pkg.lib.sdfv_fill_grid_host(C.byref(p), 0, C.byref(g), t0.ctypes.data, t1.ctypes.data)
    assert rc == -4 and b"no HIP device" in pkg.lib.sdfv_last_error()
    assert (t0 == 0).all()
    with pytest.raises(TypeError):
        pkg.fill_grid(p, g, torch.zeros(4, 4, 4, 4), torch.zeros(4, 4, 4, 4))


def test_slab_communicator_argument_checks_and_no_device(pkg):
    """sdfv_slab_comm_create validates before it touches RCCL, and without a GPU reports "no device" (no hang)."""
    lib = pkg.lib
    ident = (C.c_ubyte * 128)()
    out = C.c_void_p(123)
    assert lib.sdfv_slab_comm_create(ident, 2, 2, 0, C.byref(out)) == -1 and out.value is None
    assert lib.sdfv_slab_comm_create(ident, 0, 1, 6, C.byref(out)) == -1
    assert lib.sdfv_slab_comm_create(None, 0, 1, 0, C.byref(out)) == -1
    assert lib.sdfv_slab_comm_destroy(None) == 0
    if not torch.cuda.is_available():
        assert lib.sdfv_slab_comm_create(ident, 0, 1, 0, C.byref(out)) == -4
        assert b"no HIP device" in lib.sdfv_last_error()


def test_band_rows_is_the_partition_the_harness_uses(pkg):
    """sdfv_band_rows (pure host arithmetic: no device needed) == the rows parallel.band_rows enumerates, for every band set
    of worlds 1..9 over heights around the 16-row tile; together the sets partition the image."""
    import importlib
    par = importlib.import_module("sdf-viewer_amd.parallel")
    for h in (1, 7, 15, 16, 17, 31, 32, 33, 100, 1080, 2160):
        for world in range(1, 10):
            rows = [par.band_rows(h, r, world) for r in range(world)]
            assert sorted(sum(rows, [])) == list(range(h))
            for r in range(world):
                assert pkg.lib.sdfv_band_rows(h, r, world) == len(rows[r]), (h, r, world)
        assert pkg.lib.sdfv_band_rows(h, 0, 0) == 0 and pkg.lib.sdfv_band_rows(h, (h + 15) // 16, 1) == 0
    assert pkg.lib.sdfv_band_rows(1080, 3, 8) == 8 * 16 + 8  # bands 3, 11, ..., 67: the last one holds rows 1072..1079


def test_power_of_two_modulus_identity():
    """The kernels replace `x % 0.5` / `x % 0.25` (cube.rs:192) by a - trunc(a * (1/m)) * m and `v / 0.25`,
    `floor(r) / 4` by multiplications.  Every step is exact for a power-of-two modulus; checked here against
    fmod / true division on float32 bit patterns from every binade (the arithmetic is the same on the device)."""
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 0x7F800000, size=4_000_000, dtype=np.uint32)           # all finite non-negative floats
    a = np.concatenate([bits.view(np.float32),
                        np.array([0.0, 0.25, 0.5, 0.75, 1.0, 2.0 ** 23, 2.0 ** 24, 3.4e38, 1e-45, 1.17549435e-38,
                                  0.49999997, 0.24999999, 0.50000006], np.float32)])
    for m, inv in ((np.float32(0.5), np.float32(2.0)), (np.float32(0.25), np.float32(4.0))):
        with np.errstate(over="ignore", invalid="ignore"):
            got = a - np.trunc(a * inv) * m
            want = np.fmod(a, m)
            ok = np.isfinite(a * inv)                  # a * inv overflows only above 1.7e38 (never a texture coordinate)
        np.testing.assert_array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    s = np.concatenate([a, -a])
    with np.errstate(over="ignore"):
        times4, div025 = s * np.float32(4.0), s / np.float32(0.25)
    np.testing.assert_array_equal(times4.view(np.uint32), div025.view(np.uint32))   # including the overflow to inf
    r = np.floor(s[np.abs(s) < 1e30])
    np.testing.assert_array_equal((r * np.float32(0.25)).view(np.uint32), (r / np.float32(4.0)).view(np.uint32))
    # (p - min) / 2^k == (p - min) * 2^-k, and for power-of-two texture sizes ((p - min) * 2^-k) * N == (p - min) * (2^-k * N)
    d = rng.uniform(-3.0, 3.0, size=1_000_000).astype(np.float32)
    for size, n in ((np.float32(2.0), np.float32(256.0)), (np.float32(0.5), np.float32(64.0)), (np.float32(8.0), np.float32(1024.0))):
        inv = np.float32(1.0) / size
        np.testing.assert_array_equal((d / size).view(np.uint32), (d * inv).view(np.uint32))
        np.testing.assert_array_equal(((d * inv) * n).view(np.uint32), (d * (inv * n)).view(np.uint32))
    # symmetric box: max(-m - p, p - m) == |p| - m
    mx = np.float32(1.0)
    np.testing.assert_array_equal(np.maximum(-mx - d, d - mx).view(np.uint32), (np.abs(d) - mx).view(np.uint32))


LLVM_TOOLS = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    """Every gfx950 code object inside libsdfgrid.so, unbundled: [(path of the .co, text of its metadata notes)]."""
    import subprocess
    tmp_path = tmp_path_factory.mktemp("code_objects")
    lib = os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid.so")
    if not all(os.path.exists(os.path.join(LLVM_TOOLS, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump")):
        pytest.skip("LLVM binary tools not installed")
    fat = tmp_path / "fat.bin"
    subprocess.run([f"{LLVM_TOOLS}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib, str(tmp_path / "unused.so")], check=True)
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, blob)]
    assert starts, "no offload bundle in the library"
    out = []
    for k, at in enumerate(starts):
        piece = tmp_path / f"bundle{k}.bin"
        piece.write_bytes(blob[at:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = tmp_path / f"bundle{k}.co"
        subprocess.run([f"{LLVM_TOOLS}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={piece}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM_TOOLS}/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        out.append((co, notes))
    return out


def test_no_kernel_needs_more_than_4_kb_of_arguments(code_objects):
    """HIP documents 4 KB of kernel arguments; round 3's march carried 64 cameras in an 8 KB block.  Every gfx950
    code object inside libsdfgrid.so is unbundled and its kernels' kernarg sizes read from the metadata notes."""
    sizes = []
    for _, notes in code_objects:
        sizes += [int(v) for v in re.findall(r"\.kernarg_segment_size:\s+(\d+)", notes)]
    assert len(sizes) > 100 and max(sizes) <= 4096, (len(sizes), max(sizes))


def _kernel_table(code_objects):
    """mangled name -> {vgpr, vgpr_spill, sgpr_spill, lds, scratch, co} from the code objects' metadata."""
    table = {}
    for co, notes in code_objects:
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            def field(key, blk=blk):
                return int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
            name = re.search(r"\n\s+\.name:\s+(\S+)\n\s+\.private_segment_fixed_size", blk).group(1)
            table[name] = dict(vgpr=field("vgpr_count"), vgpr_spill=field("vgpr_spill_count"), sgpr_spill=field("sgpr_spill_count"),
                               lds=field("group_segment_fixed_size"), scratch=field("private_segment_fixed_size"), co=co)
    return table


def _opcodes(co, symbol, cache={}):
    """The instruction stream of one kernel: ["opcode", "opcode nt", ...] (llvm-objdump -d of its code object)."""
    import subprocess
    if co not in cache:
        text = subprocess.run([f"{LLVM_TOOLS}/llvm-objdump", "-d", str(co)], check=True, capture_output=True, text=True).stdout
        parts = re.split(r"\n[0-9a-f]{16} <([^>]+)>:\n", text)
        cache[co] = dict(zip(parts[1::2], parts[2::2]))
    ops = []
    for line in cache[co][symbol].split("\n"):
        code = line.split("//")[0].split()
        if code:
            ops.append(code[0] + (" nt" if "nt" in code[1:] else ""))
    return ops


def test_hot_kernels_keep_their_performance_contract(code_objects):
    """what DESIGN.md claims about the kernels the bench times is read back from the BUILT library, so that a
    live register added to the march loop, a spill, or a toolchain that stops emitting the streaming stores fails here (CPU, no
    GPU) instead of showing up as a slower bench line.  The march loop stands in for material.frag:92-128, the fill for
    scene/sdf/mod.rs:173-215."""
    from collections import Counter
    table = _kernel_table(code_objects)
    assert len(table) > 250
    # nothing in the library spills vector registers or uses scratch; a few cold kernels spill SGPRs (to VGPR lanes)
    assert max(k["vgpr_spill"] for k in table.values()) == 0 and max(k["scratch"] for k in table.values()) == 0
    # ---- the march kernels the bench and SDFViewerMaterial::render launch (hand-written gfx950 loop, NORMAL off) ----
    for cams in (1, 2, 4):
        name = f"_ZN4sdfv12_GLOBAL__N_115raymarch_kernelILi{cams}ELb1ELi2ELb1ELb0ELb1ELb0EEEvNS_12RaymarchArgsE"
        k = table[name]
        assert k["vgpr"] <= 72, (cams, k)              # 7 waves per SIMD (512 / 72); 73 would drop to 6
        assert k["sgpr_spill"] == 0 and k["lds"] == 0  # (unused dynamic LDS is the launcher's occupancy cap)
        ops = Counter(_opcodes(k["co"], name))
        # the loop's shape: EXEC = marching lanes narrowed by v_cmpx, skipped blocks by s_cbranch_execz, packed fp32 lerps
        assert ops["v_cmpx_nlt_f32_e32"] >= 1 and ops["v_cmpx_ngt_f32_e32"] >= 1 and ops["s_cbranch_execz"] >= 8, cams
        assert ops["v_pk_mul_f32"] >= 60 and ops["v_pk_add_f32"] >= 40 and ops["v_mad_u32_u24"] >= 16, cams
        assert ops["v_pk_fma_f32"] == 0, "the reference's shader arithmetic is not contracted"
        assert not any(o.startswith(("scratch_", "buffer_")) for o in ops), cams
    # ---- the dense fills: store-only, 16-byte stores, the textures streamed (nt) in the fused form ----
    def fill(tx, fused, cfg="NS_11DefaultCfgTILb0EEE", last="Lb0"):
        return f"_ZN4sdfv12_GLOBAL__N_117fill_dense_kernelILi{tx}ELb{int(fused)}E{cfg}{last}EEEvNS_8FillArgsE"
    fused = table[fill(256, True)]
    assert fused["vgpr"] <= 24 and fused["sgpr_spill"] == 0 and fused["lds"] <= 1100, fused   # the 1 KiB sRGB table + a row's (y, z)
    ops = Counter(_opcodes(fused["co"], fill(256, True)))
    assert ops["global_store_dwordx4 nt"] == 2 and ops["global_store_dwordx4"] == 0, ops     # tex0 + tex1, past L2
    assert ops["global_load_dword"] == 1 and sum(v for o, v in ops.items() if o.startswith("global_load")) == 1, ops  # the LUT staging
    assert ops["global_store_dword"] >= 1                                                      # the distance volume stays cacheable
    plain = table[fill(256, False)]
    assert plain["vgpr"] <= 24 and plain["sgpr_spill"] == 0
    ops = Counter(_opcodes(plain["co"], fill(256, False)))
    assert ops["global_store_dwordx4"] + ops["global_store_dwordx4 nt"] == 2 and sum(v for o, v in ops.items() if o.startswith("global_load")) == 1
    for n, k in table.items():
        if "fill_dense" in n:
            assert k["vgpr"] <= 40 and k["sgpr_spill"] == 0, (n, k)   # >= 12 waves per SIMD for every dense form
    # ---- the ingest kernel: LUT + a workgroup's 256 records in LDS, one 16-byte and one 12-byte store per record ----
    for rnd in (0, 1):
        name = f"_ZN4sdfv12_GLOBAL__N_119pack_samples_kernelILb{rnd}EEEvNS_8PackArgsE"
        k = table[name]
        assert k["vgpr"] <= 24 and k["sgpr_spill"] == 0 and k["lds"] == 1024 + 256 * 28, k
        ops = Counter(_opcodes(k["co"], name))
        assert ops["global_store_dwordx4"] == 1 and ops["global_store_dwordx3"] == 1, ops


def test_headers_compile_as_plain_c_and_the_library_links(tmp_path):
    """include/*.h are C headers (the reference's FFI side is `extern "C"` Rust): a C99 program, -pedantic, no C++."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "abi_smoke"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
           os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", str(exe),
           "-L", os.path.join(root, "sdf-viewer_amd"), "-lsdfgrid", "-Wl,-rpath," + os.path.join(root, "sdf-viewer_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_integration_guide_binds_every_entry_point():
    """INTEGRATION.md shows the reference-side binding for every function include/sdfgrid.h declares."""
    import re
    header = open(os.path.join(ROOT, "include", "sdfgrid.h")).read()
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared, _inline = header_functions()  # what the library EXPORTS; the header-only wrappers are the binder's to restate
    missing = sorted(f for f in declared if f not in guide)
    assert not missing, missing


def test_rccl_library_option_is_process_wide_and_final_once_loaded():
    """SDFV_OPT_RCCL_LIBRARY: the one process-wide option -- the path of the RCCL-ABI library the communicator loads (how
    tests/c/mock_rccl.cpp stands in for RCCL); a path that cannot be loaded is a clean SDFV_ERR_COMM, and once RCCL has been
    loaded (or has failed to load) the option is refused.  In a process of its own: loading is once per process."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, importlib, sys
sys.path.insert(0, %r)
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
assert C.string_at(pkg.get_option(K.OPT_RCCL_LIBRARY)) == b""
buf = C.create_string_buffer(b"/nonexistent/librccl.so")
pkg.set_option(K.OPT_RCCL_LIBRARY, C.addressof(buf))
buf.value = b"overwritten"                                   # (the library keeps its own copy)
assert C.string_at(pkg.get_option(K.OPT_RCCL_LIBRARY)) == b"/nonexistent/librccl.so"
ident = (C.c_ubyte * 128)()
assert pkg.lib.sdfv_slab_comm_unique_id(ident) == -5 and b"SDFV_OPT_RCCL_LIBRARY" in pkg.lib.sdfv_last_error()
assert pkg.lib.sdfv_set_option(K.OPT_RCCL_LIBRARY, 0) == -1 and b"already been loaded" in pkg.lib.sdfv_last_error()
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]


def test_mock_rccl_and_its_driver_compile(tmp_path):
    """tests/c/mock_rccl.cpp and tests/c/multirank_mock.cpp (the multi-rank paths between different ranks on one GPU) build with
    g++ against the HIP runtime headers and include/sdfgrid.h, and the mock exports the eleven entry points csrc/slab_comm.hip
    binds.  (They RUN in tests/test_gpu_multirank.py.)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    mock = tmp_path / "librccl.so.1"
    r = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Werror", *hip,
                        os.path.join(root, "tests", "c", "mock_rccl.cpp"), "-o", str(mock), "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-D", "--defined-only", str(mock)], capture_output=True, text=True).stdout
    bound = set(re.findall(r'bind\(r\.handle, "(nccl\w+)"', open(os.path.join(root, "sdf-viewer_amd", "csrc", "slab_comm.hip")).read()))
    assert len(bound) == 11 and all(f" T {name}\n" in nm for name in bound), (sorted(bound), nm)
    r = subprocess.run(["g++", "-O0", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", *hip, "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "c", "multirank_mock.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
