"""sdf-viewer_amd -- Python harness over libsdfgrid.so (the MI355X-native voxelise-and-raymarch path).

The product is the C-ABI shared library (include/sdfgrid.h, sdf-viewer_amd/csrc/); this package is the
thin test/bench harness above it.  PyTorch is used only as plumbing: device memory (tensors whose
data_ptr() is handed to the C ABI), the current HIP stream, and torch.distributed for the multi-GPU
halo exchange.  Import fails loudly if the library has not been built -- there is no CPU path.

The package directory name contains a hyphen; import it with
    importlib.import_module("sdf-viewer_amd")
"""
import ctypes as C
import os

# The multi-GPU fill step needs its streams on separate HIP hardware queues.  The HIP runtime reads
# GPU_MAX_HW_QUEUES once, when it initialises (the first HIP call of the process): this harness sets a default before
# importing torch and says so when that is already too late.  A C / Rust host sets it in its own environment
# (include/sdfgrid.h); the library itself never touches the environment.
import sys
import warnings

_hip_already_up = "torch" in sys.modules and getattr(sys.modules["torch"], "cuda", None) is not None \
    and sys.modules["torch"].cuda.is_initialized()
if "GPU_MAX_HW_QUEUES" not in os.environ:
    if _hip_already_up:
        warnings.warn("sdf-viewer_amd imported after the HIP runtime initialised: GPU_MAX_HW_QUEUES keeps its default "
                      "(4); the multi-GPU fill step may serialise its streams (DESIGN.md 6)", RuntimeWarning)
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"

import torch  # noqa: E402

from . import _capi
from ._capi import (Camera, DemoParams, Grid, Light, MarchAux, RenderParams, Sample, SdfvError, check, f3, lib,  # noqa: F401
                    MATERIAL_BRICK, MATERIAL_NORMAL, SDF_CUBE, SDF_DEMO, SDF_SPHERE)

AIR_DIST = lib.sdfv_air_dist()  # scene/sdf/mod.rs:42
AUX_FLOATS = C.sizeof(MarchAux) // 4


def default_params(**overrides):
    """SDFDemo::default(): the clap defaults (cube.rs:15-18, sphere.rs:11-14, demo/mod.rs:26-29)."""
    p = DemoParams()
    lib.sdfv_demo_params_default(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def make_grid(dims, bb_min=(-1.0, -1.0, -1.0), bb_max=(1.0, 1.0, 1.0), z_begin=0, z_end=None):
    g = Grid()
    g.dims = (C.c_uint32 * 3)(*[int(d) for d in dims])
    g.bb_min = f3(bb_min)
    g.bb_max = f3(bb_max)
    g.z_begin = int(z_begin)
    g.z_end = int(dims[2] if z_end is None else z_end)
    return g


def grid_from_bb(bb_min, bb_max, max_voxels_side):
    """SDFViewer::from_bb voxel sizing (scene/sdf/mod.rs:46-72)."""
    g = Grid()
    check(lib.sdfv_grid_from_bb(f3(bb_min), f3(bb_max), int(max_voxels_side), C.byref(g)))
    return g


def slab_voxels(grid):
    return int(grid.dims[0]) * int(grid.dims[1]) * (int(grid.z_end) - int(grid.z_begin))


def default_texture_skew(texture_bytes):
    """host/sdf_viewer.cpp's untuned placement: bytes between the end of tex0 and the start of tex1 in ONE block that MI355X has
    shown to be best, reproducibly, for the texture sizes that matter (EXPERIMENTS R4.1: the fill's rate is periodic in that
    distance)."""
    return {1 << 28: 12288, 1 << 30: 20480}.get(int(texture_bytes), 0)


def alloc_textures_placed(grid, device="cuda"):
    """Both textures in one block at default_texture_skew(): what SDFViewer::new_voxels allocates (no probe, deterministic)."""
    shape = (int(grid.z_end) - int(grid.z_begin), int(grid.dims[1]), int(grid.dims[0]), 4)
    n = shape[0] * shape[1] * shape[2] * 4
    skew = default_texture_skew(n * 4) // 4
    block = torch.empty(2 * n + skew + 64, dtype=torch.float32, device=device)
    pad = ((-block.data_ptr()) % 256) // 4
    return block[pad:pad + n].view(shape), block[pad + n + skew:pad + 2 * n + skew].view(shape)


def alloc_textures(grid, device="cuda"):
    """Two RGBA32F textures for the slab described by `grid` (uninitialised device memory), two separate allocations
    (alloc_textures_placed: one block, the distance between them the one MI355X fills fastest)."""
    shape = (int(grid.z_end) - int(grid.z_begin), int(grid.dims[1]), int(grid.dims[0]), 4)
    return (torch.empty(shape, dtype=torch.float32, device=device), torch.empty(shape, dtype=torch.float32, device=device))


def _stream_ptr(stream=None):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _dev_ptr(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
        raise TypeError(f"{what} must be a contiguous float32 CUDA(HIP) tensor")
    return C.c_void_p(t.data_ptr())


def grid_init(grid, tex0, tex1, stream=None):
    check(lib.sdfv_grid_init(C.byref(grid), _dev_ptr(tex0, "tex0"), _dev_ptr(tex1, "tex1"), _stream_ptr(stream)))


def grid_init_unvisited(grid, step, tex0, tex1, dist=None, stream=None, flags=0):
    """[AIR_DIST; 4] into the rows a pass with `step` does not visit (0: every row) -- sdfv_grid_init_unvisited[_ex]; flags:
    _capi.PASS_VOLUME_INTERLEAVED when `dist` is laid out y-interleaved."""
    check(lib.sdfv_grid_init_unvisited_ex(C.byref(grid), int(step), _dev_ptr(tex0, "tex0"), _dev_ptr(tex1, "tex1"),
                                          None if dist is None else _dev_ptr(dist, "dist"), int(flags), _stream_ptr(stream)))


def fill_grid(params, grid, tex0, tex1, sdf_id=SDF_DEMO, stream=None, dist=None):
    """Dense fill = final state of SDFViewer::update (scene/sdf/mod.rs:128-217).  `dist`: optional [D, H, W] tensor that
    receives the compact distance volume in the same pass (sdfv_fill_grid_commit)."""
    assert tex0.numel() == slab_voxels(grid) * 4 and tex1.numel() == slab_voxels(grid) * 4
    assert dist is None or dist.numel() == slab_voxels(grid)
    check(lib.sdfv_fill_grid_commit(C.byref(params), sdf_id, C.byref(grid), _dev_ptr(tex0, "tex0"),
                                    _dev_ptr(tex1, "tex1"), None if dist is None else _dev_ptr(dist, "dist"),
                                    _stream_ptr(stream)))


def fill_grid_pass(params, grid, step, tex0, tex1, changed_box=None, sdf_id=SDF_DEMO, stream=None, dist=None, flags=0):
    """One LoadingManager pass (loading.rs:50-76) with update_required (scene/sdf/mod.rs:184-190).  `dist`: the
    textures' compact distance volume, read instead of tex0 and kept in sync (sdfv_fill_grid_pass_ex's `dist`).  `flags`:
    _capi.PASS_FRESH_GRID / PASS_SAME_LOAD -- what the caller knows about the grid (sdfv_fill_grid_pass_ex)."""
    box = None if changed_box is None else (C.c_float * 6)(*[float(x) for x in changed_box])
    check(lib.sdfv_fill_grid_pass_ex(C.byref(params), sdf_id, C.byref(grid), int(step), box, _dev_ptr(tex0, "tex0"),
                                     _dev_ptr(tex1, "tex1"), None if dist is None else _dev_ptr(dist, "dist"), int(flags),
                                     _stream_ptr(stream)))


def pack_samples(grid, samples, tex0, tex1, indices=None, index_base=0, dist=None, flags=0, stream=None):
    """sdfv_pack_samples: SDFViewer::update's packing (scene/sdf/mod.rs:196-208) of host-taken samples, on the device.
    samples: CUDA float32 [n, 7] (SDFSample records); indices: CUDA int32/uint32-as-int32 [n] offsets from index_base, or None
    (a contiguous run from index_base)."""
    assert samples.is_cuda and samples.dtype == torch.float32 and samples.is_contiguous() and samples.shape[-1] == 7
    n = samples.numel() // 7
    assert indices is None or (indices.is_cuda and indices.is_contiguous() and indices.element_size() == 4 and indices.numel() == n)
    check(lib.sdfv_pack_samples(C.byref(grid), int(index_base), None if indices is None else C.c_void_p(indices.data_ptr()),
                                C.c_void_p(samples.data_ptr()), n, _dev_ptr(tex0, "tex0"), _dev_ptr(tex1, "tex1"),
                                None if dist is None else _dev_ptr(dist, "dist"), int(flags), _stream_ptr(stream)))


def sample_points(params, points, distance_only=False, sdf_id=SDF_DEMO, stream=None):
    """Batched SDFSurface::sample.  points: [n,3] CUDA tensor -> [n,7] (distance, rgb, metallic, roughness, occlusion)."""
    n = points.shape[0]
    out = torch.empty((n, 7), dtype=torch.float32, device=points.device)
    check(lib.sdfv_sample_points(C.byref(params), sdf_id, _dev_ptr(points, "points"), n, int(bool(distance_only)),
                                 C.c_void_p(out.data_ptr()), _stream_ptr(stream)))
    return out


def normal_points(params, points, eps=None, use_default=False, sdf_id=SDF_DEMO, stream=None):
    """Batched SDFSurface::normal(p, eps) (eps None -> <= 0 over the ABI, ffi.rs:326)."""
    n = points.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=points.device)
    check(lib.sdfv_normal_points(C.byref(params), sdf_id, _dev_ptr(points, "points"), n,
                                 float(eps) if eps else 0.0, int(bool(use_default)), C.c_void_p(out.data_ptr()),
                                 _stream_ptr(stream)))
    return out


VERTEX_FLOATS = 12  # sdfv_vertex: position, normal, color, metallic, roughness, occlusion (meshers/mesh.rs:135-143)


def source_sample_scalar(params, unit_points, bb_min=(-1.0, -1.0, -1.0), bb_max=(1.0, 1.0, 1.0), sdf_id=SDF_DEMO,
                         stream=None):
    """ScalarSource::sample_scalar over unit-cube points (meshers/isosurface.rs:78-84) -> [n] distances."""
    n = unit_points.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=unit_points.device)
    check(lib.sdfv_source_sample_scalar(C.byref(params), sdf_id, f3(bb_min), f3(bb_max),
                                        _dev_ptr(unit_points, "unit_points"), n, C.c_void_p(out.data_ptr()),
                                        _stream_ptr(stream)))
    return out


def source_sample_normal(params, unit_points, bb_min=(-1.0, -1.0, -1.0), bb_max=(1.0, 1.0, 1.0), sdf_id=SDF_DEMO,
                         stream=None):
    """HermiteSource::sample_normal over unit-cube points (meshers/isosurface.rs:87-92) -> [n, 3]."""
    n = unit_points.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=unit_points.device)
    check(lib.sdfv_source_sample_normal(C.byref(params), sdf_id, f3(bb_min), f3(bb_max),
                                        _dev_ptr(unit_points, "unit_points"), n, C.c_void_p(out.data_ptr()),
                                        _stream_ptr(stream)))
    return out


def mesh_postproc(params, vertices, sdf_id=SDF_DEMO, stream=None):
    """Mesh::postproc (meshers/mesh.rs:22-33), in place over an [n, 12] vertex tensor."""
    assert vertices.dim() == 2 and vertices.shape[1] == VERTEX_FLOATS
    check(lib.sdfv_mesh_postproc(C.byref(params), sdf_id, _dev_ptr(vertices, "vertices"), vertices.shape[0],
                                 _stream_ptr(stream)))
    return vertices


class _DeviceArray:
    """A library-owned device buffer seen through __cuda_array_interface__ (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr,
                                         "version": 2, "strides": None}


def mesh_extract(params, max_voxels_per_axis=64, bb_min=(-1.0, -1.0, -1.0), bb_max=(1.0, 1.0, 1.0), sdf_id=SDF_DEMO,
                 algorithm=0, stream=None):
    """Meshers::mesh (meshers/mod.rs:136-149): -> (vertices [n, 12] float32, indices [3 * triangles] int32), copied
    out of the library's buffers into torch tensors (the library's copy is freed before returning)."""
    m = _capi.Mesh()
    check(lib.sdfv_mesh_extract(C.byref(params), sdf_id, f3(bb_min), f3(bb_max), int(max_voxels_per_axis),
                                int(algorithm), C.byref(m), _stream_ptr(stream)))
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        if m.n_vertices:
            v = torch.as_tensor(_DeviceArray(m.vertices, (m.n_vertices, VERTEX_FLOATS), "<f4"), device=dev).clone()
        else:
            v = torch.empty((0, VERTEX_FLOATS), dtype=torch.float32, device=dev)
        if m.n_indices:
            i = torch.as_tensor(_DeviceArray(m.indices, (m.n_indices,), "<i4"), device=dev).clone()
        else:
            i = torch.empty((0,), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
    finally:
        lib.sdfv_mesh_free(C.byref(m))
    return v, i


def default_render_params(grid):
    rp = RenderParams()
    lib.sdfv_render_params_default(C.byref(rp), C.byref(grid))
    return rp


def camera_look_at(eye=(2.5, 3.0, 5.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovy_degrees=45.0, aspect=1.0,
                   z_near=0.1, z_far=1000.0):
    """Camera::new_perspective with the reference's defaults (scene/mod.rs:82-95)."""
    cam = Camera()
    check(lib.sdfv_camera_look_at(C.byref(cam), f3(eye), f3(target), f3(up), fovy_degrees, aspect, z_near, z_far))
    return cam


def orbit_cameras(n, aspect, eye0=(2.5, 3.0, 5.0)):
    """camera k on the orbit of camera 0 around +y, azimuth atan2(z, x) + 2*pi*k/n."""
    import math
    r = math.hypot(eye0[0], eye0[2])
    a0 = math.atan2(eye0[2], eye0[0])
    cams = []
    for k in range(n):
        a = a0 + 2.0 * math.pi * k / n
        cams.append(camera_look_at(eye=(r * math.cos(a), eye0[1], r * math.sin(a)), aspect=aspect))
    return cams


RAY_STATE_WORDS = C.sizeof(_capi.RayState) // 4  # pixel, iteration, pos[3], t


def raymarch_slab(rp, grid, ghost_lo, ghost_hi, tex0, tex1, camera, width, height, rgba, out_down, out_up, counters,
                  in_states=None, aux=None, stream=None):
    """One round of the sharded march on this rank (sdfv_raymarch_slab).  in_states None = first round.
    out_down/out_up: [capacity, 6] int32 device tensors, counters: [2] int32 device tensor (zeroed by the caller)."""
    n_in = 0 if in_states is None else int(in_states.shape[0])
    if in_states is not None and n_in == 0:
        return  # a continuation round in which nothing arrived (an empty tensor has no address to pass)
    check(lib.sdfv_raymarch_slab(C.byref(rp), C.byref(grid), int(ghost_lo), int(ghost_hi), _dev_ptr(tex0, "tex0"),
                                 _dev_ptr(tex1, "tex1"), C.byref(camera), width, height,
                                 None if in_states is None else C.c_void_p(in_states.data_ptr()), n_in,
                                 _dev_ptr(rgba, "rgba"), None if aux is None else C.c_void_p(aux.data_ptr()),
                                 C.c_void_p(out_down.data_ptr()), C.c_void_p(out_up.data_ptr()),
                                 int(out_down.shape[0]), C.c_void_p(counters.data_ptr()), _stream_ptr(stream)))


def ray_buffer(capacity, device="cuda"):
    """A fixed-capacity ray list with its count in band (sdfv_ray_buffer_bytes): int32 words, header first."""
    words = lib.sdfv_ray_buffer_bytes(int(capacity)) // 4
    return torch.zeros(words, dtype=torch.int32, device=device)


def raymarch_slab_round(rp, grid, ghost_lo, ghost_hi, tex0, tex1, camera, width, height, rgba, out_down, out_up, capacity,
                        in_lo=None, in_hi=None, first_round=False, aux=None, overflow=None, stream=None):
    """One round of the sharded march with device-side counts (sdfv_raymarch_slab_round): nothing to read back.
    in_lo / in_hi / out_down / out_up: ray_buffer() tensors."""
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    check(lib.sdfv_raymarch_slab_round(C.byref(rp), C.byref(grid), int(ghost_lo), int(ghost_hi), _dev_ptr(tex0, "tex0"),
                                       _dev_ptr(tex1, "tex1"), C.byref(camera), width, height, p(in_lo), p(in_hi),
                                       1 if first_round else 0, _dev_ptr(rgba, "rgba"), p(aux), p(out_down), p(out_up),
                                       int(capacity), p(overflow), _stream_ptr(stream)))


def commit_distance(grid, tex0, dist=None, stream=None):
    """Device-side commit: compact copy of tex0.r for the raymarch (sdfv_commit_distance)."""
    if dist is None:
        dist = torch.empty(tuple(tex0.shape[:-1]), dtype=torch.float32, device=tex0.device)
    check(lib.sdfv_commit_distance(C.byref(grid), _dev_ptr(tex0, "tex0"), _dev_ptr(dist, "dist"), _stream_ptr(stream)))
    return dist


def commit_pairs(grid, dist, pairs=None, stream=None):
    """The y-pair volume of the raymarch (sdfv_commit_pairs): [D, H, W, 2] floats from the compact distance volume."""
    if pairs is None:
        pairs = torch.empty(tuple(dist.shape) + (2,), dtype=torch.float32, device=dist.device)
    check(lib.sdfv_commit_pairs(C.byref(grid), _dev_ptr(dist, "dist"), _dev_ptr(pairs, "pairs"), _stream_ptr(stream)))
    return pairs


def march_volume_advice(grid):
    """"pairs", "interleaved" or None: the acceleration volume a many-frames host should build for this grid (speed only)."""
    kind = C.c_uint32(0)
    check(lib.sdfv_march_volume_advice(C.byref(grid), C.byref(kind)))
    return {0: None, 1: "pairs", 2: "interleaved"}[kind.value]


def commit_interleaved(grid, dist, ilv=None, stream=None):
    """The y-interleaved volume of the raymarch (sdfv_commit_interleaved): same floats as `dist`, rows paired."""
    if ilv is None:
        ilv = torch.empty_like(dist)
    check(lib.sdfv_commit_interleaved(C.byref(grid), _dev_ptr(dist, "dist"), _dev_ptr(ilv, "ilv"), _stream_ptr(stream)))
    return ilv


def set_option(option, value):
    """sdfv_set_option: per-thread option of the library (A/B measurements, forcing kernel specialisations in tests)."""
    check(lib.sdfv_set_option(int(option), int(value)))


def get_option(option):
    v = C.c_uint64(0)
    check(lib.sdfv_get_option(int(option), C.byref(v)))
    return int(v.value)


class options:
    """with pkg.options({pkg._capi.OPT_FILL_FORM: 2}): ...  -- sets the options, restores the previous values on exit."""

    def __init__(self, values):
        self.values = dict(values)

    def __enter__(self):
        self.saved = {k: get_option(k) for k in self.values}
        for k, v in self.values.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            set_option(k, v)
        return False


def upload_cameras(cameras, device="cuda"):
    """The cameras as a device tensor (120 bytes each) that raymarch() hands to the library as it is: an array in DEVICE memory
    is read in place whatever its length; a host array of more than 16 is staged by the launcher on every call."""
    arr = (Camera * len(cameras))(*cameras)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def raymarch(rp, tex0, tex1, cameras, width, height, y0=0, y1=None, want_aux=False, stream=None, out=None, dist=None,
             want_depth=False, depth_out=None, pairs=None, ilv=None, bands=None, rgba8=None, f32=True):
    """material.frag main() over rows [y0,y1) -- or, bands=(first, step), over the 16-row tile bands first, first + step, ...
    stored one after the other (sdfv_march_desc.band_first / band_step).  Returns rgba [n_cam, rows, W, 4] (+ aux [n_cam, rows, W, 18] words).
    cameras: a Camera, a list of them, or upload_cameras()'s device tensor (any number of cameras).
    `dist` = optional compact distance volume from commit_distance(); `pairs` = optional y-pair volume from commit_pairs()
    (sdfv_march_desc.pairs), `ilv` = the y-interleaved volume (desc.ilv).  want_depth / depth_out: also return the
    gl_FragDepth plane [n_cam, rows, W] (desc.depth); return order: rgba[, depth][, aux].  One export: sdfv_raymarch_ex.
    rgba8: "both" or a uint8 tensor [n_cam, rows, W, 4] -> the 8-bit UNORM plane is written too (desc.rgba8) and returned last;
    f32=False (or rgba8="only") -> the fp32 plane is NOT written (desc.rgba = NULL: 4 B per pixel stored instead of 16)."""
    if isinstance(cameras, Camera):
        cameras = [cameras]
    y1 = height if y1 is None else y1
    if isinstance(cameras, torch.Tensor):  # upload_cameras(): cameras in device memory, read in place
        n = cameras.numel() * cameras.element_size() // C.sizeof(Camera)
        cam_arr = C.cast(C.c_void_p(cameras.data_ptr()), C.POINTER(Camera))
    else:
        n = len(cameras)
        cam_arr = (Camera * n)(*cameras)
    d = _capi.MarchDesc()
    d.size = C.sizeof(d)
    if bands is not None:
        if (y0, y1) != (0, height):
            raise ValueError("bands and a row range exclude each other")
        if int(bands[1]) < 1:
            raise SdfvError(-1, "band_step is 0")
        bh = int(bands[2]) if len(bands) > 2 else 16  # bands = (first, step[, rows per band: 16 or 8])
        if bh not in (8, 16):
            raise SdfvError(-1, f"band_height {bh}: 8 or 16")
        rows = int(lib.sdfv_band_rows_ex(height, int(bands[0]), int(bands[1]), bh))
        d.y0, d.y1, d.band_first, d.band_step, d.band_height = 0, height, int(bands[0]), int(bands[1]), bh
    else:
        rows = y1 - y0
        d.y0, d.y1 = y0, y1
    img8 = None
    if rgba8 is not None:
        img8 = rgba8 if isinstance(rgba8, torch.Tensor) else torch.empty((n, rows, width, 4), dtype=torch.uint8, device=tex0.device)
        assert img8.dtype == torch.uint8 and img8.is_contiguous() and img8.numel() == n * rows * width * 4
    only8 = (isinstance(rgba8, str) and rgba8 == "only") or (img8 is not None and not f32)
    rgba = None if only8 else (out if out is not None else torch.empty((n, rows, width, 4), dtype=torch.float32, device=tex0.device))
    aux = torch.empty((n, rows, width, AUX_FLOATS), dtype=torch.int32, device=tex0.device) if want_aux else None
    depth = depth_out
    if depth is None and want_depth:
        depth = torch.empty((n, rows, width), dtype=torch.float32, device=tex0.device)
    def result():
        ret = (() if only8 else (rgba,)) + ((depth,) if (want_depth or depth_out is not None) else ()) + ((aux,) if want_aux else ()) + \
              (() if img8 is None else (img8,))
        return ret if len(ret) > 1 else ret[0]
    if rows == 0 and bands is not None and int(bands[1]) > 0:  # a band set below the image: nothing to render (empty tensors have no address)
        return result()
    d.rp = C.pointer(rp)
    d.tex0, d.tex1 = _dev_ptr(tex0, "tex0"), _dev_ptr(tex1, "tex1")
    d.dist = None if dist is None else _dev_ptr(dist, "dist")
    d.pairs = None if pairs is None else _dev_ptr(pairs, "pairs")
    d.ilv = None if ilv is None else _dev_ptr(ilv, "ilv")
    d.cameras, d.n_cameras, d.width, d.height = cam_arr, n, width, height
    d.rgba = None if rgba is None else rgba.data_ptr()
    d.depth = None if depth is None else _dev_ptr(depth, "depth")
    d.aux = aux.data_ptr() if want_aux else None
    d.rgba8 = None if img8 is None else img8.data_ptr()
    check(lib.sdfv_raymarch_ex(C.byref(d), _stream_ptr(stream)))
    return result()
