"""ctypes binding of include/sdfgrid.h (libsdfgrid.so).

This is the stub a maintainer of the reference would write in Rust as `extern "C"` declarations
(see INTEGRATION.md); here it is ctypes because the image has no Rust toolchain.  There is no
fallback: if the shared library is missing or was not built, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDFGRID_LIBRARY (read by this HARNESS, not by the library): load another build of the same ABI, e.g. the tuning
# build libsdfgrid_tuning.so that tools/wave_timing.py needs.
LIB_PATH = os.environ.get("SDFGRID_LIBRARY") or os.path.join(_HERE, "libsdfgrid.so")


class SdfvError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libsdfgrid error {code}: {message}")
        self.code = code


class Sample(C.Structure):  # src/sdf/mod.rs:104-118
    _fields_ = [("distance", C.c_float), ("color", C.c_float * 3), ("metallic", C.c_float),
                ("roughness", C.c_float), ("occlusion", C.c_float)]


class DemoParams(C.Structure):  # cube.rs:15-18, sphere.rs:11-14, demo/mod.rs:26-29
    _fields_ = [("cube_half_side", C.c_float), ("cube_material", C.c_uint32),
                ("sphere_radius", C.c_float), ("sphere_material", C.c_uint32),
                ("max_distance_custom_material", C.c_float), ("disable_sphere", C.c_uint32)]


class Grid(C.Structure):
    _fields_ = [("dims", C.c_uint32 * 3), ("bb_min", C.c_float * 3), ("bb_max", C.c_float * 3),
                ("z_begin", C.c_uint32), ("z_end", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("tan_half_fovy", C.c_float), ("aspect", C.c_float),
                ("bvp", C.c_float * 16)]


class Light(C.Structure):  # sdfv_light
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("intensity", C.c_float), ("direction", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("tex_size", C.c_uint32 * 3),
                ("lod_dist_between_samples", C.c_float), ("tint", C.c_float * 4), ("ambient", C.c_float * 3),
                ("gamma", C.c_float), ("tone_mapping", C.c_uint32), ("color_mapping", C.c_uint32),
                ("n_lights", C.c_uint32), ("lights", Light * 4)]


class MarchAux(C.Structure):
    _fields_ = [("status", C.c_int32), ("steps", C.c_int32), ("hit_pos", C.c_float * 3), ("t", C.c_float),
                ("raw0", C.c_float * 4), ("raw1", C.c_float * 4), ("normal", C.c_float * 3), ("depth", C.c_float)]


assert C.sizeof(Sample) == 28 and C.sizeof(MarchAux) == 72 and C.sizeof(Camera) == 120

SDF_DEMO, SDF_CUBE, SDF_SPHERE = 0, 1, 2
MATERIAL_BRICK, MATERIAL_NORMAL = 0, 1

# name -> (restype, argtypes); every symbol include/sdfgrid.h declares
class RayState(C.Structure):
    _fields_ = [("pixel", C.c_uint32), ("iteration", C.c_uint32), ("pos", C.c_float * 3), ("t", C.c_float)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("indices", C.c_void_p), ("n_vertices", C.c_size_t), ("n_indices", C.c_size_t)]


class MarchDesc(C.Structure):
    """sdfv_march_desc: the one descriptor of sdfv_raymarch_ex (size-prefixed)."""
    _fields_ = [("size", C.c_uint32), ("reserved", C.c_uint32), ("rp", C.POINTER(RenderParams)), ("tex0", C.c_void_p),
                ("tex1", C.c_void_p), ("dist", C.c_void_p), ("pairs", C.c_void_p), ("ilv", C.c_void_p),
                ("cameras", C.POINTER(Camera)), ("n_cameras", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("y0", C.c_uint32), ("y1", C.c_uint32), ("band_first", C.c_uint32), ("band_step", C.c_uint32),
                ("band_height", C.c_uint32), ("rgba", C.c_void_p), ("depth", C.c_void_p), ("aux", C.c_void_p),
                ("rgba8", C.c_void_p)]


PROTOTYPES = {
    "sdfv_raymarch_ex": (C.c_int, [C.POINTER(MarchDesc), C.c_void_p]),
    "sdfv_abi_version": (C.c_uint32, []),
    "sdfv_last_error": (C.c_char_p, []),
    "sdfv_build_id": (C.c_char_p, []),
    "sdfv_device_count": (C.c_int, []),
    "sdfv_air_dist": (C.c_float, []),
    "sdfv_set_option": (C.c_int, [C.c_uint32, C.c_uint64]),
    "sdfv_get_option": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint64)]),
    "sdfv_demo_params_default": (None, [C.POINTER(DemoParams)]),
    "sdfv_grid_from_bb": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(Grid)]),
    "sdfv_render_params_default": (None, [C.POINTER(RenderParams), C.POINTER(Grid)]),
    "sdfv_camera_look_at": (C.c_int, [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float]),
    "sdfv_grid_init": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_grid_init_unvisited": (C.c_int, [C.POINTER(Grid), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_grid_init_unvisited_ex": (C.c_int, [C.POINTER(Grid), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "sdfv_fill_grid": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(Grid), C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "sdfv_pack_samples": (C.c_int, [C.POINTER(Grid), C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint32, C.c_void_p]),
    "sdfv_fill_grid_commit": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(Grid), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "sdfv_fill_grid_pass_ex": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(Grid), C.c_uint32,
                                         C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "sdfv_sample_points": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "sdfv_normal_points": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_float,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "sdfv_commit_distance": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_commit_pairs": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_band_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "sdfv_band_rows_ex": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "sdfv_march_volume_advice": (C.c_int, [C.POINTER(Grid), C.POINTER(C.c_uint32)]),
    "sdfv_commit_interleaved": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_fill_grid_host": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(Grid), C.c_void_p, C.c_void_p]),
    "sdfv_sample_points_host": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_int,
                                          C.c_void_p]),
    "sdfv_normal_points_host": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_float,
                                          C.c_int, C.c_void_p]),
    "sdfv_raymarch_host": (C.c_int, [C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Camera),
                                     C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "sdfv_source_sample_scalar": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sdfv_source_sample_normal": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "sdfv_mesh_postproc": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdfv_mesh_postproc_host": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.c_void_p, C.c_size_t]),
    "sdfv_mesh_extract": (C.c_int, [C.POINTER(DemoParams), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.c_uint32, C.c_uint32, C.POINTER(Mesh), C.c_void_p]),
    "sdfv_mesh_free": (C.c_int, [C.POINTER(Mesh)]),
    "sdfv_mesh_trim": (C.c_int, []),
    "sdfv_raymarch_slab": (C.c_int, [C.POINTER(RenderParams), C.POINTER(Grid), C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "sdfv_ray_buffer_bytes": (C.c_size_t, [C.c_uint32]),
    "sdfv_raymarch_slab_round": (C.c_int, [C.POINTER(RenderParams), C.POINTER(Grid), C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                           C.c_void_p]),
    "sdfv_slab_march_scratch_bytes": (C.c_size_t, [C.c_uint32]),
    "sdfv_slab_march": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.POINTER(Grid), C.c_void_p, C.c_void_p,
                                  C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "sdfv_slab_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "sdfv_slab_comm_create": (C.c_int, [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "sdfv_slab_comm_destroy": (C.c_int, [C.c_void_p]),
    "sdfv_slab_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sdfv_slab_comm_ranks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sdfv_slab_halo_exchange": (C.c_int, [C.c_void_p, C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_slab_fill_step": (C.c_int, [C.c_void_p, C.POINTER(DemoParams), C.c_uint32, C.POINTER(Grid), C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "sdfv_slab_fill_step_commit": (C.c_int, [C.c_void_p, C.POINTER(DemoParams), C.c_uint32, C.POINTER(Grid), C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdfv_slab_comm_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sdfv_bands_scatter": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_void_p, C.c_void_p]),
    "sdfv_comm_gather_bands_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "sdfv_comm_gather_bands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdfv_comm_gather_cameras": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                           C.c_void_p, C.c_void_p]),
    "sdfv_comm_allgather_slabs": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}
LIGHT_AMBIENT, LIGHT_DIRECTIONAL, MAX_LIGHTS = 0, 1, 4


def raymarch_rc(rp, tex0, tex1, cameras, n_cameras, width, height, y0, y1, rgba, depth=None, aux=None, dist=None, pairs=None,
                ilv=None, band_first=0, band_step=0, stream=None, size=None, band_height=0):
    """sdfv_raymarch_ex from raw addresses (ints / c_void_p / None), returning the status code unchecked: what the argument-
    error tests and the tools that bypass the torch harness call.  cameras: a Camera, a ctypes array of them, or None."""
    d = MarchDesc()
    d.size = C.sizeof(d) if size is None else size
    d.rp = C.pointer(rp) if rp is not None else None
    d.tex0, d.tex1, d.dist, d.pairs, d.ilv = tex0, tex1, dist, pairs, ilv
    if cameras is not None:
        d.cameras = C.cast(C.pointer(cameras), C.POINTER(Camera))
    d.n_cameras, d.width, d.height, d.y0, d.y1 = n_cameras, width, height, y0, y1
    d.band_first, d.band_step, d.band_height = band_first, band_step, band_height
    d.rgba, d.depth, d.aux = rgba, depth, aux
    return lib.sdfv_raymarch_ex(C.byref(d), stream)
# sdfv_option / values (include/sdfgrid.h)
OPT_FILL_NONTEMPORAL, OPT_FILL_FORM, OPT_RAYMARCH_DISABLE, OPT_RAYMARCH_KEEP_NORMAL, OPT_SLAB_STEP_FORM = 1, 2, 3, 4, 5
OPT_RAYMARCH_TILE_GROUP = 6
OPT_RAYMARCH_BOX_FIRST = 7
OPT_RAYMARCH_WAVES_PER_SIMD = 8
OPT_RAYMARCH_BATCH_STREAMS = 9
OPT_PASS_INDEX_LIMIT = 11
OPT_RAYMARCH_CAMERA_STAGING = 12
OPT_PASS_FORM = 13  # 0 auto | 1 per-voxel kernels for unflagged passes (A/B)
OPT_RCCL_LIBRARY = 14  # process-wide: address of the path of the RCCL-ABI library to load (before the first communicator)
OPT_PASS_LOADS = 15  # 0 auto (SDFV_PASS_EXPECT_NOOP decides) | 1 cached | 2 nontemporal loads for update_required
OPT_EXT_SRGB_QUANT = 10  # Srgba::from(Vec3): 0 truncate (default) | 1 round
OPT_TUNING_WAVE_TIMING = 100
OPT_TUNING_PRIORITY_MAP = 101
OPT_TUNING_TILE_ORDER = 102
RM_NO_FAST_INDEX, RM_NO_POW2_EXTENT, RM_NO_POW2_SIZE, RM_NO_SYMMETRIC, RM_NO_ASM_LOOP, RM_NO_INTERIOR_FETCH = 1, 2, 4, 8, 16, 32
STEP_SIDE_BOUNDARY, STEP_UNPACKED, STEP_START_EVENT, STEP_DEFER_JOIN = 3, 4, 8, 16
PASS_FRESH_GRID, PASS_SAME_LOAD, PASS_VIRGIN_GRID, PASS_VOLUME_INTERLEAVED, PASS_EXPECT_NOOP = 1, 2, 4, 8, 16
FILL_FORM = {"auto": 0, "rows": 1, "flat": 2}
COMM_ID_BYTES = 128
RAY_BUFFER_HEADER_BYTES = 16
MARCH_MERGE = 1
COMM_PERIODIC = 1
COMM_HALO2 = 2


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sdf-viewer_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what the header declares
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = load()


def check(rc):
    if rc != 0:
        raise SdfvError(rc, lib.sdfv_last_error().decode("utf-8", "replace"))


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])
