"""ctypes binding of oracle/liboracle.so -- the CPU restatement used ONLY as the checker in tests,
smoke() and bench.py's cpu_baseline leg.  Never imported by the product package."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))


class Sample(C.Structure):
    _fields_ = [("distance", C.c_float), ("color", C.c_float * 3), ("metallic", C.c_float),
                ("roughness", C.c_float), ("occlusion", C.c_float)]


class DemoParams(C.Structure):
    _fields_ = [("cube_half_side", C.c_float), ("cube_material", C.c_uint32),
                ("sphere_radius", C.c_float), ("sphere_material", C.c_uint32),
                ("max_distance_custom_material", C.c_float), ("disable_sphere", C.c_uint32)]


class LoadingManager(C.Structure):
    _fields_ = [("limits", C.c_uint64 * 3), ("passes", C.c_uint64), ("step_size", C.c_uint64),
                ("next_index", C.c_uint64 * 3), ("iterations", C.c_uint64), ("total_iterations", C.c_uint64)]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("tan_half_fovy", C.c_float), ("aspect", C.c_float),
                ("bvp", C.c_float * 16)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("intensity", C.c_float), ("direction", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("tex_size", C.c_uint32 * 3),
                ("lod_dist_between_samples", C.c_float), ("tint", C.c_float * 4), ("ambient", C.c_float * 3),
                ("gamma", C.c_float), ("tone_mapping", C.c_uint32), ("color_mapping", C.c_uint32),
                ("n_lights", C.c_uint32), ("lights", Light * 4)]


AUX_DTYPE = np.dtype([("status", "<i4"), ("steps", "<i4"), ("hit_pos", "<f4", 3), ("t", "<f4"),
                      ("raw0", "<f4", 4), ("raw1", "<f4", 4), ("normal", "<f4", 3), ("depth", "<f4")])
assert AUX_DTYPE.itemsize == 72

FP = C.POINTER(C.c_float)
L.or_air_dist.restype = C.c_float
L.or_voxel_coord.restype = C.c_float
L.or_voxel_coord.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_float]
L.or_srgb_quantize.restype = C.c_uint8
L.or_srgb_quantize.argtypes = [C.c_float]
L.or_srgb_u8_to_linear.restype = C.c_float
L.or_srgb_u8_to_linear.argtypes = [C.c_uint8]
L.or_lm_next.restype = C.c_int
L.or_lm_len.restype = C.c_uint64
L.or_lm_passes_left.restype = C.c_uint64
L.or_prev_power_of_2.restype = C.c_uint32
L.or_prev_power_of_2.argtypes = [C.c_uint32]
L.or_viewer_update.restype = C.c_uint64
L.or_viewer_update.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_uint64, C.c_void_p, C.c_void_p]
L.or_fill_dense.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                            C.c_void_p, C.c_void_p, C.c_int]
L.or_raymarch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                          C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
L.or_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
L.or_normal.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_void_p]
L.or_normal_default.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_void_p]
L.or_camera_look_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                C.c_float]
L.or_tex_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.or_shade.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
L.or_pack_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
L.or_grid_dims_from_bb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
L.or_default_render_params.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

AIR_DIST = L.or_air_dist()


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def u3(v):
    return (C.c_uint32 * 3)(*[int(x) for x in v])


def default_params(**kw):
    p = DemoParams()
    L.or_demo_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def params_from(other):
    """Copy a product DemoParams (same layout) into an oracle DemoParams."""
    p = DemoParams()
    C.memmove(C.byref(p), C.byref(other), C.sizeof(p))
    return p


def sample(params, p, distance_only=False, sdf_id=0):
    s = Sample()
    L.or_sample(C.byref(params), sdf_id, f3(p), int(distance_only), C.byref(s))
    return np.array([s.distance, *s.color, s.metallic, s.roughness, s.occlusion], dtype=np.float32)


def sample_many(params, pts, distance_only=False, sdf_id=0):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty((len(pts), 7), dtype=np.float32)
    s = Sample()
    for i in range(len(pts)):
        L.or_sample(C.byref(params), sdf_id, pts[i].ctypes.data, int(distance_only), C.byref(s))
        out[i] = (s.distance, *s.color, s.metallic, s.roughness, s.occlusion)
    return out


def normal_many(params, pts, eps=0.0, sdf_id=0, use_default=False):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    out = np.empty((len(pts), 3), dtype=np.float32)
    fn = L.or_normal_default if use_default else L.or_normal
    for i in range(len(pts)):
        fn(C.byref(params), sdf_id, pts[i].ctypes.data, float(eps), out[i].ctypes.data)
    return out


def pack(sample7):
    s = Sample(float(sample7[0]), (C.c_float * 3)(*[float(x) for x in sample7[1:4]]), float(sample7[4]),
               float(sample7[5]), float(sample7[6]))
    t0 = np.zeros(4, np.float32)
    t1 = np.full(4, AIR_DIST, np.float32)
    L.or_pack_sample(C.byref(s), t0.ctypes.data, t1.ctypes.data)
    return t0, t1


def fill_dense(params, dims, bb_min=(-1, -1, -1), bb_max=(1, 1, 1), z0=0, z1=None, sdf_id=0, threads=8):
    z1 = dims[2] if z1 is None else z1
    shape = (z1 - z0, dims[1], dims[0], 4)
    t0 = np.empty(shape, np.float32)
    t1 = np.empty(shape, np.float32)
    L.or_fill_dense(C.byref(params), sdf_id, u3(dims), f3(bb_min), f3(bb_max), z0, z1, t0.ctypes.data,
                    t1.ctypes.data, threads)
    return t0, t1


def grid_init(dims):
    shape = (dims[2], dims[1], dims[0], 4)
    return np.full(shape, AIR_DIST, np.float32), np.full(shape, AIR_DIST, np.float32)


def lm_new(limits, passes):
    m = LoadingManager()
    L.or_lm_new(C.byref(m), (C.c_uint64 * 3)(*limits), passes)
    return m


def lm_next(m):
    idx = (C.c_uint64 * 3)()
    return tuple(idx) if L.or_lm_next(C.byref(m), idx) else None


def viewer_update(params, dims, lm, t0, t1, changed_box=None, max_iterations=2 ** 62, bb_min=(-1, -1, -1),
                  bb_max=(1, 1, 1), sdf_id=0):
    cb = None if changed_box is None else (C.c_float * 6)(*[float(x) for x in changed_box])
    return L.or_viewer_update(C.byref(params), sdf_id, u3(dims), f3(bb_min), f3(bb_max), C.byref(lm), cb,
                              max_iterations, t0.ctypes.data, t1.ctypes.data)


SAMPLE_FN = C.CFUNCTYPE(None, C.c_void_p, FP, C.c_int, FP)
L.or_viewer_update_fn.restype = C.c_uint64
L.or_viewer_update_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_void_p, C.c_void_p]


def viewer_update_fn(sample_fn, dims, lm, t0, t1, changed_box=None, max_iterations=2 ** 62, bb_min=(-1, -1, -1),
                     bb_max=(1, 1, 1), user=None):
    """SDFViewer::update's loop over ANY sample function: `sample_fn` is the address of a C function with or_sample_fn's
    signature (or a SAMPLE_FN-wrapped Python callable)."""
    cb = None if changed_box is None else (C.c_float * 6)(*[float(x) for x in changed_box])
    fn = C.cast(sample_fn, C.c_void_p)
    return L.or_viewer_update_fn(fn, user, u3(dims), f3(bb_min), f3(bb_max), C.byref(lm), cb, max_iterations,
                                 t0.ctypes.data, t1.ctypes.data)


def default_render_params(dims, bb_min=(-1, -1, -1), bb_max=(1, 1, 1)):
    rp = RenderParams()
    L.or_default_render_params(C.byref(rp), u3(dims), f3(bb_min), f3(bb_max))
    return rp


def camera_look_at(eye=(2.5, 3.0, 5.0), target=(0, 0, 0), up=(0, 1, 0), fovy=45.0, aspect=1.0, near=0.1, far=1000.0):
    cam = Camera()
    L.or_camera_look_at(C.byref(cam), f3(eye), f3(target), f3(up), fovy, aspect, near, far)
    return cam


def copy_struct(dst_type, src):
    d = dst_type()
    assert C.sizeof(d) == C.sizeof(src)
    C.memmove(C.byref(d), C.byref(src), C.sizeof(d))
    return d


def raymarch(rp, t0, t1, cam, width, height, y0=0, y1=None, threads=8, want_aux=True):
    y1 = height if y1 is None else y1
    rgba = np.empty((y1 - y0, width, 4), np.float32)
    aux = np.empty((y1 - y0, width), AUX_DTYPE) if want_aux else None
    L.or_raymarch(C.byref(rp), t0.ctypes.data, t1.ctypes.data, C.byref(cam), width, height, y0, y1, rgba.ctypes.data,
                  aux.ctypes.data if want_aux else None, threads)
    return rgba, aux


class MarchCounts(C.Structure):
    _fields_ = [("pixels", C.c_uint64), ("covered", C.c_uint64), ("hits", C.c_uint64), ("sum_steps", C.c_uint64),
                ("max_steps", C.c_uint64)]


L.or_raymarch_touch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
L.or_set_ext_variant.argtypes = [C.c_uint32]
L.or_get_ext_variant.restype = C.c_uint32
EXT_VARIANTS = {"srgb_pow_ulp_up": 1, "srgb_pow_ulp_down": 2, "srgb_quant_round": 4, "cgmath_normalize_div": 8,
                "glsl_mix_lerp": 16, "glsl_normalize_rsq": 32, "trilinear_weighted_sum": 64, "srgb_double_pow": 128}


def raymarch_touch(rp, t0, t1, cam, width, height, y0=0, y1=None, threads=8, maps=None):
    """The march with texel-touch recording (SURVEY 8d byte model).  maps: dict of uint8 arrays [D, H, W] keyed
    march0 / hit0 / hit1 / normal0 (created when None; pass the same dict again to accumulate over cameras).
    Returns (maps, counts dict)."""
    y1 = height if y1 is None else y1
    shape = t0.shape[:3]
    if maps is None:
        maps = {k: np.zeros(shape, np.uint8) for k in ("march0", "hit0", "hit1", "normal0")}
    c = MarchCounts()
    L.or_raymarch_touch(C.byref(rp), t0.ctypes.data, t1.ctypes.data, C.byref(cam), width, height, y0, y1,
                        maps["march0"].ctypes.data, maps["hit0"].ctypes.data, maps["hit1"].ctypes.data,
                        maps["normal0"].ctypes.data, C.byref(c), threads)
    return maps, {k: int(getattr(c, k)) for k, _ in MarchCounts._fields_}


# ---- mesher front end (oracle/mesh_front.c) ----
VERTEX_FLOATS = 12
L.or_source_scalar.restype = C.c_float
L.or_source_scalar.argtypes = [C.c_void_p, C.c_uint32, FP, FP, FP]
L.or_source_normal.argtypes = [C.c_void_p, C.c_uint32, FP, FP, FP, FP]
L.or_vert_pos_to.argtypes = [FP, FP, FP, FP]
L.or_mesh_postproc.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
L.or_ply_color_u8.restype = C.c_uint8
L.or_ply_color_u8.argtypes = [C.c_float]


def source_scalar_many(params, unit_pts, bb_min=(-1, -1, -1), bb_max=(1, 1, 1), sdf_id=0):
    out = np.empty(len(unit_pts), np.float32)
    lo, hi = f3(bb_min), f3(bb_max)
    for i, p in enumerate(unit_pts):
        out[i] = L.or_source_scalar(C.byref(params), sdf_id, lo, hi, f3(p))
    return out


def source_normal_many(params, unit_pts, bb_min=(-1, -1, -1), bb_max=(1, 1, 1), sdf_id=0):
    out = np.empty((len(unit_pts), 3), np.float32)
    lo, hi = f3(bb_min), f3(bb_max)
    n = (C.c_float * 3)()
    for i, p in enumerate(unit_pts):
        L.or_source_normal(C.byref(params), sdf_id, lo, hi, f3(p), n)
        out[i] = n[:]
    return out


def mesh_postproc(params, vertices, sdf_id=0):
    """vertices: [n, 12] float32 (sdfv_vertex layout); returns the post-processed copy."""
    v = np.ascontiguousarray(vertices, np.float32).copy()
    L.or_mesh_postproc(C.byref(params), sdf_id, v.ctypes.data, len(v))
    return v
