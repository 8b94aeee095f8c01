"""ctypes binding of the C++ host mirror (through the TEST-ONLY libsdfviewer_host_test.so = the mirror + host_capi.cpp's flat shims) and of the per-point provider library
(libsdfdemo_provider.so, the reference's ffi.rs ABI)."""
import ctypes as C
import os

import numpy as np

try:  # the GPU tests mix this library with torch in one process: torch bundles its own HIP/HSA runtime, and whichever
    import torch  # noqa: F401  runtime is loaded first serves both -- ROCm's first leaves torch without devices
except ImportError:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# SDFV_HOST_TEST_LIB: another build of the same library (a sanitizer build run under LD_PRELOAD=libasan.so, say)
H = C.CDLL(os.environ.get("SDFV_HOST_TEST_LIB") or os.path.join(ROOT, "sdf-viewer_amd", "libsdfviewer_host_test.so"))
PROVIDER_PATH = os.path.join(ROOT, "sdf-viewer_amd", "libsdfdemo_provider.so")

SZ = C.c_size_t
for name, res, args in [
    ("sdfvh_lm_new", C.c_void_p, [SZ, SZ, SZ, SZ]), ("sdfvh_lm_free", None, [C.c_void_p]),
    ("sdfvh_lm_next", C.c_int, [C.c_void_p, C.POINTER(SZ)]), ("sdfvh_lm_len", SZ, [C.c_void_p]),
    ("sdfvh_lm_total_iterations", SZ, [C.c_void_p]), ("sdfvh_lm_passes_left", SZ, [C.c_void_p]),
    ("sdfvh_lm_step_size", SZ, [C.c_void_p]), ("sdfvh_lm_finish_pass", SZ, [C.c_void_p]),
    ("sdfvh_prev_power_of_2", C.c_uint32, [C.c_uint32]),
    ("sdfvh_demo_new", C.c_void_p, [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, SZ]),
    ("sdfvh_provider_load", C.c_void_p, [C.c_char_p, C.c_char_p, SZ]),
    ("sdfvh_sdf_sample_concurrency", C.c_uint, [C.c_void_p]),
    ("sdfvh_viewer_set_ingest", None, [C.c_void_p, C.c_uint, SZ]), ("sdfvh_viewer_last_error", SZ, [C.c_void_p, C.c_char_p, SZ]),
    ("sdfvh_lm_advance", None, [C.c_void_p, SZ]), ("sdfvh_lm_cursor", SZ, [C.c_void_p]),
    ("sdfvh_lm_pass_remaining", SZ, [C.c_void_p]), ("sdfvh_lm_pass_point", None, [C.c_void_p, SZ, C.POINTER(SZ)]),
    ("sdfvh_sdf_free", None, [C.c_void_p]), ("sdfvh_sdf_id", C.c_uint32, [C.c_void_p]),
    ("sdfvh_sdf_name", SZ, [C.c_void_p, C.c_char_p, SZ]), ("sdfvh_sdf_n_children", SZ, [C.c_void_p]),
    ("sdfvh_sdf_child", C.c_void_p, [C.c_void_p, SZ]), ("sdfvh_sdf_bounding_box", None, [C.c_void_p, C.c_void_p]),
    ("sdfvh_sdf_device_params", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("sdfvh_sdf_sample", None, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("sdfvh_sdf_sample_batch", None, [C.c_void_p, C.c_void_p, SZ, C.c_int, C.c_void_p]),
    ("sdfvh_sdf_normal", None, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    ("sdfvh_sdf_normal_default", None, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    ("sdfvh_sdf_set_parameter", C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_char_p, SZ]),
    ("sdfvh_sdf_changed", C.c_int, [C.c_void_p, C.c_void_p]), ("sdfvh_sdf_parameters", SZ, [C.c_void_p, C.c_char_p, SZ]),
    ("sdfvh_viewer_from_bb", C.c_void_p, [C.c_void_p, SZ, SZ]),
    ("sdfvh_viewer_new_voxels", C.c_void_p, [SZ, SZ, SZ, C.c_void_p, SZ]),
    ("sdfvh_viewer_new_voxels_layout", C.c_void_p, [SZ, SZ, SZ, C.c_void_p, SZ, C.c_int]), ("sdfvh_viewer_free", None, [C.c_void_p]),
    ("sdfvh_viewer_dims", None, [C.c_void_p, C.c_void_p]),
    ("sdfvh_viewer_update", SZ, [C.c_void_p, C.c_void_p, C.c_double]), ("sdfvh_viewer_commit", None, [C.c_void_p]),
    ("sdfvh_viewer_lod", C.c_float, [C.c_void_p]), ("sdfvh_viewer_remaining", SZ, [C.c_void_p]),
    ("sdfvh_viewer_passes_left", SZ, [C.c_void_p]), ("sdfvh_viewer_has_changed_box", C.c_int, [C.c_void_p]),
    ("sdfvh_viewer_download", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("sdfvh_viewer_tex0", C.c_void_p, [C.c_void_p]), ("sdfvh_viewer_tex1", C.c_void_p, [C.c_void_p]),
    ("sdfvh_viewer_render", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("sdfvh_viewer_render_device", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("sdfvh_viewer_sync", C.c_int, [C.c_void_p]), ("sdfvh_viewer_pairs_valid", C.c_int, [C.c_void_p]),
    ("sdfvh_format_f32", SZ, [C.c_float, C.c_char_p, SZ]), ("sdfvh_ply_color_u8", C.c_uint32, [C.c_float]),
    ("sdfvh_mesh_sdf", C.c_void_p, [C.c_void_p, C.c_char_p, SZ, C.c_int, C.c_char_p, SZ]),
    ("sdfvh_mesh_from_arrays", C.c_void_p, [C.c_void_p, SZ, C.c_void_p, SZ]), ("sdfvh_mesh_free", None, [C.c_void_p]),
    ("sdfvh_mesh_counts", SZ, [C.c_void_p, C.POINTER(SZ)]), ("sdfvh_mesh_copy", None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("sdfvh_mesh_serialize_ply", SZ, [C.c_void_p, C.c_char_p, C.c_char_p, SZ]),
    ("sdfvh_scene_new", C.c_void_p, [C.c_void_p]), ("sdfvh_scene_free", None, [C.c_void_p]),
    ("sdfvh_scene_advance_clock", None, [C.c_void_p, C.c_longlong]),
    ("sdfvh_scene_set_sdf", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong]),
    ("sdfvh_scene_set_budget_ms", None, [C.c_void_p, C.c_longlong]),
    ("sdfvh_scene_render", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_ulonglong)]),
    ("sdfvh_scene_lod", C.c_float, [C.c_void_p]), ("sdfvh_scene_dims", None, [C.c_void_p, C.c_void_p]),
    ("sdfvh_scene_load_progress", C.c_longlong, [C.c_void_p, C.c_char_p, SZ]),
]:
    fn = getattr(H, name)
    fn.restype = res
    fn.argtypes = args


class LoadingManager:
    def __init__(self, limits, passes):
        self.h = H.sdfvh_lm_new(*limits, passes)

    def __del__(self):
        if H is not None:
            H.sdfvh_lm_free(self.h)

    def next(self):
        out = (SZ * 3)()
        return tuple(out) if H.sdfvh_lm_next(self.h, out) else None

    def len(self):
        return H.sdfvh_lm_len(self.h)

    def total_iterations(self):
        return H.sdfvh_lm_total_iterations(self.h)

    def passes_left(self):
        return H.sdfvh_lm_passes_left(self.h)

    def step_size(self):
        return H.sdfvh_lm_step_size(self.h)

    def finish_pass(self):
        return H.sdfvh_lm_finish_pass(self.h)

    def advance(self, n):
        H.sdfvh_lm_advance(self.h, n)

    def cursor(self):
        return H.sdfvh_lm_cursor(self.h)

    def pass_remaining(self):
        return H.sdfvh_lm_pass_remaining(self.h)

    def pass_point(self, k):
        out = (SZ * 3)()
        H.sdfvh_lm_pass_point(self.h, k, out)
        return tuple(out)


class SDF:
    def __init__(self, handle):
        assert handle
        self.h = handle

    def __del__(self):
        if H is not None:  # interpreter shutdown clears module globals first
            H.sdfvh_sdf_free(self.h)

    @staticmethod
    def demo(*args):
        argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
        err = C.create_string_buffer(512)
        h = H.sdfvh_demo_new(len(args), argv, err, 512)
        if not h:
            raise ValueError(err.value.decode())
        return SDF(h)

    @staticmethod
    def provider(path):
        """ProviderSDF::load: an SDF behind the per-point ABI of include/sdf_provider.h (host-sampled only)."""
        err = C.create_string_buffer(512)
        h = H.sdfvh_provider_load(str(path).encode(), err, 512)
        if not h:
            raise OSError(err.value.decode())
        return SDF(h)

    def sample_concurrency(self):
        return H.sdfvh_sdf_sample_concurrency(self.h)

    def id(self):
        return H.sdfvh_sdf_id(self.h)

    def name(self):
        b = C.create_string_buffer(128)
        H.sdfvh_sdf_name(self.h, b, 128)
        return b.value.decode()

    def children(self):
        return [SDF(H.sdfvh_sdf_child(self.h, i)) for i in range(H.sdfvh_sdf_n_children(self.h))]

    def bounding_box(self):
        out = np.zeros(6, np.float32)
        H.sdfvh_sdf_bounding_box(self.h, out.ctypes.data)
        return out

    def device_params(self, params_type):
        p = params_type()
        sid = C.c_uint32()
        assert H.sdfvh_sdf_device_params(self.h, C.byref(p), C.byref(sid)) == 0
        return p, sid.value

    def sample(self, p, distance_only=False):
        p = np.asarray(p, np.float32)
        out = np.zeros(7, np.float32)
        H.sdfvh_sdf_sample(self.h, p.ctypes.data, int(distance_only), out.ctypes.data)
        return out

    def sample_batch(self, points, distance_only=False):
        """SDFSurface::sample_batch: [n, 3] points -> [n, 7] samples."""
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        out = np.zeros((len(p), 7), np.float32)
        H.sdfvh_sdf_sample_batch(self.h, p.ctypes.data, len(p), int(distance_only), out.ctypes.data)
        return out

    def normal(self, p, eps=0.0, default=False):
        p = np.asarray(p, np.float32)
        out = np.zeros(3, np.float32)
        (H.sdfvh_sdf_normal_default if default else H.sdfvh_sdf_normal)(self.h, p.ctypes.data, eps, out.ctypes.data)
        return out

    def set_parameter(self, param_id, value):
        err = C.create_string_buffer(512)
        if isinstance(value, bool):
            rc = H.sdfvh_sdf_set_parameter(self.h, param_id, 0, int(value), 0.0, None, err, 512)
        elif isinstance(value, int):
            rc = H.sdfvh_sdf_set_parameter(self.h, param_id, 1, value, 0.0, None, err, 512)
        elif isinstance(value, float):
            rc = H.sdfvh_sdf_set_parameter(self.h, param_id, 2, 0, value, None, err, 512)
        else:
            rc = H.sdfvh_sdf_set_parameter(self.h, param_id, 3, 0, 0.0, str(value).encode(), err, 512)
        return None if rc == 0 else err.value.decode()

    def changed(self):
        out = np.zeros(6, np.float32)
        return out if H.sdfvh_sdf_changed(self.h, out.ctypes.data) else None

    def parameters(self):
        b = C.create_string_buffer(4096)
        H.sdfvh_sdf_parameters(self.h, b, 4096)
        return [line.split("|") for line in b.value.decode().strip().split("\n") if line]


class Viewer:
    def __init__(self, handle):
        assert handle, "SDFViewer creation failed (no GPU?)"
        self.h = handle

    def __del__(self):
        if H is not None:
            H.sdfvh_viewer_free(self.h)

    @staticmethod
    def from_bb(bb, max_voxels_side, passes):
        bb = np.asarray(bb, np.float32).reshape(6)
        return Viewer(H.sdfvh_viewer_from_bb(bb.ctypes.data, max_voxels_side, passes))

    @staticmethod
    def new_voxels(dims, bb, passes, layout="auto"):
        """layout: how the distance volume is laid out -- "auto" (the march's preference), "plain", "interleaved"."""
        bb = np.asarray(bb, np.float32).reshape(6)
        return Viewer(H.sdfvh_viewer_new_voxels_layout(dims[0], dims[1], dims[2], bb.ctypes.data, passes,
                                                       {"auto": 0, "plain": 1, "interleaved": 2}[layout]))

    def texture_gap(self):
        """Bytes between the end of tex0 and the start of tex1 (the placement the constructor used)."""
        w, h, d = self.dims()
        return H.sdfvh_viewer_tex1(self.h) - H.sdfvh_viewer_tex0(self.h) - w * h * d * 16

    def dims(self):
        out = (C.c_uint32 * 3)()
        H.sdfvh_viewer_dims(self.h, out)
        return tuple(out)

    def update(self, sdf, max_delta_seconds):
        return H.sdfvh_viewer_update(self.h, sdf.h, max_delta_seconds)

    def set_ingest(self, host_threads=0, capacity=0):
        """Knobs of the ingest path (host-sampled SDFs): worker threads (0 = what the SDF allows), records per buffer."""
        H.sdfvh_viewer_set_ingest(self.h, host_threads, capacity)

    def last_error(self):
        b = C.create_string_buffer(512)
        H.sdfvh_viewer_last_error(self.h, b, 512)
        return b.value.decode()

    def commit(self):
        H.sdfvh_viewer_commit(self.h)

    def lod(self):
        return H.sdfvh_viewer_lod(self.h)

    def remaining(self):
        return H.sdfvh_viewer_remaining(self.h)

    def has_changed_box(self):
        return bool(H.sdfvh_viewer_has_changed_box(self.h))

    def download(self):
        w, h, d = self.dims()
        t0 = np.empty((d, h, w, 4), np.float32)
        t1 = np.empty_like(t0)
        assert H.sdfvh_viewer_download(self.h, t0.ctypes.data, t1.ctypes.data) == 0
        return t0, t1

    def render_device(self, width, height, rgba_ptr, eye=None):
        """Enqueue the frame into a device image (address of width x height x 4 floats); sync() waits for it."""
        e = None if eye is None else np.asarray(eye, np.float32)
        assert H.sdfvh_viewer_render_device(self.h, width, height, None if e is None else e.ctypes.data, rgba_ptr) == 0

    def sync(self):
        assert H.sdfvh_viewer_sync(self.h) == 0

    def pairs_valid(self):
        return bool(H.sdfvh_viewer_pairs_valid(self.h))

    def march_volume(self):
        return {0: "distance", 1: "pairs", 2: "interleaved"}[H.sdfvh_viewer_pairs_valid(self.h)]

    def render(self, width, height, eye=None):
        out = np.empty((height, width, 4), np.float32)
        e = None if eye is None else np.asarray(eye, np.float32)
        assert H.sdfvh_viewer_render(self.h, width, height, None if e is None else e.ctypes.data, out.ctypes.data) == 0
        return out


class Scene:
    """SDFViewerAppScene with a manual clock."""

    def __init__(self, sdf):
        self.h = H.sdfvh_scene_new(sdf.h)
        assert self.h, "SDFViewerAppScene creation failed (no GPU?)"

    def __del__(self):
        if H is not None:
            H.sdfvh_scene_free(self.h)

    def advance_clock(self, ms):
        H.sdfvh_scene_advance_clock(self.h, ms)

    def set_sdf(self, sdf, max_voxels_side=None, loading_passes=None):
        assert H.sdfvh_scene_set_sdf(self.h, sdf.h, -1 if max_voxels_side is None else max_voxels_side,
                                     -1 if loading_passes is None else loading_passes) == 0

    def set_budget_ms(self, ms):
        H.sdfvh_scene_set_budget_ms(self.h, ms)

    def render(self, width=0, height=0, draw=False):
        out = (C.c_ulonglong * 4)()
        img = np.empty((height, width, 4), np.float32) if draw else None
        assert H.sdfvh_scene_render(self.h, width, height, img.ctypes.data if draw else None, out) == 0
        rep = dict(cpu_updates=out[0], committed=bool(out[1]), last_chunk=bool(out[2]), request_repaint=bool(out[3]))
        return (rep, img) if draw else rep

    def lod(self):
        return H.sdfvh_scene_lod(self.h)

    def dims(self):
        out = (C.c_uint32 * 3)()
        H.sdfvh_scene_dims(self.h, out)
        return tuple(out)

    def load_progress(self):
        b = C.create_string_buffer(256)
        p = H.sdfvh_scene_load_progress(self.h, b, 256)
        return None if p < 0 else (p / 1e6, b.value.decode())


class Mesh:
    """sdfviewer::Mesh (host/mesh.hpp = meshers/mesh.rs)."""

    def __init__(self, handle):
        assert handle
        self.h = handle

    def __del__(self):
        if H is not None:
            H.sdfvh_mesh_free(self.h)

    @staticmethod
    def from_sdf(sdf, mesher="marching-cubes", max_voxels_per_axis=64, postproc=True):
        err = C.create_string_buffer(512)
        h = H.sdfvh_mesh_sdf(sdf.h, mesher.encode(), max_voxels_per_axis, int(postproc), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return Mesh(h)

    @staticmethod
    def from_arrays(vertices, indices):
        import numpy as np
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 12)
        i = np.ascontiguousarray(indices, np.uint32)
        return Mesh(H.sdfvh_mesh_from_arrays(v.ctypes.data, len(v), i.ctypes.data, len(i)))

    def arrays(self):
        import numpy as np
        ni = SZ()
        nv = H.sdfvh_mesh_counts(self.h, C.byref(ni))
        v = np.empty((nv, 12), np.float32)
        i = np.empty(ni.value, np.uint32)
        H.sdfvh_mesh_copy(self.h, v.ctypes.data, i.ctypes.data)
        return v, i

    def serialize_ply(self, version_info="test"):
        need = H.sdfvh_mesh_serialize_ply(self.h, version_info.encode(), None, 0)
        buf = C.create_string_buffer(need)
        H.sdfvh_mesh_serialize_ply(self.h, version_info.encode(), buf, need)
        return buf.raw[:need].decode()


def format_f32(v):
    b = C.create_string_buffer(128)
    H.sdfvh_format_f32(v, b, 128)
    return b.value.decode()


# ---- per-point provider (reference ffi.rs ABI) ----
class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class PointerLength(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("len_bytes", C.c_size_t)]


class _KindInt(C.Structure):
    _fields_ = [("range_start", C.c_int32), ("range_end", C.c_int32), ("step", C.c_int32)]


class _KindFloat(C.Structure):
    _fields_ = [("range_start", C.c_float), ("range_end", C.c_float), ("step", C.c_float)]


class _KindUnion(C.Union):
    _fields_ = [("int_", _KindInt), ("float_", _KindFloat), ("choices", PointerLength)]


class ParamKindC(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("v", _KindUnion)]


class _ValueUnion(C.Union):
    _fields_ = [("boolean", C.c_bool), ("int_", C.c_int32), ("float_", C.c_float), ("string_", PointerLength)]


class ParamValueC(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("v", _ValueUnion)]


class ParamC(C.Structure):
    _fields_ = [("id", C.c_uint32), ("name", PointerLength), ("kind", ParamKindC), ("value", ParamValueC),
                ("description", PointerLength)]


class SetParameterResult(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("error", PointerLength)]


class ChangedResult(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("bounds", C.c_float * 6)]


PROVIDER_SYMBOLS = ["init", "bounding_box", "bounding_box_free", "sample", "sample_free", "children", "children_free",
                    "name", "name_free", "parameters", "parameters_free", "set_parameter", "set_parameter_free",
                    "changed", "changed_free", "normal", "normal_free"]


def load_provider():
    P = C.CDLL(PROVIDER_PATH)
    P.init.restype = None
    P.init_with_args.restype = C.c_int
    P.init_with_args.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    P.bounding_box.restype = C.POINTER(C.c_float * 6)
    P.bounding_box.argtypes = [C.c_uint32]
    P.sample.restype = C.POINTER(C.c_float * 7)
    P.sample.argtypes = [C.c_uint32, Vec3, C.c_bool]
    P.children.restype = C.POINTER(PointerLength)
    P.children.argtypes = [C.c_uint32]
    P.name.restype = C.POINTER(PointerLength)
    P.name.argtypes = [C.c_uint32]
    P.parameters.restype = C.POINTER(PointerLength)
    P.parameters.argtypes = [C.c_uint32]
    P.set_parameter.restype = C.POINTER(SetParameterResult)
    P.set_parameter.argtypes = [C.c_uint32, C.c_uint32, ParamValueC]
    P.changed.restype = C.POINTER(ChangedResult)
    P.changed.argtypes = [C.c_uint32]
    P.normal.restype = C.POINTER(C.c_float * 3)
    P.normal.argtypes = [C.c_uint32, Vec3, C.c_float]
    for f in ("bounding_box_free", "sample_free", "children_free", "name_free", "parameters_free",
              "set_parameter_free", "changed_free", "normal_free"):
        getattr(P, f).restype = None
        getattr(P, f).argtypes = [C.c_void_p]
    return P


def pl_bytes(pl):
    return C.string_at(pl.ptr, pl.len_bytes) if pl.ptr else b""
