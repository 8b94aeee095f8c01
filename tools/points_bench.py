#!/usr/bin/env python3
"""Throughput of the batched point kernels (SDFSurface::sample / ::normal over point lists, the mesher's ScalarSource /
HermiteSource, Mesh::postproc) on 2^26 points; with a second build of the library as argument, both in one process and the
bits compared.  python tools/points_bench.py [sdf-viewer_amd/libsdfgrid_prev.so]"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd")
prev = C.CDLL(os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else None
prm = pkg.default_params()
n = 64 * 1024 * 1024
pts = (torch.rand((n, 3), device="cuda") * 2.4 - 1.2).contiguous()
unit = torch.rand((n, 3), device="cuda").contiguous()
verts = torch.zeros((n // 4, 12), device="cuda"); verts[:, :3] = pts[:n // 4]
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
f3 = lambda v: (C.c_float * 3)(*v)
def timed(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
lo, hi = f3((-1, -1, -1)), f3((1, 1, 1))
cases = [
    ("sample(p, false)", n, 40, (n, 7), lambda L, o: L.sdfv_sample_points(C.byref(prm), 0, p(pts), C.c_size_t(n), 0, p(o), st())),
    ("sample(p, true)", n, 40, (n, 7), lambda L, o: L.sdfv_sample_points(C.byref(prm), 0, p(pts), C.c_size_t(n), 1, p(o), st())),
    ("normal(p)", n, 24, (n, 3), lambda L, o: L.sdfv_normal_points(C.byref(prm), 0, p(pts), C.c_size_t(n), C.c_float(0.0), 0, p(o), st())),
    ("normal_default(p, 0.001)", n, 24, (n, 3), lambda L, o: L.sdfv_normal_points(C.byref(prm), 0, p(pts), C.c_size_t(n), C.c_float(0.001), 1, p(o), st())),
    ("source_sample_scalar", n, 16, (n,), lambda L, o: L.sdfv_source_sample_scalar(C.byref(prm), 0, lo, hi, p(unit), C.c_size_t(n), p(o), st())),
    ("source_sample_normal", n, 24, (n, 3), lambda L, o: L.sdfv_source_sample_normal(C.byref(prm), 0, lo, hi, p(unit), C.c_size_t(n), p(o), st())),
]
res = {"points": n}
for name, cnt, bytes_per, shape, call in cases:
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    def run(L, o):
        assert call(L, o) == 0
    ms = timed(lambda: run(pkg.lib, out))
    row = {"ms": round(ms, 4), "GB_s_algorithmic": round(cnt * bytes_per / ms / 1e6), "frac_of_8TBs": round(cnt * bytes_per / ms / 8e9, 3)}
    if prev is not None:
        out2 = torch.empty_like(out)
        row["prev_ms"] = round(timed(lambda: run(prev, out2)), 4)
        row["same_bits"] = bool(torch.equal(out.view(torch.int32), out2.view(torch.int32)))
    res[name] = row
    print(name, row, file=sys.stderr, flush=True)
v2 = verts.clone()
ms = timed(lambda: pkg.lib.sdfv_mesh_postproc(C.byref(prm), 0, p(verts), C.c_size_t(n // 4), st()))
row = {"vertices": n // 4, "ms": round(ms, 4), "GB_s_algorithmic": round((n // 4) * 84 / ms / 1e6), "frac_of_8TBs": round((n // 4) * 84 / ms / 8e9, 3),
       "bytes_per_vertex": "48 read + 36 written"}
if prev is not None:
    row["prev_ms"] = round(timed(lambda: prev.sdfv_mesh_postproc(C.byref(prm), 0, p(v2), C.c_size_t(n // 4), st())), 4)
    row["same_bits"] = bool(torch.equal(verts.view(torch.int32), v2.view(torch.int32)))
res["mesh_postproc"] = row
print("mesh_postproc", row, file=sys.stderr, flush=True)
print(json.dumps(res))
