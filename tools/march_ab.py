#!/usr/bin/env python3
"""The march of two builds of the library in ONE process (box-to-box spread is +-8 %): 12 views x {1080p / 256^3, 4K / 512^3}
over the distance, pair and interleaved volumes + the 64-camera batch, alternating rounds, bits compared.
python tools/march_ab.py sdf-viewer_amd/libsdfgrid_prev.so"""
import ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd")
K = pkg._capi
prev = C.CDLL(os.path.abspath(sys.argv[1]))
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
prev.sdfv_raymarch_ex.restype = C.c_int
prev.sdfv_raymarch_ex.argtypes = [C.POINTER(K.MarchDesc), C.c_void_p]
def call(lib, rp, t0, t1, dist, pairs, cams, W, H, out, ilv=None):
    arr = (K.Camera * len(cams))(*cams)
    d = K.MarchDesc()
    d.size = C.sizeof(d)
    d.rp = C.pointer(rp)
    d.tex0, d.tex1, d.dist, d.pairs, d.ilv = t0.data_ptr(), t1.data_ptr(), p(dist), p(pairs), p(ilv)
    d.cameras, d.n_cameras, d.width, d.height, d.y0, d.y1 = arr, len(cams), W, H, 0, H
    d.rgba = out.data_ptr()
    rc = lib.sdfv_raymarch_ex(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
def run(fn, n=40, warm=0.05):
    t_end = time.perf_counter() + warm
    while time.perf_counter() < t_end: fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
res = {}
prm = pkg.default_params()
for side, W, H in ((256, 1920, 1080), (512, 3840, 2160)):
    g = pkg.make_grid((side,) * 3)
    t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
    pkg.fill_grid(prm, g, t0, t1, dist=dist); pairs = pkg.commit_pairs(g, dist); ilv = pkg.commit_interleaved(g, dist)
    rp = pkg.default_render_params(g)
    out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda"); ref = torch.empty_like(out)
    views = {"default": pkg.camera_look_at(aspect=W / H)}
    for k, c in enumerate(pkg.orbit_cameras(8, aspect=W / H)[1:]): views[f"orbit{k + 1}"] = c
    views["close"] = pkg.camera_look_at(eye=(1.2, 1.5, 2.4), aspect=W / H)
    views["far"] = pkg.camera_look_at(eye=(5.0, 6.0, 10.0), aspect=W / H)
    views["axis"] = pkg.camera_look_at(eye=(0.0, 0.0, 5.0), aspect=W / H)
    views["inside"] = pkg.camera_look_at(eye=(0.2, 0.1, 0.3), target=(1.0, 0.5, -1.0), aspect=W / H)
    for vol, pr in (("dist", None), ("pairs", pairs), ("ilv", None)):
        ratios = []
        iv = ilv if vol == "ilv" else None
        for name, cam in views.items():
            ms = {"new": [], "prev": []}
            for rnd in range(3):
                ms["prev"].append(run(lambda: call(prev, rp, t0, t1, dist, pr, [cam], W, H, ref, iv)))
                ms["new"].append(run(lambda: call(pkg.lib, rp, t0, t1, dist, pr, [cam], W, H, out, iv)))
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            res[f"{side}_{vol}_{name}"] = {"prev_ms": round(min(ms["prev"]), 4), "new_ms": round(min(ms["new"]), 4), "same_bits": same}
            ratios.append(min(ms["new"]) / min(ms["prev"]))
            print(f"{side} {vol:5s} {name:8s} prev {min(ms['prev']):.4f}  new {min(ms['new']):.4f}  ratio {ratios[-1]:.3f}  same bits {same}", file=sys.stderr, flush=True)
        res[f"{side}_{vol}_geomean_ratio"] = round(float(torch.tensor(ratios).log().mean().exp()), 4)
        print(f"{side} {vol} geomean ratio {res[f'{side}_{vol}_geomean_ratio']}", file=sys.stderr, flush=True)
    if side == 256:
        cams = pkg.orbit_cameras(64, aspect=W / H)
        big = torch.empty((64, H, W, 4), dtype=torch.float32, device="cuda"); big2 = torch.empty_like(big)
        for vol, pr in (("dist", None), ("pairs", pairs)):
            a = min(run(lambda: call(prev, rp, t0, t1, dist, pr, cams, W, H, big), n=5) for _ in range(3))
            b = min(run(lambda: call(pkg.lib, rp, t0, t1, dist, pr, cams, W, H, big2), n=5) for _ in range(3))
            res[f"256_batch64_{vol}"] = {"prev_ms": round(a, 4), "new_ms": round(b, 4), "same_bits": bool(torch.equal(big.view(torch.int32), big2.view(torch.int32)))}
            print("batch64", vol, res[f"256_batch64_{vol}"], file=sys.stderr, flush=True)
        del big, big2
    del t0, t1, dist, pairs, ilv
print(json.dumps(res))
