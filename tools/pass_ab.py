#!/usr/bin/env python3
"""A/B of the progressive pass kernels between two builds of libsdfgrid in ONE process on one box (box-to-box spread is
+-8 %): the current library against sdf-viewer_amd/libsdfgrid_prev.so (or argv[1]), alternating rounds, through the C ABI
both export (sdfv_fill_grid_pass_dist).  python tools/pass_ab.py [prev.so] [side]"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
prev_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid_prev.so")
side = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prev = C.CDLL(prev_path)
for name in ("sdfv_fill_grid_pass_ex", "sdfv_grid_init", "sdfv_fill_grid_commit"):
    getattr(prev, name).restype = C.c_int
    getattr(prev, name).argtypes = K.PROTOTYPES[name][1]
libs = {"new": pkg.lib, "prev": prev}
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def pass_(lib, step, box=None):
    b = None if box is None else (C.c_float * 6)(*box)
    assert lib.sdfv_fill_grid_pass_ex(C.byref(prm), 0, C.byref(g), step, b, P(t0), P(t1), P(dist), 0, st) == 0
def fresh():
    pkg.grid_init(g, t0, t1); dist.fill_(pkg.AIR_DIST)
def loaded():
    pkg.fill_grid(prm, g, t0, t1, dist=dist)
def timed(fn, setup, reps=9):
    ts = []
    for _ in range(reps):
        setup(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
eighth = (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5)
def after2(lib):
    def f():
        fresh(); pass_(lib, 2)
    return f
cases = {"fresh_step_1": (lambda lib: (lambda: pass_(lib, 1), fresh)),
         "fresh_step_2": (lambda lib: (lambda: pass_(lib, 2), fresh)),
         "fresh_step_4": (lambda lib: (lambda: pass_(lib, 4), fresh)),
         "step_1_after_step_2": (lambda lib: (lambda: pass_(lib, 1), after2(lib))),
         "noop_step_1": (lambda lib: (lambda: pass_(lib, 1), loaded)),
         "noop_step_2": (lambda lib: (lambda: pass_(lib, 2), loaded)),
         "box8_step_1": (lambda lib: (lambda: pass_(lib, 1, eighth), loaded)),
         "box8_step_2": (lambda lib: (lambda: pass_(lib, 2, eighth), loaded)),
         "box8_step_4": (lambda lib: (lambda: pass_(lib, 4, eighth), loaded)),
         "dense_fused": (lambda lib: (lambda: lib.sdfv_fill_grid_commit(C.byref(prm), 0, C.byref(g), P(t0), P(t1), P(dist), st), (lambda: None)))}
# same texels from both builds: a fresh 3-pass load, then a boxed edit with other parameters
states = {}
for label, lib in libs.items():
    fresh()
    for stp in (4, 2, 1):
        pass_(lib, stp)
    keep = (t0.clone(), t1.clone(), dist.clone())
    prm2 = pkg.default_params(sphere_radius=0.8, cube_material=1)
    prm_saved, prm = prm, prm2
    for stp in (4, 2, 1):
        pass_(lib, stp, eighth)
    prm = prm_saved
    states[label] = keep + (t0.clone(), t1.clone(), dist.clone())
torch.cuda.synchronize()
assert all(torch.equal(x, y) for x, y in zip(states["new"], states["prev"])), "the two builds disagree"
del states
res = {}
for name, mk in cases.items():
    r = {"new": [], "prev": []}
    for rnd in range(3):
        for label, lib in libs.items():
            fn, setup = mk(lib)
            r[label].append(timed(fn, setup))
    res[name] = {k: round(min(v), 4) for k, v in r.items()}
    print(f"{side}^3 {name:22s} new {res[name]['new']:.4f}  prev {res[name]['prev']:.4f}  ratio {res[name]['new'] / res[name]['prev']:.3f}", file=sys.stderr, flush=True)
print(json.dumps({"side": side, "ms": res}))
