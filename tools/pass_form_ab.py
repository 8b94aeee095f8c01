#!/usr/bin/env python3
"""A/B of the unflagged strided passes inside ONE build: SDFV_OPT_PASS_FORM 0 (auto: the whole-rows kernel whose waves decide on
the volume they read) against 1 (the per-voxel kernel of round 4), alternating rounds, both volume layouts, states compared bit
for bit.  python tools/pass_form_ab.py [side=256]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("sdf-viewer_amd"); K = pkg._capi
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prm = pkg.default_params(); g = pkg.make_grid((side,) * 3)
t0, t1 = pkg.alloc_textures_placed(g); dist = torch.empty((side,) * 3, dtype=torch.float32, device="cuda")
FORMS = {"adaptive": 0, "per_voxel": 1}


def timed(fn, setup, reps=9):
    ts = []
    for _ in range(reps):
        setup(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


out = {"side": side, "layouts": {}}
for layout, lflag in (("plain", 0), ("interleaved", K.PASS_VOLUME_INTERLEAVED)):
    def pass_(step, flags=0):
        pkg.fill_grid_pass(prm, g, step, t0, t1, dist=dist, flags=flags | lflag)

    def fresh():
        pkg.grid_init(g, t0, t1); dist.fill_(pkg.AIR_DIST)

    def loaded():
        pkg.fill_grid_pass(prm, g, 1, t0, t1, dist=dist, flags=K.PASS_VIRGIN_GRID | lflag)

    def after(step):
        def f():
            fresh(); pass_(step)
        return f

    cases = {"fresh_step_2": (lambda: pass_(2), fresh), "fresh_step_4": (lambda: pass_(4), fresh), "fresh_step_8": (lambda: pass_(8), fresh),
             "step_2_after_step_4": (lambda: pass_(2), after(4)), "noop_step_2": (lambda: pass_(2), loaded),
             "noop_step_4": (lambda: pass_(4), loaded),
             "fresh_load_2_1": (lambda: (pass_(2), pass_(1)), fresh),
             "flagged_fresh_step_2 (reference point)": (lambda: pass_(2, K.PASS_FRESH_GRID | K.PASS_SAME_LOAD), fresh)}
    states = {}
    for label, form in FORMS.items():
        with pkg.options({K.OPT_PASS_FORM: form}):
            fresh()
            snaps = []
            for stp in (8, 4, 2, 1):
                pass_(stp)
                snaps.append(int(t0.view(torch.int32).sum(dtype=torch.int64).item()) ^ int(t1.view(torch.int32).sum(dtype=torch.int64).item()) ^ int(dist.view(torch.int32).sum(dtype=torch.int64).item()))
            states[label] = (snaps, t0.clone(), dist.clone())
    assert states["adaptive"][0] == states["per_voxel"][0] and torch.equal(states["adaptive"][1], states["per_voxel"][1]) and \
        torch.equal(states["adaptive"][2], states["per_voxel"][2]), "the two forms disagree"
    del states
    res = {}
    for name, (fn, setup) in cases.items():
        r = {k: [] for k in FORMS}
        for rnd in range(3):
            for label, form in FORMS.items():
                with pkg.options({K.OPT_PASS_FORM: form}):
                    r[label].append(timed(fn, setup))
        res[name] = {k: round(min(v), 4) for k, v in r.items()}
        print(f"{side}^3 {layout:11s} {name:40s} adaptive {res[name]['adaptive']:.4f}  per-voxel {res[name]['per_voxel']:.4f}  "
              f"ratio {res[name]['adaptive'] / res[name]['per_voxel']:.3f}", file=sys.stderr, flush=True)
    out["layouts"][layout] = res
print(json.dumps(out))
