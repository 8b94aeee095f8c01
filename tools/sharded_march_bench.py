#!/usr/bin/env python3
"""The sharded march with no host in the loop (sdfv_raymarch_slab_round: ray buffers with in-band counts, what sdfv_slab_march
enqueues per rank), all ranks of an 8-slab grid on ONE GPU, whole-buffer device copies in place of the RCCL send / receive:
ms per 1080p frame, eager and as a replayed HIP graph, for a few ray-list capacities, next to the round-1 host-driven loop
(parallel.ShardedMarch: counter read-back every round) and to the single-GPU march over the whole grid.
python tools/sharded_march_bench.py [side=256] [world=8]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
pkg = importlib.import_module("sdf-viewer_amd")
par = importlib.import_module("sdf-viewer_amd.parallel")
import test_gpu_sharded_march as T
side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
dims, bb = (side,) * 3, ((-1, -1, -1), (1, 1, 1))
prm = pkg.default_params()
full = pkg.make_grid(dims)
f0, f1 = pkg.alloc_textures(full)
pkg.fill_grid(prm, full, f0, f1)
rp = pkg.default_render_params(full)
cam = pkg.camera_look_at(aspect=W / H)
want = pkg.raymarch(rp, f0, f1, cam, W, H)
slabs, grids = T.build_slabs(pkg, par, prm, dims, world, bb)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / n * 1e3, 4)
res = {"side": side, "world": world, "image": [W, H]}
res["single_gpu_march_tex0_ms"] = timed(lambda: pkg.raymarch(rp, f0, f1, cam, W, H))
res["host_driven_rounds_ms"] = timed(lambda: T.run_lockstep(pkg, par, rp, slabs, grids, cam, W, H, want_aux=False), n=3)
res["rays_handed_between_ranks"] = T.run_lockstep(pkg, par, rp, slabs, grids, cam, W, H, want_aux=False)[2]
for cap in (W * H, W * H // 8, W * H // 32):
    enqueue, rgba, aux, out_down, out_up, overflow = T.run_lockstep_inband(pkg, rp, slabs, grids, cam, W, H, cap, want_aux=False)
    eager = timed(enqueue)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        enqueue()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        enqueue()
    replay = timed(g.replay)
    got, _ = T.merged(pkg, par, rgba, aux)
    res[f"inband_capacity_{cap}"] = {"eager_ms": eager, "graph_replay_ms": replay, "overflow": int(overflow.sum()),
                                     "bit_identical_to_single_gpu": bool(torch.equal(got.view(torch.int32), want[0].view(torch.int32))),
                                     "buffer_bytes_per_neighbour_and_round": 16 + 24 * cap}
    del g
res["note"] = ("all ranks share one GPU and run one after the other, so these are SUMS over ranks of what each rank would do "
               "concurrently on its own GPU; the copies stand in for xGMI transfers (buffer bytes / ~150 GB/s per direction on a real link)")
print(json.dumps(res))
