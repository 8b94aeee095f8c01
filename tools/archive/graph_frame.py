#!/usr/bin/env python3
"""Launch-bound case (configs[0] on the GPU: 64^3 grid, 512x512 frame): one frame = fill + commit + raymarch, three
short kernels.  Eager enqueue vs a captured hipGraph replay (the library's enqueue calls are capture-safe: no
allocation, no synchronisation)."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sdf-viewer_amd")


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    W = H = 512
    prm = pkg.default_params()
    g = pkg.make_grid((side, side, side))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((side, side, side), dtype=torch.float32, device="cuda")
    rp = pkg.default_render_params(g)
    cam = pkg.camera_look_at(aspect=W / H)
    out = torch.empty((1, H, W, 4), dtype=torch.float32, device="cuda")

    def frame():
        pkg.fill_grid(prm, g, t0, t1)
        pkg.commit_distance(g, t0, dist=dist)
        pkg.raymarch(rp, t0, t1, cam, W, H, out=out, dist=dist)

    def timeit(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6

    frame()
    torch.cuda.synchronize()
    ref = out.clone()
    eager = timeit(frame)
    side_stream = torch.cuda.Stream()
    with torch.cuda.stream(side_stream):
        frame()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side_stream):
        frame()
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    same = torch.equal(out.view(torch.int32), ref.view(torch.int32))
    replay = timeit(graph.replay)
    print(f"{side}^3 + {W}x{H}: eager {eager:.1f} us/frame, hipGraph replay {replay:.1f} us/frame, identical output: {same}")


if __name__ == "__main__":
    main()
