#!/usr/bin/env python3
"""What does one halo exchange cost on the launch side?  One GPU box, backend nccl (= RCCL), world size 1: every
send is matched by a receive from the same rank in the same group, so the whole c10d + RCCL launch path runs
(no xGMI transfer -- the copy stays in HBM; transfer time has to be added from the link rate).
Prints wall time per exchange for the eager path with a rebuilt and a prebuilt op list."""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_msgs = 4  # tex0/tex1 x lo/hi
    send = [torch.full((side, side, 4), float(i + 1), device="cuda") for i in range(n_msgs)]
    recv = [torch.zeros((side, side, 4), device="cuda") for _ in range(n_msgs)]

    def make_ops():
        ops = []
        for s, r in zip(send, recv):
            ops.append(dist.P2POp(dist.isend, s, 0))
            ops.append(dist.P2POp(dist.irecv, r, 0))
        return ops

    def exchange(ops=None):
        for req in dist.batch_isend_irecv(ops or make_ops()):
            req.wait()

    exchange()
    torch.cuda.synchronize()
    for s, r in zip(send, recv):
        assert torch.equal(s, r)
    print(f"loopback ok; {n_msgs} messages of {send[0].numel() * 4 / 1e6:.2f} MB")

    def timeit(fn, n=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        return t_issue / n * 1e6, t_all / n * 1e6

    print("eager, ops rebuilt each call : issue %.1f us, total %.1f us per exchange" % timeit(exchange))
    ops = make_ops()
    print("eager, prebuilt op list      : issue %.1f us, total %.1f us per exchange" % timeit(lambda: exchange(ops)))

    # single message pair for the floor
    one = [dist.P2POp(dist.isend, send[0], 0), dist.P2POp(dist.irecv, recv[0], 0)]
    print("eager, one send/recv pair    : issue %.1f us, total %.1f us per exchange" % timeit(lambda: exchange(one)))

    # the fill beside it, for scale
    pkg = importlib.import_module("sdf-viewer_amd")
    prm = pkg.default_params()
    g = pkg.make_grid((side, side, side))
    t0_, t1_ = pkg.alloc_textures(g)
    print("fill %d^3                   : issue %.1f us, total %.1f us per fill" % ((side,) + timeit(
        lambda: pkg.fill_grid(prm, g, t0_, t1_))))

    # the library's own communicator: one C call per fill step (boundary fills, exchange on its stream, interior)
    par = importlib.import_module("sdf-viewer_amd.parallel")
    comm = par.SlabComm(pkg, 0, 1, periodic=True)
    slab = par.alloc_slab((side, side, side), 0, 1, "cuda", periodic=True)
    print("sdfv_slab_fill_step %d^3    : issue %.1f us, total %.1f us per step (fill + 4 sends + 4 receives)" % (
        (side,) + timeit(lambda: comm.fill_step(prm, g, slab))))
    print("sdfv_slab_halo_exchange      : issue %.1f us, total %.1f us per exchange" % timeit(
        lambda: comm.halo_exchange(g, slab)))
    filler = par.SlabFiller(pkg, prm, (side, side, side), par.alloc_slab((side, side, side), 0, 1, "cuda"), 0, 1)
    print("SlabFiller.step (world 1)    : issue %.1f us, total %.1f us per step" % timeit(filler.step))
    comm.close()

    # hipGraph capture of the exchange (torch.cuda.graph around batch_isend_irecv) was tried here: capture_end
    # segfaults inside the runtime on this image (ROCm 7.2 / torch 2.10), so the step stays eager.
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
