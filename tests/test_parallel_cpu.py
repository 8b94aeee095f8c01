"""world_size-2 and -3 gloo tests (CPU) of the multi-GPU layout: slab partition, halo exchange, replica gather,
camera split.  Slab contents are produced by the oracle here (no GPU in this container); the exchange code
is the same one bench.py runs over RCCL."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dims, q, halo_hi=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = importlib.import_module("sdf-viewer_amd.parallel")
        import oracle_binding as oracle
        prm = oracle.default_params()
        slab = par.alloc_slab(dims, rank, world, "cpu", fill_value=float("nan"), halo_hi=halo_hi)
        o0, o1 = oracle.fill_dense(prm, dims, z0=slab.z_begin, z1=slab.z_end, threads=1)
        slab.owned0.copy_(torch.from_numpy(o0))
        slab.owned1.copy_(torch.from_numpy(o1))
        sent = par.halo_exchange(slab, rank, world)
        # ghosts must equal what a local recompute of the neighbour's slice gives (the SDF is analytic)
        ok = True
        if slab.ghost_lo:
            g0, g1 = oracle.fill_dense(prm, dims, z0=slab.z_begin - 1, z1=slab.z_begin, threads=1)
            ok &= np.array_equal(slab.tex0[0].numpy().view(np.uint32), g0[0].view(np.uint32))
            ok &= np.array_equal(slab.tex1[0].numpy().view(np.uint32), g1[0].view(np.uint32))
        if slab.ghost_hi:
            assert slab.ghost_hi == halo_hi
            g0, g1 = oracle.fill_dense(prm, dims, z0=slab.z_end, z1=slab.z_end + slab.ghost_hi, threads=1)
            ok &= np.array_equal(slab.tex0[-slab.ghost_hi:].numpy().view(np.uint32), g0.view(np.uint32))
            ok &= np.array_equal(slab.tex1[-slab.ghost_hi:].numpy().view(np.uint32), g1.view(np.uint32))
        down = halo_hi if rank > 0 else 0        # slices sent to the lower neighbour, per texture
        up = 1 if rank < world - 1 else 0
        expect_sent = (down + up) * 2 * dims[0] * dims[1] * 16
        ok &= sent == expect_sent
        # owned region untouched by the exchange
        ok &= np.array_equal(slab.owned0.numpy().view(np.uint32), o0.view(np.uint32))
        # replica gather == dense fill
        r0, r1 = par.gather_replica(slab, dims, world)
        d0, d1 = oracle.fill_dense(prm, dims, threads=1)
        ok &= np.array_equal(r0.numpy().view(np.uint32), d0.view(np.uint32))
        ok &= np.array_equal(r1.numpy().view(np.uint32), d1.view(np.uint32))
        # camera split + image gather (config 5): rank r's cameras carry their global index as pixel value
        n_cams = 7
        mine = list(par.split_cameras(n_cams, rank, world))
        imgs = torch.stack([torch.full((3, 5, 4), float(c)) for c in mine]) if mine else torch.zeros((0, 3, 5, 4))
        got = par.gather_images(imgs, n_cams, rank, world, dst=0)
        if rank == 0:
            ok &= got.shape == (n_cams, 3, 5, 4) and all(bool((got[c] == float(c)).all()) for c in range(n_cams))
        else:
            ok &= got is None
        # image-tile split (config 5): rank r's band of rows carries r + 1 in every pixel
        Himg = 50
        y0, y1 = par.split_rows(Himg, rank, world)
        band = torch.full((2, y1 - y0, 6, 4), float(rank + 1))
        whole = par.gather_rows(band, Himg, rank, world, dst=0)
        if rank == 0:
            ok &= whole.shape == (2, Himg, 6, 4)
            for r in range(world):
                a, b = par.split_rows(Himg, r, world)
                ok &= bool((whole[:, a:b] == float(r + 1)).all())
        else:
            ok &= whole is None
        # the balanced image-tile split: rank r's 16-row tile bands r, r + world, ...; every pixel carries its image row
        for Himg in (50, 64, 7):
            first, step, bh = par.split_bands(Himg, rank, world)
            rows = par.band_rows(Himg, first, step, bh)
            part = torch.tensor(rows, dtype=torch.float32).reshape(1, -1, 1, 1).expand(2, -1, 6, 4).contiguous()
            whole = par.gather_bands(part, Himg, rank, world, dst=0)
            if rank == 0:
                ok &= whole.shape == (2, Himg, 6, 4)
                ok &= bool((whole == torch.arange(Himg, dtype=torch.float32).reshape(1, -1, 1, 1)).all())
            else:
                ok &= whole is None
        # ray hand-over of the sharded march: lists of data-dependent length to both neighbours, some empty.
        # Rank r sends r + 1 rays down and 2 * r rays up (rank 0 none up); every word carries (sender, direction, k).
        def rays(n, direction):
            t = torch.zeros((n, 6), dtype=torch.int32)
            t[:, 0], t[:, 1], t[:, 2] = rank, direction, torch.arange(n, dtype=torch.int32)
            return t
        inc = par._exchange_rays(rays(rank + 1 if rank > 0 else 0, 0), rays(2 * rank if rank < world - 1 else 0, 1),
                                 rank, world, None)
        want = []
        if rank > 0 and 2 * (rank - 1) > 0:                     # what the lower neighbour sent up
            want += [(rank - 1, 1, k) for k in range(2 * (rank - 1))]
        if rank < world - 1:                                     # what the upper neighbour sent down
            want += [(rank + 1, 0, k) for k in range(rank + 2)]
        ok &= sorted(map(tuple, inc[:, :3].tolist())) == sorted(want) and inc.shape == (len(want), 6)
        q.put((rank, bool(ok), slab.z_begin, slab.z_end))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,halo_hi", [(2, (12, 10, 16), 1), (3, (8, 6, 11), 1), (3, (8, 6, 11), 2)])
def test_halo_exchange_and_gather_gloo(world, dims, halo_hi):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world + 10 * halo_hi
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, q, halo_hi)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results), results
    # slabs tile [0, D) exactly
    assert results[0][2] == 0 and results[-1][3] == dims[2]
    assert all(a[3] == b[2] for a, b in zip(results[:-1], results[1:]))


def test_partition_helpers():
    par = importlib.import_module("sdf-viewer_amd.parallel")
    for depth in (1, 7, 64, 1024):
        for world in (1, 2, 3, 8):
            rs = [par.slab_range(depth, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == depth
            assert all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:]))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    assert par.weak_scaling_dims(256, 1) == par.weak_scaling_dims(256, 1, "cube") == (256, 256, 256)
    assert par.weak_scaling_dims(256, 8) == (256, 256, 2048)             # every rank fills the N=1 slab
    assert par.slab_range(2048, 3, 8) == (768, 1024)
    assert par.weak_scaling_dims(256, 2, "cube") == (256, 256, 512)
    assert par.weak_scaling_dims(256, 4, "cube") == (256, 512, 512)
    assert par.weak_scaling_dims(512, 8, "cube") == (1024, 1024, 1024)   # BASELINE.json config 4
    for h in (1, 15, 16, 17, 1080, 2160):
        for world in (1, 2, 3, 8):
            bands = [par.split_rows(h, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == h and all(a[1] == b[0] for a, b in zip(bands[:-1], bands[1:]))
            assert all(y0 % 16 == 0 for y0, _ in bands)
    assert par.split_rows(1080, 3, 8) == (400, 544)
    for h in (1, 15, 16, 17, 100, 1080, 2160):
        for world in (1, 2, 3, 8, 80):
            rows = [par.band_rows(h, *par.split_bands(h, r, world)) for r in range(world)]
            assert sorted(sum(rows, [])) == list(range(h))              # a partition of the image's rows
            assert all(r == sorted(r) for r in rows)                     # each set in image order
    assert par.band_rows(1080, 3, 8)[:17] == list(range(48, 64)) + [176] and len(par.band_rows(1080, 3, 8)) == 8 * 16 + 8
    cams = [list(par.split_cameras(64, r, 8)) for r in range(8)]
    assert sum(cams, []) == list(range(64)) and all(len(c) == 8 for c in cams)


def _comm_failure_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("sdf-viewer_amd")
        par = importlib.import_module("sdf-viewer_amd.parallel")
        try:
            par.SlabComm(pkg, rank, world)
            q.put((rank, "created"))
        except pkg.SdfvError as e:
            q.put((rank, f"SdfvError: {e}"))
        dist.barrier()  # every rank is still in step with the others: nobody was left inside a collective
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.is_available(), reason="exercises the failure path that a box without GPUs takes")
def test_slab_comm_creation_fails_on_every_rank_together():
    """SlabComm.__init__ is collective.  Without a GPU sdfv_slab_comm_create fails (no device) -- after the
    id was drawn and broadcast -- and the agreement step must turn that into an exception on EVERY rank, with all ranks
    still able to meet in a barrier afterwards (the old code left the healthy ranks inside a collective)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_failure_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert set(results) == {0, 1}
    assert all(v.startswith("SdfvError") and "sdfv_slab_comm_create" in v for v in results.values()), results
