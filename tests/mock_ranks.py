#!/usr/bin/env python3
"""The PYTHON layer's multi-rank code (parallel.SlabComm / SlabFiller(transport="rccl") / raymarch_sharded / gather_bands /
gather_images / gather_replica over the library communicator) between DIFFERENT ranks on one GPU: the ranks are threads of this
process, the library loads tests/c/mock_rccl.cpp in RCCL's place (SDFV_OPT_RCCL_LIBRARY), the 128-byte id travels by a Python
variable (SlabComm(ident=...): no torch.distributed anywhere).  Everything a rank ends up with is compared bit for bit with the
single-device result of the same library.  Run by tests/test_gpu_multirank.py in a process of its own (RCCL is loaded once per
process, and the other tests of the suite load the real one):   python tests/mock_ranks.py <mock librccl path> <world>"""
import ctypes as C
import importlib
import os
import sys
import threading
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("sdf-viewer_amd")
par = importlib.import_module("sdf-viewer_amd.parallel")
K = pkg._capi
mock_path, world = sys.argv[1], int(sys.argv[2])
path_buf = C.create_string_buffer(mock_path.encode())
pkg.set_option(K.OPT_RCCL_LIBRARY, C.addressof(path_buf))

dims = (64, 48, 41)
W, H, n_cam = 160, 88, 5
prm = pkg.default_params()
full = pkg.make_grid(dims)
f0, f1 = pkg.alloc_textures(full)
fd = torch.empty(dims[::-1], dtype=torch.float32, device="cuda")
pkg.fill_grid(prm, full, f0, f1, dist=fd)
rp = pkg.default_render_params(full)
cam = pkg.camera_look_at(eye=(1.5, 2.0, 3.5), aspect=W / H)
cams = pkg.orbit_cameras(n_cam, aspect=W / H)
want_frame, want_aux = pkg.raymarch(rp, f0, f1, cam, W, H, want_aux=True)
want_batch = pkg.raymarch(rp, f0, f1, cams, W, H)
torch.cuda.synchronize()
assert int((want_frame[0][..., 3] > 0).sum()) > 500
ident = par.SlabComm.unique_id(pkg)
errors, notes = [], []
lock = threading.Lock()


def same(a, b):
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


def rank_main(rank):
    try:
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            comm = par.SlabComm(pkg, rank, world, ident=ident, halo_hi=2)
            assert comm.rccl_ranks == (rank, world)
            slab = par.alloc_slab(dims, rank, world, "cuda", halo_hi=2)
            assert (slab.ghost_lo, slab.ghost_hi) == (comm.ghost_lo, comm.ghost_hi)
            grid = pkg.make_grid(dims, z_begin=slab.z_begin, z_end=slab.z_end)
            vol = torch.empty(tuple(slab.tex0.shape[:3]), dtype=torch.float32, device="cuda")
            for t in (slab.tex0, slab.tex1, vol):
                t.fill_(-7.0)
            filler = par.SlabFiller(pkg, prm, dims, slab, rank, world, transport="rccl", comm=comm, dist=vol)
            for _ in range(3):
                filler.step()
            st.synchronize()
            lo, hi = slab.z_begin - slab.ghost_lo, slab.z_end + slab.ghost_hi
            assert same(slab.tex0, f0[lo:hi]) and same(slab.tex1, f1[lo:hi]) and same(vol, fd[lo:hi]), "SlabFiller.step over the library communicator"
            # the march where the grid lies; no process group: the ray lists hold every pixel
            got = par.raymarch_sharded(pkg, rp, grid, slab, cam, W, H, rank, world, comm=comm)
            st.synchronize()
            assert same(got, want_frame[0]), "raymarch_sharded over the library communicator"
            rgba, aux, status = comm.march(rp, grid, slab, cam, W, H, want_aux=True, capacity=W * H)
            st.synchronize()
            assert status.tolist() == [0, 0] and same(rgba, want_frame[0])
            ga, wa = aux.cpu().numpy(), want_aux[0].cpu().numpy()  # (normals stay 0 in the sharded record, depth of unreported pixels 1)
            assert (ga[..., :14] == wa[..., :14]).all() and (ga[..., 17] == wa[..., 17]).all(), "merged aux record"
            # replicas, then config 5's two splits with their gathers, to rank 0 and to the last rank
            r0, r1 = par.gather_replica(slab, dims, world, comm=comm)
            st.synchronize()
            assert same(r0, f0) and same(r1, f1), "gather_replica over the library communicator"
            g0, g1, gd = comm.allgather_slabs(slab, dims, dist=vol)
            st.synchronize()
            assert same(gd, fd) and same(g0, f0)
            for dst in (0, world - 1):
                mine = [cams[i] for i in par.split_cameras(n_cam, rank, world)]
                part = pkg.raymarch(rp, r0, r1, mine, W, H) if mine else torch.empty((0, H, W, 4), dtype=torch.float32, device="cuda")
                out = par.gather_images(part, n_cam, rank, world, dst=dst, comm=comm)
                st.synchronize()
                assert (out is not None) == (rank == dst)
                if rank == dst:
                    assert same(out, want_batch), "gather_images (whole cameras)"
                for bh in (8, 16):
                    bands = par.split_bands(H, rank, world, band_height=bh)
                    bpart = pkg.raymarch(rp, r0, r1, cams, W, H, bands=bands, dist=gd)
                    out = par.gather_bands(bpart, H, rank, world, dst=dst, comm=comm, band_height=bh)
                    st.synchronize()
                    if rank == dst:
                        assert same(out, want_batch), f"gather_bands ({bh}-row bands)"
                # the 8-bit UNORM plane (sdfv_march_desc.rgba8): a quarter of the bytes through the same collectives
                want8 = want_batch.clamp(0.0, 1.0).mul(255.0).round().to(torch.uint8)
                bands = par.split_bands(H, rank, world, band_height=8)
                b8 = pkg.raymarch(rp, r0, r1, cams, W, H, bands=bands, dist=gd, rgba8="only")
                out = par.gather_bands(b8, H, rank, world, dst=dst, comm=comm, band_height=8)
                st.synchronize()
                if rank == dst:
                    assert out.dtype == torch.uint8 and torch.equal(out, want8), "gather_bands of the rgba8 plane"
                p8 = pkg.raymarch(rp, r0, r1, mine, W, H, rgba8="only") if mine else torch.empty((0, H, W, 4), dtype=torch.uint8, device="cuda")
                out = par.gather_images(p8, n_cam, rank, world, dst=dst, comm=comm)
                st.synchronize()
                if rank == dst:
                    assert torch.equal(out, want8), "gather_images of the rgba8 plane"
            st.synchronize()
            comm.close()
        with lock:
            notes.append(f"rank {rank}: slab z {slab.z_begin}..{slab.z_end}, ghosts {slab.ghost_lo}/{slab.ghost_hi}")
    except BaseException as e:  # noqa: BLE001
        with lock:
            errors.append(f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}")


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join()
if errors:
    print("FAILED\n" + "\n".join(errors))
    sys.exit(1)
print(f"ok {world} ranks as threads over the mock RCCL through parallel.py: " + "; ".join(sorted(notes)))
