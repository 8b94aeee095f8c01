"""Multi-GPU layout of the path: z-slab sharding of the grid fill with a one-voxel halo exchange, and
camera/row splitting of the raymarch over replicated grids (the reference has no sharding
of any kind -- its hot loop carries `// TODO: Cross-platform parallel iteration?`, scene/sdf/mod.rs:174).

One process per GPU.  Voxels are independent, so the fill needs NO data-path collective; the only exchange
is the halo: after the fill each rank sends its first/last owned z-slice of both textures to its z-neighbour
and receives the neighbour's into a ghost slice, so trilinear sampling (material.frag:42-45) at a slab
boundary reads local memory.  On GPUs `torch.distributed` backend "nccl" is RCCL: the grouped
isend/irecv below become one ncclGroupStart/End of ncclSend/ncclRecv pairs, each neighbour pair riding one
xGMI link per direction.  On CPU (tests) the same code runs over gloo.

Two transports carry the halo: `SlabComm` (the library's own RCCL communicator, include/sdfgrid.h sdfv_slab_*: one
C call enqueues a whole fill step, ~10 us of host time) and torch.distributed P2P ops (c10d spends ~120 us of
host time per exchange, more than the 256^3 fill takes on the GPU -- EXPERIMENTS, round 2).  `SlabFiller` uses
the first on GPUs with the nccl backend and the second everywhere else (gloo tests).
"""
import ctypes as C
import time
from dataclasses import dataclass

import torch
import torch.distributed as c10d  # not `dist`: that name is the distance volume in this package's signatures


# Every collective stage names itself before it blocks, so that a hang on first contact with real xGMI (a peer that never
# joins, P2P unavailable) is attributable in ONE run: bench.py's watchdog prints current_stage() of the rank that is stuck.
_stage = {"name": "idle", "since": time.monotonic(), "count": 0}


def enter_stage(name):
    _stage["name"], _stage["since"] = name, time.monotonic()
    _stage["count"] += 1


def current_stage():
    """-> (name of the collective stage this rank entered last, seconds since, stages entered so far)"""
    return _stage["name"], time.monotonic() - _stage["since"], _stage["count"]


def slab_range(depth, rank, world):
    """Slices [z_begin, z_end) owned by `rank`: contiguous, balanced to within one slice."""
    return depth * rank // world, depth * (rank + 1) // world


def weak_scaling_dims(side, world, geometry="slab"):
    """Global grid with side^3 voxels per rank.
    "slab": the grid grows along the sharded axis only, (side, side, side*world): every rank fills exactly the
            slab the single-GPU run fills (same launch shape, same rows), and the halo slice stays side^2.
    "cube": doubles z, then y, then x as world doubles (1,2,4,8 -> a cube of 2*side at 8 ranks = BASELINE.json
            config 4 when side = 512); the halo slice grows to (2*side)^2 while the slab gets thinner."""
    if geometry == "slab":
        return (side, side, side * world)
    assert geometry == "cube", geometry
    dims = [side, side, side]
    axis, w = 2, world
    while w > 1:
        assert w % 2 == 0, "weak-scaling layout is defined for power-of-two world sizes"
        dims[axis] *= 2
        axis = (axis - 1) % 3
        w //= 2
    return tuple(dims)


@dataclass
class SlabTextures:
    """tex0/tex1 of one rank's slab plus ghost slices: [ghost_lo?][owned z_begin..z_end)[ghost_hi?]."""
    tex0: torch.Tensor
    tex1: torch.Tensor
    z_begin: int
    z_end: int
    ghost_lo: int  # 1 if a lower neighbour exists
    ghost_hi: int  # upper ghost slices actually present (0 at the top of the grid)
    halo_hi: int = 1  # upper halo depth of the layout: every rank sends this many slices down

    @property
    def owned0(self):
        return self.tex0[self.ghost_lo:self.ghost_lo + (self.z_end - self.z_begin)]

    @property
    def owned1(self):
        return self.tex1[self.ghost_lo:self.ghost_lo + (self.z_end - self.z_begin)]


def alloc_slab(dims, rank, world, device, fill_value=None, periodic=False, pkg=None, halo_hi=1):
    """pkg given (and a GPU device): the two textures share one block, tex1 at the distance from tex0's end that MI355X fills
    fastest for textures of this size (pkg.alloc_textures_placed = what SDFViewer::new_voxels allocates; no probe).
    halo_hi = 2: two ghost slices on the upper side (what sdfNormal's taps need in the sharded march): the library's
    communicator fills them when created with SDFV_COMM_HALO2 (SlabComm(halo_hi=2)), halo_exchange() below handles either
    depth over torch.distributed."""
    z0, z1 = slab_range(dims[2], rank, world)
    glo = 1 if (periodic or rank > 0) else 0
    ghi = halo_hi if periodic else min(halo_hi, dims[2] - z1)
    shape = (glo + (z1 - z0) + ghi, dims[1], dims[0], 4)
    if pkg is not None and torch.device(device).type == "cuda":
        with torch.cuda.device(torch.device(device)):
            t0, t1 = pkg.alloc_textures_placed(pkg.make_grid((dims[0], dims[1], shape[0])), device=device)
    else:
        t0 = torch.empty(shape, dtype=torch.float32, device=device)
        t1 = torch.empty(shape, dtype=torch.float32, device=device)
    if fill_value is not None:
        t0.fill_(fill_value)
        t1.fill_(fill_value)
    return SlabTextures(t0, t1, z0, z1, glo, ghi, halo_hi)


def _needs_host_staging(t, group=None):
    """gloo cannot send/recv device tensors: when the path is exercised over gloo with GPU tensors (two ranks
    sharing the one GPU of a test box) the slices are staged through host memory.  Never taken with nccl/RCCL."""
    return t.is_cuda and c10d.get_backend(group) == "gloo"


def halo_exchange(slab, rank, world, group=None):
    """Halo of both textures with ranks rank-1 / rank+1; non-periodic ends.  Downwards a rank sends its first
    `slab.halo_hi` owned slices (the lower neighbour's upper ghosts: 1 = the one-voxel halo, 2 = what sdfNormal's taps
    need in the sharded march), upwards its last owned slice.  Returns the number of bytes this rank sent."""
    if world == 1:
        return 0
    staged = _needs_host_staging(slab.tex0, group)
    ops, copies = [], []  # copies: (ghost slices, host buffer) to copy back after the wait when staging
    sent = 0
    n_owned = slab.z_end - slab.z_begin
    depth = slab.halo_hi
    assert n_owned >= depth, "a halo deeper than a neighbour's slab would need a second hop"

    def send(block, peer):
        nonlocal sent
        ops.append(c10d.P2POp(c10d.isend, block.cpu() if staged else block, peer, group))
        sent += block.numel() * 4

    def recv(block, peer):
        if staged:
            buf = torch.empty(block.shape, dtype=block.dtype)
            ops.append(c10d.P2POp(c10d.irecv, buf, peer, group))
            copies.append((block, buf))
        else:
            ops.append(c10d.P2POp(c10d.irecv, block, peer, group))

    lo = slab.ghost_lo
    for t in (slab.tex0, slab.tex1):
        # sends down then up, receives from above then from below: messages between a pair match in posting order
        if rank > 0:
            send(t[lo:lo + depth], rank - 1)
        if rank < world - 1:
            send(t[lo + n_owned - 1:lo + n_owned], rank + 1)
        if rank < world - 1:
            recv(t[lo + n_owned:lo + n_owned + slab.ghost_hi], rank + 1)
        if rank > 0:
            recv(t[0:lo], rank - 1)
    enter_stage(f"halo_exchange: batch_isend_irecv of {len(ops)} ops with ranks {rank - 1}/{rank + 1}")
    for req in c10d.batch_isend_irecv(ops):
        req.wait()
    enter_stage("halo_exchange: done")
    for dst, buf in copies:
        dst.copy_(buf)
    return sent


class SlabComm:
    """The library's RCCL communicator for the slab halo (sdfv_slab_comm_*).  Rank 0 draws the id and
    torch.distributed only carries its 128 bytes to the other ranks; after that no torch call is on the step path.
    `periodic` wraps the end ranks around -- with world 1 a rank exchanges with itself (single-GPU tests)."""

    def __init__(self, pkg, rank, world, group=None, periodic=False, halo_hi=1, ident=None):
        """Collective over the ranks.  Every stage that can fail on one rank only (drawing the id, creating the RCCL
        communicator) is followed by an agreement (all-reduce of a flag) so that a failure raises on EVERY rank instead of
        leaving the others inside a collective nobody else joins.
        ident: the 128 bytes of SlabComm.unique_id(), carried to every rank by the HOST's own channel -- then nothing here touches
        torch.distributed (include/sdfgrid.h: "the HOST hands the 128 bytes to every rank by whatever channel it has"): ranks
        that are threads of one process, an MPI host, a test."""
        capi = pkg._capi
        self.pkg, self.rank, self.world, self.periodic, self.halo_hi = pkg, rank, world, periodic, halo_hi
        self.handle = None
        assert halo_hi in (1, 2)
        if ident is not None:
            ident = bytes(ident)
            assert len(ident) == capi.COMM_ID_BYTES
            handle = C.c_void_p()
            flags = (capi.COMM_PERIODIC if periodic else 0) | (capi.COMM_HALO2 if halo_hi == 2 else 0)
            enter_stage(f"SlabComm.__init__: sdfv_slab_comm_create(rank {rank} of {world}) with the host's own id")
            pkg.check(pkg.lib.sdfv_slab_comm_create((C.c_ubyte * capi.COMM_ID_BYTES)(*ident), rank, world, flags, C.byref(handle)))
            self.handle = handle
            return
        on_gpu = world > 1 and c10d.get_backend(group) == "nccl"
        dev = "cuda" if on_gpu else "cpu"

        def agree(ok, what):
            if world > 1:
                enter_stage(f"SlabComm.__init__: agreement all_reduce after {what}")
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                c10d.all_reduce(t, op=c10d.ReduceOp.MIN, group=group)
                everyone = bool(t.item())
            else:
                everyone = ok
            if not everyone:
                raise pkg.SdfvError(-5, f"{what} failed on {'this rank' if not ok else 'another rank'}: "
                                        f"{pkg.lib.sdfv_last_error().decode('utf-8', 'replace') if not ok else ''}")

        ident = (C.c_ubyte * capi.COMM_ID_BYTES)()
        rc = pkg.lib.sdfv_slab_comm_unique_id(ident) if rank == 0 else 0
        agree(rc == 0, "sdfv_slab_comm_unique_id")
        if world > 1:
            enter_stage("SlabComm.__init__: broadcast of the 128-byte RCCL id from rank 0")
            t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
            c10d.broadcast(t, src=c10d.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = (C.c_ubyte * capi.COMM_ID_BYTES)(*t.cpu().tolist())
        handle = C.c_void_p()
        flags = (capi.COMM_PERIODIC if periodic else 0) | (capi.COMM_HALO2 if halo_hi == 2 else 0)
        enter_stage(f"SlabComm.__init__: sdfv_slab_comm_create = ncclCommInitRank(rank {rank} of {world}) on the library's communicator")
        rc = pkg.lib.sdfv_slab_comm_create(ident, rank, world, flags, C.byref(handle))
        if rc == 0:
            self.handle = handle
        try:
            agree(rc == 0, "sdfv_slab_comm_create")
        except Exception:
            self.close()
            raise
        enter_stage("SlabComm.__init__: done")

    @staticmethod
    def unique_id(pkg):
        """sdfv_slab_comm_unique_id: the 128 bytes one rank draws and the host hands to the others (SlabComm(ident=...))."""
        ident = (C.c_ubyte * pkg._capi.COMM_ID_BYTES)()
        pkg.check(pkg.lib.sdfv_slab_comm_unique_id(ident))
        return bytes(ident)

    @property
    def ghost_lo(self):
        return 1 if (self.periodic or self.rank > 0) else 0

    @property
    def ghost_hi(self):
        return self.halo_hi if (self.periodic or self.rank < self.world - 1) else 0

    @property
    def rccl_ranks(self):
        """(rank, world) as RCCL reports them for the library's communicator (ncclCommUserRank, ncclCommCount)."""
        r, n = C.c_int(-1), C.c_int(0)
        self.pkg.check(self.pkg.lib.sdfv_slab_comm_ranks(self.handle, C.byref(r), C.byref(n)))
        return int(r.value), int(n.value)

    @property
    def wait_value_capable(self):
        v = C.c_uint32(0)
        self.pkg.check(self.pkg.lib.sdfv_slab_comm_info(self.handle, None, None, C.byref(v)))
        return bool(v.value)

    def _args(self, grid, slab, stream):
        n_owned = int(grid.z_end) - int(grid.z_begin)
        assert slab.ghost_lo == self.ghost_lo and slab.ghost_hi == self.ghost_hi
        assert slab.tex0.shape[0] == self.ghost_lo + n_owned + self.ghost_hi
        stream = torch.cuda.current_stream() if stream is None else stream
        return (C.byref(grid), C.c_void_p(slab.tex0.data_ptr()), C.c_void_p(slab.tex1.data_ptr()),
                C.c_void_p(stream.cuda_stream))

    def halo_exchange(self, grid, slab, stream=None):
        self.pkg.check(self.pkg.lib.sdfv_slab_halo_exchange(self.handle, *self._args(grid, slab, stream)))

    def fill_step(self, params, grid, slab, sdf_id=0, stream=None, dist=None):
        """dist: optional [slices incl. ghosts, H, W] float32 tensor = the slab's compact distance volume, written in the
        same pass (sdfv_slab_fill_step_commit: the fused fill per rank, ghost slices' share once the halo is in)."""
        g, t0, t1, st = self._args(grid, slab, stream)
        if dist is None:
            self.pkg.check(self.pkg.lib.sdfv_slab_fill_step(self.handle, C.byref(params), sdf_id, g, t0, t1, st))
            return
        assert dist.is_cuda and dist.dtype == torch.float32 and dist.is_contiguous() and \
            tuple(dist.shape) == tuple(slab.tex0.shape[:3])
        self.pkg.check(self.pkg.lib.sdfv_slab_fill_step_commit(self.handle, C.byref(params), sdf_id, g, t0, t1,
                                                               C.c_void_p(dist.data_ptr()), st))

    def march(self, rp, grid, slab, camera, width, height, want_aux=False, capacity=None, merge=True, stream=None):
        """sdfv_slab_march: the whole sharded march of one frame enqueued in one call -- `world` rounds back to back, the ray
        buffers (count in band) exchanged over this communicator, optionally the integer all-reduce that merges the ranks'
        images.  Nothing is read back in between.  -> (rgba [H, W, 4], aux or None, status int32[2] DEVICE tensor:
        [overflow flag, rays left over]); the caller checks status when it synchronises anyway."""
        pkg = self.pkg
        dev = slab.tex0.device
        # A message is 16 + 24 * capacity bytes per neighbour and round WHATEVER it carries (width * height made it
        # 49.8 MB at 1080p).  Default: an eighth of the pixels -- no view measured hands more than a few per cent of its rays
        # across one slab boundary in one round; raymarch_sharded() retries with width * height should status[0] report overflow.
        capacity = max(4096, width * height // 8) if capacity is None else int(capacity)
        n = pkg.lib.sdfv_slab_march_scratch_bytes(capacity)
        stream = torch.cuda.current_stream() if stream is None else stream
        with torch.cuda.stream(stream):  # allocations, the call and the merge on the stream the rounds run on
            scratch = torch.empty(n // 4, dtype=torch.int32, device=dev)
            rgba = torch.empty((height, width, 4), dtype=torch.float32, device=dev)
            aux = torch.empty((height, width, pkg.AUX_FLOATS), dtype=torch.int32, device=dev) if want_aux else None
            status = torch.zeros(2, dtype=torch.int32, device=dev)
            pkg.check(pkg.lib.sdfv_slab_march(self.handle, C.byref(rp), C.byref(grid), C.c_void_p(slab.tex0.data_ptr()),
                                              C.c_void_p(slab.tex1.data_ptr()), C.byref(camera), width, height,
                                              C.c_void_p(rgba.data_ptr()), None if aux is None else C.c_void_p(aux.data_ptr()),
                                              C.c_void_p(scratch.data_ptr()), n, capacity, pkg._capi.MARCH_MERGE if merge else 0,
                                              C.c_void_p(status.data_ptr()), C.c_void_p(stream.cuda_stream)))
            self._march_scratch = scratch  # alive until the stream has run the rounds
            merged_aux = merge_sharded_aux(aux) if (want_aux and merge) else aux
        return rgba, merged_aux, status

    # ---- config 5's collectives over this communicator; every rank calls, nothing synchronises ----
    def gather_bands(self, part, height, dst=0, stream=None, band_height=16):
        """sdfv_comm_gather_bands: part = [n_cam, band rows of this rank, W, C] rendered with bands=(rank, world, band_height) -> on
        `dst` the images [n_cam, height, W, C] (None elsewhere)."""
        pkg = self.pkg
        stream = torch.cuda.current_stream() if stream is None else stream
        if part.dtype == torch.uint8:  # the 8-bit UNORM plane (sdfv_march_desc.rgba8): 4 bytes per pixel = one word per pixel
            out = self.gather_bands(part.view(torch.float32), height, dst=dst, stream=stream, band_height=band_height)
            return None if out is None else out.view(torch.uint8)
        n_cam, _, width, ch = (int(v) for v in part.shape)
        with torch.cuda.stream(stream):
            out = scratch = None
            n = int(pkg.lib.sdfv_comm_gather_bands_scratch_bytes(self.handle, dst, band_height, n_cam, width, height, ch))
            if self.rank == dst:
                out = torch.empty((n_cam, height, width, ch), dtype=torch.float32, device=part.device)
                scratch = torch.empty(max(n // 4, 4), dtype=torch.float32, device=part.device)
            enter_stage("SlabComm.gather_bands: sdfv_comm_gather_bands over the library communicator")
            pkg.check(pkg.lib.sdfv_comm_gather_bands(self.handle, C.c_void_p(part.data_ptr()) if part.numel() else None, band_height, n_cam,
                                                     width, height, ch, dst, None if out is None else C.c_void_p(out.data_ptr()),
                                                     None if scratch is None else C.c_void_p(scratch.data_ptr()), n,
                                                     C.c_void_p(stream.cuda_stream)))
        self._gather_scratch = scratch  # alive until the stream has run the scatter
        return out

    def gather_cameras(self, part, n_cameras, dst=0, stream=None):
        """sdfv_comm_gather_cameras: part = [cameras of this rank, H, W, C] (split_cameras) -> on `dst` [n_cameras, H, W, C]."""
        pkg = self.pkg
        stream = torch.cuda.current_stream() if stream is None else stream
        if part.dtype == torch.uint8:  # the 8-bit UNORM plane: one 4-byte word per pixel
            out = self.gather_cameras(part.view(torch.float32), n_cameras, dst=dst, stream=stream)
            return None if out is None else out.view(torch.uint8)
        _, height, width, ch = (int(v) for v in part.shape)
        with torch.cuda.stream(stream):
            out = torch.empty((n_cameras, height, width, ch), dtype=torch.float32, device=part.device) if self.rank == dst else None
            enter_stage("SlabComm.gather_cameras: sdfv_comm_gather_cameras over the library communicator")
            pkg.check(pkg.lib.sdfv_comm_gather_cameras(self.handle, C.c_void_p(part.data_ptr()) if part.numel() else None, n_cameras,
                                                       width, height, ch, dst, None if out is None else C.c_void_p(out.data_ptr()),
                                                       C.c_void_p(stream.cuda_stream)))
        return out

    def allgather_slabs(self, slab, dims, dist=None, stream=None):
        """sdfv_comm_allgather_slabs: the whole grid on every rank from the ranks' owned slices (slab_range split) ->
        (tex0, tex1[, dist]) of the whole grid.  dist: the slab's distance volume incl. ghost slices, or None."""
        pkg = self.pkg
        stream = torch.cuda.current_stream() if stream is None else stream
        bounds = [slab_range(dims[2], r, self.world)[0] for r in range(self.world)] + [int(dims[2])]
        dev = slab.tex0.device
        with torch.cuda.stream(stream):
            out0 = torch.empty((dims[2], dims[1], dims[0], 4), dtype=torch.float32, device=dev)
            out1 = torch.empty_like(out0)
            outd = torch.empty((dims[2], dims[1], dims[0]), dtype=torch.float32, device=dev) if dist is not None else None
            own_d = None if dist is None else dist[slab.ghost_lo:]
            enter_stage("SlabComm.allgather_slabs: sdfv_comm_allgather_slabs over the library communicator")
            pkg.check(pkg.lib.sdfv_comm_allgather_slabs(self.handle, (C.c_uint32 * 3)(*[int(d) for d in dims]),
                                                        (C.c_uint32 * len(bounds))(*bounds), C.c_void_p(slab.owned0.data_ptr()),
                                                        C.c_void_p(slab.owned1.data_ptr()),
                                                        None if own_d is None else C.c_void_p(own_d.data_ptr()),
                                                        C.c_void_p(out0.data_ptr()), C.c_void_p(out1.data_ptr()),
                                                        None if outd is None else C.c_void_p(outd.data_ptr()),
                                                        C.c_void_p(stream.cuda_stream)))
        return (out0, out1) if dist is None else (out0, out1, outd)

    def join(self, stream=None):
        """sdfv_slab_comm_join: `stream` waits for the latest exchange (after steps taken with STEP_DEFER_JOIN)."""
        stream = torch.cuda.current_stream() if stream is None else stream
        self.pkg.check(self.pkg.lib.sdfv_slab_comm_join(self.handle, C.c_void_p(stream.cuda_stream)))

    def close(self):
        if self.handle:
            self.pkg.lib.sdfv_slab_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class SlabFiller:
    """One fill step of a rank: fill the slab, exchange the halo, with the exchange hidden behind the fill.

    The two boundary slices (the ones the neighbours need) are filled first; as soon as they are done the halo
    exchange starts on a second HIP stream while the interior of the slab is still being filled on the main
    stream.  Per neighbour pair the exchange moves 2 x W*H*16 B per direction over one xGMI link (~0.1 ms for
    512^2 slices, ~0.5 ms for 1024^2), the interior fill takes W*H*depth*32 B / ~6.5 TB/s, so for slabs of a few
    dozen slices the two are of the same order and overlapping them is what keeps weak scaling near linear.
    """

    def __init__(self, pkg, params, dims, slab, rank, world, sdf_id=0, group=None, transport="auto", periodic=False,
                 comm=None, dist=None):
        """transport: "rccl" = the library's communicator (one C call per step), "torch" = torch.distributed P2P
        ops on a second stream, "auto" = rccl on GPUs under the nccl backend, torch otherwise."""
        self.pkg, self.params, self.dims, self.slab = pkg, params, dims, slab
        self.dist = dist  # optional compact distance volume of the slab incl. ghosts: every step is the fused fill
        self.rank, self.world, self.sdf_id, self.group = rank, world, sdf_id, group
        if transport == "auto":
            transport = "rccl" if (world > 1 and slab.tex0.is_cuda and c10d.get_backend(group) == "nccl") else "torch"
        self.transport = transport
        # comm: an existing SlabComm to reuse (one communicator serves any number of slabs; the caller closes it)
        self.comm = (comm or SlabComm(pkg, rank, world, group, periodic=periodic, halo_hi=slab.halo_hi)) \
            if transport == "rccl" else None
        self.overlap = (transport == "torch" and world > 1 and slab.tex0.is_cuda
                        and (slab.z_end - slab.z_begin) >= 3)
        # highest priority: HIP keeps streams of different priorities on different hardware queues; with equal
        # priorities the exchange and the interior fill can land on one queue and run back to back
        self.comm_stream = torch.cuda.Stream(device=slab.tex0.device, priority=-1) if self.overlap else None
        z0, z1 = slab.z_begin, slab.z_end
        self.whole = pkg.make_grid(dims, z_begin=z0, z_end=z1)
        if self.overlap:
            o0, o1 = slab.owned0, slab.owned1
            # (grid, tex0 view, tex1 view) of the first slice, the last slice and the interior
            self.parts = [(pkg.make_grid(dims, z_begin=z0, z_end=z0 + 1), o0[:1], o1[:1]),
                          (pkg.make_grid(dims, z_begin=z1 - 1, z_end=z1), o0[-1:], o1[-1:]),
                          (pkg.make_grid(dims, z_begin=z0 + 1, z_end=z1 - 1), o0[1:-1], o1[1:-1])]

    def step(self):
        pkg, slab = self.pkg, self.slab
        if self.comm is not None:
            self.comm.fill_step(self.params, self.whole, slab, sdf_id=self.sdf_id, dist=self.dist)
            return
        n_owned = slab.z_end - slab.z_begin
        own_dist = None if self.dist is None else self.dist[slab.ghost_lo:slab.ghost_lo + n_owned]
        if not self.overlap:
            pkg.fill_grid(self.params, self.whole, slab.owned0, slab.owned1, sdf_id=self.sdf_id, dist=own_dist)
            if self.world > 1:
                halo_exchange(slab, self.rank, self.world, self.group)
                self._ghost_distances()
            return
        main = torch.cuda.current_stream()
        for k, (grid, t0, t1) in enumerate(self.parts[:2]):
            pkg.fill_grid(self.params, grid, t0, t1, sdf_id=self.sdf_id,
                          dist=None if own_dist is None else (own_dist[:1] if k == 0 else own_dist[-1:]))
        boundary_done = torch.cuda.Event()
        boundary_done.record(main)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(boundary_done)
            halo_exchange(slab, self.rank, self.world, self.group)  # its waits block the comm stream only
            self._ghost_distances()
            halo_done = torch.cuda.Event()
            halo_done.record(self.comm_stream)
        grid, t0, t1 = self.parts[2]
        pkg.fill_grid(self.params, grid, t0, t1, sdf_id=self.sdf_id,
                      dist=None if own_dist is None else own_dist[1:-1])  # overlaps the exchange
        main.wait_event(halo_done)

    def _ghost_distances(self):
        """torch transport: the ghost slices' share of the distance volume = tex0.r of the slices just received."""
        if self.dist is None:
            return
        slab = self.slab
        n_owned = slab.z_end - slab.z_begin
        if slab.ghost_lo:
            self.dist[:slab.ghost_lo].copy_(slab.tex0[:slab.ghost_lo, ..., 0])
        if slab.ghost_hi:
            self.dist[slab.ghost_lo + n_owned:].copy_(slab.tex0[slab.ghost_lo + n_owned:, ..., 0])


class ShardedMarch:
    """One rank's side of the raymarch over a z-sharded grid (sdfv_raymarch_slab): the grid stays where the fill put
    it, rays are handed between z-neighbours.  `round(incoming)` runs one round and returns the rays leaving through
    the low and the high face of the slab; after `world` rounds every ray has ended on exactly one rank and the image
    is the merge of the ranks' images (all-zero bits except on the rank the ray ended on).  `raymarch_sharded` below
    drives it over torch.distributed."""

    def __init__(self, pkg, rp, grid, slab, camera, width, height, want_aux=False):
        self.pkg, self.rp, self.grid, self.slab, self.camera = pkg, rp, grid, slab, camera
        self.width, self.height = width, height
        dev = slab.tex0.device
        cap = width * height
        self.rgba = torch.empty((height, width, 4), dtype=torch.float32, device=dev)
        self.aux = torch.empty((height, width, pkg.AUX_FLOATS), dtype=torch.int32, device=dev) if want_aux else None
        self.out = [torch.empty((cap, pkg.RAY_STATE_WORDS), dtype=torch.int32, device=dev) for _ in range(2)]
        self.counters = torch.zeros(2, dtype=torch.int32, device=dev)

    def round(self, incoming=None):
        """incoming None = first round; otherwise an [n, 6] int32 tensor of ray states (n may be 0).  Returns (down, up)."""
        self.counters.zero_()
        self.pkg.raymarch_slab(self.rp, self.grid, self.slab.ghost_lo, self.slab.ghost_hi, self.slab.tex0, self.slab.tex1,
                               self.camera, self.width, self.height, self.rgba, self.out[0], self.out[1], self.counters,
                               in_states=incoming, aux=self.aux)
        n_down, n_up = (int(v) for v in self.counters.tolist())  # synchronises
        return self.out[0][:n_down], self.out[1][:n_up]


def _exchange_rays(down, up, rank, world, group):
    """Rays leaving downwards go to rank-1, upwards to rank+1; returns what the two neighbours sent here, concatenated.
    Counts travel first (the lists are data dependent), then the payloads."""
    dev = down.device
    staged = _needs_host_staging(down, group)
    cdev = torch.device("cpu") if staged else dev
    peers = [(rank - 1, down), (rank + 1, up)]
    peers = [(p, t) for p, t in peers if 0 <= p < world]
    send_n = [torch.tensor([t.shape[0]], dtype=torch.int64, device=cdev) for _, t in peers]
    recv_n = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in peers]
    ops = []
    for (p, _), sn, rn in zip(peers, send_n, recv_n):
        ops += [c10d.P2POp(c10d.isend, sn, p, group), c10d.P2POp(c10d.irecv, rn, p, group)]
    enter_stage(f"_exchange_rays: ray counts with {[p for p, _ in peers]}")
    for req in c10d.batch_isend_irecv(ops):
        req.wait()
    ops, bufs = [], []
    for (p, t), rn in zip(peers, recv_n):
        n = int(rn.item())
        if t.shape[0]:
            ops.append(c10d.P2POp(c10d.isend, t.cpu() if staged else t.contiguous(), p, group))
        if n:
            buf = torch.empty((n, t.shape[1]), dtype=t.dtype, device=cdev)
            ops.append(c10d.P2POp(c10d.irecv, buf, p, group))
            bufs.append(buf)
    if ops:
        enter_stage(f"_exchange_rays: ray payloads ({len(ops)} ops)")
        for req in c10d.batch_isend_irecv(ops):
            req.wait()
    enter_stage("_exchange_rays: done")
    if not bufs:
        return torch.empty((0, down.shape[1]), dtype=down.dtype, device=dev)
    return torch.cat([b.to(dev) for b in bufs], dim=0)


def merge_sharded_aux(aux_sum):
    """Merge of the ranks' aux images (summed bit patterns) -> the single-GPU record: pixels no rank reports (status 0) get the
    cleared record's depth of 1.0 back (aux words: status, steps, hit_pos[3], t, raw0[4], raw1[4], normal[3], depth)."""
    out = aux_sum.clone()
    depth = out[..., -1].view(torch.float32)
    depth[out[..., 0] == 0] = 1.0
    return out


def raymarch_sharded(pkg, rp, grid, slab, camera, width, height, rank, world, group=None, want_aux=False, comm=None):
    """Raymarch of a grid that stays z-sharded across the ranks (slab + ghosts as left by the halo exchange).
    Every rank gets the full image (all-reduce of the per-rank images' bit patterns: each pixel is written by one rank).
    Returns rgba [H, W, 4] (+ the merged aux image); bit-identical to pkg.raymarch over the whole grid, except that
    aux.normal stays (0, 0, 0).  comm: a (non-periodic) SlabComm -- the whole march then runs inside the library over its
    RCCL communicator (sdfv_slab_march: no host round trip per round); otherwise torch.distributed carries the rays, with a
    counter read-back and a count exchange per round (the gloo / CPU-test transport)."""
    if comm is not None and comm.handle and not comm.periodic:
        # Ray lists: the bounded default (an eighth of the pixels) first, all pixels after an overflow -- remembered per image
        # size, so that the frames that follow start where this one ended.  The retry needs every rank to agree
        # (an overflow on ONE rank sends all of them round again): that takes a torch.distributed group; with the library
        # communicator alone the lists simply hold every pixel and nothing can overflow.
        # A message of the march is 16 + 24 * capacity bytes WHATEVER it carries: every rank of the communicator must pass the same
        # capacity, or the sends and receives differ in size.  The remembered capacity therefore lives ON the communicator (not in
        # a process-wide table shared by other communicators, restarted peers or ranks that run as threads), and the ranks agree
        # on it -- all-reduce(MAX) -- before the first march of every frame.
        agree = world > 1 and c10d.is_available() and c10d.is_initialized()
        remembered = comm.__dict__.setdefault("_sharded_capacity", {})  # (width, height) -> capacity the last frame needed
        key = (int(width), int(height))
        if world > 1 and not agree:
            attempts = [width * height]
        else:
            first = remembered.get(key) or 0  # 0 = the library's bounded default
            if agree:
                want = torch.tensor([first], dtype=torch.int64, device=slab.tex0.device if c10d.get_backend(group) == "nccl" else "cpu")
                enter_stage("raymarch_sharded: all_reduce(MAX) of the ray-list capacity")
                c10d.all_reduce(want, op=c10d.ReduceOp.MAX, group=group)
                first = int(want.item())
            attempts = [first or None, width * height] if first != width * height else [width * height]
        overflow = left = 0
        for capacity in attempts:
            rgba, aux, status = comm.march(rp, grid, slab, camera, width, height, want_aux=want_aux, capacity=capacity)
            st = status.clone()
            if agree:
                enter_stage("raymarch_sharded: all_reduce(MAX) of the march status")
                c10d.all_reduce(st, op=c10d.ReduceOp.MAX, group=group)
            overflow, left = (int(v) for v in st.tolist())  # synchronises: the caller wants the image now
            if not (overflow or left):
                return (rgba, aux) if want_aux else rgba
            if not overflow:
                break  # rays left over after `world` rounds without any list overflowing: larger lists cannot help
            remembered[key] = width * height
        raise pkg.SdfvError(-1, f"sdfv_slab_march: ray lists overflowed ({overflow}) / {left} rays left over")
    m = ShardedMarch(pkg, rp, grid, slab, camera, width, height, want_aux)
    incoming = None
    for _ in range(world):
        down, up = m.round(incoming)
        incoming = _exchange_rays(down, up, rank, world, group) if world > 1 else down[:0]
    staged = world > 1 and _needs_host_staging(m.rgba, group)

    def merge(t):
        """Each pixel is written by one rank and is all-zero bits elsewhere, so adding the BIT PATTERNS as integers
        over the ranks reproduces them exactly (a float sum would turn -0.0 into +0.0; RCCL has no bitwise OR)."""
        if world == 1:
            return t
        bits = t.view(torch.int32)
        h = bits.cpu() if staged else bits
        enter_stage("raymarch_sharded: all_reduce (integer sum of bit patterns) of the ranks' images")
        c10d.all_reduce(h, op=c10d.ReduceOp.SUM, group=group)
        return (h.to(t.device) if staged else h).view(t.dtype)

    rgba = merge(m.rgba)
    if not want_aux:
        return rgba
    return rgba, merge_sharded_aux(merge(m.aux))


def gather_replica(slab, dims, world, group=None, comm=None):
    """Full grid on every rank from the slabs (all-gather; slabs may differ by one slice, so each is
    padded to the deepest slab for the collective and trimmed afterwards).  comm: a (non-periodic) SlabComm -- the library's
    own collective then (sdfv_comm_allgather_slabs); torch.distributed is the gloo / CPU-test transport."""
    if comm is not None and comm.handle and not comm.periodic:
        return comm.allgather_slabs(slab, dims)
    ranges = [slab_range(dims[2], r, world) for r in range(world)]
    deepest = max(z1 - z0 for z0, z1 in ranges)
    outs = []
    staged = _needs_host_staging(slab.tex0, group)
    for owned in (slab.owned0, slab.owned1):
        dev = torch.device("cpu") if staged else owned.device
        padded = torch.zeros((deepest, dims[1], dims[0], 4), dtype=owned.dtype, device=dev)
        padded[:owned.shape[0]] = owned
        parts = [torch.empty_like(padded) for _ in range(world)]
        enter_stage("gather_replica: all_gather of the slabs")
        c10d.all_gather(parts, padded, group=group)
        outs.append(torch.cat([p[:z1 - z0] for p, (z0, z1) in zip(parts, ranges)], dim=0).to(owned.device))
    return outs[0], outs[1]


def split_cameras(n_cameras, rank, world):
    """Cameras dealt to ranks in contiguous blocks (config 5: 64 cameras over 8 GPUs = 8 each)."""
    return range(n_cameras * rank // world, n_cameras * (rank + 1) // world)


def split_rows(height, rank, world, tile=16):
    """Image-tile split of ONE image over the ranks (config 5's "image-tile split"): contiguous bands of rows, cut on
    multiples of the kernel's 16-row workgroup tile so no workgroup straddles two ranks; -> (y0, y1)."""
    tiles = (height + tile - 1) // tile
    return min(height, tiles * rank // world * tile), min(height, tiles * (rank + 1) // world * tile)


def band_height_for(height, world):
    """Rows per band of the image-tile split: a workgroup's 16, or a wave's 8 when a rank would get fewer than 12 bands of 16
    (the rows under the object are then dealt too coarsely: 8 ranks at 1080p hold 4 or 5 object bands each -- tools/
    split_balance.py: 4.9x -> 5.4x with 8-row bands; 4 ranks lose 2 % to the finer bands and keep 16)."""
    return 8 if (height + 15) // 16 < 12 * world else 16


def split_bands(height, rank, world, band_height=None):
    """The BALANCED image-tile split of config 5: rank r renders the tile bands r, r + world, r + 2 world, ... of every image
    (sdfv_march_desc.band_*) -- the rows under the object cost ten times the background's, so contiguous ranges leave the outer
    ranks idle; -> (band_first, band_step, band_height) for raymarch(bands=...)."""
    return rank, world, band_height_for(height, world) if band_height is None else band_height


def band_rows(height, first, step, tile=16):
    """Image rows of the bands (first, step), in the order raymarch(bands=...) stores them."""
    rows = []
    for t in range(first, (height + tile - 1) // tile, step):
        rows.extend(range(t * tile, min(height, (t + 1) * tile)))
    return rows


def assemble_bands(parts, height, band_height=None):
    """The image [n_cam, height, W, ...] from the `world` band sets of split_bands, parts[r] = what rank r rendered."""
    world = len(parts)
    band_height = band_height_for(height, world) if band_height is None else band_height
    out = torch.empty((parts[0].shape[0], height) + tuple(parts[0].shape[2:]), dtype=parts[0].dtype, device=parts[0].device)
    for r, part in enumerate(parts):
        out[:, torch.as_tensor(band_rows(height, r, world, band_height), dtype=torch.long, device=out.device)] = part
    return out


def gather_bands(part, height, rank, world, dst=0, group=None, comm=None, band_height=None):
    """Collect the band sets of split_bands on rank `dst` -> [n_cam, height, W, 4] (None elsewhere).  comm: a SlabComm -- the
    library's own collective then (sdfv_comm_gather_bands); torch.distributed is the gloo / CPU-test transport."""
    band_height = band_height_for(height, world) if band_height is None else band_height  # (split_bands' default)
    if world == 1 and comm is None:
        return part
    if comm is not None and comm.handle:
        return comm.gather_bands(part, height, dst=dst, band_height=band_height)
    deepest = max(len(band_rows(height, r, world, band_height)) for r in range(world))
    staged = _needs_host_staging(part, group)
    dev = torch.device("cpu") if staged else part.device
    padded = torch.zeros((part.shape[0], deepest) + tuple(part.shape[2:]), dtype=part.dtype, device=dev)
    padded[:, :part.shape[1]] = part
    got = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    enter_stage("gather_bands: gather of the tile bands")
    c10d.gather(padded, got, dst=dst, group=group)
    if rank != dst:
        return None
    return assemble_bands([g[:, :len(band_rows(height, r, world, band_height))] for r, g in enumerate(got)], height, band_height).to(part.device)


def gather_rows(band, height, rank, world, dst=0, group=None, tile=16):
    """Collect the row bands of split_rows ([n_cam, rows, W, 4] each) on rank `dst` -> [n_cam, height, W, 4]."""
    if world == 1:
        return band
    ranges = [split_rows(height, r, world, tile) for r in range(world)]
    deepest = max(y1 - y0 for y0, y1 in ranges)
    staged = _needs_host_staging(band, group)
    dev = torch.device("cpu") if staged else band.device
    padded = torch.zeros((band.shape[0], deepest) + tuple(band.shape[2:]), dtype=band.dtype, device=dev)
    padded[:, :band.shape[1]] = band
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    enter_stage("gather_rows: gather of the row bands")
    c10d.gather(padded, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p[:, :y1 - y0] for p, (y0, y1) in zip(parts, ranges)], dim=1).to(band.device)


def gather_images(rgba, n_cameras, rank, world, dst=0, group=None, comm=None):
    """Config 5's optional last step: collect every rank's rendered cameras ([n_local, H, W, 4]) on rank `dst` in
    camera order.  Camera counts may differ by one between ranks, so each block is padded to the largest count.
    Returns [n_cameras, H, W, 4] on `dst`, None elsewhere.  comm: a SlabComm -- sdfv_comm_gather_cameras then."""
    if world == 1 and comm is None:
        return rgba
    if comm is not None and comm.handle:
        return comm.gather_cameras(rgba, n_cameras, dst=dst)
    counts = [len(split_cameras(n_cameras, r, world)) for r in range(world)]
    staged = _needs_host_staging(rgba, group)
    dev = torch.device("cpu") if staged else rgba.device
    padded = torch.zeros((max(counts),) + tuple(rgba.shape[1:]), dtype=rgba.dtype, device=dev)
    padded[:rgba.shape[0]] = rgba
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    enter_stage("gather_images: gather of the cameras' images")
    c10d.gather(padded, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).to(rgba.device)
