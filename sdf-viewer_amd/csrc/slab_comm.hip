// slab_comm.hip -- z-slab sharding of the grid fill: one-voxel halo exchange over RCCL, overlapped with the fill.
// Implements the sdfv_slab_* part of include/sdfgrid.h.  The reference has nothing of the kind (single thread,
// scene/sdf/mod.rs:173-215); the layout follows SURVEY.md 8(e).
//
// RCCL is bound at run time: dlopen("librccl.so.1") returns the copy the process already holds (PyTorch ships its
// own under the same soname) or loads ROCm's.  Only the point-to-point calls are used; a rank talks to rank-1 and
// rank+1, each pair riding one xGMI link per direction.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <new>

#include "../../include/sdfgrid.h"
#include "api_internal.h"

namespace {

// The slice of rccl.h this file needs (/opt/rocm/include/rccl/rccl.h; the ABI is NCCL's).
constexpr int kNcclSuccess = 0;
constexpr int kNcclFloat = 7;  // ncclFloat32
struct NcclUniqueId {
    char internal[SDFV_COMM_ID_BYTES];
};
using NcclComm = void*;

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* load_error = nullptr;
};

template <typename F>
bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

const Rccl* rccl() {
    static const Rccl lib = [] {
        Rccl r;
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            r.load_error = "librccl.so.1 not found";
            return r;
        }
        bool ok = true;  // bind every entry point, then judge
        ok = bind(r.handle, "ncclGetUniqueId", r.GetUniqueId) && ok;
        ok = bind(r.handle, "ncclCommInitRank", r.CommInitRank) && ok;
        ok = bind(r.handle, "ncclCommDestroy", r.CommDestroy) && ok;
        ok = bind(r.handle, "ncclGroupStart", r.GroupStart) && ok;
        ok = bind(r.handle, "ncclGroupEnd", r.GroupEnd) && ok;
        ok = bind(r.handle, "ncclSend", r.Send) && ok;
        ok = bind(r.handle, "ncclRecv", r.Recv) && ok;
        ok = bind(r.handle, "ncclGetErrorString", r.GetErrorString) && ok;
        if (!ok) r.load_error = "librccl.so.1 lacks an ncclSend/ncclRecv entry point";
        return r;
    }();
    return &lib;
}

int need_rccl(const Rccl*& lib) {
    lib = rccl();
    if (lib->load_error) return sdfv::set_error(SDFV_ERR_COMM, "RCCL: %s", lib->load_error);
    return SDFV_OK;
}

#define SDFV_RCCL(lib, call)                                                                            \
    do {                                                                                                \
        int r_ = (lib)->call;                                                                           \
        if (r_ != kNcclSuccess) return sdfv::set_error(SDFV_ERR_COMM, "RCCL %s: %s", #call, (lib)->GetErrorString(r_)); \
    } while (0)

#define SDFV_HIPC(call)                                                                                 \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return sdfv::set_error(SDFV_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct sdfv_slab_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    bool periodic = false;
    hipStream_t comm_stream = nullptr;
    hipEvent_t boundary_done = nullptr, halo_done = nullptr;

    bool has_lo() const { return periodic || rank > 0; }
    bool has_hi() const { return periodic || rank < world - 1; }
    int lo_peer() const { return (rank + world - 1) % world; }
    int hi_peer() const { return (rank + 1) % world; }
};

namespace {

int check_slab(const sdfv_slab_comm* c, const sdfv_grid* g, const float* tex0, const float* tex1) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "communicator is NULL");
    if (!g || !tex0 || !tex1) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "grid or texture pointer is NULL");
    if (g->z_begin >= g->z_end || g->z_end > g->dims[2] || g->dims[0] == 0 || g->dims[1] == 0)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "slab [%u, %u) is empty or outside the %u slices of the grid",
                               g->z_begin, g->z_end, g->dims[2]);
    return SDFV_OK;
}

// One ncclGroup: first owned slice down, last owned slice up, the neighbours' into the ghosts; both textures.
int enqueue_exchange(const Rccl* lib, const sdfv_slab_comm* c, const sdfv_grid* g, float* tex0, float* tex1,
                     hipStream_t stream) {
    const size_t slice = (size_t)g->dims[0] * g->dims[1] * 4;  // floats per z-slice of one texture
    const size_t owned = g->z_end - g->z_begin;
    const size_t lo = c->has_lo() ? 1 : 0;
    if (!c->has_lo() && !c->has_hi()) return SDFV_OK;
    SDFV_RCCL(lib, GroupStart());
    int first_error = kNcclSuccess;  // a group that was opened is always closed, whatever happens inside it
    auto post = [&](int r) {
        if (first_error == kNcclSuccess) first_error = r;
    };
    for (float* t : {tex0, tex1}) {
        float* first_owned = t + lo * slice;
        float* last_owned = t + (lo + owned - 1) * slice;
        // Sends go down then up, receives come from above then from below: messages between one pair of ranks
        // match in posting order, and with a periodic world of 1 or 2 both neighbours are the same rank.
        if (c->has_lo()) post(lib->Send(first_owned, slice, kNcclFloat, c->lo_peer(), c->comm, stream));
        if (c->has_hi()) post(lib->Send(last_owned, slice, kNcclFloat, c->hi_peer(), c->comm, stream));
        if (c->has_hi()) post(lib->Recv(t + (lo + owned) * slice, slice, kNcclFloat, c->hi_peer(), c->comm, stream));
        if (c->has_lo()) post(lib->Recv(t, slice, kNcclFloat, c->lo_peer(), c->comm, stream));
    }
    post(lib->GroupEnd());
    if (first_error != kNcclSuccess)
        return sdfv::set_error(SDFV_ERR_COMM, "RCCL halo exchange: %s", lib->GetErrorString(first_error));
    return SDFV_OK;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int sdfv_slab_comm_unique_id(unsigned char id_out[SDFV_COMM_ID_BYTES]) {
    if (!id_out) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "id_out is NULL");
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    NcclUniqueId id;
    SDFV_RCCL(lib, GetUniqueId(&id));
    memcpy(id_out, id.internal, SDFV_COMM_ID_BYTES);
    return SDFV_OK;
}

int sdfv_slab_comm_create(const unsigned char id[SDFV_COMM_ID_BYTES], int rank, int world, uint32_t flags,
                          sdfv_slab_comm** out) {
    if (!id || !out) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "id or out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "rank %d is not in a world of %d", rank, world);
    if (flags & ~SDFV_COMM_PERIODIC) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0) return sdfv::set_error(SDFV_ERR_NO_DEVICE, "no HIP device for the communicator");
    sdfv_slab_comm* c = new (std::nothrow) sdfv_slab_comm;
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "out of host memory");
    c->rank = rank;
    c->world = world;
    c->periodic = (flags & SDFV_COMM_PERIODIC) != 0;
    NcclUniqueId uid;
    memcpy(uid.internal, id, SDFV_COMM_ID_BYTES);
    int r = lib->CommInitRank(&c->comm, world, uid, rank);
    if (r != kNcclSuccess) {
        delete c;
        return sdfv::set_error(SDFV_ERR_COMM, "RCCL ncclCommInitRank(rank %d of %d): %s", rank, world, lib->GetErrorString(r));
    }
    // Highest priority: the exchange is short and the neighbours wait for it; HIP also keeps streams of different
    // priorities on different hardware queues, without which the exchange and the interior fill (both enqueued
    // back to back) land on one queue and run one after the other (profiles/r01_slab_step_timeline.txt).
    int prio_low = 0, prio_high = 0;
    e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_high);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->boundary_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->halo_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        sdfv_slab_comm_destroy(c);
        return sdfv::set_error(SDFV_ERR_HIP, "communicator stream/events: %s", hipGetErrorString(e));
    }
    *out = c;
    return SDFV_OK;
}

int sdfv_slab_comm_destroy(sdfv_slab_comm* c) {
    if (!c) return SDFV_OK;
    const Rccl* lib = rccl();
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm && !lib->load_error) (void)lib->CommDestroy(c->comm);
    if (c->boundary_done) (void)hipEventDestroy(c->boundary_done);
    if (c->halo_done) (void)hipEventDestroy(c->halo_done);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    delete c;
    return SDFV_OK;
}

int sdfv_slab_halo_exchange(sdfv_slab_comm* c, const sdfv_grid* slab, float* tex0, float* tex1, void* stream) {
    if (int rc = check_slab(c, slab, tex0, tex1)) return rc;
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    return enqueue_exchange(lib, c, slab, tex0, tex1, (hipStream_t)stream);
}

int sdfv_slab_fill_step(sdfv_slab_comm* c, const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab,
                        float* tex0, float* tex1, void* stream) {
    if (int rc = check_slab(c, slab, tex0, tex1)) return rc;
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    hipStream_t main = (hipStream_t)stream;
    const size_t slice = (size_t)slab->dims[0] * slab->dims[1] * 4;
    const uint32_t z0 = slab->z_begin, z1 = slab->z_end, owned = z1 - z0;
    float* o0 = tex0 + (c->has_lo() ? slice : 0);  // first owned slice
    float* o1 = tex1 + (c->has_lo() ? slice : 0);
    const bool exchange = c->has_lo() || c->has_hi();

    auto fill = [&](uint32_t za, uint32_t zb) {
        sdfv_grid part = *slab;
        part.z_begin = za;
        part.z_end = zb;
        return sdfv_fill_grid(params, sdf_id, &part, o0 + (size_t)(za - z0) * slice, o1 + (size_t)(za - z0) * slice, main);
    };

    if (!exchange || owned < 3) {  // nothing to hide the exchange behind
        if (int rc = fill(z0, z1)) return rc;
        return exchange ? enqueue_exchange(lib, c, slab, tex0, tex1, main) : SDFV_OK;
    }
    // boundary slices first: they are what the neighbours wait for
    if (int rc = sdfv::fill_boundary_slices(params, sdf_id, slab, o0, o1, main)) return rc;
    SDFV_HIPC(hipEventRecord(c->boundary_done, main));
    SDFV_HIPC(hipStreamWaitEvent(c->comm_stream, c->boundary_done, 0));
    if (int rc = enqueue_exchange(lib, c, slab, tex0, tex1, c->comm_stream)) return rc;
    SDFV_HIPC(hipEventRecord(c->halo_done, c->comm_stream));
    if (int rc = fill(z0 + 1, z1 - 1)) return rc;  // overlaps the exchange
    SDFV_HIPC(hipStreamWaitEvent(main, c->halo_done, 0));
    return SDFV_OK;
}

}  // extern "C"
#pragma GCC visibility pop
