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
constexpr int kNcclInt8 = 0, kNcclInt32 = 2, kNcclSum = 0;
struct NcclUniqueId {
    char internal[SDFV_COMM_ID_BYTES];
};
using NcclComm = void*;

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommCount)(const NcclComm, int*) = nullptr;
    int (*CommUserRank)(const NcclComm, int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
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
        const char* path = sdfv::claim_rccl_library_path();  // (from here on SDFV_OPT_RCCL_LIBRARY is refused: one RCCL per process)
        if (path && path[0]) {  // SDFV_OPT_RCCL_LIBRARY: this file and nothing else
            r.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) {
                const char* why = dlerror();
                static std::string message;  // (this initialiser runs once per process)
                message = std::string("the library named by SDFV_OPT_RCCL_LIBRARY could not be loaded: ") + (why ? why : "dlopen failed");
                r.load_error = message.c_str();
                return r;
            }
        } else {
            r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.handle) {
            r.load_error = "librccl.so.1 not found";
            return r;
        }
        bool ok = true;  // bind every entry point, then judge
        ok = bind(r.handle, "ncclGetUniqueId", r.GetUniqueId) && ok;
        ok = bind(r.handle, "ncclCommInitRank", r.CommInitRank) && ok;
        ok = bind(r.handle, "ncclCommDestroy", r.CommDestroy) && ok;
        ok = bind(r.handle, "ncclCommCount", r.CommCount) && ok;
        ok = bind(r.handle, "ncclCommUserRank", r.CommUserRank) && ok;
        ok = bind(r.handle, "ncclGroupStart", r.GroupStart) && ok;
        ok = bind(r.handle, "ncclGroupEnd", r.GroupEnd) && ok;
        ok = bind(r.handle, "ncclSend", r.Send) && ok;
        ok = bind(r.handle, "ncclRecv", r.Recv) && ok;
        ok = bind(r.handle, "ncclAllReduce", r.AllReduce) && ok;
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
    uint32_t halo_hi = 1;  // upper halo depth: every rank sends this many leading slices down
    hipStream_t comm_stream = nullptr;
    hipEvent_t boundary_done = nullptr, halo_done = nullptr;
    // packed messages: one block holding [send_lo | send_hi | recv_hi | recv_lo], sized for stage_slice texels per slice
    float* stage = nullptr;
    size_t stage_slice = 0;
    // the word the fill launch stores the step number to when it starts, and the communicator's stream waits on
    uint32_t* signal = nullptr;
    uint32_t step = 0;
    bool can_wait_value = false;

    bool has_lo() const { return periodic || rank > 0; }
    bool has_hi() const { return periodic || rank < world - 1; }
    int lo_peer() const { return (rank + world - 1) % world; }
    int hi_peer() const { return (rank + 1) % world; }
    uint32_t ghost_lo() const { return has_lo() ? 1u : 0u; }
    uint32_t ghost_hi() const { return has_hi() ? halo_hi : 0u; }
    // staging layout, in floats (a slice of one texture = 4 * stage_slice floats)
    size_t slice_f() const { return stage_slice * 4; }
    float* send_lo() const { return stage; }                                                  // 2 * halo_hi slices
    float* send_hi() const { return stage + 2 * halo_hi * slice_f(); }                        // 2 slices
    float* recv_hi() const { return stage + (2 * halo_hi + 2) * slice_f(); }                  // 2 * halo_hi slices
    float* recv_lo() const { return stage + (4 * halo_hi + 2) * slice_f(); }                  // 2 slices
    size_t stage_floats() const { return (4 * halo_hi + 4) * slice_f(); }
};

namespace {

int check_slab(const sdfv_slab_comm* c, const sdfv_grid* g, const float* tex0, const float* tex1) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "communicator is NULL");
    if (!g || !tex0 || !tex1) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "grid or texture pointer is NULL");
    if (((uintptr_t)tex0 | (uintptr_t)tex1) & 15)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "texel buffers (tex0, tex1) must be 16-byte aligned");
    if (g->z_begin >= g->z_end || g->z_end > g->dims[2] || g->dims[0] == 0 || g->dims[1] == 0)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "slab [%u, %u) is empty or outside the %u slices of the grid",
                               g->z_begin, g->z_end, g->dims[2]);
    if ((c->has_lo() || c->has_hi()) && g->z_end - g->z_begin < c->halo_hi)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "a halo of %u slices needs slabs of at least %u slices",
                               c->halo_hi, c->halo_hi);
    return SDFV_OK;
}

int ensure_staging(sdfv_slab_comm* c, size_t slice_texels) {
    if (c->stage && c->stage_slice == slice_texels) return SDFV_OK;
    if (c->stage) {
        SDFV_HIPC(hipStreamSynchronize(c->comm_stream));
        (void)hipFree(c->stage);
        c->stage = nullptr;
    }
    c->stage_slice = slice_texels;
    SDFV_HIPC(hipMalloc((void**)&c->stage, c->stage_floats() * sizeof(float)));
    return SDFV_OK;
}

// One ncclGroup straight on the textures: the first halo_hi owned slices down, the last owned slice up, the neighbours'
// into the ghosts; one message per texture, neighbour and direction.
int enqueue_exchange_direct(const Rccl* lib, const sdfv_slab_comm* c, const sdfv_grid* g, float* tex0, float* tex1,
                            hipStream_t stream) {
    const size_t slice = (size_t)g->dims[0] * g->dims[1] * 4;  // floats per z-slice of one texture
    const size_t owned = g->z_end - g->z_begin;
    const size_t lo = c->ghost_lo();
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
        if (c->has_lo()) post(lib->Send(first_owned, c->halo_hi * slice, kNcclFloat, c->lo_peer(), c->comm, stream));
        if (c->has_hi()) post(lib->Send(last_owned, slice, kNcclFloat, c->hi_peer(), c->comm, stream));
        if (c->has_hi()) post(lib->Recv(t + (lo + owned) * slice, c->halo_hi * slice, kNcclFloat, c->hi_peer(), c->comm, stream));
        if (c->has_lo()) post(lib->Recv(t, slice, kNcclFloat, c->lo_peer(), c->comm, stream));
    }
    post(lib->GroupEnd());
    if (first_error != kNcclSuccess)
        return sdfv::set_error(SDFV_ERR_COMM, "RCCL halo exchange: %s", lib->GetErrorString(first_error));
    return SDFV_OK;
}

// One ncclGroup over the packed staging buffers (filled by the boundary workgroups of the ordered fill): ONE message per
// neighbour and direction carrying both textures' slices, then one launch that copies the received slices into the ghosts.
// dist (optional): the slab's compact distance volume incl. ghost slices -- the copy out of the receive buffers writes the
// ghost slices' share (tex0.r) in the same pass
int enqueue_exchange_packed(const Rccl* lib, const sdfv_slab_comm* c, const sdfv_grid* g, float* tex0, float* tex1,
                            float* dist, hipStream_t stream) {
    const size_t slice = c->slice_f();
    const size_t owned = g->z_end - g->z_begin;
    const size_t lo = c->ghost_lo();
    if (!c->has_lo() && !c->has_hi()) return SDFV_OK;
#ifdef SDFV_TUNING  // diagnostics: drop the RCCL group (8) or the ghost copy (16) to price them (results undefined)
    const bool no_rccl = (sdfv::options().wave_timing & 8) != 0, no_copy = (sdfv::options().wave_timing & 16) != 0;
#else
    const bool no_rccl = false, no_copy = false;
#endif
    int first_error = kNcclSuccess;
    auto post = [&](int r) {
        if (first_error == kNcclSuccess) first_error = r;
    };
    if (!no_rccl) {
    SDFV_RCCL(lib, GroupStart());
    if (c->has_lo()) post(lib->Send(c->send_lo(), 2 * c->halo_hi * slice, kNcclFloat, c->lo_peer(), c->comm, stream));
    if (c->has_hi()) post(lib->Send(c->send_hi(), 2 * slice, kNcclFloat, c->hi_peer(), c->comm, stream));
    if (c->has_hi()) post(lib->Recv(c->recv_hi(), 2 * c->halo_hi * slice, kNcclFloat, c->hi_peer(), c->comm, stream));
    if (c->has_lo()) post(lib->Recv(c->recv_lo(), 2 * slice, kNcclFloat, c->lo_peer(), c->comm, stream));
    post(lib->GroupEnd());
    }
    if (first_error != kNcclSuccess)
        return sdfv::set_error(SDFV_ERR_COMM, "RCCL halo exchange: %s", lib->GetErrorString(first_error));
    if (no_copy) return SDFV_OK;
    const float* src[4] = {nullptr, nullptr, nullptr, nullptr};
    float* dst[4] = {nullptr, nullptr, nullptr, nullptr};
    float* r_out[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t n[4] = {0, 0, 0, 0};
    if (c->has_hi()) {  // [tex0 halo_hi slices | tex1 halo_hi slices] from above -> the upper ghosts
        src[0] = c->recv_hi();                          dst[0] = tex0 + (lo + owned) * slice; n[0] = c->halo_hi * c->stage_slice;
        if (dist) r_out[0] = dist + (lo + owned) * (slice / 4);
        src[1] = c->recv_hi() + c->halo_hi * slice;     dst[1] = tex1 + (lo + owned) * slice; n[1] = c->halo_hi * c->stage_slice;
    }
    if (c->has_lo()) {  // [tex0 slice | tex1 slice] from below -> the lower ghost
        src[2] = c->recv_lo();                          dst[2] = tex0; n[2] = c->stage_slice;
        if (dist) r_out[2] = dist;
        src[3] = c->recv_lo() + slice;                  dst[3] = tex1; n[3] = c->stage_slice;
    }
    return sdfv::copy_texel_segments(src, dst, n, r_out, stream);
}

// A band set is stored as [camera][band k][<= B rows][width], B = 16 or 8: pixel row j of the set is row
// B * (band_first + (j / B) * band_step) + j % B of the image.  One thread per 16-byte group of a pixel row.
__global__ __launch_bounds__(256) void bands_scatter_kernel(const float4* __restrict__ part, float4* __restrict__ out,
                                                            uint32_t row_vec4, uint32_t rows_part, uint32_t height,
                                                            uint32_t band_first, uint32_t band_step, uint32_t band_shift) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, cam = blockIdx.z;
    if (x >= row_vec4) return;
    const uint32_t y = ((band_first + (j >> band_shift) * band_step) << band_shift) + (j & ((1u << band_shift) - 1u));
    out[((size_t)cam * height + y) * row_vec4 + x] = part[((size_t)cam * rows_part + j) * row_vec4 + x];
}
__global__ __launch_bounds__(256) void bands_scatter_scalar_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                   uint32_t row_floats, uint32_t rows_part, uint32_t height,
                                                                   uint32_t band_first, uint32_t band_step, uint32_t band_shift) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, cam = blockIdx.z;
    if (x >= row_floats) return;
    const uint32_t y = ((band_first + (j >> band_shift) * band_step) << band_shift) + (j & ((1u << band_shift) - 1u));
    out[((size_t)cam * height + y) * row_floats + x] = part[((size_t)cam * rows_part + j) * row_floats + x];
}

int scatter_bands(const float* part, uint32_t band_first, uint32_t band_step, uint32_t band_height, uint32_t n_cameras, uint32_t width,
                  uint32_t height, uint32_t channels, float* out, hipStream_t st) {
    if (band_height != 0 && band_height != 8 && band_height != 16)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "band_height %u: 8 or 16 (0 = 16)", band_height);
    const uint32_t shift = band_height == 8 ? 3u : 4u;
    const uint32_t rows = sdfv_band_rows_ex(height, band_first, band_step, band_height);
    if (rows == 0 || n_cameras == 0 || width == 0) return SDFV_OK;
    const uint64_t row_floats = (uint64_t)width * channels;
    if (rows > 65535u || n_cameras > 65535u || row_floats >= (1ull << 32))
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "band set of %u rows x %u cameras is too large for one launch", rows, n_cameras);
    if (row_floats % 4 == 0 && !(((uintptr_t)part | (uintptr_t)out) & 15)) {
        const uint32_t v = (uint32_t)(row_floats / 4);
        hipLaunchKernelGGL(bands_scatter_kernel, dim3((v + 255) / 256, rows, n_cameras), dim3(256), 0, st,
                           reinterpret_cast<const float4*>(part), reinterpret_cast<float4*>(out), v, rows, height, band_first, band_step, shift);
    } else {
        hipLaunchKernelGGL(bands_scatter_scalar_kernel, dim3(((uint32_t)row_floats + 255) / 256, rows, n_cameras), dim3(256), 0, st,
                           part, out, (uint32_t)row_floats, rows, height, band_first, band_step, shift);
    }
    SDFV_HIPC(hipGetLastError());
    return SDFV_OK;
}

// floats of rank r's band set (band_first = r, band_step = world) of n_cameras images
size_t band_set_floats(uint32_t height, int r, int world, uint32_t band_height, uint32_t n_cameras, uint32_t width, uint32_t channels) {
    return (size_t)n_cameras * sdfv_band_rows_ex(height, (uint32_t)r, (uint32_t)world, band_height) * width * channels;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

// ---- config 5's collectives ("gather of RGBA tiles to rank 0", replicas "by ncclAllGather of slabs") ----

int sdfv_bands_scatter(const float* part, uint32_t band_first, uint32_t band_step, uint32_t band_height, uint32_t n_cameras,
                       uint32_t width, uint32_t height, uint32_t channels, float* out, void* stream) {
    if (!part || !out) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (band_step == 0 || channels == 0) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "band_step or channels is 0");
    if (((uintptr_t)part | (uintptr_t)out) & 3) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "buffers must be 4-byte aligned");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        (void)hipGetLastError();
        return sdfv::set_error(SDFV_ERR_NO_DEVICE, "no HIP device visible: libsdfgrid has no CPU path");
    }
    return scatter_bands(part, band_first, band_step, band_height, n_cameras, width, height, channels, out, (hipStream_t)stream);
}

size_t sdfv_comm_gather_bands_scratch_bytes(const sdfv_slab_comm* c, int dst, uint32_t band_height, uint32_t n_cameras, uint32_t width,
                                            uint32_t height, uint32_t channels) {
    if (!c || c->rank != dst) return 0;
    size_t floats = 0;
    for (int r = 0; r < c->world; ++r)
        if (r != dst) floats += (band_set_floats(height, r, c->world, band_height, n_cameras, width, channels) + 3) & ~(size_t)3;  // 16-byte aligned parts
    return floats * sizeof(float);
}

int sdfv_comm_gather_bands(sdfv_slab_comm* c, const float* part, uint32_t band_height, uint32_t n_cameras, uint32_t width,
                           uint32_t height, uint32_t channels, int dst, float* out, void* scratch, size_t scratch_bytes, void* stream) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "communicator is NULL");
    if (dst < 0 || dst >= c->world) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "dst %d is not a rank of a world of %d", dst, c->world);
    if (channels == 0) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "channels is 0");
    if (band_height != 0 && band_height != 8 && band_height != 16)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "band_height %u: 8 or 16 (0 = 16)", band_height);
    const size_t mine = band_set_floats(height, c->rank, c->world, band_height, n_cameras, width, channels);
    if (mine && !part) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "part is NULL");
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (c->rank != dst) {  // one message: the whole band set
        if (mine) SDFV_RCCL(lib, Send(part, mine, kNcclFloat, dst, c->comm, st));
        return SDFV_OK;
    }
    if (!out) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "out is NULL on the gathering rank");
    const size_t need = sdfv_comm_gather_bands_scratch_bytes(c, dst, band_height, n_cameras, width, height, channels);
    if (need && (!scratch || scratch_bytes < need || ((uintptr_t)scratch & 15)))
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "scratch: 16-byte aligned, at least sdfv_comm_gather_bands_scratch_bytes() = %zu bytes", need);
    // every peer's set arrives whole, side by side in the scratch block (all links at once), then one launch per set moves
    // its bands to their rows; this rank's own set goes straight from `part`
    SDFV_RCCL(lib, GroupStart());
    int first_error = kNcclSuccess;
    size_t at = 0;
    for (int r = 0; r < c->world; ++r) {
        if (r == dst) continue;
        const size_t n = band_set_floats(height, r, c->world, band_height, n_cameras, width, channels);
        if (n && first_error == kNcclSuccess) first_error = lib->Recv(static_cast<float*>(scratch) + at, n, kNcclFloat, r, c->comm, st);
        at += (n + 3) & ~(size_t)3;
    }
    const int end = lib->GroupEnd();
    if (first_error == kNcclSuccess) first_error = end;
    if (first_error != kNcclSuccess) return sdfv::set_error(SDFV_ERR_COMM, "RCCL gather of the tile bands: %s", lib->GetErrorString(first_error));
    at = 0;
    for (int r = 0; r < c->world; ++r) {
        const float* src = r == dst ? part : static_cast<const float*>(scratch) + at;
        if (int rc = scatter_bands(src, (uint32_t)r, (uint32_t)c->world, band_height, n_cameras, width, height, channels, out, st)) return rc;
        if (r != dst) at += (band_set_floats(height, r, c->world, band_height, n_cameras, width, channels) + 3) & ~(size_t)3;
    }
    return SDFV_OK;
}

int sdfv_comm_gather_cameras(sdfv_slab_comm* c, const float* part, uint32_t n_cameras, uint32_t width, uint32_t height,
                             uint32_t channels, int dst, float* out, void* stream) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "communicator is NULL");
    if (dst < 0 || dst >= c->world) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "dst %d is not a rank of a world of %d", dst, c->world);
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t image = (size_t)width * height * channels;
    auto first_of = [&](int r) { return (size_t)n_cameras * r / c->world; };  // rank r renders cameras [first_of(r), first_of(r + 1))
    const size_t mine = (first_of(c->rank + 1) - first_of(c->rank)) * image;
    if (mine && !part) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "part is NULL");
    if (c->rank != dst) {
        if (mine) SDFV_RCCL(lib, Send(part, mine, kNcclFloat, dst, c->comm, st));
        return SDFV_OK;
    }
    if (!out) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "out is NULL on the gathering rank");
    SDFV_RCCL(lib, GroupStart());
    int first_error = kNcclSuccess;
    for (int r = 0; r < c->world; ++r) {  // whole cameras are contiguous in camera order: received in place
        const size_t n = (first_of(r + 1) - first_of(r)) * image;
        if (r == dst || n == 0 || first_error != kNcclSuccess) continue;
        first_error = lib->Recv(out + first_of(r) * image, n, kNcclFloat, r, c->comm, st);
    }
    const int end = lib->GroupEnd();
    if (first_error == kNcclSuccess) first_error = end;
    if (first_error != kNcclSuccess) return sdfv::set_error(SDFV_ERR_COMM, "RCCL gather of the cameras: %s", lib->GetErrorString(first_error));
    if (mine && out + first_of(dst) * image != part)
        SDFV_HIPC(hipMemcpyAsync(out + first_of(dst) * image, part, mine * sizeof(float), hipMemcpyDeviceToDevice, st));
    return SDFV_OK;
}

int sdfv_comm_allgather_slabs(sdfv_slab_comm* c, const uint32_t dims[3], const uint32_t* z_bounds, const float* tex0_owned,
                              const float* tex1_owned, const float* dist_owned, float* out0, float* out1, float* out_dist,
                              void* stream) {
    if (!c || !dims || !z_bounds) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!tex0_owned || !tex1_owned || !out0 || !out1) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if ((dist_owned == nullptr) != (out_dist == nullptr))
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "dist_owned and out_dist go together (both or neither)");
    if (z_bounds[0] != 0 || z_bounds[c->world] != dims[2]) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "z_bounds must run from 0 to the grid's depth");
    for (int r = 0; r < c->world; ++r)
        if (z_bounds[r] > z_bounds[r + 1]) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "z_bounds must not decrease");
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t slice = (size_t)dims[0] * dims[1];
    // Point-to-point only, in ONE group (the primitive the halo exchange already rests on; slabs may differ in depth, which
    // ncclAllGather does not take): this rank's slab goes to every peer, every peer's slab arrives at its place in the output;
    // this rank's own share is a local copy.
    struct Buf { const float* mine; float* out; size_t per_voxel; };
    const Buf bufs[3] = {{tex0_owned, out0, 4}, {tex1_owned, out1, 4}, {dist_owned, out_dist, 1}};
    const size_t my_z0 = z_bounds[c->rank], my_n = (size_t)(z_bounds[c->rank + 1] - z_bounds[c->rank]) * slice;
    for (const Buf& b : bufs)
        if (b.out && my_n && b.out + my_z0 * slice * b.per_voxel != b.mine)
            SDFV_HIPC(hipMemcpyAsync(b.out + my_z0 * slice * b.per_voxel, b.mine, my_n * b.per_voxel * sizeof(float), hipMemcpyDeviceToDevice, st));
    SDFV_RCCL(lib, GroupStart());
    int first_error = kNcclSuccess;
    auto post = [&](int r) {
        if (first_error == kNcclSuccess) first_error = r;
    };
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        const size_t z0 = z_bounds[r], n = (size_t)(z_bounds[r + 1] - z_bounds[r]) * slice;
        for (const Buf& b : bufs) {
            if (!b.out) continue;
            if (my_n) post(lib->Send(b.mine, my_n * b.per_voxel, kNcclFloat, r, c->comm, st));
            if (n) post(lib->Recv(b.out + z0 * slice * b.per_voxel, n * b.per_voxel, kNcclFloat, r, c->comm, st));
        }
    }
    post(lib->GroupEnd());
    if (first_error != kNcclSuccess) return sdfv::set_error(SDFV_ERR_COMM, "RCCL all-gather of the slabs: %s", lib->GetErrorString(first_error));
    return SDFV_OK;
}

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
    if (flags & ~(SDFV_COMM_PERIODIC | SDFV_COMM_HALO2))
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
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
    c->halo_hi = (flags & SDFV_COMM_HALO2) ? 2u : 1u;
    NcclUniqueId uid;
    memcpy(uid.internal, id, SDFV_COMM_ID_BYTES);
    int r = lib->CommInitRank(&c->comm, world, uid, rank);
    if (r != kNcclSuccess) {
        delete c;
        return sdfv::set_error(SDFV_ERR_COMM, "RCCL ncclCommInitRank(rank %d of %d): %s", rank, world, lib->GetErrorString(r));
    }
    // Highest priority: the exchange is short and the neighbours wait for it; HIP also keeps streams of different
    // priorities on different hardware queues, without which the exchange and the interior fill (both enqueued
    // back to back) land on one queue and run one after the other (profiles/r01/slab_step_timeline.txt).
    int prio_low = 0, prio_high = 0;
    e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_high);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->boundary_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->halo_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        sdfv_slab_comm_destroy(c);
        return sdfv::set_error(SDFV_ERR_HIP, "communicator stream/events: %s", hipGetErrorString(e));
    }
    // The start signal needs hipStreamWaitValue32 on a word of signal memory; without either, the event form.
    int dev = 0, can = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can &&
        hipExtMallocWithFlags((void**)&c->signal, 8, hipMallocSignalMemory) == hipSuccess &&
        hipMemset(c->signal, 0, 8) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
        c->can_wait_value = true;
    } else {
        (void)hipGetLastError();
        if (c->signal) (void)hipFree(c->signal);
        c->signal = nullptr;
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
    if (c->stage) (void)hipFree(c->stage);
    if (c->signal) (void)hipFree(c->signal);
    delete c;
    return SDFV_OK;
}

int sdfv_slab_comm_info(const sdfv_slab_comm* c, uint32_t* ghost_lo, uint32_t* ghost_hi, uint32_t* wait_value_capable) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "communicator is NULL");
    if (ghost_lo) *ghost_lo = c->ghost_lo();
    if (ghost_hi) *ghost_hi = c->ghost_hi();
    if (wait_value_capable) *wait_value_capable = c->can_wait_value ? 1u : 0u;
    return SDFV_OK;
}

int sdfv_slab_comm_ranks(const sdfv_slab_comm* c, int* rank, int* world) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "communicator is NULL");
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    int n = 0, r = -1;
    SDFV_RCCL(lib, CommCount(c->comm, &n));
    SDFV_RCCL(lib, CommUserRank(c->comm, &r));
    if (rank) *rank = r;
    if (world) *world = n;
    return SDFV_OK;
}

int sdfv_slab_halo_exchange(sdfv_slab_comm* c, const sdfv_grid* slab, float* tex0, float* tex1, void* stream) {
    if (int rc = check_slab(c, slab, tex0, tex1)) return rc;
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    return enqueue_exchange_direct(lib, c, slab, tex0, tex1, (hipStream_t)stream);
}

int sdfv_slab_fill_step(sdfv_slab_comm* c, const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab,
                        float* tex0, float* tex1, void* stream) {
    return sdfv_slab_fill_step_commit(c, params, sdf_id, slab, tex0, tex1, nullptr, stream);
}

int sdfv_slab_fill_step_commit(sdfv_slab_comm* c, const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab,
                               float* tex0, float* tex1, float* dist, void* stream) {
    if (int rc = check_slab(c, slab, tex0, tex1)) return rc;
    if ((uintptr_t)dist & 3) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    hipStream_t main = (hipStream_t)stream;
    const size_t slice_texels = (size_t)slab->dims[0] * slab->dims[1];
    const size_t slice = slice_texels * 4;
    const uint32_t z0 = slab->z_begin, z1 = slab->z_end, owned = z1 - z0;
    float* o0 = tex0 + c->ghost_lo() * slice;  // first owned slice
    float* o1 = tex1 + c->ghost_lo() * slice;
    float* od = dist ? dist + c->ghost_lo() * slice_texels : nullptr;  // the volume follows the textures' slice layout
    const bool exchange = c->has_lo() || c->has_hi();
    // the ghost slices' share of the distance volume = tex0.r of the slices just received, on the stream that received them
    auto ghost_distances = [&](hipStream_t st) -> int {
        if (!dist) return SDFV_OK;
        if (c->has_lo())
            if (int rc = sdfv::extract_distance(tex0, dist, slice_texels, st)) return rc;
        if (c->has_hi())
            if (int rc = sdfv::extract_distance(tex0 + (c->ghost_lo() + owned) * slice, dist + (c->ghost_lo() + owned) * slice_texels,
                                                c->halo_hi * slice_texels, st))
                return rc;
        return SDFV_OK;
    };
    const uint32_t lead = c->halo_hi;

    // Which form: the boundary launch needs whole workgroups per slice and an interior to hide the exchange behind.
    uint32_t form = sdfv::options().slab_step_form;
    bool packed = !(form & SDFV_STEP_UNPACKED);
    const bool start_event = (form & SDFV_STEP_START_EVENT) != 0 || !c->can_wait_value;
    const bool want_defer = (form & SDFV_STEP_DEFER_JOIN) != 0;
    form &= ~(SDFV_STEP_UNPACKED | SDFV_STEP_START_EVENT | SDFV_STEP_DEFER_JOIN);
    uint32_t bps = 0, total = 0;
    if (int rc = sdfv::ordered_fill_blocks(slab, &bps, &total)) return rc;
    if (!exchange || owned < lead + 2 || bps == 0) {  // nothing to hide the exchange behind / shape not supported
        sdfv_grid part = *slab;
        if (int rc = sdfv_fill_grid_commit(params, sdf_id, &part, o0, o1, od, main)) return rc;
        if (!exchange) return SDFV_OK;
        if (int rc = enqueue_exchange_direct(lib, c, slab, tex0, tex1, main)) return rc;
        return ghost_distances(main);
    }
    // Measured in loopback (tools/slab_step_probe.py): packed messages halve the exchange kernel's time, which is what
    // a short fill (256^3: 80 us) needs to hide it; a long fill (512^3: 580 us) hides either and is disturbed
    // less by per-texture messages straight into the ghosts (no staging stores, no ghost copy: 16 MB less traffic).
    if (form == 0 && (uint64_t)slice_texels * owned >= (1ull << 26)) packed = false;
    // SDFV_STEP_DEFER_JOIN: `main` is not made to wait for the exchange (sdfv_slab_comm_join does that on demand).  Only
    // with packed messages: there `main` fills every owned slice itself and the exchange touches nothing but the staging
    // block and the ghosts.  With per-texture messages the communicator's stream writes the OWNED boundary slices, which
    // work put on `main` after the step may read -- that form always joins.
    const bool defer_join = want_defer && packed;
    if (packed)
        if (int rc = ensure_staging(c, slice_texels)) return rc;

    sdfv::OrderedFill of;
    of.lead = lead;
    of.dist = od;
    if (packed) {
        of.stage_lo = c->has_lo() ? c->send_lo() : nullptr;
        of.stage_hi = c->has_hi() ? c->send_hi() : nullptr;
    }
    const uint32_t nb = (lead + 1) * bps;  // the boundary workgroups: what the neighbours wait for
    auto exchange_on = [&](hipStream_t st) -> int {
        if (packed) return enqueue_exchange_packed(lib, c, slab, tex0, tex1, dist, st);  // ghost copy + their distances in one launch
        if (int rc = enqueue_exchange_direct(lib, c, slab, tex0, tex1, st)) return rc;
        return ghost_distances(st);
    };

    // `main` runs the plain dense fill, nothing before it.  The communicator's stream starts when `main` reaches
    // this step (work enqueued earlier on `main` may still be reading what the exchange overwrites), computes the
    // boundary slices into the packed send buffers only, exchanges, and copies the received slices into the ghosts.
#ifdef SDFV_TUNING  // diagnostics of the tuning build: drop one of the two stream dependencies to price it (results undefined)
    const bool no_start = (sdfv::options().wave_timing & 1) != 0, no_wait = (sdfv::options().wave_timing & 2) != 0;
#else
    const bool no_start = false, no_wait = false;
#endif
    // How the communicator's stream learns that `main` has reached this step: the packed form's fill launch says so
    // itself (its first workgroup stores the step number to a word of signal memory the moment the launch starts,
    // and the communicator's stream waits on that word) -- no event is recorded on `main`, which then carries
    // nothing but the fill and the final wait; otherwise an event.
    const bool signal_start = packed && !start_event;
    if (signal_start) {
        // the fill goes first in host order: should the two streams ever share a hardware queue, the launch that signals
        // is ahead of the packet that waits for it
        c->step += 1;
        if (int rc = sdfv::fill_grid_signalling_start(params, sdf_id, slab, o0, o1, od, c->signal, c->step, main)) return rc;
        SDFV_HIPC(hipStreamWaitValue32(c->comm_stream, c->signal, c->step, hipStreamWaitValueGte, 0xffffffffu));
    } else if (!no_start) {
        SDFV_HIPC(hipEventRecord(c->boundary_done, main));
        SDFV_HIPC(hipStreamWaitEvent(c->comm_stream, c->boundary_done, 0));
    }
    if (packed) {
        of.stage_only = true;
#ifdef SDFV_TUNING
        if (!(sdfv::options().wave_timing & 4))  // diagnostics: drop the boundary launch
#endif
        if (int rc = sdfv::fill_slab_ordered(params, sdf_id, slab, o0, o1, of, 0, nb, c->comm_stream)) return rc;
        if (int rc = exchange_on(c->comm_stream)) return rc;
        SDFV_HIPC(hipEventRecord(c->halo_done, c->comm_stream));
        if (!signal_start)
            if (int rc = sdfv_fill_grid_commit(params, sdf_id, slab, o0, o1, od, main)) return rc;
    } else {
        // per-texture messages straight out of / into the textures: the communicator's stream fills the boundary
        // slices in place, `main` everything else (the same boundary-first grid, split between the two streams)
        if (int rc = sdfv::fill_slab_ordered(params, sdf_id, slab, o0, o1, of, 0, nb, c->comm_stream)) return rc;
        if (int rc = exchange_on(c->comm_stream)) return rc;
        SDFV_HIPC(hipEventRecord(c->halo_done, c->comm_stream));
        if (int rc = sdfv::fill_slab_ordered(params, sdf_id, slab, o0, o1, of, nb, total, main)) return rc;
    }
    if (!no_wait && !defer_join) SDFV_HIPC(hipStreamWaitEvent(main, c->halo_done, 0));
    return SDFV_OK;
}

// The sharded march with nobody in the loop but the stream: `world` rounds enqueued back to back -- a round's kernel, then ONE
// RCCL group that ships the two fixed-capacity ray buffers (count in their header) to the z-neighbours and receives theirs.
// No counter is read back, nothing synchronises; every rank enqueues the same sequence, so the groups match.
int sdfv_slab_march(sdfv_slab_comm* c, const sdfv_render_params* rp, const sdfv_grid* slab, const float* tex0,
                    const float* tex1, const sdfv_camera* camera, uint32_t width, uint32_t height, float* rgba,
                    sdfv_march_aux* aux, void* scratch, size_t scratch_bytes, uint32_t capacity, uint32_t flags,
                    uint32_t* status, void* stream) {
    if (!c || !scratch || !status) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    if (c->periodic) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "the sharded march needs a non-periodic communicator");
    if (flags & ~SDFV_MARCH_MERGE) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    if (capacity == 0) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "capacity is 0");
    const size_t one = sdfv_ray_buffer_bytes(capacity);
    const size_t one_aligned = (one + 255) & ~(size_t)255;
    if (((uintptr_t)scratch & 15) || scratch_bytes < 4 * one_aligned)
        return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "scratch: 16-byte aligned, at least sdfv_slab_march_scratch_bytes(capacity)");
    const Rccl* lib;
    if (int rc = need_rccl(lib)) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* base = static_cast<char*>(scratch);
    void *out_down = base, *out_up = base + one_aligned, *in_lo = base + 2 * one_aligned, *in_hi = base + 3 * one_aligned;
    const bool lo = c->rank > 0, hi = c->rank < c->world - 1;
    SDFV_HIPC(hipMemsetAsync(status, 0, 2 * sizeof(uint32_t), st));
    for (int round = 0; round < c->world; ++round) {
        const bool last = round == c->world - 1;
        if (int rc = sdfv_internal_raymarch_slab_round(rp, slab, c->ghost_lo(), c->ghost_hi(), tex0, tex1, camera, width, height,
                                               round == 0 ? nullptr : (lo ? in_lo : nullptr), round == 0 ? nullptr : (hi ? in_hi : nullptr),
                                               round == 0, rgba, aux, out_down, out_up, capacity, status, last ? status + 1 : nullptr, st))
            return rc;
        if (round == c->world - 1 || (!lo && !hi)) break;  // z is monotonic along a ray: after `world` rounds every ray has ended
        SDFV_RCCL(lib, GroupStart());
        int first_error = kNcclSuccess;
        auto post = [&](int r) {
            if (first_error == kNcclSuccess) first_error = r;
        };
        // whole buffers, header and all: the count travels in band, the size of a message never depends on it
        if (lo) post(lib->Send(out_down, one, kNcclInt8, c->lo_peer(), c->comm, st));
        if (hi) post(lib->Send(out_up, one, kNcclInt8, c->hi_peer(), c->comm, st));
        if (hi) post(lib->Recv(in_hi, one, kNcclInt8, c->hi_peer(), c->comm, st));
        if (lo) post(lib->Recv(in_lo, one, kNcclInt8, c->lo_peer(), c->comm, st));
        post(lib->GroupEnd());
        if (first_error != kNcclSuccess)
            return sdfv::set_error(SDFV_ERR_COMM, "RCCL ray exchange: %s", lib->GetErrorString(first_error));
    }
    // status[1] (counted by the last round's kernel): rays handed on after `world` rounds -- none, unless a list overflowed
    if (flags & SDFV_MARCH_MERGE) {
        // every pixel was written by exactly one rank and is all-zero bits elsewhere: the integer sum of the bit patterns IS
        // the image (a float sum would turn -0.0 into +0.0; RCCL has no bitwise OR)
        SDFV_RCCL(lib, AllReduce(rgba, rgba, (size_t)width * height * 4, kNcclInt32, kNcclSum, c->comm, st));
        if (aux) SDFV_RCCL(lib, AllReduce(aux, aux, (size_t)width * height * (sizeof(sdfv_march_aux) / 4), kNcclInt32, kNcclSum, c->comm, st));
    }
    return SDFV_OK;
}

size_t sdfv_slab_march_scratch_bytes(uint32_t capacity) { return 4 * ((sdfv_ray_buffer_bytes(capacity) + 255) & ~(size_t)255); }

int sdfv_slab_comm_join(sdfv_slab_comm* c, void* stream) {
    if (!c) return sdfv::set_error(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    SDFV_HIPC(hipStreamWaitEvent((hipStream_t)stream, c->halo_done, 0));  // never recorded yet: returns at once
    return SDFV_OK;
}

}  // extern "C"
#pragma GCC visibility pop
