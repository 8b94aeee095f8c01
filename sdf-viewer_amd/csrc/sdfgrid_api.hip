// sdfgrid_api.hip -- the extern "C" surface declared in include/sdfgrid.h.
// Validates arguments, builds the kernel argument blocks and enqueues the gfx950 kernels.  There is
// no CPU fallback: without a HIP device every compute entry point fails with SDFV_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "../../include/sdfgrid.h"
#include "api_internal.h"
#include "fill_kernels.h"
#include "ingest_kernels.h"
#include "mesh_kernels.h"
#include "points_kernels.h"
#include "raymarch_kernels.h"

namespace {

thread_local char g_err[512] = "";
thread_local sdfv::Options g_options;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice)
        return fail(SDFV_ERR_NO_DEVICE, "%s: no HIP device (%s)", what, hipGetErrorString(e));
    return fail(SDFV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define SDFV_HIP(call)                                   \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

bool known_sdf(uint32_t id) { return id <= SDFV_SDF_SPHERE; }

int check_params(const sdfv_demo_params* p, uint32_t sdf_id) {
    if (!p) return fail(SDFV_ERR_INVALID_ARGUMENT, "params is NULL");
    if (!known_sdf(sdf_id)) {
        fprintf(stderr, "Failed to find SDF with ID %u\n", sdf_id);  // ffi.rs:47
        return fail(SDFV_ERR_UNKNOWN_SDF, "Failed to find SDF with ID %u", sdf_id);
    }
    if (p->cube_material > SDFV_MATERIAL_NORMAL || p->sphere_material > SDFV_MATERIAL_NORMAL)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "Invalid cube material");  // cube.rs:33
    return SDFV_OK;
}

int check_grid(const sdfv_grid* g) {
    if (!g) return fail(SDFV_ERR_INVALID_ARGUMENT, "grid is NULL");
    if (g->z_begin > g->z_end || g->z_end > g->dims[2])
        return fail(SDFV_ERR_INVALID_ARGUMENT, "slab [%u,%u) outside depth %u", g->z_begin, g->z_end, g->dims[2]);
    return SDFV_OK;
}

// Every entry point that reinterprets a float* as float4* (dwordx4 loads and stores) checks this first: a misaligned
// caller buffer is an argument error, not a GPU fault.
int check_texel_alignment(const void* a, const void* b = nullptr, const void* c = nullptr) {
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "texel buffers (tex0, tex1, rgba) must be 16-byte aligned");
    return SDFV_OK;
}

int need_device() {
    static int cached = 0;  // only a positive answer is cached
    if (cached > 0) return SDFV_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(SDFV_ERR_NO_DEVICE, "no HIP device visible: libsdfgrid has no CPU path");
    }
    cached = n;
    return SDFV_OK;
}

float air_dist() {
    volatile float a = 1e-1f, b = 0.001234f;  // scene/sdf/mod.rs:42, f32 const evaluation
    return a + b;
}

sdfv::FillArgs make_fill_args(const sdfv_demo_params& p, uint32_t sdf_id, const sdfv_grid& g, float* tex0,
                              float* tex1) {
    sdfv::FillArgs a;
    memset(&a, 0, sizeof(a));
    a.prm = p;
    a.sdf_id = sdf_id;
    a.W = g.dims[0];
    a.H = g.dims[1];
    a.D = g.dims[2];
    a.z_begin = g.z_begin;
    a.slab_d = g.z_end - g.z_begin;
    for (int i = 0; i < 3; ++i) {
        a.dm1[i] = (float)g.dims[i] - 1.0f;           // scene/sdf/mod.rs:168
        a.bb_size[i] = g.bb_max[i] - g.bb_min[i];      // scene/sdf/mod.rs:167
        a.bb_min[i] = g.bb_min[i];
    }
    a.air_dist = air_dist();
    a.z_step = 1;
    a.srgb_round = g_options.ext_srgb_quant;
    a.tex0 = reinterpret_cast<float4*>(tex0);
    a.tex1 = reinterpret_cast<float4*>(tex1);
    return a;
}

// Store policy and index form of the dense fill (SDFV_OPT_FILL_*).  Store policy "auto": a launch that also writes the
// compact distance volume streams the two textures past L2 (nt) -- nothing re-reads them before the march's few texels
// under the hits, while the distance volume, which the march gathers from, keeps its place in the caches: the fused
// fill itself runs 6 % faster (0.097 -> 0.091 ms at 256^3) and fill + march 0.188 -> 0.177 ms (EXPERIMENTS, round 2 store-policy probe).
// The plain fill keeps plain stores (nt: within noise alone, +3 % on the 256^3 pipeline, -3 % on the 512^3 one).
struct DeviceFacts;
const DeviceFacts& device_facts();
static bool device_has_eight_xcds();
sdfv::FillLaunch fill_launch_config(bool writes_distance_volume) {
    sdfv::FillLaunch c;
    c.nontemporal = g_options.fill_nontemporal == 1 || (g_options.fill_nontemporal == 0 && writes_distance_volume);
    c.force_rows = g_options.fill_form == 1;
    c.force_flat = g_options.fill_form == 2;
    c.force_paired = g_options.fill_form == 3;
    c.force_pairrows = g_options.fill_form == 4;
    // the interleaved-volume fill with the rows of a pair on one XCD rests on workgroup b -> XCD b % 8 (as the march's tile orders do)
    c.xcd_pairing = device_has_eight_xcds();
    return c;
}

// The per-thread device caches below belong to the HIP device that was current when they were allocated.  A thread
// that later calls in with another device current (one process driving several GPUs) must not launch kernels on
// device B over pointers of device A: each cache records its owner and is released and re-made on a mismatch.
int current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return d;
}

// What the raymarch launcher needs to know about the device it launches on (cached per host thread and device): the
// hand-written march loop is gfx950 machine code, the XCD-aware tile order assumes workgroup L -> XCD L % xcds, and the
// occupancy rule counts wave slots.
struct DeviceFacts {
    int device = -2;
    bool gfx950 = false;
    uint32_t cus = 0, xcds = 0;
    uint64_t last_level_cache_bytes = 0;
};
const DeviceFacts& device_facts() {
    thread_local DeviceFacts f;
    const int now = current_device();
    if (f.device == now) return f;
    f = DeviceFacts();
    f.device = now;
    hipDeviceProp_t prop;
    if (now >= 0 && hipGetDeviceProperties(&prop, now) == hipSuccess) {
        f.gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        f.cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 0u;
        int x = 0;
        if (hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, now) == hipSuccess && x > 0) f.xcds = (uint32_t)x;
        // MI355X: 256 MB of Infinity Cache behind the eight 4 MB L2s (MI355X_MICROARCH.md); HIP reports the L2 only
        f.last_level_cache_bytes = f.gfx950 ? (256ull << 20) : (uint64_t)(prop.l2CacheSize > 0 ? prop.l2CacheSize : 0);
    } else {
        (void)hipGetLastError();
    }
    return f;
}

static bool device_has_eight_xcds() { return device_facts().xcds == 8; }

struct MeshScratch {
    void* p = nullptr;
    size_t bytes = 0;
    int device = -1;
};
// A batch of more than kMaxCamerasPerLaunch cameras is several launches (the cameras ride in the kernel-argument block).  When
// each is small -- low-resolution views, a rank's share of a split batch -- it is as long as its longest waves, not as its
// work, and running them one after the other multiplies that: such launches go to side streams forked from and joined back
// into the caller's stream (events only: capturable), where they overlap.  Created on first use, per thread and device.
struct BatchStreams {
    static constexpr int kSide = 3;
    hipStream_t side[kSide] = {};
    hipEvent_t fork = nullptr, join[kSide] = {};
    int device = -1;
    void release() {
        for (int i = 0; i < kSide; ++i) {
            if (side[i]) (void)hipStreamDestroy(side[i]);
            if (join[i]) (void)hipEventDestroy(join[i]);
        }
        if (fork) (void)hipEventDestroy(fork);
        *this = BatchStreams{};
    }
    hipError_t ensure(int device_now) {
        if (device == device_now && fork) return hipSuccess;
        release();
        hipError_t e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
        for (int i = 0; i < kSide && e == hipSuccess; ++i) {
            e = hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&join[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) release();
        else device = device_now;
        return e;
    }
};
thread_local BatchStreams g_batch_streams;  // released by sdfv_mesh_trim(); a thread that exits without it leaks three streams

// Cameras of a launch of more than kInlineCameras that arrive as a HOST array (the kernel-argument block carries 16): the
// launch reads them from a slot of this per-thread ring of device memory, written on the launch's own stream by small
// kernels that carry 32 cameras each as THEIR arguments -- no copy engine, no allocation per call, nothing the host waits
// for.  A slot is reused every kSlots launches; the event recorded behind its last reader orders the rewrite after it,
// whatever stream that reader ran on.  Allocated on first use (122 KB), released by sdfv_mesh_trim().
struct CameraRing {
    static constexpr uint32_t kSlots = 16;
    sdfv_camera* base = nullptr;  // kSlots x kMaxCamerasPerLaunch
    hipEvent_t read[kSlots] = {};
    bool recorded[kSlots] = {};
    uint32_t next = 0;
    int device = -1;
    void release() {
        for (uint32_t i = 0; i < kSlots; ++i)
            if (read[i]) (void)hipEventDestroy(read[i]);
        if (base) (void)hipFree(base);
        *this = CameraRing{};
    }
    hipError_t ensure(int device_now) {
        if (device == device_now && base) return hipSuccess;
        release();
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), (size_t)kSlots * sdfv::kMaxCamerasPerLaunch * sizeof(sdfv_camera));
        for (uint32_t i = 0; i < kSlots && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&read[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            release();
        } else {
            device = device_now;
        }
        return e;
    }
};
thread_local CameraRing g_camera_ring;
thread_local MeshScratch g_mesh_scratch;  // freed by sdfv_mesh_trim(); a thread that exits without it leaks the block

// Per-point callers (the reference's ffi.rs ABI: one sample() per call) would otherwise pay two hipMalloc/hipFree per
// point.  Small requests of the *_host conveniences run through a per-thread staging area instead: a device block
// and a pinned host block allocated once, asynchronous copies on the null stream, one synchronisation.
struct SmallStage {
    static constexpr size_t kBytes = 64 << 10;  // in + out of up to 1024 points
    char* dev = nullptr;
    char* host = nullptr;
    int device = -1;
    bool ensure() {
        const int now = current_device();
        if (dev && device != now) {  // the staging block lives on another GPU than the one this call runs on
            (void)hipFree(dev);
            dev = nullptr;
        }
        device = now;
        if (dev && host) return true;
        if (!dev && hipMalloc((void**)&dev, kBytes) != hipSuccess) dev = nullptr;
        if (!host && hipHostMalloc((void**)&host, kBytes, hipHostMallocDefault) != hipSuccess) host = nullptr;
        return dev && host;
    }
};
thread_local SmallStage g_stage;  // never freed: lives as long as the thread's device context

struct DeviceBuf {
    void* p = nullptr;
    ~DeviceBuf() {
        if (p) (void)hipFree(p);
    }
};

// Runs `enqueue(dev_in, dev_out)` over host buffers: in_bytes up, out_bytes down.  Small requests use the per-thread
// staging area, large ones a pair of temporary device buffers.
template <typename Enqueue>
int run_over_host_buffers(const void* in_host, size_t in_bytes, void* out_host, size_t out_bytes, Enqueue enqueue) {
    const size_t in_up = (in_bytes + 255) & ~(size_t)255;
    if (in_up + out_bytes <= SmallStage::kBytes && g_stage.ensure()) {
        memcpy(g_stage.host, in_host, in_bytes);
        SDFV_HIP(hipMemcpyAsync(g_stage.dev, g_stage.host, in_bytes, hipMemcpyHostToDevice, nullptr));
        if (int rc = enqueue(g_stage.dev, g_stage.dev + in_up)) return rc;
        SDFV_HIP(hipMemcpyAsync(g_stage.host + in_up, g_stage.dev + in_up, out_bytes, hipMemcpyDeviceToHost, nullptr));
        SDFV_HIP(hipStreamSynchronize(nullptr));
        memcpy(out_host, g_stage.host + in_up, out_bytes);
        return SDFV_OK;
    }
    DeviceBuf din, dout;
    SDFV_HIP(hipMalloc(&din.p, in_bytes));
    SDFV_HIP(hipMalloc(&dout.p, out_bytes));
    SDFV_HIP(hipMemcpy(din.p, in_host, in_bytes, hipMemcpyHostToDevice));
    if (int rc = enqueue(din.p, dout.p)) return rc;
    SDFV_HIP(hipMemcpy(out_host, dout.p, out_bytes, hipMemcpyDeviceToHost));
    return SDFV_OK;
}

// Everything RaymarchArgs derives from the render parameters alone (exactness flags, tap sizes, cull sphere).
void derive_raymarch_args(const sdfv_render_params* rp, sdfv::RaymarchArgs& a) {
    memset(&a, 0, sizeof(a));
    a.rp = *rp;
    const uint32_t off = g_options.raymarch_disable;
    a.pow2_extent = (off & SDFV_RM_NO_POW2_EXTENT) ? 0u : 1u;
    a.fast_index = (off & SDFV_RM_NO_FAST_INDEX) ? 0u : 1u;
    a.pow2_size = 1;
    a.symmetric_box = 1;
    a.fast_normal = 1;
    float inv_h2 = 0.0f, radius2 = 0.0f;
    for (int i = 0; i < 3; ++i) {
        const float s = (float)rp->tex_size[i] / rp->lod_dist_between_samples;
        inv_h2 += s * s;
    }
    const float h_world = 1.0f / sqrtf(inv_h2);  // sdfNormal's tap offset, material.frag:74
    for (int i = 0; i < 3; ++i) {
        a.bsize[i] = rp->bounds_max[i] - rp->bounds_min[i];
        a.inv_bsize[i] = 1.0f / a.bsize[i];
        const float n = (float)rp->tex_size[i];
        // a marching ray stays within 1e-4 of the box (material.frag:106): floor(u) in [-1, N-1] needs
        // 1e-4 * N / size well below 0.5; the normal's taps sit h_world further out
        if (!(a.bsize[i] > 0.0f) || !(1e-4f * n / a.bsize[i] <= 0.25f)) a.fast_index = 0;
        if (!(a.bsize[i] > 0.0f) || !((1e-4f + h_world) * n / a.bsize[i] <= 0.45f)) a.fast_normal = 0;
        int e = 0;
        // x / 2^k == x * 2^-k exactly (both correctly rounded), provided 2^-k is itself normal
        if (!(a.bsize[i] > 0.0f) || frexpf(a.bsize[i], &e) != 0.5f || e < -100 || e > 100) a.pow2_extent = 0;
        if (rp->tex_size[i] >= (1u << 24)) a.pow2_size = 0;  // (N as a float is exact below 2^24)
        if (rp->bounds_min[i] != -rp->bounds_max[i]) a.symmetric_box = 0;
        a.cull_center[i] = 0.5f * (rp->bounds_min[i] + rp->bounds_max[i]);
        radius2 += 0.25f * a.bsize[i] * a.bsize[i];
    }
    a.cull_radius2 = radius2 * 1.0201f + 1e-12f;  // (1.01 r)^2
    if (!(a.cull_radius2 > 0.0f) || !std::isfinite(a.cull_radius2)) a.cull_radius2 = INFINITY;  // never cull
    a.asm_loop = (off & SDFV_RM_NO_ASM_LOOP) ? 0u : 1u;
    a.no_interior_fetch = (off & SDFV_RM_NO_INTERIOR_FETCH) ? 1u : 0u;
    a.cube_box = (a.symmetric_box && rp->bounds_max[0] == rp->bounds_max[1] && rp->bounds_max[1] == rp->bounds_max[2]) ? 1u : 0u;
    if (off & SDFV_RM_NO_SYMMETRIC) a.symmetric_box = 0;
    if (off & SDFV_RM_NO_POW2_SIZE) a.pow2_size = 0;
}

// The light list: the scene's ambient light (rp->ambient) plus rp->lights.  Only ambient entries can be rendered.
int check_lights(const sdfv_render_params* rp) {
    if (rp->n_lights > SDFV_MAX_LIGHTS)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "n_lights %u exceeds SDFV_MAX_LIGHTS", rp->n_lights);
    for (uint32_t i = 0; i < rp->n_lights; ++i) {
        if (rp->lights[i].kind == SDFV_LIGHT_DIRECTIONAL)
            return fail(SDFV_ERR_INVALID_ARGUMENT,
                        "lights[%u] is directional: three-d 0.18.2 shader source not available (scene/mod.rs:107-112 keeps "
                        "its directional lights commented out; their BRDF is not restated from memory)", i);
        if (rp->lights[i].kind != SDFV_LIGHT_AMBIENT)
            return fail(SDFV_ERR_INVALID_ARGUMENT, "lights[%u] has unknown kind %u", i, rp->lights[i].kind);
    }
    return SDFV_OK;
}

}  // namespace

namespace sdfv {
const Options& options() { return g_options; }

namespace {
std::mutex g_rccl_path_mutex;
std::string g_rccl_path;         // SDFV_OPT_RCCL_LIBRARY (process-wide: RCCL is loaded once per process)
std::atomic<bool> g_rccl_loaded{false};
}  // namespace
const char* rccl_library_path() {
    std::lock_guard<std::mutex> lock(g_rccl_path_mutex);
    return g_rccl_path.c_str();  // (stable: the string is not modified once RCCL has been loaded)
}
std::string rccl_library_path_copy() {
    std::lock_guard<std::mutex> lock(g_rccl_path_mutex);
    return g_rccl_path;
}
bool rccl_loaded() { return g_rccl_loaded.load(); }
const char* claim_rccl_library_path() {
    std::lock_guard<std::mutex> lock(g_rccl_path_mutex);
    g_rccl_loaded.store(true);
    return g_rccl_path.c_str();
}
static bool set_rccl_library_path(const char* path) {
    std::lock_guard<std::mutex> lock(g_rccl_path_mutex);
    if (g_rccl_loaded.load()) return false;  // (decided under the same lock the loader claims the path under)
    g_rccl_path = path ? path : "";
    return true;
}
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int ordered_fill_blocks(const sdfv_grid* slab, uint32_t* per_slice, uint32_t* total) {
    if (int rc = check_grid(slab)) return rc;
    sdfv_demo_params prm;
    sdfv_demo_params_default(&prm);
    const FillArgs a = make_fill_args(prm, SDFV_SDF_DEMO, *slab, nullptr, nullptr);
    const OrderedBlocks b = ordered_blocks(a);
    *per_slice = b.per_slice;
    *total = b.total;
    return SDFV_OK;
}

int fill_slab_ordered(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab, float* o0, float* o1,
                      const OrderedFill& of, uint32_t block_begin, uint32_t block_end, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = check_grid(slab)) return rc;
    if (int rc = check_texel_alignment(o0, o1)) return rc;
    if (int rc = check_texel_alignment(of.stage_lo, of.stage_hi)) return rc;
    if (int rc = need_device()) return rc;
    FillArgs a = make_fill_args(*params, sdf_id, *slab, o0, o1);
    a.order_lead = of.lead;
    a.stage_only = of.stage_only ? 1u : 0u;
    a.dist = of.dist;
    a.stage_lo = reinterpret_cast<float4*>(of.stage_lo);
    a.stage_hi = reinterpret_cast<float4*>(of.stage_hi);
    SDFV_HIP(launch_fill_dense_ordered(a, block_begin, block_end, (hipStream_t)stream));
    return SDFV_OK;
}

int copy_texel_segments(const float* const src[4], float* const dst[4], const size_t n[4], float* const r_out[4],
                        void* stream) {
    if (int rc = need_device()) return rc;
    CopySegments c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < 4; ++i) {
        if (n[i] >= (1ull << 32)) return fail(SDFV_ERR_INVALID_ARGUMENT, "segment too large");
        if (n[i] == 0) continue;
        if (int rc = check_texel_alignment(src[i], dst[i])) return rc;  // the kernel moves float4 texels
        if (r_out && ((uintptr_t)r_out[i] & 3)) return fail(SDFV_ERR_INVALID_ARGUMENT, "r_out must be 4-byte aligned");
        c.src[i] = reinterpret_cast<const float4*>(src[i]);
        c.dst[i] = reinterpret_cast<float4*>(dst[i]);
        c.n[i] = (uint32_t)n[i];
        c.r_out[i] = r_out ? r_out[i] : nullptr;
    }
    SDFV_HIP(launch_copy_segments(c, (hipStream_t)stream));
    return SDFV_OK;
}

int extract_distance(const float* tex0, float* dist, size_t n, void* stream) {
    if (n == 0) return SDFV_OK;
    if (int rc = check_texel_alignment(tex0)) return rc;  // read as float4 texels
    if ((uintptr_t)dist & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    SDFV_HIP(launch_commit_distance(tex0, dist, n, (hipStream_t)stream));
    return SDFV_OK;
}

int fill_grid_signalling_start(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* grid, float* tex0,
                               float* tex1, float* dist, uint32_t* signal, uint32_t value, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = check_grid(grid)) return rc;
    if (!tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (int rc = check_texel_alignment(tex0, tex1)) return rc;
    if (int rc = need_device()) return rc;
    if ((uintptr_t)dist & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    FillArgs a = make_fill_args(*params, sdf_id, *grid, tex0, tex1);
    a.dist = dist;
    a.signal = signal;
    a.signal_value = value;
    if ((uint64_t)a.H * a.slab_d > 0x7fffffffull || a.W > 0x7fffffffu)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "slab of %u x %u rows is too large for one launch", a.H, a.slab_d);
    SDFV_HIP(launch_fill_dense(a, fill_launch_config(dist != nullptr), (hipStream_t)stream));
    return SDFV_OK;
}

int fill_boundary_slices(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab, float* o0, float* o1,
                         void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = check_grid(slab)) return rc;
    if (slab->z_end - slab->z_begin < 2) return fail(SDFV_ERR_INVALID_ARGUMENT, "a slab of one slice has one boundary");
    if (int rc = check_texel_alignment(o0, o1)) return rc;
    if (int rc = need_device()) return rc;
    FillArgs a = make_fill_args(*params, sdf_id, *slab, o0, o1);
    a.z_step = a.slab_d - 1;
    a.slab_d = 2;
    SDFV_HIP(launch_fill_slices(a, (hipStream_t)stream));
    return SDFV_OK;
}
}  // namespace sdfv

#pragma GCC visibility push(default)
static int raymarch_rows(const sdfv_render_params* rp, const float* tex0, const float* tex1, const float* dist,
                         const float* pairs, const float* ilv, const sdfv_camera* cameras, uint32_t n_cameras, uint32_t width,
                         uint32_t height, uint32_t y0, uint32_t y1, uint32_t band_step, uint32_t band_height, float* rgba, float* depth,
                         sdfv_march_aux* aux, uint32_t* rgba8, void* stream);

extern "C" {

uint32_t sdfv_abi_version(void) { return SDFV_ABI_VERSION; }

const char* sdfv_last_error(void) { return g_err; }

#ifndef SDFV_BUILD_ID
#define SDFV_BUILD_ID "unknown"
#endif
const char* sdfv_build_id(void) { return SDFV_BUILD_ID; }

int sdfv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

float sdfv_air_dist(void) { return air_dist(); }

int sdfv_set_option(uint32_t option, uint64_t value) {
    switch (option) {
        case SDFV_OPT_FILL_NONTEMPORAL:
            if (value > 2) break;
            g_options.fill_nontemporal = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_FILL_FORM:
            if (value > 4) break;
            g_options.fill_form = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_DISABLE:
            if (value & ~(uint64_t)(SDFV_RM_NO_FAST_INDEX | SDFV_RM_NO_POW2_EXTENT | SDFV_RM_NO_POW2_SIZE | SDFV_RM_NO_SYMMETRIC |
                                     SDFV_RM_NO_ASM_LOOP | SDFV_RM_NO_INTERIOR_FETCH)) break;
            g_options.raymarch_disable = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_KEEP_NORMAL:
            if (value > 1) break;
            g_options.raymarch_keep_normal = value != 0;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_TILE_GROUP:
            if (value > 5) break;
            g_options.raymarch_tile_group = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_BOX_FIRST:
            if (value > 1) break;
            g_options.raymarch_box_first = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_BATCH_STREAMS:
            if (value > 1) break;
            g_options.raymarch_batch_streams = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_CAMERA_STAGING:
            if (value > 1) break;
            g_options.raymarch_camera_staging = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_SLAB_STEP_FORM: {
            const uint64_t form = value & ~(uint64_t)(SDFV_STEP_UNPACKED | SDFV_STEP_START_EVENT | SDFV_STEP_DEFER_JOIN);
            if (form != 0 && form != SDFV_STEP_SIDE_BOUNDARY) break;
            g_options.slab_step_form = (uint32_t)value;
            return SDFV_OK;
        }
        case SDFV_OPT_PASS_INDEX_LIMIT:
            if (value != 0 && (value < 2 || value > (1ull << 32))) break;
            g_options.pass_index_limit = value;
            return SDFV_OK;
        case SDFV_OPT_EXT_SRGB_QUANT:
            if (value > 1) break;
            g_options.ext_srgb_quant = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RCCL_LIBRARY: {
            // value = address of a NUL-terminated path (copied), or 0 = librccl.so.1 by name; process-wide, before the first
            // communicator -- RCCL is loaded once
            const char* path = reinterpret_cast<const char*>((uintptr_t)value);
            if (path && strnlen(path, 4096) >= 4096) break;
            if (!sdfv::set_rccl_library_path(path))
                return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_OPT_RCCL_LIBRARY: RCCL has already been loaded in this process");
            return SDFV_OK;
        }
        case SDFV_OPT_PASS_FORM:
            if (value > 1) break;
            g_options.pass_form = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_PASS_LOADS:
            if (value > 2) break;
            g_options.pass_loads = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_RAYMARCH_WAVES_PER_SIMD:
            if (value == 1 || value > 7) break;  // 0 auto (the launcher's rule) | 2..6 cap | 7 never cap
            g_options.raymarch_waves_per_simd = (uint32_t)value;
            return SDFV_OK;
        case SDFV_OPT_TUNING_TILE_ORDER:
#ifdef SDFV_TUNING
            g_options.tile_order = value;
            return SDFV_OK;
#else
            return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_OPT_TUNING_TILE_ORDER needs the tuning build (make tuning)");
#endif
        case SDFV_OPT_TUNING_PRIORITY_MAP:
#ifdef SDFV_TUNING
            g_options.priority_map = value;
            return SDFV_OK;
#else
            return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_OPT_TUNING_PRIORITY_MAP needs the tuning build (make tuning)");
#endif
        case SDFV_OPT_TUNING_WAVE_TIMING:
#ifdef SDFV_TUNING
            g_options.wave_timing = value;
            return SDFV_OK;
#else
            return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_OPT_TUNING_WAVE_TIMING needs the tuning build (make tuning)");
#endif
        default:
            return fail(SDFV_ERR_INVALID_ARGUMENT, "unknown option %u", option);
    }
    return fail(SDFV_ERR_INVALID_ARGUMENT, "value %llu is out of range for option %u", (unsigned long long)value, option);
}

int sdfv_get_option(uint32_t option, uint64_t* value) {
    if (!value) return fail(SDFV_ERR_INVALID_ARGUMENT, "value is NULL");
    switch (option) {
        case SDFV_OPT_FILL_NONTEMPORAL: *value = g_options.fill_nontemporal; return SDFV_OK;
        case SDFV_OPT_FILL_FORM: *value = g_options.fill_form; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_DISABLE: *value = g_options.raymarch_disable; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_KEEP_NORMAL: *value = g_options.raymarch_keep_normal; return SDFV_OK;
        case SDFV_OPT_SLAB_STEP_FORM: *value = g_options.slab_step_form; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_TILE_GROUP: *value = g_options.raymarch_tile_group; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_BOX_FIRST: *value = g_options.raymarch_box_first; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_BATCH_STREAMS: *value = g_options.raymarch_batch_streams; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_CAMERA_STAGING: *value = g_options.raymarch_camera_staging; return SDFV_OK;
        case SDFV_OPT_TUNING_WAVE_TIMING: *value = g_options.wave_timing; return SDFV_OK;
        case SDFV_OPT_TUNING_PRIORITY_MAP: *value = g_options.priority_map; return SDFV_OK;
        case SDFV_OPT_TUNING_TILE_ORDER: *value = g_options.tile_order; return SDFV_OK;
        case SDFV_OPT_RAYMARCH_WAVES_PER_SIMD: *value = g_options.raymarch_waves_per_simd; return SDFV_OK;
        case SDFV_OPT_EXT_SRGB_QUANT: *value = g_options.ext_srgb_quant; return SDFV_OK;
        case SDFV_OPT_PASS_INDEX_LIMIT: *value = g_options.pass_index_limit; return SDFV_OK;
        case SDFV_OPT_PASS_FORM: *value = g_options.pass_form; return SDFV_OK;
        case SDFV_OPT_PASS_LOADS: *value = g_options.pass_loads; return SDFV_OK;
        case SDFV_OPT_RCCL_LIBRARY: {  // address of a copy that belongs to the CALLING THREAD ("" = by name): valid until this
            thread_local std::string copy;  // thread asks again, whatever another thread sets meanwhile
            copy = sdfv::rccl_library_path_copy();
            *value = (uint64_t)(uintptr_t)copy.c_str();
            return SDFV_OK;
        }
        default: return fail(SDFV_ERR_INVALID_ARGUMENT, "unknown option %u", option);
    }
}

void sdfv_demo_params_default(sdfv_demo_params* p) {
    if (!p) return;
    p->cube_half_side = 0.95f;                // cube.rs:17
    p->cube_material = SDFV_MATERIAL_BRICK;   // cube.rs:15
    p->sphere_radius = 1.05f;                 // sphere.rs:13
    p->sphere_material = SDFV_MATERIAL_NORMAL;  // sphere.rs:11
    p->max_distance_custom_material = 0.05f;  // demo/mod.rs:26
    p->disable_sphere = 0;                    // demo/mod.rs:28
}

int sdfv_grid_from_bb(const float bb_min[3], const float bb_max[3], uint32_t max_voxels_side, sdfv_grid* out) {
    if (!bb_min || !bb_max || !out) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    // scene/sdf/mod.rs:46-72: the longest axis gets max_voxels_side (ties: the LAST longest axis, as
    // Iterator::max_by), the others (N as f32 * size_i / size_max) as usize.
    float size[3] = {bb_max[0] - bb_min[0], bb_max[1] - bb_min[1], bb_max[2] - bb_min[2]};
    int max_dim = 0;
    for (int i = 1; i < 3; ++i)
        if (size[i] >= size[max_dim]) max_dim = i;
    for (int i = 0; i < 3; ++i) {
        if (i == max_dim) {
            out->dims[i] = max_voxels_side;
        } else {
            float v = (float)max_voxels_side * size[i] / size[max_dim];
            out->dims[i] = v > 0.0f ? (uint32_t)v : 0u;
        }
        out->bb_min[i] = bb_min[i];
        out->bb_max[i] = bb_max[i];
    }
    out->z_begin = 0;
    out->z_end = out->dims[2];
    return SDFV_OK;
}

void sdfv_render_params_default(sdfv_render_params* rp, const sdfv_grid* grid) {
    if (!rp) return;
    memset(rp, 0, sizeof(*rp));
    for (int i = 0; i < 3; ++i) {
        if (grid) {
            rp->bounds_min[i] = grid->bb_min[i];
            rp->bounds_max[i] = grid->bb_max[i];
            rp->tex_size[i] = grid->dims[i];
        }
        rp->ambient[i] = 1.0f;  // AmbientLight::new(&ctx, 1.0, Srgba::WHITE), scene/mod.rs:106
    }
    rp->lod_dist_between_samples = 1.0f;  // material.rs:28
    rp->tint[0] = rp->tint[1] = rp->tint[2] = rp->tint[3] = 1.0f;  // Srgba::WHITE, material.rs:29
    rp->gamma = 0.0f;
    rp->tone_mapping = 2;   // three-d 0.18 default (ACES)
    rp->color_mapping = 1;  // three-d 0.18 default (compute to sRGB)
}

int sdfv_camera_look_at(sdfv_camera* cam, const float eye[3], const float target[3], const float up[3],
                        float fovy_degrees, float aspect, float z_near, float z_far) {
    if (!cam || !eye || !target || !up) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    // cgmath look_at_rh: f = normalize(target - eye), s = normalize(f x up), u = s x f
    float f[3] = {target[0] - eye[0], target[1] - eye[1], target[2] - eye[2]};
    float fl = 1.0f / sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& c : f) c *= fl;
    float s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    float sl = 1.0f / sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (float& c : s) c *= sl;
    float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    for (int i = 0; i < 3; ++i) {
        cam->eye[i] = eye[i];
        cam->right[i] = s[i];
        cam->up[i] = u[i];
        cam->forward[i] = f[i];
    }
    cam->tan_half_fovy = tanf(fovy_degrees * (3.14159265358979323846f / 180.0f) / 2.0f);
    cam->aspect = aspect;
    const float view[16] = {s[0], u[0], -f[0], 0.0f, s[1], u[1], -f[1], 0.0f, s[2], u[2], -f[2], 0.0f,
                            -(eye[0] * s[0] + eye[1] * s[1] + eye[2] * s[2]),
                            -(eye[0] * u[0] + eye[1] * u[1] + eye[2] * u[2]),
                            (eye[0] * f[0] + eye[1] * f[1] + eye[2] * f[2]), 1.0f};
    const float ct = 1.0f / cam->tan_half_fovy;
    const float proj[16] = {ct / aspect, 0, 0, 0, 0, ct, 0, 0,
                            0, 0, (z_far + z_near) / (z_near - z_far), -1.0f,
                            0, 0, (2.0f * z_far * z_near) / (z_near - z_far), 0};
    const float bias[16] = {0.5f, 0, 0, 0, 0, 0.5f, 0, 0, 0, 0, 0.5f, 0, 0.5f, 0.5f, 0.5f, 1.0f};  // material.rs:90-95
    float pv[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += proj[k * 4 + r] * view[c * 4 + k];
            pv[c * 4 + r] = acc;
        }
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += bias[k * 4 + r] * pv[c * 4 + k];
            cam->bvp[c * 4 + r] = acc;
        }
    return SDFV_OK;
}

int sdfv_grid_init(const sdfv_grid* grid, float* tex0, float* tex1, void* stream) {
    if (int rc = check_grid(grid)) return rc;
    if (!tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (int rc = check_texel_alignment(tex0, tex1)) return rc;
    if (int rc = need_device()) return rc;
    const uint64_t n = (uint64_t)grid->dims[0] * grid->dims[1] * (grid->z_end - grid->z_begin);
    SDFV_HIP(sdfv::launch_grid_init(tex0, tex1, n, air_dist(), (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_grid_init_unvisited(const sdfv_grid* grid, uint32_t step, float* tex0, float* tex1, float* dist, void* stream) {
    return sdfv_grid_init_unvisited_ex(grid, step, tex0, tex1, dist, 0u, stream);
}

int sdfv_grid_init_unvisited_ex(const sdfv_grid* grid, uint32_t step, float* tex0, float* tex1, float* dist, uint32_t flags,
                                void* stream) {
    if (int rc = check_grid(grid)) return rc;
    if (flags & ~SDFV_PASS_VOLUME_INTERLEAVED) return fail(SDFV_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    if ((flags & SDFV_PASS_VOLUME_INTERLEAVED) && dist && ((grid->dims[1] & 1u) || ((uintptr_t)dist & 7)))
        return fail(SDFV_ERR_INVALID_ARGUMENT, "the interleaved volume pairs rows: H even, 8-byte aligned");
    if (!tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (step & (step - 1)) return fail(SDFV_ERR_INVALID_ARGUMENT, "step %u is neither 0 nor a power of two", step);
    if (int rc = check_texel_alignment(tex0, tex1)) return rc;
    if ((uintptr_t)dist & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    if (step == 1) return SDFV_OK;  // a step-1 pass wrote every row
    SDFV_HIP(sdfv::launch_grid_init_unvisited(tex0, tex1, dist, grid->dims[0], grid->dims[1], grid->z_begin,
                                              grid->z_end - grid->z_begin, step, air_dist(),
                                              (flags & SDFV_PASS_VOLUME_INTERLEAVED) ? 1u : 0u, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_fill_grid_commit(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* grid, float* tex0,
                          float* tex1, float* dist, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = check_grid(grid)) return rc;
    if (!tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (int rc = check_texel_alignment(tex0, tex1)) return rc;
    if ((uintptr_t)dist & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    sdfv::FillArgs a = make_fill_args(*params, sdf_id, *grid, tex0, tex1);
    a.dist = dist;
    if ((uint64_t)a.H * a.slab_d > 0x7fffffffull || a.W > 0x7fffffffu)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "slab of %u x %u rows is too large for one launch", a.H, a.slab_d);
    SDFV_HIP(sdfv::launch_fill_dense(a, fill_launch_config(dist != nullptr), (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_fill_grid(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* grid, float* tex0, float* tex1,
                   void* stream) {
    return sdfv_fill_grid_commit(params, sdf_id, grid, tex0, tex1, nullptr, stream);
}

int sdfv_fill_grid_pass_ex(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* grid, uint32_t step,
                           const float* changed_box, float* tex0, float* tex1, float* dist, uint32_t flags, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = check_grid(grid)) return rc;
    if (!tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (step == 0 || (step & (step - 1))) return fail(SDFV_ERR_INVALID_ARGUMENT, "step %u is not a power of two", step);
    if (flags & ~(SDFV_PASS_FRESH_GRID | SDFV_PASS_SAME_LOAD | SDFV_PASS_VIRGIN_GRID | SDFV_PASS_VOLUME_INTERLEAVED | SDFV_PASS_EXPECT_NOOP))
        return fail(SDFV_ERR_INVALID_ARGUMENT, "unknown pass flags 0x%x", flags);
    if (flags & SDFV_PASS_VOLUME_INTERLEAVED) {
        if (!dist) return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_PASS_VOLUME_INTERLEAVED without a volume");
        if ((grid->dims[1] & 1u) || ((uintptr_t)dist & 7))
            return fail(SDFV_ERR_INVALID_ARGUMENT, "the interleaved volume pairs rows: H = %u must be even and the volume 8-byte aligned", grid->dims[1]);
    }
    if ((flags & SDFV_PASS_VIRGIN_GRID) && changed_box)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_PASS_VIRGIN_GRID with a changed box: a box test reads the grid (sdfv_grid_init_unvisited first)");
    if (int rc = check_texel_alignment(tex0, tex1)) return rc;
    if ((uintptr_t)dist & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    sdfv::FillArgs a = make_fill_args(*params, sdf_id, *grid, tex0, tex1);
    a.dist_ilv = (flags & SDFV_PASS_VOLUME_INTERLEAVED) ? 1u : 0u;
    // what the caller KNOWS about the grid (the layout bit and the hint say nothing about update_required)
    const uint32_t knowledge = flags & ~(SDFV_PASS_VOLUME_INTERLEAVED | SDFV_PASS_EXPECT_NOOP);
    sdfv::PassArgs p;
    memset(&p, 0, sizeof(p));
    p.step = step;
    p.nx = (a.W + step - 1) / step;  // loading.rs:82: ceil(limit / step) visits per axis
    p.ny = (a.H + step - 1) / step;
    p.z_first = ((grid->z_begin + step - 1) / step) * step;
    p.nz = p.z_first < grid->z_end ? (grid->z_end - p.z_first + step - 1) / step : 0;
    p.has_box = changed_box != nullptr;
    p.dist = dist;
    if (changed_box) memcpy(p.box, changed_box, sizeof(p.box));
    // Does update_required (scene/sdf/mod.rs:184-190) hold for EVERY visited voxel?  Then the pass reads nothing.
    //  * the caller says so (flags): a fresh grid is all AIR_DIST; within one load a stored voxel already holds what this pass
    //    would write;
    //  * the changed box contains every voxel of the slab -- decided on the voxels' own first and last coordinates per axis
    //    (the kernels' arithmetic: idx / (dim - 1) * size + min, three roundings; monotonic in idx, so the ends decide; a NaN
    //    coordinate fails the comparison and keeps the general path).  The demo reports its whole bounding box on every
    //    parameter edit (demo/mod.rs:135-144), so its edits take this path.
    bool covers = changed_box != nullptr;
    for (int i = 0; i < 3 && covers; ++i) {
        const uint32_t first_idx = i == 2 ? grid->z_begin : 0u, last_idx = i == 2 ? grid->z_end - 1 : grid->dims[i] - 1;
        float first = (float)first_idx / a.dm1[i];
        first = first * a.bb_size[i];
        first = first + a.bb_min[i];
        float last = (float)last_idx / a.dm1[i];
        last = last * a.bb_size[i];
        last = last + a.bb_min[i];
        covers = first >= changed_box[i] && first <= changed_box[3 + i] && last >= changed_box[i] && last <= changed_box[3 + i];
    }
    p.fresh = (flags & SDFV_PASS_FRESH_GRID) ? 1u : 0u;
    p.virgin = (flags & SDFV_PASS_VIRGIN_GRID) ? 1u : 0u;
    p.index_limit = g_options.pass_index_limit;
    p.no_adaptive = g_options.pass_form == 1 ? 1u : 0u;
    // The scan's loads.  Auto: nontemporal when the caller expects a no-op pass AND the volume is larger than the last-level
    // cache -- a smaller one is still resident from the fill that wrote it, and cached loads hit (same box, tools/pass_loads_ab.py:
    // no-op pass at 512^3 0.128 -> 0.085 ms with nt loads, at 256^3 0.0128 -> 0.0150; passes that update most voxels +7..13 %).
    {
        const uint64_t volume_bytes = (uint64_t)a.W * a.H * a.slab_d * 4u, llc = device_facts().last_level_cache_bytes;
        const bool hinted = (flags & SDFV_PASS_EXPECT_NOOP) != 0 && (llc == 0 || volume_bytes > llc);
        p.stream_loads = g_options.pass_loads == 0 ? (hinted ? 1u : 0u) : (g_options.pass_loads == 2 ? 1u : 0u);
    }
    p.all_required = (knowledge != 0 || covers) ? 1u : 0u;
    SDFV_HIP(sdfv::launch_fill_pass(a, p, fill_launch_config(dist != nullptr), (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_pack_samples(const sdfv_grid* grid, uint64_t index_base, const uint32_t* indices, const sdfv_sample* samples, size_t n,
                      float* tex0, float* tex1, float* dist, uint32_t flags, void* stream) {
    if (int rc = check_grid(grid)) return rc;
    if (!tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (n && !samples) return fail(SDFV_ERR_INVALID_ARGUMENT, "samples is NULL");
    if (flags & ~SDFV_PASS_VOLUME_INTERLEAVED) return fail(SDFV_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
    if (flags & SDFV_PASS_VOLUME_INTERLEAVED) {
        if (!dist) return fail(SDFV_ERR_INVALID_ARGUMENT, "SDFV_PASS_VOLUME_INTERLEAVED without a volume");
        if ((grid->dims[1] & 1u) || ((uintptr_t)dist & 7))
            return fail(SDFV_ERR_INVALID_ARGUMENT, "the interleaved volume pairs rows: H = %u must be even and the volume 8-byte aligned", grid->dims[1]);
    }
    if (int rc = check_texel_alignment(tex0, tex1)) return rc;
    if (((uintptr_t)dist | (uintptr_t)samples | (uintptr_t)indices) & 3)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "samples, indices and dist must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    sdfv::PackArgs a;
    memset(&a, 0, sizeof(a));
    a.samples = samples;
    a.indices = indices;
    a.index_base = index_base;
    a.n = n;
    a.n_voxels = (uint64_t)grid->dims[0] * grid->dims[1] * (grid->z_end - grid->z_begin);
    a.W = grid->dims[0];
    a.tex0 = reinterpret_cast<float4*>(tex0);
    a.tex1 = tex1;
    a.dist = dist;
    a.dist_ilv = (flags & SDFV_PASS_VOLUME_INTERLEAVED) ? 1u : 0u;
    a.srgb_round = g_options.ext_srgb_quant;
    SDFV_HIP(sdfv::launch_pack_samples(a, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_sample_points(const sdfv_demo_params* params, uint32_t sdf_id, const float* points, size_t n,
                       int distance_only, sdfv_sample* out, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (n && (!points || !out)) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = need_device()) return rc;
    SDFV_HIP(sdfv::launch_sample_points(*params, sdf_id, points, n, distance_only != 0, out, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_normal_points(const sdfv_demo_params* params, uint32_t sdf_id, const float* points, size_t n, float eps,
                       int use_default, float* out, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (n && (!points || !out)) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = need_device()) return rc;
    SDFV_HIP(sdfv::launch_normal_points(*params, sdf_id, nullptr, nullptr, points, n, eps, use_default != 0, out,
                                        (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_source_sample_scalar(const sdfv_demo_params* params, uint32_t sdf_id, const float bb_min[3],
                              const float bb_max[3], const float* unit_points, size_t n, float* dist_out, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (!bb_min || !bb_max) return fail(SDFV_ERR_INVALID_ARGUMENT, "bounding box is NULL");
    if (n && (!unit_points || !dist_out)) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = need_device()) return rc;
    SDFV_HIP(sdfv::launch_source_scalar(*params, sdf_id, bb_min, bb_max, unit_points, n, dist_out, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_source_sample_normal(const sdfv_demo_params* params, uint32_t sdf_id, const float bb_min[3],
                              const float bb_max[3], const float* unit_points, size_t n, float* normal_out,
                              void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (!bb_min || !bb_max) return fail(SDFV_ERR_INVALID_ARGUMENT, "bounding box is NULL");
    if (n && (!unit_points || !normal_out)) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = need_device()) return rc;
    SDFV_HIP(sdfv::launch_normal_points(*params, sdf_id, bb_min, bb_max, unit_points, n, 0.0f, false, normal_out,
                                        (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_mesh_postproc(const sdfv_demo_params* params, uint32_t sdf_id, sdfv_vertex* vertices, size_t n, void* stream) {
    if (int rc = check_params(params, sdf_id)) return rc;
    if (n && !vertices) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if ((uintptr_t)vertices & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "vertices must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    SDFV_HIP(sdfv::launch_mesh_postproc(*params, sdf_id, vertices, n, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_mesh_extract(const sdfv_demo_params* params, uint32_t sdf_id, const float bb_min[3], const float bb_max[3],
                      uint32_t max_voxels_per_axis, uint32_t algorithm, sdfv_mesh* out, void* stream) {
    if (!out) return fail(SDFV_ERR_INVALID_ARGUMENT, "out is NULL");
    memset(out, 0, sizeof(*out));
    if (int rc = check_params(params, sdf_id)) return rc;
    if (!bb_min || !bb_max) return fail(SDFV_ERR_INVALID_ARGUMENT, "bounding box is NULL");
    if (algorithm != SDFV_MESHER_MARCHING_CUBES)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "Unsupported algorithm %u", algorithm);  // isosurface.rs:49
    if (max_voxels_per_axis < 1 || max_voxels_per_axis > 1024)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "max_voxels_per_axis %u is outside [1, 1024]", max_voxels_per_axis);
    if (int rc = need_device()) return rc;
    hipStream_t st = (hipStream_t)stream;
    sdfv::MeshGrid g;
    for (int i = 0; i < 3; ++i) {
        g.cells[i] = max_voxels_per_axis;
        g.bb_min[i] = bb_min[i];
        g.bb_size[i] = bb_max[i] - bb_min[i];
    }
    const size_t n_points = (size_t)(g.cells[0] + 1) * (g.cells[1] + 1) * (g.cells[2] + 1);
    const size_t n_cells = (size_t)g.cells[0] * g.cells[1] * g.cells[2];
    // One scratch block per host thread, grown on demand and kept between calls (allocating ~13 B per lattice point
    // afresh costs more than the extraction itself); sdfv_mesh_trim() gives it back.
    sdfv::MeshWork w{};
    w.scan_tmp_bytes = sdfv::mesh_scan_tmp_bytes(n_points);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_dist = 0, o_first = o_dist + up(n_points * 4), o_cfirst = o_first + up(n_points * 4),
                 o_mask = o_cfirst + up(n_cells * 4), o_tmp = o_mask + up(n_points), o_totals = o_tmp + up(w.scan_tmp_bytes),
                 need = o_totals + 256;
    const int device_now = current_device();
    if (g_mesh_scratch.bytes < need || g_mesh_scratch.device != device_now) {
        if (g_mesh_scratch.p) (void)hipFree(g_mesh_scratch.p);
        g_mesh_scratch = MeshScratch{};
        SDFV_HIP(hipMalloc(&g_mesh_scratch.p, need));
        g_mesh_scratch.bytes = need;
        g_mesh_scratch.device = device_now;
    }
    char* base = static_cast<char*>(g_mesh_scratch.p);
    w.dist = (float*)(base + o_dist);
    w.point_first = (uint32_t*)(base + o_first);
    w.cell_first = (uint32_t*)(base + o_cfirst);
    w.point_mask = (uint8_t*)(base + o_mask);
    w.scan_tmp = base + o_tmp;
    struct { void* p; } totals{base + o_totals};
    SDFV_HIP(sdfv::launch_mesh_count(*params, sdf_id, g, w, (uint32_t*)totals.p, st));
    uint32_t n[2] = {0, 0};
    SDFV_HIP(hipMemcpyAsync(n, totals.p, 8, hipMemcpyDeviceToHost, st));
    SDFV_HIP(hipStreamSynchronize(st));
    sdfv_mesh m{};
    m.n_vertices = n[0];
    m.n_indices = (size_t)n[1] * 3;
    if (m.n_vertices) SDFV_HIP(hipMalloc((void**)&m.vertices, m.n_vertices * sizeof(sdfv_vertex)));
    if (m.n_indices) {
        hipError_t e = hipMalloc((void**)&m.indices, m.n_indices * 4);
        if (e != hipSuccess) {
            (void)hipFree(m.vertices);
            return hip_fail(e, "hipMalloc(indices)");
        }
    }
    hipError_t e = sdfv::launch_mesh_emit(*params, sdf_id, g, w, m.vertices, m.indices, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the next extraction on this thread reuses the scratch
    if (e != hipSuccess) {
        (void)hipFree(m.vertices);
        (void)hipFree(m.indices);
        return hip_fail(e, "mesh emit");
    }
    *out = m;
    return SDFV_OK;
}

int sdfv_mesh_trim(void) {
    if (g_mesh_scratch.p) (void)hipFree(g_mesh_scratch.p);
    g_mesh_scratch = MeshScratch{};
    g_batch_streams.release();
    g_camera_ring.release();
    return SDFV_OK;
}

int sdfv_mesh_free(sdfv_mesh* mesh) {
    if (!mesh) return SDFV_OK;
    if (mesh->vertices) (void)hipFree(mesh->vertices);
    if (mesh->indices) (void)hipFree(mesh->indices);
    memset(mesh, 0, sizeof(*mesh));
    return SDFV_OK;
}

int sdfv_commit_distance(const sdfv_grid* grid, const float* tex0, float* dist, void* stream) {
    if (int rc = check_grid(grid)) return rc;
    if (!tex0 || !dist) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = check_texel_alignment(tex0)) return rc;
    if ((uintptr_t)dist & 3) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist must be 4-byte aligned");
    if (int rc = need_device()) return rc;
    const uint64_t n = (uint64_t)grid->dims[0] * grid->dims[1] * (grid->z_end - grid->z_begin);
    SDFV_HIP(sdfv::launch_commit_distance(tex0, dist, n, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_commit_pairs(const sdfv_grid* grid, const float* dist, float* pairs, void* stream) {
    if (int rc = check_grid(grid)) return rc;
    if (!dist || !pairs) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist or pairs is NULL");
    if (((uintptr_t)dist & 3) || ((uintptr_t)pairs & 7)) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist: 4-byte, pairs: 8-byte aligned");
    if (grid->z_begin != 0 || grid->z_end != grid->dims[2])
        return fail(SDFV_ERR_INVALID_ARGUMENT, "the pair volume is built over the whole grid (the march reads the whole grid)");
    if (int rc = need_device()) return rc;
    const uint64_t n = (uint64_t)grid->dims[0] * grid->dims[1] * grid->dims[2];
    SDFV_HIP(sdfv::launch_commit_pairs(dist, pairs, grid->dims[0], grid->dims[1], n, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_commit_interleaved(const sdfv_grid* grid, const float* dist, float* ilv, void* stream) {
    if (int rc = check_grid(grid)) return rc;
    if (!dist || !ilv) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist or ilv is NULL");
    if (((uintptr_t)dist & 3) || ((uintptr_t)ilv & 7)) return fail(SDFV_ERR_INVALID_ARGUMENT, "dist: 4-byte, ilv: 8-byte aligned");
    if (grid->z_begin != 0 || grid->z_end != grid->dims[2])
        return fail(SDFV_ERR_INVALID_ARGUMENT, "the interleaved volume is built over the whole grid");
    if (grid->dims[1] & 1) return fail(SDFV_ERR_INVALID_ARGUMENT, "the interleaved volume pairs rows: H = %u is odd", grid->dims[1]);
    if (int rc = need_device()) return rc;
    const uint64_t n = (uint64_t)grid->dims[0] * grid->dims[1] * grid->dims[2];
    SDFV_HIP(sdfv::launch_commit_interleaved(dist, ilv, grid->dims[0], n, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_march_volume_advice(const sdfv_grid* grid, uint32_t* kind) {
    if (int rc = check_grid(grid)) return rc;
    if (!kind) return fail(SDFV_ERR_INVALID_ARGUMENT, "kind is NULL");
    if (int rc = need_device()) return rc;
    const uint64_t n = (uint64_t)grid->dims[0] * grid->dims[1] * grid->dims[2];
    const uint64_t llc = device_facts().last_level_cache_bytes;
    // only the hand-written gfx950 loop reads these volumes -- on another device, or with that loop switched off, a
    // host would allocate 4-8 B/voxel and re-run the commit after every fill for a volume no march ever touches.  ADVICE r04:
    // and the loop's OTHER conditions (power-of-two extents, symmetric box, clamp-for-mirror, its addressing limits) are
    // asked of the launcher itself, over the render parameters a march of this grid would carry (sdfv_render_params_default).
    sdfv_render_params rp;
    memset(&rp, 0, sizeof(rp));
    for (int i = 0; i < 3; ++i) rp.tex_size[i] = grid->dims[i], rp.bounds_min[i] = grid->bb_min[i], rp.bounds_max[i] = grid->bb_max[i];
    rp.lod_dist_between_samples = 1.0f;
    sdfv::RaymarchArgs a;
    derive_raymarch_args(&rp, a);
    if (!device_facts().gfx950) a.asm_loop = 0;
    const bool cubic = grid->dims[0] == grid->dims[1] && grid->dims[1] == grid->dims[2];
    const bool pairs_ok = sdfv::march_volume_applicable(a, 3), ilv_ok = sdfv::march_volume_applicable(a, 4);
    if (!cubic)
        *kind = SDFV_MARCH_VOLUME_NONE;  // the two-gather cell fetch is the cubic grid's: the distance volume marches fastest here
    else if (ilv_ok && (!pairs_ok || (llc && n * 8u > llc)))
        *kind = SDFV_MARCH_VOLUME_INTERLEAVED;
    else
        *kind = pairs_ok ? SDFV_MARCH_VOLUME_PAIRS : SDFV_MARCH_VOLUME_NONE;
    return SDFV_OK;
}

int sdfv_raymarch_ex(const sdfv_march_desc* desc, void* stream) {
    if (!desc) return fail(SDFV_ERR_INVALID_ARGUMENT, "desc is NULL");
    // size-prefixed: read what the caller's header knew, take the rest as 0 / NULL (a binder built against an older header)
    if (desc->size < offsetof(sdfv_march_desc, depth))
        return fail(SDFV_ERR_INVALID_ARGUMENT, "sdfv_march_desc.size = %u is smaller than the descriptor's first version (%zu)",
                    desc->size, offsetof(sdfv_march_desc, depth));
    sdfv_march_desc d;
    memset(&d, 0, sizeof(d));
    memcpy(&d, desc, desc->size < sizeof(d) ? desc->size : sizeof(d));
    if (d.reserved != 0) return fail(SDFV_ERR_INVALID_ARGUMENT, "sdfv_march_desc: reserved fields must be 0");
    // a NEWER caller's descriptor: fields this library does not know may only be zero (0 / NULL = "not used" by the
    // descriptor's rule); anything else would be dropped silently
    for (uint32_t i = (uint32_t)sizeof(d); i < desc->size; ++i)
        if (reinterpret_cast<const unsigned char*>(desc)[i] != 0)
            return fail(SDFV_ERR_INVALID_ARGUMENT, "sdfv_march_desc.size = %u: this library knows %zu bytes and byte %u beyond them is not 0",
                        desc->size, sizeof(d), i);
    if (d.band_step == 0 && d.band_height != 0)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "band_height %u with band_step 0: a band set needs band_step >= 1 (0 = rows [y0, y1))", d.band_height);
    if (d.band_step == 0)
        return raymarch_rows(d.rp, d.tex0, d.tex1, d.dist, d.pairs, d.ilv, d.cameras, d.n_cameras, d.width, d.height, d.y0, d.y1, 1, 16,
                             d.rgba, d.depth, d.aux, d.rgba8, stream);
    const uint32_t B = d.band_height ? d.band_height : 16u;
    if (B != 8u && B != 16u) return fail(SDFV_ERR_INVALID_ARGUMENT, "band_height %u: 8 or 16 (0 = 16)", d.band_height);
    if (sdfv_band_rows_ex(d.height, d.band_first, d.band_step, B) == 0) return SDFV_OK;  // a band set below the image: nothing to render
    return raymarch_rows(d.rp, d.tex0, d.tex1, d.dist, d.pairs, d.ilv, d.cameras, d.n_cameras, d.width, d.height, d.band_first * B,
                         d.height, d.band_step, B, d.rgba, d.depth, d.aux, d.rgba8, stream);
}

uint32_t sdfv_band_rows_ex(uint32_t height, uint32_t band_first, uint32_t band_step, uint32_t band_height) {
    const uint32_t B = band_height ? band_height : 16u;
    if (B != 8u && B != 16u) return 0;
    const uint32_t bands = (height + B - 1) / B;
    if (band_step == 0 || band_first >= bands) return 0;
    const uint32_t n = (bands - band_first + band_step - 1) / band_step, last = band_first + (n - 1) * band_step;
    return (n - 1) * B + (height - last * B < B ? height - last * B : B);
}
uint32_t sdfv_band_rows(uint32_t height, uint32_t band_first, uint32_t band_step) {
    return sdfv_band_rows_ex(height, band_first, band_step, 16u);
}

}  // extern "C"
// Rows [y0, y1) of the image (band_step 1), or the 16-row bands y0 / 16, y0 / 16 + band_step, ... below y1 == height.
static int raymarch_rows(const sdfv_render_params* rp, const float* tex0, const float* tex1, const float* dist,
                         const float* pairs, const float* ilv, const sdfv_camera* cameras, uint32_t n_cameras, uint32_t width,
                         uint32_t height, uint32_t y0, uint32_t y1, uint32_t band_step, uint32_t band_height, float* rgba, float* depth,
                         sdfv_march_aux* aux, uint32_t* rgba8, void* stream) {
    if (!rp || !tex0 || !tex1) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!rgba && !rgba8) return fail(SDFV_ERR_INVALID_ARGUMENT, "no colour output: rgba and rgba8 are both NULL");
    if (int rc = check_lights(rp)) return rc;
    if (int rc = check_texel_alignment(tex0, tex1, rgba)) return rc;
    if ((uintptr_t)dist & 3 || (uintptr_t)depth & 3 || (uintptr_t)aux & 3 || (uintptr_t)rgba8 & 3)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "dist, depth, aux and rgba8 must be 4-byte aligned");
    if (((uintptr_t)pairs | (uintptr_t)ilv) & 7) return fail(SDFV_ERR_INVALID_ARGUMENT, "pairs and ilv must be 8-byte aligned");
    if (n_cameras && !cameras) return fail(SDFV_ERR_INVALID_ARGUMENT, "cameras is NULL");
    if (y0 > y1 || y1 > height) return fail(SDFV_ERR_INVALID_ARGUMENT, "rows [%u,%u) outside height %u", y0, y1, height);
    if (rp->tex_size[0] == 0 || rp->tex_size[1] == 0 || rp->tex_size[2] == 0)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "empty texture");
    if (!(rp->lod_dist_between_samples >= 1.0f)) return fail(SDFV_ERR_INVALID_ARGUMENT, "lod_dist_between_samples < 1");
    if ((uint64_t)rp->tex_size[0] * rp->tex_size[1] * rp->tex_size[2] >= (1ull << 32))
        return fail(SDFV_ERR_INVALID_ARGUMENT, "texture too large for 32-bit texel indexing");
    if (int rc = need_device()) return rc;
    sdfv::RaymarchArgs a;
    derive_raymarch_args(rp, a);
    a.dist = dist;
    a.pairs = pairs;
    a.ilv = ilv;
    a.tex0 = reinterpret_cast<const float4*>(tex0);
    a.tex1 = reinterpret_cast<const float4*>(tex1);
    a.width = width;
    a.height = height;
    a.y0 = y0;
    a.y1 = y1;
    a.band_skip = band_height * (band_step - 1u);
    a.band_shift = band_height == 8u ? 3u : 4u;
    a.rows_out = band_step > 1 ? sdfv_band_rows_ex(height, y0 / band_height, band_step, band_height) : y1 - y0;
    // sdfNormal's result only feeds calculate_lighting (material.frag:155,163), and the one AmbientLight the scene
    // configures (scene/mod.rs:106-112) does not read it: dead code a GLSL compiler removes.  It is evaluated when the
    // aux record asks for it; SDFV_OPT_RAYMARCH_KEEP_NORMAL evaluates it per hit regardless (what it would cost once a
    // directional light uses it).
    a.compute_normal = g_options.raymarch_keep_normal ? 1u : 0u;
    // tile order: auto = for a single frame the launcher's XCD-aware choice (groups of 2 x 2 tiles, those under the projected
    // bounding box first; profiles/r02/box_first_*.json, EXPERIMENTS R2-R3), launch order for batches of cameras, which
    // lose 3-8 % with any grouping
    // w waves per SIMD = w workgroups per CU: each asks for a w-th of the CU's 160 KB of LDS (less a little for rounding)
    const uint32_t w = g_options.raymarch_waves_per_simd;
    a.waves_per_simd = w;  // 0 = the launcher's occupancy rule (launch_raymarch), 7 = never cap, 2..6 = that cap
    a.box_first = g_options.raymarch_box_first;
    // The device's facts gate what is machine- or topology-specific: the hand-written loop is gfx950 code (anything else
    // takes the compiler's loop, bit-identical), and the XCD-aware tile orders assume the observed workgroup -> XCD
    // placement L % 8 of an eight-XCD part (speed only; launch order elsewhere).
    const DeviceFacts& dev = device_facts();
    if (!dev.gfx950) a.asm_loop = 0;
    a.wave_slots_per_simd_unit = dev.cus * 4u;  // SIMDs: resident waves at w per SIMD = w * this
    a.last_level_cache_bytes = dev.last_level_cache_bytes;
    const bool xcd_order_ok = dev.xcds == 8;
    a.group_shift = g_options.raymarch_tile_group == 0 ? (n_cameras == 1 && xcd_order_ok ? sdfv::kGroupAuto : 0u)
                                                       : (g_options.raymarch_tile_group == 1 ? 0u : g_options.raymarch_tile_group - 1u);
    // launches that keep launch order (batches, tile bands): tile columns rotated by row and camera, so that the XCDs' static
    // shares balance (auto only: SDFV_OPT_RAYMARCH_TILE_GROUP 1 is the plain launch order, for A/B runs)
    a.rotate_columns = g_options.raymarch_tile_group == 0 && xcd_order_ok ? 1u : 0u;
#ifdef SDFV_TUNING
    a.wave_timing = reinterpret_cast<unsigned long long*>(g_options.wave_timing);  // 32 B per wave, or 0
    a.priority_map = reinterpret_cast<const unsigned char*>(g_options.priority_map);
    a.tile_order = n_cameras == 1 ? reinterpret_cast<const uint32_t*>(g_options.tile_order) : nullptr;
#endif
    const uint64_t pixels_per_cam = (uint64_t)a.rows_out * width;
    hipStream_t main = (hipStream_t)stream;
    // Cameras.  Up to kInlineCameras ride in the kernel-argument block.  A larger batch is read from device memory, so that
    // kMaxCamerasPerLaunch of them still make ONE launch: the caller's own array if that is where it lies, else a slot of the
    // camera ring per launch (CameraRing).  While the stream is being captured (the ring's events do not belong in somebody's
    // graph), with the option off, or if the ring cannot be had, the batch goes out as launches of kInlineCameras instead
    // (same pixels).
    a.camera_list = nullptr;
    const sdfv_camera* device_cameras = nullptr;
    bool ring = false;
    // (asked for ANY count -- ADVICE r04: a device array of <= kInlineCameras cameras, e.g. a rank's share of a resident batch,
    // used to be memcpy'd from on the host; the kernel reads camera_list for any count)
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, cameras) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)) {
        device_cameras = cameras;
    } else {
        (void)hipGetLastError();
        if (n_cameras > sdfv::kInlineCameras) {
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(main, &capturing) != hipSuccess) (void)hipGetLastError();
            ring = capturing == hipStreamCaptureStatusNone && g_options.raymarch_camera_staging &&
                   g_camera_ring.ensure(current_device()) == hipSuccess;
        }
    }
    const uint32_t per_launch = device_cameras || ring ? sdfv::kMaxCamerasPerLaunch : sdfv::kInlineCameras;
    // several launches, each small (tools/split_balance.py: a launch of 17 000 workgroups gains 34 % from overlapping with its
    // siblings, one of 35 000 11 %, one of 65 000 nothing, one of 130 000 loses 4 %): side streams, see BatchStreams
    const uint64_t groups_per_launch = (uint64_t)((width + 15) / 16) * ((a.rows_out + 15) / 16) * per_launch;
    const bool overlap = n_cameras > per_launch && groups_per_launch <= 40000 && g_options.raymarch_batch_streams;
    uint32_t used = 0;  // side streams that carry a launch of this call
    int rc = SDFV_OK;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == SDFV_OK) rc = hip_fail(e, what);
        return e == hipSuccess;
    };
    if (overlap) hip_ok(g_batch_streams.ensure(current_device()), "batch streams") && hip_ok(hipEventRecord(g_batch_streams.fork, main), "hipEventRecord");
    uint32_t launch = 0;
    for (uint32_t c0 = 0; c0 < n_cameras && rc == SDFV_OK; c0 += per_launch, ++launch) {
        const uint32_t nc = n_cameras - c0 < per_launch ? n_cameras - c0 : per_launch;
        a.n_cameras = nc;
        if (device_cameras) a.camera_list = device_cameras + c0;
        else if (!ring) memcpy(a.cameras, cameras + c0, nc * sizeof(sdfv_camera));
        a.rgba = rgba ? reinterpret_cast<float4*>(rgba) + c0 * pixels_per_cam : nullptr;
        a.rgba8 = rgba8 ? rgba8 + c0 * pixels_per_cam : nullptr;
        a.aux = aux ? aux + c0 * pixels_per_cam : nullptr;
        a.depth = depth ? depth + c0 * pixels_per_cam : nullptr;
        hipStream_t on = main;
        const uint32_t lane = launch % (BatchStreams::kSide + 1);  // 0 = the caller's stream
        if (overlap && lane != 0) {
            on = g_batch_streams.side[lane - 1];
            if (lane > used) {
                if (!hip_ok(hipStreamWaitEvent(on, g_batch_streams.fork, 0), "hipStreamWaitEvent")) break;
                used = lane;
            }
        }
        if (ring) {
            CameraRing& r = g_camera_ring;
            const uint32_t slot = r.next++ % CameraRing::kSlots;
            sdfv_camera* at = r.base + (size_t)slot * sdfv::kMaxCamerasPerLaunch;
            if (r.recorded[slot] && !hip_ok(hipStreamWaitEvent(on, r.read[slot], 0), "hipStreamWaitEvent")) break;
            if (!hip_ok(sdfv::launch_store_cameras(cameras + c0, nc, at, on), "store_cameras")) break;
            a.camera_list = at;
            hip_ok(sdfv::launch_raymarch(a, on), "launch_raymarch") && hip_ok(hipEventRecord(r.read[slot], on), "hipEventRecord");
            r.recorded[slot] = true;  // (a failed record leaves the event as it was: waiting on it is harmless)
            continue;
        }
        hip_ok(sdfv::launch_raymarch(a, on), "launch_raymarch");
    }
    for (uint32_t i = 0; i < used; ++i)  // (joined even after an error: the side streams must not run on behind the caller's)
        hip_ok(hipEventRecord(g_batch_streams.join[i], g_batch_streams.side[i]), "hipEventRecord") &&
            hip_ok(hipStreamWaitEvent(main, g_batch_streams.join[i], 0), "hipStreamWaitEvent");
    return rc;
}
extern "C" {

// Argument checks and RaymarchArgs / SlabMarchArgs shared by the two forms of a round of the sharded march.
static int slab_round_args(const sdfv_render_params* rp, const sdfv_grid* slab, uint32_t ghost_lo, uint32_t ghost_hi,
                           const float* tex0, const float* tex1, const sdfv_camera* camera, uint32_t width, uint32_t height,
                           float* rgba, sdfv_march_aux* aux, sdfv::RaymarchArgs& a, sdfv::SlabMarchArgs& s) {
    if (!rp || !slab || !tex0 || !tex1 || !camera || !rgba) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    if (int rc = check_grid(slab)) return rc;
    if (int rc = check_lights(rp)) return rc;
    if (int rc = check_texel_alignment(tex0, tex1, rgba)) return rc;
    for (int i = 0; i < 3; ++i)
        if (rp->tex_size[i] != slab->dims[i])
            return fail(SDFV_ERR_INVALID_ARGUMENT, "render parameters describe a %ux%ux%u grid, the slab a %ux%ux%u one",
                        rp->tex_size[0], rp->tex_size[1], rp->tex_size[2], slab->dims[0], slab->dims[1], slab->dims[2]);
    if (slab->z_begin >= slab->z_end) return fail(SDFV_ERR_INVALID_ARGUMENT, "empty slab");
    if (ghost_lo > slab->z_begin || slab->z_end + ghost_hi > slab->dims[2])
        return fail(SDFV_ERR_INVALID_ARGUMENT, "ghost slices reach outside the grid");
    if (ghost_lo > 1 || ghost_hi > 2) return fail(SDFV_ERR_INVALID_ARGUMENT, "at most 1 lower and 2 upper ghost slices");
    if (slab->z_end < slab->dims[2] && ghost_hi == 0)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "an interior slab needs its upper ghost slice (run the halo exchange)");
    if (rp->lod_dist_between_samples != 1.0f)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "the sharded march renders loaded grids only (lod_dist_between_samples == 1)");
    if ((uint64_t)width * height >= (1ull << 32)) return fail(SDFV_ERR_INVALID_ARGUMENT, "image too large");
    const uint64_t resident = (uint64_t)slab->dims[0] * slab->dims[1] * (ghost_lo + (slab->z_end - slab->z_begin) + ghost_hi);
    if (resident >= (1ull << 32)) return fail(SDFV_ERR_INVALID_ARGUMENT, "slab too large for 32-bit texel indexing");
    if (int rc = need_device()) return rc;
    derive_raymarch_args(rp, a);
    if (!a.fast_index)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "box too small for the sharded march: 1e-4 * N / size must be <= 0.25");
    a.tex0 = reinterpret_cast<const float4*>(tex0);
    a.tex1 = reinterpret_cast<const float4*>(tex1);
    a.width = width;
    a.height = height;
    a.y0 = 0;
    a.y1 = height;
    a.rows_out = height;
    a.band_skip = 0;
    a.band_shift = 4;
    a.n_cameras = 1;
    a.camera_list = nullptr;
    a.cameras[0] = *camera;
    a.rgba = reinterpret_cast<float4*>(rgba);
    a.aux = aux;
    s = sdfv::SlabMarchArgs{};
    s.z_lo = slab->z_begin - ghost_lo;
    s.z_count = ghost_lo + (slab->z_end - slab->z_begin) + ghost_hi;
    s.own_begin = slab->z_begin;
    s.own_end = slab->z_end;
    return SDFV_OK;
}

int sdfv_raymarch_slab(const sdfv_render_params* rp, const sdfv_grid* slab, uint32_t ghost_lo, uint32_t ghost_hi,
                       const float* tex0, const float* tex1, const sdfv_camera* camera, uint32_t width, uint32_t height,
                       const sdfv_ray_state* in_states, uint32_t n_in, float* rgba, sdfv_march_aux* aux,
                       sdfv_ray_state* out_down, sdfv_ray_state* out_up, uint32_t capacity, uint32_t* counters,
                       void* stream) {
    if (!out_down || !out_up || !counters) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    if (in_states == nullptr && n_in != 0) return fail(SDFV_ERR_INVALID_ARGUMENT, "n_in without in_states");
    sdfv::RaymarchArgs a;
    sdfv::SlabMarchArgs s;
    if (int rc = slab_round_args(rp, slab, ghost_lo, ghost_hi, tex0, tex1, camera, width, height, rgba, aux, a, s)) return rc;
    s.in = in_states;
    s.n_in = n_in;
    s.out_down = out_down;
    s.out_up = out_up;
    s.count_down = counters;
    s.count_up = counters + 1;
    s.capacity = capacity;
    if (in_states && n_in == 0) return SDFV_OK;  // nothing arrived this round
    SDFV_HIP(sdfv::launch_raymarch_slab(a, s, (hipStream_t)stream));
    return SDFV_OK;
}

size_t sdfv_ray_buffer_bytes(uint32_t capacity) { return SDFV_RAY_BUFFER_HEADER_BYTES + (size_t)capacity * sizeof(sdfv_ray_state); }

int sdfv_raymarch_slab_round(const sdfv_render_params* rp, const sdfv_grid* slab, uint32_t ghost_lo, uint32_t ghost_hi,
                             const float* tex0, const float* tex1, const sdfv_camera* camera, uint32_t width, uint32_t height,
                             const void* in_lo, const void* in_hi, int first_round, float* rgba, sdfv_march_aux* aux,
                             void* out_down, void* out_up, uint32_t capacity, uint32_t* overflow, void* stream) {
    return sdfv_internal_raymarch_slab_round(rp, slab, ghost_lo, ghost_hi, tex0, tex1, camera, width, height, in_lo, in_hi,
                                             first_round, rgba, aux, out_down, out_up, capacity, overflow, nullptr, stream);
}

__attribute__((visibility("hidden"))) int sdfv_internal_raymarch_slab_round(
    const sdfv_render_params* rp, const sdfv_grid* slab, uint32_t ghost_lo, uint32_t ghost_hi, const float* tex0, const float* tex1,
    const sdfv_camera* camera, uint32_t width, uint32_t height, const void* in_lo, const void* in_hi, int first_round, float* rgba,
    sdfv_march_aux* aux, void* out_down, void* out_up, uint32_t capacity, uint32_t* overflow, uint32_t* leftover, void* stream) {
    if (!out_down || !out_up) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL ray buffer");
    if (((uintptr_t)in_lo | (uintptr_t)in_hi | (uintptr_t)out_down | (uintptr_t)out_up | (uintptr_t)overflow) & 3)
        return fail(SDFV_ERR_INVALID_ARGUMENT, "ray buffers must be 4-byte aligned");
    if (first_round && (in_lo || in_hi)) return fail(SDFV_ERR_INVALID_ARGUMENT, "the first round takes no incoming rays");
    sdfv::RaymarchArgs a;
    sdfv::SlabMarchArgs s;
    if (int rc = slab_round_args(rp, slab, ghost_lo, ghost_hi, tex0, tex1, camera, width, height, rgba, aux, a, s)) return rc;
    auto count_of = [](const void* b) { return reinterpret_cast<const uint32_t*>(b); };
    auto rays_of = [](const void* b) {
        return reinterpret_cast<const sdfv_ray_state*>(static_cast<const char*>(b) + SDFV_RAY_BUFFER_HEADER_BYTES);
    };
    // this round's outgoing lists start empty: their headers are cleared on the stream, ahead of the kernel
    SDFV_HIP(hipMemsetAsync(out_down, 0, SDFV_RAY_BUFFER_HEADER_BYTES, (hipStream_t)stream));
    SDFV_HIP(hipMemsetAsync(out_up, 0, SDFV_RAY_BUFFER_HEADER_BYTES, (hipStream_t)stream));
    if (!first_round) {
        if (!in_lo && !in_hi) return SDFV_OK;  // a rank without neighbours has nothing to continue
        s.in_count[0] = in_lo ? count_of(in_lo) : nullptr;
        s.in_rays[0] = in_lo ? rays_of(in_lo) : nullptr;
        s.in_count[1] = in_hi ? count_of(in_hi) : nullptr;
        s.in_rays[1] = in_hi ? rays_of(in_hi) : nullptr;
        // every pixel's ray is in exactly one place, so the two lists together hold at most width * height rays
        const uint64_t most = (uint64_t)capacity * ((in_lo ? 1u : 0u) + (in_hi ? 1u : 0u));
        const uint64_t pixels = (uint64_t)width * height;
        s.max_in = (uint32_t)(most < pixels ? most : pixels);
        if (s.max_in == 0) return SDFV_OK;
    }
    s.out_down = const_cast<sdfv_ray_state*>(rays_of(out_down));
    s.out_up = const_cast<sdfv_ray_state*>(rays_of(out_up));
    s.count_down = reinterpret_cast<uint32_t*>(out_down);
    s.count_up = reinterpret_cast<uint32_t*>(out_up);
    s.overflow = overflow;
    s.leftover = leftover;
    s.capacity = capacity;
    SDFV_HIP(sdfv::launch_raymarch_slab(a, s, (hipStream_t)stream));
    return SDFV_OK;
}

int sdfv_fill_grid_host(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* grid, float* tex0_host,
                        float* tex1_host) {
    if (int rc = check_grid(grid)) return rc;
    if (!tex0_host || !tex1_host) return fail(SDFV_ERR_INVALID_ARGUMENT, "texture pointer is NULL");
    if (int rc = need_device()) return rc;
    const size_t bytes = (size_t)grid->dims[0] * grid->dims[1] * (grid->z_end - grid->z_begin) * 16;
    if (bytes == 0) return SDFV_OK;
    DeviceBuf d0, d1;
    SDFV_HIP(hipMalloc(&d0.p, bytes));
    SDFV_HIP(hipMalloc(&d1.p, bytes));
    if (int rc = sdfv_fill_grid(params, sdf_id, grid, (float*)d0.p, (float*)d1.p, nullptr)) return rc;
    SDFV_HIP(hipMemcpy(tex0_host, d0.p, bytes, hipMemcpyDeviceToHost));
    SDFV_HIP(hipMemcpy(tex1_host, d1.p, bytes, hipMemcpyDeviceToHost));
    return SDFV_OK;
}

int sdfv_sample_points_host(const sdfv_demo_params* params, uint32_t sdf_id, const float* points_host, size_t n,
                            int distance_only, sdfv_sample* out_host) {
    if (n && (!points_host || !out_host)) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = need_device()) return rc;
    if (n == 0) return SDFV_OK;
    return run_over_host_buffers(points_host, n * 12, out_host, n * sizeof(sdfv_sample), [&](void* in, void* out) {
        return sdfv_sample_points(params, sdf_id, (const float*)in, n, distance_only, (sdfv_sample*)out, nullptr);
    });
}

int sdfv_normal_points_host(const sdfv_demo_params* params, uint32_t sdf_id, const float* points_host, size_t n,
                            float eps, int use_default, float* out_host) {
    if (n && (!points_host || !out_host)) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = need_device()) return rc;
    if (n == 0) return SDFV_OK;
    return run_over_host_buffers(points_host, n * 12, out_host, n * 12, [&](void* in, void* out) {
        return sdfv_normal_points(params, sdf_id, (const float*)in, n, eps, use_default, (float*)out, nullptr);
    });
}

int sdfv_mesh_postproc_host(const sdfv_demo_params* params, uint32_t sdf_id, sdfv_vertex* vertices_host, size_t n) {
    if (n && !vertices_host) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (int rc = check_params(params, sdf_id)) return rc;
    if (int rc = need_device()) return rc;
    if (n == 0) return SDFV_OK;
    DeviceBuf dv;
    SDFV_HIP(hipMalloc(&dv.p, n * sizeof(sdfv_vertex)));
    SDFV_HIP(hipMemcpy(dv.p, vertices_host, n * sizeof(sdfv_vertex), hipMemcpyHostToDevice));
    if (int rc = sdfv_mesh_postproc(params, sdf_id, (sdfv_vertex*)dv.p, n, nullptr)) return rc;
    SDFV_HIP(hipMemcpy(vertices_host, dv.p, n * sizeof(sdfv_vertex), hipMemcpyDeviceToHost));
    return SDFV_OK;
}

int sdfv_raymarch_host(const sdfv_render_params* rp, const float* tex0_host, const float* tex1_host,
                       const sdfv_camera* cameras, uint32_t n_cameras, uint32_t width, uint32_t height,
                       float* rgba_host, sdfv_march_aux* aux_host) {
    if (!rp || !tex0_host || !tex1_host || !rgba_host) return fail(SDFV_ERR_INVALID_ARGUMENT, "NULL argument");
    if (int rc = need_device()) return rc;
    const size_t tex_bytes = (size_t)rp->tex_size[0] * rp->tex_size[1] * rp->tex_size[2] * 16;
    const size_t px = (size_t)n_cameras * width * height;
    if (tex_bytes == 0 || px == 0) return SDFV_OK;
    DeviceBuf d0, d1, dr, da;
    SDFV_HIP(hipMalloc(&d0.p, tex_bytes));
    SDFV_HIP(hipMalloc(&d1.p, tex_bytes));
    SDFV_HIP(hipMalloc(&dr.p, px * 16));
    if (aux_host) SDFV_HIP(hipMalloc(&da.p, px * sizeof(sdfv_march_aux)));
    SDFV_HIP(hipMemcpy(d0.p, tex0_host, tex_bytes, hipMemcpyHostToDevice));
    SDFV_HIP(hipMemcpy(d1.p, tex1_host, tex_bytes, hipMemcpyHostToDevice));
    if (int rc = sdfv_raymarch(rp, (const float*)d0.p, (const float*)d1.p, cameras, n_cameras, width, height, 0, height,
                               (float*)dr.p, (sdfv_march_aux*)da.p, nullptr))
        return rc;
    SDFV_HIP(hipMemcpy(rgba_host, dr.p, px * 16, hipMemcpyDeviceToHost));
    if (aux_host) SDFV_HIP(hipMemcpy(aux_host, da.p, px * sizeof(sdfv_march_aux), hipMemcpyDeviceToHost));
    return SDFV_OK;
}

}  // extern "C"
#pragma GCC visibility pop
