// Does the fused fill's store rate depend on HOW MANY store streams a launch runs, and is ONE interleaved stream immune to
// where the allocator puts things?  ("one store stream instead of three".)  Synthetic stand-ins for the
// fused dense fill of a side^3 grid (36 B/voxel: tex0 16 + tex1 16 + volume 4), same workgroup shape (256 voxels, memset
// order), ~100 VALU instructions per voxel, nontemporal texture stores:
//   sep     three allocations (tex0, tex1, volume)                        -- the product's layout
//   block   one allocation, tex1 `skew` bytes after tex0's end, volume after tex1
//   row     ONE buffer, row-pitched: [tex0 row | tex1 row | volume row], 36 W bytes per row
//   chunk   ONE buffer, workgroup-chunked: [4 KiB tex0 | 4 KiB tex1 | 1 KiB volume] per 256 voxels
//   flat    ONE buffer of 36 B/voxel written front to back as float4 (what a memset of those bytes does)
// Run in several FRESH processes (the physical pages differ from process to process): tools/ubench/streams.sh.
//   hipcc --offload-arch=gfx950 -O3 streams.hip -o streams && ./streams [side=512] [dummy_MiB=0] [skew=0]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float work(float x) {
    float v = x;
#pragma unroll
    for (int i = 0; i < 48; ++i) v = v * 1.0001f + 0.5f;
    return v;
}
__device__ __forceinline__ void st4nt(void* p, float v) {
    v4f t = {v, v, v, v};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// three base pointers, texel i at base + 16 i (volume: 4 i)
__global__ __launch_bounds__(256) void k_sep(char* t0, char* t1, char* d, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = work((float)i);
    st4nt(t0 + (size_t)i * 16, v);
    st4nt(t1 + (size_t)i * 16, v);
    *reinterpret_cast<float*>(d + (size_t)i * 4) = v;
}
// k_sep with the launch's workgroups dealt over S equal parts of the grid (workgroup b -> part b % S, chunk b / S of it): S times
// as many store streams in flight, each in memory order.  vol = 0: the plain fill's two streams only.
__global__ __launch_bounds__(256) void k_split(char* t0, char* t1, char* d, uint32_t n, uint32_t S, uint32_t chunks_per_part) {
    const uint32_t b = (blockIdx.x % S) * chunks_per_part + blockIdx.x / S;
    const uint32_t i = b * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = work((float)i);
    st4nt(t0 + (size_t)i * 16, v);
    st4nt(t1 + (size_t)i * 16, v);
    if (d) *reinterpret_cast<float*>(d + (size_t)i * 4) = v;
}
// one buffer; a unit of U voxels (a row: U = W; a workgroup chunk: U = 256) is [16 U | 16 U | 4 U] bytes
__global__ __launch_bounds__(256) void k_unit(char* base, uint32_t n, uint32_t unit_shift) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = work((float)i);
    const uint32_t u = i >> unit_shift, x = i & ((1u << unit_shift) - 1u);
    const size_t U = (size_t)1 << unit_shift;
    char* p = base + (size_t)u * 36 * U;
    st4nt(p + (size_t)x * 16, v);
    st4nt(p + 16 * U + (size_t)x * 16, v);
    *reinterpret_cast<float*>(p + 32 * U + (size_t)x * 4) = v;
}
// the same, the volume's 1 KiB of a 256-voxel chunk stored as float4 by the first wave's lanes (whole lines everywhere)
__global__ __launch_bounds__(256) void k_flat(char* base, uint32_t n) {
    // a workgroup writes 9216 contiguous bytes = 576 float4: lanes 0..255 two each, lanes 0..63 one more
    const size_t b = (size_t)blockIdx.x * 9216;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = work((float)i);
    st4nt(base + b + (size_t)threadIdx.x * 16, v);
    st4nt(base + b + 4096 + (size_t)threadIdx.x * 16, v);
    if (threadIdx.x < 64) st4nt(base + b + 8192 + (size_t)threadIdx.x * 16, v);
}

// Virtual-memory-management placement: the three arrays at their usual virtual addresses inside one
// reservation, backed by physical chunks of `chunk` bytes created in the order tex0[k], tex1[k], volume[k / 4] -- if the
// allocator hands out physical memory front to back, the chunks the three lockstep streams write at any moment are neighbours.
// chunk == 0: ONE physical handle for everything (what hipMalloc does, minus its sub-allocator).
struct Vmm {
    char* va = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool ok = false;
};
static Vmm vmm_alloc(size_t tex, size_t vol, size_t chunk) {
    Vmm m;
    int dev = 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return m;
    if (chunk && chunk % gran) chunk = ((chunk + gran - 1) / gran) * gran;
    m.bytes = 2 * tex + vol;
    if (hipMemAddressReserve((void**)&m.va, m.bytes, 0, nullptr, 0) != hipSuccess) return m;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto map = [&](size_t off, size_t len) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, len, &prop, 0) != hipSuccess) return false;
        m.handles.push_back(h);
        return hipMemMap(m.va + off, len, 0, h, 0) == hipSuccess;
    };
    bool good = true;
    if (chunk == 0) {
        good = map(0, m.bytes);
    } else {
        const size_t nchunks = tex / chunk;  // (tex is a multiple of every chunk size tried)
        for (size_t k = 0; k < nchunks && good; ++k) {
            good = map(k * chunk, chunk) && map(tex + k * chunk, chunk);
            if (good && k % 4 == 3) good = map(2 * tex + (k / 4) * chunk, chunk);
        }
    }
    if (good) good = hipMemSetAccess(m.va, m.bytes, &acc, 1) == hipSuccess;
    m.ok = good;
    if (!good) (void)hipGetLastError();
    return m;
}
static void vmm_free(Vmm& m) {
    if (m.va) {
        (void)hipMemUnmap(m.va, m.bytes);
        for (auto h : m.handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(m.va, m.bytes);
    }
    (void)hipGetLastError();
}

template <typename F>
static float timed(F launch, int reps, int warm = 300) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) launch();  // clocks
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / reps);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2];
}

// ./streams sweep [side]: ONE block, tex1 at tex0's end + skew for a coarse ladder of skews (the fine structure below 64 KiB is
// EXPERIMENTS R4.1's); where the three-allocation form lands by itself is printed by the default mode (addresses).
static int sweep(uint32_t side) {
    const uint32_t n = side * side * side;
    const size_t tex = (size_t)n * 16, vol = (size_t)n * 4, slack = (size_t)1200 << 20;
    const dim3 grid(n / 256), block(256);
    char* blk;
    CK(hipMalloc(&blk, 2 * tex + vol + slack + (1 << 20)));
    const int reps = side >= 512 ? 10 : 50;
    std::vector<size_t> skews;
    for (size_t k = 0; k <= 32; ++k) skews.push_back(k << 21);                      // 0 .. 64 MiB in 2 MiB steps
    for (size_t k = 5; k <= 16; ++k) skews.push_back((k << 24));                    // 80 .. 256 MiB in 16 MiB steps
    for (size_t k = 5; k <= 16; ++k) skews.push_back((k << 26));                    // 320 MiB .. 1 GiB in 64 MiB steps
    for (size_t k : {1, 3, 5, 7, 9, 11, 13, 15}) skews.push_back(((size_t)k << 16)); // 64 KiB .. 960 KiB
    for (size_t k : {17, 33, 65, 129, 257}) skews.push_back(((size_t)k << 21) + (12 << 10));
    printf("{\"side\": %u, \"base\": \"%p\", \"sweep\": [", side, (void*)blk);
    bool first = true;
    for (size_t skew : skews) {
        if (skew > slack) continue;
        char *t0 = blk, *t1 = blk + tex + skew, *d = blk + 2 * tex + slack;
        const float ms = timed([&] { hipLaunchKernelGGL(k_sep, grid, block, 0, 0, t0, t1, d, n); }, reps, first ? 300 : 20);
        printf("%s[%zu, %.4f]", first ? "" : ", ", skew, ms);
        first = false;
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}

// ./streams split [side]: how many streams per texture should a launch run?
static int split(uint32_t side) {
    const uint32_t n = side * side * side;
    const size_t tex = (size_t)n * 16, vol = (size_t)n * 4;
    const dim3 grid(n / 256), block(256);
    char* blk;
    CK(hipMalloc(&blk, 2 * tex + vol + (1 << 20)));
    const int reps = side >= 512 ? 10 : 50;
    printf("{\"side\": %u, \"split\": [", side);
    bool first = true;
    for (uint32_t S : {1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
        const uint32_t cpp = (n / 256) / S;
        const float fused = timed([&] { hipLaunchKernelGGL(k_split, grid, block, 0, 0, blk, blk + tex, blk + 2 * tex + (1 << 19), n, S, cpp); }, reps, first ? 300 : 30);
        const float plain = timed([&] { hipLaunchKernelGGL(k_split, grid, block, 0, 0, blk, blk + tex, (char*)nullptr, n, S, cpp); }, reps, 30);
        printf("%s{\"S\": %u, \"fused_ms\": %.4f, \"fused_bus_TBps\": %.2f, \"plain_ms\": %.4f, \"plain_TBps\": %.2f}", first ? "" : ", ", S, fused,
               36.0 * n / 1e9 / fused, plain, 32.0 * n / 1e9 / plain);
        first = false;
        fflush(stdout);
    }
    printf("]}\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "split")) return split(argc > 2 ? atoi(argv[2]) : 512);
    if (argc > 1 && !strcmp(argv[1], "sweep")) return sweep(argc > 2 ? atoi(argv[2]) : 512);
    const uint32_t side = argc > 1 ? atoi(argv[1]) : 512;
    const size_t dummy = (argc > 2 ? (size_t)atoll(argv[2]) : 0) << 20;
    const size_t skew = argc > 3 ? (size_t)atoll(argv[3]) : 0;
    const uint32_t n = side * side * side;
    const size_t tex = (size_t)n * 16, vol = (size_t)n * 4;
    uint32_t row_shift = 0;
    while ((1u << row_shift) < side) ++row_shift;
    const dim3 grid(n / 256), block(256);
    void* pad = nullptr;
    if (dummy) CK(hipMalloc(&pad, dummy));  // moves everything that follows to other physical pages
    const int reps = side >= 512 ? 20 : 100;
    char *a0, *a1, *ad, *blk, *one;
    CK(hipMalloc(&a0, tex));
    CK(hipMalloc(&a1, tex));
    CK(hipMalloc(&ad, vol));
    const float sep = timed([&] { hipLaunchKernelGGL(k_sep, grid, block, 0, 0, a0, a1, ad, n); }, reps);
    fprintf(stderr, "sep: tex0 %p tex1 %p (tex1 - tex0 - bytes = %lld) volume %p (vol - tex1 - bytes = %lld)\n", (void*)a0, (void*)a1,
            (long long)(a1 - a0) - (long long)tex, (void*)ad, (long long)(ad - a1) - (long long)tex);
    CK(hipFree(a0)); CK(hipFree(a1)); CK(hipFree(ad));
    CK(hipMalloc(&blk, 2 * tex + vol + (1 << 20)));
    const float blk0 = timed([&] { hipLaunchKernelGGL(k_sep, grid, block, 0, 0, blk, blk + tex + skew, blk + 2 * tex + (1 << 19), n); }, reps);
    CK(hipFree(blk));
    float vmm_ms[3] = {-1.0f, -1.0f, -1.0f};
    const size_t chunks[3] = {0, (size_t)2 << 20, (size_t)32 << 20};
    for (int c = 0; c < 3; ++c) {
        Vmm m = vmm_alloc(tex, vol, chunks[c]);
        if (m.ok) vmm_ms[c] = timed([&] { hipLaunchKernelGGL(k_sep, grid, block, 0, 0, m.va, m.va + tex, m.va + 2 * tex, n); }, reps);
        vmm_free(m);
    }
    CK(hipMalloc(&one, 2 * tex + vol));
    const float row = timed([&] { hipLaunchKernelGGL(k_unit, grid, block, 0, 0, one, n, row_shift); }, reps);
    const float chunk = timed([&] { hipLaunchKernelGGL(k_unit, grid, block, 0, 0, one, n, 8u); }, reps);
    const float flat = timed([&] { hipLaunchKernelGGL(k_flat, grid, block, 0, 0, one, n); }, reps);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) CK(hipMemsetAsync(one, 0, 2 * tex + vol, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float mset;
    CK(hipEventElapsedTime(&mset, e0, e1));
    mset /= reps;
    const double gb = 36.0 * n / 1e9;
    auto tb = [&](float ms) { return gb / ms; };  // TB/s on the bus
    printf("{\"side\": %u, \"dummy_MiB\": %zu, \"skew\": %zu, \"ms\": {\"sep\": %.4f, \"block\": %.4f, \"row\": %.4f, \"chunk\": %.4f, \"flat\": %.4f, "
           "\"memset\": %.4f, \"vmm_one\": %.4f, \"vmm_2M\": %.4f, \"vmm_32M\": %.4f}, \"bus_TBps\": {\"sep\": %.2f, \"block\": %.2f, \"row\": %.2f, \"chunk\": %.2f, \"flat\": %.2f, \"memset\": %.2f}}\n",
           side, dummy >> 20, skew, sep, blk0, row, chunk, flat, mset, vmm_ms[0], vmm_ms[1], vmm_ms[2], tb(sep), tb(blk0), tb(row), tb(chunk), tb(flat), tb(mset));
    return 0;
}
