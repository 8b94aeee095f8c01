// What does a 4 B/voxel read stream cost next to a 36 B/voxel store stream?  Synthetic stand-ins for the step-1 progressive
// pass (reads the distance volume for update_required, rewrites tex0 + tex1 + the volume) on a 256^3 / 512^3 grid.
//   hipcc --offload-arch=gfx950 -O3 rw_mix.hip -o rw_mix && ./rw_mix [side]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float work(float x) {  // ~100 VALU instructions, like fill_voxel
    float v = x;
#pragma unroll
    for (int i = 0; i < 48; ++i) v = v * 1.0001f + 0.5f;
    return v;
}

template <bool NT>
__device__ __forceinline__ void st4(float4* p, float v) {
    if (NT) { v4f t = {v, v, v, v}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); }
    else *p = make_float4(v, v, v, v);
}

// V0: store only (the fused dense fill's traffic)
template <bool NT>
__global__ __launch_bounds__(256) void k_store(float4* t0, float4* t1, float* d, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = work((float)i);
    st4<NT>(t0 + i, v); st4<NT>(t1 + i, v); d[i] = v;
}
// V1: one dword of the volume per lane, stores depend on it (MODE 0 plain load, 1 nt load, 2 sc1 load)
template <bool NT, int MODE>
__global__ __launch_bounds__(256) void k_dword(float4* t0, float4* t1, float* d, uint32_t n, float air) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float f;
    if (MODE == 1) f = __builtin_nontemporal_load(d + i);
    else if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(f) : "v"(d + i) : "memory");
    else f = d[i];
    if (f != air) return;
    const float v = work((float)i);
    st4<NT>(t0 + i, v); st4<NT>(t1 + i, v); d[i] = v;
}
// V2: the quad structure (one float4 of the volume per lane, four rounds of 64 consecutive voxels per wave)
template <bool NT, bool COMPUTE_FIRST>
__global__ __launch_bounds__(256) void k_quad(float4* t0, float4* t1, float* d, uint32_t n, float air) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const float4 f = *reinterpret_cast<const float4*>(d + (size_t)q * 4);
    const uint32_t span0 = (q - lane) * 4;
    float pre = 0.0f;
    if (COMPUTE_FIRST) pre = work((float)(span0 + lane));  // round 0's arithmetic before anyone waits for the load
    uint32_t bits = (f.x == air) | ((f.y == air) << 1) | ((f.z == air) << 2) | ((f.w == air) << 3);
    if (__ballot(bits != 0) == 0ull) return;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t b = (uint32_t)__shfl((int)bits, (int)(j * 16 + (lane >> 2)));
        const uint32_t v = span0 + j * 64 + lane;
        if (!((b >> (lane & 3)) & 1u)) continue;
        const float x = (COMPUTE_FIRST && j == 0) ? pre : work((float)v);
        st4<NT>(t0 + v, x); st4<NT>(t1 + v, x); d[v] = x;
    }
}
// V3: a 1-bit-per-voxel mask instead of the volume (2 MB for 256^3): one 32-bit word per 32 voxels
template <bool NT>
__global__ __launch_bounds__(256) void k_mask(float4* t0, float4* t1, float* d, const uint32_t* mask, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = mask[i >> 5];
    if (!((w >> (i & 31)) & 1u)) return;
    const float v = work((float)i);
    st4<NT>(t0 + i, v); st4<NT>(t1 + i, v); d[i] = v;
}
// V4: dword per lane + a prefetch of the volume `ahead` workgroups further on (lands in L2 / Infinity Cache)
template <bool NT>
__global__ __launch_bounds__(256) void k_prefetch(float4* t0, float4* t1, float* d, uint32_t n, float air, uint32_t ahead) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = i + ahead * 256;
    float pf = 0.0f;
    if (j < n && (threadIdx.x & 31) == 0) pf = __builtin_nontemporal_load(d + j);  // one lane per 128-byte line
    const float f = d[i];
    if (f == air) {
        const float v = work((float)i);
        st4<NT>(t0 + i, v); st4<NT>(t1 + i, v); d[i] = v;
    }
    if (pf == 12345.678f) t0[0] = make_float4(pf, pf, pf, pf);  // the wave waits for its prefetch only here, at its end
}
// V5: read-only (the no-op pass)
__global__ __launch_bounds__(256) void k_read(const float* d, uint32_t n, float air, uint32_t* out) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    const float4 f = *reinterpret_cast<const float4*>(d + (size_t)q * 4);
    if (f.x == air && f.y == 123.0f) out[0] = q;
}
__global__ void k_fill(float* d, uint32_t n, float v) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = v;
}

int main(int argc, char** argv) {
    const uint32_t side = argc > 1 ? atoi(argv[1]) : 256;
    const uint32_t n = side * side * side;
    float4 *t0, *t1; float* d; uint32_t* mask; uint32_t* out;
    CK(hipMalloc(&t0, (size_t)n * 16)); CK(hipMalloc(&t1, (size_t)n * 16)); CK(hipMalloc(&d, (size_t)n * 4));
    CK(hipMalloc(&mask, n / 8)); CK(hipMalloc(&out, 64));
    CK(hipMemset(mask, 0xff, n / 8));
    const float air = 0.101234004f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = n / 256;
    auto timed = [&](const char* name, auto launch, int bytes_per_voxel) {
        std::vector<float> ts;
        for (int r = 0; r < 12; ++r) {
            k_fill<<<blocks, 256>>>(d, n, air);
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        const float ms = ts[ts.size() / 2];
        printf("%-44s %.4f ms  %6.0f GB/s on %d B/voxel\n", name, ms, (double)n * bytes_per_voxel / ms / 1e6, bytes_per_voxel);
    };
    for (int warm = 0; warm < 200; ++warm) k_store<false><<<blocks, 256>>>(t0, t1, d, n);
    timed("store only, plain", [&] { k_store<false><<<blocks, 256>>>(t0, t1, d, n); }, 36);
    timed("store only, nt textures", [&] { k_store<true><<<blocks, 256>>>(t0, t1, d, n); }, 36);
    timed("dword read -> stores, plain", [&] { k_dword<false, 0><<<blocks, 256>>>(t0, t1, d, n, air); }, 40);
    timed("dword read -> stores, nt textures", [&] { k_dword<true, 0><<<blocks, 256>>>(t0, t1, d, n, air); }, 40);
    timed("dword nt read -> nt stores", [&] { k_dword<true, 1><<<blocks, 256>>>(t0, t1, d, n, air); }, 40);
    timed("dword sc1 read -> nt stores", [&] { k_dword<true, 2><<<blocks, 256>>>(t0, t1, d, n, air); }, 40);
    timed("quad read, 4 rounds, plain", [&] { k_quad<false, false><<<blocks / 4, 256>>>(t0, t1, d, n, air); }, 40);
    timed("quad read, 4 rounds, nt textures", [&] { k_quad<true, false><<<blocks / 4, 256>>>(t0, t1, d, n, air); }, 40);
    timed("quad read, compute first, nt textures", [&] { k_quad<true, true><<<blocks / 4, 256>>>(t0, t1, d, n, air); }, 40);
    timed("bit mask read -> stores, plain", [&] { k_mask<false><<<blocks, 256>>>(t0, t1, d, mask, n); }, 36);
    timed("bit mask read -> stores, nt textures", [&] { k_mask<true><<<blocks, 256>>>(t0, t1, d, mask, n); }, 36);
    for (uint32_t ahead : {512u, 2048u, 8192u, 32768u}) {
        char name[64]; snprintf(name, sizeof(name), "dword + prefetch %u WGs ahead, nt tex", ahead);
        timed(name, [&] { k_prefetch<true><<<blocks, 256>>>(t0, t1, d, n, air, ahead); }, 40);
    }
    timed("read only (no-op pass)", [&] { k_read<<<blocks / 4, 256>>>(d, n, air, out); }, 4);
    return 0;
}
