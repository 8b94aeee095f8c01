// What is the fastest a gfx950 kernel can READ a buffer once, cold?  (the step-1 no-op pass reads the
// 4 B/voxel distance volume of a loaded grid and does nothing else; 512^3: 537 MB in 0.143 ms cold = 3.75 TB/s, 0.116 ms in a
// chain of passes.)  Variants of a read-and-test kernel (each wave ORs a comparison of what it loaded and leaves; one lane in
// 2^20 writes, so that nothing is optimised away), over `MiB` mebibytes:
//   b16        16 B per lane, one workgroup of 256 lanes per 4 KiB, memory order            (the pass kernel's shape)
//   b32 / b64  2 / 4 loads of 16 B per lane issued before any is looked at, 8 / 16 KiB per workgroup
//   nt16       b16 with nontemporal loads
//   wg1024     16 B per lane, workgroups of 1024 lanes
//   persist    (CUs x 8) workgroups of 256 lanes striding over the buffer, 4 loads in flight per lane
// each timed COLD (after a 1 GiB memset of another buffer: nothing of the buffer in L2 / Infinity Cache) and CHAINED (8 launches).
//   hipcc --offload-arch=gfx950 -O3 read_stream.hip -o read_stream && ./read_stream [MiB=512]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ bool hit(v4f d, float key) { return d.x == key || d.y == key || d.z == key || d.w == key; }

template <int LOADS, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_read(const v4f* __restrict__ src, uint64_t n16, float key, uint32_t* out) {
    const uint64_t base = (uint64_t)blockIdx.x * BLOCK * LOADS + threadIdx.x;
    v4f d[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const uint64_t at = base + (uint64_t)i * BLOCK;
        d[i] = at < n16 ? (NT ? __builtin_nontemporal_load(src + at) : src[at]) : v4f{0, 0, 0, 0};
    }
    bool any = false;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) any = any || hit(d[i], key);
    if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(out, 1u);
}

__global__ __launch_bounds__(256) void k_persist(const v4f* __restrict__ src, uint64_t n16, float key, uint32_t* out) {
    const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
    bool any = false;
    for (uint64_t base = (uint64_t)blockIdx.x * 256 * 4 + threadIdx.x; base < n16; base += stride) {
        v4f d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = base + i * 256 < n16 ? src[base + i * 256] : v4f{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) any = any || hit(d[i], key);
    }
    if (__ballot(any) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(out, 1u);
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? strtoul(argv[1], nullptr, 10) : 512;
    const size_t bytes = mib << 20;
    const uint64_t n16 = bytes / 16;
    char *buf, *flush;
    uint32_t* out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&flush, (size_t)1 << 30));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0x11, bytes));
    CK(hipMemset(out, 0, 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const float key = 123.0f;
    auto blocks = [&](int per_block) { return (uint32_t)((n16 + per_block - 1) / per_block); };
    struct V { const char* name; void (*launch)(const v4f*, uint64_t, float, uint32_t*, uint32_t, int); int per_block; int cus; };
    auto run = [&](const char* name, auto launch) {
        std::vector<float> cold, chain;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipMemsetAsync(flush, rep, (size_t)1 << 30, nullptr));  // 1 GiB through the caches
            CK(hipEventRecord(e0, nullptr));
            launch();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            cold.push_back(ms);
            launch();
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 8; ++i) launch();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            chain.push_back(ms / 8);
        }
        std::sort(cold.begin(), cold.end());
        std::sort(chain.begin(), chain.end());
        printf("%-10s cold %.4f ms %6.0f GB/s   chained %.4f ms %6.0f GB/s\n", name, cold[3], bytes / (cold[3] * 1e-3) / 1e9, chain[3],
               bytes / (chain[3] * 1e-3) / 1e9);
    };
    const v4f* src = reinterpret_cast<const v4f*>(buf);
    printf("%zu MiB, %d CUs (an event pair alone costs ~5 us of the cold figures)\n", mib, prop.multiProcessorCount);
    run("b16", [&] { hipLaunchKernelGGL((k_read<1, false, 256>), dim3(blocks(256)), dim3(256), 0, nullptr, src, n16, key, out); });
    run("b32", [&] { hipLaunchKernelGGL((k_read<2, false, 256>), dim3(blocks(512)), dim3(256), 0, nullptr, src, n16, key, out); });
    run("b64", [&] { hipLaunchKernelGGL((k_read<4, false, 256>), dim3(blocks(1024)), dim3(256), 0, nullptr, src, n16, key, out); });
    run("b128", [&] { hipLaunchKernelGGL((k_read<8, false, 256>), dim3(blocks(2048)), dim3(256), 0, nullptr, src, n16, key, out); });
    run("nt16", [&] { hipLaunchKernelGGL((k_read<1, true, 256>), dim3(blocks(256)), dim3(256), 0, nullptr, src, n16, key, out); });
    run("nt64", [&] { hipLaunchKernelGGL((k_read<4, true, 256>), dim3(blocks(1024)), dim3(256), 0, nullptr, src, n16, key, out); });
    run("wg1024", [&] { hipLaunchKernelGGL((k_read<1, false, 1024>), dim3(blocks(1024)), dim3(1024), 0, nullptr, src, n16, key, out); });
    run("persist", [&] { hipLaunchKernelGGL(k_persist, dim3(prop.multiProcessorCount * 8), dim3(256), 0, nullptr, src, n16, key, out); });
    uint32_t h = 0;
    CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("(hits %u)\n", h);
    return 0;
}
