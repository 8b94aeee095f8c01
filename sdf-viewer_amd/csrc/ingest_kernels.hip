// ingest_kernels.hip -- the device half of SDFViewer::update for an SDF that can only be sampled on the HOST
// (reference src/app/scene/sdf/mod.rs:193-208 with `sdf` a wasm / FFI provider, src/sdf/wasm/native.rs:188-217).
//
// The host calls SDFSurface::sample(pos, false) and ships the raw 28-byte SDFSample records with the flat index of the
// voxel each belongs to; everything update() does with a sample after that -- the distance offset and clamp, the all-zero
// colour default, Srgba::from + to_linear_srgb through the 256-entry table, the occlusion default -- happens here, with
// the SAME pack_sample the fill kernels use, into tex0 / tex1 (and the compact distance volume).  tex1.a is not touched,
// as in the reference.  One thread per record.
//
// Traffic: 28 B + 4 B read (contiguous), 16 B + 12 B (+ 4 B) written wherever the indices point.  A workgroup's 256
// records are 7 KiB of contiguous memory: they cross global memory as whole-wave dword rows (256 B per instruction) and
// are re-sliced per record in LDS (a stride of 7 dwords is conflict-free), like points_kernels.hip's staged form.
#include "ingest_kernels.h"

#include "demo_sdf_device.h"

namespace sdfv {

namespace {

constexpr int kBlock = 256;

__constant__ float c_ingest_srgb_lut[256] = {
#include "srgb_lut.inc"
};

struct LdsLut {
    const float* p;
    __device__ __forceinline__ float operator[](uint32_t i) const { return p[i]; }
};

template <bool ROUND>
__global__ __launch_bounds__(kBlock) void pack_samples_kernel(PackArgs a) {
    __shared__ float s_lut[256];
    __shared__ float s_rec[kBlock * 7];
    const uint32_t t = threadIdx.x;
    const uint64_t first = (uint64_t)blockIdx.x * kBlock;          // first record of this workgroup
    const uint64_t here = a.n - first < kBlock ? a.n - first : kBlock;  // records it holds
    s_lut[t] = c_ingest_srgb_lut[t];
    const float* src = reinterpret_cast<const float*>(a.samples) + first * 7;
    const uint32_t words = (uint32_t)here * 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uint32_t w = k * kBlock + t;
        if (w < words) s_rec[w] = __builtin_nontemporal_load(src + w);
    }
    __syncthreads();
    if (t >= here) return;
    const uint64_t i = first + t;
    const uint64_t flat = a.index_base + (a.indices ? (uint64_t)a.indices[i] : i);
    if (flat >= a.n_voxels) return;  // not a voxel of this slab: skipped (a host may mark records it does not want stored so)
    const float* r = s_rec + t * 7;
    Sample s;
    s.distance = r[0];
    s.m.r = r[1]; s.m.g = r[2]; s.m.b = r[3];
    s.m.metallic = r[4]; s.m.roughness = r[5]; s.m.occlusion = r[6];
    float4 t0, t1;
    pack_sample<ROUND>(s, LdsLut{s_lut}, 0.0f, t0, t1);
    a.tex0[flat] = t0;
    float* o1 = a.tex1 + flat * 4;  // .rgb only: global_store_dwordx3
    typedef float v3f __attribute__((ext_vector_type(3)));
    v3f m = {t1.x, t1.y, t1.z};
    *reinterpret_cast<v3f*>(o1) = m;
    if (a.dist) {
        uint64_t at = flat;
        if (a.dist_ilv) {  // FillArgs::dist_ilv: rows 2p, 2p + 1 of the slab as one row of pairs
            const uint64_t row = flat / a.W;
            const uint64_t x = flat - row * a.W;
            at = ((row >> 1) * a.W + x) * 2 + (row & 1);
        }
        a.dist[at] = t0.x;
    }
}

}  // namespace

hipError_t launch_pack_samples(const PackArgs& a, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    const uint64_t blocks = (a.n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (a.srgb_round)
        hipLaunchKernelGGL(pack_samples_kernel<true>, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, a);
    else
        hipLaunchKernelGGL(pack_samples_kernel<false>, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace sdfv
