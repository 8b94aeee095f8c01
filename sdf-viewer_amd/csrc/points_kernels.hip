// points_kernels.hip -- the "Batched sampling" the reference leaves as a TODO (src/sdf/mod.rs:39):
// SDFSurface::sample(p, distance_only) and SDFSurface::normal(p, eps) for n arbitrary points, one
// thread per point.  Gather-style front end for the meshers (src/sdf/meshers/isosurface.rs:78-92,
// src/sdf/meshers/mesh.rs:22-33) and the per-point C ABI (src/sdf/ffi.rs:57-65,322-332).
#include "points_kernels.h"

#include "demo_sdf_device.h"

namespace sdfv {
namespace {

constexpr int kBlock = 256;

// Scalar form: any alignment, any n (also finishes the last partial workgroup of the staged form).
__global__ __launch_bounds__(kBlock) void sample_points_kernel(sdfv_demo_params prm, uint32_t sdf_id,
                                                               const float* __restrict__ points, size_t first,
                                                               size_t n, bool distance_only, float* __restrict__ out) {
    const size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float px = points[i * 3 + 0], py = points[i * 3 + 1], pz = points[i * 3 + 2];
    Sample s = demo_sample(prm, sdf_id, px, py, pz, distance_only);
    float* o = out + i * 7;  // #[repr(C)] SDFSample, 28 bytes
    o[0] = s.distance;
    o[1] = s.m.r; o[2] = s.m.g; o[3] = s.m.b;
    o[4] = s.m.metallic; o[5] = s.m.roughness; o[6] = s.m.occlusion;
}

// Staged form for whole workgroups of 256 points: the 12-byte points and the 28-byte samples are arrays of
// structures, so per-lane accesses are 3 and 7 dword operations at a 12 / 28-byte stride.  A workgroup's input
// (3 KiB) and output (7 KiB) are contiguous, though: they cross global memory as dwordx4 and are re-sliced per
// point in LDS (strides of 3 and 7 dwords are conflict-free).  Memory order, one point per thread.
__global__ __launch_bounds__(kBlock) void sample_points_staged_kernel(sdfv_demo_params prm, uint32_t sdf_id,
                                                                      const float4* __restrict__ points,
                                                                      bool distance_only, float4* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_in[kBlock * 3];
    __shared__ __attribute__((aligned(16))) float s_out[kBlock * 7];
    const uint32_t t = threadIdx.x;
    const size_t in4 = (size_t)blockIdx.x * (kBlock * 3 / 4), out4 = (size_t)blockIdx.x * (kBlock * 7 / 4);
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (t < kBlock * 3 / 4) reinterpret_cast<v4f*>(s_in)[t] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(points) + in4 + t);
    __syncthreads();
    const float px = s_in[t * 3 + 0], py = s_in[t * 3 + 1], pz = s_in[t * 3 + 2];
    Sample s = demo_sample(prm, sdf_id, px, py, pz, distance_only);
    float* o = s_out + t * 7;
    o[0] = s.distance;
    o[1] = s.m.r; o[2] = s.m.g; o[3] = s.m.b;
    o[4] = s.m.metallic; o[5] = s.m.roughness; o[6] = s.m.occlusion;
    __syncthreads();
    // streamed: the read stream and the store stream get along better when the stores pass L2 by (EXPERIMENTS R3.4)
    __builtin_nontemporal_store(reinterpret_cast<const v4f*>(s_out)[t], reinterpret_cast<v4f*>(out) + out4 + t);
    if (t < kBlock * 7 / 4 - kBlock)
        __builtin_nontemporal_store(reinterpret_cast<const v4f*>(s_out)[kBlock + t], reinterpret_cast<v4f*>(out) + out4 + kBlock + t);
}

__global__ __launch_bounds__(kBlock) void normal_points_kernel(sdfv_demo_params prm, uint32_t sdf_id, SourceBox box,
                                                               const float* __restrict__ points, size_t first, size_t n,
                                                               float eps, bool use_default,
                                                               float* __restrict__ out) {
    const size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float px = points[i * 3 + 0], py = points[i * 3 + 1], pz = points[i * 3 + 2];
    box.to_world(px, py, pz);
    float nx, ny, nz;
    demo_normal(prm, sdf_id, px, py, pz, eps, use_default, nx, ny, nz);
    out[i * 3 + 0] = nx; out[i * 3 + 1] = ny; out[i * 3 + 2] = nz;
}

// The same for whole workgroups of 256 points, staged like sample_points_staged_kernel: 3 KiB in and 3 KiB out cross global
// memory as streamed dwordx4 and are re-sliced per point in LDS.
__global__ __launch_bounds__(kBlock) void normal_points_staged_kernel(sdfv_demo_params prm, uint32_t sdf_id, SourceBox box,
                                                                      const float4* __restrict__ points, float eps,
                                                                      bool use_default, float4* __restrict__ out) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float s_io[kBlock * 3];
    const uint32_t t = threadIdx.x;
    const size_t at4 = (size_t)blockIdx.x * (kBlock * 3 / 4);
    if (t < kBlock * 3 / 4) reinterpret_cast<v4f*>(s_io)[t] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(points) + at4 + t);
    __syncthreads();
    float px = s_io[t * 3 + 0], py = s_io[t * 3 + 1], pz = s_io[t * 3 + 2];
    box.to_world(px, py, pz);
    float nx, ny, nz;
    demo_normal(prm, sdf_id, px, py, pz, eps, use_default, nx, ny, nz);
    s_io[t * 3 + 0] = nx; s_io[t * 3 + 1] = ny; s_io[t * 3 + 2] = nz;  // a thread's own three words: no barrier in between
    __syncthreads();
    if (t < kBlock * 3 / 4) __builtin_nontemporal_store(reinterpret_cast<const v4f*>(s_io)[t], reinterpret_cast<v4f*>(out) + at4 + t);
}

// ScalarSource::sample_scalar, meshers/isosurface.rs:78-84: distance only, 12 B in, 4 B out per point.
__global__ __launch_bounds__(kBlock) void source_scalar_kernel(sdfv_demo_params prm, uint32_t sdf_id, SourceBox box,
                                                               const float* __restrict__ points, size_t first, size_t n,
                                                               float* __restrict__ out) {
    const size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float px = points[i * 3 + 0], py = points[i * 3 + 1], pz = points[i * 3 + 2];
    box.to_world(px, py, pz);
    out[i] = demo_sample(prm, sdf_id, px, py, pz, true).distance;
}

// Whole workgroups: the 12-byte points through LDS as streamed dwordx4; the 4-byte results are coalesced as they are.
__global__ __launch_bounds__(kBlock) void source_scalar_staged_kernel(sdfv_demo_params prm, uint32_t sdf_id, SourceBox box,
                                                                      const float4* __restrict__ points,
                                                                      float* __restrict__ out) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float s_in[kBlock * 3];
    const uint32_t t = threadIdx.x;
    if (t < kBlock * 3 / 4)
        reinterpret_cast<v4f*>(s_in)[t] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(points) + (size_t)blockIdx.x * (kBlock * 3 / 4) + t);
    __syncthreads();
    float px = s_in[t * 3 + 0], py = s_in[t * 3 + 1], pz = s_in[t * 3 + 2];
    box.to_world(px, py, pz);
    __builtin_nontemporal_store(demo_sample(prm, sdf_id, px, py, pz, true).distance, out + (size_t)blockIdx.x * kBlock + t);
}

// Mesh::postproc, meshers/mesh.rs:22-33, in place over #[repr(Rust)]-free 48-byte vertices (sdfv_vertex):
// material from sample(position, false); normal from sdf.normal(position, None) where the mesher left |n|^2 < 1e-4.
__device__ __forceinline__ void postproc_vertex(const sdfv_demo_params& prm, uint32_t sdf_id, float* v) {
    const float px = v[0], py = v[1], pz = v[2];
    Sample s = demo_sample(prm, sdf_id, px, py, pz, false);
    const float dx = v[3] - 0.0f, dy = v[4] - 0.0f, dz = v[5] - 0.0f;  // distance2(Vector3::zero())
    if (dx * dx + dy * dy + dz * dz < 0.0001f) {
        float nx, ny, nz;
        demo_normal(prm, sdf_id, px, py, pz, 0.0f, false, nx, ny, nz);
        v[3] = nx; v[4] = ny; v[5] = nz;
    }
    v[6] = s.m.r; v[7] = s.m.g; v[8] = s.m.b;
    v[9] = s.m.metallic; v[10] = s.m.roughness; v[11] = s.m.occlusion;
}

__global__ __launch_bounds__(kBlock) void mesh_postproc_kernel(sdfv_demo_params prm, uint32_t sdf_id,
                                                               float* __restrict__ vertices, size_t first, size_t n) {
    const size_t i = first + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float v[12];
    for (int k = 0; k < 12; ++k) v[k] = vertices[i * 12 + k];
    postproc_vertex(prm, sdf_id, v);
    for (int k = 3; k < 12; ++k) vertices[i * 12 + k] = v[k];
}

// Whole workgroups: 256 vertices = 12 KiB contiguous, moved as dwordx4 and re-sliced per vertex in LDS.
__global__ __launch_bounds__(kBlock) void mesh_postproc_staged_kernel(sdfv_demo_params prm, uint32_t sdf_id,
                                                                      float4* __restrict__ vertices) {
    __shared__ __attribute__((aligned(16))) float s_v[kBlock * 12];
    const uint32_t t = threadIdx.x;
    float4* base = vertices + (size_t)blockIdx.x * (kBlock * 3);
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (int k = 0; k < 3; ++k)
        reinterpret_cast<v4f*>(s_v)[k * kBlock + t] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base) + k * kBlock + t);
    __syncthreads();
    float v[12];
    for (int k = 0; k < 12; ++k) v[k] = s_v[t * 12 + k];
    postproc_vertex(prm, sdf_id, v);
    for (int k = 3; k < 12; ++k) s_v[t * 12 + k] = v[k];
    __syncthreads();
    for (int k = 0; k < 3; ++k)
        __builtin_nontemporal_store(reinterpret_cast<const v4f*>(s_v)[k * kBlock + t], reinterpret_cast<v4f*>(base) + k * kBlock + t);
}

uint32_t blocks_for(size_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

}  // namespace

hipError_t launch_sample_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* points, size_t n,
                                bool distance_only, sdfv_sample* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    float* o = reinterpret_cast<float*>(out);
    size_t done = 0;
    const size_t whole = n / kBlock;
    if (whole > 0 && whole <= 0x7fffffffull && (((uintptr_t)points | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(sample_points_staged_kernel, dim3((uint32_t)whole), dim3(kBlock), 0, stream, prm, sdf_id,
                           reinterpret_cast<const float4*>(points), distance_only, reinterpret_cast<float4*>(o));
        done = whole * kBlock;
    }
    if (done < n) {
        const size_t blocks = (n - done + kBlock - 1) / kBlock;
        if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(sample_points_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, prm, sdf_id, points,
                           done, n, distance_only, o);
    }
    return hipGetLastError();
}

namespace {
SourceBox make_box(const float* bb_min, const float* bb_max) {
    SourceBox b{};
    b.unit_cube = bb_min != nullptr && bb_max != nullptr;
    for (int i = 0; i < 3 && b.unit_cube; ++i) {
        b.bb_min[i] = bb_min[i];
        b.bb_size[i] = bb_max[i] - bb_min[i];
    }
    return b;
}
}  // namespace

hipError_t launch_normal_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* bb_min,
                                const float* bb_max, const float* points, size_t n, float eps, bool use_default,
                                float* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if ((n + kBlock - 1) / kBlock > 0x7fffffffull) return hipErrorInvalidValue;
    const SourceBox box = make_box(bb_min, bb_max);
    size_t done = 0;
    const size_t whole = n / kBlock;
    if (whole > 0 && (((uintptr_t)points | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(normal_points_staged_kernel, dim3((uint32_t)whole), dim3(kBlock), 0, stream, prm, sdf_id, box,
                           reinterpret_cast<const float4*>(points), eps, use_default, reinterpret_cast<float4*>(out));
        done = whole * kBlock;
    }
    if (done < n)
        hipLaunchKernelGGL(normal_points_kernel, dim3(blocks_for(n - done)), dim3(kBlock), 0, stream, prm, sdf_id, box, points,
                           done, n, eps, use_default, out);
    return hipGetLastError();
}

hipError_t launch_source_scalar(const sdfv_demo_params& prm, uint32_t sdf_id, const float* bb_min,
                                const float* bb_max, const float* points, size_t n, float* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if ((n + kBlock - 1) / kBlock > 0x7fffffffull) return hipErrorInvalidValue;
    const SourceBox box = make_box(bb_min, bb_max);
    size_t done = 0;
    const size_t whole = n / kBlock;
    if (whole > 0 && (((uintptr_t)points & 15) | ((uintptr_t)out & 3)) == 0) {
        hipLaunchKernelGGL(source_scalar_staged_kernel, dim3((uint32_t)whole), dim3(kBlock), 0, stream, prm, sdf_id, box,
                           reinterpret_cast<const float4*>(points), out);
        done = whole * kBlock;
    }
    if (done < n)
        hipLaunchKernelGGL(source_scalar_kernel, dim3(blocks_for(n - done)), dim3(kBlock), 0, stream, prm, sdf_id, box, points,
                           done, n, out);
    return hipGetLastError();
}

hipError_t launch_mesh_postproc(const sdfv_demo_params& prm, uint32_t sdf_id, sdfv_vertex* vertices, size_t n,
                                hipStream_t stream) {
    if (n == 0) return hipSuccess;
    float* v = reinterpret_cast<float*>(vertices);
    size_t done = 0;
    const size_t whole = n / kBlock;
    if (whole > 0 && whole <= 0x7fffffffull && ((uintptr_t)vertices & 15) == 0) {
        hipLaunchKernelGGL(mesh_postproc_staged_kernel, dim3((uint32_t)whole), dim3(kBlock), 0, stream, prm, sdf_id,
                           reinterpret_cast<float4*>(v));
        done = whole * kBlock;
    }
    if (done < n) {
        const size_t blocks = (n - done + kBlock - 1) / kBlock;
        if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(mesh_postproc_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, prm, sdf_id, v, done, n);
    }
    return hipGetLastError();
}

}  // namespace sdfv
