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
    if (t < kBlock * 3 / 4) reinterpret_cast<float4*>(s_in)[t] = points[in4 + t];
    __syncthreads();
    const float px = s_in[t * 3 + 0], py = s_in[t * 3 + 1], pz = s_in[t * 3 + 2];
    Sample s = demo_sample(prm, sdf_id, px, py, pz, distance_only);
    float* o = s_out + t * 7;
    o[0] = s.distance;
    o[1] = s.m.r; o[2] = s.m.g; o[3] = s.m.b;
    o[4] = s.m.metallic; o[5] = s.m.roughness; o[6] = s.m.occlusion;
    __syncthreads();
    out[out4 + t] = reinterpret_cast<const float4*>(s_out)[t];
    if (t < kBlock * 7 / 4 - kBlock) out[out4 + kBlock + t] = reinterpret_cast<const float4*>(s_out)[kBlock + t];
}

// SDFDemoCube::normal, cube.rs:164-177
__device__ __forceinline__ void cube_normal(const sdfv_demo_params& prm, float px, float py, float pz,
                                            float& nx, float& ny, float& nz) {
    float side = prm.cube_half_side;
    nx = fabsf(px) > side ? signum_f32(px) : 0.0f;
    ny = fabsf(py) > side ? signum_f32(py) : 0.0f;
    nz = fabsf(pz) > side ? signum_f32(pz) : 0.0f;
}

// cgmath normalize: v * (1 / |v|)
__device__ __forceinline__ void normalize3(float x, float y, float z, float& nx, float& ny, float& nz) {
    float inv = 1.0f / vec_length(x, y, z);
    nx = x * inv; ny = y * inv; nz = z * inv;
}

__global__ __launch_bounds__(kBlock) void normal_points_kernel(sdfv_demo_params prm, uint32_t sdf_id,
                                                               const float* __restrict__ points, size_t n,
                                                               float eps, bool use_default,
                                                               float* __restrict__ out) {
    {
        const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
        if (i >= n) return;
        const float px = points[i * 3 + 0], py = points[i * 3 + 1], pz = points[i * 3 + 2];
        float nx, ny, nz;
        if (use_default) {
            // normal_default_impl, defaults.rs:49-56: eps.unwrap_or(0.001), 4 taps of sample(.., true)
            const float e = eps > 0.0f ? eps : 0.001f;
            float d1 = demo_sample(prm, sdf_id, px + e, py + -1.0f * e, pz + -1.0f * e, true).distance;
            float d2 = demo_sample(prm, sdf_id, px + -1.0f * e, py + e, pz + -1.0f * e, true).distance;
            float d3 = demo_sample(prm, sdf_id, px + -1.0f * e, py + -1.0f * e, pz + e, true).distance;
            float d4 = demo_sample(prm, sdf_id, px + e, py + e, pz + e, true).distance;
            normalize3(d1 + -d2 + -d3 + d4, -d1 + d2 + -d3 + d4, -d1 + -d2 + d3 + d4, nx, ny, nz);
        } else if (sdf_id == SDFV_SDF_CUBE) {
            cube_normal(prm, px, py, pz, nx, ny, nz);
        } else if (sdf_id == SDFV_SDF_SPHERE) {
            normalize3(px, py, pz, nx, ny, nz);  // sphere.rs:122-124
        } else {
            // SDFDemo::normal, demo/mod.rs:147-156: normal of the closest surface, sphere negated
            float d_box = cube_distance(prm, px, py, pz);
            float d_sph = vec_length(px, py, pz) - prm.sphere_radius;
            if (fabsf(d_box) < fabsf(d_sph)) {
                cube_normal(prm, px, py, pz, nx, ny, nz);
            } else {
                normalize3(px, py, pz, nx, ny, nz);
                nx = -nx; ny = -ny; nz = -nz;
            }
        }
        out[i * 3 + 0] = nx; out[i * 3 + 1] = ny; out[i * 3 + 2] = nz;
    }
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

hipError_t launch_normal_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* points, size_t n,
                                float eps, bool use_default, float* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if ((n + kBlock - 1) / kBlock > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(normal_points_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, prm, sdf_id, points, n,
                       eps, use_default, out);
    return hipGetLastError();
}

}  // namespace sdfv
