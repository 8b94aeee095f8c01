// points_kernels.hip -- the "Batched sampling" the reference leaves as a TODO (src/sdf/mod.rs:39):
// SDFSurface::sample(p, distance_only) and SDFSurface::normal(p, eps) for n arbitrary points, one
// thread per point.  Gather-style front end for the meshers (src/sdf/meshers/isosurface.rs:78-92,
// src/sdf/meshers/mesh.rs:22-33) and the per-point C ABI (src/sdf/ffi.rs:57-65,322-332).
#include "points_kernels.h"

#include "demo_sdf_device.h"

namespace sdfv {
namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void sample_points_kernel(sdfv_demo_params prm, uint32_t sdf_id,
                                                               const float* __restrict__ points, size_t n,
                                                               bool distance_only, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float px = points[i * 3 + 0], py = points[i * 3 + 1], pz = points[i * 3 + 2];
        Sample s = demo_sample(prm, sdf_id, px, py, pz, distance_only);
        float* o = out + i * 7;  // #[repr(C)] SDFSample, 28 bytes
        o[0] = s.distance;
        o[1] = s.m.r; o[2] = s.m.g; o[3] = s.m.b;
        o[4] = s.m.metallic; o[5] = s.m.roughness; o[6] = s.m.occlusion;
    }
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
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
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

uint32_t blocks_for(size_t n) {
    size_t b = (n + kBlock - 1) / kBlock;
    return (uint32_t)(b > 8192 ? 8192 : b);
}

}  // namespace

hipError_t launch_sample_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* points, size_t n,
                                bool distance_only, sdfv_sample* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sample_points_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, prm, sdf_id, points, n,
                       distance_only, reinterpret_cast<float*>(out));
    return hipGetLastError();
}

hipError_t launch_normal_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* points, size_t n,
                                float eps, bool use_default, float* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(normal_points_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, stream, prm, sdf_id, points, n,
                       eps, use_default, out);
    return hipGetLastError();
}

}  // namespace sdfv
