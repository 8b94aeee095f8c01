// raymarch_kernels.h -- launch interface of the sphere-tracing kernel (see raymarch_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdfgrid.h"

namespace sdfv {

struct RaymarchArgs {
    sdfv_render_params rp;
    float bsize[3];              // bounds_max - bounds_min (material.frag:44)
    float inv_bsize[3];          // exact reciprocals, valid when pow2_extent
    uint32_t pow2_extent;        // every bsize[i] is an exact power of two
    uint32_t fast_index;         // 1e-4 * N / size <= 0.25 on every axis: MirroredRepeat == clamp while marching
    const float* dist;           // compact copy of tex0.r (sdfv_commit_distance) or nullptr
    uint32_t pow2_size;          // every tex_size[i] is a power of two (with pow2_extent: one fused scale per axis)
    uint32_t symmetric_box;      // bounds_min == -bounds_max on every axis
    uint32_t fast_normal;        // the normal's 4 taps also keep floor(u) in [-1, N-1]
    float cull_center[3];        // bounding sphere of the box, radius inflated by 1 % (conservative tile cull)
    float cull_radius2;
    const float4* tex0;          // full grid, rp.tex_size
    const float4* tex1;
    uint32_t n_cameras;          // cameras in this launch (<= kMaxCamerasPerLaunch), by value in kernarg
    uint32_t width, height;      // full image
    uint32_t y0, y1;             // rows rendered by this launch
    uint32_t compute_normal;     // evaluate sdfNormal per hit even when no aux is stored
    float4* rgba;                // n_cameras x (y1-y0) x width
    sdfv_march_aux* aux;         // same layout or nullptr
    unsigned long long* wave_timing;  // tuning only: per wave {start, end, iterations, xcc|cu} or nullptr
    sdfv_camera cameras[16];
};

constexpr uint32_t kMaxCamerasPerLaunch = 16;

hipError_t launch_raymarch(const RaymarchArgs& a, hipStream_t stream);

}  // namespace sdfv
