// fill_kernels.h -- launch interface of the grid-fill kernels (see fill_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdfgrid.h"

namespace sdfv {

struct FillArgs {
    sdfv_demo_params prm;
    uint32_t sdf_id;
    uint32_t W, H, D;        // global dims
    uint32_t z_begin;        // first global slice held by tex0/tex1
    uint32_t slab_d;         // slices held
    float dm1[3];            // (float)dim - 1.0f          (scene/sdf/mod.rs:168)
    float bb_size[3];        // bb[1] - bb[0]              (scene/sdf/mod.rs:167)
    float bb_min[3];
    float air_dist;
    uint32_t x_chunks;       // set by the launcher: ceil(W / TX)
    unsigned long long w_magic;  // set by the launcher (flat form): floor(2^64 / W) + 1
    uint32_t z_step;         // strided flat form only: local slice k is global slice z_begin + k * z_step and
                             // lives k * z_step slices into the textures (1 everywhere else)
    float4* tex0;
    float4* tex1;
    float* dist;             // optional compact copy of tex0.r written in the same pass (fused SDFViewer::commit)
};

struct PassArgs {
    uint32_t step;
    uint32_t nx, ny, nz;     // visited voxels per axis in this slab
    uint32_t z_first;        // first visited GLOBAL z (multiple of step, >= z_begin)
    uint32_t has_box;
    float box[6];
    float approx_scale[3];   // set by the launcher: bb_size / (dim - 1), for the cheap coordinate estimate
    float approx_margin[3];  // set by the launcher: bound on |estimate - exact voxel coordinate|, with slack
    float* dist;             // optional compact copy of tex0.r, in sync with tex0: read for update_required
                             // instead of the 16-byte texel and rewritten with it (nullptr = read tex0 itself)
};

struct FillLaunch {
    bool nontemporal;  // global_store_dwordx4 ... nt (measured: within noise of plain stores)
    bool force_flat = false, force_rows = false;  // A/B runs: pin the kernel form
};

hipError_t launch_fill_dense(const FillArgs& a, const FillLaunch& cfg, hipStream_t stream);
// slab_d slices z_begin + k * z_step in ONE launch (the two boundary slices of a slab: slab_d = 2).
hipError_t launch_fill_slices(const FillArgs& a, hipStream_t stream);
hipError_t launch_fill_pass(const FillArgs& a, const PassArgs& p, hipStream_t stream);
hipError_t launch_commit_distance(const float* tex0, float* dist, uint64_t n_voxels, hipStream_t stream);
hipError_t launch_grid_init(float* tex0, float* tex1, uint64_t n_voxels, float air, hipStream_t stream);

}  // namespace sdfv
