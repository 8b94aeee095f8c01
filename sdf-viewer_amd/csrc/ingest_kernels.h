// ingest_kernels.h -- launch interface of the sample-ingest kernel (see ingest_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdfgrid.h"

namespace sdfv {

struct PackArgs {
    const sdfv_sample* samples;  // n records of 28 bytes (src/sdf/mod.rs:104-118)
    const uint32_t* indices;     // n offsets from index_base, or nullptr: record i belongs to voxel index_base + i
    uint64_t index_base;
    uint64_t n;
    uint64_t n_voxels;           // voxels of the slab: a record addressed beyond them is skipped
    uint32_t W;                  // row length (the interleaved volume's index needs it)
    float4* tex0;
    float* tex1;                 // written 12 bytes per voxel: .a keeps whatever the grid holds (scene/sdf/mod.rs:205-208)
    float* dist;                 // optional distance volume kept in sync with tex0.r
    uint32_t dist_ilv;           // its layout (FillArgs::dist_ilv)
    uint32_t srgb_round;         // Srgba::from policy (SDFV_OPT_EXT_SRGB_QUANT)
};

hipError_t launch_pack_samples(const PackArgs& a, hipStream_t stream);

}  // namespace sdfv
