// points_kernels.h -- batched SDFSurface::sample / ::normal over arbitrary point lists.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdfgrid.h"

namespace sdfv {

hipError_t launch_sample_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* points, size_t n,
                                bool distance_only, sdfv_sample* out, hipStream_t stream);
// bb_min/bb_max non-NULL: the points are in the unit cube and go through vert_pos_to first (isosurface.rs:95-99).
hipError_t launch_normal_points(const sdfv_demo_params& prm, uint32_t sdf_id, const float* bb_min,
                                const float* bb_max, const float* points, size_t n, float eps, bool use_default,
                                float* out, hipStream_t stream);
hipError_t launch_source_scalar(const sdfv_demo_params& prm, uint32_t sdf_id, const float* bb_min,
                                const float* bb_max, const float* points, size_t n, float* out, hipStream_t stream);
hipError_t launch_mesh_postproc(const sdfv_demo_params& prm, uint32_t sdf_id, sdfv_vertex* vertices, size_t n,
                                hipStream_t stream);

}  // namespace sdfv
