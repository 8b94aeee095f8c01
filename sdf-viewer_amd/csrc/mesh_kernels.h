// mesh_kernels.h -- cube-by-cube isosurface extraction over the unit-cube lattice the reference's meshers sample
// (src/sdf/meshers/isosurface.rs:16-66: MarchingCubes::<Signed>::new(max_voxels_per_axis) over ScalarSource /
// HermiteSource).  The extractor itself lives in the un-vendored `isosurface` crate; this is the build's own.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdfgrid.h"

namespace sdfv {

struct MeshGrid {
    uint32_t cells[3];     // cells per axis; lattice points = cells + 1
    float bb_min[3], bb_size[3];
};

struct MeshWork {           // device scratch of one extraction, sized by mesh_work_bytes()
    float* dist;            // [points] ScalarSource at every lattice point
    uint32_t* point_first;  // [points] first vertex id of the point's edges (exclusive scan of the edge counts)
    uint8_t* point_mask;    // [points] bit a: the edge towards +axis a crosses the surface
    uint32_t* cell_first;   // [cells] first triangle of the cell (exclusive scan of the triangle counts)
    void* scan_tmp;
    size_t scan_tmp_bytes;
};

size_t mesh_scan_tmp_bytes(size_t n);
// Phase 1: lattice distances, edge masks, both scans.  Leaves the totals in totals_dev[0] (vertices), [1] (triangles).
hipError_t launch_mesh_count(const sdfv_demo_params& prm, uint32_t sdf_id, const MeshGrid& g, const MeshWork& w,
                             uint32_t* totals_dev, hipStream_t stream);
// Phase 2: vertices (position + HermiteSource normal, the rest zero) and triangle indices.
hipError_t launch_mesh_emit(const sdfv_demo_params& prm, uint32_t sdf_id, const MeshGrid& g, const MeshWork& w,
                            sdfv_vertex* vertices, uint32_t* indices, hipStream_t stream);

}  // namespace sdfv
