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
    uint32_t srgb_round;     // Srgba::from policy (SDFV_OPT_EXT_SRGB_QUANT): 0 truncate (default), 1 round
    uint32_t dist_ilv;       // layout of `dist` (and PassArgs::dist): 0 = one float per voxel in texture order; 1 = y-interleaved
                             // (rows 2p, 2p + 1 of the slab as one row of pairs: entry ((row >> 1) * W + x) * 2 + (row & 1), H even) --
                             // the layout the march's hand-written loop gathers fastest from beyond the last-level cache, written
                             // by the fill itself instead of by a commit pass
    // ---- boundary-first order (multi-GPU fill step, launch_fill_dense_ordered) ----
    // The slab's `order_lead` first slices and its last slice are the ones the z-neighbours wait for: the workgroups
    // that fill them come FIRST in dispatch order, the interior follows in memory order.
    uint32_t order_bps;      // workgroups per slice (0 = plain memory order)
    uint32_t order_lead;     // leading boundary slices: 1 (one-voxel halo) or 2 (two ghost slices on the upper side)
    uint32_t block_base;     // first logical workgroup of this launch (the boundary workgroups and the rest are separate launches)
    uint32_t stage_only;     // boundary workgroups write the packed copies only, not the textures (another launch does)
    float4* stage_lo;        // packed copy of the lead slices for the lower neighbour: [tex0 lead slices | tex1 lead slices]
    float4* stage_hi;        // packed copy of the last slice for the upper neighbour:  [tex0 slice | tex1 slice]
    uint32_t* signal;        // plain (not ordered) launches: word the first workgroup stores signal_value to when the launch
    uint32_t signal_value;   // starts -- the communicator's stream waits on it (hipStreamWaitValue32); nullptr = none
};

// Exact division of a 32-bit index by a launch constant (fill_kernels.hip div_u32).
struct DivU32 {
    unsigned long long magic;  // floor((2^64 - 1) / d) + 1
    uint32_t shift;            // log2(d) when d is a power of two
    uint32_t pow2;
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
    uint32_t all_required;   // update_required is known to hold for every visited voxel: nothing is read
    uint32_t fresh;          // every voxel of the slab holds [AIR_DIST; 4] on entry (first pass of a fresh load)
    uint32_t virgin;         // SDFV_PASS_VIRGIN_GRID: the slab's contents are undefined wherever no pass of this load has written
    uint32_t no_adaptive;    // SDFV_OPT_PASS_FORM 1: unflagged passes take the per-voxel kernels only (A/B runs)
    uint32_t stream_loads;   // the update_required test reads the volume / tex0.r with nontemporal loads (a scan that expects to
                             // leave most of what it reads alone: fill_kernels.hip load_once)
    uint64_t index_limit;    // 0 = 2^32: slabs of this many voxels or more are passed over in pieces of whole slices
    // set by the launcher:
    uint32_t n_visited;      // nx * ny * nz
    DivU32 div_nx, div_ny, div_w;
};

struct FillLaunch {
    bool nontemporal;  // global_store_dwordx4 ... nt (measured: within noise of plain stores)
    bool force_flat = false, force_rows = false;  // A/B runs: pin the kernel form
    bool force_paired = false, force_pairrows = false;  // interleaved-volume fill: pin the XCD-paired / the thread-per-pair form
    bool xcd_pairing = false;                     // the device places workgroup b on XCD b % 8 (eight XCDs)
};

hipError_t launch_fill_dense(const FillArgs& a, const FillLaunch& cfg, hipStream_t stream);
// Boundary-first order.  ordered_blocks() = {workgroups per slice, total} or {0, 0} when the slab's shape does not
// allow it (rows that do not fill whole workgroups); launch_fill_dense_ordered runs logical workgroups
// [block_begin, block_end) of that order -- [0, total) is one launch, [0, (lead + 1) * bps) are the boundary slices.
struct OrderedBlocks {
    uint32_t per_slice, total;
};
OrderedBlocks ordered_blocks(const FillArgs& a);
hipError_t launch_fill_dense_ordered(const FillArgs& a, uint32_t block_begin, uint32_t block_end, hipStream_t stream);
// Ghost slices out of the packed receive buffers: up to four contiguous copies of float4 texels in one launch.
struct CopySegments {
    const float4* src[4];
    float4* dst[4];
    uint32_t n[4];  // texels
    float* r_out[4];  // optional: the texels' first component as a compact array (the distance volume's ghost slices)
};
hipError_t launch_copy_segments(const CopySegments& c, hipStream_t stream);
// slab_d slices z_begin + k * z_step in ONE launch (the two boundary slices of a slab: slab_d = 2).
hipError_t launch_fill_slices(const FillArgs& a, hipStream_t stream);
// dense_cfg: store policy / index form of the dense kernel, which an all-required step-1 pass is
hipError_t launch_fill_pass(const FillArgs& a, const PassArgs& p, const FillLaunch& dense_cfg, hipStream_t stream);
hipError_t launch_commit_distance(const float* tex0, float* dist, uint64_t n_voxels, hipStream_t stream);
// pairs[i] = (dist[i], dist[i + W] or, in the last row of a slice, dist[i]) over a whole grid of W x H x (n / (W * H)) voxels
hipError_t launch_commit_pairs(const float* dist, float* pairs, uint32_t W, uint32_t H, uint64_t n_voxels, hipStream_t stream);
// ilv[(z * H/2 + p) * W + x] = (dist[z][2p][x], dist[z][2p+1][x]) over a whole grid with an even H
hipError_t launch_commit_interleaved(const float* dist, float* ilv, uint32_t W, uint64_t n_voxels, hipStream_t stream);
hipError_t launch_grid_init(float* tex0, float* tex1, uint64_t n_voxels, float air, hipStream_t stream);
// [air; 4] (+ dist = air) into every row of the slab whose y or global z is not a multiple of step (step 0: every row)
hipError_t launch_grid_init_unvisited(float* tex0, float* tex1, float* dist, uint32_t W, uint32_t H, uint32_t z_begin,
                                      uint32_t slab_d, uint32_t step, float air, uint32_t dist_ilv, hipStream_t stream);

}  // namespace sdfv
