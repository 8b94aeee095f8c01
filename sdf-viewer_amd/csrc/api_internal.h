// api_internal.h -- shared by the translation units behind include/sdfgrid.h (not part of the ABI).
#pragma once

#include <cstdint>
#include <string>

#include "../../include/sdfgrid.h"

namespace sdfv {
// The calling thread's options (sdfv_set_option); every entry point copies what it needs once per call.
struct Options {
    uint32_t fill_nontemporal = 0;   // 0 auto (nt when the launch also writes the distance volume), 1 always, 2 never
    uint32_t fill_form = 0;          // 0 auto, 1 rows, 2 flat
    uint32_t raymarch_disable = 0;   // SDFV_RM_NO_*
    bool raymarch_keep_normal = false;
    uint32_t raymarch_batch_streams = 1;  // batches of several SMALL launches overlap on side streams
    uint32_t raymarch_camera_staging = 1;  // host arrays of more than kInlineCameras cameras are copied to stream-ordered device memory
    uint32_t raymarch_box_first = 1;   // single frames: groups under the projected bounding box are launched first
    uint32_t raymarch_waves_per_simd = 0;  // 0 = no cap
    uint32_t raymarch_tile_group = 0;  // 0 auto, 1 launch order, v >= 2: XCD-aware order over groups of 2^(v-1) x 2^(v-1) tiles
    uint32_t slab_step_form = 0;     // 0 auto, SDFV_STEP_* otherwise
    uint32_t ext_srgb_quant = 0;     // Srgba::from(Vec3): 0 truncate (default), 1 round
    uint32_t pass_form = 0;          // 0 auto, 1 = unflagged passes take the per-voxel kernels only
    uint32_t pass_loads = 0;         // 0 auto (SDFV_PASS_EXPECT_NOOP decides), 1 cached loads, 2 nontemporal loads for update_required
    unsigned long long pass_index_limit = 0;  // 0 = 2^32: voxels per piece of a pass over a slab too large for 32-bit indices
    unsigned long long wave_timing = 0;  // tuning build only
    unsigned long long priority_map = 0;  // tuning build only
    unsigned long long tile_order = 0;    // tuning build only
};
const Options& options();
// SDFV_OPT_RCCL_LIBRARY: process-wide path of the RCCL-ABI library the communicator loads (empty = librccl.so.1 by name)
// (claim_rccl_library_path: the path, and from that moment on SDFV_OPT_RCCL_LIBRARY is refused -- one RCCL per process)
const char* rccl_library_path();
std::string rccl_library_path_copy();
bool rccl_loaded();
const char* claim_rccl_library_path();
// Formats the thread-local message sdfv_last_error() returns and hands `code` back.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// The first and the last slice of `slab` (the ones the z-neighbours need) in one launch; o0/o1 address the first
// owned slice.  Same texels as sdfv_fill_grid over those two slices.  Requires at least two owned slices.
// Dense fill of `slab` in boundary-first workgroup order (fill_kernels.h): the `lead` first slices and the last one come
// first, optionally copied into the packed staging buffers.
struct OrderedFill {
    uint32_t lead = 1;
    bool stage_only = false;  // write the packed copies only (the textures are another launch's)
    float* dist = nullptr;    // compact distance volume of the owned slices (fused commit), or nullptr
    float* stage_lo = nullptr;
    float* stage_hi = nullptr;
};
// Workgroups per slice and in total of that order for this slab; per_slice = 0: the shape does not allow it.
int ordered_fill_blocks(const sdfv_grid* slab, uint32_t* per_slice, uint32_t* total);
// Logical workgroups [block_begin, block_end) of the ordered fill; o0/o1 address the first owned slice.
int fill_slab_ordered(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab, float* o0, float* o1,
                      const OrderedFill& of, uint32_t block_begin, uint32_t block_end, void* stream);
// dst[i] = src[i] over up to four segments of `n` 16-byte texels (ghost slices out of the packed receive buffers).
// r_out (or its entries) may be NULL; r_out[i] receives the first component of segment i's texels as a compact array
int copy_texel_segments(const float* const src[4], float* const dst[4], const size_t n[4], float* const r_out[4],
                        void* stream);
// sdfv_fill_grid whose first workgroup stores `value` to `signal` (signal memory) as soon as the launch starts.
int fill_grid_signalling_start(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* grid, float* tex0,
                               float* tex1, float* dist, uint32_t* signal, uint32_t value, void* stream);
// dist[i] = tex0[i].r over n texels (the ghost slices' share of the compact distance volume)
int extract_distance(const float* tex0, float* dist, size_t n, void* stream);
int fill_boundary_slices(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab, float* o0, float* o1,
                         void* stream);
}  // namespace sdfv

// sdfv_raymarch_slab_round with a counter of the rays the round hands on (`leftover`, DEVICE or NULL): sdfv_slab_march passes
// it in its last round, after which no ray may be in flight.  Not part of the ABI (hidden).
extern "C" __attribute__((visibility("hidden"))) int sdfv_internal_raymarch_slab_round(
    const sdfv_render_params* rp, const sdfv_grid* slab, uint32_t ghost_lo, uint32_t ghost_hi, const float* tex0, const float* tex1,
    const sdfv_camera* camera, uint32_t width, uint32_t height, const void* in_lo, const void* in_hi, int first_round, float* rgba,
    sdfv_march_aux* aux, void* out_down, void* out_up, uint32_t capacity, uint32_t* overflow, uint32_t* leftover, void* stream);
