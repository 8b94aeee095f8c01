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
    const float* pairs;          // y-pair volume (sdfv_commit_pairs: 2 floats per texel) or nullptr
    const float* ilv;            // y-interleaved volume (sdfv_commit_interleaved: 1 float per texel, rows paired) or nullptr
    uint32_t pow2_size;          // with pow2_extent: one fused scale (1/size) * N per axis -- exact for ANY N below 2^24, since
                                 // 1/size is a power of two (the name dates from when only power-of-two N took it)
    uint32_t symmetric_box;      // bounds_min == -bounds_max on every axis
    uint32_t fast_normal;        // the normal's 4 taps also keep floor(u) in [-1, N-1]
    float cull_center[3];        // bounding sphere of the box, radius inflated by 1 % (conservative tile cull)
    float cull_radius2;
    const float4* tex0;          // full grid, rp.tex_size
    const float4* tex1;
    uint32_t n_cameras;          // cameras in this launch (<= kMaxCamerasPerLaunch): cameras[] or camera_list, see below
    uint32_t width, height;      // full image
    uint32_t y0, y1;             // rows rendered by this launch
    uint32_t compute_normal;     // evaluate sdfNormal per hit even when no aux is stored
    uint32_t asm_loop;           // use the hand-written march loop where its specialisation applies (default 1)
    uint32_t group_shift;        // 0 = tiles in launch order; g > 0: XCD-aware order over groups of 2^g x 2^g tiles (kGroupAuto: launcher's choice)
    uint32_t tiles_x, tiles_y, groups_x;  // set by the launcher when group_shift > 0
    // box-first order (single camera, set by the launcher; first_w == 0: off): the groups inside the screen rectangle
    // [first_gx0, +first_w) x [first_gy0, +first_h) of the projected bounding box are launched before all others.
    // m_* = ceil(2^32 / d): n / d == mulhi(n, m) for n, d < 2^16
    uint32_t box_first;          // option: use that order (default 1)
    uint32_t first_gx0, first_gy0, first_w, first_h, m_groups_x, m_first_w, m_rest_w;
    uint32_t no_interior_fetch;  // hand-written loop: always take the clamping fetch block (tests, A/B)
    uint32_t rotate_columns;     // launch order only (group_shift == 0): workgroup (x, y, z) renders tile column (x + y + z) mod
                                 // tiles_x -- see raymarch_kernel; set by the API layer on eight-XCD parts, order only
    uint32_t lds_cap_bytes;      // unused dynamic LDS per workgroup: caps the resident waves per SIMD (0 = no cap); set by the launcher
    uint32_t waves_per_simd;     // option: 0 = the launcher's occupancy rule, 7 = never cap, 2..6 = that cap
    uint32_t wave_slots_per_simd_unit;  // SIMDs of the device (CUs x 4): resident waves at w per SIMD = w x this; 0 = unknown
    uint64_t last_level_cache_bytes;    // Infinity Cache (MI355X: 256 MB); 0 = unknown
    uint32_t cube_box;           // symmetric box with bounds_max[0] == [1] == [2]: two-instruction out-of-bounds test
    uint32_t rows_out;           // rows per camera in the outputs: y1 - y0, or the rows of the rendered bands (band_skip != 0)
    uint32_t band_skip;          // 0: rows [y0, y1).  B * (step - 1): the B-row bands y0/B, y0/B + step, ... below y1, stored one
                                 // after the other (the balanced image-tile split); B = 1 << band_shift = 16 or 8 (a wave's tile)
    uint32_t band_shift;
    float4* rgba;                // n_cameras x rows_out x width, or nullptr (rgba8 only)
    uint32_t* rgba8;             // same pixels as 8-bit UNORM RGBA (rint(clamp(c, 0, 1) * 255), R in the low byte), or nullptr
    sdfv_march_aux* aux;         // same layout or nullptr
    float* depth;                // gl_FragDepth plane, same pixel layout, or nullptr
#ifdef SDFV_TUNING
    const uint32_t* tile_order;  // tuning build only: workgroup L renders tile tile_order[L] (single camera)
    const unsigned char* priority_map;  // tuning build only: one byte per tile, non-zero = raise the waves' priority
    unsigned long long* wave_timing;  // tuning build only: per wave {start, end, iterations, covered mask} or nullptr
#endif
    // Cameras: a launch of up to kInlineCameras carries them by value in this block (16 x 120 B; the whole block stays below
    // HIP's documented 4 KB of kernel arguments -- round 3 carried 64 in an 8 KB block); a larger launch reads them
    // from `camera_list` (DEVICE, n_cameras entries: the caller's own array when it lies in device memory, else a stream-ordered
    // copy the launcher makes).  Wave-uniform either way: scalar loads into SGPRs.
    const sdfv_camera* camera_list;  // nullptr: cameras[]
    sdfv_camera cameras[16];
};

constexpr uint32_t kInlineCameras = 16;
// One launch for BASELINE config 5's 64-camera batch (1.59 -> 1.48 ms against four launches of 16, and a rank's share of it
// is one launch too); larger batches are several launches of this many (small ones overlap on side streams).
constexpr uint32_t kMaxCamerasPerLaunch = 64;
constexpr uint32_t kGroupAuto = 255;  // RaymarchArgs::group_shift as handed to launch_raymarch: let the launcher choose

// One round of the march over a z-slab of the grid (multi-GPU: the grid stays sharded, rays move between ranks).
struct SlabMarchArgs {
    uint32_t z_lo, z_count;        // resident slices [z_lo, z_lo + z_count) (owned + ghosts); tex0/tex1 address z_lo
    uint32_t own_begin, own_end;   // owned slices: this rank marches a ray while clamp(floor(w), 0, D-1) is in here
    const sdfv_ray_state* in;      // continuation round: rays handed over by the neighbours; nullptr = first round
    uint32_t n_in;                 // ... their number, known to the host (sdfv_raymarch_slab)
    // continuation round whose counts live on the DEVICE (sdfv_raymarch_slab_round: no read-back between rounds): up to two
    // ray buffers {count, ...header..., states[]}; thread t takes ray t of the first while t < its count, then of the second
    const uint32_t* in_count[2];   // nullptr = not used (with in == nullptr and both null: first round)
    const sdfv_ray_state* in_rays[2];
    uint32_t max_in;               // threads launched for such a round (>= the two counts' sum; <= width * height)
    sdfv_ray_state* out_down;      // rays leaving through the low / high face of the slab
    sdfv_ray_state* out_up;
    uint32_t* count_down;          // rays in out_down / out_up (atomically appended)
    uint32_t* count_up;
    uint32_t* overflow;            // optional: set to 1 when a ray did not fit into `capacity`
    uint32_t* leftover;            // optional: counts every ray this round hands on (the LAST round of a march: must stay 0)
    uint32_t capacity;             // entries each out list can hold
};
hipError_t launch_raymarch_slab(const RaymarchArgs& a, const SlabMarchArgs& s, hipStream_t stream);

hipError_t launch_raymarch(const RaymarchArgs& a, hipStream_t stream);
// would the launcher march over a pair (kind 3) / y-interleaved (kind 4) volume for these arguments, pointers apart?
bool march_volume_applicable(const RaymarchArgs& a, int kind);
// n cameras from a HOST array into DEVICE memory, stream-ordered, without a copy engine: launches that carry 32 cameras each
// in their kernel arguments (the host array is free again on return)
hipError_t launch_store_cameras(const sdfv_camera* host, uint32_t n, sdfv_camera* device, hipStream_t stream);

}  // namespace sdfv
