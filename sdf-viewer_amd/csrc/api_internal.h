// api_internal.h -- shared by the translation units behind include/sdfgrid.h (not part of the ABI).
#pragma once

#include <cstdint>

#include "../../include/sdfgrid.h"

namespace sdfv {
// The calling thread's options (sdfv_set_option); every entry point copies what it needs once per call.
struct Options {
    bool fill_nontemporal = false;
    uint32_t fill_form = 0;          // 0 auto, 1 rows, 2 flat
    uint32_t raymarch_disable = 0;   // SDFV_RM_NO_*
    bool raymarch_keep_normal = false;
    uint32_t slab_step_form = 0;     // 0 auto, SDFV_STEP_* otherwise
    unsigned long long wave_timing = 0;  // tuning build only
};
const Options& options();
// Formats the thread-local message sdfv_last_error() returns and hands `code` back.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// The first and the last slice of `slab` (the ones the z-neighbours need) in one launch; o0/o1 address the first
// owned slice.  Same texels as sdfv_fill_grid over those two slices.  Requires at least two owned slices.
int fill_boundary_slices(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab, float* o0, float* o1,
                         void* stream);
}  // namespace sdfv
