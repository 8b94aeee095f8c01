// api_internal.h -- shared by the translation units behind include/sdfgrid.h (not part of the ABI).
#pragma once

#include <cstdint>

#include "../../include/sdfgrid.h"

namespace sdfv {
// Formats the thread-local message sdfv_last_error() returns and hands `code` back.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// The first and the last slice of `slab` (the ones the z-neighbours need) in one launch; o0/o1 address the first
// owned slice.  Same texels as sdfv_fill_grid over those two slices.  Requires at least two owned slices.
int fill_boundary_slices(const sdfv_demo_params* params, uint32_t sdf_id, const sdfv_grid* slab, float* o0, float* o1,
                         void* stream);
}  // namespace sdfv
