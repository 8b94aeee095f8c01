// loading_manager.hpp -- the progressive-LOD schedule of a grid load as the GPU path sees it: a list of PASSES, each a
// lattice of the grid (every step-th voxel on every axis, step = 2^(passes-1) ... 2, 1), each launched as one kernel.
//
// It stands in for the reference's LoadingManager (src/app/scene/sdf/loading.rs:5-115), whose observable behaviour the drop-in
// SDFViewer needs -- the sequence next() yields (x fastest, then y, then z, pass after pass), len(), total_iterations(),
// passes_left() -- and the reference's own five unit tests (loading.rs:117-171) run against this class
// (tests/test_host_cpu.py).  The reference keeps a 3-D "next index" and carries it voxel by voxel; here the state is the
// pass and a CURSOR into it, the index is decoded from the cursor on demand, and everything a whole-pass launcher asks
// (how many voxels a pass holds, how many are left) is closed form.  Per-voxel next() exists for hosts that sample on the CPU.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <optional>

namespace sdfviewer {

// The largest power of two <= x (0 for 0): loading.rs:108-115 computes the same by bit smearing.
inline uint32_t prev_power_of_2(uint32_t x) { return x ? 1u << (31 - __builtin_clz(x)) : 0u; }

class LoadingManager {
   public:
    using Index = std::array<size_t, 3>;

    LoadingManager(Index limits_, size_t passes_) : limits(limits_), passes(passes_) { reset(passes_); }

    // Back to the first (coarsest) pass: step 2^(max(passes, 1) - 1), nothing handed out.
    void reset(size_t passes_) {
        passes = passes_;
        step_ = (size_t)1 << ((passes_ > 1 ? passes_ : 1) - 1);
        cursor_ = 0;
        handed_out_ = 0;
    }

    // The next voxel of the schedule, or nothing once every pass is through.
    std::optional<Index> next() {
        if (step_ == 0) return std::nullopt;
        const Index at = lattice_point(step_, cursor_);
        ++handed_out_;
        if (++cursor_ == pass_points(step_)) next_pass();
        return at;
    }

    // Voxels still to be handed out: the passes from the current one down to step 1, less the cursor.
    size_t len() const {
        size_t left = 0;
        for (size_t s = step_; s != 0; s >>= 1) left += pass_len(s);
        return left - cursor_;
    }

    size_t total_iterations() const { return handed_out_; }

    // 1 + log2(step) while loading, 0 when loaded: the shader reads it as lod_dist_between_samples = 2^passes_left.
    size_t passes_left() const { return step_ ? (size_t)(64 - __builtin_clzll((unsigned long long)step_)) : 0; }

    // ---- whole passes (what the GPU path launches) ----
    size_t step_size() const { return step_; }
    bool at_pass_start() const { return cursor_ == 0; }
    // Lattice points of a pass, as the reference's len() counts them: ceil(limit / step) per axis.
    size_t pass_len(size_t step) const { return axis_points(0, step) * axis_points(1, step) * axis_points(2, step); }
    // Hand out the rest of the current pass at once; returns how many voxels that were.
    size_t finish_pass() {
        if (step_ == 0) return 0;
        const size_t n = pass_len(step_) - cursor_;
        handed_out_ += n;
        next_pass();
        return n;
    }

    // ---- runs of a pass (what a host that samples on the CPU hands to its worker threads) ----
    size_t cursor() const { return cursor_; }
    // Points of the current pass not handed out yet (0 once loaded).
    size_t pass_remaining() const { return step_ ? pass_points(step_) - cursor_ : 0; }
    // The k-th point of the current pass, in next()'s order.
    Index pass_point(size_t k) const { return lattice_point(step_, k); }
    // Points per axis of the walk of the current pass (x fastest, then y, then z).
    Index pass_walk() const { return {walk_points(0, step_), walk_points(1, step_), walk_points(2, step_)}; }
    // Hand out the next n points of the current pass at once (n <= pass_remaining()): n calls of next().
    void advance(size_t n) {
        if (step_ == 0 || n == 0) return;
        handed_out_ += n;
        cursor_ += n;
        if (cursor_ >= pass_points(step_)) next_pass();
    }

    Index limits;   // voxels per axis
    size_t passes;  // as configured (--loading-passes)

   private:
    size_t axis_points(int axis, size_t step) const { return (limits[axis] + step - 1) / step; }
    // What next() walks through: an axis without voxels still yields its index 0 (the reference returns the index before it
    // tests the limit), so the walk uses at least one point per axis.  Never differs from pass_len on a grid that has voxels.
    size_t walk_points(int axis, size_t step) const { return axis_points(axis, step) ? axis_points(axis, step) : 1; }
    size_t pass_points(size_t step) const { return walk_points(0, step) * walk_points(1, step) * walk_points(2, step); }
    // The cursor-th point of the pass: x fastest, then y, then z.
    Index lattice_point(size_t step, size_t cursor) const {
        const size_t nx = walk_points(0, step), ny = walk_points(1, step);
        return {(cursor % nx) * step, (cursor / nx % ny) * step, (cursor / (nx * ny)) * step};
    }
    void next_pass() {
        step_ >>= 1;  // ... 4, 2, 1, 0 = loaded
        cursor_ = 0;
    }

    size_t step_ = 0;        // lattice spacing of the current pass; 0 = loaded
    size_t cursor_ = 0;      // points of the current pass already handed out
    size_t handed_out_ = 0;  // since the last reset
};

}  // namespace sdfviewer
