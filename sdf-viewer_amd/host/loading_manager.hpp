// loading_manager.hpp -- C++ mirror of the progressive-LOD iterator,
// reference src/app/scene/sdf/loading.rs:5-115 (struct LoadingManager, Iterator, ExactSizeIterator).
// Same field names and semantics; `pass_*` helpers expose a whole pass at once, which is the unit the
// GPU path launches (one kernel per pass instead of one sample() per next()).
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <optional>

namespace sdfviewer {

// loading.rs:108-115
inline uint32_t prev_power_of_2(uint32_t x) {
    x = x | (x >> 1);
    x = x | (x >> 2);
    x = x | (x >> 4);
    x = x | (x >> 8);
    x = x | (x >> 16);
    return x - (x >> 1);
}

class LoadingManager {
   public:
    using Index = std::array<size_t, 3>;

    // loading.rs:23-35
    LoadingManager(Index limits_, size_t passes_) : limits(limits_), passes(passes_) { reset(passes_); }

    // loading.rs:38-44
    void reset(size_t passes_) {
        passes = passes_;
        const uint32_t p = passes_ > 1 ? (uint32_t)passes_ : 1u;
        step_size_ = (size_t)1 << (p - 1);  // 2usize.pow(max(passes, 1) - 1)
        next_index_ = {0, 0, 0};
        iterations_ = 0;
        total_iterations_ = 0;
    }

    // Iterator::next, loading.rs:50-76
    std::optional<Index> next() {
        if (step_size_ == 0) return std::nullopt;
        iterations_ += 1;
        total_iterations_ += 1;
        const Index res = next_index_;
        next_index_[0] += step_size_;
        if (next_index_[0] >= limits[0]) {
            next_index_[0] = 0;
            next_index_[1] += step_size_;
            if (next_index_[1] >= limits[1]) {
                next_index_[1] = 0;
                next_index_[2] += step_size_;
                if (next_index_[2] >= limits[2]) {
                    step_size_ = prev_power_of_2((uint32_t)(step_size_ - 1));
                    next_index_ = {0, 0, 0};
                    iterations_ = 0;
                }
            }
        }
        return res;
    }

    // ExactSizeIterator::len, loading.rs:80-89
    size_t len() const {
        size_t step = step_size_, iterations = 0;
        while (step > 0) {
            iterations += pass_len(step);
            step = prev_power_of_2((uint32_t)(step - 1));
        }
        return iterations - iterations_;
    }

    size_t total_iterations() const { return total_iterations_; }  // loading.rs:93-95

    // loading.rs:99-105
    size_t passes_left() const {
        if (step_size_ == 0) return 0;
        return (size_t)std::log2((float)step_size_) + 1;
    }

    // ---- whole-pass view (GPU path) ----
    size_t step_size() const { return step_size_; }
    bool at_pass_start() const { return iterations_ == 0; }
    size_t pass_len(size_t step) const {
        return ((limits[0] + step - 1) / step) * ((limits[1] + step - 1) / step) * ((limits[2] + step - 1) / step);
    }
    // Consume the rest of the current pass exactly as pass_len - iterations calls of next() would.
    size_t finish_pass() {
        if (step_size_ == 0) return 0;
        const size_t n = pass_len(step_size_) - iterations_;
        total_iterations_ += n;
        step_size_ = prev_power_of_2((uint32_t)(step_size_ - 1));
        next_index_ = {0, 0, 0};
        iterations_ = 0;
        return n;
    }

    Index limits;   // pub(crate) limits
    size_t passes;  // pub(crate) passes

   private:
    size_t step_size_ = 0;
    Index next_index_{0, 0, 0};
    size_t iterations_ = 0;
    size_t total_iterations_ = 0;
};

}  // namespace sdfviewer
