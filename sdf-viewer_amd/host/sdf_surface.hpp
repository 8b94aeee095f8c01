// sdf_surface.hpp -- C++ mirror of the reference's SDF abstraction (L1 in SURVEY.md):
//   trait SDFSurface            src/sdf/mod.rs:32-101   (same method names, argument meaning, defaults)
//   struct SDFSample            src/sdf/mod.rs:104-126
//   SDFParam / Kind / Value     src/sdf/mod.rs:129-173
//   default method bodies       src/sdf/defaults.rs:5-72
// The reference is Rust; this image has no Rust toolchain, so the host side above the C ABI is C++
// (see INTEGRATION.md for the `extern "C"` block a Rust maintainer would write instead).
//
// One addition over the trait: device_params(), through which an SDF that libsdfgrid can evaluate on
// the GPU describes itself to the batched kernels (the "Batched sampling" TODO, src/sdf/mod.rs:39).
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <variant>
#include <vector>

#include "../../include/sdfgrid.h"

namespace sdfviewer {

struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};
using BoundingBox = std::array<Vec3, 2>;  // [min, max]

// #[repr(C)] pub struct SDFSample -- layout-compatible with sdfv_sample (28 bytes)
struct SDFSample {
    float distance = 0.0f;
    Vec3 color;
    float metallic = 0.0f;
    float roughness = 0.0f;
    float occlusion = 0.0f;
    // SDFSample::new(distance, color): the other properties default to 0 (src/sdf/mod.rs:120-126)
    static SDFSample make(float distance, Vec3 color) {
        SDFSample s;
        s.distance = distance;
        s.color = color;
        return s;
    }
};
static_assert(sizeof(SDFSample) == sizeof(sdfv_sample), "SDFSample must stay repr(C)-compatible");

// src/sdf/mod.rs:143-162
struct SDFParamKind {
    enum class Tag : uint32_t { Boolean = 0, Int = 1, Float = 2, String = 3 };
    Tag tag = Tag::Boolean;
    int32_t int_lo = 0, int_hi = 0, int_step = 0;          // Int { range: lo..=hi, step }
    float float_lo = 0.0f, float_hi = 0.0f, float_step = 0.0f;  // Float { range, step }
    std::vector<std::string> choices;                      // String { choices }
};

// src/sdf/mod.rs:165-173
using SDFParamValue = std::variant<bool, int32_t, float, std::string>;

// src/sdf/mod.rs:129-141
struct SDFParam {
    uint32_t id = 0;
    std::string name;
    SDFParamKind kind;
    SDFParamValue value;
    std::string description;
};

// Result<(), String> of set_parameter
struct SetParameterResult {
    bool ok = true;
    std::string error;
    static SetParameterResult Ok() { return {}; }
    static SetParameterResult Err(std::string e) { return {false, std::move(e)}; }
};

// What libsdfgrid needs to evaluate an SDF on the device.
struct DeviceSDF {
    sdfv_demo_params params;
    uint32_t sdf_id;
};

class SDFSurface {
   public:
    virtual ~SDFSurface() = default;

    // ============ REQUIRED CORE ============ (src/sdf/mod.rs:34-43)
    virtual BoundingBox bounding_box() const = 0;
    virtual SDFSample sample(Vec3 p, bool distance_only) const = 0;

    // ============ OPTIONAL: HIERARCHY ============ (defaults.rs:7-21)
    virtual std::vector<std::shared_ptr<SDFSurface>> children() const { return {}; }
    virtual uint32_t id() const { return 0; }
    virtual std::string name() const { return "Object"; }

    // ============ OPTIONAL: PARAMETERS ============ (defaults.rs:25-45)
    virtual std::vector<SDFParam> parameters() const { return {}; }
    virtual SetParameterResult set_parameter(uint32_t /*param_id*/, const SDFParamValue& /*value*/) {
        return SetParameterResult::Err("no parameters implemented by default, overwrite this method");
    }
    // changed_default_impl: the first child reporting a change wins, the others report on later calls
    virtual std::optional<BoundingBox> changed() {
        for (auto& ch : children()) {
            if (auto b = ch->changed()) return b;
        }
        return std::nullopt;
    }

    // ============ OPTIONAL: UTILITIES ============ (defaults.rs:49-56)
    virtual Vec3 normal(Vec3 p, std::optional<float> eps) const;

    // ============ batched sampling (the trait's own TODO, src/sdf/mod.rs:39: "Batched sampling to speed up operations") ============
    // out[i] = sample(p[i], distance_only) for i in [0, n).  The default IS that loop; an SDF that can answer many points for
    // the price of one call overrides it (ProviderSDF: one FFI call and no allocation per point when the library exports
    // `sample_batch`; the demo SDF: one batch of the device kernel).  SDFViewer::update's host path samples through it.
    virtual void sample_batch(const Vec3* p, size_t n, bool distance_only, SDFSample* out) const {
        for (size_t i = 0; i < n; ++i) out[i] = sample(p[i], distance_only);
    }

    // ============ host sampling (not in the reference) ============
    // How many host threads may call sample() on this object at once.  1 (the default) = only the thread that calls
    // SDFViewer::update, which is all the reference's trait promises (its SDFs are !Send: Rc<RefCell>, demo/mod.rs:160-198,
    // and its loop is single-threaded with "TODO: Cross-platform parallel iteration?", scene/sdf/mod.rs:174).  A stateless
    // SDF answers with the parallelism it tolerates; SDFViewer::update then samples a pass on that many threads.
    virtual unsigned sample_concurrency() const { return 1; }

    // ============ batched / device evaluation (not in the reference) ============
    // nullopt = this SDF can only be sampled point by point on the host (e.g. a wasm provider).
    virtual std::optional<DeviceSDF> device_sdf() const { return std::nullopt; }
};

// merge_bounding_boxes, defaults.rs:59-72
BoundingBox merge_bounding_boxes(const BoundingBox& a, const BoundingBox& b);

}  // namespace sdfviewer
