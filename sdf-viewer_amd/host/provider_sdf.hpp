// provider_sdf.hpp -- an SDFSurface backed by a native library that exports the reference's per-point SDF ABI
// (include/sdf_provider.h = src/sdf/ffi.rs:42-337).  The consumer side of that ABI: what WasmerSDF is for a .wasm module
// (src/sdf/wasm/native.rs:23-521) this class is for a shared object -- same required / optional exports, same defaults when an
// optional export is missing or a call fails, same copy-then-*_free ownership.
//
// Such an SDF can only be sampled point by point on the host (device_sdf() = nullopt): SDFViewer::update takes the ingest
// path for it (sdf_viewer_ingest.cpp).
#pragma once

#include <memory>
#include <string>

#include "sdf_surface.hpp"

namespace sdfviewer {

class ProviderSDF : public SDFSurface {
   public:
    // dlopen()s `path`, calls its init() if it exports one (native.rs:51-56) and returns the root SDF (id 0, native.rs:78).
    // bounding_box and sample are required (native.rs:60,62); nullptr + *error when the library or one of them is missing.
    // The provider's registry is thread-local (ffi.rs:15-17): use the object on the thread that loaded it.  A native library's
    // globals are the PROCESS's (dlopen counts references): two loads of one path share them -- and init() of the second resets
    // what the first set up -- where two instances of a wasm module would not.
    static std::shared_ptr<ProviderSDF> load(const std::string& path, std::string* error);

    BoundingBox bounding_box() const override;
    SDFSample sample(Vec3 p, bool distance_only) const override;
    // through the optional export `sample_batch` (sdf_provider.h) when the library has it, else the per-point loop
    void sample_batch(const Vec3* p, size_t n, bool distance_only, SDFSample* out) const override;
    std::vector<std::shared_ptr<SDFSurface>> children() const override;
    uint32_t id() const override { return sdf_id_; }
    std::string name() const override;
    std::vector<SDFParam> parameters() const override;
    SetParameterResult set_parameter(uint32_t param_id, const SDFParamValue& value) override;
    std::optional<BoundingBox> changed() override;
    Vec3 normal(Vec3 p, std::optional<float> eps) const override;
    // the optional export `uint32_t sample_concurrency(void)` (an extension, sdf_provider.h), else 1
    unsigned sample_concurrency() const override;

    struct Library;  // the dlopen handle and the resolved exports, shared by the root and the children it hands out

   private:
    ProviderSDF(std::shared_ptr<Library> lib, uint32_t sdf_id) : lib_(std::move(lib)), sdf_id_(sdf_id) {}
    std::shared_ptr<Library> lib_;
    uint32_t sdf_id_;
};

}  // namespace sdfviewer
