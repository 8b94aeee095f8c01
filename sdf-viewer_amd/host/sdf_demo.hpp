// sdf_demo.hpp -- C++ mirror of the embedded demo SDF provider (L0 in SURVEY.md):
//   SDFDemo        src/sdf/demo/mod.rs:20-157   (id 0, "Demo")
//   SDFDemoCube    src/sdf/demo/cube.rs:13-178  (id 1, "DemoCube")
//   SDFDemoSphere  src/sdf/demo/sphere.rs:9-125 (id 2, "DemoSphere")
// Parameters live in shared cells (Rc<RefCell<..>> in the reference, demo/mod.rs:160-198) so that the
// children handed out by SDFDemo::children() see and make the same modifications.  All arithmetic of
// sample()/normal() runs on the GPU through libsdfgrid (one-point batches here; use the batched calls
// or SDFViewer for anything hot).
#pragma once

#include <memory>

#include "sdf_surface.hpp"

namespace sdfviewer {

enum class Material : uint32_t { Brick = SDFV_MATERIAL_BRICK, Normal = SDFV_MATERIAL_NORMAL };  // cube.rs:20-24
// FromStr / Display, cube.rs:26-46
std::optional<Material> material_from_str(const std::string& s);
std::string material_to_string(Material m);

// The state shared by SDFDemo and the children it hands out.
struct DemoState {
    Material cube_material = Material::Brick;   // -t, cube.rs:15
    float cube_half_side = 0.95f;               // -c, cube.rs:17
    Material sphere_material = Material::Normal;  // -l, sphere.rs:11
    float sphere_radius = 1.05f;                // -s, sphere.rs:13
    float max_distance_custom_material = 0.05f;  // -m, demo/mod.rs:26
    bool disable_sphere = false;                // -d, demo/mod.rs:28
    bool cube_changed = false, sphere_changed = false, demo_changed = false;
    sdfv_demo_params to_device() const;
};

class SDFDemoBase : public SDFSurface {
   public:
    explicit SDFDemoBase(std::shared_ptr<DemoState> st) : st_(std::move(st)) {}
    BoundingBox bounding_box() const override { return {Vec3{-1, -1, -1}, Vec3{1, 1, 1}}; }
    SDFSample sample(Vec3 p, bool distance_only) const override;
    void sample_batch(const Vec3* p, size_t n, bool distance_only, SDFSample* out) const override;  // ONE device batch
    Vec3 normal(Vec3 p, std::optional<float> eps) const override;
    std::optional<DeviceSDF> device_sdf() const override { return DeviceSDF{st_->to_device(), id()}; }
    const std::shared_ptr<DemoState>& state() const { return st_; }

   protected:
    std::shared_ptr<DemoState> st_;
};

class SDFDemoCube : public SDFDemoBase {
   public:
    static constexpr uint32_t ID_MATERIAL = 0, ID_HALF_SIDE = 1;
    using SDFDemoBase::SDFDemoBase;
    SDFDemoCube() : SDFDemoBase(std::make_shared<DemoState>()) {}
    uint32_t id() const override { return 1; }
    std::string name() const override { return "DemoCube"; }
    std::vector<SDFParam> parameters() const override;
    SetParameterResult set_parameter(uint32_t param_id, const SDFParamValue& value) override;
    std::optional<BoundingBox> changed() override;
};

class SDFDemoSphere : public SDFDemoBase {
   public:
    static constexpr uint32_t ID_MATERIAL = 0, ID_RADIUS = 1;
    using SDFDemoBase::SDFDemoBase;
    SDFDemoSphere() : SDFDemoBase(std::make_shared<DemoState>()) {}
    uint32_t id() const override { return 2; }
    std::string name() const override { return "DemoSphere"; }
    std::vector<SDFParam> parameters() const override;
    SetParameterResult set_parameter(uint32_t param_id, const SDFParamValue& value) override;
    std::optional<BoundingBox> changed() override;
};

class SDFDemo : public SDFDemoBase {
   public:
    static constexpr uint32_t ID_MAX_DISTANCE_CUSTOM_MATERIAL = 0, ID_DISABLE_SPHERE = 1;
    SDFDemo() : SDFDemoBase(std::make_shared<DemoState>()) {}
    explicit SDFDemo(std::shared_ptr<DemoState> st) : SDFDemoBase(std::move(st)) {}
    // clap-style construction from the reference's flags: -t -c -l -s -m -d (long names accepted too)
    static std::shared_ptr<SDFDemo> from_args(const std::vector<std::string>& args, std::string* error);
    uint32_t id() const override { return 0; }
    std::string name() const override { return "Demo"; }
    std::vector<std::shared_ptr<SDFSurface>> children() const override;
    std::vector<SDFParam> parameters() const override;
    SetParameterResult set_parameter(uint32_t param_id, const SDFParamValue& value) override;
    std::optional<BoundingBox> changed() override;
};

std::string param_value_debug(const SDFParamValue& v);  // Rust {:?} of SDFParamValue

}  // namespace sdfviewer
