// sdf_demo.cpp -- see sdf_demo.hpp.
#include "sdf_demo.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace sdfviewer {

std::optional<Material> material_from_str(const std::string& s) {  // cube.rs:26-36
    std::string up = s;
    std::transform(up.begin(), up.end(), up.begin(), [](unsigned char c) { return (char)std::toupper(c); });
    if (up == "BRICK") return Material::Brick;
    if (up == "NORMAL") return Material::Normal;
    return std::nullopt;  // Err("Invalid cube material")
}

std::string material_to_string(Material m) { return m == Material::Brick ? "Brick" : "Normal"; }  // cube.rs:39-46

sdfv_demo_params DemoState::to_device() const {
    sdfv_demo_params p;
    p.cube_half_side = cube_half_side;
    p.cube_material = (uint32_t)cube_material;
    p.sphere_radius = sphere_radius;
    p.sphere_material = (uint32_t)sphere_material;
    p.max_distance_custom_material = max_distance_custom_material;
    p.disable_sphere = disable_sphere ? 1u : 0u;
    return p;
}

std::string param_value_debug(const SDFParamValue& v) {
    char buf[64];
    if (auto b = std::get_if<bool>(&v)) return std::string("Boolean(") + (*b ? "true" : "false") + ")";
    if (auto i = std::get_if<int32_t>(&v)) {
        snprintf(buf, sizeof(buf), "Int(%d)", *i);
        return buf;
    }
    if (auto f = std::get_if<float>(&v)) {
        snprintf(buf, sizeof(buf), "Float(%g)", (double)*f);
        std::string s = buf;
        if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos &&
            s.find("nan") == std::string::npos)
            s.insert(s.size() - 1, ".0");  // Rust prints 1.0, not 1
        return s;
    }
    return "String(\"" + std::get<std::string>(v) + "\")";
}

static SetParameterResult unknown_parameter(uint32_t param_id, const SDFParamValue& v) {
    // format!("Unknown parameter {param_id} with value {param_value:?}")  (demo/mod.rs:131)
    return SetParameterResult::Err("Unknown parameter " + std::to_string(param_id) + " with value " + param_value_debug(v));
}

// ---- sampling: one-point batches on the GPU ----
SDFSample SDFDemoBase::sample(Vec3 p, bool distance_only) const {
    sdfv_demo_params prm = st_->to_device();
    SDFSample out;
    if (sdfv_sample_points_host(&prm, id(), &p.x, 1, distance_only ? 1 : 0, reinterpret_cast<sdfv_sample*>(&out)) != 0) {
        fprintf(stderr, "SDFDemo::sample: %s\n", sdfv_last_error());  // wasm/native.rs:196-203: log + default
        return SDFSample::make(1.0f, Vec3{});
    }
    return out;
}

void SDFDemoBase::sample_batch(const Vec3* p, size_t n, bool distance_only, SDFSample* out) const {
    if (n == 0) return;
    sdfv_demo_params prm = st_->to_device();
    if (sdfv_sample_points_host(&prm, id(), &p[0].x, n, distance_only ? 1 : 0, reinterpret_cast<sdfv_sample*>(out)) != 0) {
        fprintf(stderr, "SDFDemo::sample_batch: %s\n", sdfv_last_error());
        for (size_t i = 0; i < n; ++i) out[i] = SDFSample::make(1.0f, Vec3{});
    }
}

Vec3 SDFDemoBase::normal(Vec3 p, std::optional<float> eps) const {
    // the demo's overrides (demo/mod.rs:147-156, cube.rs:164-177, sphere.rs:122-124) ignore eps
    (void)eps;
    sdfv_demo_params prm = st_->to_device();
    Vec3 out;
    if (sdfv_normal_points_host(&prm, id(), &p.x, 1, 0.0f, 0, &out.x) != 0) {
        fprintf(stderr, "SDFDemo::normal: %s\n", sdfv_last_error());
        return Vec3{};
    }
    return out;
}

// normal_default_impl, defaults.rs:49-56 (host arithmetic over 4 sample(.., true) calls)
Vec3 SDFSurface::normal(Vec3 p, std::optional<float> eps_opt) const {
    const float eps = eps_opt.value_or(0.001f);
    const float k[4][3] = {{1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, 1, 1}};
    Vec3 acc;
    for (int i = 0; i < 4; ++i) {
        Vec3 q{p.x + k[i][0] * eps, p.y + k[i][1] * eps, p.z + k[i][2] * eps};
        float d = sample(q, true).distance;
        Vec3 term{k[i][0] * d, k[i][1] * d, k[i][2] * d};
        acc = i == 0 ? term : Vec3{acc.x + term.x, acc.y + term.y, acc.z + term.z};
    }
    float inv = 1.0f / std::sqrt(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z);
    return Vec3{acc.x * inv, acc.y * inv, acc.z * inv};
}

BoundingBox merge_bounding_boxes(const BoundingBox& a, const BoundingBox& b) {
    return {Vec3{std::fmin(a[0].x, b[0].x), std::fmin(a[0].y, b[0].y), std::fmin(a[0].z, b[0].z)},
            Vec3{std::fmax(a[1].x, b[1].x), std::fmax(a[1].y, b[1].y), std::fmax(a[1].z, b[1].z)}};
}

static SDFParam material_param(uint32_t id, Material current, const char* desc) {
    SDFParam p;
    p.id = id;
    p.name = "material";
    p.kind.tag = SDFParamKind::Tag::String;
    p.kind.choices = {material_to_string(Material::Brick), material_to_string(Material::Normal)};
    p.value = material_to_string(current);
    p.description = desc;
    return p;
}

static std::optional<BoundingBox> take_flag(bool& flag, const BoundingBox& bb) {
    if (flag) {
        flag = false;
        return bb;
    }
    return std::nullopt;
}

// ---- cube ---- (cube.rs:101-161)
std::vector<SDFParam> SDFDemoCube::parameters() const {
    SDFParam half;
    half.id = ID_HALF_SIDE;
    half.name = "half_side";
    half.kind.tag = SDFParamKind::Tag::Int;  // "Should be float, but testing the int parameter"
    half.kind.int_lo = 0;
    half.kind.int_hi = 100;
    half.kind.int_step = 1;
    half.value = (int32_t)(st_->cube_half_side * 100.0f);
    half.description = "Half the length of a side of the cube (mapped from [0-100] to [0.0,1.0]).";
    return {material_param(ID_MATERIAL, st_->cube_material, "The material to use for the cube."), half};
}

SetParameterResult SDFDemoCube::set_parameter(uint32_t param_id, const SDFParamValue& value) {
    if (param_id == ID_MATERIAL) {
        if (auto s = std::get_if<std::string>(&value)) {
            auto m = material_from_str(*s);
            if (!m) return SetParameterResult::Err("Invalid cube material");  // the reference panics here (expect)
            st_->cube_material = *m;
            st_->cube_changed = true;
            return SetParameterResult::Ok();
        }
    } else if (param_id == ID_HALF_SIDE) {
        if (auto i = std::get_if<int32_t>(&value)) {
            st_->cube_half_side = (float)*i / 100.0f;
            st_->cube_changed = true;
            return SetParameterResult::Ok();
        }
    }
    return unknown_parameter(param_id, value);
}

std::optional<BoundingBox> SDFDemoCube::changed() { return take_flag(st_->cube_changed, bounding_box()); }

// ---- sphere ---- (sphere.rs:59-119)
std::vector<SDFParam> SDFDemoSphere::parameters() const {
    SDFParam r;
    r.id = ID_RADIUS;
    r.name = "sphere_radius";
    r.kind.tag = SDFParamKind::Tag::Float;
    r.kind.float_lo = 0.0f;
    r.kind.float_hi = 1.25f;
    r.kind.float_step = 0.01f;
    r.value = st_->sphere_radius;
    r.description = "The radius of the sphere.";
    return {material_param(ID_MATERIAL, st_->sphere_material, "The material to use for the sphere."), r};
}

SetParameterResult SDFDemoSphere::set_parameter(uint32_t param_id, const SDFParamValue& value) {
    if (param_id == ID_MATERIAL) {
        if (auto s = std::get_if<std::string>(&value)) {
            auto m = material_from_str(*s);
            if (!m) return SetParameterResult::Err("Invalid cube material");
            st_->sphere_material = *m;
            st_->sphere_changed = true;
            return SetParameterResult::Ok();
        }
    } else if (param_id == ID_RADIUS) {
        if (auto f = std::get_if<float>(&value)) {
            st_->sphere_radius = *f;
            st_->sphere_changed = true;
            return SetParameterResult::Ok();
        }
    }
    return unknown_parameter(param_id, value);
}

std::optional<BoundingBox> SDFDemoSphere::changed() { return take_flag(st_->sphere_changed, bounding_box()); }

// ---- demo ---- (demo/mod.rs:78-144)
std::vector<std::shared_ptr<SDFSurface>> SDFDemo::children() const {
    // "cheap clone with shared references to parameters (to receive modifications)"
    return {std::make_shared<SDFDemoCube>(st_), std::make_shared<SDFDemoSphere>(st_)};
}

std::vector<SDFParam> SDFDemo::parameters() const {
    SDFParam a;
    a.id = ID_MAX_DISTANCE_CUSTOM_MATERIAL;
    a.name = "max_distance_custom_material";
    a.kind.tag = SDFParamKind::Tag::Float;
    a.kind.float_lo = 0.0f;
    a.kind.float_hi = 0.25f;
    a.kind.float_step = 0.01f;
    a.value = st_->max_distance_custom_material;
    a.description = "The maximum distance between both surfaces at which the two materials are merged.";
    SDFParam b;
    b.id = ID_DISABLE_SPHERE;
    b.name = "disable_sphere";
    b.kind.tag = SDFParamKind::Tag::Boolean;
    b.value = st_->disable_sphere;
    b.description = "Whether to hide the sphere or not.";
    return {a, b};
}

SetParameterResult SDFDemo::set_parameter(uint32_t param_id, const SDFParamValue& value) {
    if (param_id == ID_MAX_DISTANCE_CUSTOM_MATERIAL) {
        if (auto f = std::get_if<float>(&value)) {
            st_->max_distance_custom_material = *f;
            st_->demo_changed = true;
            return SetParameterResult::Ok();
        }
    } else if (param_id == ID_DISABLE_SPHERE) {
        if (auto b = std::get_if<bool>(&value)) {
            st_->disable_sphere = *b;
            st_->demo_changed = true;
            return SetParameterResult::Ok();
        }
    }
    return unknown_parameter(param_id, value);
}

std::optional<BoundingBox> SDFDemo::changed() {
    // changed_default_impl(self).or_else(own flag), demo/mod.rs:135-144
    if (auto b = SDFSurface::changed()) return b;
    return take_flag(st_->demo_changed, bounding_box());
}

std::shared_ptr<SDFDemo> SDFDemo::from_args(const std::vector<std::string>& args, std::string* error) {
    auto st = std::make_shared<DemoState>();
    auto fail = [&](const std::string& e) {
        if (error) *error = e;
        return std::shared_ptr<SDFDemo>();
    };
    for (size_t i = 0; i < args.size(); ++i) {
        std::string a = args[i], v;
        auto eq = a.find('=');
        bool has_inline = a.rfind("--", 0) == 0 && eq != std::string::npos;
        if (has_inline) {
            v = a.substr(eq + 1);
            a = a.substr(0, eq);
        }
        auto value = [&]() -> std::optional<std::string> {
            if (has_inline) return v;
            if (i + 1 < args.size()) return args[++i];
            return std::nullopt;
        };
        auto need = [&](const char* what) -> std::optional<std::string> {
            auto x = value();
            if (!x) fail(std::string("The argument '") + what + "' requires a value but none was supplied");
            return x;
        };
        if (a == "-t" || a == "--cube-material") {
            auto x = need("--cube-material <CUBE_MATERIAL>");
            if (!x) return nullptr;
            auto m = material_from_str(*x);
            if (!m) return fail("Invalid value \"" + *x + "\" for '--cube-material <CUBE_MATERIAL>': Invalid cube material");
            st->cube_material = *m;
        } else if (a == "-l" || a == "--sphere-material") {
            auto x = need("--sphere-material <SPHERE_MATERIAL>");
            if (!x) return nullptr;
            auto m = material_from_str(*x);
            if (!m) return fail("Invalid value \"" + *x + "\" for '--sphere-material <SPHERE_MATERIAL>': Invalid cube material");
            st->sphere_material = *m;
        } else if (a == "-c" || a == "--cube-half-side" || a == "-s" || a == "--sphere-radius" || a == "-m" ||
                   a == "--max-distance-custom-material") {
            auto x = need(a.c_str());
            if (!x) return nullptr;
            char* end = nullptr;
            float f = strtof(x->c_str(), &end);
            if (end == x->c_str() || *end) return fail("Invalid value \"" + *x + "\" for '" + a + "': invalid float literal");
            if (a == "-c" || a == "--cube-half-side") st->cube_half_side = f;
            else if (a == "-s" || a == "--sphere-radius") st->sphere_radius = f;
            else st->max_distance_custom_material = f;
        } else if (a == "-d" || a == "--disable-sphere") {
            auto x = need("--disable-sphere <DISABLE_SPHERE>");
            if (!x) return nullptr;
            if (*x == "true") st->disable_sphere = true;
            else if (*x == "false") st->disable_sphere = false;
            else return fail("Invalid value \"" + *x + "\" for '--disable-sphere <DISABLE_SPHERE>': provided string was not `true` or `false`");
        } else {
            return fail("Found argument '" + a + "' which wasn't expected, or isn't valid in this context");
        }
    }
    return std::make_shared<SDFDemo>(st);
}

}  // namespace sdfviewer
