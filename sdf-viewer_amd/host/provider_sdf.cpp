// provider_sdf.cpp -- see provider_sdf.hpp.  Every method follows the WasmerSDF method of the same name
// (reference src/sdf/wasm/native.rs) with "wasm linear memory" replaced by the process's own: call, copy what came back,
// hand the same pointer to the matching *_free if the library exports one.
#include "provider_sdf.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>

// (types only: the functions are resolved with dlsym, the header's prototypes are never linked against)
#include "../../include/sdf_provider.h"

namespace sdfviewer {

struct ProviderSDF::Library {
    void* handle = nullptr;
    SDFBoundingBox* (*f_bounding_box)(uint32_t) = nullptr;
    void (*f_bounding_box_free)(SDFBoundingBox*) = nullptr;
    ::SDFSample* (*f_sample)(uint32_t, SDFVec3, bool) = nullptr;
    void (*f_sample_free)(::SDFSample*) = nullptr;
    PointerLength* (*f_children)(uint32_t) = nullptr;
    void (*f_children_free)(PointerLength*) = nullptr;
    PointerLength* (*f_name)(uint32_t) = nullptr;
    void (*f_name_free)(PointerLength*) = nullptr;
    PointerLength* (*f_parameters)(uint32_t) = nullptr;
    void (*f_parameters_free)(PointerLength*) = nullptr;
    SDFSetParameterResult* (*f_set_parameter)(uint32_t, uint32_t, SDFParamValueC) = nullptr;
    void (*f_set_parameter_free)(SDFSetParameterResult*) = nullptr;
    SDFChangedResult* (*f_changed)(uint32_t) = nullptr;
    void (*f_changed_free)(SDFChangedResult*) = nullptr;
    SDFVec3* (*f_normal)(uint32_t, SDFVec3, float) = nullptr;
    void (*f_normal_free)(SDFVec3*) = nullptr;
    uint32_t (*f_sample_concurrency)(void) = nullptr;
    void (*f_sample_batch)(uint32_t, const SDFVec3*, size_t, bool, ::SDFSample*) = nullptr;
    ~Library() {
        if (handle) dlclose(handle);
    }
};

namespace {
template <typename F>
void resolve(void* handle, const char* symbol, F& out) {
    out = reinterpret_cast<F>(dlsym(handle, symbol));
}
std::string pl_string(const PointerLength& p) {
    return p.ptr ? std::string(static_cast<const char*>(p.ptr), p.len_bytes) : std::string();
}
}  // namespace

std::shared_ptr<ProviderSDF> ProviderSDF::load(const std::string& path, std::string* error) {
    auto lib = std::make_shared<Library>();
    lib->handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib->handle) {
        const char* why = dlerror();
        if (error) *error = std::string("cannot load SDF provider '") + path + "': " + (why ? why : "dlopen failed");
        return nullptr;
    }
    void* h = lib->handle;
    resolve(h, "bounding_box", lib->f_bounding_box);
    resolve(h, "sample", lib->f_sample);
    if (!lib->f_bounding_box || !lib->f_sample) {  // native.rs:60,62: the two required exports
        if (error) *error = std::string("SDF provider '") + path + "' does not export " + (lib->f_bounding_box ? "sample" : "bounding_box");
        return nullptr;
    }
    resolve(h, "bounding_box_free", lib->f_bounding_box_free);
    resolve(h, "sample_free", lib->f_sample_free);
    resolve(h, "children", lib->f_children);
    resolve(h, "children_free", lib->f_children_free);
    resolve(h, "name", lib->f_name);
    resolve(h, "name_free", lib->f_name_free);
    resolve(h, "parameters", lib->f_parameters);
    resolve(h, "parameters_free", lib->f_parameters_free);
    resolve(h, "set_parameter", lib->f_set_parameter);
    resolve(h, "set_parameter_free", lib->f_set_parameter_free);
    resolve(h, "changed", lib->f_changed);
    resolve(h, "changed_free", lib->f_changed_free);
    resolve(h, "normal", lib->f_normal);
    resolve(h, "normal_free", lib->f_normal_free);
    resolve(h, "sample_concurrency", lib->f_sample_concurrency);
    resolve(h, "sample_batch", lib->f_sample_batch);
    void (*f_init)(void) = nullptr;
    resolve(h, "init", f_init);  // "Call init() to initialize the module (optional)", native.rs:51-56
    if (f_init) f_init();
    return std::shared_ptr<ProviderSDF>(new ProviderSDF(std::move(lib), 0));  // the root SDF is id 0 (native.rs:78)
}

BoundingBox ProviderSDF::bounding_box() const {  // native.rs:164-186
    BoundingBox res{Vec3{0.0f, 0.0f, 0.0f}, Vec3{1.0f, 1.0f, 1.0f}};
    SDFBoundingBox* ret = lib_->f_bounding_box(sdf_id_);
    if (!ret) {
        fprintf(stderr, "Failed to get bounding box of SDF with ID %u\n", sdf_id_);
        return res;
    }
    res[0] = Vec3{ret->min.x, ret->min.y, ret->min.z};
    res[1] = Vec3{ret->max.x, ret->max.y, ret->max.z};
    if (lib_->f_bounding_box_free) lib_->f_bounding_box_free(ret);
    return res;
}

SDFSample ProviderSDF::sample(Vec3 p, bool distance_only) const {  // native.rs:188-217
    ::SDFSample* ret = lib_->f_sample(sdf_id_, SDFVec3{p.x, p.y, p.z}, distance_only);
    if (!ret) return SDFSample::make(1.0f, Vec3{});  // native.rs:203
    SDFSample s;
    static_assert(sizeof(s) == sizeof(*ret), "SDFSample is 28 bytes on both sides");
    memcpy(static_cast<void*>(&s), ret, sizeof(s));
    if (lib_->f_sample_free) lib_->f_sample_free(ret);
    return s;
}

void ProviderSDF::sample_batch(const Vec3* p, size_t n, bool distance_only, SDFSample* out) const {
    static_assert(sizeof(Vec3) == sizeof(SDFVec3) && sizeof(SDFSample) == sizeof(::SDFSample), "repr(C) on both sides");
    if (!lib_->f_sample_batch) return SDFSurface::sample_batch(p, n, distance_only, out);
    lib_->f_sample_batch(sdf_id_, reinterpret_cast<const SDFVec3*>(p), n, distance_only, reinterpret_cast<::SDFSample*>(out));
}

std::vector<std::shared_ptr<SDFSurface>> ProviderSDF::children() const {  // native.rs:219-253
    std::vector<std::shared_ptr<SDFSurface>> out;
    if (!lib_->f_children) return out;
    PointerLength* ret = lib_->f_children(sdf_id_);
    if (!ret) return out;
    const uint32_t* ids = static_cast<const uint32_t*>(ret->ptr);
    for (size_t i = 0; ids && i < ret->len_bytes / sizeof(uint32_t); ++i) {
        if (ids[i] == sdf_id_) {
            fprintf(stderr, "Children of SDF with ID %u include itself! Skipping, but this should be fixed.\n", sdf_id_);
            continue;
        }
        out.push_back(std::shared_ptr<ProviderSDF>(new ProviderSDF(lib_, ids[i])));
    }
    if (lib_->f_children_free) lib_->f_children_free(ret);
    return out;
}

std::string ProviderSDF::name() const {  // native.rs:259-281
    if (!lib_->f_name) return SDFSurface::name();
    PointerLength* ret = lib_->f_name(sdf_id_);
    if (!ret) return SDFSurface::name();
    std::string s = pl_string(*ret);
    if (lib_->f_name_free) lib_->f_name_free(ret);
    return s;
}

std::vector<SDFParam> ProviderSDF::parameters() const {  // native.rs:283-386
    std::vector<SDFParam> out;
    if (!lib_->f_parameters) return out;
    PointerLength* ret = lib_->f_parameters(sdf_id_);
    if (!ret) return out;
    const SDFParamC* params = static_cast<const SDFParamC*>(ret->ptr);
    for (size_t i = 0; params && i < ret->len_bytes / sizeof(SDFParamC); ++i) {
        const SDFParamC& c = params[i];
        SDFParam p;
        p.id = c.id;
        p.name = pl_string(c.name);
        p.description = pl_string(c.description);
        bool known = true;
        switch (c.kind.tag) {
        case 0: p.kind.tag = SDFParamKind::Tag::Boolean; break;
        case 1:
            p.kind.tag = SDFParamKind::Tag::Int;
            p.kind.int_lo = c.kind.v.int_.range_start;
            p.kind.int_hi = c.kind.v.int_.range_end;
            p.kind.int_step = c.kind.v.int_.step;
            break;
        case 2:
            p.kind.tag = SDFParamKind::Tag::Float;
            p.kind.float_lo = c.kind.v.float_.range_start;
            p.kind.float_hi = c.kind.v.float_.range_end;
            p.kind.float_step = c.kind.v.float_.step;
            break;
        case 3: {
            p.kind.tag = SDFParamKind::Tag::String;
            const PointerLength* items = static_cast<const PointerLength*>(c.kind.v.string_.choices.ptr);
            for (size_t k = 0; items && k < c.kind.v.string_.choices.len_bytes / sizeof(PointerLength); ++k)
                p.kind.choices.push_back(pl_string(items[k]));
            break;
        }
        default:  // native.rs: "Unknown SDF param kind enum type" -> the parameter is dropped
            fprintf(stderr, "Unknown SDF param kind enum type %u\n", c.kind.tag);
            known = false;
        }
        switch (c.value.tag) {
        case 0: p.value = (bool)c.value.v.boolean; break;
        case 1: p.value = (int32_t)c.value.v.int_; break;
        case 2: p.value = (float)c.value.v.float_; break;
        case 3: p.value = pl_string(c.value.v.string_); break;
        default:
            fprintf(stderr, "Unknown SDF param value enum type %u\n", c.value.tag);
            known = false;
        }
        if (known) out.push_back(std::move(p));
    }
    if (lib_->f_parameters_free) lib_->f_parameters_free(ret);
    return out;
}

SetParameterResult ProviderSDF::set_parameter(uint32_t param_id, const SDFParamValue& value) {  // native.rs:388-448
    if (!lib_->f_set_parameter) return SDFSurface::set_parameter(param_id, value);
    SDFParamValueC c;
    memset(&c, 0, sizeof(c));
    c.tag = (uint32_t)value.index();
    if (auto b = std::get_if<bool>(&value)) c.v.boolean = *b;
    else if (auto i = std::get_if<int32_t>(&value)) c.v.int_ = *i;
    else if (auto f = std::get_if<float>(&value)) c.v.float_ = *f;
    else {  // the string stays ours: the callee copies it (sdf_provider.h, "One deliberate difference")
        const std::string& s = std::get<std::string>(value);
        c.v.string_ = PointerLength{s.data(), s.size()};
    }
    SDFSetParameterResult* ret = lib_->f_set_parameter(sdf_id_, param_id, c);
    if (!ret) return SDFSurface::set_parameter(param_id, value);
    SetParameterResult res = SetParameterResult::Ok();
    if (ret->tag == 1) res = SetParameterResult::Err(pl_string(ret->error));
    else if (ret->tag != 0) res = SetParameterResult::Err("Unknown SDF set parameter result kind enum type");
    if (lib_->f_set_parameter_free) lib_->f_set_parameter_free(ret);
    return res;
}

std::optional<BoundingBox> ProviderSDF::changed() {  // native.rs:450-492: no export = None (not changed_default_impl)
    if (!lib_->f_changed) return std::nullopt;
    SDFChangedResult* ret = lib_->f_changed(sdf_id_);
    if (!ret) return std::nullopt;
    std::optional<BoundingBox> res;
    if (ret->tag == 1)
        res = BoundingBox{Vec3{ret->bounds.min.x, ret->bounds.min.y, ret->bounds.min.z},
                          Vec3{ret->bounds.max.x, ret->bounds.max.y, ret->bounds.max.z}};
    else if (ret->tag != 0)
        fprintf(stderr, "Unknown SDF changed result kind enum type %u\n", ret->tag);
    if (lib_->f_changed_free) lib_->f_changed_free(ret);
    return res;
}

Vec3 ProviderSDF::normal(Vec3 p, std::optional<float> eps) const {  // native.rs:494-521: no export = zero
    if (!lib_->f_normal) return Vec3{};
    SDFVec3* ret = lib_->f_normal(sdf_id_, SDFVec3{p.x, p.y, p.z}, eps ? *eps : -1.0f);
    if (!ret) return Vec3{};
    Vec3 n{ret->x, ret->y, ret->z};
    if (lib_->f_normal_free) lib_->f_normal_free(ret);
    return n;
}

unsigned ProviderSDF::sample_concurrency() const {
    const uint32_t n = lib_->f_sample_concurrency ? lib_->f_sample_concurrency() : 1u;
    return n ? n : 1u;
}

}  // namespace sdfviewer
