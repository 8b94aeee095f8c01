// provider.cpp -- libsdfdemo_provider.so: the reference's per-point SDF ABI (src/sdf/ffi.rs:42-337) over the
// C++ mirror of the demo SDF.  Symbol names, argument order, ownership and error behaviour follow ffi.rs;
// see include/sdf_provider.h.
#include "../../include/sdf_provider.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "sdf_demo.hpp"

namespace sv = sdfviewer;
using sv::BoundingBox;
using sv::SDFDemo;
using sv::SDFParamKind;
using sv::SDFParamValue;
using sv::SDFSurface;
using sv::SetParameterResult;
using sv::Vec3;

namespace {

// thread_local! { static REGISTRY: RefCell<HashMap<u32, Box<dyn SDFSurface>>> }  (ffi.rs:15-17)
thread_local std::map<uint32_t, std::shared_ptr<SDFSurface>> g_registry;

// set_root_sdf, ffi.rs:20-33: the root and all of its descendants, keyed by id()
void set_root_sdf(std::shared_ptr<SDFSurface> root) {
    g_registry.clear();
    std::vector<std::shared_ptr<SDFSurface>> to_process{std::move(root)};
    while (!to_process.empty()) {
        auto cur = to_process.back();
        to_process.pop_back();
        for (auto& ch : cur->children()) to_process.push_back(ch);
        g_registry[cur->id()] = cur;
    }
}

SDFSurface* find(uint32_t sdf_id) {
    auto it = g_registry.find(sdf_id);
    if (it == g_registry.end()) {
        fprintf(stderr, "Failed to find SDF with ID %u\n", sdf_id);  // ffi.rs:47
        return nullptr;
    }
    return it->second.get();
}

// PointerLength::from_vec, ffi.rs:84-90: leak a byte copy until *_free
PointerLength pl_from_bytes(const void* data, size_t len_bytes) {
    PointerLength p;
    void* mem = len_bytes ? malloc(len_bytes) : nullptr;
    if (len_bytes) memcpy(mem, data, len_bytes);
    p.ptr = mem;
    p.len_bytes = len_bytes;
    return p;
}
PointerLength pl_from_string(const std::string& s) { return pl_from_bytes(s.data(), s.size()); }
PointerLength pl_null() { return PointerLength{nullptr, 0}; }  // ffi.rs:101-107
void pl_free(const PointerLength& p) { free(const_cast<void*>(p.ptr)); }  // own_again + drop

SDFParamKindC kind_from_api(const SDFParamKind& k) {  // ffi.rs:188-199
    SDFParamKindC c;
    memset(&c, 0, sizeof(c));
    c.tag = (uint32_t)k.tag;
    switch (k.tag) {
    case SDFParamKind::Tag::Boolean: break;
    case SDFParamKind::Tag::Int:
        c.v.int_.range_start = k.int_lo;
        c.v.int_.range_end = k.int_hi;
        c.v.int_.step = k.int_step;
        break;
    case SDFParamKind::Tag::Float:
        c.v.float_.range_start = k.float_lo;
        c.v.float_.range_end = k.float_hi;
        c.v.float_.step = k.float_step;
        break;
    case SDFParamKind::Tag::String: {
        std::vector<PointerLength> items;
        for (auto& s : k.choices) items.push_back(pl_from_string(s));
        c.v.string_.choices = pl_from_bytes(items.data(), items.size() * sizeof(PointerLength));
        break;
    }
    }
    return c;
}

SDFParamValueC value_from_api(const SDFParamValue& v) {  // ffi.rs:212-221
    SDFParamValueC c;
    memset(&c, 0, sizeof(c));
    c.tag = (uint32_t)v.index();
    if (auto b = std::get_if<bool>(&v)) c.v.boolean = *b;
    else if (auto i = std::get_if<int32_t>(&v)) c.v.int_ = *i;
    else if (auto f = std::get_if<float>(&v)) c.v.float_ = *f;
    else c.v.string_ = pl_from_string(std::get<std::string>(v));
    return c;
}

// ffi.rs:223-231.  Deliberate difference, documented in sdf_provider.h: a String payload is COPIED and stays the caller's
// (the reference re-owns it as a Rust Vec, which is only sound when caller and callee share an allocator); nothing is freed
// here and nothing leaks on this side.
SDFParamValue value_to_api(const SDFParamValueC& c) {
    switch (c.tag) {
    case 0: return SDFParamValue{(bool)c.v.boolean};
    case 1: return SDFParamValue{(int32_t)c.v.int_};
    case 2: return SDFParamValue{(float)c.v.float_};
    default: return SDFParamValue{std::string(static_cast<const char*>(c.v.string_.ptr), c.v.string_.len_bytes)};
    }
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

void init(void) { set_root_sdf(std::make_shared<SDFDemo>()); }  // demo/ffi.rs:5-8

int init_with_args(int argc, const char* const* argv) {
    std::vector<std::string> args(argv, argv + (argc > 0 ? argc : 0));
    std::string err;
    auto demo = SDFDemo::from_args(args, &err);
    if (!demo) {
        fprintf(stderr, "error: %s\n", err.c_str());
        return -1;
    }
    set_root_sdf(demo);
    return 0;
}

// the registry is thread-local (ffi.rs:15-17) and sample() drives the GPU through one-point batches: one thread
uint32_t sample_concurrency(void) { return 1; }

SDFBoundingBox* bounding_box(uint32_t sdf_id) {
    auto* ret = static_cast<SDFBoundingBox*>(calloc(1, sizeof(SDFBoundingBox)));  // [Vector3::zero(); 2] on failure
    if (auto* sdf = find(sdf_id)) {
        BoundingBox bb = sdf->bounding_box();
        ret->min = SDFVec3{bb[0].x, bb[0].y, bb[0].z};
        ret->max = SDFVec3{bb[1].x, bb[1].y, bb[1].z};
    }
    return ret;
}
void bounding_box_free(SDFBoundingBox* ret) { free(ret); }

SDFSample* sample(uint32_t sdf_id, SDFVec3 p, bool distance_only) {
    auto* ret = static_cast<SDFSample*>(calloc(1, sizeof(SDFSample)));  // SDFSample::new(0.0, zero) on failure
    if (auto* sdf = find(sdf_id)) {
        sv::SDFSample s = sdf->sample(Vec3{p.x, p.y, p.z}, distance_only);
        memcpy(ret, &s, sizeof(SDFSample));
    }
    return ret;
}
void sample_free(SDFSample* ret) { free(ret); }

// extension (sdf_provider.h): n points for the price of one call -- for this provider, of one device batch
void sample_batch(uint32_t sdf_id, const SDFVec3* points, size_t n, bool distance_only, SDFSample* out) {
    static_assert(sizeof(SDFVec3) == sizeof(Vec3) && sizeof(SDFSample) == sizeof(sv::SDFSample), "repr(C) on both sides");
    if (auto* sdf = find(sdf_id)) sdf->sample_batch(reinterpret_cast<const Vec3*>(points), n, distance_only, reinterpret_cast<sv::SDFSample*>(out));
    else if (n) memset(out, 0, n * sizeof(SDFSample));
}

PointerLength* children(uint32_t sdf_id) {
    auto* ret = static_cast<PointerLength*>(malloc(sizeof(PointerLength)));
    *ret = pl_null();
    if (auto* sdf = find(sdf_id)) {
        std::vector<uint32_t> ids;
        for (auto& ch : sdf->children()) ids.push_back(ch->id());
        *ret = pl_from_bytes(ids.data(), ids.size() * sizeof(uint32_t));
    }
    return ret;
}
void children_free(PointerLength* ret) {
    if (!ret) return;
    pl_free(*ret);
    free(ret);
}

PointerLength* name(uint32_t sdf_id) {
    auto* ret = static_cast<PointerLength*>(malloc(sizeof(PointerLength)));
    *ret = pl_null();
    if (auto* sdf = find(sdf_id)) *ret = pl_from_string(sdf->name());
    return ret;
}
void name_free(PointerLength* ret) {
    if (!ret) return;
    pl_free(*ret);
    free(ret);
}

PointerLength* parameters(uint32_t sdf_id) {
    auto* ret = static_cast<PointerLength*>(malloc(sizeof(PointerLength)));
    *ret = pl_null();
    if (auto* sdf = find(sdf_id)) {
        std::vector<SDFParamC> out;
        for (auto& p : sdf->parameters()) {
            SDFParamC c;
            memset(&c, 0, sizeof(c));
            c.id = p.id;
            c.name = pl_from_string(p.name);
            c.kind = kind_from_api(p.kind);
            c.value = value_from_api(p.value);
            c.description = pl_from_string(p.description);
            out.push_back(c);
        }
        *ret = pl_from_bytes(out.data(), out.size() * sizeof(SDFParamC));
    }
    return ret;
}
void parameters_free(PointerLength* ret) {  // ffi.rs:258-283
    if (!ret) return;
    auto* params = static_cast<const SDFParamC*>(ret->ptr);
    for (size_t i = 0; i < ret->len_bytes / sizeof(SDFParamC); ++i) {
        pl_free(params[i].name);
        pl_free(params[i].description);
        if (params[i].kind.tag == 3) {
            auto* items = static_cast<const PointerLength*>(params[i].kind.v.string_.choices.ptr);
            for (size_t k = 0; k < params[i].kind.v.string_.choices.len_bytes / sizeof(PointerLength); ++k) pl_free(items[k]);
            pl_free(params[i].kind.v.string_.choices);
        }
        if (params[i].value.tag == 3) pl_free(params[i].value.v.string_);
    }
    pl_free(*ret);
    free(ret);
}

SDFSetParameterResult* set_parameter(uint32_t sdf_id, uint32_t param_id, SDFParamValueC value) {
    auto* ret = static_cast<SDFSetParameterResult*>(calloc(1, sizeof(SDFSetParameterResult)));
    std::string err;
    if (auto* sdf = find(sdf_id)) {
        SetParameterResult r = sdf->set_parameter(param_id, value_to_api(value));
        if (!r.ok) err = r.error;
        else return ret;  // Ok(())
    } else {
        err = "Failed to find SDF with ID " + std::to_string(sdf_id);  // ffi.rs:294-296
    }
    ret->tag = 1;
    ret->error = pl_from_string(err);
    return ret;
}
void set_parameter_free(SDFSetParameterResult* ret) {
    if (!ret) return;
    if (ret->tag == 1) pl_free(ret->error);
    free(ret);
}

SDFChangedResult* changed(uint32_t sdf_id) {
    auto* ret = static_cast<SDFChangedResult*>(calloc(1, sizeof(SDFChangedResult)));  // None
    if (auto* sdf = find(sdf_id)) {
        if (auto bb = sdf->changed()) {
            ret->tag = 1;
            ret->bounds.min = SDFVec3{(*bb)[0].x, (*bb)[0].y, (*bb)[0].z};
            ret->bounds.max = SDFVec3{(*bb)[1].x, (*bb)[1].y, (*bb)[1].z};
        }
    }
    return ret;
}
void changed_free(SDFChangedResult* ret) { free(ret); }

SDFVec3* normal(uint32_t sdf_id, SDFVec3 p, float eps) {
    auto* ret = static_cast<SDFVec3*>(calloc(1, sizeof(SDFVec3)));  // Vector3::zero() on failure
    if (auto* sdf = find(sdf_id)) {
        Vec3 n = sdf->normal(Vec3{p.x, p.y, p.z}, eps > 0.0f ? std::optional<float>(eps) : std::nullopt);  // ffi.rs:326
        *ret = SDFVec3{n.x, n.y, n.z};
    }
    return ret;
}
void normal_free(SDFVec3* ret) { free(ret); }

}  // extern "C"
#pragma GCC visibility pop
