// host_capi.cpp -- flat C handles over the C++ host mirror (LoadingManager, SDFDemo, SDFViewer,
// SDFViewerMaterial) so that the pytest suite can drive the same classes a C++ application links.
// Test/tooling surface only; the product ABI is include/sdfgrid.h + include/sdf_provider.h.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <sstream>

#include "loading_manager.hpp"
#include "mesh.hpp"
#include "provider_sdf.hpp"
#include "sdf_demo.hpp"
#include "scene.hpp"
#include "sdf_viewer.hpp"

using namespace sdfviewer;

#pragma GCC visibility push(default)
extern "C" {

// ---- LoadingManager ----
void* sdfvh_lm_new(size_t lx, size_t ly, size_t lz, size_t passes) { return new LoadingManager({lx, ly, lz}, passes); }
void sdfvh_lm_free(void* m) { delete static_cast<LoadingManager*>(m); }
int sdfvh_lm_next(void* m, size_t out[3]) {
    auto r = static_cast<LoadingManager*>(m)->next();
    if (!r) return 0;
    out[0] = (*r)[0]; out[1] = (*r)[1]; out[2] = (*r)[2];
    return 1;
}
size_t sdfvh_lm_len(void* m) { return static_cast<LoadingManager*>(m)->len(); }
size_t sdfvh_lm_total_iterations(void* m) { return static_cast<LoadingManager*>(m)->total_iterations(); }
size_t sdfvh_lm_passes_left(void* m) { return static_cast<LoadingManager*>(m)->passes_left(); }
size_t sdfvh_lm_step_size(void* m) { return static_cast<LoadingManager*>(m)->step_size(); }
size_t sdfvh_lm_finish_pass(void* m) { return static_cast<LoadingManager*>(m)->finish_pass(); }
void sdfvh_lm_advance(void* m, size_t n) { static_cast<LoadingManager*>(m)->advance(n); }
size_t sdfvh_lm_cursor(void* m) { return static_cast<LoadingManager*>(m)->cursor(); }
size_t sdfvh_lm_pass_remaining(void* m) { return static_cast<LoadingManager*>(m)->pass_remaining(); }
void sdfvh_lm_pass_point(void* m, size_t k, size_t out[3]) {
    auto p = static_cast<LoadingManager*>(m)->pass_point(k);
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}
uint32_t sdfvh_prev_power_of_2(uint32_t x) { return prev_power_of_2(x); }

// ---- SDFDemo (handle = shared_ptr<SDFSurface>*) ----
void* sdfvh_demo_new(int argc, const char* const* argv, char* err, size_t err_len) {
    std::vector<std::string> args(argv, argv + (argc > 0 ? argc : 0));
    std::string e;
    auto d = SDFDemo::from_args(args, &e);
    if (!d) {
        if (err && err_len) {
            strncpy(err, e.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return nullptr;
    }
    return new std::shared_ptr<SDFSurface>(d);
}
// ---- ProviderSDF (same handle type) ----
void* sdfvh_provider_load(const char* path, char* err, size_t err_len) {
    std::string e;
    auto p = ProviderSDF::load(path ? path : "", &e);
    if (!p) {
        if (err && err_len) {
            strncpy(err, e.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return nullptr;
    }
    return new std::shared_ptr<SDFSurface>(p);
}
void sdfvh_sdf_free(void* h) { delete static_cast<std::shared_ptr<SDFSurface>*>(h); }
static SDFSurface& S(void* h) { return **static_cast<std::shared_ptr<SDFSurface>*>(h); }
uint32_t sdfvh_sdf_id(void* h) { return S(h).id(); }
size_t sdfvh_sdf_name(void* h, char* out, size_t n) {
    std::string s = S(h).name();
    if (out && n) {
        strncpy(out, s.c_str(), n - 1);
        out[n - 1] = 0;
    }
    return s.size();
}
size_t sdfvh_sdf_n_children(void* h) { return S(h).children().size(); }
void* sdfvh_sdf_child(void* h, size_t i) {
    auto ch = S(h).children();
    return i < ch.size() ? new std::shared_ptr<SDFSurface>(ch[i]) : nullptr;
}
void sdfvh_sdf_bounding_box(void* h, float out[6]) {
    auto bb = S(h).bounding_box();
    memcpy(out, &bb, 24);
}
int sdfvh_sdf_device_params(void* h, sdfv_demo_params* p, uint32_t* sdf_id) {
    auto d = S(h).device_sdf();
    if (!d) return -1;
    *p = d->params;
    *sdf_id = d->sdf_id;
    return 0;
}
void sdfvh_sdf_sample(void* h, const float p[3], int distance_only, float out[7]) {
    SDFSample s = S(h).sample(Vec3{p[0], p[1], p[2]}, distance_only != 0);
    memcpy(out, &s, 28);
}
void sdfvh_sdf_sample_batch(void* h, const float* p, size_t n, int distance_only, float* out) {
    S(h).sample_batch(reinterpret_cast<const Vec3*>(p), n, distance_only != 0, reinterpret_cast<SDFSample*>(out));
}
void sdfvh_sdf_normal(void* h, const float p[3], float eps, float out[3]) {
    Vec3 n = S(h).normal(Vec3{p[0], p[1], p[2]}, eps > 0 ? std::optional<float>(eps) : std::nullopt);
    memcpy(out, &n, 12);
}
void sdfvh_sdf_normal_default(void* h, const float p[3], float eps, float out[3]) {
    Vec3 n = S(h).SDFSurface::normal(Vec3{p[0], p[1], p[2]}, eps > 0 ? std::optional<float>(eps) : std::nullopt);
    memcpy(out, &n, 12);
}
// kind: 0 bool, 1 int, 2 float, 3 string.  Returns 0 = Ok, 1 = Err (message copied to err).
int sdfvh_sdf_set_parameter(void* h, uint32_t param_id, int kind, int ival, float fval, const char* sval, char* err,
                            size_t err_len) {
    SDFParamValue v;
    if (kind == 0) v = (bool)(ival != 0);
    else if (kind == 1) v = (int32_t)ival;
    else if (kind == 2) v = fval;
    else v = std::string(sval ? sval : "");
    auto r = S(h).set_parameter(param_id, v);
    if (!r.ok && err && err_len) {
        strncpy(err, r.error.c_str(), err_len - 1);
        err[err_len - 1] = 0;
    }
    return r.ok ? 0 : 1;
}
int sdfvh_sdf_changed(void* h, float out[6]) {
    auto b = S(h).changed();
    if (!b) return 0;
    memcpy(out, &*b, 24);
    return 1;
}
// parameters as a text block: one line per parameter "id|name|kind|value|description"
size_t sdfvh_sdf_parameters(void* h, char* out, size_t n) {
    std::string s;
    for (auto& p : S(h).parameters()) {
        s += std::to_string(p.id) + "|" + p.name + "|" + std::to_string((uint32_t)p.kind.tag) + "|" +
             param_value_debug(p.value) + "|" + p.description + "\n";
    }
    if (out && n) {
        strncpy(out, s.c_str(), n - 1);
        out[n - 1] = 0;
    }
    return s.size();
}

// ---- SDFViewer ----
void* sdfvh_viewer_from_bb(const float bb[6], size_t max_voxels_side, size_t loading_passes) {
    BoundingBox b{Vec3{bb[0], bb[1], bb[2]}, Vec3{bb[3], bb[4], bb[5]}};
    return SDFViewer::from_bb(b, max_voxels_side, loading_passes).release();
}
void* sdfvh_viewer_new_voxels(size_t w, size_t h, size_t d, const float bb[6], size_t loading_passes) {
    BoundingBox b{Vec3{bb[0], bb[1], bb[2]}, Vec3{bb[3], bb[4], bb[5]}};
    return SDFViewer::new_voxels({w, h, d}, b, loading_passes).release();
}
// layout: 0 auto, 1 texture order, 2 y-interleaved
void* sdfvh_viewer_new_voxels_layout(size_t w, size_t h, size_t d, const float bb[6], size_t loading_passes, int layout) {
    BoundingBox b{Vec3{bb[0], bb[1], bb[2]}, Vec3{bb[3], bb[4], bb[5]}};
    return SDFViewer::new_voxels({w, h, d}, b, loading_passes, (SDFViewer::VolumeLayout)layout).release();
}
void sdfvh_viewer_free(void* v) { delete static_cast<SDFViewer*>(v); }
static SDFViewer& V(void* v) { return *static_cast<SDFViewer*>(v); }
void sdfvh_viewer_dims(void* v, uint32_t out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = V(v).material.tex_size[i];
}
size_t sdfvh_viewer_update(void* v, void* sdf, double max_delta_seconds) {
    return V(v).update(S(sdf), std::chrono::nanoseconds((long long)(max_delta_seconds * 1e9)));
}
// ingest path knobs (0 = keep): host threads, records per transfer buffer
void sdfvh_viewer_set_ingest(void* v, unsigned host_threads, size_t capacity) {
    V(v).host_threads = host_threads;
    if (capacity) V(v).ingest_capacity = capacity;
}
size_t sdfvh_viewer_last_error(void* v, char* out, size_t n) {
    const std::string s = V(v).last_error();
    if (out && n) {
        strncpy(out, s.c_str(), n - 1);
        out[n - 1] = 0;
    }
    return s.size();
}
unsigned sdfvh_sdf_sample_concurrency(void* h) { return S(h).sample_concurrency(); }
void sdfvh_viewer_commit(void* v) { V(v).commit(); }
float sdfvh_viewer_lod(void* v) { return V(v).material.lod_dist_between_samples; }
size_t sdfvh_viewer_remaining(void* v) { return V(v).loading_mgr.len(); }
size_t sdfvh_viewer_passes_left(void* v) { return V(v).loading_mgr.passes_left(); }
int sdfvh_viewer_has_changed_box(void* v) { return V(v).changed_box ? 1 : 0; }
int sdfvh_viewer_download(void* v, float* tex0, float* tex1) { return V(v).download(tex0, tex1); }
void* sdfvh_viewer_tex0(void* v) { return V(v).tex0_device(); }
void* sdfvh_viewer_tex1(void* v) { return V(v).tex1_device(); }
// SDFViewerMaterial::render with the scene's default camera (scene/mod.rs:82-95) at width x height
int sdfvh_viewer_render(void* v, uint32_t width, uint32_t height, const float eye[3], float* rgba_host) {
    Camera cam;
    if (eye) cam.position = Vec3{eye[0], eye[1], eye[2]};
    cam.set_viewport(width, height);
    DeviceBuffer out((size_t)width * height * 16);
    if (!out.ok()) return -1;
    int rc = V(v).material.render(cam, out.f32(), nullptr, V(v).stream);
    if (rc != 0) return rc;
    // the frame was enqueued on the viewer's stream, which may be a non-blocking one: wait for it before the copy
    if (hipStreamSynchronize((hipStream_t)V(v).stream) != hipSuccess) return -1;
    return hipMemcpy(rgba_host, out.get(), out.bytes(), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

// Same frame, left in a caller-owned DEVICE image (width x height x 4 floats); enqueue only, no synchronisation.
int sdfvh_viewer_render_device(void* v, uint32_t width, uint32_t height, const float eye[3], float* rgba_device) {
    Camera cam;
    if (eye) cam.position = Vec3{eye[0], eye[1], eye[2]};
    cam.set_viewport(width, height);
    return V(v).material.render(cam, rgba_device, nullptr, V(v).stream);
}
// 0: the frames march over the distance volume; 1: over the pair volume commit() built; 2: over the interleaved volume
int sdfvh_viewer_pairs_valid(void* v) {
    if (V(v).material.dist && V(v).material.dist_interleaved) return V(v).material.lod_dist_between_samples == 1.0f ? 2 : 0;
    return V(v).material.pairs && V(v).material.pairs_valid ? (V(v).material.pairs_interleaved ? 2 : 1) : 0;
}
int sdfvh_viewer_sync(void* v) { return hipStreamSynchronize((hipStream_t)V(v).stream) == hipSuccess ? 0 : -1; }

// ---- SDFViewerAppScene (a manual clock, in milliseconds, makes the 500 ms commit spacing testable) ----
struct SceneHandle {
    long long now_ms = 0;
    std::unique_ptr<SDFViewerAppScene> scene;
};
void* sdfvh_scene_new(void* sdf) {
    auto* h = new SceneHandle();
    auto clock = [h] { return std::chrono::steady_clock::time_point(std::chrono::milliseconds(h->now_ms)); };
    h->scene.reset(new SDFViewerAppScene(*static_cast<std::shared_ptr<SDFSurface>*>(sdf), clock));
    if (!h->scene->sdf_viewer) {
        delete h;
        return nullptr;
    }
    return h;
}
void sdfvh_scene_free(void* h) { delete static_cast<SceneHandle*>(h); }
void sdfvh_scene_advance_clock(void* h, long long ms) { static_cast<SceneHandle*>(h)->now_ms += ms; }
int sdfvh_scene_set_sdf(void* h, void* sdf, long long max_voxels_side, long long loading_passes) {
    auto& sc = *static_cast<SceneHandle*>(h)->scene;
    return sc.set_sdf(*static_cast<std::shared_ptr<SDFSurface>*>(sdf),
                      max_voxels_side >= 0 ? std::optional<size_t>((size_t)max_voxels_side) : std::nullopt,
                      loading_passes >= 0 ? std::optional<size_t>((size_t)loading_passes) : std::nullopt) ? 0 : -1;
}
void sdfvh_scene_set_budget_ms(void* h, long long budget_ms) {
    static_cast<SceneHandle*>(h)->scene->load_budget = std::chrono::milliseconds(budget_ms);
}
// out[4] = {cpu_updates, committed, last_chunk, request_repaint}; rgba_host may be NULL (no drawing)
int sdfvh_scene_render(void* h, uint32_t width, uint32_t height, float* rgba_host, unsigned long long out[4]) {
    auto& sc = *static_cast<SceneHandle*>(h)->scene;
    DeviceBuffer img(rgba_host ? (size_t)width * height * 16 : 0);
    if (!img.ok()) return -1;
    RenderReport r = sc.render(width, height, rgba_host ? img.f32() : nullptr);
    out[0] = r.cpu_updates; out[1] = r.committed; out[2] = r.last_chunk; out[3] = r.request_repaint;
    if (rgba_host && hipStreamSynchronize((hipStream_t)sc.sdf_viewer->stream) != hipSuccess) return -1;
    if (rgba_host && hipMemcpy(rgba_host, img.get(), img.bytes(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}
float sdfvh_scene_lod(void* h) { return static_cast<SceneHandle*>(h)->scene->sdf_viewer->material.lod_dist_between_samples; }
void sdfvh_scene_dims(void* h, uint32_t out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = static_cast<SceneHandle*>(h)->scene->sdf_viewer->material.tex_size[i];
}
// returns -1 when not loading, else writes the progress text and returns progress * 1e6
long long sdfvh_scene_load_progress(void* h, char* text, size_t n) {
    auto p = static_cast<SceneHandle*>(h)->scene->load_progress();
    if (!p) return -1;
    if (text && n) {
        strncpy(text, p->second.c_str(), n - 1);
        text[n - 1] = 0;
    }
    return (long long)(p->first * 1e6f);
}

// ---- Mesh (meshers/mesh.rs) ----
size_t sdfvh_format_f32(float v, char* out, size_t n) {
    const std::string s = format_f32(v);
    if (out && n) {
        strncpy(out, s.c_str(), n - 1);
        out[n - 1] = 0;
    }
    return s.size();
}
uint32_t sdfvh_ply_color_u8(float c) { return ply_color_u8(c); }
// Mesher::mesh + (optionally) Mesh::postproc of an SDF handle; returns a Mesh handle or NULL (err filled).
void* sdfvh_mesh_sdf(void* sdf, const char* mesher, size_t max_voxels_per_axis, int postproc, char* err, size_t err_len) {
    auto fail = [&](const std::string& m) -> void* {
        if (err && err_len) {
            strncpy(err, m.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
        return nullptr;
    };
    auto which = mesher_from_name(mesher ? mesher : "");
    if (!which) return fail(std::string("unknown mesher '") + (mesher ? mesher : "") + "'");
    std::string e;
    MesherConfig cfg;
    cfg.max_voxels_per_axis = max_voxels_per_axis;
    auto m = mesh_sdf(*which, S(sdf), cfg, &e);
    if (!m) return fail(e);
    if (postproc && m->postproc(S(sdf)) != 0) return fail(sdfv_last_error());
    return new Mesh(std::move(*m));
}
// Builds a Mesh from caller arrays (serialisation tests need no GPU).
void* sdfvh_mesh_from_arrays(const float* vertices12, size_t n_vertices, const uint32_t* indices, size_t n_indices) {
    auto* m = new Mesh;
    m->vertices.resize(n_vertices);
    if (n_vertices) memcpy(static_cast<void*>(m->vertices.data()), vertices12, n_vertices * sizeof(Vertex));
    m->indices.assign(indices, indices + n_indices);
    return m;
}
void sdfvh_mesh_free(void* m) { delete static_cast<Mesh*>(m); }
size_t sdfvh_mesh_counts(void* m, size_t* n_indices) {
    *n_indices = static_cast<Mesh*>(m)->indices.size();
    return static_cast<Mesh*>(m)->vertices.size();
}
void sdfvh_mesh_copy(void* m, float* vertices12, uint32_t* indices) {
    auto& mesh = *static_cast<Mesh*>(m);
    if (!mesh.vertices.empty()) memcpy(vertices12, mesh.vertices.data(), mesh.vertices.size() * sizeof(Vertex));
    if (!mesh.indices.empty()) memcpy(indices, mesh.indices.data(), mesh.indices.size() * 4);
}
// Mesh::serialize_ply into out (capacity n); returns the bytes the PLY needs.
size_t sdfvh_mesh_serialize_ply(void* m, const char* version_info, char* out, size_t n) {
    std::ostringstream os;
    const size_t bytes = static_cast<Mesh*>(m)->serialize_ply(os, version_info ? version_info : "");
    const std::string s = os.str();
    if (out && n) memcpy(out, s.data(), s.size() < n ? s.size() : n);
    return bytes;
}

}  // extern "C"
#pragma GCC visibility pop
