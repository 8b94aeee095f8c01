// sdf_viewer.cpp -- see sdf_viewer.hpp.
#include "sdf_viewer.hpp"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>

namespace sdfviewer {

// ---- DeviceBuffer ----
DeviceBuffer::DeviceBuffer(size_t bytes) : bytes_(bytes) {
    if (bytes && hipMalloc(&ptr_, bytes) != hipSuccess) {
        (void)hipGetLastError();
        ptr_ = nullptr;
    }
}
DeviceBuffer::~DeviceBuffer() {
    if (owned_ && ptr_) (void)hipFree(ptr_);
}
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
        if (owned_ && ptr_) (void)hipFree(ptr_);
        ptr_ = o.ptr_;
        bytes_ = o.bytes_;
        owned_ = o.owned_;
        o.ptr_ = nullptr;
        o.bytes_ = 0;
    }
    return *this;
}

// ---- Camera ----
Camera Camera::new_perspective(uint32_t width, uint32_t height, Vec3 position, Vec3 target, Vec3 up,
                               float fovy_degrees, float z_near, float z_far) {
    Camera c;
    c.viewport_width = width;
    c.viewport_height = height;
    c.position = position;
    c.target = target;
    c.up = up;
    c.fovy_degrees = fovy_degrees;
    c.z_near = z_near;
    c.z_far = z_far;
    return c;
}

sdfv_camera Camera::to_device() const {
    sdfv_camera cam;
    const float aspect = viewport_height ? (float)viewport_width / (float)viewport_height : 1.0f;
    sdfv_camera_look_at(&cam, &position.x, &target.x, &up.x, fovy_degrees, aspect, z_near, z_far);
    return cam;
}

// ---- SDFViewerMaterial ----
sdfv_render_params SDFViewerMaterial::uniforms() const {
    sdfv_render_params rp;
    sdfv_render_params_default(&rp, nullptr);
    rp.bounds_min[0] = voxels_bounds[0].x; rp.bounds_min[1] = voxels_bounds[0].y; rp.bounds_min[2] = voxels_bounds[0].z;
    rp.bounds_max[0] = voxels_bounds[1].x; rp.bounds_max[1] = voxels_bounds[1].y; rp.bounds_max[2] = voxels_bounds[1].z;
    for (int i = 0; i < 3; ++i) rp.tex_size[i] = tex_size[i];
    rp.lod_dist_between_samples = lod_dist_between_samples;
    for (int i = 0; i < 4; ++i) rp.tint[i] = color[i];
    rp.gamma = gamma;
    return rp;
}

int SDFViewerMaterial::materialize(void* stream) const {
    if (!undefined_rows) return 0;
    sdfv_grid g{};
    for (int i = 0; i < 3; ++i) g.dims[i] = tex_size[i];
    g.z_end = g.dims[2];  // (the bounding box plays no part in an initialisation)
    const int rc = sdfv_grid_init_unvisited_ex(&g, defined_step, tex0->f32(), tex1->f32(), dist ? dist->f32() : nullptr,
                                               dist && dist_interleaved ? SDFV_PASS_VOLUME_INTERLEAVED : 0u, stream);
    if (rc == 0) undefined_rows = false;
    return rc;
}

int SDFViewerMaterial::render(const Camera& camera, float* rgba_device, sdfv_march_aux* aux_device, void* stream) const {
    // a frame in the middle of a virgin load reads voxels no pass has written (NEAREST snaps onto the lattice, but
    // MirroredRepeat folds coordinate 1.0 onto voxel N - 1): they must hold the AIR the reference's textures hold
    if (int rc = materialize(stream)) return rc;
    const sdfv_render_params rp = uniforms();
    const sdfv_camera cam = camera.to_device();
    // the distance volume is only meaningful for the fully loaded grid (LINEAR filter, lod == 1)
    const float* d = (dist && lod_dist_between_samples == 1.0f) ? dist->f32() : nullptr;
    if (d && dist_interleaved)  // the volume the fill wrote IS the march's y-interleaved volume
        return sdfv_raymarch_volumes(&rp, tex0->f32(), tex1->f32(), nullptr, nullptr, d, &cam, 1, camera.viewport_width,
                                     camera.viewport_height, 0, camera.viewport_height, rgba_device, nullptr, aux_device, stream);
    // a viewer renders many frames per load: the pair volume commit() built halves the march's gathers (same bits)
    const float* p = (d && pairs && pairs_valid) ? pairs->f32() : nullptr;
    return sdfv_raymarch_volumes(&rp, tex0->f32(), tex1->f32(), d, pairs_interleaved ? nullptr : p, pairs_interleaved ? p : nullptr,
                                 &cam, 1, camera.viewport_width, camera.viewport_height, 0, camera.viewport_height, rgba_device,
                                 nullptr, aux_device, stream);
}

// ---- SDFViewer ----
namespace {
// The distance of tex1 from tex0's end (see the constructor).
constexpr size_t kPlacementSlack = 64u << 10;
size_t default_skew(size_t texture_bytes) {
    return texture_bytes == ((size_t)1 << 28) ? 12288 : (texture_bytes == ((size_t)1 << 30) ? 20480 : 0);
}
}  // namespace

SDFViewer::SDFViewer(std::array<size_t, 3> voxels, const BoundingBox& bb, size_t passes)
    : loading_mgr(voxels, passes), bounding_box(bb) {
    const size_t bytes = voxels[0] * voxels[1] * voxels[2] * 16;
    // Both textures in one block (allocation only, nothing is launched or waited for here).  If the block cannot be had, two
    // plain allocations do.  The distance between them: what MI355X boxes have shown reproducibly since round 1 for the
    // texture sizes that matter (the fill's rate is periodic in the distance between the textures, EXPERIMENTS R4.1,
    // profiles/r04_place_width.json): textures of 256 MiB (256^3 and every other shape of that size) run 6-8 % faster with
    // tex1 12 KiB after tex0's end, textures of 1 GiB with 20 KiB, 4 GiB (512^3) with none.  Anything else: none.  (Rounds 3-5
    // also shipped a run-time probe, SDFViewer::tune(): in two driver runs it cost 100 ms and never beat these constants.)
    const size_t skew = default_skew(bytes);
    block_ = std::make_shared<DeviceBuffer>(2 * bytes + kPlacementSlack);
    if (bytes > 0 && block_->ok()) {
        material.tex0 = std::make_shared<DeviceBuffer>(static_cast<char*>(block_->get()), bytes);
        material.tex1 = std::make_shared<DeviceBuffer>(static_cast<char*>(block_->get()) + bytes + skew, bytes);
    } else {
        block_.reset();
        material.tex0 = std::make_shared<DeviceBuffer>(bytes);
        material.tex1 = std::make_shared<DeviceBuffer>(bytes);
    }
    material.tex_size = {(uint32_t)voxels[0], (uint32_t)voxels[1], (uint32_t)voxels[2]};
    material.voxels_bounds = bb;
}

std::unique_ptr<SDFViewer> SDFViewer::from_bb(const BoundingBox& bb, size_t max_voxels_side, size_t loading_passes) {
    sdfv_grid g;
    if (sdfv_grid_from_bb(&bb[0].x, &bb[1].x, (uint32_t)max_voxels_side, &g) != 0) return nullptr;
    fprintf(stderr, "Using %ux%ux%u voxels (dimensions: %gx%gx%g)\n", g.dims[0], g.dims[1], g.dims[2],
            (double)(bb[1].x - bb[0].x), (double)(bb[1].y - bb[0].y), (double)(bb[1].z - bb[0].z));  // :69-70
    return new_voxels({g.dims[0], g.dims[1], g.dims[2]}, bb, loading_passes);
}

std::unique_ptr<SDFViewer> SDFViewer::new_voxels(std::array<size_t, 3> voxels, const BoundingBox& bb,
                                                 size_t loading_passes, VolumeLayout layout) {
    std::unique_ptr<SDFViewer> v(new SDFViewer(voxels, bb, loading_passes));
    if (!v->material.tex0->ok() || !v->material.tex1->ok()) return nullptr;
    // The reference fills both textures with [AIR_DIST; 4] here (:76-77).  This grid is VIRGIN instead: that state is
    // recorded, not written -- the first update() overwrites every byte of it anyway (36 B/voxel the load never pays for).
    v->material.undefined_rows = true;
    v->material.defined_step = 0;
    // The compact distance volume (tex0.r, 4 B/voxel) lives next to the textures from the start and every fill keeps
    // it in sync: passes read it for update_required instead of tex0's 16-byte texels, commit() has nothing to derive.
    v->material.dist = std::make_shared<DeviceBuffer>(v->material.tex0->bytes() / 4);
    v->dist_synced_ = v->material.dist->ok();
    if (!v->dist_synced_) v->material.dist.reset();  // out of memory: march tex0.r in place, passes read tex0
    // Beyond the last-level cache the march gathers fastest from the y-interleaved volume (sdfv_march_volume_advice): the
    // fills and passes of THIS viewer write the volume in that layout from the start, so a load ends with the march's volume
    // in place and commit() builds nothing (512^3: 0.70 + 0.20 ms -> 0.70).  Smaller cubic grids keep the plain volume and
    // let commit() derive the pair volume from it.
    if (v->dist_synced_ && layout == VolumeLayout::Auto) {
        const sdfv_grid g = v->grid();
        uint32_t kind = SDFV_MARCH_VOLUME_NONE;
        if (sdfv_march_volume_advice(&g, &kind) == 0 && kind == SDFV_MARCH_VOLUME_INTERLEAVED) v->material.dist_interleaved = true;
    } else if (v->dist_synced_ && layout == VolumeLayout::Interleaved) {
        if (voxels[1] & 1) return nullptr;  // rows are paired
        v->material.dist_interleaved = true;
    }
    return v;
}

sdfv_grid SDFViewer::grid() const {
    sdfv_grid g;
    for (int i = 0; i < 3; ++i) g.dims[i] = material.tex_size[i];
    g.bb_min[0] = bounding_box[0].x; g.bb_min[1] = bounding_box[0].y; g.bb_min[2] = bounding_box[0].z;
    g.bb_max[0] = bounding_box[1].x; g.bb_max[1] = bounding_box[1].y; g.bb_max[2] = bounding_box[1].z;
    g.z_begin = 0;
    g.z_end = g.dims[2];
    return g;
}

size_t SDFViewer::update(SDFSurface& sdf, std::chrono::nanoseconds max_delta_time) {
    error_.clear();  // last_error() describes THIS call
    // Check whether the SDF self-reports updates.  (:130-141)
    bool just_changed_box = false;
    // the passes of ONE load share one SDF and one set of parameters: a different device SDF (or parameter block) than the
    // one the load began with, or any reported change, ends it
    if (const auto dev_now = sdf.device_sdf()) {
        if (!load_sdf_ || memcmp(&*load_sdf_, &*dev_now, sizeof(*dev_now)) != 0) {
            if (!fresh_) same_load_ = false;
            load_sdf_ = *dev_now;
        }
    }
    if (auto new_box = sdf.changed()) {
        same_load_ = false;
        changed_box = changed_box ? merge_bounding_boxes(*changed_box, *new_box) : *new_box;
        changed_box_while_loading = loading_mgr.len() > 0 || changed_box_while_loading;
        just_changed_box = true;
    }
    // Another full (3-pass) manager while changes are pending and the manager is idle.  (:146-156)
    if (changed_box) {
        if (loading_mgr.len() == 0) {
            loading_mgr = LoadingManager(loading_mgr.limits, 3);
            if (!just_changed_box) {
                if (!changed_box_while_loading) changed_box.reset();
                changed_box_while_loading = false;
            }
        }
    }

    const size_t start_iter = loading_mgr.total_iterations();
    const auto dev = sdf.device_sdf();
    if (!dev) return update_host(sdf, max_delta_time);  // any `impl SDFSurface`: sampled on the host, packed on the device
    host_mirror_valid_ = false;  // (whatever runs below rewrites tex0.r on the device)
    const sdfv_grid g = grid();
    const auto start_time = std::chrono::steady_clock::now();
    // Fresh grid, nothing pending, and a budget that lets every pass be enqueued in this call anyway (a pass is
    // one asynchronous launch): the state all passes converge to is the dense fill, which moves 32 B/voxel once
    // instead of re-reading and partially rewriting the grid once per pass.  No intermediate state is observable
    // inside one update() call; the LoadingManager is advanced exactly as the passes would have advanced it.
    // The same holds for a re-sampling whose changed box contains every voxel of the grid (the demo SDF reports its
    // whole bounding box on any parameter edit, demo/mod.rs:135-144): inside the box update_required is true for
    // every visited voxel, so the passes' common final state is again the dense fill.  "Contains every voxel" is
    // decided on the voxels' own coordinates (first and last index per axis, same arithmetic as the kernels).
    auto box_covers_grid = [&]() {
        if (!changed_box) return false;
        const float lo[3] = {(*changed_box)[0].x, (*changed_box)[0].y, (*changed_box)[0].z};
        const float hi[3] = {(*changed_box)[1].x, (*changed_box)[1].y, (*changed_box)[1].z};
        for (int i = 0; i < 3; ++i) {
            const float dm1 = (float)g.dims[i] - 1.0f, size = g.bb_max[i] - g.bb_min[i];
            float first = 0.0f / dm1;  // scene/sdf/mod.rs:179-182, three separately rounded steps
            first = first * size;
            first = first + g.bb_min[i];
            float last = ((float)g.dims[i] - 1.0f) / dm1;
            last = last * size;
            last = last + g.bb_min[i];
            if (!(first >= lo[i] && first <= hi[i] && last >= lo[i] && last <= hi[i])) return false;  // NaN: no
        }
        return true;
    };
    const bool all_passes_fit = loading_mgr.total_iterations() == 0 && loading_mgr.step_size() != 0 &&
                                max_delta_time >= std::chrono::milliseconds(1);
    if (all_passes_fit && ((fresh_ && !changed_box) || box_covers_grid())) {
        // The fill writes the distance volume in the same pass (+4 B/voxel instead of a second pass over tex0).
        float* dist_out = dist_synced_ ? material.dist->f32() : nullptr;
        int rc;
        if (dist_out && material.dist_interleaved) {
            // the same dense launch through the pass entry point, which carries the volume's layout: a step-1 pass in which
            // update_required holds everywhere (a virgin / fresh grid, or a box that covers it) IS the dense fused fill
            float box[6];
            const float* box_ptr = nullptr;
            uint32_t flags = SDFV_PASS_VOLUME_INTERLEAVED;
            if (changed_box) {
                box[0] = (*changed_box)[0].x; box[1] = (*changed_box)[0].y; box[2] = (*changed_box)[0].z;
                box[3] = (*changed_box)[1].x; box[4] = (*changed_box)[1].y; box[5] = (*changed_box)[1].z;
                box_ptr = box;
                if (material.materialize(stream) != 0) {  // (a box test is no virgin pass; the dense launch overwrites it all anyway)
                    error_ = sdfv_last_error();
                    return 0;
                }
            } else {
                flags |= (material.undefined_rows ? SDFV_PASS_VIRGIN_GRID : SDFV_PASS_FRESH_GRID) | SDFV_PASS_SAME_LOAD;
            }
            rc = sdfv_fill_grid_pass_ex(&dev->params, dev->sdf_id, &g, 1u, box_ptr, tex0_device(), tex1_device(), dist_out, flags, stream);
        } else {
            rc = sdfv_fill_grid_commit(&dev->params, dev->sdf_id, &g, tex0_device(), tex1_device(), dist_out, stream);
        }
        if (rc != 0) {
            error_ = sdfv_last_error();
            return 0;
        }
        fresh_ = false;
        material.undefined_rows = false;  // the dense fill wrote every voxel
        material.pairs_valid = false;
        while (loading_mgr.step_size() != 0) loading_mgr.finish_pass();
        loaded_once_ = true;
        publish_lod();
        return loading_mgr.total_iterations() - start_iter;
    }
    bool first = true;
    // "while first || start_time.elapsed() < max_delta_time" with a pass as the unit of work  (:173)
    while (first || std::chrono::steady_clock::now() - start_time < max_delta_time) {
        first = false;
        const size_t step = loading_mgr.step_size();
        if (step == 0) break;  // No more work to do!
        float box[6];
        const float* box_ptr = nullptr;
        if (changed_box) {
            box[0] = (*changed_box)[0].x; box[1] = (*changed_box)[0].y; box[2] = (*changed_box)[0].z;
            box[3] = (*changed_box)[1].x; box[4] = (*changed_box)[1].y; box[5] = (*changed_box)[1].z;
            box_ptr = box;
        }
        // What this LoadingManager knows about the grid (sdfv_fill_grid_pass_ex): the first pass of a load over the grid
        // new_voxels initialised sees AIR_DIST everywhere; the later passes of that load revisit only what ITS earlier
        // passes wrote (same SDF, same parameters -- a changed box or another SDF ends the load, see set below).
        // A virgin grid (nothing written by new_voxels): the same passes, told so -- they write the rows they visit whole
        // and read nothing.  A pass that must READ the grid (a changed box, another SDF mid-load) first gets the initial
        // state written into the rows no pass has reached.
        uint32_t flags = 0;
        if (same_load_ && !changed_box)
            flags = material.undefined_rows ? (SDFV_PASS_VIRGIN_GRID | SDFV_PASS_SAME_LOAD)
                                            : ((fresh_ ? SDFV_PASS_FRESH_GRID : 0u) | SDFV_PASS_SAME_LOAD);
        else if (material.materialize(stream) != 0) {
            error_ = sdfv_last_error();
            break;
        }
        if (dist_synced_ && material.dist_interleaved) flags |= SDFV_PASS_VOLUME_INTERLEAVED;
        // the manager the reference runs once a changed box has been worked off (:146-156, no box any more) scans a loaded grid
        // and finds nothing: say so (a hint: the scan streams the volume past the caches)
        if (loaded_once_ && !changed_box && flags == (flags & SDFV_PASS_VOLUME_INTERLEAVED)) flags |= SDFV_PASS_EXPECT_NOOP;
        if (sdfv_fill_grid_pass_ex(&dev->params, dev->sdf_id, &g, (uint32_t)step, box_ptr, tex0_device(), tex1_device(),
                                   dist_synced_ ? material.dist->f32() : nullptr, flags, stream) != 0) {
            error_ = sdfv_last_error();
            break;
        }
        fresh_ = false;
        if (material.undefined_rows) {  // (only virgin passes get here with the flag still set)
            material.defined_step = (uint32_t)step;
            if (step == 1) material.undefined_rows = false;
        }
        material.pairs_valid = false;
        loading_mgr.finish_pass();
        if (loading_mgr.step_size() == 0) loaded_once_ = true;
        publish_lod();
    }
    return loading_mgr.total_iterations() - start_iter;
}

// The reference's textures change only in commit(), together with the LOD uniform (scene/sdf/mod.rs:220-239).  Here a
// pass rewrites the device textures the moment it runs, so the uniform that tells the shader how to read them (NEAREST
// snap to the coarse lattice while loading, LINEAR + distance volume when loaded) must move with the data, not with the
// scene's 500 ms commit throttle: a frame rendered between two commits would otherwise sample a half-rewritten grid
// with the filter of the previous state.
void SDFViewer::publish_lod() {
    material.lod_dist_between_samples = std::pow(2.0f, (float)(uint8_t)loading_mgr.passes_left());  // :226
}

void SDFViewer::commit() {
    // tex0.fill / tex1.fill re-upload nothing here: the textures already live on the device.  (:222-234)
    publish_lod();  // idempotent: update() has already published it with the data
    // lod == 1 switches the GL filter to LINEAR (:227-230): the kernel selects the filter from the same uniform.
    // Where the reference uploads both textures there is nothing to move, and nothing to derive either: the compact
    // distance volume the LINEAR march reads has been kept in sync by every fill.  Without one (allocation failed at
    // creation) the march reads tex0.r in place.
    // What a commit of the fully loaded grid does derive is the pair volume: the frames that follow (the reference renders
    // one per repaint) march over it.  12 B/voxel of traffic -- against the reference's re-upload of 32 B/voxel over PCIe.
    if (dist_synced_ && material.dist_interleaved) return;  // the fills wrote the march's volume themselves
    if (dist_synced_ && loading_mgr.step_size() == 0 && !material.pairs_valid) {
        const sdfv_grid g = grid();
        if (!material.pairs && !material.no_march_volume) {
            // beyond the last-level cache the interleaved volume (4 B/voxel) marches faster than the pair volume (8)
            uint32_t kind = SDFV_MARCH_VOLUME_PAIRS;
            (void)sdfv_march_volume_advice(&g, &kind);
            material.pairs_interleaved = kind == SDFV_MARCH_VOLUME_INTERLEAVED;
            material.no_march_volume = kind == SDFV_MARCH_VOLUME_NONE;  // (a grid that is not cubic: the distance volume it is)
            if (!material.no_march_volume)
                material.pairs = std::make_shared<DeviceBuffer>(material.dist->bytes() * (material.pairs_interleaved ? 1 : 2));
        }
        if (!material.pairs) return;
        const int rc = !material.pairs->ok() ? -1
                       : material.pairs_interleaved ? sdfv_commit_interleaved(&g, material.dist->f32(), material.pairs->f32(), stream)
                                                    : sdfv_commit_pairs(&g, material.dist->f32(), material.pairs->f32(), stream);
        if (rc == 0)
            material.pairs_valid = true;
        else {
            material.pairs.reset();  // out of memory: the march keeps reading the distance volume ...
            material.no_march_volume = true;  // ... and later commits do not retry the allocation
        }
    }
}

int SDFViewer::download(float* tex0_host, float* tex1_host) const {
    if (material.materialize(stream) != 0) return -1;  // a virgin grid shows new_voxels' [AIR_DIST; 4] like any other
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    if (hipMemcpy(tex0_host, tex0_device(), material.tex0->bytes(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(tex1_host, tex1_device(), material.tex1->bytes(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

}  // namespace sdfviewer
