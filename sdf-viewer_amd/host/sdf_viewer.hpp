// sdf_viewer.hpp -- C++ mirror of the reference's viewer controller and raymarch material (L2):
//   SDFViewer          src/app/scene/sdf/mod.rs:20-251   {from_bb, new_voxels, update, commit}
//   SDFViewerMaterial  src/app/scene/sdf/material.rs:8-86
//   Camera             three-d Camera::new_perspective as used by src/app/scene/mod.rs:82-95
// with the two textures living in HBM instead of a CPU Vec + GL texture pair:
//   - update() launches LoadingManager passes as kernels (sdfv_fill_grid_pass) -- or the dense kernel when a
//     fresh grid can be finished within the call -- instead of calling sample() once per voxel;
//   - every pass that update() enqueues rewrites the device textures at once, so update() also publishes
//     lod_dist_between_samples = 2^passes_left (the uniform that tells the shader how to read them) with the data;
//     commit() has nothing to upload (the reference re-uploads both whole textures and sets the uniform there,
//     :220-239) and nothing to derive: the compact distance volume is kept in sync by every fill and pass;
//   - SDFViewerMaterial::render() is the fragment shader over every pixel (sdfv_raymarch);
//   - new_voxels() allocates and writes NOTHING ("virgin" grid): the reference's initial state [AIR_DIST; 4] is only recorded.
//     A load that runs all its passes never pays for it (the dense fill, or the step-1 pass, writes every byte); the passes
//     before it write the rows they visit whole (SDFV_PASS_VIRGIN_GRID) and whatever READS the whole grid in between -- a
//     frame at an intermediate LOD, download(), a pass with a changed box -- first writes AIR into the rows no pass reached.
#pragma once

#include <chrono>
#include <memory>
#include <optional>

#include "loading_manager.hpp"
#include "sdf_surface.hpp"

namespace sdfviewer {

// three-d Camera (perspective) + the defaults of SDFViewerAppScene::new, scene/mod.rs:82-95
struct Camera {
    Vec3 position{2.5f, 3.0f, 5.0f};
    Vec3 target{0.0f, 0.0f, 0.0f};
    Vec3 up{0.0f, 1.0f, 0.0f};
    float fovy_degrees = 45.0f;
    float z_near = 0.1f, z_far = 1000.0f;
    uint32_t viewport_width = 0, viewport_height = 0;  // "Updated at runtime"
    static Camera new_perspective(uint32_t width, uint32_t height, Vec3 position, Vec3 target, Vec3 up,
                                  float fovy_degrees, float z_near, float z_far);
    void set_viewport(uint32_t width, uint32_t height) {
        viewport_width = width;
        viewport_height = height;
    }
    sdfv_camera to_device() const;
};

// Device memory owner (hipMalloc/hipFree) or a view over caller-owned device memory.
class DeviceBuffer {
   public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes);
    DeviceBuffer(void* external, size_t bytes) : ptr_(external), bytes_(bytes), owned_(false) {}
    ~DeviceBuffer();
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept { *this = std::move(o); }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept;
    float* f32() const { return static_cast<float*>(ptr_); }
    void* get() const { return ptr_; }
    size_t bytes() const { return bytes_; }
    bool ok() const { return ptr_ != nullptr || bytes_ == 0; }

   private:
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
    bool owned_ = true;
};

// material.rs:8-31
struct SDFViewerMaterial {
    std::shared_ptr<DeviceBuffer> tex0;  // distance (R), colour (GBA)
    std::shared_ptr<DeviceBuffer> tex1;  // material properties (RGB)
    std::shared_ptr<DeviceBuffer> dist;  // the distance volume (tex0.r, 4 B/voxel), kept in sync by every fill and pass (may be null)
    bool dist_interleaved = false;       // ... laid out y-interleaved (SDFV_PASS_VOLUME_INTERLEAVED): for grids beyond the last-
                                         // level cache the fill writes the volume the march gathers fastest from ITSELF -- commit()
                                         // then has nothing to build, render() passes it as the descriptor's `ilv`
    std::shared_ptr<DeviceBuffer> pairs;  // y-pair volume (sdfv_commit_pairs) of the LOADED grid, or null / stale
    bool pairs_valid = false;             // pairs mirrors dist: set by SDFViewer::commit, cleared by every fill
    bool no_march_volume = false;         // sdfv_march_volume_advice said neither pays for this grid: commit() builds none
    bool pairs_interleaved = false;       // `pairs` holds the y-interleaved volume instead (sdfv_march_volume_advice)
    // Virgin load: the rows of the textures (and of dist) no pass has written yet hold undefined bytes, logically [AIR_DIST; 4].
    mutable bool undefined_rows = false;
    mutable uint32_t defined_step = 0;    // smallest step of the passes run so far: rows with y and z multiples of it are defined (0: none)
    // Writes new_voxels' initial state into the rows still undefined (sdfv_grid_init_unvisited); a no-op otherwise.
    int materialize(void* stream) const;
    std::array<uint32_t, 3> tex_size{0, 0, 0};
    BoundingBox voxels_bounds;
    float lod_dist_between_samples = 1.0f;
    float color[4] = {1.0f, 1.0f, 1.0f, 1.0f};  // Srgba::WHITE.to_linear_srgb()
    float gamma = 0.0f;                          // env "gamma" (material.rs:39); <= 0: not defined

    sdfv_render_params uniforms() const;  // use_uniforms, material.rs:50-73
    // Draws the volume: one ray per pixel of camera's viewport; rgba_device holds W*H*4 floats.
    int render(const Camera& camera, float* rgba_device, sdfv_march_aux* aux_device, void* stream) const;
};

class SDFViewer {
   public:
    // scene/sdf/mod.rs:46-72
    static std::unique_ptr<SDFViewer> from_bb(const BoundingBox& bb, size_t max_voxels_side, size_t loading_passes);
    // scene/sdf/mod.rs:75-101 (allocates both textures on the device and fills them with AIR_DIST)
    // `layout`: how the 4 B/voxel distance volume next to the textures is laid out.  Auto = what the march gathers fastest
    // from for this grid (sdfv_march_volume_advice: y-interleaved beyond the last-level cache, texture order below); the
    // other two pin it (Interleaved needs an even height) -- same textures, same frames; A/B runs and tests.
    enum class VolumeLayout { Auto, Plain, Interleaved };
    static std::unique_ptr<SDFViewer> new_voxels(std::array<size_t, 3> voxels, const BoundingBox& bb,
                                                 size_t loading_passes, VolumeLayout layout = VolumeLayout::Auto);

    // scene/sdf/mod.rs:128-217.  Returns the number of LoadingManager iterations consumed, like the reference.
    // An SDF with a device form (device_sdf()): the time budget is checked between passes (the GPU does a whole pass per
    // launch).  ANY other SDFSurface (a ProviderSDF, an application's own class): the ingest path -- sample() runs on the
    // host, on sample_concurrency() threads, run after run of the LoadingManager's order until the budget is spent (at least
    // one voxel, like the reference); the raw 28-byte samples go to the device through pinned double buffers and
    // sdfv_pack_samples does update()'s packing there.  update_required is decided on a host mirror of tex0.r.  (sample() is
    // the caller's code: an exception it throws on a worker thread ends the process, as a panic ends the reference's loop.)
    size_t update(SDFSurface& sdf, std::chrono::nanoseconds max_delta_time);
    // Ingest path knobs: host threads (0 = what the SDF allows, at most the machine's), records per transfer buffer (0 = 16 Ki
    // per thread, between 64 Ki and 4 Mi: a run must outlast the fork/join of its workers by far; 32 B of pinned memory each).
    unsigned host_threads = 0;
    size_t ingest_capacity = 0;
    // Where the ingest path's host time went since the viewer was created (seconds; runs = fork/joins of the workers).
    struct IngestStats {
        double wait_buffer = 0, sample = 0, ship = 0;
        size_t runs = 0, records = 0, visited = 0;
    };
    IngestStats ingest_stats;
    // scene/sdf/mod.rs:220-239
    void commit();
    // lod_dist_between_samples = 2^passes_left (scene/sdf/mod.rs:226), published with the data it describes
    void publish_lod();

    sdfv_grid grid() const;
    float* tex0_device() const { return material.tex0->f32(); }
    float* tex1_device() const { return material.tex1->f32(); }
    int download(float* tex0_host, float* tex1_host) const;  // D2H copy of both textures (debug / GL interop)
    const char* last_error() const { return error_.c_str(); }  // of the most recent update(): "" when it went through

    SDFViewerMaterial material;      // volume.material
    LoadingManager loading_mgr;
    BoundingBox bounding_box;
    std::optional<BoundingBox> changed_box;
    bool changed_box_while_loading = false;
    void* stream = nullptr;          // hipStream_t the kernels are enqueued on

   private:
    SDFViewer(std::array<size_t, 3> voxels, const BoundingBox& bb, size_t passes);
    size_t update_host(SDFSurface& sdf, std::chrono::nanoseconds max_delta_time);  // sdf_viewer_ingest.cpp
    struct Ingest;  // pinned / device transfer buffers, the host mirror of tex0.r, the worker threads
    struct IngestDeleter {
        void operator()(Ingest* p) const;
    };
    std::unique_ptr<Ingest, IngestDeleter> ingest_;  // created by the first update() with a host-only SDF
    bool host_mirror_valid_ = false;  // ingest_'s mirror equals tex0.r (cleared by every fill the device path runs)
    std::string error_;
    bool fresh_ = true;  // both textures still hold new_voxels' AIR_DIST everywhere
    bool loaded_once_ = false;  // some LoadingManager has run to its end over this grid: a later pass without a box finds nothing to do
    bool same_load_ = true;  // every pass so far belongs to ONE load: the SDF and parameters of load_sdf_, no change reported
    std::optional<DeviceSDF> load_sdf_;  // what that load samples
    bool dist_synced_ = false;  // material.dist exists and mirrors tex0.r (kept so by every fill)
    std::shared_ptr<DeviceBuffer> block_;  // owns tex0 and tex1 when they share one allocation
};

}  // namespace sdfviewer
