// scene.hpp -- C++ mirror of the reference's scene object, the caller of the hot path once per frame:
//   SDFViewerAppScene::{new, set_sdf, render, load_progress}   src/app/scene/mod.rs:34-248
// What it fixes for the path: the default camera and the single ambient light (scene/mod.rs:82-112), the 30 ms
// per-frame loading budget (:168), the >= 500 ms spacing of commits while loading (:171-174), the final commit
// (:194-200) and the progress text (:228-247).  Windowing / egui / camera interaction are out of scope; the
// clock is injectable so the scheduling can be tested deterministically.
#pragma once

#include <chrono>
#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <utility>

#include "sdf_surface.hpp"
#include "sdf_viewer.hpp"

namespace sdfviewer {

struct RenderReport {          // what the reference logs per frame (scene/mod.rs:180-197)
    size_t cpu_updates = 0;    // "Loaded SDF chunk ({} updates)"
    bool committed = false;    // "... + {:?} (GPU)" vs "+ skipped (GPU)"
    bool last_chunk = false;   // "Loaded last SDF chunk"
    bool request_repaint = false;
};

class SDFViewerAppScene {
   public:
    using Clock = std::function<std::chrono::steady_clock::time_point()>;

    // scene/mod.rs:80-136: default camera, one AmbientLight(1.0, WHITE), a 32^3 / 2-pass placeholder viewer
    explicit SDFViewerAppScene(std::shared_ptr<SDFSurface> sdf, Clock clock = nullptr);

    // scene/mod.rs:139-156: (re)create the viewer for this SDF; None keeps the previous value
    bool set_sdf(std::shared_ptr<SDFSurface> sdf, std::optional<size_t> max_voxels_side,
                 std::optional<size_t> loading_passes);

    // scene/mod.rs:158-225: one frame -- load within the budget, commit sparingly, draw the volume into
    // rgba_device (width*height*4 floats, DEVICE).  Returns what happened.
    RenderReport render(uint32_t width, uint32_t height, float* rgba_device);

    // scene/mod.rs:228-247
    std::optional<std::pair<float, std::string>> load_progress() const;

    Camera camera;                         // CameraController::camera
    std::unique_ptr<SDFViewer> sdf_viewer;
    std::shared_ptr<SDFSurface> sdf;
    std::chrono::milliseconds load_budget{30};      // scene/mod.rs:168
    std::chrono::milliseconds commit_interval{500};  // scene/mod.rs:173

   private:
    Clock clock_;
    std::optional<std::chrono::steady_clock::time_point> sdf_viewer_last_commit_;
    size_t max_voxels_side_ = 32, loading_passes_ = 2;
};

}  // namespace sdfviewer
