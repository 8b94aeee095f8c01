// scene.cpp -- see scene.hpp.
#include "scene.hpp"

#include <cstdio>

namespace sdfviewer {

SDFViewerAppScene::SDFViewerAppScene(std::shared_ptr<SDFSurface> s, Clock clock)
    : sdf(std::move(s)), clock_(clock ? std::move(clock) : [] { return std::chrono::steady_clock::now(); }) {
    camera = Camera::new_perspective(0, 0, Vec3{2.5f, 3.0f, 5.0f}, Vec3{0, 0, 0}, Vec3{0, 1, 0}, 45.0f, 0.1f, 1000.0f);
    sdf_viewer = SDFViewer::from_bb(sdf->bounding_box(), 32, 2);  // scene/mod.rs:102
}

bool SDFViewerAppScene::set_sdf(std::shared_ptr<SDFSurface> s, std::optional<size_t> max_voxels_side,
                                std::optional<size_t> loading_passes) {
    if (max_voxels_side) max_voxels_side_ = *max_voxels_side;
    if (loading_passes) loading_passes_ = *loading_passes;
    sdf = std::move(s);
    sdf_viewer = SDFViewer::from_bb(sdf->bounding_box(), max_voxels_side_, loading_passes_);
    sdf_viewer_last_commit_.reset();
    return sdf_viewer != nullptr;
}

RenderReport SDFViewerAppScene::render(uint32_t width, uint32_t height, float* rgba_device) {
    RenderReport rep;
    if (!sdf_viewer) return rep;
    camera.set_viewport(width, height);  // camera.update(&frame_input, ..): viewport only
    // Load more of the SDF in real time (if needed)
    rep.cpu_updates = sdf_viewer->update(*sdf, load_budget);
    if (rep.cpu_updates > 0) {
        // Update the GPU texture sparingly
        const auto now = clock_();
        if (!sdf_viewer_last_commit_ || now - *sdf_viewer_last_commit_ > commit_interval) {
            sdf_viewer->commit();
            sdf_viewer_last_commit_ = clock_();
            rep.committed = true;
        }
        rep.request_repaint = true;
    } else if (sdf_viewer_last_commit_) {
        sdf_viewer->commit();
        sdf_viewer_last_commit_.reset();
        rep.committed = true;
        rep.last_chunk = true;
        rep.request_repaint = true;
    }
    if (rgba_device && width && height) sdf_viewer->material.render(camera, rgba_device, nullptr, sdf_viewer->stream);
    return rep;
}

std::optional<std::pair<float, std::string>> SDFViewerAppScene::load_progress() const {
    if (!sdf_viewer || !sdf_viewer_last_commit_) return std::nullopt;
    const size_t remaining = sdf_viewer->loading_mgr.len();
    const size_t done = sdf_viewer->loading_mgr.total_iterations();
    const size_t total = done + remaining;
    const float progress = (float)done / (float)total;
    char buf[160];
    snprintf(buf, sizeof(buf), "Loading SDF %.2f%% (%zu levels of detail left, evaluations: %zu / %zu)",
             (double)(progress * 100.0f), sdf_viewer->loading_mgr.passes_left(), done, total);
    return std::make_pair(progress, std::string(buf));
}

}  // namespace sdfviewer
