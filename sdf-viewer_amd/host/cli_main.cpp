// cli_main.cpp -- `sdf-viewer-gpu`: the reference's `app` command line on the MI355X path, headless.
//
//   sdf-viewer-gpu app [--max-voxels-side N] [--loading-passes P] demo [-t M] [-c F] [-l M] [-s F] [-m F] [-d B]
//                  [--width W] [--height H] [--out image.ppm] [--dump-textures prefix] [--frames K]
//   sdf-viewer-gpu app [...] url <library.so | file://library.so>
//
// `url` is the reference's second provider (CliSDFProvider::Url, src/app/cli/mod.rs:41-46: a WebAssembly file exporting the
// per-point ABI).  There is no wasm runtime here; what takes its place is a NATIVE library exporting the same ABI
// (include/sdf_provider.h), loaded through ProviderSDF: the host samples it, the device packs (SDFViewer's ingest path).
// http(s):// and ?wait=true (the file watcher) are out of scope.
//
// Flag names and defaults are the reference's: CliApp (src/app/cli/mod.rs:10-22: max_voxels_side 64,
// loading_passes 2) and the demo SDF's flags (src/sdf/demo/cube.rs:15-18, sphere.rs:11-14, demo/mod.rs:26-29).
//   sdf-viewer-gpu mesh [-o mesh.ply] [-v 64] [marching-cubes] [demo flags after `demo`]
//
// `mesh` is the reference's CliMesher (src/sdf/meshers/mod.rs:22-89): -o/--output (default mesh.ply, "-" = stdout,
// refuses to overwrite), -v/--max-voxels-per-axis (default 64), the mesher subcommand (default marching-cubes); the
// input is the embedded demo SDF instead of -i <wasm>.  Mesh -> postproc -> serialize_ply, as run_custom_out does.
// Where the reference opens a window, this loads the SDF into the two device textures
// (SDFViewer::from_bb/update/commit), renders the default scene camera (scene/mod.rs:82-95) with the raymarch
// kernel and writes the frame as a binary PPM.  The log lines follow scene/mod.rs:180-197.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fstream>
#include <iostream>

#include "mesh.hpp"
#include "provider_sdf.hpp"
#include "sdf_demo.hpp"
#include "sdf_viewer.hpp"

using namespace sdfviewer;

static int usage(const char* msg) {
    if (msg) fprintf(stderr, "error: %s\n\n", msg);
    fprintf(stderr,
            "USAGE:\n    sdf-viewer-gpu app [--max-voxels-side <N>] [--loading-passes <P>] demo [demo flags]\n"
            "                   [--width <W>] [--height <H>] [--out <file.ppm>] [--dump-textures <prefix>] [--frames <K>]\n"
            "    sdf-viewer-gpu app [...] url <library.so>      a native SDF provider (include/sdf_provider.h), sampled on the host\n"
            "    sdf-viewer-gpu mesh [-o <mesh.ply|->] [-v <max-voxels-per-axis>] [marching-cubes] [demo [demo flags]]\n"
            "demo flags: -t/--cube-material <brick|normal>  -c/--cube-half-side <f>  -l/--sphere-material <brick|normal>\n"
            "            -s/--sphere-radius <f>  -m/--max-distance-custom-material <f>  -d/--disable-sphere <true|false>\n");
    return msg ? 2 : 0;
}

// CliMesher::run_cli / run_custom_out, src/sdf/meshers/mod.rs:40-89
static int run_mesh(const std::vector<std::string>& args) {
    std::string output = "mesh.ply";   // meshers/mod.rs:30-31
    MesherConfig cfg;                   // -v default 64, meshers/mod.rs:96-97
    Meshers mesher = Meshers::MarchingCubes;  // Default for Meshers, meshers/mod.rs:130-134
    std::vector<std::string> demo_args;
    bool in_demo = false;
    for (size_t i = 1; i < args.size(); ++i) {
        const std::string& a = args[i];
        auto next = [&](const char* what) -> std::string {
            if (i + 1 >= args.size()) {
                fprintf(stderr, "error: The argument '%s' requires a value but none was supplied\n", what);
                exit(2);
            }
            return args[++i];
        };
        if (in_demo) demo_args.push_back(a);
        else if (a == "-o" || a == "--output") output = next("--output <OUTPUT_FILE>");
        else if (a == "-v" || a == "--max-voxels-per-axis") cfg.max_voxels_per_axis = strtoul(next("--max-voxels-per-axis").c_str(), nullptr, 10);
        else if (a == "-i" || a == "--input") return usage("-i <wasm>: arbitrary wasm cannot run on the GPU; the input is the embedded demo SDF");
        else if (a == "demo") in_demo = true;
        else if (auto m = mesher_from_name(a)) mesher = *m;
        else return usage(("Found argument '" + a + "' which wasn't expected").c_str());
    }
    std::string err;
    auto sdf = SDFDemo::from_args(demo_args, &err);
    if (!sdf) return usage(err.c_str());
    if (sdfv_device_count() == 0) {
        fprintf(stderr, "error: no HIP device visible: sdf-viewer-gpu has no CPU path\n");
        return 1;
    }
    const bool to_stdout = output.empty() || output == "-";
    if (!to_stdout && std::ifstream(output).good()) {
        fprintf(stderr, "Error: Output file already exists\n");  // meshers/mod.rs:52-54
        return 1;
    }
    fprintf(stderr, "Running the meshing algorithm with Config { max_voxels_per_axis: %zu }...\n", cfg.max_voxels_per_axis);
    const auto t0 = std::chrono::steady_clock::now();
    auto mesh = mesh_sdf(mesher, *sdf, cfg, &err);
    if (!mesh) {
        fprintf(stderr, "error: %s\n", err.c_str());
        return 1;
    }
    fprintf(stderr, "Post-processing the mesh (%zu vertices, %zu triangles)...\n", mesh->vertices.size(), mesh->indices.size() / 3);
    if (mesh->postproc(*sdf) != 0) {
        fprintf(stderr, "error: %s\n", sdfv_last_error());
        return 1;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "Serializing output mesh... (meshed and post-processed in %.3fms on the GPU)\n", ms);
    const std::string version = "sdf-viewer-gpu 0.1.0 (MI355X)";  // metadata.rs:13-16 short_version_info()
    size_t bytes;
    if (to_stdout) {
        bytes = mesh->serialize_ply(std::cout, version);
        std::cout.flush();
    } else {
        std::ofstream f(output, std::ios::binary | std::ios::trunc);
        if (!f) {
            perror(output.c_str());
            return 1;
        }
        bytes = mesh->serialize_ply(f, version);
    }
    fprintf(stderr, "Wrote %zu bytes\n", bytes);
    return 0;
}

int main(int argc, char** argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    if (args.empty() || args[0] == "-h" || args[0] == "--help") return usage(nullptr);
    if (args[0] == "mesh") return run_mesh(args);
    if (args[0] != "app") return usage("only the `app` and `mesh` subcommands have a GPU path (server is out of scope)");
    size_t max_voxels_side = 64, loading_passes = 2;  // src/app/cli/mod.rs:13-18
    uint32_t width = 1280, height = 720;
    int frames = 1;
    std::string out = "sdf-viewer-gpu.ppm", dump, url;
    std::vector<std::string> demo_args;
    bool in_demo = false;
    for (size_t i = 1; i < args.size(); ++i) {
        const std::string& a = args[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= args.size()) {
                fprintf(stderr, "error: The argument '%s' requires a value but none was supplied\n", what);
                exit(2);
            }
            return args[++i].c_str();
        };
        if (a == "--max-voxels-side") max_voxels_side = strtoul(next("--max-voxels-side <MAX_VOXELS_SIDE>"), nullptr, 10);
        else if (a == "--loading-passes") loading_passes = strtoul(next("--loading-passes <LOADING_PASSES>"), nullptr, 10);
        else if (a == "--width") width = (uint32_t)strtoul(next("--width"), nullptr, 10);
        else if (a == "--height") height = (uint32_t)strtoul(next("--height"), nullptr, 10);
        else if (a == "--frames") frames = atoi(next("--frames"));
        else if (a == "--out") out = next("--out");
        else if (a == "--dump-textures") dump = next("--dump-textures");
        else if (a == "demo") in_demo = true;
        else if (a == "url") url = next("url <URL>");
        else if (in_demo) demo_args.push_back(a);
        else return usage(("Found argument '" + a + "' which wasn't expected").c_str());
    }
    std::string err;
    std::shared_ptr<SDFSurface> sdf;
    if (!url.empty()) {
        if (url.rfind("http://", 0) == 0 || url.rfind("https://", 0) == 0)
            return usage("url: only local provider libraries are supported (no network, no wasm runtime)");
        if (url.rfind("file://", 0) == 0) url = url.substr(7);
        sdf = ProviderSDF::load(url, &err);
        if (!sdf) {
            fprintf(stderr, "error: %s\n", err.c_str());
            return 1;
        }
    } else {
        sdf = SDFDemo::from_args(demo_args, &err);
        if (!sdf) return usage(err.c_str());
    }

    if (sdfv_device_count() == 0) {
        fprintf(stderr, "error: no HIP device visible: sdf-viewer-gpu has no CPU path\n");
        return 1;
    }
    // set_root_sdf -> scene.set_sdf -> SDFViewer::from_bb (app/mod.rs:99-109, scene/mod.rs:139-156)
    auto viewer = SDFViewer::from_bb(sdf->bounding_box(), max_voxels_side, loading_passes);
    if (!viewer) {
        fprintf(stderr, "error: cannot create the device textures: %s\n", sdfv_last_error());
        return 1;
    }
    // SDFViewerAppScene::render's loading half (scene/mod.rs:166-200): update until nothing is left, commit
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t updates = viewer->update(*sdf, std::chrono::milliseconds(30));
        if (*viewer->last_error()) {
            fprintf(stderr, "error: %s\n", viewer->last_error());
            return 1;
        }
        if (updates == 0) break;
        viewer->commit();
        (void)hipStreamSynchronize((hipStream_t)viewer->stream);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "Loaded SDF chunk (%zu updates) in %.3fms\n", updates, ms);
    }
    viewer->commit();
    fprintf(stderr, "Loaded last SDF chunk (lod %g)\n", (double)viewer->material.lod_dist_between_samples);

    Camera cam;  // scene/mod.rs:82-95 defaults
    cam.set_viewport(width, height);
    DeviceBuffer rgba((size_t)width * height * 16);
    if (!rgba.ok()) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f) {
        if (viewer->material.render(cam, rgba.f32(), nullptr, viewer->stream) != 0) {
            fprintf(stderr, "error: %s\n", sdfv_last_error());
            return 1;
        }
    }
    (void)hipStreamSynchronize((hipStream_t)viewer->stream);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "Rendered %d frame(s) of %ux%u in %.3fms (%.1f Mrays/s)\n", frames, width, height, ms,
            frames * (double)width * height / ms / 1e3);

    std::vector<float> host((size_t)width * height * 4);
    if (hipMemcpy(host.data(), rgba.get(), rgba.bytes(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    // outColor over a black background (blend TRANSPARENCY, material.rs:75-81): rgb * a, quantised to 8 bits
    FILE* fp = fopen(out.c_str(), "wb");
    if (!fp) {
        perror(out.c_str());
        return 1;
    }
    fprintf(fp, "P6\n%u %u\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const float* p = &host[((size_t)y * width + x) * 4];
            for (int c = 0; c < 3; ++c) {
                float v = p[c] * p[3];
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                row[x * 3 + c] = (unsigned char)std::lround(v * 255.0f);
            }
        }
        fwrite(row.data(), 1, row.size(), fp);
    }
    fclose(fp);
    fprintf(stderr, "Wrote %s\n", out.c_str());
    if (!dump.empty()) {
        const size_t n = viewer->material.tex0->bytes() / 4;
        std::vector<float> t0h(n), t1h(n);
        if (viewer->download(t0h.data(), t1h.data()) != 0) return 1;
        for (int k = 0; k < 2; ++k) {
            const std::string path = dump + (k ? ".tex1.f32" : ".tex0.f32");
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) return 1;
            fwrite(k ? t1h.data() : t0h.data(), 4, n, f);
            fclose(f);
        }
    }
    return 0;
}
