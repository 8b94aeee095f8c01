// host_load_bench.cpp -- what a host that drops libsdfviewer_host.so in actually pays for a load:
//   SDFViewer::from_bb -> update(sdf, budget) [-> update ...] -> commit        (reference: src/app/scene/mod.rs:139-156 set_sdf,
//   src/app/scene/sdf/mod.rs:46-101 from_bb / new_voxels, :128-217 update, :220-239 commit)
// through the C++ mirror of the reference's classes, NOT through the kernels' own entry points.  bench.py runs this binary and
// attaches its JSON as "host_load".  Timed per load, HIP events on the viewer's stream + the host's wall clock:
//   create_ms      from_bb alone (hipMalloc of both textures + the distance volume; nothing is launched or waited for)
//   load_ms        update() to completion + commit()'s march volume, GPU time between events    <- the number to compare with
//                  the fused fill kernel (bench.py "ms_per_step_fill")
//   update_ms      ... the update() share
//   enqueue_us     host time spent inside update() + commit() (they only enqueue)
// Forms: "dense" = one update() with the reference's 30 ms budget (every pass fits: the dense shortcut); "progressive" = one
// pass per update() call (zero budget), the reference's order of passes.  A fresh viewer per repetition (a load starts from
// new_voxels), created outside the timed region.
// --ingest <provider.so>: the same load for an SDF that only the HOST can sample (a library exporting include/sdf_provider.h,
// loaded through ProviderSDF): update() samples on `threads` host threads and the device packs (sdf_viewer_ingest.cpp).  CPU-bound
// by construction; reported: whole-load Mvoxels/s with an unlimited budget, and the reference's frame loop (30 ms per call).
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "provider_sdf.hpp"
#include "worker_pool.hpp"
#include "sdf_demo.hpp"
#include "sdf_viewer.hpp"

using namespace sdfviewer;
using Clock = std::chrono::steady_clock;

static double median(std::vector<double> v) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

struct Result {
    double create_ms, load_ms, update_ms, commit_ms, enqueue_us;
    size_t iterations;
    int passes_run;
};

static bool one_load(size_t side, size_t passes, bool progressive, hipEvent_t e0, hipEvent_t e1, hipEvent_t e2, Result& r) {
    std::string err;
    auto demo = SDFDemo::from_args({}, &err);
    if (!demo) return false;
    const BoundingBox bb = demo->bounding_box();
    const auto t0 = Clock::now();
    auto v = SDFViewer::from_bb(bb, side, passes);
    const auto t1 = Clock::now();
    if (!v) return false;
    r.create_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    hipStream_t st = (hipStream_t)v->stream;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    r.iterations = 0;
    r.passes_run = 0;
    const auto h0 = Clock::now();
    (void)hipEventRecord(e0, st);
    for (;;) {
        const size_t n = v->update(*demo, progressive ? std::chrono::nanoseconds(0) : std::chrono::milliseconds(30));
        if (n == 0) break;
        r.iterations += n;
        ++r.passes_run;
    }
    (void)hipEventRecord(e1, st);
    v->commit();
    (void)hipEventRecord(e2, st);
    const auto h1 = Clock::now();
    if (hipEventSynchronize(e2) != hipSuccess) return false;
    float a = 0.0f, b = 0.0f;
    (void)hipEventElapsedTime(&a, e0, e1);
    (void)hipEventElapsedTime(&b, e1, e2);
    r.update_ms = a;
    r.commit_ms = b;
    r.load_ms = a + b;
    r.enqueue_us = std::chrono::duration<double, std::micro>(h1 - h0).count();
    return v->material.lod_dist_between_samples == 1.0f && !v->material.undefined_rows;
}

// One load of a host-sampled SDF.  budget_ms < 0: one update() with an unlimited budget; otherwise the frame loop.
// (the median of `repeats` loads by load time: the host cores are shared and rationed -- one load in three runs 30 % off)
static int ingest_load_once(SDFSurface& sdf, size_t side, size_t passes, unsigned threads, double budget_ms, const char* key,
                            std::string* line, double* load_ms);
static int ingest_load(SDFSurface& sdf, size_t side, size_t passes, unsigned threads, double budget_ms, const char* key, int repeats = 3) {
    std::vector<std::pair<double, std::string>> loads;
    for (int i = 0; i < repeats; ++i) {
        std::string line;
        double ms = 0.0;
        if (ingest_load_once(sdf, side, passes, threads, budget_ms, key, &line, &ms) != 0) return 1;
        loads.emplace_back(ms, line);
    }
    std::sort(loads.begin(), loads.end());
    printf("%s", loads[loads.size() / 2].second.c_str());
    return 0;
}
static int ingest_load_once(SDFSurface& sdf, size_t side, size_t passes, unsigned threads, double budget_ms, const char* key,
                            std::string* line, double* load_ms) {
    auto v = SDFViewer::from_bb(sdf.bounding_box(), side, passes);
    if (!v) return 1;
    v->host_threads = threads;
    hipStream_t st = (hipStream_t)v->stream;
    size_t iterations = 0, calls = 0;
    double worst_call_ms = 0.0;
    const auto budget = budget_ms < 0 ? std::chrono::nanoseconds(std::chrono::hours(24))
                                      : std::chrono::nanoseconds((long long)(budget_ms * 1e6));
    // set-up (pinned + device transfer buffers, the host mirror, the worker threads) happens in the first call: a zero budget
    // makes it sample one voxel per worker and return, so that it can be told apart from the sampling rate
    const auto s0 = Clock::now();
    iterations += v->update(sdf, std::chrono::nanoseconds(0));
    const double setup_ms = std::chrono::duration<double, std::milli>(Clock::now() - s0).count();
    const auto t0 = Clock::now();
    for (;;) {
        const auto c0 = Clock::now();
        const size_t n = v->update(sdf, budget);
        worst_call_ms = std::max(worst_call_ms, std::chrono::duration<double, std::milli>(Clock::now() - c0).count());
        if (n == 0) break;
        iterations += n;
        ++calls;
    }
    if (*v->last_error()) {
        fprintf(stderr, "ingest failed: %s\n", v->last_error());
        return 1;
    }
    v->commit();
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    const double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    const double voxels = (double)v->material.tex_size[0] * v->material.tex_size[1] * v->material.tex_size[2];
    const auto& is = v->ingest_stats;
    fprintf(stderr, "%s: runs %zu, records %zu of %zu visited; host ms: wait for a buffer %.2f, sample %.2f, ship %.2f\n", key, is.runs,
            is.records, is.visited, is.wait_buffer * 1e3, is.sample * 1e3, is.ship * 1e3);
    char buf[512];
    snprintf(buf, sizeof buf, ", \"%s\": {\"setup_ms\": %.2f, \"load_ms\": %.2f, \"update_calls\": %zu, \"iterations\": %zu, \"worst_call_ms\": %.2f, "
             "\"Mvoxels_per_s\": %.2f}", key, setup_ms, ms, calls, iterations, worst_call_ms, ms > 0 ? voxels / (ms * 1e-3) / 1e6 : 0.0);
    *line = buf;
    *load_ms = ms;
    return 0;
}

int main(int argc, char** argv) {
    size_t side = 256, passes = 2;
    int reps = 20;
    const char* ingest = nullptr;
    unsigned threads = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--side") && i + 1 < argc) side = strtoul(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--passes") && i + 1 < argc) passes = strtoul(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--ingest") && i + 1 < argc) ingest = argv[++i];
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = (unsigned)atoi(argv[++i]);
        else {
            fprintf(stderr, "usage: %s [--side N] [--passes P] [--reps R] [--ingest provider.so [--threads T]]\n", argv[0]);
            return 2;
        }
    }
    if (sdfv_device_count() <= 0) {
        fprintf(stderr, "no HIP device\n");
        return 3;
    }
    if (ingest) {
        std::string err;
        auto sdf = ProviderSDF::load(ingest, &err);
        if (!sdf) {
            fprintf(stderr, "%s\n", err.c_str());
            return 2;
        }
        const unsigned hw = WorkerPool::usable_cpus();
        const unsigned used = std::max(1u, std::min(sdf->sample_concurrency(), threads ? threads : hw));
        printf("{\"side\": %zu, \"loading_passes\": %zu, \"provider\": \"%s\", \"host_cores\": %u, \"threads\": %u", side, passes,
               sdf->name().c_str(), hw, used);
        if (ingest_load(*sdf, side, passes, threads, -1.0, "whole_load") != 0) return 1;
        if (ingest_load(*sdf, side, passes, threads, 30.0, "frame_loop_30ms") != 0) return 1;
        if (ingest_load(*sdf, side, passes, 1, -1.0, "whole_load_1_thread", 1) != 0) return 1;
        printf("}\n");
        return 0;
    }
    hipEvent_t e0, e1, e2;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess) return 3;
    const double voxels = (double)side * side * side;
    printf("{\"side\": %zu, \"loading_passes\": %zu, \"reps\": %d", side, passes, reps);
    for (int form = 0; form < 2; ++form) {
        const bool progressive = form == 1;
        std::vector<double> load, update, commit, create, enqueue;
        Result r{};
        bool ok = true;
        // keep the device busy before the timed loads (it idles at a few hundred MHz; bench.py does the same)
        for (int i = 0; i < 3 && ok; ++i) ok = one_load(side, passes, progressive, e0, e1, e2, r);
        for (int i = 0; i < reps && ok; ++i) {
            ok = one_load(side, passes, progressive, e0, e1, e2, r);
            load.push_back(r.load_ms);
            update.push_back(r.update_ms);
            commit.push_back(r.commit_ms);
            create.push_back(r.create_ms);
            enqueue.push_back(r.enqueue_us);
        }
        if (!ok) {
            fprintf(stderr, "load failed: %s\n", sdfv_last_error());
            return 1;
        }
        // algorithmic bytes of the update() share: every voxel's two texels + its distance-volume entry once (36 B) for the
        // dense form; the progressive form also writes the rows its coarser passes visit (36 B per voxel of those rows)
        double bytes = 36.0 * voxels;
        if (progressive)
            for (size_t step = (size_t)1 << (passes - 1); step > 1; step >>= 1)
                bytes += 36.0 * (double)side * (double)((side + step - 1) / step) * (double)((side + step - 1) / step);
        const double up = median(update);
        printf(", \"%s\": {\"load_ms\": %.4f, \"update_ms\": %.4f, \"commit_ms\": %.4f, \"create_ms\": %.3f, \"enqueue_us\": %.1f, "
               "\"update_calls\": %d, \"iterations\": %zu, \"update_bytes\": %.0f, \"update_TBps\": %.3f, \"Mvoxels_per_s\": %.0f}",
               progressive ? "progressive" : "dense", median(load), up, median(commit), median(create), median(enqueue),
               r.passes_run, r.iterations, bytes, up > 0 ? bytes / (up * 1e-3) / 1e12 : 0.0,
               median(load) > 0 ? voxels / (median(load) * 1e-3) / 1e6 : 0.0);
    }
    printf("}\n");
    return 0;
}
