// sdf_viewer_ingest.cpp -- SDFViewer::update for an SDFSurface that only the host can sample (see sdf_viewer.hpp).
//
// The reference's loop (src/app/scene/sdf/mod.rs:173-215) does five things per voxel: take the next index of the
// LoadingManager, compute its position, decide update_required, call sdf.sample(pos, false), pack the sample into the two
// textures.  Here the first four stay on the host -- they ARE the host's part: sample() is arbitrary user code -- and the
// fifth moves to the device:
//   - the LoadingManager's order is cut into RUNS of consecutive points of one pass; a run is split among the worker
//     threads (sample_concurrency() of them; the calling thread is worker 0, and the only one for an SDF that does not
//     say it tolerates more), each writing the records it produces into its own stretch of a PINNED buffer;
//   - update_required reads a host mirror of tex0.r (4 B/voxel, kept by this file; rebuilt from the device's distance
//     volume when the device path has written the grid in between) -- never the device textures;
//   - a finished run is copied to the device (async, the viewer's stream) and sdfv_pack_samples packs it into tex0 / tex1 /
//     the distance volume while the workers sample the next run into the other buffer;
//   - the time budget is honoured between runs: the first run is one voxel per worker ("performs at least one update"),
//     every further run is sized from the measured rate to end within half of what is left.
// The set of voxels visited by a call is a prefix of the LoadingManager's remaining order, and the return value counts
// them, exactly as in the reference.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "sdf_viewer.hpp"
#include "worker_pool.hpp"

namespace sdfviewer {

namespace {

// f32::clamp(0.0, 1.0) of scene/sdf/mod.rs:196 (a NaN stays a NaN), the same three lines the device packs with
inline float clamp01_rust(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

}  // namespace

struct SDFViewer::Ingest {
    struct Buffer {
        sdfv_sample* samples = nullptr;    // pinned host
        uint32_t* indices = nullptr;       // pinned host
        sdfv_sample* d_samples = nullptr;  // device
        uint32_t* d_indices = nullptr;     // device
        hipEvent_t done = nullptr;         // the pack launch that read the device half
        bool in_flight = false;
    };
    Buffer buf[2];
    size_t capacity = 0;  // records per buffer
    int next = 0;
    std::vector<float> mirror;     // tex0.r of every voxel, texture order
    std::vector<float> coords[3];  // the voxels' coordinates per axis (scene/sdf/mod.rs:179-182)
    WorkerPool pool;

    ~Ingest() { release(); }
    void release() {
        for (auto& b : buf) {
            if (b.in_flight) (void)hipEventSynchronize(b.done);
            if (b.done) (void)hipEventDestroy(b.done);
            if (b.samples) (void)hipHostFree(b.samples);
            if (b.indices) (void)hipHostFree(b.indices);
            if (b.d_samples) (void)hipFree(b.d_samples);
            if (b.d_indices) (void)hipFree(b.d_indices);
            b = Buffer();
        }
        capacity = 0;
    }
    bool reserve(size_t records) {
        if (capacity == records) return true;
        release();
        for (auto& b : buf) {
            if (hipHostMalloc(reinterpret_cast<void**>(&b.samples), records * sizeof(sdfv_sample), hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc(reinterpret_cast<void**>(&b.indices), records * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&b.d_samples), records * sizeof(sdfv_sample)) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&b.d_indices), records * sizeof(uint32_t)) != hipSuccess ||
                hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                release();
                return false;
            }
        }
        capacity = records;
        return true;
    }
};

void SDFViewer::IngestDeleter::operator()(Ingest* p) const { delete p; }

size_t SDFViewer::update_host(SDFSurface& sdf, std::chrono::nanoseconds max_delta_time) {
    const size_t start_iter = loading_mgr.total_iterations();
    const auto start_time = std::chrono::steady_clock::now();
    if (loading_mgr.step_size() == 0) return 0;  // No more work to do!
    const sdfv_grid g = grid();
    const size_t W = g.dims[0], H = g.dims[1], D = g.dims[2];
    const size_t n_voxels = W * H * D;
    if (n_voxels == 0) return 0;
    if (n_voxels > 0xffffffffull) {
        error_ = "host-sampled grids are addressed with 32-bit voxel indices: at most 2^32 voxels";
        return 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!ingest_) ingest_.reset(new Ingest());
    Ingest& in = *ingest_;
    unsigned threads = sdf.sample_concurrency();
    const unsigned hw = WorkerPool::usable_cpus();
    threads = std::max(1u, std::min({threads, host_threads ? host_threads : hw, 0xffffu}));
    const size_t auto_capacity = std::min<size_t>(std::max<size_t>((size_t)threads << 14, (size_t)1 << 16), (size_t)1 << 22);
    if (!in.reserve(ingest_capacity ? ingest_capacity : std::min(auto_capacity, n_voxels))) {
        error_ = "cannot allocate the ingest buffers (pinned host + device)";
        return 0;
    }
    // A record only rewrites tex0 and tex1.rgb of ITS voxel: everything else must already hold what the reference's
    // textures hold (new_voxels' [AIR_DIST; 4] wherever nothing has been sampled, and tex1.a everywhere).
    if (material.materialize(st) != 0) {
        error_ = sdfv_last_error();
        return 0;
    }
    const float air = sdfv_air_dist();
    // ---- the host mirror of tex0.r ----
    if (!host_mirror_valid_) {
        if (fresh_) {  // nothing has been sampled into this grid yet
            in.mirror.assign(n_voxels, air);
        } else {  // the device path wrote it: fetch the 4 B/voxel volume (or derive one from tex0)
            in.mirror.resize(n_voxels);
            DeviceBuffer derived;
            const float* src = nullptr;
            bool ilv = false;
            if (dist_synced_) {
                src = material.dist->f32();
                ilv = material.dist_interleaved;
            } else {
                derived = DeviceBuffer(n_voxels * 4);
                if (!derived.ok() || sdfv_commit_distance(&g, tex0_device(), derived.f32(), st) != 0) {
                    error_ = "cannot read tex0.r back for the host mirror";
                    return 0;
                }
                src = derived.f32();
            }
            std::vector<float> tmp;
            float* dst = in.mirror.data();
            if (ilv) {
                tmp.resize(n_voxels);
                dst = tmp.data();
            }
            if (hipMemcpyAsync(dst, src, n_voxels * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                (void)hipGetLastError();
                error_ = "cannot read the distance volume back for the host mirror";
                return 0;
            }
            if (ilv)  // entry ((row >> 1) * W + x) * 2 + (row & 1) holds voxel x of row = z * H + y
                for (size_t row = 0; row < H * D; ++row)
                    for (size_t x = 0; x < W; ++x) in.mirror[row * W + x] = tmp[((row >> 1) * W + x) * 2 + (row & 1)];
        }
        host_mirror_valid_ = true;
    }
    // ---- voxel coordinates: idx as f32 / (dim - 1) * size + min, three separately rounded steps (scene/sdf/mod.rs:179-182) ----
    const float bbmin[3] = {bounding_box[0].x, bounding_box[0].y, bounding_box[0].z};
    const float bbmax[3] = {bounding_box[1].x, bounding_box[1].y, bounding_box[1].z};
    for (int a = 0; a < 3; ++a) {
        in.coords[a].resize(g.dims[a]);
        const float dm1 = (float)g.dims[a] - 1.0f, size = bbmax[a] - bbmin[a];
        for (uint32_t i = 0; i < g.dims[a]; ++i) {
            float p = (float)i;
            p = p / dm1;
            p = p * size;
            p = p + bbmin[a];
            in.coords[a][i] = p;
        }
    }
    // this load is no longer one the device path can make assumptions about (SDFV_PASS_SAME_LOAD / FRESH_GRID)
    same_load_ = false;
    load_sdf_.reset();
    fresh_ = false;
    material.pairs_valid = false;

    const uint32_t pack_flags = (dist_synced_ && material.dist_interleaved) ? SDFV_PASS_VOLUME_INTERLEAVED : 0u;
    float* dist_dev = dist_synced_ ? material.dist->f32() : nullptr;
    const bool has_box = changed_box.has_value();
    const BoundingBox box = has_box ? *changed_box : BoundingBox{};

    // A run that has been sampled into its buffer.
    struct Run {
        Ingest::Buffer* b = nullptr;
        size_t n = 0;
        unsigned workers = 0;
        std::vector<size_t> counts;
    };
    Run runs[2];
    for (auto& r : runs) r.counts.resize(threads);
    bool failed = false;
    // Ship a run: the workers' stretches packed back to back on the device (full adjacent stretches in one copy), one launch.
    auto ship = [&](Run& r) {
        if (!r.b || failed) return;
        const auto t0 = std::chrono::steady_clock::now();
        Ingest::Buffer& b = *r.b;
        size_t total = 0;
        for (unsigned t = 0; t < r.workers && !failed; ++t) {
            const size_t lo = r.n * t / r.workers;
            size_t records = r.counts[t];
            unsigned last = t;
            while (last + 1 < r.workers && r.counts[last] == r.n * (last + 1) / r.workers - r.n * last / r.workers) {
                ++last;
                records += r.counts[last];
            }
            if (records) {
                failed = hipMemcpyAsync(b.d_samples + total, b.samples + lo, records * sizeof(sdfv_sample), hipMemcpyHostToDevice, st) != hipSuccess ||
                         hipMemcpyAsync(b.d_indices + total, b.indices + lo, records * sizeof(uint32_t), hipMemcpyHostToDevice, st) != hipSuccess;
                total += records;
            }
            t = last;
        }
        if (!failed && total)
            failed = sdfv_pack_samples(&g, 0, b.d_indices, b.d_samples, total, tex0_device(), tex1_device(), dist_dev, pack_flags, stream) != 0;
        if (failed) {
            (void)hipGetLastError();
            error_ = std::string("ingest: ") + sdfv_last_error();
            host_mirror_valid_ = false;  // the mirror holds samples the device never received
        } else if (total && hipEventRecord(b.done, st) == hipSuccess) {
            b.in_flight = true;
        }
        ingest_stats.ship += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ingest_stats.records += total;
        r.b = nullptr;
    };
    // (Shipping run k - 1 from the calling thread while the workers sample run k was built and measured: on the GPU boxes'
    // 16-CPU quota the 17th thread costs the samplers what the overlap saves -- 69-71 ms against 60-66 for the load of R6.1's
    // grid.  The calling thread is worker 0, then ships.)
    size_t run_len = threads;  // the first run: one voxel per worker
    bool first = true;
    in.pool.begin(threads);
    struct EndSession {
        WorkerPool& pool;
        ~EndSession() { pool.end(); }
    } end_session{in.pool};
    // "while first || start_time.elapsed() < max_delta_time" with a run as the unit of work  (:173)
    while (!failed && (first || std::chrono::steady_clock::now() - start_time < max_delta_time)) {
        first = false;
        const size_t step = loading_mgr.step_size();
        if (step == 0) break;  // No more work to do!
        const size_t n = std::min({run_len, loading_mgr.pass_remaining(), in.capacity});
        Run& run = runs[in.next];
        Ingest::Buffer& b = in.buf[in.next];
        in.next ^= 1;
        const auto wait_start = std::chrono::steady_clock::now();
        if (b.in_flight) {  // the launch that read this buffer two runs ago
            (void)hipEventSynchronize(b.done);
            b.in_flight = false;
        }
        const auto run_start = std::chrono::steady_clock::now();
        ingest_stats.wait_buffer += std::chrono::duration<double>(run_start - wait_start).count();
        const size_t c0 = loading_mgr.cursor();
        const LoadingManager::Index walk = loading_mgr.pass_walk();
        const unsigned workers = (unsigned)std::min<size_t>(threads, n);
        run.b = &b;
        run.n = n;
        run.workers = workers;
        auto sample_stretch = [&](unsigned t) {
            const size_t lo = n * t / workers, hi = n * (t + 1) / workers;  // this worker's stretch of the run (and of the buffer)
            sdfv_sample* out_s = b.samples + lo;
            uint32_t* out_i = b.indices + lo;
            size_t count = 0;
            size_t k = c0 + lo;
            size_t kx = k % walk[0], ky = k / walk[0] % walk[1], kz = k / (walk[0] * walk[1]);
            // The voxels update_required lets through are gathered kGather at a time and sampled with ONE sample_batch call
            // (src/sdf/mod.rs:39's "Batched sampling"; for an SDF that does not override it, the loop of sample() calls it
            // stands for), straight into the pinned records.  A run lies within one pass, so no voxel is visited twice in it:
            // the mirror may be brought up to date after the batch.
            constexpr size_t kGather = 2048;
            Vec3 pos_buf[kGather];
            size_t m = 0;
            auto flush = [&] {
                if (m == 0) return;
                SDFSample* rec = reinterpret_cast<SDFSample*>(out_s + count);
                sdf.sample_batch(pos_buf, m, false, rec);  // :193
                for (size_t j = 0; j < m; ++j)
                    in.mirror[out_i[count + j]] = clamp01_rust(1e-1f + rec[j].distance);  // :196, what the device stores in tex0.r
                count += m;
                m = 0;
            };
            for (size_t i = lo; i < hi; ++i) {
                const size_t x = kx * step, y = ky * step, z = kz * step;
                const size_t flat = (z * H + y) * W + x;  // :177
                const Vec3 pos{in.coords[0][x], in.coords[1][y], in.coords[2][z]};
                // Check if the update is required: was AIR on initial load, or has changed since.  (:184-190)
                bool update_required = in.mirror[flat] == air;
                if (has_box)
                    update_required = update_required || (pos.x >= box[0].x && pos.x <= box[1].x && pos.y >= box[0].y &&
                                                          pos.y <= box[1].y && pos.z >= box[0].z && pos.z <= box[1].z);
                if (update_required) {
                    pos_buf[m] = pos;
                    out_i[count + m] = (uint32_t)flat;
                    if (++m == kGather) flush();
                }
                if (++kx == walk[0]) {
                    kx = 0;
                    if (++ky == walk[1]) {
                        ky = 0;
                        ++kz;
                    }
                }
            }
            flush();
            run.counts[t] = count;
        };
        if (workers > 1) in.pool.run(workers, sample_stretch);
        else sample_stretch(0);
        ingest_stats.sample += std::chrono::duration<double>(std::chrono::steady_clock::now() - run_start).count();
        ship(run);
        if (failed) break;
        loading_mgr.advance(n);
        if (loading_mgr.step_size() == 0) loaded_once_ = true;
        publish_lod();
        ingest_stats.runs += 1;
        ingest_stats.visited += n;
        // ---- the next run: sized to end within half of the budget that is left, growing by at most 8x ----
        const auto now = std::chrono::steady_clock::now();
        const double per_voxel = std::chrono::duration<double>(now - run_start).count() / (double)n;
        const double left = std::chrono::duration<double>(max_delta_time - (now - start_time)).count();
        double want = per_voxel > 0.0 ? 0.5 * left / per_voxel : (double)in.capacity;
        want = std::min(want, 8.0 * (double)n);
        run_len = want < 1.0 ? 1 : (size_t)std::min(want, (double)in.capacity);
    }
    return loading_mgr.total_iterations() - start_iter;
}

}  // namespace sdfviewer
