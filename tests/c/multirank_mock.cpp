// tests/c/multirank_mock.cpp -- the library's MULTI-RANK code on ONE GPU: N ranks = N host threads of this process, all on
// device 0, each with its own sdfv_slab_comm over tests/c/mock_rccl.cpp (run with that mock's directory first in
// LD_LIBRARY_PATH: the library dlopens "librccl.so.1").  RCCL itself refuses two ranks on one device and no run of this
// repository has ever had two devices, so the non-periodic branches -- a rank with a lower neighbour only, with an upper one
// only, with both; the ghost offsets; the gathers' destination rank; sdfv_slab_march's rounds between DIFFERENT ranks -- had
// never executed.  What is checked, per world size and per rank, against the SAME library's single-device results (which the
// parity suite pins to the oracle):
//   * sdfv_slab_fill_step / _commit in both message forms, sdfv_slab_halo_exchange after a plain fill, one and two upper ghost
//     slices: [ghost | owned | ghost] of tex0, tex1 and the volume == those slices of the dense fill of the whole grid;
//   * sdfv_comm_allgather_slabs: every rank's replica == the dense fill;
//   * sdfv_slab_march with SDFV_MARCH_MERGE: every rank's image == sdfv_raymarch over the whole grid, bit for bit, status 0 0;
//   * sdfv_comm_gather_cameras / sdfv_comm_gather_bands (8- and 16-row bands) to rank 0 and to the last rank == the batch rendered
//     in one piece.
// usage: multirank_mock [world=3]    exit code 0 and a line "ok ..." when everything matched
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "sdfgrid.h"

#define HIP(x)                                                                                  \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)
#define SDFV(x)                                                                                              \
    do {                                                                                                     \
        int r_ = (x);                                                                                        \
        if (r_ != 0) {                                                                                       \
            fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, sdfv_last_error());          \
            exit(3);                                                                                         \
        }                                                                                                    \
    } while (0)

static uint32_t W = 64, H = 48, D = 41;  // set per geometry in main(): rows of whole workgroups (the boundary-first step applies) /
                                         // rows that are not and slabs of two slices (the step falls back to fill, then exchange)
static const uint32_t IW = 160, IH = 88;       // image: 5.5 bands of 16 rows
static std::atomic<int> g_failures{0};

static void expect(bool ok, int rank, const char* what) {
    if (!ok) {
        fprintf(stderr, "MISMATCH rank %d: %s\n", rank, what);
        g_failures += 1;
    }
}

struct Reference {
    std::vector<float> tex0, tex1, dist;  // the whole grid
    std::vector<float> frame;             // one camera, sdfv_raymarch over the whole grid
    std::vector<float> batch;             // n_cams cameras
    sdfv_camera cam, cams[5];
    sdfv_render_params rp;
    sdfv_demo_params prm;
    sdfv_grid grid;
};

static void z_range(int rank, int world, uint32_t* z0, uint32_t* z1) {
    *z0 = (uint32_t)((uint64_t)D * rank / world);
    *z1 = (uint32_t)((uint64_t)D * (rank + 1) / world);
}

static void rank_main(int rank, int world, const unsigned char* id, uint32_t comm_flags, const Reference* ref) {
    HIP(hipSetDevice(0));
    hipStream_t st;
    HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    sdfv_slab_comm* comm = nullptr;
    SDFV(sdfv_slab_comm_create(id, rank, world, comm_flags, &comm));
    uint32_t glo = 0, ghi = 0;
    SDFV(sdfv_slab_comm_info(comm, &glo, &ghi, nullptr));
    int r_rank = -1, r_world = -1;
    SDFV(sdfv_slab_comm_ranks(comm, &r_rank, &r_world));
    expect(r_rank == rank && r_world == world, rank, "sdfv_slab_comm_ranks");
    expect(glo == (rank > 0 ? 1u : 0u) && ghi == (rank < world - 1 ? ((comm_flags & SDFV_COMM_HALO2) ? 2u : 1u) : 0u), rank, "ghost depths");
    uint32_t z0, z1;
    z_range(rank, world, &z0, &z1);
    sdfv_grid slab = ref->grid;
    slab.z_begin = z0;
    slab.z_end = z1;
    const size_t slice = (size_t)W * H, held = (z1 - z0) + glo + ghi;
    float *t0, *t1, *dv;
    HIP(hipMalloc((void**)&t0, held * slice * 16));
    HIP(hipMalloc((void**)&t1, held * slice * 16));
    HIP(hipMalloc((void**)&dv, held * slice * 4));
    std::vector<float> h0(held * slice * 4), h1(held * slice * 4), hd(held * slice);
    const size_t first = z0 - glo;  // first global slice held
    auto check_slab = [&](const char* what, bool with_dist) {
        HIP(hipStreamSynchronize(st));
        HIP(hipMemcpy(h0.data(), t0, h0.size() * 4, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(h1.data(), t1, h1.size() * 4, hipMemcpyDeviceToHost));
        expect(memcmp(h0.data(), ref->tex0.data() + first * slice * 4, h0.size() * 4) == 0, rank, (std::string(what) + ": tex0 incl. ghosts").c_str());
        expect(memcmp(h1.data(), ref->tex1.data() + first * slice * 4, h1.size() * 4) == 0, rank, (std::string(what) + ": tex1 incl. ghosts").c_str());
        if (with_dist) {
            HIP(hipMemcpy(hd.data(), dv, hd.size() * 4, hipMemcpyDeviceToHost));
            expect(memcmp(hd.data(), ref->dist.data() + first * slice, hd.size() * 4) == 0, rank, (std::string(what) + ": volume incl. ghosts").c_str());
        }
    };
    auto poison = [&]() {
        HIP(hipMemsetAsync(t0, 0xA5, held * slice * 16, st));
        HIP(hipMemsetAsync(t1, 0xA5, held * slice * 16, st));
        HIP(hipMemsetAsync(dv, 0xA5, held * slice * 4, st));
    };
    // ---- the fill step, both message forms, with and without the fused volume ----
    const uint64_t forms[4] = {SDFV_STEP_SIDE_BOUNDARY, SDFV_STEP_SIDE_BOUNDARY | SDFV_STEP_UNPACKED, SDFV_STEP_SIDE_BOUNDARY | SDFV_STEP_START_EVENT,
                               SDFV_STEP_SIDE_BOUNDARY | SDFV_STEP_DEFER_JOIN};
    const char* form_names[4] = {"packed messages", "per-texture messages", "packed, start event", "packed, deferred join"};
    for (int f = 0; f < 4; ++f) {
        SDFV(sdfv_set_option(SDFV_OPT_SLAB_STEP_FORM, forms[f]));
        poison();
        SDFV(sdfv_slab_fill_step_commit(comm, &ref->prm, SDFV_SDF_DEMO, &slab, t0, t1, dv, st));
        if (f == 3) {  // two more steps without a join, then ONE join before anything reads the ghosts
            SDFV(sdfv_slab_fill_step_commit(comm, &ref->prm, SDFV_SDF_DEMO, &slab, t0, t1, dv, st));
            SDFV(sdfv_slab_fill_step_commit(comm, &ref->prm, SDFV_SDF_DEMO, &slab, t0, t1, dv, st));
            SDFV(sdfv_slab_comm_join(comm, st));
        }
        check_slab((std::string("fused step, ") + form_names[f]).c_str(), true);
        poison();
        SDFV(sdfv_slab_fill_step(comm, &ref->prm, SDFV_SDF_DEMO, &slab, t0, t1, st));
        if (f == 3) SDFV(sdfv_slab_comm_join(comm, st));
        check_slab((std::string("step, ") + form_names[f]).c_str(), false);
    }
    SDFV(sdfv_set_option(SDFV_OPT_SLAB_STEP_FORM, 0));
    for (int k = 0; k < 3; ++k) SDFV(sdfv_slab_fill_step_commit(comm, &ref->prm, SDFV_SDF_DEMO, &slab, t0, t1, dv, st));  // back to back
    check_slab("three default steps back to back", true);
    // ---- the exchange alone after a plain fill of the owned slices ----
    poison();
    SDFV(sdfv_fill_grid(&ref->prm, SDFV_SDF_DEMO, &slab, t0 + glo * slice * 4, t1 + glo * slice * 4, st));
    SDFV(sdfv_slab_halo_exchange(comm, &slab, t0, t1, st));
    check_slab("sdfv_fill_grid + sdfv_slab_halo_exchange", false);
    SDFV(sdfv_slab_fill_step_commit(comm, &ref->prm, SDFV_SDF_DEMO, &slab, t0, t1, dv, st));
    // ---- replicas by all-gather ----
    {
        std::vector<uint32_t> zb(world + 1);
        for (int r = 0; r <= world; ++r) zb[r] = (uint32_t)((uint64_t)D * r / world);
        float *o0, *o1, *od;
        HIP(hipMalloc((void**)&o0, (size_t)D * slice * 16));
        HIP(hipMalloc((void**)&o1, (size_t)D * slice * 16));
        HIP(hipMalloc((void**)&od, (size_t)D * slice * 4));
        const uint32_t dims[3] = {W, H, D};
        SDFV(sdfv_comm_allgather_slabs(comm, dims, zb.data(), t0 + glo * slice * 4, t1 + glo * slice * 4, dv + glo * slice, o0, o1, od, st));
        HIP(hipStreamSynchronize(st));
        std::vector<float> g0((size_t)D * slice * 4), gd((size_t)D * slice);
        HIP(hipMemcpy(g0.data(), o0, g0.size() * 4, hipMemcpyDeviceToHost));
        expect(memcmp(g0.data(), ref->tex0.data(), g0.size() * 4) == 0, rank, "allgather_slabs: tex0 replica");
        HIP(hipMemcpy(g0.data(), o1, g0.size() * 4, hipMemcpyDeviceToHost));
        expect(memcmp(g0.data(), ref->tex1.data(), g0.size() * 4) == 0, rank, "allgather_slabs: tex1 replica");
        HIP(hipMemcpy(gd.data(), od, gd.size() * 4, hipMemcpyDeviceToHost));
        expect(memcmp(gd.data(), ref->dist.data(), gd.size() * 4) == 0, rank, "allgather_slabs: volume replica");
        // ---- config 5's gathers over the replica: whole cameras, then tile bands of 8 and of 16 rows ----
        const uint32_t n_cams = 5;
        const size_t image = (size_t)IW * IH * 4;
        for (int dst : {0, world - 1}) {
            const uint32_t c0 = n_cams * rank / world, c1 = n_cams * (rank + 1) / world;
            float *part, *out = nullptr;
            HIP(hipMalloc((void**)&part, (c1 - c0 ? c1 - c0 : 1) * image * 4));
            if (rank == dst) HIP(hipMalloc((void**)&out, n_cams * image * 4));
            if (c1 > c0) SDFV(sdfv_raymarch(&ref->rp, o0, o1, ref->cams + c0, c1 - c0, IW, IH, 0, IH, part, nullptr, st));
            SDFV(sdfv_comm_gather_cameras(comm, part, n_cams, IW, IH, 4, dst, out, st));
            HIP(hipStreamSynchronize(st));
            if (rank == dst) {
                std::vector<float> got(n_cams * image);
                HIP(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                expect(memcmp(got.data(), ref->batch.data(), got.size() * 4) == 0, rank, "gather_cameras == the batch in one piece");
                HIP(hipFree(out));
            }
            HIP(hipFree(part));
            for (uint32_t bh : {8u, 16u}) {
                const uint32_t rows = sdfv_band_rows_ex(IH, (uint32_t)rank, (uint32_t)world, bh);
                float *bpart, *bout = nullptr;
                void* scratch = nullptr;
                HIP(hipMalloc((void**)&bpart, ((size_t)n_cams * rows * IW * 4 + 4) * 4));
                const size_t sb = sdfv_comm_gather_bands_scratch_bytes(comm, dst, bh, n_cams, IW, IH, 4);
                if (rank == dst) {
                    HIP(hipMalloc((void**)&bout, n_cams * image * 4));
                    HIP(hipMalloc(&scratch, sb ? sb : 16));
                }
                sdfv_march_desc d;
                memset(&d, 0, sizeof d);
                d.size = sizeof d;
                d.rp = &ref->rp;
                d.tex0 = o0, d.tex1 = o1, d.dist = od;
                d.cameras = ref->cams, d.n_cameras = n_cams;
                d.width = IW, d.height = IH, d.y0 = 0, d.y1 = IH;
                d.band_first = (uint32_t)rank, d.band_step = (uint32_t)world, d.band_height = bh;
                d.rgba = bpart;
                SDFV(sdfv_raymarch_ex(&d, st));
                SDFV(sdfv_comm_gather_bands(comm, bpart, bh, n_cams, IW, IH, 4, dst, bout, scratch, sb, st));
                HIP(hipStreamSynchronize(st));
                if (rank == dst) {
                    std::vector<float> got(n_cams * image);
                    HIP(hipMemcpy(got.data(), bout, got.size() * 4, hipMemcpyDeviceToHost));
                    expect(memcmp(got.data(), ref->batch.data(), got.size() * 4) == 0, rank, bh == 8 ? "gather_bands (8 rows) == the batch" : "gather_bands (16 rows) == the batch");
                    HIP(hipFree(bout));
                    HIP(hipFree(scratch));
                }
                HIP(hipFree(bpart));
            }
        }
        HIP(hipFree(o0));
        HIP(hipFree(o1));
        HIP(hipFree(od));
    }
    // ---- the march over the sharded grid, rays handed between the ranks, images merged ----
    for (uint32_t capacity : {IW * IH, IW * IH / 4}) {
        const size_t sb = sdfv_slab_march_scratch_bytes(capacity);
        void* scratch;
        float* rgba;
        uint32_t* status;
        HIP(hipMalloc(&scratch, sb));
        HIP(hipMalloc((void**)&rgba, (size_t)IW * IH * 16));
        HIP(hipMalloc((void**)&status, 8));
        SDFV(sdfv_slab_march(comm, &ref->rp, &slab, t0, t1, &ref->cam, IW, IH, rgba, nullptr, scratch, sb, capacity, SDFV_MARCH_MERGE, status, st));
        HIP(hipStreamSynchronize(st));
        uint32_t hs[2];
        HIP(hipMemcpy(hs, status, 8, hipMemcpyDeviceToHost));
        std::vector<float> got((size_t)IW * IH * 4);
        HIP(hipMemcpy(got.data(), rgba, got.size() * 4, hipMemcpyDeviceToHost));
        if (hs[0] == 0) {  // (a list that overflowed a bounded capacity says so; the full-capacity run must not)
            expect(hs[1] == 0, rank, "sdfv_slab_march: rays left over");
            expect(memcmp(got.data(), ref->frame.data(), got.size() * 4) == 0, rank, "sdfv_slab_march (merged) == sdfv_raymarch over the whole grid");
        } else {
            expect(capacity != IW * IH, rank, "sdfv_slab_march: overflow with lists that hold every pixel");
        }
        HIP(hipFree(scratch));
        HIP(hipFree(rgba));
        HIP(hipFree(status));
    }
    HIP(hipStreamSynchronize(st));
    SDFV(sdfv_slab_comm_destroy(comm));
    HIP(hipFree(t0));
    HIP(hipFree(t1));
    HIP(hipFree(dv));
    HIP(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 3;
    if (world < 2 || world > 8) {
        fprintf(stderr, "world must be 2..8\n");
        return 1;
    }
    HIP(hipSetDevice(0));
    int sets = 0;
    size_t hits = 0;
    const uint32_t geometries[2][3] = {{64, 48, 41}, {40, 20, 17}};
    for (int gi = 0; gi < 2; ++gi) {
    W = geometries[gi][0], H = geometries[gi][1], D = geometries[gi][2];
    if (D < (uint32_t)world * 2) continue;  // (every rank needs two slices: the halo's second slice comes from ONE neighbour)
    Reference ref;
    sdfv_demo_params_default(&ref.prm);
    memset(&ref.grid, 0, sizeof ref.grid);
    ref.grid.dims[0] = W, ref.grid.dims[1] = H, ref.grid.dims[2] = D;
    for (int i = 0; i < 3; ++i) ref.grid.bb_min[i] = -1.0f, ref.grid.bb_max[i] = 1.0f;
    ref.grid.z_begin = 0, ref.grid.z_end = D;
    const size_t n = (size_t)W * H * D;
    float *t0, *t1, *dv, *img;
    HIP(hipMalloc((void**)&t0, n * 16));
    HIP(hipMalloc((void**)&t1, n * 16));
    HIP(hipMalloc((void**)&dv, n * 4));
    SDFV(sdfv_fill_grid_commit(&ref.prm, SDFV_SDF_DEMO, &ref.grid, t0, t1, dv, nullptr));
    ref.tex0.resize(n * 4), ref.tex1.resize(n * 4), ref.dist.resize(n);
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(ref.tex0.data(), t0, n * 16, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ref.tex1.data(), t1, n * 16, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ref.dist.data(), dv, n * 4, hipMemcpyDeviceToHost));
    sdfv_render_params_default(&ref.rp, &ref.grid);
    const float target[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    const float eye[3] = {1.5f, 2.0f, 3.5f};
    SDFV(sdfv_camera_look_at(&ref.cam, eye, target, up, 45.0f, (float)IW / IH, 0.1f, 1000.0f));
    for (int k = 0; k < 5; ++k) {
        const float e[3] = {2.5f - 1.2f * k, 3.0f - 0.4f * k, 5.0f - 2.1f * k};
        SDFV(sdfv_camera_look_at(&ref.cams[k], e, target, up, 45.0f, (float)IW / IH, 0.1f, 1000.0f));
    }
    const size_t image = (size_t)IW * IH * 4;
    HIP(hipMalloc((void**)&img, 5 * image * 4));
    SDFV(sdfv_raymarch(&ref.rp, t0, t1, &ref.cam, 1, IW, IH, 0, IH, img, nullptr, nullptr));
    HIP(hipDeviceSynchronize());
    ref.frame.resize(image);
    HIP(hipMemcpy(ref.frame.data(), img, image * 4, hipMemcpyDeviceToHost));
    SDFV(sdfv_raymarch(&ref.rp, t0, t1, ref.cams, 5, IW, IH, 0, IH, img, nullptr, nullptr));
    HIP(hipDeviceSynchronize());
    ref.batch.resize(5 * image);
    HIP(hipMemcpy(ref.batch.data(), img, 5 * image * 4, hipMemcpyDeviceToHost));
    hits = 0;
    for (size_t i = 3; i < image; i += 4) hits += ref.frame[i] > 0.0f;
    if (hits < 500) {
        fprintf(stderr, "the reference frame shows almost nothing (%zu hits)\n", hits);
        return 1;
    }
    HIP(hipFree(t0));
    HIP(hipFree(t1));
    HIP(hipFree(dv));
    HIP(hipFree(img));
    for (uint32_t flags : {0u, (uint32_t)SDFV_COMM_HALO2}) {
        unsigned char id[SDFV_COMM_ID_BYTES];
        SDFV(sdfv_slab_comm_unique_id(id));
        std::vector<std::thread> ranks;
        for (int r = 0; r < world; ++r) ranks.emplace_back(rank_main, r, world, id, flags, &ref);
        for (auto& t : ranks) t.join();
        sets += 1;
    }
    }
    if (g_failures.load() != 0) {
        printf("FAILED: %d mismatches over %d communicator sets of %d ranks\n", g_failures.load(), sets, world);
        return 1;
    }
    printf("ok %d ranks x %d communicator sets (two geometries; one / two upper ghost slices) on one device over the mock RCCL: fill steps "
           "in four forms, halo, all-gather, both gathers, sharded march (%zu hit pixels in the last reference frame)\n", world, sets, hits);
    return 0;
}
