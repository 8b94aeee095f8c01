/* gpu_roundtrip.c -- the hot path driven from plain C with nothing but the HIP runtime API for memory: what a host in
 * any language with a C FFI does.  hipMalloc -> sdfv_fill_grid_commit -> sdfv_raymarch_accel -> hipMemcpy back; the raw
 * results go to <prefix>.tex0.f32 / .tex1.f32 / .rgba.f32 for tests/test_gpu_host.py to compare with the oracle.
 * Then the reference's default 2-pass progressive load as its LoadingManager would drive it (sdfv_grid_init, two
 * sdfv_fill_grid_pass_ex with the flags a host knows) into a second pair of textures, the y-pair volume of a viewer
 * (sdfv_commit_pairs) and a frame over it (sdfv_raymarch_pairs): <prefix>.p_tex0.f32 / .p_tex1.f32 / .p_rgba.f32.
 * Last, what a multi-GPU host does per rank and a host of a large grid per load: the same frame as the two band sets of a
 * world of 2 (sdfv_raymarch_bands over the volume sdfv_march_volume_advice names: <prefix>.b0_rgba.f32, .b1_rgba.f32) and
 * over the y-interleaved volume (sdfv_commit_interleaved, sdfv_raymarch_volumes: <prefix>.i_rgba.f32). */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "sdfgrid.h"

#define DIE(msg)                                                        \
    do {                                                                \
        fprintf(stderr, "%s: %s\n", msg, sdfv_last_error());            \
        return 1;                                                       \
    } while (0)

static int dump(const char *prefix, const char *suffix, const void *dev, size_t bytes) {
    char path[1024];
    void *host = malloc(bytes);
    FILE *f;
    if (!host || hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    snprintf(path, sizeof path, "%s.%s", prefix, suffix);
    f = fopen(path, "wb");
    if (!f || fwrite(host, 1, bytes, f) != bytes) return 1;
    fclose(f);
    free(host);
    return 0;
}

int main(int argc, char **argv) {
    const char *prefix = argc > 1 ? argv[1] : "roundtrip";
    const uint32_t W = 64, H = 48;
    const float lo[3] = {-1.0f, -1.0f, -1.0f}, hi[3] = {1.0f, 1.0f, 1.0f};
    const float eye[3] = {2.5f, 3.0f, 5.0f}, target[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};
    sdfv_demo_params prm;
    sdfv_grid grid;
    sdfv_render_params rp;
    sdfv_camera cam;
    float *tex0 = NULL, *tex1 = NULL, *dist = NULL, *rgba = NULL;
    size_t voxels, tex_bytes;

    if (sdfv_device_count() < 1) DIE("no device");
    sdfv_demo_params_default(&prm);
    if (sdfv_grid_from_bb(lo, hi, 32, &grid) != SDFV_OK) DIE("grid_from_bb");
    voxels = (size_t)grid.dims[0] * grid.dims[1] * grid.dims[2];
    tex_bytes = voxels * 16;
    if (hipMalloc((void **)&tex0, tex_bytes) != hipSuccess || hipMalloc((void **)&tex1, tex_bytes) != hipSuccess ||
        hipMalloc((void **)&dist, voxels * 4) != hipSuccess || hipMalloc((void **)&rgba, (size_t)W * H * 16) != hipSuccess)
        return 1;
    if (sdfv_fill_grid_commit(&prm, SDFV_SDF_DEMO, &grid, tex0, tex1, dist, NULL) != SDFV_OK) DIE("fill_grid_commit");
    sdfv_render_params_default(&rp, &grid);
    if (sdfv_camera_look_at(&cam, eye, target, up, 45.0f, (float)W / (float)H, 0.1f, 1000.0f) != SDFV_OK) DIE("camera");
    if (sdfv_raymarch_accel(&rp, tex0, tex1, dist, &cam, 1, W, H, 0, H, rgba, NULL, NULL) != SDFV_OK) DIE("raymarch");
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (dump(prefix, "tex0.f32", tex0, tex_bytes) || dump(prefix, "tex1.f32", tex1, tex_bytes) ||
        dump(prefix, "rgba.f32", rgba, (size_t)W * H * 16))
        return 1;
    {
        float *p0 = NULL, *p1 = NULL, *pd = NULL, *pairs = NULL;
        if (hipMalloc((void **)&p0, tex_bytes) != hipSuccess || hipMalloc((void **)&p1, tex_bytes) != hipSuccess ||
            hipMalloc((void **)&pd, voxels * 4) != hipSuccess || hipMalloc((void **)&pairs, voxels * 8) != hipSuccess)
            return 1;
        if (sdfv_grid_init(&grid, p0, p1, NULL) != SDFV_OK) DIE("grid_init");
        if (sdfv_commit_distance(&grid, p0, pd, NULL) != SDFV_OK) DIE("commit_distance");
        if (sdfv_fill_grid_pass_ex(&prm, SDFV_SDF_DEMO, &grid, 2, NULL, p0, p1, pd, SDFV_PASS_FRESH_GRID | SDFV_PASS_SAME_LOAD, NULL) != SDFV_OK)
            DIE("pass step 2");
        if (sdfv_fill_grid_pass_ex(&prm, SDFV_SDF_DEMO, &grid, 1, NULL, p0, p1, pd, SDFV_PASS_SAME_LOAD, NULL) != SDFV_OK)
            DIE("pass step 1");
        if (sdfv_commit_pairs(&grid, pd, pairs, NULL) != SDFV_OK) DIE("commit_pairs");
        if (sdfv_raymarch_pairs(&rp, p0, p1, pd, pairs, &cam, 1, W, H, 0, H, rgba, NULL, NULL, NULL) != SDFV_OK) DIE("raymarch_pairs");
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        if (dump(prefix, "p_tex0.f32", p0, tex_bytes) || dump(prefix, "p_tex1.f32", p1, tex_bytes) ||
            dump(prefix, "p_rgba.f32", rgba, (size_t)W * H * 16))
            return 1;
        {
            uint32_t kind = 99, r;
            float *ilv = NULL, *band = NULL;
            if (sdfv_march_volume_advice(&grid, &kind) != SDFV_OK || kind != SDFV_MARCH_VOLUME_PAIRS) DIE("march_volume_advice");
            for (r = 0; r < 2; ++r) {  /* rank r of 2: tile bands r, r + 2, ... into a compact buffer */
                const uint32_t rows = sdfv_band_rows(H, r, 2);
                char name[32];
                if (rows != (r == 0 ? 32u : 16u)) DIE("band_rows");  /* H = 48: bands 0, 2 | band 1 */
                if (hipMalloc((void **)&band, (size_t)rows * W * 16) != hipSuccess) return 1;
                if (sdfv_raymarch_bands(&rp, p0, p1, pd, pairs, NULL, &cam, 1, W, H, r, 2, band, NULL, NULL, NULL) != SDFV_OK) DIE("raymarch_bands");
                if (hipDeviceSynchronize() != hipSuccess) return 1;
                snprintf(name, sizeof name, "b%u_rgba.f32", r);
                if (dump(prefix, name, band, (size_t)rows * W * 16)) return 1;
                hipFree(band);
            }
            if (hipMalloc((void **)&ilv, voxels * 4) != hipSuccess) return 1;
            if (sdfv_commit_interleaved(&grid, pd, ilv, NULL) != SDFV_OK) DIE("commit_interleaved");
            if (sdfv_raymarch_volumes(&rp, p0, p1, pd, NULL, ilv, &cam, 1, W, H, 0, H, rgba, NULL, NULL, NULL) != SDFV_OK) DIE("raymarch_volumes");
            if (hipDeviceSynchronize() != hipSuccess) return 1;
            if (dump(prefix, "i_rgba.f32", rgba, (size_t)W * H * 16)) return 1;
            hipFree(ilv);
        }
        hipFree(p0);
        hipFree(p1);
        hipFree(pd);
        hipFree(pairs);
    }
    hipFree(tex0);
    hipFree(tex1);
    hipFree(dist);
    hipFree(rgba);
    printf("ok %ux%ux%u\n", grid.dims[0], grid.dims[1], grid.dims[2]);
    return 0;
}
