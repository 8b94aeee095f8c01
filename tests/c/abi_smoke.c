/* abi_smoke.c -- the two public headers must be usable from plain C (C99, -pedantic) and the library must link and
 * answer without a GPU.  Built and run by tests/test_abi.py; prints "ok" or the first failed check. */
#include <stdio.h>
#include <string.h>

#include "sdf_provider.h"
#include "sdfgrid.h"

#define CHECK(cond)                                           \
    do {                                                      \
        if (!(cond)) {                                        \
            printf("FAILED line %d: %s\n", __LINE__, #cond);  \
            return 1;                                         \
        }                                                     \
    } while (0)

int main(void) {
    sdfv_demo_params p;
    sdfv_grid g;
    sdfv_render_params rp;
    sdfv_camera cam;
    sdfv_mesh mesh;
    const float lo[3] = {-1.0f, -1.0f, -1.0f}, hi[3] = {1.0f, 1.0f, 1.0f};
    const float eye[3] = {2.5f, 3.0f, 5.0f}, target[3] = {0.0f, 0.0f, 0.0f}, up[3] = {0.0f, 1.0f, 0.0f};

    CHECK(sdfv_abi_version() == SDFV_ABI_VERSION);
    CHECK(sizeof(sdfv_sample) == 28 && sizeof(sdfv_demo_params) == 24 && sizeof(sdfv_vertex) == 48);
    CHECK(sizeof(sdfv_ray_state) == 24 && sizeof(sdfv_march_aux) == 72);
    sdfv_demo_params_default(&p);
    CHECK(p.cube_half_side == 0.95f && p.sphere_radius == 1.05f && p.max_distance_custom_material == 0.05f);
    CHECK(sdfv_grid_from_bb(lo, hi, 64, &g) == SDFV_OK && g.dims[0] == 64 && g.dims[1] == 64 && g.dims[2] == 64);
    sdfv_render_params_default(&rp, &g);
    CHECK(rp.tex_size[2] == 64 && rp.lod_dist_between_samples == 1.0f);
    CHECK(sdfv_camera_look_at(&cam, eye, target, up, 45.0f, 16.0f / 9.0f, 0.1f, 1000.0f) == SDFV_OK);
    CHECK(sdfv_air_dist() > 0.1012f && sdfv_air_dist() < 0.1013f);
    /* argument errors are status codes with a message, never a crash */
    CHECK(sdfv_fill_grid(NULL, 0, &g, NULL, NULL, NULL) == SDFV_ERR_INVALID_ARGUMENT);
    CHECK(strlen(sdfv_last_error()) > 0);
    CHECK(sdfv_mesh_extract(&p, 0, lo, hi, 8, 3, &mesh, NULL) == SDFV_ERR_INVALID_ARGUMENT);
    CHECK(strstr(sdfv_last_error(), "Unsupported algorithm") != NULL);
    CHECK(sdfv_mesh_free(&mesh) == SDFV_OK && sdfv_mesh_trim() == SDFV_OK);
    CHECK(sdfv_slab_comm_destroy(NULL) == SDFV_OK);
    if (sdfv_device_count() == 0) {
        float t[4 * 4 * 4 * 4];
        sdfv_grid small = g;
        small.dims[0] = small.dims[1] = small.dims[2] = 4;
        small.z_end = 4;
        memset(t, 0, sizeof t);
        CHECK(sdfv_fill_grid_host(&p, 0, &small, t, t) == SDFV_ERR_NO_DEVICE); /* no CPU fallback */
        CHECK(t[0] == 0.0f);
    }
    printf("ok\n");
    return 0;
}
