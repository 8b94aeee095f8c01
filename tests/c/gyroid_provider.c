/* gyroid_provider.c -- TEST FIXTURE: a host-only SDF behind the reference's per-point ABI (include/sdf_provider.h =
 * src/sdf/ffi.rs:42-337), the kind of library the ingest path of SDFViewer::update exists for.  Nothing here runs on the
 * GPU and nothing in the product knows this SDF.  Exports the two REQUIRED functions and their frees, parameters /
 * set_parameter / changed (one float parameter whose edit reports a sub-box), and deliberately NOT children, name, normal
 * (a consumer must fall back to the trait's defaults, src/sdf/wasm/native.rs:219-281,494-500).
 * gyroid_sample_raw is the same function with the oracle's callback signature (oracle/sdf_oracle.h or_sample_fn), so that
 * the oracle's update loop and the product's ingest path are fed the same samples.
 * Not the reference's code: a gyroid shell with a procedural colour, written for these tests. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "sdf_provider.h"

#define EXPORT __attribute__((visibility("default")))

static float g_thickness = 0.15f;
static int g_changed = 0;
static const SDFBoundingBox k_bounds = {{-1.0f, -0.5f, -0.75f}, {1.0f, 0.5f, 0.75f}};

static void gyroid(const float p[3], int distance_only, float out[7]) {
    const float k = 6.0f;
    const float x = p[0] * k, y = p[1] * k, z = p[2] * k;
    float g = sinf(x) * cosf(y) + sinf(y) * cosf(z) + sinf(z) * cosf(x);
    float d = fabsf(g) / k - g_thickness * 0.5f;
    d = d * 2.5f; /* stretches the range so that both clamps of 0.1 + d are hit */
    if (p[0] == k_bounds.min.x && p[1] == k_bounds.min.y && p[2] == k_bounds.min.z) d = NAN; /* f32::clamp keeps a NaN */
    memset(out, 0, 7 * sizeof(float));
    out[0] = d;
    if (distance_only) return;
    /* colour: bands that are exactly black (update() replaces an all-zero colour by 0.5 grey), values beyond [0, 1] */
    const float band = floorf((p[0] + 1.0f) * 4.0f);
    if (fmodf(band, 3.0f) != 0.0f) {
        out[1] = 0.5f + 0.6f * sinf(y);  /* < 0 and > 1 occur: Srgba::from saturates */
        out[2] = fabsf(cosf(z));
        out[3] = (p[2] > 0.0f) ? 1.0f : 0.25f;
    }
    out[4] = 0.5f + 0.5f * cosf(x);                 /* metallic */
    out[5] = p[1] > 0.0f ? 0.3f : 0.0f;             /* roughness */
    out[6] = p[0] > 0.25f ? 0.0f : (p[0] < -0.25f ? -1.0f : 0.6f); /* occlusion: <= 0 becomes 1 */
}

EXPORT void gyroid_sample_raw(void *user, const float p[3], int distance_only, float out[7]) {
    (void)user;
    gyroid(p, distance_only, out);
}

EXPORT SDFBoundingBox *bounding_box(uint32_t sdf_id) {
    SDFBoundingBox *ret = (SDFBoundingBox *)calloc(1, sizeof *ret);
    if (sdf_id == 0) *ret = k_bounds;
    return ret;
}
EXPORT void bounding_box_free(SDFBoundingBox *ret) { free(ret); }

EXPORT SDFSample *sample(uint32_t sdf_id, SDFVec3 p, bool distance_only) {
    SDFSample *ret = (SDFSample *)calloc(1, sizeof *ret);
    if (sdf_id == 0) {
        const float q[3] = {p.x, p.y, p.z};
        gyroid(q, distance_only, (float *)ret);
    }
    return ret;
}
EXPORT void sample_free(SDFSample *ret) { free(ret); }

/* extension (sdf_provider.h): batched sampling into the caller's array.  Built only with -DGYROID_BATCH: the suite loads the
 * fixture both ways, so that a consumer is checked with and without the export. */
#ifdef GYROID_BATCH
EXPORT void sample_batch(uint32_t sdf_id, const SDFVec3 *points, size_t n, bool distance_only, SDFSample *out) {
    for (size_t i = 0; i < n; ++i) {
        if (sdf_id == 0) {
            const float q[3] = {points[i].x, points[i].y, points[i].z};
            gyroid(q, distance_only, (float *)&out[i]);
        } else {
            memset(&out[i], 0, sizeof out[i]);
        }
    }
}
#endif

static PointerLength pl_copy(const void *data, size_t n) {
    PointerLength p = {NULL, n};
    if (n) {
        void *m = malloc(n);
        memcpy(m, data, n);
        p.ptr = m;
    }
    return p;
}

EXPORT PointerLength *parameters(uint32_t sdf_id) {
    PointerLength *ret = (PointerLength *)calloc(1, sizeof *ret);
    if (sdf_id != 0) return ret;
    SDFParamC prm;
    memset(&prm, 0, sizeof prm);
    prm.id = 0;
    prm.name = pl_copy("thickness", 9);
    prm.kind.tag = 2;
    prm.kind.v.float_.range_start = 0.0f;
    prm.kind.v.float_.range_end = 1.0f;
    prm.kind.v.float_.step = 0.01f;
    prm.value.tag = 2;
    prm.value.v.float_ = g_thickness;
    prm.description = pl_copy("shell thickness", 15);
    *ret = pl_copy(&prm, sizeof prm);
    return ret;
}
EXPORT void parameters_free(PointerLength *ret) {
    if (!ret) return;
    const SDFParamC *prm = (const SDFParamC *)ret->ptr;
    for (size_t i = 0; i < ret->len_bytes / sizeof(SDFParamC); ++i) {
        free((void *)prm[i].name.ptr);
        free((void *)prm[i].description.ptr);
    }
    free((void *)ret->ptr);
    free(ret);
}

EXPORT SDFSetParameterResult *set_parameter(uint32_t sdf_id, uint32_t param_id, SDFParamValueC value) {
    SDFSetParameterResult *ret = (SDFSetParameterResult *)calloc(1, sizeof *ret);
    if (sdf_id == 0 && param_id == 0 && value.tag == 2) {
        g_thickness = value.v.float_;
        g_changed = 1;
        return ret;
    }
    ret->tag = 1;
    ret->error = pl_copy("unknown parameter", 17);
    return ret;
}
EXPORT void set_parameter_free(SDFSetParameterResult *ret) {
    if (!ret) return;
    free((void *)ret->error.ptr);
    free(ret);
}

/* an edit reports the lower-x part of the box once (a box that does NOT cover the grid: update_required mixes both tests) */
EXPORT SDFChangedResult *changed(uint32_t sdf_id) {
    SDFChangedResult *ret = (SDFChangedResult *)calloc(1, sizeof *ret);
    if (sdf_id == 0 && g_changed) {
        g_changed = 0;
        ret->tag = 1;
        ret->bounds = k_bounds;
        ret->bounds.max.x = 0.1f;
        ret->bounds.min.y = -0.3f;
    }
    return ret;
}
EXPORT void changed_free(SDFChangedResult *ret) { free(ret); }

/* extension (sdf_provider.h): sample() is a pure function of its arguments and g_thickness -- any number of threads */
EXPORT uint32_t sample_concurrency(void) { return 64; }
