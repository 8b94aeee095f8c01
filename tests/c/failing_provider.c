/* failing_provider.c -- TEST FIXTURE: a provider library whose every call goes wrong in one of the ways the reference's host
 * guards against (src/sdf/wasm/native.rs:164-521: a failed call is logged and mapped to the trait's default).  With
 * fail_mode() == 0 every call returns NULL; with 1 the calls return well-formed blocks that carry values a consumer must
 * refuse: unknown enum tags, a child list that names the SDF itself, byte lengths that are not a whole number of records,
 * NULL data pointers with a non-zero length.  No *_free is exported (they are optional, ffi.rs:52-55): nothing is freed.
 * Not the reference's code. */
#include <stdlib.h>
#include <string.h>

#include "sdf_provider.h"

#define EXPORT __attribute__((visibility("default")))

static int g_mode = 0;
EXPORT void set_fail_mode(int mode) { g_mode = mode; }

EXPORT SDFBoundingBox *bounding_box(uint32_t sdf_id) {
    (void)sdf_id;
    return NULL;
}

EXPORT SDFSample *sample(uint32_t sdf_id, SDFVec3 p, bool distance_only) {
    (void)sdf_id, (void)p, (void)distance_only;
    return NULL;
}

EXPORT PointerLength *children(uint32_t sdf_id) {
    if (g_mode == 0) return NULL;
    PointerLength *ret = (PointerLength *)calloc(1, sizeof *ret);
    if (g_mode == 2) { /* a length without data */
        ret->len_bytes = 12;
        return ret;
    }
    uint32_t *ids = (uint32_t *)calloc(3, sizeof *ids);
    ids[0] = sdf_id; /* "Children of SDF with ID .. include itself! Skipping" (native.rs:241-244) */
    ids[1] = sdf_id + 5;
    ids[2] = 0xdeadbeefu; /* cut off by the ragged length below */
    ret->ptr = ids;
    ret->len_bytes = 2 * sizeof *ids + 3; /* two whole ids and three stray bytes */
    return ret;
}

EXPORT PointerLength *name(uint32_t sdf_id) {
    (void)sdf_id;
    if (g_mode == 0) return NULL;
    PointerLength *ret = (PointerLength *)calloc(1, sizeof *ret);
    ret->len_bytes = g_mode == 2 ? 9 : 0; /* NULL data either way */
    return ret;
}

EXPORT PointerLength *parameters(uint32_t sdf_id) {
    (void)sdf_id;
    if (g_mode == 0) return NULL;
    PointerLength *ret = (PointerLength *)calloc(1, sizeof *ret);
    if (g_mode == 2) {
        ret->len_bytes = 5 * sizeof(SDFParamC);
        return ret;
    }
    SDFParamC *prm = (SDFParamC *)calloc(4, sizeof *prm);
    /* 0: unknown kind tag -> dropped; 1: unknown value tag -> dropped; 2: a good one; 3: cut off by the ragged length */
    prm[0].id = 10;
    prm[0].kind.tag = 9;
    prm[0].value.tag = 0;
    prm[1].id = 11;
    prm[1].kind.tag = 0;
    prm[1].value.tag = 77;
    prm[2].id = 12;
    prm[2].name.ptr = "ok";
    prm[2].name.len_bytes = 2;
    prm[2].kind.tag = 1;
    prm[2].kind.v.int_.range_start = -3;
    prm[2].kind.v.int_.range_end = 3;
    prm[2].kind.v.int_.step = 1;
    prm[2].value.tag = 1;
    prm[2].value.v.int_ = 2;
    prm[3].id = 13;
    ret->ptr = prm;
    ret->len_bytes = 3 * sizeof *prm + sizeof *prm / 2;
    return ret;
}

EXPORT SDFSetParameterResult *set_parameter(uint32_t sdf_id, uint32_t param_id, SDFParamValueC value) {
    (void)sdf_id, (void)param_id, (void)value;
    if (g_mode == 0) return NULL;
    SDFSetParameterResult *ret = (SDFSetParameterResult *)calloc(1, sizeof *ret);
    ret->tag = g_mode == 2 ? 1 : 5; /* 2: an error without a message; 1: an unknown result kind */
    return ret;
}

EXPORT SDFChangedResult *changed(uint32_t sdf_id) {
    (void)sdf_id;
    if (g_mode == 0) return NULL;
    SDFChangedResult *ret = (SDFChangedResult *)calloc(1, sizeof *ret);
    ret->tag = 3; /* neither "no change" nor "changed" */
    ret->bounds.max.x = 1.0f;
    return ret;
}

/* what a consumer samples instead (native.rs:203: SDFSample::new(1.0, 0)), with the oracle's callback signature (oracle/sdf_oracle.h) */
EXPORT void failing_sample_raw(void *user, const float p[3], int distance_only, float out[7]) {
    (void)user, (void)p, (void)distance_only;
    memset(out, 0, 7 * sizeof(float));
    out[0] = 1.0f;
}

EXPORT SDFVec3 *normal(uint32_t sdf_id, SDFVec3 p, float eps) {
    (void)sdf_id, (void)p, (void)eps;
    return NULL;
}

EXPORT uint32_t sample_concurrency(void) { return 0; /* "no answer": a consumer must still use one thread */ }
