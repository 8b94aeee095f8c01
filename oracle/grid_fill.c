/*
 * grid_fill.c -- oracle restatement of the grid fill controller and the progressive-LOD iterator.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see sdf_oracle.h).
 * Follows /root/reference/src/app/scene/sdf/mod.rs and src/app/scene/sdf/loading.rs.
 */
#include "sdf_oracle.h"

#include <math.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* scene/sdf/mod.rs:42  const AIR_DIST: f32 = 1e-1 + 0.001234; (f32 const-evaluated) */
float or_air_dist(void) {
    volatile float a = 1e-1f, b = 0.001234f;
    return a + b;
}

/* scene/sdf/mod.rs:46-72.  max_by keeps the LAST maximal element on ties. */
void or_grid_dims_from_bb(const float bb_min[3], const float bb_max[3], uint32_t max_voxels_side, uint32_t dims[3]) {
    float size[3] = {bb_max[0] - bb_min[0], bb_max[1] - bb_min[1], bb_max[2] - bb_min[2]};
    int max_dim = 0;
    for (int i = 1; i < 3; ++i)
        if (size[i] >= size[max_dim]) max_dim = i;
    for (int i = 0; i < 3; ++i) {
        if (i == max_dim) {
            dims[i] = max_voxels_side;
        } else {
            float v = (float)max_voxels_side * size[i] / size[max_dim]; /* (a*b)/c, then `as usize` */
            dims[i] = v > 0.0f ? (uint32_t)v : 0u;
        }
    }
}

/* scene/sdf/mod.rs:178-182: three separately rounded steps per axis */
float or_voxel_coord(uint32_t idx, uint32_t dim, float bb_min, float bb_max) {
    float size = bb_max - bb_min;
    float dim_minus_1 = (float)dim - 1.0f; /* :168 `width as f32 - 1.` */
    float pos = (float)idx;
    pos = pos / dim_minus_1;
    pos = pos * size;
    pos = pos + bb_min;
    return pos;
}

/* three-d-asset 0.9.2 [EXT]: impl From<Vec3> for Srgba -> (c * 255.0) as u8 (truncating, saturating,
 * NaN -> 0) */
extern uint32_t or_ext_variant_flags;
uint8_t or_srgb_quantize(float c) {
    float v = c * 255.0f;
    if (or_ext_variant_flags & OR_EXT_SRGB_QUANT_ROUND) v = v + 0.5f;
    if (!(v > 0.0f)) return 0; /* negatives, -0, NaN */
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

/* three-d-asset 0.9.2 [EXT]: Srgba::to_linear_srgb, per channel */
float or_srgb_u8_to_linear(uint8_t c8) {
    float c = (float)c8 / 255.0f;
    if (c < 0.04045f) return c / 12.92f;
    if (or_ext_variant_flags & OR_EXT_SRGB_DOUBLE_POW) return (float)pow(((double)c + 0.055) / 1.055, 2.4);
    float r = powf((c + 0.055f) / 1.055f, 2.4f);
    if (or_ext_variant_flags & OR_EXT_SRGB_POW_ULP_UP) r = nextafterf(r, 2.0f);
    if (or_ext_variant_flags & OR_EXT_SRGB_POW_ULP_DOWN) r = nextafterf(r, -1.0f);
    return r;
}

/* scene/sdf/mod.rs:196-208 */
void or_pack_sample(const OrSample *s_in, float tex0[4], float tex1[4]) {
    OrSample s = *s_in;
    float d = 1e-1f + s.distance;
    /* f32::clamp(0.0, 1.0) */
    if (d < 0.0f) d = 0.0f;
    if (d > 1.0f) d = 1.0f;
    tex0[0] = d;
    if (s.color[0] == 0.0f && s.color[1] == 0.0f && s.color[2] == 0.0f) {
        s.color[0] = s.color[1] = s.color[2] = 0.5f;
    }
    tex0[1] = or_srgb_u8_to_linear(or_srgb_quantize(s.color[0]));
    tex0[2] = or_srgb_u8_to_linear(or_srgb_quantize(s.color[1]));
    tex0[3] = or_srgb_u8_to_linear(or_srgb_quantize(s.color[2]));
    tex1[0] = s.metallic;
    tex1[1] = s.roughness;
    tex1[2] = s.occlusion <= 0.0f ? 1.0f : s.occlusion;
    /* tex1[3] is never written by update(): it keeps new_voxels' AIR_DIST */
}

void or_grid_init(float *tex0, float *tex1, size_t n_voxels) {
    float air = or_air_dist();
    for (size_t i = 0; i < n_voxels * 4; ++i) {
        tex0[i] = air;
        tex1[i] = air;
    }
}

static void fill_voxel(const OrDemoParams *prm, uint32_t sdf_id, const uint32_t dims[3],
                       const float bb_min[3], const float bb_max[3],
                       uint32_t x, uint32_t y, uint32_t z, float *t0, float *t1) {
    float pos[3] = {or_voxel_coord(x, dims[0], bb_min[0], bb_max[0]),
                    or_voxel_coord(y, dims[1], bb_min[1], bb_max[1]),
                    or_voxel_coord(z, dims[2], bb_min[2], bb_max[2])};
    OrSample s;
    or_sample(prm, sdf_id, pos, 0, &s);
    or_pack_sample(&s, t0, t1);
}

void or_fill_dense(const OrDemoParams *prm, uint32_t sdf_id, const uint32_t dims[3],
                   const float bb_min[3], const float bb_max[3],
                   uint32_t z0, uint32_t z1, float *tex0, float *tex1, int n_threads) {
    const size_t W = dims[0], H = dims[1];
    const float air = or_air_dist();
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t z = z0; z < (int64_t)z1; ++z) {
        for (size_t y = 0; y < H; ++y) {
            for (size_t x = 0; x < W; ++x) {
                size_t flat = (((size_t)z - z0) * H + y) * W + x; /* :177 row-major, x fastest */
                float *t0 = tex0 + flat * 4, *t1 = tex1 + flat * 4;
                t1[3] = air; /* new_voxels init value survives */
                fill_voxel(prm, sdf_id, dims, bb_min, bb_max, (uint32_t)x, (uint32_t)y, (uint32_t)z, t0, t1);
            }
        }
    }
}

/* ---------------- LoadingManager, loading.rs ---------------- */

/* loading.rs:108-115 */
uint32_t or_prev_power_of_2(uint32_t x) {
    x = x | (x >> 1);
    x = x | (x >> 2);
    x = x | (x >> 4);
    x = x | (x >> 8);
    x = x | (x >> 16);
    return x - (x >> 1);
}

/* loading.rs:23-44 (new + reset) */
void or_lm_new(OrLoadingManager *m, const uint64_t limits[3], uint64_t passes) {
    memcpy(m->limits, limits, sizeof(m->limits));
    m->passes = passes;
    uint32_t p = (uint32_t)passes;
    if (p < 1) p = 1;
    m->step_size = (uint64_t)1 << (p - 1); /* 2usize.pow(max(passes,1) - 1) */
    m->next_index[0] = m->next_index[1] = m->next_index[2] = 0;
    m->iterations = 0;
    m->total_iterations = 0;
}

/* loading.rs:50-76 */
int or_lm_next(OrLoadingManager *m, uint64_t out_index[3]) {
    if (m->step_size == 0) return 0;
    m->iterations += 1;
    m->total_iterations += 1;
    out_index[0] = m->next_index[0];
    out_index[1] = m->next_index[1];
    out_index[2] = m->next_index[2];
    m->next_index[0] += m->step_size;
    if (m->next_index[0] >= m->limits[0]) {
        m->next_index[0] = 0;
        m->next_index[1] += m->step_size;
        if (m->next_index[1] >= m->limits[1]) {
            m->next_index[1] = 0;
            m->next_index[2] += m->step_size;
            if (m->next_index[2] >= m->limits[2]) {
                m->step_size = or_prev_power_of_2((uint32_t)(m->step_size - 1));
                m->next_index[0] = m->next_index[1] = m->next_index[2] = 0;
                m->iterations = 0;
            }
        }
    }
    return 1;
}

/* loading.rs:80-89 */
uint64_t or_lm_len(const OrLoadingManager *m) {
    uint64_t step = m->step_size, iterations = 0;
    while (step > 0) {
        uint64_t sx = (m->limits[0] + step - 1) / step;
        uint64_t sy = (m->limits[1] + step - 1) / step;
        uint64_t sz = (m->limits[2] + step - 1) / step;
        iterations += sx * sy * sz;
        step = or_prev_power_of_2((uint32_t)(step - 1));
    }
    return iterations - m->iterations;
}

/* loading.rs:99-105 */
uint64_t or_lm_passes_left(const OrLoadingManager *m) {
    if (m->step_size == 0) return 0;
    return (uint64_t)log2f((float)m->step_size) + 1;
}

/* scene/sdf/mod.rs:173-215 with `sdf` ANY SDFSurface: sample() is the caller's function */
uint64_t or_viewer_update_fn(or_sample_fn sample, void *user, const uint32_t dims[3],
                             const float bb_min[3], const float bb_max[3],
                             OrLoadingManager *lm, const float *cb, uint64_t max_iterations,
                             float *tex0, float *tex1) {
    const float air = or_air_dist();
    uint64_t start = lm->total_iterations;
    while (lm->total_iterations - start < max_iterations) {
        uint64_t idx[3];
        if (!or_lm_next(lm, idx)) break;
        size_t flat = ((size_t)idx[2] * dims[1] + idx[1]) * dims[0] + idx[0];
        float pos[3] = {or_voxel_coord((uint32_t)idx[0], dims[0], bb_min[0], bb_max[0]),
                        or_voxel_coord((uint32_t)idx[1], dims[1], bb_min[1], bb_max[1]),
                        or_voxel_coord((uint32_t)idx[2], dims[2], bb_min[2], bb_max[2])};
        int update_required = tex0[flat * 4] == air;
        if (cb) {
            update_required = update_required ||
                              (pos[0] >= cb[0] && pos[0] <= cb[3] && pos[1] >= cb[1] && pos[1] <= cb[4] &&
                               pos[2] >= cb[2] && pos[2] <= cb[5]);
        }
        if (update_required) {
            OrSample s;
            sample(user, pos, 0, &s);
            or_pack_sample(&s, tex0 + flat * 4, tex1 + flat * 4);
        }
    }
    return lm->total_iterations - start;
}

/* ... with `sdf` the demo */
typedef struct {
    const OrDemoParams *prm;
    uint32_t sdf_id;
} DemoSdf;
static void demo_sample_fn(void *user, const float p[3], int distance_only, OrSample *out) {
    const DemoSdf *d = (const DemoSdf *)user;
    or_sample(d->prm, d->sdf_id, p, distance_only, out);
}
uint64_t or_viewer_update(const OrDemoParams *prm, uint32_t sdf_id, const uint32_t dims[3],
                          const float bb_min[3], const float bb_max[3],
                          OrLoadingManager *lm, const float *cb, uint64_t max_iterations,
                          float *tex0, float *tex1) {
    DemoSdf d = {prm, sdf_id};
    return or_viewer_update_fn(demo_sample_fn, &d, dims, bb_min, bb_max, lm, cb, max_iterations, tex0, tex1);
}
