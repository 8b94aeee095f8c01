/*
 * demo_sdf.c -- oracle restatement of the embedded demo SDF (cube minus sphere).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see sdf_oracle.h).
 * Follows /root/reference/src/sdf/demo/{mod,cube,sphere}.rs and src/sdf/{mod,defaults}.rs.
 */
#include "sdf_oracle.h"

#include <math.h>
#include <string.h>

void or_demo_default_params(OrDemoParams *p) {
    p->cube_half_side = 0.95f;               /* cube.rs:17 */
    p->cube_material = 0;                    /* cube.rs:15 "brick" */
    p->sphere_radius = 1.05f;                /* sphere.rs:13 */
    p->sphere_material = 1;                  /* sphere.rs:11 "normal" */
    p->max_distance_custom_material = 0.05f; /* demo/mod.rs:26 */
    p->disable_sphere = 0;                   /* demo/mod.rs:28 */
}

/* SDFSample::new, src/sdf/mod.rs:120-126 */
static OrSample sample_new(float distance, float r, float g, float b) {
    OrSample s;
    s.distance = distance;
    s.color[0] = r;
    s.color[1] = g;
    s.color[2] = b;
    s.metallic = 0.0f;
    s.roughness = 0.0f;
    s.occlusion = 0.0f;
    return s;
}

/* f32::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN */
static float signum_f32(float x) {
    if (isnan(x)) return x;
    return signbit(x) ? -1.0f : 1.0f;
}

/* compute_tex2d closure, cube.rs:189-202.  `%` on f32 is fmodf. */
static void brick_tex2d(float u, float v, float rgb[3], float *metallic, float *roughness, float *occlusion) {
    const float BRICK_WIDTH = 0.5f, BRICK_HEIGHT = 0.25f, CEMENT_THICKNESS = 0.2f;
    float row_num = v / BRICK_HEIGHT;
    float brick_offset = floorf(row_num) / 4.0f;
    float bx = fmodf(fabsf(u + brick_offset), BRICK_WIDTH);
    float by = fmodf(fabsf(v), BRICK_HEIGHT);
    float max_cement_displacement = CEMENT_THICKNESS / 2.0f * BRICK_HEIGHT;
    if (bx < max_cement_displacement || bx > BRICK_WIDTH - max_cement_displacement ||
        by < max_cement_displacement || by > BRICK_HEIGHT - max_cement_displacement) {
        rgb[0] = 56.0f / 255.0f; /* CEMENT_COLOR cube.rs:185 */
        rgb[1] = 70.0f / 255.0f;
        rgb[2] = 60.0f / 255.0f;
        *metallic = 0.4f;
        *roughness = 0.5f;
        *occlusion = 1.0f;
    } else {
        rgb[0] = 150.0f / 255.0f; /* BRICK_COLOR cube.rs:182 */
        rgb[1] = 24.0f / 255.0f;
        rgb[2] = 10.0f / 255.0f;
        *metallic = 0.2f;
        *roughness = 0.8f;
        *occlusion = 0.0f;
    }
}

/* sample_brick_texture, cube.rs:181-222: tri-planar pick by normal */
static OrSample sample_brick_texture(const float p[3], const float n[3], float distance) {
    OrSample s;
    float u, v;
    if (fabsf(n[0]) > fabsf(n[1])) {
        if (fabsf(n[0]) > fabsf(n[2])) { u = p[2]; v = p[1]; }
        else { u = p[0]; v = p[1]; }
    } else if (fabsf(n[1]) > fabsf(n[2])) { u = p[2]; v = p[0]; }
    else { u = p[0]; v = p[1]; }
    s.distance = distance;
    brick_tex2d(u, v, s.color, &s.metallic, &s.roughness, &s.occlusion);
    return s;
}

/* Material::render, cube.rs:51-58 */
static OrSample material_render(uint32_t material, float dist, const float p[3], const float n[3]) {
    if (material == 0) return sample_brick_texture(p, n, dist);
    return sample_new(dist, fabsf(n[0]), fabsf(n[1]), fabsf(n[2]));
}

/* SDFDemoCube::normal, cube.rs:164-177 */
static void cube_normal(const OrDemoParams *prm, const float p[3], float n[3]) {
    float side = prm->cube_half_side;
    n[0] = n[1] = n[2] = 0.0f;
    if (fabsf(p[0]) > side) n[0] = signum_f32(p[0]);
    if (fabsf(p[1]) > side) n[1] = signum_f32(p[1]);
    if (fabsf(p[2]) > side) n[2] = signum_f32(p[2]);
}

/* SDFDemoCube::sample, cube.rs:79-89 */
static OrSample cube_sample(const OrDemoParams *prm, const float p[3], int distance_only) {
    float dist_box = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2])) - prm->cube_half_side;
    distance_only = distance_only || dist_box > 0.1f;
    if (distance_only) return sample_new(dist_box, 0.0f, 0.0f, 0.0f);
    float n[3];
    cube_normal(prm, p, n);
    return material_render(prm->cube_material, dist_box, p, n);
}

/* cgmath 0.18 [EXT]: magnitude = sqrt(dot(v,v)), dot = (x*x + y*y) + z*z */
static float vec_len(const float v[3]) {
    return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

uint32_t or_ext_variant_flags = 0; /* [EXT] sensitivity switches, see sdf_oracle.h */
void or_set_ext_variant(uint32_t flags) { or_ext_variant_flags = flags; }
uint32_t or_get_ext_variant(void) { return or_ext_variant_flags; }

/* SDFDemoSphere::normal, sphere.rs:122-124; cgmath normalize = v * (1 / |v|) [EXT] */
static void sphere_normal(const float p[3], float n[3]) {
    if (or_ext_variant_flags & OR_EXT_CGMATH_NORMALIZE_DIV) {
        float l = vec_len(p);
        n[0] = p[0] / l; n[1] = p[1] / l; n[2] = p[2] / l;
        return;
    }
    float inv = 1.0f / vec_len(p);
    n[0] = p[0] * inv;
    n[1] = p[1] * inv;
    n[2] = p[2] * inv;
}

/* SDFDemoSphere::sample, sphere.rs:37-47.  p.distance(0) = |0 - p| */
static OrSample sphere_sample(const OrDemoParams *prm, const float p[3], int distance_only) {
    float q[3] = {0.0f - p[0], 0.0f - p[1], 0.0f - p[2]};
    float dist_sphere = vec_len(q) - prm->sphere_radius;
    distance_only = distance_only || dist_sphere > 0.1f;
    if (distance_only) return sample_new(dist_sphere, 0.0f, 0.0f, 0.0f);
    float n[3];
    sphere_normal(p, n);
    return material_render(prm->sphere_material, dist_sphere, p, n);
}

/* SDFDemo::sample, demo/mod.rs:51-75 */
static OrSample demo_sample(const OrDemoParams *prm, const float p[3], int distance_only) {
    OrSample sample_box = cube_sample(prm, p, distance_only);
    if (prm->disable_sphere) return sample_box;
    OrSample sample_sphere = sphere_sample(prm, p, distance_only);
    float dist = fmaxf(sample_box.distance, -sample_sphere.distance);
    float inter_surface_dist = fabsf(sample_box.distance) - fabsf(sample_sphere.distance);
    OrSample sample = inter_surface_dist < 0.0f ? sample_box : sample_sphere;
    if (fabsf(inter_surface_dist) <= prm->max_distance_custom_material) {
        sample.color[0] = 0.5f;
        sample.color[1] = 0.6f;
        sample.color[2] = 0.7f;
        sample.metallic = 0.5f;
        sample.roughness = 0.0f;
        sample.occlusion = 0.0f;
    }
    sample.distance = dist;
    return sample;
}

void or_sample(const OrDemoParams *prm, uint32_t sdf_id, const float p[3], int distance_only, OrSample *out) {
    switch (sdf_id) {
    case OR_SDF_DEMO: *out = demo_sample(prm, p, distance_only); break;
    case OR_SDF_CUBE: *out = cube_sample(prm, p, distance_only); break;
    case OR_SDF_SPHERE: *out = sphere_sample(prm, p, distance_only); break;
    default: *out = sample_new(0.0f, 0.0f, 0.0f, 0.0f); break; /* ffi.rs:61-64 unknown id */
    }
}

void or_normal_default(const OrDemoParams *prm, uint32_t sdf_id, const float p[3], float eps, float out[3]) {
    /* defaults.rs:49-56; Vector3 * scalar then + left to right, then normalize */
    if (!(eps > 0.0f)) eps = 0.001f;
    static const float k[4][3] = {{1.f, -1.f, -1.f}, {-1.f, 1.f, -1.f}, {-1.f, -1.f, 1.f}, {1.f, 1.f, 1.f}};
    float acc[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        float q[3] = {p[0] + k[i][0] * eps, p[1] + k[i][1] * eps, p[2] + k[i][2] * eps};
        OrSample s;
        or_sample(prm, sdf_id, q, 1, &s);
        for (int c = 0; c < 3; ++c) {
            float term = k[i][c] * s.distance;
            acc[c] = (i == 0) ? term : acc[c] + term;
        }
    }
    sphere_normal(acc, out); /* .normalize() */
}

void or_normal(const OrDemoParams *prm, uint32_t sdf_id, const float p[3], float eps, float out[3]) {
    (void)eps; /* the demo's overrides ignore eps */
    switch (sdf_id) {
    case OR_SDF_DEMO: { /* demo/mod.rs:147-156 */
        OrSample sb = cube_sample(prm, p, 1);
        OrSample ss = sphere_sample(prm, p, 1);
        if (fabsf(sb.distance) < fabsf(ss.distance)) {
            cube_normal(prm, p, out);
        } else {
            sphere_normal(p, out);
            out[0] = -out[0];
            out[1] = -out[1];
            out[2] = -out[2];
        }
        break;
    }
    case OR_SDF_CUBE: cube_normal(prm, p, out); break;
    case OR_SDF_SPHERE: sphere_normal(p, out); break;
    default: out[0] = out[1] = out[2] = 0.0f; break; /* ffi.rs:328-331 */
    }
}
