/*
 * raymarch.c -- oracle restatement of the per-pixel sphere-tracing fragment shader.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see sdf_oracle.h).
 * Follows /root/reference/src/app/scene/sdf/material.frag (whole file), material.rs:50-97,
 * src/app/scene/mod.rs:82-112 and src/app/scene/sdf/mod.rs:103-117,254-282.
 *
 * Restatement choices where the reference defers to GL / three-d (all [EXT], isolated below):
 *  - `pos` (the rasterised bbox fragment) = ray/AABB slab test through the pixel centre; the front
 *    face is used when the camera is outside the box, the back face when inside (Cull::None,
 *    material.rs:78; the other face's fragment is transparent with depth 1.0 and changes nothing).
 *  - texture() LINEAR = GL texel-centre convention, MirroredRepeat, full fp32 mix() x then y then z.
 *  - normalize(v) = v / length(v); mix(a,b,t) = a*(1-t) + b*t; round() = half away from zero.
 */
#include "sdf_oracle.h"

#include <math.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 v3_from(const float *p) { return v3_make(p[0], p[1], p[2]); }
static v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 v3_madd(v3 a, v3 d, float t) { return v3_make(a.x + d.x * t, a.y + d.y * t, a.z + d.z * t); }
static float v3_len(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
extern uint32_t or_ext_variant_flags; /* [EXT] sensitivity switches (sdf_oracle.h); 0 in every parity test */
static v3 v3_normalize(v3 a) {
    float l = v3_len(a);
    if (or_ext_variant_flags & OR_EXT_GLSL_NORMALIZE_RSQ) {
        float r = 1.0f / l;
        return v3_make(a.x * r, a.y * r, a.z * r);
    }
    return v3_make(a.x / l, a.y / l, a.z / l);
}
static float mixf(float a, float b, float t) {
    if (or_ext_variant_flags & OR_EXT_GLSL_MIX_LERP) return a + t * (b - a);
    return a * (1.0f - t) + b * t;
}

/* Texel-touch recording for or_raymarch_touch: the map the current phase of the current thread marks, or NULL. */
static _Thread_local uint8_t *tl_touch_map = NULL;
static _Thread_local const float *tl_touch_tex = NULL;

void or_default_render_params(OrRenderParams *rp, const uint32_t dims[3], const float bb_min[3], const float bb_max[3]) {
    memset(rp, 0, sizeof(*rp));
    for (int i = 0; i < 3; ++i) {
        rp->bounds_min[i] = bb_min[i];
        rp->bounds_max[i] = bb_max[i];
        rp->tex_size[i] = dims[i];
        rp->ambient[i] = 1.0f; /* AmbientLight::new(ctx, 1.0, Srgba::WHITE), scene/mod.rs:106 */
    }
    rp->lod_dist_between_samples = 1.0f; /* material.rs:28 */
    rp->tint[0] = rp->tint[1] = rp->tint[2] = rp->tint[3] = 1.0f; /* Srgba::WHITE, material.rs:29 */
    rp->gamma = 0.0f;
    rp->tone_mapping = 2;  /* three-d 0.18 default: Aces [EXT] */
    rp->color_mapping = 1; /* three-d 0.18 default: ComputeToSrgb [EXT] */
}

/* cgmath look_at_rh + perspective [EXT], three-d Camera::new_perspective; scene/mod.rs:82-95 */
void or_camera_look_at(OrCamera *cam, const float eye[3], const float target[3], const float up[3],
                       float fovy_degrees, float aspect, float z_near, float z_far) {
    v3 e = v3_from(eye), t = v3_from(target), u0 = v3_from(up);
    v3 f = v3_sub(t, e);
    float fl = v3_len(f);
    f = v3_make(f.x * (1.0f / fl), f.y * (1.0f / fl), f.z * (1.0f / fl));
    v3 s = v3_make(f.y * u0.z - f.z * u0.y, f.z * u0.x - f.x * u0.z, f.x * u0.y - f.y * u0.x);
    float sl = v3_len(s);
    s = v3_make(s.x * (1.0f / sl), s.y * (1.0f / sl), s.z * (1.0f / sl));
    v3 u = v3_make(s.y * f.z - s.z * f.y, s.z * f.x - s.x * f.z, s.x * f.y - s.y * f.x);
    cam->eye[0] = e.x; cam->eye[1] = e.y; cam->eye[2] = e.z;
    cam->right[0] = s.x; cam->right[1] = s.y; cam->right[2] = s.z;
    cam->up[0] = u.x; cam->up[1] = u.y; cam->up[2] = u.z;
    cam->forward[0] = f.x; cam->forward[1] = f.y; cam->forward[2] = f.z;
    float half = fovy_degrees * (3.14159265358979323846f / 180.0f) / 2.0f;
    cam->tan_half_fovy = tanf(half);
    cam->aspect = aspect;
    /* view (column-major) */
    float view[16] = {s.x, u.x, -f.x, 0.0f, s.y, u.y, -f.y, 0.0f, s.z, u.z, -f.z, 0.0f,
                      -(e.x * s.x + e.y * s.y + e.z * s.z), -(e.x * u.x + e.y * u.y + e.z * u.z),
                      (e.x * f.x + e.y * f.y + e.z * f.z), 1.0f};
    float ct = 1.0f / cam->tan_half_fovy;
    float proj[16] = {ct / aspect, 0, 0, 0, 0, ct, 0, 0,
                      0, 0, (z_far + z_near) / (z_near - z_far), -1.0f,
                      0, 0, (2.0f * z_far * z_near) / (z_near - z_far), 0};
    static const float bias[16] = {0.5f, 0, 0, 0, 0, 0.5f, 0, 0, 0, 0, 0.5f, 0, 0.5f, 0.5f, 0.5f, 1.0f};
    float pv[16];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += proj[k * 4 + r] * view[c * 4 + k];
            pv[c * 4 + r] = acc;
        }
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += bias[k * 4 + r] * pv[c * 4 + k];
            cam->bvp[c * 4 + r] = acc;
        }
}

/* GL MIRRORED_REPEAT on an integer texel index */
static int64_t mirror_index(int64_t i, int64_t n) {
    int64_t period = 2 * n;
    int64_t m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - 1 - m;
}

static const float *texel(const float *tex, const uint32_t size[3], int64_t i, int64_t j, int64_t k) {
    i = mirror_index(i, size[0]);
    j = mirror_index(j, size[1]);
    k = mirror_index(k, size[2]);
    const size_t at = ((size_t)k * size[1] + (size_t)j) * size[0] + (size_t)i;
    if (tl_touch_map && tex == tl_touch_tex) tl_touch_map[at] = 1;
    return tex + at * 4;
}

/* texture(sampler3D, p01) with min/mag LINEAR (scene/sdf/mod.rs:241-250), wrap MirroredRepeat (:113-115) */
static void tex_linear(const float *tex, const uint32_t size[3], const float p01[3], float out[4]) {
    float u = p01[0] * (float)size[0] - 0.5f;
    float v = p01[1] * (float)size[1] - 0.5f;
    float w = p01[2] * (float)size[2] - 0.5f;
    float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    float ax = u - fu, ay = v - fv, az = w - fw;
    int64_t i0 = (int64_t)fu, j0 = (int64_t)fv, k0 = (int64_t)fw;
    const float *t000 = texel(tex, size, i0, j0, k0), *t100 = texel(tex, size, i0 + 1, j0, k0);
    const float *t010 = texel(tex, size, i0, j0 + 1, k0), *t110 = texel(tex, size, i0 + 1, j0 + 1, k0);
    const float *t001 = texel(tex, size, i0, j0, k0 + 1), *t101 = texel(tex, size, i0 + 1, j0, k0 + 1);
    const float *t011 = texel(tex, size, i0, j0 + 1, k0 + 1), *t111 = texel(tex, size, i0 + 1, j0 + 1, k0 + 1);
    if (or_ext_variant_flags & OR_EXT_TRILINEAR_WEIGHTED) { /* GL 4.6 spec 8.14.2, eq. for TEXTURE_3D LINEAR */
        float bx = 1.0f - ax, by = 1.0f - ay, bz = 1.0f - az;
        for (int c = 0; c < 4; ++c)
            out[c] = bx * by * bz * t000[c] + ax * by * bz * t100[c] + bx * ay * bz * t010[c] + ax * ay * bz * t110[c] +
                     bx * by * az * t001[c] + ax * by * az * t101[c] + bx * ay * az * t011[c] + ax * ay * az * t111[c];
        return;
    }
    for (int c = 0; c < 4; ++c) {
        float c00 = mixf(t000[c], t100[c], ax);
        float c10 = mixf(t010[c], t110[c], ax);
        float c01 = mixf(t001[c], t101[c], ax);
        float c11 = mixf(t011[c], t111[c], ax);
        float c0 = mixf(c00, c10, ay);
        float c1 = mixf(c01, c11, ay);
        out[c] = mixf(c0, c1, az);
    }
}

/* texture(sampler3D, p01) with NEAREST (scene/sdf/mod.rs:110-111) */
static void tex_nearest(const float *tex, const uint32_t size[3], const float p01[3], float out[4]) {
    int64_t i = (int64_t)floorf(p01[0] * (float)size[0]);
    int64_t j = (int64_t)floorf(p01[1] * (float)size[1]);
    int64_t k = (int64_t)floorf(p01[2] * (float)size[2]);
    memcpy(out, texel(tex, size, i, j, k), 4 * sizeof(float));
}

/* sdfSampleRawInterp / sdfSampleRawNearest, material.frag:27-53 */
static void sdf_sample_raw_interp(const float *tex, const OrRenderParams *rp, v3 p, float out[4]) {
    float p01[3] = {(p.x - rp->bounds_min[0]) / (rp->bounds_max[0] - rp->bounds_min[0]),
                    (p.y - rp->bounds_min[1]) / (rp->bounds_max[1] - rp->bounds_min[1]),
                    (p.z - rp->bounds_min[2]) / (rp->bounds_max[2] - rp->bounds_min[2])};
    if (rp->lod_dist_between_samples == 1.0f) {
        tex_linear(tex, rp->tex_size, p01, out);
    } else {
        float q[3];
        for (int c = 0; c < 3; ++c) {
            float round_steps = (float)rp->tex_size[c] / rp->lod_dist_between_samples;
            q[c] = roundf(p01[c] * round_steps) / round_steps;
        }
        tex_nearest(tex, rp->tex_size, q, out);
    }
}

void or_tex_sample(const float *tex, const OrRenderParams *rp, const float p[3], float out[4]) {
    sdf_sample_raw_interp(tex, rp, v3_from(p), out);
}

/* sdfOutOfBoundsDist, material.frag:83-88 */
static float oob_dist(const OrRenderParams *rp, v3 p) {
    float ox = fmaxf(rp->bounds_min[0] - p.x, p.x - rp->bounds_max[0]);
    float oy = fmaxf(rp->bounds_min[1] - p.y, p.y - rp->bounds_max[1]);
    float oz = fmaxf(rp->bounds_min[2] - p.z, p.z - rp->bounds_max[2]);
    return fmaxf(ox, fmaxf(oy, oz));
}

/* sdfNormal, material.frag:73-80 */
static v3 sdf_normal(const float *tex0, const OrRenderParams *rp, v3 p) {
    float sx = (float)rp->tex_size[0] / rp->lod_dist_between_samples;
    float sy = (float)rp->tex_size[1] / rp->lod_dist_between_samples;
    float sz = (float)rp->tex_size[2] / rp->lod_dist_between_samples;
    float h = 1.0f / sqrtf(sx * sx + sy * sy + sz * sz);
    static const float k[4][3] = {{1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, 1, 1}}; /* xyy, yyx, yxy, xxx */
    v3 acc = v3_make(0, 0, 0);
    for (int i = 0; i < 4; ++i) {
        v3 q = v3_make(p.x + k[i][0] * h, p.y + k[i][1] * h, p.z + k[i][2] * h);
        float raw[4];
        sdf_sample_raw_interp(tex0, rp, q, raw);
        float d = raw[0] - 1e-1f;
        v3 term = v3_make(k[i][0] * d, k[i][1] * d, k[i][2] * d);
        acc = (i == 0) ? term : v3_make(acc.x + term.x, acc.y + term.y, acc.z + term.z);
    }
    return v3_normalize(acc);
}

/* three-d 0.18.2 [EXT]: ambient-only calculate_lighting, tone_mapping, color_mapping; material.frag:158-173 */
static float tone_map(uint32_t type, float c) {
    if (type == 1) c = c / (c + 1.0f);
    else if (type == 2) c = (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f);
    else if (type == 3) {
        float x = fmaxf(0.0f, c - 0.004f);
        c = (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
        c = powf(c, 2.2f);
    }
    return fminf(fmaxf(c, 0.0f), 1.0f);
}
static float color_map(uint32_t type, float c) {
    if (type != 1) return c;
    float ginv = 1.0f / 2.4f;
    float select = c >= 0.0031308f ? 1.0f : 0.0f; /* step(edge, x) */
    float lo = c * 12.92f;
    float hi = 1.055f * powf(c, ginv) - 0.055f;
    return mixf(lo, hi, select);
}

void or_shade(const OrRenderParams *rp, const float raw0[4], const float raw1[4], float rgba[4]) {
    float metallic = raw1[0], occlusion = raw1[2];
    for (int c = 0; c < 3; ++c) {
        float albedo = raw0[1 + c] * rp->tint[c];                              /* :158-159 */
        float lit = occlusion * rp->ambient[c] * mixf(albedo, 0.0f, metallic); /* :163, ambient light */
        for (uint32_t l = 0; l < rp->n_lights && l < OR_MAX_LIGHTS; ++l)       /* further ambient lights, summed */
            lit += occlusion * (rp->lights[l].intensity * rp->lights[l].color[c]) * mixf(albedo, 0.0f, metallic);
        lit = tone_map(rp->tone_mapping, lit);                                 /* :167 */
        lit = color_map(rp->color_mapping, lit);                               /* :168 */
        if (rp->gamma > 0.0f) lit = powf(lit, rp->gamma);                      /* :171-173 */
        rgba[c] = lit;
    }
    rgba[3] = rp->tint[3]; /* :169 */
}

/* maps of the recording run (NULL outside or_raymarch_touch) */
typedef struct { uint8_t *march0, *hit0, *hit1, *normal0; } TouchMaps;
static const TouchMaps *g_touch = NULL;
static void touch_phase(const float *tex, uint8_t *map) { tl_touch_tex = tex; tl_touch_map = map; }

static void march_pixel(const OrRenderParams *rp, const float *tex0, const float *tex1, const OrCamera *cam,
                        uint32_t W, uint32_t H, uint32_t px, uint32_t py, float rgba[4], OrMarchAux *aux) {
    OrMarchAux a;
    memset(&a, 0, sizeof(a));
    a.depth = 1.0f;
    rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f;

    /* primary ray through the pixel centre; image row 0 is the top row */
    float ndc_x = (((float)px + 0.5f) / (float)W) * 2.0f - 1.0f;
    float ndc_y = 1.0f - (((float)py + 0.5f) / (float)H) * 2.0f;
    float sx = ndc_x * cam->aspect * cam->tan_half_fovy;
    float sy = ndc_y * cam->tan_half_fovy;
    v3 eye = v3_from(cam->eye);
    v3 d0 = v3_make(cam->forward[0] + cam->right[0] * sx + cam->up[0] * sy,
                    cam->forward[1] + cam->right[1] * sx + cam->up[1] * sy,
                    cam->forward[2] + cam->right[2] * sx + cam->up[2] * sy);
    d0 = v3_normalize(d0);

    /* bbox fragment: slab test (stands in for rasterising cube_with_bounds, scene/sdf/mod.rs:254-282) */
    float tx1 = (rp->bounds_min[0] - eye.x) / d0.x, tx2 = (rp->bounds_max[0] - eye.x) / d0.x;
    float ty1 = (rp->bounds_min[1] - eye.y) / d0.y, ty2 = (rp->bounds_max[1] - eye.y) / d0.y;
    float tz1 = (rp->bounds_min[2] - eye.z) / d0.z, tz2 = (rp->bounds_max[2] - eye.z) / d0.z;
    float tnear = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    float tfar = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    if (!(tfar >= tnear && tfar > 0.0f)) { /* pixel not covered by the box */
        if (aux) *aux = a;
        return;
    }
    float tfrag = tnear > 0.0f ? tnear : tfar;
    v3 pos = v3_madd(eye, d0, tfrag);

    /* main(), material.frag:130-182 */
    v3 ray_origin = pos;
    v3 ray_dir = v3_normalize(v3_sub(ray_origin, eye));
    if (oob_dist(rp, v3_madd(ray_origin, ray_dir, 0.2f)) > 0.0f) {
        ray_origin = v3_madd(eye, ray_dir, 0.2f);
    }

    /* sdfRaycast(rayOrigin, rayDir, 256), material.frag:92-128 */
    const int max_steps = 256;
    v3 ray_pos = ray_origin;
    float dist_from_origin = 0.0f;
    float hit_w = 0.0f; /* vec4(0.0) if the loop ends without a break: cannot happen for maxSteps=256 */
    float raw0[4] = {0, 0, 0, 0};
    int steps = 0, hit = 0;
    if (g_touch) touch_phase(tex0, g_touch->march0);
    for (int i = 0; i < max_steps; ++i) {
        if (i >= max_steps - 1) { hit_w = -1.0f; break; }
        if (oob_dist(rp, ray_pos) > 1e-4f) { hit_w = -2.0f; break; }
        float s[4];
        sdf_sample_raw_interp(tex0, rp, ray_pos, s);
        ++steps;
        float sample_dist = s[0] - 1e-1f;
        if (sample_dist < 1e-5f) {
            hit_w = dist_from_origin;
            memcpy(raw0, s, sizeof(raw0));
            hit = 1;
            break;
        }
        dist_from_origin += sample_dist;
        ray_pos = v3_madd(ray_pos, ray_dir, sample_dist);
    }
    if (g_touch) touch_phase(NULL, NULL);
    a.steps = steps;
    a.hit_pos[0] = ray_pos.x; a.hit_pos[1] = ray_pos.y; a.hit_pos[2] = ray_pos.z;
    a.t = dist_from_origin;
    if (!hit || hit_w < 0.0f) { /* :145-149; a hit whose accumulated distance is negative is also dropped */
        a.status = hit ? -3 : (int32_t)hit_w;
        if (aux) *aux = a;
        return;
    }
    a.status = 1;
    memcpy(a.raw0, raw0, sizeof(raw0));
    if (g_touch) { /* the texels under the hit: tex0's are those of the march's last fetch */
        float again[4];
        touch_phase(tex0, g_touch->hit0);
        sdf_sample_raw_interp(tex0, rp, ray_pos, again);
        touch_phase(tex1, g_touch->hit1);
    }
    sdf_sample_raw_interp(tex1, rp, ray_pos, a.raw1); /* :154 */
    if (g_touch) touch_phase(tex0, g_touch->normal0);
    v3 n = sdf_normal(tex0, rp, ray_pos);              /* :155 */
    if (g_touch) touch_phase(NULL, NULL);
    a.normal[0] = n.x; a.normal[1] = n.y; a.normal[2] = n.z;
    or_shade(rp, a.raw0, a.raw1, rgba);
    /* :180-181 */
    const float *m = cam->bvp;
    float hz = m[2] * ray_pos.x + m[6] * ray_pos.y + m[10] * ray_pos.z + m[14];
    float hw = m[3] * ray_pos.x + m[7] * ray_pos.y + m[11] * ray_pos.z + m[15];
    a.depth = hz / hw;
    if (aux) *aux = a;
}

void or_raymarch(const OrRenderParams *rp, const float *tex0, const float *tex1, const OrCamera *cam,
                 uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                 float *rgba, OrMarchAux *aux, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t y = y0; y < (int64_t)y1; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            size_t o = ((size_t)(y - y0)) * width + x;
            march_pixel(rp, tex0, tex1, cam, width, height, x, (uint32_t)y, rgba + o * 4, aux ? aux + o : NULL);
        }
    }
}

void or_raymarch_touch(const OrRenderParams *rp, const float *tex0, const float *tex1, const OrCamera *cam,
                       uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                       uint8_t *march0, uint8_t *hit0, uint8_t *hit1, uint8_t *normal0,
                       OrMarchCounts *counts, int n_threads) {
    (void)n_threads;
    const TouchMaps maps = {march0, hit0, hit1, normal0};
    g_touch = &maps; /* marks are idempotent byte stores of 1: threads may race on them */
    uint64_t covered = 0, hits = 0, sum_steps = 0, max_steps = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1) \
    reduction(+ : covered, hits, sum_steps) reduction(max : max_steps)
#endif
    for (int64_t y = y0; y < (int64_t)y1; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            float rgba[4];
            OrMarchAux a;
            march_pixel(rp, tex0, tex1, cam, width, height, x, (uint32_t)y, rgba, &a);
            covered += a.status != 0;
            hits += a.status == 1;
            sum_steps += (uint64_t)a.steps;
            if ((uint64_t)a.steps > max_steps) max_steps = (uint64_t)a.steps;
        }
    }
    g_touch = NULL;
    if (counts) {
        counts->pixels = (uint64_t)(y1 - y0) * width;
        counts->covered = covered;
        counts->hits = hits;
        counts->sum_steps = sum_steps;
        counts->max_steps = max_steps;
    }
}
