/*
 * mesh_front.c -- oracle restatement of the sampling front end of the reference's meshers.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see sdf_oracle.h).
 * Follows /root/reference/src/sdf/meshers/isosurface.rs:68-99 and src/sdf/meshers/mesh.rs:22-33,106-108.
 * The meshing algorithms themselves live in the un-vendored `isosurface` crate and are not restated.
 */
#include "sdf_oracle.h"

/* isosurface.rs:95-99: vec3(p).mul_element_wise(bb[1].sub(bb[0])).add(bb[0]) -- three roundings per axis */
void or_vert_pos_to(const float bb_min[3], const float bb_max[3], const float p[3], float out[3]) {
    for (int i = 0; i < 3; ++i) {
        float size = bb_max[i] - bb_min[i];
        float scaled = p[i] * size;
        out[i] = scaled + bb_min[i];
    }
}

/* isosurface.rs:78-84: sdf.sample(vert_pos_to(p), true).distance */
float or_source_scalar(const OrDemoParams *prm, uint32_t sdf_id, const float bb_min[3], const float bb_max[3],
                       const float p[3]) {
    float w[3];
    OrSample s;
    or_vert_pos_to(bb_min, bb_max, p, w);
    or_sample(prm, sdf_id, w, 1, &s);
    return s.distance;
}

/* isosurface.rs:87-92: sdf.normal(vert_pos_to(p), None) */
void or_source_normal(const OrDemoParams *prm, uint32_t sdf_id, const float bb_min[3], const float bb_max[3],
                      const float p[3], float out[3]) {
    float w[3];
    or_vert_pos_to(bb_min, bb_max, p, w);
    or_normal(prm, sdf_id, w, 0.0f, out);
}

/* mesh.rs:22-33 */
void or_mesh_postproc(const OrDemoParams *prm, uint32_t sdf_id, OrVertex *vertices, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        OrVertex *v = &vertices[i];
        OrSample s;
        or_sample(prm, sdf_id, v->position, 0, &s);
        /* v.normal.distance2(Vector3::zero()): (n - 0) . (n - 0), summed x, y, z */
        float dx = v->normal[0] - 0.0f, dy = v->normal[1] - 0.0f, dz = v->normal[2] - 0.0f;
        float d2 = dx * dx + dy * dy;
        d2 = d2 + dz * dz;
        if (d2 < 0.0001f) or_normal(prm, sdf_id, v->position, 0.0f, v->normal);
        v->color[0] = s.color[0];
        v->color[1] = s.color[1];
        v->color[2] = s.color[2];
        v->metallic = s.metallic;
        v->roughness = s.roughness;
        v->occlusion = s.occlusion;
    }
}

/* mesh.rs:106-108: (c * 255.9999) as u8 -- saturating, NaN -> 0 */
uint8_t or_ply_color_u8(float c) {
    float v = c * 255.9999f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
