/*
 * sdf_oracle.h -- CPU restatement ("oracle") of sdf-viewer's voxelise-and-raymarch path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as the
 * checker / the timed CPU baseline.  The product (libsdfgrid.so) never links or calls it.
 *
 * PARITY UNPINNED.  The reference is Rust + GLSL and cannot be built or run in this image (no
 * cargo/rustc, no GL, no wasm runtime), and its own tests hold no numeric vectors for this path
 * (the only unit tests are LoadingManager's, src/app/scene/sdf/loading.rs:117-171, which ARE
 * restated and pass against this oracle).  Three pieces of arithmetic live in crates that are not
 * under /root/reference and are restated here from their published behaviour:
 *   - three-d-asset 0.9.2  Srgba::from(Vec3) / Srgba::to_linear_srgb   (call site scene/sdf/mod.rs:201)
 *   - three-d 0.18.2       ambient calculate_lighting, ACES tone_mapping, sRGB color_mapping,
 *                          Camera::new_perspective                      (material.frag:163-168)
 *   - cgmath 0.18.0        distance / normalize op order                (sphere.rs:39,123)
 * plus the GL driver's trilinear filter (material.frag:19-23), restated as full-fp32 lerps.
 * Each of those is an isolated function below (srgb_u8_to_linear, shade_*, trilinear_*).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Build: plain C11, -O3 -ffp-contract=off (Rust never contracts a*b+c), scalar fp32, no fast-math.
 */
#ifndef SDF_ORACLE_H
#define SDF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- [EXT] sensitivity: the arithmetic restated from crates / GL that are not under the reference tree can be
 * switched, one piece at a time, to its plausible alternative (tools/ext_sensitivity.py measures how far the outputs
 * move).  0 = the restatement the parity tests use.  Process-wide; not for concurrent use with other oracle calls. */
enum {
    OR_EXT_SRGB_POW_ULP_UP      = 1u << 0, /* to_linear_srgb's powf result one ulp up (a libm that is not correctly rounded) */
    OR_EXT_SRGB_POW_ULP_DOWN    = 1u << 1, /* ... one ulp down */
    OR_EXT_SRGB_QUANT_ROUND     = 1u << 2, /* Srgba::from(Vec3): round(c * 255) instead of the truncating `as u8` */
    OR_EXT_CGMATH_NORMALIZE_DIV = 1u << 3, /* cgmath normalize as v / |v| instead of v * (1 / |v|) */
    OR_EXT_GLSL_MIX_LERP        = 1u << 4, /* mix(a, b, t) as a + t * (b - a) instead of a * (1 - t) + b * t */
    OR_EXT_GLSL_NORMALIZE_RSQ   = 1u << 5, /* GLSL normalize as v * (1 / sqrt(dot)) instead of v / length(v) */
    OR_EXT_TRILINEAR_WEIGHTED   = 1u << 6, /* GL spec form: sum of 8 weighted texels instead of nested mix x, y, z */
    OR_EXT_SRGB_DOUBLE_POW      = 1u << 7  /* to_linear_srgb evaluated in f64 and rounded once */
};
void     or_set_ext_variant(uint32_t flags);
uint32_t or_get_ext_variant(void);

/* src/sdf/mod.rs:104-118  #[repr(C)] SDFSample, 28 bytes */
typedef struct {
    float distance;
    float color[3];
    float metallic;
    float roughness;
    float occlusion;
} OrSample;

/* clap flags: cube.rs:15-18, sphere.rs:11-14, demo/mod.rs:26-29.  material: 0 = Brick, 1 = Normal */
typedef struct {
    float    cube_half_side;               /* 0.95 */
    uint32_t cube_material;                /* brick */
    float    sphere_radius;                /* 1.05 */
    uint32_t sphere_material;              /* normal */
    float    max_distance_custom_material; /* 0.05 */
    uint32_t disable_sphere;               /* false */
} OrDemoParams;

/* ids: demo/mod.rs:84 (0), cube.rs:92-94 (1), sphere.rs:50-52 (2) */
enum { OR_SDF_DEMO = 0, OR_SDF_CUBE = 1, OR_SDF_SPHERE = 2 };

void or_demo_default_params(OrDemoParams *p);
/* sample(id, p, distance_only): demo/mod.rs:51-75, cube.rs:79-89, sphere.rs:37-47 */
void or_sample(const OrDemoParams *prm, uint32_t sdf_id, const float p[3], int distance_only, OrSample *out);
/* normal(id, p, eps): demo/mod.rs:147-156, cube.rs:164-177, sphere.rs:122-124; eps<=0 -> None (ffi.rs:326) */
void or_normal(const OrDemoParams *prm, uint32_t sdf_id, const float p[3], float eps, float out[3]);
/* normal_default_impl, defaults.rs:49-56 (4-tap tetrahedral over sample(.., true)) */
void or_normal_default(const OrDemoParams *prm, uint32_t sdf_id, const float p[3], float eps, float out[3]);

/* ---- mesher front end (src/sdf/meshers): the reference's Vertex and the calls its meshers make ---- */
typedef struct OrVertex { /* meshers/mesh.rs:135-143 */
    float position[3], normal[3], color[3], metallic, roughness, occlusion;
} OrVertex;
/* SDFSurfaceWrapper::vert_pos_to, meshers/isosurface.rs:95-99: unit cube -> bounding box */
void or_vert_pos_to(const float bb_min[3], const float bb_max[3], const float p[3], float out[3]);
/* ScalarSource::sample_scalar, meshers/isosurface.rs:78-84 */
float or_source_scalar(const OrDemoParams *prm, uint32_t sdf_id, const float bb_min[3], const float bb_max[3], const float p[3]);
/* HermiteSource::sample_normal, meshers/isosurface.rs:87-92 */
void or_source_normal(const OrDemoParams *prm, uint32_t sdf_id, const float bb_min[3], const float bb_max[3], const float p[3], float out[3]);
/* Mesh::postproc, meshers/mesh.rs:22-33 */
void or_mesh_postproc(const OrDemoParams *prm, uint32_t sdf_id, OrVertex *vertices, size_t n);
/* (v.color[i] * 255.9999) as u8, meshers/mesh.rs:106-108 */
uint8_t or_ply_color_u8(float c);

/* ---- grid ---- */
/* AIR_DIST, scene/sdf/mod.rs:42 */
float or_air_dist(void);
/* from_bb voxel sizing, scene/sdf/mod.rs:46-72 */
void or_grid_dims_from_bb(const float bb_min[3], const float bb_max[3], uint32_t max_voxels_side, uint32_t dims[3]);
/* voxel index -> position, scene/sdf/mod.rs:179-182 */
float or_voxel_coord(uint32_t idx, uint32_t dim, float bb_min, float bb_max);
/* three-d-asset Srgba::from + to_linear_srgb, one channel [EXT] */
uint8_t or_srgb_quantize(float c);
float   or_srgb_u8_to_linear(uint8_t c);
/* packing of one sample into the two texels, scene/sdf/mod.rs:196-208 (tex1[3] untouched) */
void or_pack_sample(const OrSample *s, float tex0[4], float tex1[4]);
/* new_voxels init: both textures = [AIR_DIST;4], scene/sdf/mod.rs:76-77 */
void or_grid_init(float *tex0, float *tex1, size_t n_voxels);

/* Dense fill of z-slices [z0,z1) of a dims grid; tex pointers address slice z0.  Final state of
 * SDFViewer::update (scene/sdf/mod.rs:128-217) once the LoadingManager is exhausted on a fresh grid. */
void or_fill_dense(const OrDemoParams *prm, uint32_t sdf_id, const uint32_t dims[3],
                   const float bb_min[3], const float bb_max[3],
                   uint32_t z0, uint32_t z1, float *tex0, float *tex1, int n_threads);

/* LoadingManager, scene/sdf/loading.rs:5-115 */
typedef struct {
    uint64_t limits[3];
    uint64_t passes;
    uint64_t step_size;
    uint64_t next_index[3];
    uint64_t iterations;
    uint64_t total_iterations;
} OrLoadingManager;
void     or_lm_new(OrLoadingManager *m, const uint64_t limits[3], uint64_t passes);
int      or_lm_next(OrLoadingManager *m, uint64_t out_index[3]); /* 1 = Some, 0 = None */
uint64_t or_lm_len(const OrLoadingManager *m);
uint64_t or_lm_passes_left(const OrLoadingManager *m);
uint32_t or_prev_power_of_2(uint32_t x);

/* Faithful SDFViewer::update loop, scene/sdf/mod.rs:173-215: visits voxels in LoadingManager order for
 * at most max_iterations iterator steps (stand-in for the 30 ms budget), honours update_required
 * (tex0.r == AIR_DIST or inside changed_box; changed_box may be NULL).  Returns iterator steps consumed. */
uint64_t or_viewer_update(const OrDemoParams *prm, uint32_t sdf_id, const uint32_t dims[3],
                          const float bb_min[3], const float bb_max[3],
                          OrLoadingManager *lm, const float *changed_box /* 6 floats or NULL */,
                          uint64_t max_iterations, float *tex0, float *tex1);

/* The same loop over ANY SDFSurface (`sdf: impl SDFSurface`, scene/sdf/mod.rs:128): sample() is a callback with the trait
 * method's meaning (src/sdf/mod.rs:43).  The checker of the product's ingest path (host-sampled SDFs). */
typedef void (*or_sample_fn)(void *user, const float p[3], int distance_only, OrSample *out);
uint64_t or_viewer_update_fn(or_sample_fn sample, void *user, const uint32_t dims[3],
                             const float bb_min[3], const float bb_max[3],
                             OrLoadingManager *lm, const float *changed_box /* 6 floats or NULL */,
                             uint64_t max_iterations, float *tex0, float *tex1);

/* ---- raymarch ---- */
/* three-d Camera::new_perspective restated as a POD: the library and the oracle are handed the same
 * block, so the basis/tan/matrix arithmetic is not part of the parity surface. */
typedef struct {
    float eye[3];
    float right[3];
    float up[3];
    float forward[3];
    float tan_half_fovy;
    float aspect;
    float bvp[16]; /* column-major bias*projection*view, material.rs:89-97 */
} OrCamera;

/* one further entry of the scene's light list (scene/mod.rs:106-112); only ambient lights are restated */
typedef struct {
    uint32_t kind;       /* 0 ambient, 1 directional (not restated: three-d 0.18.2 shader text is not in the tree) */
    float color[3];
    float intensity;
    float direction[3];
} OrLight;
#define OR_MAX_LIGHTS 4

typedef struct {
    float bounds_min[3];
    float bounds_max[3];
    uint32_t tex_size[3];
    float lod_dist_between_samples; /* 1 when loaded, 2^passes_left while loading */
    float tint[4];                  /* surfaceColorTint, Srgba::WHITE -> (1,1,1,1) */
    float ambient[3];               /* intensity*color of the one AmbientLight, scene/mod.rs:106 */
    float gamma;                    /* GAMMA_CORRECTION define; <=0 = not defined */
    uint32_t tone_mapping;          /* 0 none, 1 reinhard, 2 aces(default), 3 filmic */
    uint32_t color_mapping;         /* 0 none, 1 compute-to-srgb(default) */
    uint32_t n_lights;              /* further AMBIENT lights; calculate_lighting sums the lights' contributions [EXT] */
    OrLight lights[OR_MAX_LIGHTS];
} OrRenderParams;

/* Per-pixel march record for parity on quantities fully determined by in-tree source. */
typedef struct {
    int32_t status;    /* 1 hit, -1 out of steps, -2 out of bounds, 0 ray misses the box */
    int32_t steps;     /* number of tex0 fetches performed by sdfRaycast */
    float   hit_pos[3];
    float   t;         /* distanceFromOrigin */
    float   raw0[4];   /* tex0 sample at the hit */
    float   raw1[4];   /* tex1 sample at the hit */
    float   normal[3];
    float   depth;     /* gl_FragDepth */
} OrMarchAux;

void or_default_render_params(OrRenderParams *rp, const uint32_t dims[3], const float bb_min[3], const float bb_max[3]);
/* look_at_rh + perspective(fovy deg) restated [EXT three-d 0.18.2 / cgmath]; scene/mod.rs:82-95 constants */
void or_camera_look_at(OrCamera *cam, const float eye[3], const float target[3], const float up[3],
                       float fovy_degrees, float aspect, float z_near, float z_far);
/* texture(sampler3D) LINEAR / NEAREST with MirroredRepeat, material.frag:19-53 */
void or_tex_sample(const float *tex, const OrRenderParams *rp, const float p[3], float out[4]);
/* material.frag main(), rows [y0,y1) of a W x H image (row 0 = top); rgba/aux address row y0 */
void or_raymarch(const OrRenderParams *rp, const float *tex0, const float *tex1, const OrCamera *cam,
                 uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                 float *rgba, OrMarchAux *aux /* may be NULL */, int n_threads);
/* The same march, recording WHICH texels it reads (SURVEY.md 8d's raymarch byte model): one byte per texel of the grid
 * in each map, set to 1 when touched (maps may be NULL).  march0: tex0 texels read by sdfRaycast's trilinear fetches;
 * hit0 / hit1: the 8 texels of tex0 / tex1 under each hit (material.frag:118,154); normal0: tex0 texels under
 * sdfNormal's four taps (material.frag:155).  Counts are deterministic. */
typedef struct {
    uint64_t pixels;     /* rows * width */
    uint64_t covered;    /* pixels whose ray meets the box */
    uint64_t hits;       /* status == 1 */
    uint64_t sum_steps;  /* tex0 fetches of sdfRaycast over all pixels */
    uint64_t max_steps;
} OrMarchCounts;
void or_raymarch_touch(const OrRenderParams *rp, const float *tex0, const float *tex1, const OrCamera *cam,
                       uint32_t width, uint32_t height, uint32_t y0, uint32_t y1,
                       uint8_t *march0, uint8_t *hit0, uint8_t *hit1, uint8_t *normal0,
                       OrMarchCounts *counts, int n_threads);
/* shading tail only (material.frag:158-173) for unit tests of the [EXT] restatement */
void or_shade(const OrRenderParams *rp, const float raw0[4], const float raw1[4], float rgba[4]);

#ifdef __cplusplus
}
#endif
#endif
