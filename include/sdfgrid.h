/*
 * sdfgrid.h -- C ABI of libsdfgrid.so: the MI355X-native voxelise-and-raymarch path of sdf-viewer.
 *
 * This is the drop-in boundary for the reference's hot path (paths relative to the reference root):
 *   - grid fill   src/app/scene/sdf/mod.rs:128-217  (SDFViewer::update, one SDFSurface::sample per voxel)
 *   - raymarch    src/app/scene/sdf/material.frag:92-182 (per-pixel sphere tracing of the two textures)
 * for the embedded demo SDF (src/sdf/demo/).  The reference exposes its SDFs over a per-point C ABI
 * (src/sdf/ffi.rs:42-337) and leaves batching as a TODO (src/sdf/mod.rs:39); the sdfv_* calls below
 * are those batched entry points.  The per-point ABI itself (bounding_box/sample/children/... with the
 * reference's unprefixed names) is exported by the companion libsdfdemo_provider.so, see sdf_provider.h.
 *
 * Conventions
 *   - plain C types only; every pointer documented as HOST or DEVICE (HIP device memory, gfx950).
 *   - output buffers are owned by the caller; the library never frees them (reference: callee-allocates
 *     + *_free, ffi.rs:52-55; the batched API inverts that because buffers are GBs and live on device).
 *   - every call returns 0 on success or a negative sdfv_status; sdfv_last_error() returns a
 *     thread-local message.  Nothing aborts (reference convention: log + zero result, ffi.rs:46-49).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only enqueue work;
 *     the caller synchronises the stream.  A handle/stream pair is single-owner, like the reference's
 *     thread-local registry (ffi.rs:15-17).
 *   - textures are RGBA32F, row-major with x fastest: flat = (z*H + y)*W + x (scene/sdf/mod.rs:177);
 *     tex0 = (clamped distance + 0.1, linear r, g, b), tex1 = (metallic, roughness, occlusion, AIR_DIST).
 */
#ifndef SDFGRID_H
#define SDFGRID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 3): the step forms other than SDFV_STEP_SIDE_BOUNDARY are gone; added sdfv_fill_grid_pass_ex, the pair and
 * interleaved volumes (sdfv_commit_pairs / _interleaved, sdfv_raymarch_pairs / _volumes, sdfv_march_volume_advice),
 * sdfv_raymarch_bands, the ray buffers with their count in band (sdfv_raymarch_slab_round, sdfv_slab_march). */
/* 4 (round 4): ONE raymarch entry point (sdfv_raymarch_ex over sdfv_march_desc) and ONE pass entry point (sdfv_fill_grid_pass_ex);
 * the positional forms of version 3 are header-only wrappers (source compatible, no longer exported).  Added SDFV_PASS_VIRGIN_GRID,
 * sdfv_grid_init_unvisited, SDFV_OPT_EXT_SRGB_QUANT, SDFV_OPT_PASS_INDEX_LIMIT, sdfv_bands_scatter and the sdfv_comm_* collectives.
 * The descriptor is size-prefixed: further layouts / outputs are new fields, not new versions. */
/* 5 (round 6): added sdfv_pack_samples (the device half of SDFViewer::update for host-sampled SDFs); removed
 * sdfv_tune_texture_placement / SDFV_PLACEMENT_SLACK (a probe that never beat the fixed placement in the driver's runs). */
#define SDFV_ABI_VERSION 5

typedef enum sdfv_status {
    SDFV_OK = 0,
    SDFV_ERR_INVALID_ARGUMENT = -1,
    SDFV_ERR_UNKNOWN_SDF = -2, /* ffi.rs:46-49 "Failed to find SDF with ID" */
    SDFV_ERR_HIP = -3,
    SDFV_ERR_NO_DEVICE = -4,
    SDFV_ERR_COMM = -5 /* RCCL could not be loaded, or one of its calls failed */
} sdfv_status;

/* src/sdf/mod.rs:104-118: #[repr(C)] struct SDFSample, 28 bytes */
typedef struct sdfv_sample {
    float distance;
    float color[3];
    float metallic;
    float roughness;
    float occlusion;
} sdfv_sample;

/* SDF ids of the demo hierarchy: demo/mod.rs:84 (root 0), cube.rs:92-94 (1), sphere.rs:50-52 (2) */
enum { SDFV_SDF_DEMO = 0, SDFV_SDF_CUBE = 1, SDFV_SDF_SPHERE = 2 };
/* cube.rs:20-24 enum Material */
enum { SDFV_MATERIAL_BRICK = 0, SDFV_MATERIAL_NORMAL = 1 };

/* The demo SDF's parameters = its clap flags (cube.rs:15-18, sphere.rs:11-14, demo/mod.rs:26-29). */
typedef struct sdfv_demo_params {
    float    cube_half_side;               /* -c, default 0.95 */
    uint32_t cube_material;                /* -t, default brick */
    float    sphere_radius;                /* -s, default 1.05 */
    uint32_t sphere_material;              /* -l, default normal */
    float    max_distance_custom_material; /* -m, default 0.05 */
    uint32_t disable_sphere;               /* -d, default false */
} sdfv_demo_params;

/* A voxel grid, or a z-slab of one (multi-GPU: rank r holds slices [z_begin, z_end) of the global
 * grid; the texture pointers handed to the calls address slice z_begin).  scene/sdf/mod.rs:46-117. */
typedef struct sdfv_grid {
    uint32_t dims[3];   /* global W, H, D */
    float    bb_min[3]; /* SDFSurface::bounding_box()[0] */
    float    bb_max[3]; /* SDFSurface::bounding_box()[1] */
    uint32_t z_begin;   /* first slice held */
    uint32_t z_end;     /* one past the last slice held; dense grid: 0, dims[2] */
} sdfv_grid;

/* three-d Camera::new_perspective (scene/mod.rs:82-95) flattened to what the shader consumes. */
typedef struct sdfv_camera {
    float eye[3];     /* cameraPosition, material.rs:62 */
    float right[3];   /* orthonormal view basis */
    float up[3];
    float forward[3];
    float tan_half_fovy;
    float aspect;     /* width / height of the FULL image */
    float bvp[16];    /* column-major bias * projection * view, material.rs:89-97 */
} sdfv_camera;

/* An entry of the scene's light list (`lights: Vec<Box<dyn Light>>`, scene/mod.rs:106-112).  The reference
 * configures ONE AmbientLight (intensity 1.0, white); its two DirectionalLights are commented out (scene/mod.rs:107-112).
 * Only SDFV_LIGHT_AMBIENT is implemented: what a directional light adds to calculate_lighting (material.frag:163) is
 * three-d 0.18.2's PBR shader text, which is not under the reference tree -- a call whose light list holds a
 * directional entry fails with SDFV_ERR_INVALID_ARGUMENT instead of rendering a guessed BRDF. */
enum { SDFV_LIGHT_AMBIENT = 0, SDFV_LIGHT_DIRECTIONAL = 1 };
#define SDFV_MAX_LIGHTS 4
typedef struct sdfv_light {
    uint32_t kind;         /* SDFV_LIGHT_* */
    float    color[3];     /* linear rgb */
    float    intensity;
    float    direction[3]; /* directional lights only */
} sdfv_light;

/* The shader's uniforms (material.rs:50-73) + three-d's tone/colour mapping selectors. */
typedef struct sdfv_render_params {
    float    bounds_min[3];            /* sdfBoundsMin */
    float    bounds_max[3];            /* sdfBoundsMax */
    uint32_t tex_size[3];              /* sdfTexSize */
    float    lod_dist_between_samples; /* sdfLODDistBetweenSamples: 1 = loaded (LINEAR), >1 = loading (NEAREST snap) */
    float    tint[4];                  /* surfaceColorTint */
    float    ambient[3];               /* AmbientLight intensity*colour, scene/mod.rs:106 */
    float    gamma;                    /* GAMMA_CORRECTION define (env "gamma", material.rs:39); <= 0 = undefined */
    uint32_t tone_mapping;             /* 0 none, 1 Reinhard, 2 ACES (three-d default), 3 filmic */
    uint32_t color_mapping;            /* 0 none, 1 compute-to-sRGB (three-d default) */
    uint32_t n_lights;                 /* FURTHER lights, after the scene's ambient light above; default 0 */
    sdfv_light lights[SDFV_MAX_LIGHTS]; /* each ambient entry adds occlusion * (intensity * color) * mix(albedo, 0, metallic) */
} sdfv_render_params;

/* Optional per-pixel march record (parity/debug): everything main() knows before shading. */
typedef struct sdfv_march_aux {
    int32_t status;   /* 1 hit, -1 out of steps, -2 out of bounds (material.frag:99-109), 0 pixel off the box */
    int32_t steps;    /* tex0 fetches done by sdfRaycast */
    float   hit_pos[3];
    float   t;        /* distanceFromOrigin */
    float   raw0[4];  /* tex0 sample at the hit */
    float   raw1[4];  /* tex1 sample at the hit */
    float   normal[3];
    float   depth;    /* gl_FragDepth, material.frag:180-181 */
} sdfv_march_aux;

/* ---- library ---- */
uint32_t    sdfv_abi_version(void);
const char *sdfv_last_error(void);  /* thread-local, never NULL */
const char *sdfv_build_id(void);    /* 12 hex digits: sha256 over the kernel sources this library was built from (csrc/Makefile;
                                      * tools/source_hash.py computes it from a tree) -- profiles/ stamps its evidence with it */
int         sdfv_device_count(void); /* number of HIP devices visible, 0 if none */
float       sdfv_air_dist(void);     /* AIR_DIST, scene/sdf/mod.rs:42 */

/* ---- per-thread options ----
 * Everything here defaults to what the product uses; the options exist for A/B measurements and so that the parity
 * tests can force every kernel specialisation.  Thread-local (like the error string and the reference's registry,
 * ffi.rs:15-17), read once per call; the library reads no environment variables. */
typedef enum sdfv_option {
    SDFV_OPT_FILL_NONTEMPORAL = 1,     /* texture stores of the dense fill: 0 auto (default: nt when the same launch also
                                        * writes the distance volume, plain otherwise) | 1 always nt | 2 never */
    SDFV_OPT_FILL_FORM = 2,            /* 0 auto (default) | 1 row-chunk form | 2 flat form of the dense fill | interleaved-volume fill of
                                        * widths that are multiples of 256 only: 3 one row per workgroup, pairs on one XCD | 4 a thread per
                                        * x of both rows of a pair.  A/B runs and tests; same texels */
    SDFV_OPT_RAYMARCH_DISABLE = 3,     /* mask of SDFV_RM_NO_*: exact-arithmetic specialisations left out (default 0) */
    SDFV_OPT_RAYMARCH_KEEP_NORMAL = 4, /* 0 (default) | 1: evaluate sdfNormal per hit although nothing consumes it */
    SDFV_OPT_SLAB_STEP_FORM = 5,       /* sdfv_slab_fill_step: 0 auto (default: packed messages for slabs below 2^26 voxels, per-texture
                                        * messages from there) | SDFV_STEP_SIDE_BOUNDARY, optionally with the SDFV_STEP_* flags below */
    SDFV_OPT_RAYMARCH_TILE_GROUP = 6,  /* raymarch workgroup-tile order: 0 auto (default: for a single frame XCD-aware groups of
                                        * 2 x 2 tiles with the box-first order below, 4 x 4 where that does not apply; launch order
                                        * with the tile columns rotated by row and camera for camera batches and tile bands, so that
                                        * the eight XCDs' shares of a launch balance) | 1 plain launch order | v = 2..5: XCD-aware
                                        * groups of 2^(v-1) x 2^(v-1) tiles (each group's tiles run on one XCD = one L2) */
    SDFV_OPT_RAYMARCH_BOX_FIRST = 7,   /* 1 (default) | 0: with an XCD-aware tile order and one camera, the groups of tiles under
                                        * the screen rectangle of the projected bounding box are launched before the others (the
                                        * frame is as long as its longest wave; those all start at once then).  Order only */
    SDFV_OPT_RAYMARCH_WAVES_PER_SIMD = 8, /* resident waves per SIMD of the raymarch kernel (unused dynamic LDS caps them): 0 (default)
                                        * = the launcher's rule: 4 for a single frame over a volume larger than the Infinity Cache
                                        * whose projected bounding box holds 1x .. 3.5x the machine's wave slots (latency-bound on
                                        * its long waves: 1440p over 512^3 -17 %, 4K -6 %), otherwise no cap | 2..6: that cap |
                                        * 7: never cap (what the register file allows).  Speed only */
    SDFV_OPT_RAYMARCH_BATCH_STREAMS = 9, /* 1 (default) | 0: a batch of more than 64 cameras is several launches (the cameras ride
                                        * in the kernel arguments); when each is small (at most 40 000 workgroups: low-resolution
                                        * views, a rank's share of a split batch) they run on side streams forked from and joined
                                        * back into the caller's stream, where they overlap -- a small launch is as long as its
                                        * longest waves, not as its work.  Events only: capturable.  Order only */
    SDFV_OPT_EXT_SRGB_QUANT = 10,      /* how the fill evaluates three-d-asset's Srgba::from(Vector3<f32>) (scene/sdf/mod.rs:201) -- a crate
                                        * whose source is not under the reference tree, so the restatement is unpinned: 0 (default)
                                        * (c * 255.0) as u8, truncating | 1 (c * 255.0 + 0.5) as u8, rounding.  The only restated
                                        * piece whose alternative moves visible output (19 M tex0 words at 256^3, profiles/
                                        * ext_sensitivity.json): both are compiled in, so whoever runs tools/ref_golden/ against the
                                        * real crate flips a flag instead of patching kernels.  Affects every fill and pass call */
    SDFV_OPT_PASS_INDEX_LIMIT = 11,    /* 0 (default) = 2^32 | v in [2, 2^32]: a LoadingManager pass over a slab of v voxels or more runs as
                                        * several launches over pieces of whole slices, each below v (the pass kernels index with 32
                                        * bits; slabs from ~1626^3 voxels up, which fit in 288 GB, need it).  A test hook: small
                                        * values exercise the piecewise path on small grids.  Same texels */
    SDFV_OPT_RAYMARCH_CAMERA_STAGING = 12, /* 1 (default) | 0: a launch carries up to 16 cameras in its kernel arguments; a larger batch handed
                                        * over as a HOST array is written into a per-thread ring of device memory on the caller's
                                        * stream (kernels that carry 32 cameras each as arguments: no copy engine, no allocation
                                        * per call), so that 64 cameras are one launch; 0 = launches of 16 cameras instead (what a
                                        * stream under capture gets anyway).  A DEVICE array is always read in place.  Same pixels */
    SDFV_OPT_PASS_FORM = 13,           /* 0 auto (default) | 1: a pass the caller says nothing about (no flags, no box) with step 2..8 takes
                                        * the per-voxel kernel (one texel in every `step`) instead of the whole-rows kernel whose
                                        * waves decide on the volume they read (sdfv_fill_grid_pass_ex).  A/B runs; same texels --
                                        * with one caveat the whole-rows form shares with the reference's own reading of the value:
                                        * where 64 consecutive voxels of a visited row ALL store exactly AIR_DIST it takes them for
                                        * new_voxels' state and writes the ones between lattice points as [AIR_DIST; 4]; a voxel
                                        * SAMPLED to exactly 0.1 + d == AIR_DIST there (update_required re-samples such a voxel on
                                        * every visit anyway, scene/sdf/mod.rs:184) would lose its colour until then.  Option 1 and
                                        * the reference leave off-lattice texels alone */
    SDFV_OPT_PASS_LOADS = 15,          /* how a pass reads the volume (or tex0.r) for update_required: 0 auto (default: nontemporal loads
                                        * when the caller passes SDFV_PASS_EXPECT_NOOP, cached loads otherwise) | 1 cached | 2
                                        * nontemporal.  A/B runs; same texels */
    SDFV_OPT_RCCL_LIBRARY = 14,        /* PROCESS-wide, before the first communicator: value = address of a NUL-terminated path of the RCCL-ABI
                                        * library sdfv_slab_comm_* loads (copied; 0 = "librccl.so.1" by name, the default).  For
                                        * installations whose RCCL is not on the loader's path -- and how tests/c/mock_rccl.cpp stands
                                        * in for RCCL so that several ranks can run on one device.  Refused once RCCL is loaded (a
                                        * path that fails to load is final too: one attempt per process, its dlerror() text is in
                                        * sdfv_last_error()).  sdfv_get_option hands back the address of a copy of the path that
                                        * belongs to the calling thread, valid until that thread asks again */
    SDFV_OPT_TUNING_WAVE_TIMING = 100, /* tuning build only (-DSDFV_TUNING): DEVICE address of 32 B per raymarch wave */
    SDFV_OPT_TUNING_TILE_ORDER = 102,  /* tuning build only: DEVICE address of tiles_x * tiles_y uint32 tile numbers (row-major
                                        * tile index by * tiles_x + bx): workgroup L of a single-camera launch renders tile
                                        * order[L] (an oracle for longest-first scheduling experiments) */
    SDFV_OPT_TUNING_PRIORITY_MAP = 101 /* tuning build only: DEVICE address of one byte per 16 x 16 raymarch tile (row-major);
                                        * waves of tiles whose byte is non-zero raise their issue priority at start */
} sdfv_option;
#define SDFV_RM_NO_FAST_INDEX  1u /* general kernel: full MirroredRepeat, the shader's nested loop */
#define SDFV_RM_NO_POW2_EXTENT 2u /* (p - min) / size by IEEE divide instead of the exact reciprocal */
#define SDFV_RM_NO_POW2_SIZE   4u /* no fused (1/size)*N scale */
#define SDFV_RM_NO_SYMMETRIC   8u /* max(min - p, p - max) instead of |p| - max */
#define SDFV_RM_NO_ASM_LOOP   16u /* the compiler's march loop instead of the hand-written gfx950 one */
#define SDFV_RM_NO_INTERIOR_FETCH 32u /* hand-written loop: always the clamping cell fetch, never the interior fast path */
#define SDFV_STEP_SIDE_BOUNDARY 3u /* the one step form (values 1 and 2, round 2's two-launch / one-launch forms, lost every
                                    * measurement and are gone): the caller's stream runs the plain dense fill of the whole
                                    * slab; the communicator's stream computes the boundary slices once more, into the packed
                                    * send buffers only, exchanges, and copies what it received into the ghosts */
#define SDFV_STEP_UNPACKED     4u /* flag: 2 messages per texture and neighbour straight out of / into the textures (no
                                   * staging): the communicator's stream fills the boundary slices in place, the caller's
                                   * stream everything else */
#define SDFV_STEP_START_EVENT  8u /* flag: release the communicator's stream with an event recorded on the caller's stream
                                   * instead of the fill launch's own "started" signal (the fallback where
                                   * hipStreamWaitValue32 is unavailable) */
#define SDFV_STEP_DEFER_JOIN  16u /* flag (packed messages only; ignored with per-texture messages, where the communicator's
                                   * stream writes OWNED slices): the step does NOT make the caller's stream wait for the
                                   * exchange; sdfv_slab_comm_join() does, whenever the ghost slices are needed.  Steps on the
                                   * same communicator order themselves after the previous exchange on their own */
int sdfv_set_option(uint32_t option, uint64_t value); /* unknown option / value out of range: SDFV_ERR_INVALID_ARGUMENT */
int sdfv_get_option(uint32_t option, uint64_t *value);

void sdfv_demo_params_default(sdfv_demo_params *p);
/* SDFViewer::from_bb voxel sizing, scene/sdf/mod.rs:46-72 */
int  sdfv_grid_from_bb(const float bb_min[3], const float bb_max[3], uint32_t max_voxels_side, sdfv_grid *out);
/* SDFViewerMaterial::new defaults + scene lights, material.rs:23-31, scene/mod.rs:106 */
void sdfv_render_params_default(sdfv_render_params *rp, const sdfv_grid *grid);
/* Camera::new_perspective(eye, target, up, degrees(fovy), near, far), scene/mod.rs:82-95 */
int  sdfv_camera_look_at(sdfv_camera *cam, const float eye[3], const float target[3], const float up[3],
                         float fovy_degrees, float aspect, float z_near, float z_far);

/* ---- grid fill (DEVICE pointers, 16 B per voxel per texture, 16-byte aligned) ---- */

/* SDFViewer::new_voxels initial state: both textures = [AIR_DIST; 4] (scene/sdf/mod.rs:76-77). */
int sdfv_grid_init(const sdfv_grid *grid, float *tex0, float *tex1, void *stream);

/* The lazy half of new_voxels for a grid loaded with SDFV_PASS_VIRGIN_GRID passes (below): [AIR_DIST; 4] into every row of the
 * slab that a pass with `step` (a power of two; the SMALLEST step run so far) does not visit -- y or global z not a multiple of
 * step -- and AIR_DIST into those rows of `dist` (DEVICE, one float per voxel, or NULL).  step 0: no pass has run, every row
 * (= sdfv_grid_init + the volume); step 1: nothing to do.  A host calls it before anything READS the whole grid while a
 * virgin load is unfinished (a frame at an intermediate LOD, a download, a pass with a changed box). */
int sdfv_grid_init_unvisited(const sdfv_grid *grid, uint32_t step, float *tex0, float *tex1, float *dist, void *stream);
/* ... the same over a grid whose volume is y-interleaved (SDFV_PASS_VOLUME_INTERLEAVED): flags = that bit, or 0 */
int sdfv_grid_init_unvisited_ex(const sdfv_grid *grid, uint32_t step, float *tex0, float *tex1, float *dist, uint32_t flags,
                                void *stream);

/* Dense fill: the state SDFViewer::update (scene/sdf/mod.rs:128-217) converges to on a fresh grid once
 * the LoadingManager is exhausted.  Store-only (32 B/voxel), does not read the textures; writes tex1.a =
 * AIR_DIST itself, so no prior sdfv_grid_init is needed. */
int sdfv_fill_grid(const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *grid,
                   float *tex0, float *tex1, void *stream);

/* sdfv_fill_grid and sdfv_commit_distance in ONE pass: the dense fill also writes the compact distance volume
 * (dist: DEVICE, one float per voxel of the slab, or NULL = plain sdfv_fill_grid).  +4 B/voxel of stores instead of a
 * second pass that re-reads tex0 (SDFViewer::update to completion followed by SDFViewer::commit, scene/sdf/mod.rs:128-239). */
int sdfv_fill_grid_commit(const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *grid, float *tex0,
                          float *tex1, float *dist, void *stream);
/* One LoadingManager pass (loading.rs:50-76) with step `step` (a power of two >= 1) over the slab: sdfv_fill_grid_pass_ex below is
 * the one exported form; sdfv_fill_grid_pass (no distance volume, no flags) and sdfv_fill_grid_pass_dist (no flags) are
 * header-only wrappers at the end of this file.  The pass visits voxels whose x, y and GLOBAL z are multiples of step and
 * applies update_required (scene/sdf/mod.rs:184-190): tex0.r == AIR_DIST, or position inside changed_box (HOST, 6 floats
 * min.xyz max.xyz, may be NULL).  Requires initialised textures (or SDFV_PASS_VIRGIN_GRID).
 * `dist` (DEVICE, one float per voxel, or NULL): the textures' compact distance volume, equal to tex0.r on entry -- as
 * sdfv_fill_grid_commit or sdfv_commit_distance leave it, also right after sdfv_grid_init.  update_required then reads 4 bytes
 * instead of a 16-byte texel, updated voxels rewrite their dist entry, and tex1.a is written as the AIR_DIST it holds in any
 * grid this library initialised or filled (the reference never writes it, scene/sdf/mod.rs:205-208) instead of being read
 * back.  Same texels; a no-op pass over a loaded 256^3 grid moves 67 MB instead of 268 MB. */

/* `flags`: what the CALLER knows about the grid (0 = nothing).  Whenever update_required
 * is known to hold for every visited voxel the pass reads nothing: at step 1 it is the dense fill (+ the distance volume),
 * at larger steps a store-only strided pass.  The library finds one such case itself -- a changed_box that contains every
 * voxel of the slab (what the demo reports on any parameter edit, demo/mod.rs:135-144) -- the flags name the two a host
 * that drives a LoadingManager knows (host/sdf_viewer.cpp does):
 *   SDFV_PASS_FRESH_GRID  every voxel of the slab holds new_voxels' initial state [AIR_DIST; 4] (scene/sdf/mod.rs:76-77,
 *                         sdfv_grid_init) on entry: the FIRST pass of a load.  The voxels between the visited ones are then
 *                         known as well, and a pass with step 2..8 writes the visited rows whole (visited x: the sample,
 *                         the others: the AIR texel they hold) -- whole 128-byte lines instead of one texel in every
 *                         `step`, no read-modify-write of partly written lines.
 *   SDFV_PASS_SAME_LOAD   every stored (non-AIR) voxel the pass visits was written by an earlier pass of the SAME load --
 *                         same SDF, same parameters, no changed box since -- so rewriting it stores the bits it holds:
 *                         the later passes of a load (loading.rs:50-76 revisits what coarser passes sampled).
 * The textures after the call are those sdfv_fill_grid_pass_dist leaves, bit for bit, provided the flags are true; a false
 * flag overwrites voxels the reference would have kept. */
#define SDFV_PASS_FRESH_GRID 1u
#define SDFV_PASS_SAME_LOAD  2u
/*   SDFV_PASS_VIRGIN_GRID the slab is a fresh ALLOCATION: new_voxels' initial state has not been WRITTEN (sdfv_grid_init was
 *                         skipped), its voxels are only LOGICALLY [AIR_DIST; 4] wherever no earlier pass of this load stored a
 *                         sample.  The pass reads nothing and writes the rows it visits whole -- samples on its lattice,
 *                         [AIR_DIST; 4] between them (and the distance volume's entries alike) -- so after it the rows with y
 *                         and global z multiples of `step` hold exactly what the reference's textures hold there; the other
 *                         rows stay UNDEFINED until a pass with a smaller step writes them or sdfv_grid_init_unvisited(step)
 *                         does.  Implies FRESH_GRID's knowledge; on the later passes of the load combine it with SAME_LOAD.
 *                         A step-1 pass is the dense fill and leaves nothing undefined: a load that runs all its passes
 *                         never pays for the initial state (36 B/voxel) at all.  changed_box must be NULL. */
#define SDFV_PASS_VIRGIN_GRID 4u
/*   SDFV_PASS_VOLUME_INTERLEAVED  `dist` is laid out as the Y-INTERLEAVED volume (sdfv_march_desc.ilv; sdfv_commit_interleaved's
 *                         layout: entry ((row >> 1) * W + x) * 2 + (row & 1) with row = z_local * H + y; H even, 8-byte aligned)
 *                         instead of one float per voxel in texture order -- on entry and on exit.  The fill then writes the
 *                         volume the march gathers fastest from beyond the last-level cache ITSELF (a step-1 pass with
 *                         SDFV_PASS_VIRGIN_GRID is the dense fused fill: textures + this volume, 36 B/voxel, no commit pass
 *                         afterwards: 512^3 0.70 + 0.20 ms -> 0.70), and every later pass reads and maintains it in place.
 *                         Pass the same buffer as sdfv_march_desc.ilv. */
#define SDFV_PASS_VOLUME_INTERLEAVED 8u
/*   SDFV_PASS_EXPECT_NOOP a HINT, not knowledge (a wrong hint costs time, never texels): the caller expects the pass to leave most of
 *                         what it visits alone -- the passes a LoadingManager runs over a loaded grid once a changed box has been
 *                         worked off (scene/sdf/mod.rs:146-156: one more full manager, without the box).  The pass then scans the
 *                         volume with NONTEMPORAL loads where the volume is larger than the last-level cache: nothing is
 *                         allocated in L2 / the Infinity Cache, so the scan does not push the previous fill's dirty lines out
 *                         first -- a step-1 no-op pass over 512^3 reads at 6.3 TB/s instead of 4.2 (0.085 against 0.128 ms, same
 *                         box).  A pass that does update most voxels runs 7-13 % slower with such loads (its partial-line stores
 *                         find the lines gone), and a volume that fits the cache is read faster through it: leave the hint off
 *                         for loads; the library ignores it for small volumes. */
#define SDFV_PASS_EXPECT_NOOP 16u
int sdfv_fill_grid_pass_ex(const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *grid, uint32_t step,
                           const float *changed_box, float *tex0, float *tex1, float *dist, uint32_t flags, void *stream);

/* ---- ingest: SDFViewer::update for an SDF that only the HOST can sample (any `impl SDFSurface`: a wasm / FFI provider,
 * src/sdf/wasm/native.rs:188-217, src/sdf/ffi.rs:57-65) ----
 * The host evaluates SDFSurface::sample(pos, false) for the voxels its LoadingManager visits and update_required lets through
 * (scene/sdf/mod.rs:173-193) and hands the RAW 28-byte records over; everything update() does with a sample afterwards
 * (scene/sdf/mod.rs:196-208: 0.1 + distance clamped to [0, 1], an all-zero colour replaced by 0.5 grey, Srgba::from +
 * to_linear_srgb, occlusion <= 0 -> 1) runs on the device, with the packing code of the fill kernels, bit for bit.
 *   samples     DEVICE, n records (4-byte aligned)
 *   indices     DEVICE, n uint32, or NULL: record i belongs to the voxel with flat index index_base + indices[i] (NULL: index_base
 *               + i, a contiguous run) of the SLAB the texture pointers address -- flat = ((z - z_begin) * H + y) * W + x
 *               (scene/sdf/mod.rs:177).  The indices of one call must be distinct (two records for one voxel race).  A record
 *               addressed beyond the slab's voxels is skipped: a host may mark records it does not want stored that way
 *   tex0, tex1  DEVICE, the slab's textures; tex0 is written whole, tex1 .rgb only -- .a keeps what the grid holds (the
 *               reference never writes it, scene/sdf/mod.rs:205-208), so the textures must be initialised (sdfv_grid_init, a
 *               fill, or sdfv_grid_init_unvisited for a virgin grid)
 *   dist        DEVICE or NULL: the compact distance volume, kept equal to tex0.r
 *   flags       SDFV_PASS_VOLUME_INTERLEAVED (the layout of dist) or 0
 * SDFV_OPT_EXT_SRGB_QUANT applies.  Enqueues one launch; `samples` / `indices` may be reused once it has run. */
int sdfv_pack_samples(const sdfv_grid *grid, uint64_t index_base, const uint32_t *indices, const sdfv_sample *samples, size_t n,
                      float *tex0, float *tex1, float *dist, uint32_t flags, void *stream);

/* ---- batched point sampling (the "Batched sampling" TODO, src/sdf/mod.rs:39) ---- */
/* points: DEVICE, n x 3 floats.  out: DEVICE, n x sdfv_sample.  SDFSurface::sample(p, distance_only). */
int sdfv_sample_points(const sdfv_demo_params *params, uint32_t sdf_id, const float *points, size_t n,
                       int distance_only, sdfv_sample *out, void *stream);
/* out: DEVICE, n x 3 floats.  SDFSurface::normal(p, eps): eps <= 0 means None (ffi.rs:326).
 * use_default != 0 evaluates normal_default_impl (defaults.rs:49-56) instead of the demo's overrides. */
int sdfv_normal_points(const sdfv_demo_params *params, uint32_t sdf_id, const float *points, size_t n,
                       float eps, int use_default, float *out, void *stream);

/* ---- mesher front end (src/sdf/meshers): what the isosurface extractors and Mesh::postproc ask of the SDF ---- */
/* Vertex, meshers/mesh.rs:135-143 (12 floats, 48 bytes) */
typedef struct sdfv_vertex {
    float position[3];
    float normal[3];
    float color[3];
    float metallic;
    float roughness;
    float occlusion;
} sdfv_vertex;
/* ScalarSource::sample_scalar for n points of the UNIT cube (meshers/isosurface.rs:78-84): each point goes through
 * vert_pos_to (p * (bb_max - bb_min) + bb_min, isosurface.rs:95-99), then sample(p, true).distance.
 * unit_points: DEVICE n x 3 floats; dist_out: DEVICE n floats. */
int sdfv_source_sample_scalar(const sdfv_demo_params *params, uint32_t sdf_id, const float bb_min[3],
                              const float bb_max[3], const float *unit_points, size_t n, float *dist_out, void *stream);
/* HermiteSource::sample_normal (meshers/isosurface.rs:87-92): normal(vert_pos_to(p), None).  out: DEVICE n x 3. */
int sdfv_source_sample_normal(const sdfv_demo_params *params, uint32_t sdf_id, const float bb_min[3],
                              const float bb_max[3], const float *unit_points, size_t n, float *normal_out, void *stream);
/* Mesh::postproc (meshers/mesh.rs:22-33) in place: colour/metallic/roughness/occlusion from sample(position, false);
 * normal from normal(position, None) where the mesher left |normal|^2 < 1e-4.  vertices: DEVICE, n x sdfv_vertex. */
int sdfv_mesh_postproc(const sdfv_demo_params *params, uint32_t sdf_id, sdfv_vertex *vertices, size_t n, void *stream);

/* An indexed triangle mesh in DEVICE memory, allocated by sdfv_mesh_extract and owned by the library until
 * sdfv_mesh_free (the reference's callee-allocates + *_free convention, ffi.rs:52-55).  Mesh, meshers/mesh.rs:11-17. */
typedef struct sdfv_mesh {
    sdfv_vertex *vertices; /* DEVICE, n_vertices */
    uint32_t    *indices;  /* DEVICE, n_indices = 3 * triangles, counter-clockwise seen from outside */
    size_t       n_vertices;
    size_t       n_indices;
} sdfv_mesh;
#define SDFV_MESHER_MARCHING_CUBES 0u /* Meshers::MarchingCubes, meshers/mod.rs:118-119 (the default, :130-134) */
/* Meshers::mesh (meshers/mod.rs:136-149 -> isosurface.rs:16-66) on the device: max_voxels_per_axis^3 cells over the
 * unit cube mapped onto the bounding box, distances from ScalarSource, one vertex per crossing lattice edge with
 * its HermiteSource normal, material fields zero (Vertex::default) until sdfv_mesh_postproc.  The extraction
 * algorithm itself is the build's own (the reference delegates to the un-vendored `isosurface` crate); algorithms
 * other than marching cubes are rejected like the reference's "Unsupported algorithm" (isosurface.rs:49).
 * Synchronises `stream` (the output size is data dependent).  Vertex and triangle counts are 32-bit: a surface has
 * O(N^2) of them, far below 2^32 for max_voxels_per_axis <= 1024; a field crossing zero on nearly every lattice edge
 * (not a distance field) at the largest sizes would overflow them. */
int sdfv_mesh_extract(const sdfv_demo_params *params, uint32_t sdf_id, const float bb_min[3], const float bb_max[3],
                      uint32_t max_voxels_per_axis, uint32_t algorithm, sdfv_mesh *out, void *stream);
int sdfv_mesh_free(sdfv_mesh *mesh);
/* sdfv_mesh_extract keeps its scratch (about 13 bytes per lattice point) for the calling thread's next extraction;
 * this releases it -- and the three side streams a batch of more than 64 cameras makes (SDFV_OPT_RAYMARCH_BATCH_STREAMS) and the
 * camera ring (SDFV_OPT_RAYMARCH_CAMERA_STAGING). */
int sdfv_mesh_trim(void);

/* ---- raymarch ----
 * ONE exported entry point, one descriptor: sdfv_raymarch_ex.  The forms earlier ABI versions exported one by one
 * (sdfv_raymarch, _accel, _depth, _pairs, _volumes, _bands) are header-only wrappers at the end of this file that fill the
 * descriptor; a new acceleration layout or output plane adds a FIELD, not a function and not an ABI version. */

/* material.frag main() for every pixel of rows [y0, y1) -- or of the 16-row tile bands band_first, band_first + band_step, ...
 * -- of n_cameras W x H images (row 0 = top).  `size` = sizeof(sdfv_march_desc) as the CALLER compiled it: the library reads
 * that many bytes and takes every field beyond them as 0 / NULL, so a binder built against an older header keeps working.
 *   rp                uniforms (material.rs:50-73); a light-list entry that is not SDFV_LIGHT_AMBIENT fails the call
 *                     (SDFV_ERR_INVALID_ARGUMENT: three-d 0.18.2's shader source is not available)
 *   tex0, tex1        DEVICE, the FULL grid rp->tex_size
 *   dist              DEVICE or NULL: compact distance volume (sdfv_commit_distance / the fused fill): the march gathers
 *                     4-byte distances instead of the r channel of 16-byte texels; bit-identical results
 *   pairs             DEVICE (8-byte aligned) or NULL: y-pair volume (sdfv_commit_pairs) -- texel (x, y, z) = (d[y], d[min(y+1,
 *                     H-1)]): a cell's z-level is 16 contiguous bytes, two 16-byte gathers per cell instead of four 8-byte
 *                     ones (1080p over 256^3 -6.5 %, 64-camera batch -15 %).  8 B/voxel.  Bit-identical
 *   ilv               DEVICE (8-byte aligned) or NULL: y-interleaved volume (sdfv_commit_interleaved) -- rows 2p, 2p + 1 of a
 *                     slice as ONE row of pairs: 2 cache lines per cell for even y, 4 for odd, at 4 B/voxel (4K over 512^3
 *                     -7..-13 %).  Bit-identical.  Given several volumes the launcher applies sdfv_march_volume_advice's
 *                     rule; the pair / interleaved volumes serve the hand-written gfx950 loop (any grid size, power-of-two
 *                     extents, symmetric box, <= 2^28 texels) on a CUBIC grid, every other launch reads dist / tex0.r
 *   cameras           HOST array of n_cameras, free again when the call returns (up to 16 ride in a launch's kernel arguments; a
 *                     larger batch goes through the library's ring of device memory, SDFV_OPT_RAYMARCH_CAMERA_STAGING, so that
 *                     64 cameras are one launch).  The array may instead lie in DEVICE (or managed) memory, whatever its
 *                     length: read in place by the launches, nothing copied (the library asks hipPointerGetAttributes).  One
 *                     consequence: a SINGLE camera handed over in device memory renders in plain tile order -- the launcher's
 *                     box-first order and occupancy rule read the camera on the host (same pixels; hand single frames over
 *                     as a host struct)
 *   y0, y1            rows [y0, y1) when band_step == 0
 *   band_first, band_step, band_height   band_step >= 1: the balanced image-tile split of BASELINE config 5 -- rank r of N renders
 *                     (r, N): the bands band_first, band_first + band_step, ... of band_height = 16 (also 0) or 8 rows -- a
 *                     workgroup's / a wave's tile -- stored one after the other (outputs hold n_cameras x sdfv_band_rows_ex(height,
 *                     band_first, band_step, band_height) x width pixels; 8-row bands deal the rows under the object more evenly when
 *                     a rank gets few bands: 8 ranks at 1080p 4.9x -> 5.4x, 4 ranks -2 %: tools/split_balance.py; a band set that starts below the
 *                     image renders nothing and succeeds); y0 / y1 are ignored.  Contiguous row ranges leave the outer ranks
 *                     with background only (8 ranges of a 1080p orbit view scale 2.4x on 8 GPUs, 8 band sets 5-6x)
 *   rgba              DEVICE, n_cameras x rows x W x 4 floats (outColor)
 *   depth             DEVICE or NULL, n_cameras x rows x W floats: gl_FragDepth (material.frag:180-181: (BVP * vec4(hit, 1)).z /
 *                     .w at a hit; 1.0 for a fragment that hits nothing, :147; 1.0 where the ray misses the box) -- equal bit for
 *                     bit to sdfv_march_aux.depth without the 72-byte record
 *   aux               DEVICE or NULL, same pixel layout: everything main() knows before shading
 *   rgba8             DEVICE or NULL, n_cameras x rows x W pixels of 4 bytes (R, G, B, A in memory order): outColor as the 8-bit
 *                     UNORM framebuffer the reference renders into holds it (src/app/frameinput.rs:19-24; no FRAMEBUFFER_SRGB
 *                     re-encoding: the shader's own colour mapping already produced display values) -- each channel
 *                     rint(clamp(c, 0, 1) * 255), a NaN as 0.  With rgba8 given, `rgba` may be NULL: the launch then stores 4
 *                     bytes per pixel instead of 16 -- what a 64-camera batch wants when the images go on to a display, an
 *                     encoder or another rank (a quarter of the store traffic and of the gather: 0.53 GB instead of 2.1).
 *                     fp32 `rgba` stays the parity output; at least one of the two must be given */
typedef struct sdfv_march_desc {
    uint32_t size;
    uint32_t reserved; /* 0 */
    const sdfv_render_params *rp;
    const float *tex0, *tex1;
    const float *dist, *pairs, *ilv;
    const sdfv_camera *cameras;
    uint32_t n_cameras;
    uint32_t width, height;
    uint32_t y0, y1;
    uint32_t band_first, band_step;
    uint32_t band_height; /* rows per band: 16 (also 0) or 8 */
    float *rgba;
    float *depth;
    sdfv_march_aux *aux;
    uint32_t *rgba8; /* (added in ABI 5) */
} sdfv_march_desc;
int sdfv_raymarch_ex(const sdfv_march_desc *desc, void *stream);
/* rows a band set holds: bands of band_height (16 or 8; 0 = 16) rows, the last one of the image possibly short */
uint32_t sdfv_band_rows_ex(uint32_t height, uint32_t band_first, uint32_t band_step, uint32_t band_height);
uint32_t sdfv_band_rows(uint32_t height, uint32_t band_first, uint32_t band_step); /* = ..._ex(..., 16) */

/* Device-side analogue of SDFViewer::commit (scene/sdf/mod.rs:220-239).  The textures already live in HBM, so
 * there is nothing to upload; what a commit can do instead is derive the raymarch's acceleration data: `dist`
 * (DEVICE, W*H*D floats) receives a compact copy of tex0.r. */
int sdfv_commit_distance(const sdfv_grid *grid, const float *tex0, float *dist, void *stream);
/* The y-pair volume from the compact distance volume (12 B/voxel of traffic, once per load: 0.03 ms at 256^3; 8 B/voxel of
 * memory; `pairs`: DEVICE, 8-byte aligned, 2 floats per voxel of the WHOLE grid).  NOT part of the fill: the fused fill stays
 * at 36 B/voxel. */
int sdfv_commit_pairs(const sdfv_grid *grid, const float *dist, float *pairs, void *stream);
/* The y-interleaved volume from the compact distance volume (8 B/voxel of traffic; `ilv`: DEVICE, 8-byte aligned, one float per
 * voxel of the WHOLE grid, H even): ilv[((z * H/2 + (y >> 1)) * W + x) * 2 + (y & 1)] = dist[z][y][x]. */
int sdfv_commit_interleaved(const sdfv_grid *grid, const float *dist, float *ilv, void *stream);
/* Which of the two a host that renders many frames per load should build for this grid on the current device:
 * the pair volume while its 8 B/voxel fit the last-level cache (MI355X: 256 MB -- up to 256^3 x 2), the interleaved volume
 * beyond (4K over 512^3: -10 % against either of the others; 1080p over 256^3: pairs -6 %, interleaved +-1 %); NONE for a grid
 * that is not cubic, on a device that is not gfx950, or when the hand-written loop is switched off (the march would never read
 * them).  Speed only. */
#define SDFV_MARCH_VOLUME_NONE 0u
#define SDFV_MARCH_VOLUME_PAIRS 1u
#define SDFV_MARCH_VOLUME_INTERLEAVED 2u
int sdfv_march_volume_advice(const sdfv_grid *grid, uint32_t *kind);

/* ---- raymarch over a z-sharded grid (multi-GPU; the consumer of the slab halo) ----
 * The grid stays sharded: rank r holds [ghost_lo][owned z_begin..z_end)[ghost_hi] as laid out for sdfv_slab_*.
 * A ray is marched by the rank that owns the cell it is in (clamp(floor(w), 0, D-1) in [z_begin, z_end); the
 * trilinear fetch then needs at most slice z_end = ghost_hi) and is handed to the z-neighbour, state and all, when
 * it leaves the slab.  z is monotonic along a ray, so after `world` rounds every ray has ended on exactly one rank;
 * that rank writes the pixel, all others leave it all-zero bits, and the image is their merge (OR, or integer sum, of the bit patterns).
 * The arithmetic and its order are those of sdfv_raymarch: the merged image is bit-identical to the single-GPU one.
 * Restrictions: loaded grids only (lod_dist_between_samples == 1), boxes large enough that a marching ray's
 * floor(w) stays in [-1, D-1] (1e-4 * N / size <= 0.25), one camera per call.  aux.normal needs the slices around its
 * four taps, which can be one further up than the march's own fetch: it is filled where a SECOND upper ghost slice
 * (ghost_hi = 2) makes them resident, and left (0,0,0) otherwise -- the one-voxel halo suffices for RGBA. */
typedef struct sdfv_ray_state {
    uint32_t pixel;     /* y * width + x */
    uint32_t iteration; /* sdfRaycast's loop counter i = tex0 fetches done so far */
    float    pos[3];    /* rayPos */
    float    t;         /* distanceFromOrigin */
} sdfv_ray_state;
/* One round on this rank.  in_states == NULL: first round (every pixel's primary ray; rays that start in another
 * rank's slab are left to that rank, and every pixel of rgba/aux is initialised).  Otherwise: continue the n_in rays
 * received from the neighbours.  Rays that leave the slab are appended to out_down / out_up (DEVICE, `capacity`
 * entries each; width*height always suffices) and counted in counters[0] / counters[1] (DEVICE, zeroed by the
 * caller).  rp describes the GLOBAL grid; slab gives dims and [z_begin, z_end); tex0/tex1 address the start of the
 * slab allocation (ghost_lo slices before z_begin, ghost_hi after z_end).  rgba: DEVICE height x width x 4. */
int sdfv_raymarch_slab(const sdfv_render_params *rp, const sdfv_grid *slab, uint32_t ghost_lo, uint32_t ghost_hi,
                       const float *tex0, const float *tex1, const sdfv_camera *camera, uint32_t width,
                       uint32_t height, const sdfv_ray_state *in_states, uint32_t n_in, float *rgba,
                       sdfv_march_aux *aux, sdfv_ray_state *out_down, sdfv_ray_state *out_up, uint32_t capacity,
                       uint32_t *counters, void *stream);

/* The same round with NOTHING for the host to read between rounds: ray lists are fixed-capacity buffers with their count
 * in band -- SDFV_RAY_BUFFER_HEADER_BYTES of header (word 0: the count, written by the round that fills the buffer) followed
 * by `capacity` sdfv_ray_state entries; sdfv_ray_buffer_bytes(capacity) bytes, DEVICE, 4-byte aligned.  in_lo / in_hi: the
 * buffers the lower / upper neighbour filled as its out_up / out_down (either may be NULL: no such neighbour); first_round
 * != 0: every pixel's primary ray instead (in_lo = in_hi = NULL).  out_down / out_up are cleared by the call itself.
 * `overflow` (DEVICE word, may be NULL) is set to 1 when a ray did not fit into `capacity` (width * height always suffices).
 * A whole march is `world` such rounds per rank with the buffers moved between neighbours by any stream-ordered copy
 * (sdfv_slab_march below does it over RCCL); capture-safe, no synchronisation. */
#define SDFV_RAY_BUFFER_HEADER_BYTES 16u
size_t sdfv_ray_buffer_bytes(uint32_t capacity);
int sdfv_raymarch_slab_round(const sdfv_render_params *rp, const sdfv_grid *slab, uint32_t ghost_lo, uint32_t ghost_hi,
                             const float *tex0, const float *tex1, const sdfv_camera *camera, uint32_t width,
                             uint32_t height, const void *in_lo, const void *in_hi, int first_round, float *rgba,
                             sdfv_march_aux *aux, void *out_down, void *out_up, uint32_t capacity, uint32_t *overflow,
                             void *stream);

/* ---- host-buffer conveniences (allocate, run, copy back, synchronise; PCIe-inclusive) ---- */
int sdfv_fill_grid_host(const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *grid,
                        float *tex0_host, float *tex1_host);
int sdfv_sample_points_host(const sdfv_demo_params *params, uint32_t sdf_id, const float *points_host, size_t n,
                            int distance_only, sdfv_sample *out_host);
int sdfv_normal_points_host(const sdfv_demo_params *params, uint32_t sdf_id, const float *points_host, size_t n,
                            float eps, int use_default, float *out_host);
int sdfv_mesh_postproc_host(const sdfv_demo_params *params, uint32_t sdf_id, sdfv_vertex *vertices_host, size_t n);
int sdfv_raymarch_host(const sdfv_render_params *rp, const float *tex0_host, const float *tex1_host,
                       const sdfv_camera *cameras, uint32_t n_cameras, uint32_t width, uint32_t height,
                       float *rgba_host, sdfv_march_aux *aux_host);

/* ---- multi-GPU: z-slab sharding with a one-voxel halo over RCCL (one process per GPU) ----------------------
 * The reference has no sharding (its loop carries "TODO: Cross-platform parallel iteration?", scene/sdf/mod.rs:174).
 * Rank r of `world` owns slices [z_begin, z_end) of the global grid.  Its textures carry one ghost slice per
 * existing z-neighbour:   [ghost_lo (rank > 0)] [owned z_begin .. z_end) [ghost_hi (rank < world-1)]
 * and the tex0/tex1 pointers handed to the two calls below address the START of that allocation.  After the
 * exchange ghost_lo holds the last slice of rank-1 and ghost_hi the first slice of rank+1, so trilinear
 * sampling (material.frag:42-45) across a slab boundary reads local memory.  Ends are not periodic.
 * SDFV_COMM_PERIODIC wraps rank 0 / world-1 around (both ghosts always present); with world = 1 every send is
 * matched by a receive of the same rank, which is how the single-GPU tests drive the RCCL path.
 * RCCL is loaded at run time (librccl.so.1), so hosts that never create a communicator do not need it. */
#define SDFV_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
#define SDFV_COMM_PERIODIC 1u
#define SDFV_COMM_HALO2    2u /* TWO ghost slices on the upper side: [ghost_lo (0/1)] [owned] [ghost_hi (0/2)], every rank
                               * sends its first two owned slices down.  What sdfNormal's taps need in sdfv_raymarch_slab
                               * (material.frag:73-80 reaches one slice beyond the march's own fetch). */
typedef struct sdfv_slab_comm sdfv_slab_comm;

/* Rank 0 makes the id; the HOST hands the 128 bytes to every rank by whatever channel it has. */
int sdfv_slab_comm_unique_id(unsigned char id_out[SDFV_COMM_ID_BYTES]);
/* Collective over all ranks (ncclCommInitRank on the current HIP device).  Owns a second HIP stream + events.
 * Hardware queues: HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The fill
 * step involves the caller's stream, the communicator's high-priority stream and RCCL's internal ones; when two of
 * them land on one queue the step serialises (measured 265 us instead of 106 at 256^3; which streams collide depends on
 * creation order -- e.g. a stream used before the process's first RCCL communicator came up).  Start the process with
 * GPU_MAX_HW_QUEUES=8 (read by the HIP runtime at start-up): every creation order measured is then fast. */
int sdfv_slab_comm_create(const unsigned char id[SDFV_COMM_ID_BYTES], int rank, int world, uint32_t flags,
                          sdfv_slab_comm **out);
int sdfv_slab_comm_destroy(sdfv_slab_comm *comm);
/* Ghost slices this rank's textures carry below / above the owned slices, and whether the step can be released by the
 * fill launch's own start signal on this device (hipStreamWaitValue32; otherwise the event form is used).  Any output
 * pointer may be NULL. */
int sdfv_slab_comm_info(const sdfv_slab_comm *comm, uint32_t *ghost_lo, uint32_t *ghost_hi, uint32_t *wait_value_capable);
/* What RCCL itself says about the communicator: ncclCommUserRank / ncclCommCount -- the number of ranks RCCL actually
 * connected (what a reader of a multi-GPU measurement needs to know first).  Either output pointer may be NULL. */
int sdfv_slab_comm_ranks(const sdfv_slab_comm *comm, int *rank, int *world);
/* The halo exchange alone, enqueued on `stream` (one ncclGroup of up to 4 sends + 4 receives straight on the textures:
 * the first 1 (2 with SDFV_COMM_HALO2) owned slices down, the last owned slice up). DEVICE pointers. */
int sdfv_slab_halo_exchange(sdfv_slab_comm *comm, const sdfv_grid *slab, float *tex0, float *tex1, void *stream);
/* One fill step of this rank = sdfv_fill_grid over the owned slab + the halo exchange, with the exchange hidden
 * behind the fill: `stream` runs the plain dense fill of the whole slab -- the same single launch as sdfv_fill_grid,
 * nothing before it -- while the communicator's high-priority stream, released by the fill launch's own start signal (or
 * by an event recorded on `stream`), (1) computes the boundary slices (the first 1 or 2 and the last owned one: under
 * 1 % of the slab, computed twice) straight into packed send buffers, (2) exchanges ONE RCCL message per neighbour and
 * direction carrying both textures' slices, (3) copies the received slices into the ghosts; `stream` then waits for
 * that chain, which is shorter than the fill it runs under.  From 2^26 voxels per slab the messages go per texture
 * straight out of / into the textures instead (SDFV_STEP_UNPACKED; SDFV_OPT_SLAB_STEP_FORM pins either).  Slabs too
 * thin to have an interior, or whose rows do not fill whole workgroups, are filled and then exchanged.  On return
 * everything is enqueued; work later put on `stream` sees owned and ghost slices complete. */
int sdfv_slab_fill_step(sdfv_slab_comm *comm, const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *slab,
                        float *tex0, float *tex1, void *stream);
/* The same step with the fused commit (sdfv_fill_grid_commit per rank): `dist` (DEVICE, one float per voxel of the slab
 * allocation INCLUDING its ghost slices, same slice layout as the textures; NULL = sdfv_slab_fill_step) receives the
 * compact distance volume of the owned slices in the fill's own pass (36 B/voxel stored, the textures streamed past L2)
 * and, once the halo has arrived, tex0.r of the ghost slices. */
int sdfv_slab_fill_step_commit(sdfv_slab_comm *comm, const sdfv_demo_params *params, uint32_t sdf_id,
                               const sdfv_grid *slab, float *tex0, float *tex1, float *dist, void *stream);

/* The whole sharded march of one frame over the communicator, enqueued in ONE call: `world` rounds of
 * sdfv_raymarch_slab_round back to back on `stream`, between them one RCCL group per rank that sends its two ray buffers
 * (whole, fixed size: the count is in their header) to the z-neighbours and receives theirs.  No counter read-back, no host
 * round trip, nothing synchronises.  tex0 / tex1: the slab allocation as sdfv_slab_fill_step leaves it (ghosts filled; the
 * communicator must not be periodic).  scratch: DEVICE, 16-byte aligned, sdfv_slab_march_scratch_bytes(capacity) bytes (four
 * ray buffers); capacity: rays a list can hold -- width * height always suffices, a message is 16 + 24 * capacity bytes per
 * neighbour and round whatever it carries, so a caller that knows its views can bound it.  status: DEVICE, 2 words --
 * [0] = 1 if any list overflowed (the image is then incomplete: repeat with a larger capacity), [1] = rays left over after
 * the last round (0 otherwise).  SDFV_MARCH_MERGE: afterwards the ranks' images (each pixel written by the one rank its ray
 * ended on, zero bits elsewhere) are summed as integers in place (ncclAllReduce), so every rank holds the whole frame --
 * bit-identical to sdfv_raymarch over the whole grid (aux.normal as described for sdfv_raymarch_slab; aux.depth of pixels no
 * rank reports stays 0 in the merged record, as with sdfv_raymarch_slab). */
#define SDFV_MARCH_MERGE 1u
size_t sdfv_slab_march_scratch_bytes(uint32_t capacity);
int sdfv_slab_march(sdfv_slab_comm *comm, const sdfv_render_params *rp, const sdfv_grid *slab, const float *tex0,
                    const float *tex1, const sdfv_camera *camera, uint32_t width, uint32_t height, float *rgba,
                    sdfv_march_aux *aux, void *scratch, size_t scratch_bytes, uint32_t capacity, uint32_t flags,
                    uint32_t *status, void *stream);

/* ---- config 5's collectives on the library communicator ("gather of RGBA tiles to rank 0", replicas by
 * all-gather of slabs).  Every rank of the communicator calls; everything is enqueued on `stream`, nothing synchronises. ---- */

/* The inverse of sdfv_raymarch_bands' output layout, on one device (no communicator): moves the band set `part` (DEVICE,
 * n_cameras x sdfv_band_rows_ex(height, band_first, band_step, band_height) x width x channels floats) to its rows of the images `out` (DEVICE,
 * n_cameras x height x width x channels).  channels: 4 for rgba, 1 for the depth plane, 18 for the aux record.  What
 * sdfv_comm_gather_bands runs per received set; a host with another transport (MPI, torch.distributed) calls it itself. */
int sdfv_bands_scatter(const float *part, uint32_t band_first, uint32_t band_step, uint32_t band_height, uint32_t n_cameras,
                       uint32_t width, uint32_t height, uint32_t channels, float *out, void *stream);
/* The image-tile split's gather: rank r passes the band set it rendered with (band_first = r, band_step = world, band_height); rank `dst` receives every other rank's set whole (one message per peer, all xGMI links at once) into `scratch` and
 * assembles the n_cameras images in `out` (DEVICE, n_cameras x height x width x channels floats; ignored elsewhere, may be
 * NULL).  scratch: DEVICE, 16-byte aligned, sdfv_comm_gather_bands_scratch_bytes() bytes on dst (0 elsewhere).  64 cameras x
 * 1080p are 2.1 GB into ONE rank: tens of milliseconds over seven links against the 0.3 ms a rank spends rendering its share --
 * a host that can consume the bands where they were rendered should. */
size_t sdfv_comm_gather_bands_scratch_bytes(const sdfv_slab_comm *comm, int dst, uint32_t band_height, uint32_t n_cameras,
                                            uint32_t width, uint32_t height, uint32_t channels);
int sdfv_comm_gather_bands(sdfv_slab_comm *comm, const float *part, uint32_t band_height, uint32_t n_cameras, uint32_t width,
                           uint32_t height, uint32_t channels, int dst, float *out, void *scratch, size_t scratch_bytes, void *stream);
/* The camera split's gather: rank r rendered cameras [n_cameras * r / world, n_cameras * (r + 1) / world) whole (`part`); dst
 * receives them in place, in camera order (`out`: n_cameras x height x width x channels floats).  No scratch. */
int sdfv_comm_gather_cameras(sdfv_slab_comm *comm, const float *part, uint32_t n_cameras, uint32_t width, uint32_t height,
                             uint32_t channels, int dst, float *out, void *stream);
/* Replicas from a z-sharded grid: every rank ends up with the whole dims[0] x dims[1] x dims[2] grid.  z_bounds: HOST, world + 1
 * entries, rank r owns slices [z_bounds[r], z_bounds[r + 1]) (depths may differ); tex0_owned / tex1_owned / dist_owned: DEVICE,
 * this rank's FIRST OWNED slice (past any ghost slice; dist_owned and out_dist may both be NULL); out0 / out1 / out_dist:
 * DEVICE, the whole grid.  (For the demo SDF a redundant local fill of the replica is cheaper than moving it -- 34 GB at 8 TB/s
 * against 30 GB per rank over xGMI, SURVEY 8(e) -- this is for grids whose samples are not free to recompute.) */
int sdfv_comm_allgather_slabs(sdfv_slab_comm *comm, const uint32_t dims[3], const uint32_t *z_bounds, const float *tex0_owned,
                              const float *tex1_owned, const float *dist_owned, float *out0, float *out1, float *out_dist,
                              void *stream);

/* Makes `stream` wait for the most recent exchange the communicator has enqueued (a no-op when there is none).  Needed
 * only after steps taken with SDFV_STEP_DEFER_JOIN: before anything put on `stream` reads the ghost slices of the
 * textures (or the ghost slices' share of the distance volume).  The owned slices never need it: the flag is honoured
 * only where the caller's stream fills every owned slice itself (packed messages). */
int sdfv_slab_comm_join(sdfv_slab_comm *comm, void *stream);

/* ---- header-only convenience forms (static inline over the exported calls above; NOT symbols of the library).  A binder in
 * another language binds sdfv_raymarch_ex / sdfv_fill_grid_pass_ex and writes these few lines in its own idiom. ---- */
#if defined(__GNUC__) || defined(__clang__)
#define SDFV_INLINE static inline __attribute__((unused))
#else
#define SDFV_INLINE static inline
#endif
SDFV_INLINE int sdfv_fill_grid_pass_dist(const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *grid, uint32_t step,
                                         const float *changed_box, float *tex0, float *tex1, float *dist, void *stream) {
    return sdfv_fill_grid_pass_ex(params, sdf_id, grid, step, changed_box, tex0, tex1, dist, 0u, stream);
}
SDFV_INLINE int sdfv_fill_grid_pass(const sdfv_demo_params *params, uint32_t sdf_id, const sdfv_grid *grid, uint32_t step,
                                    const float *changed_box, float *tex0, float *tex1, void *stream) {
    return sdfv_fill_grid_pass_ex(params, sdf_id, grid, step, changed_box, tex0, tex1, (float *)0, 0u, stream);
}
/* The most general positional form: any of the three acceleration volumes may be NULL. */
SDFV_INLINE int sdfv_raymarch_volumes(const sdfv_render_params *rp, const float *tex0, const float *tex1, const float *dist,
                                      const float *pairs, const float *ilv, const sdfv_camera *cameras, uint32_t n_cameras,
                                      uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, float *rgba, float *depth,
                                      sdfv_march_aux *aux, void *stream) {
    sdfv_march_desc d;
    d.size = (uint32_t)sizeof d;
    d.reserved = 0;
    d.rp = rp;
    d.tex0 = tex0;
    d.tex1 = tex1;
    d.dist = dist;
    d.pairs = pairs;
    d.ilv = ilv;
    d.cameras = cameras;
    d.n_cameras = n_cameras;
    d.width = width;
    d.height = height;
    d.y0 = y0;
    d.y1 = y1;
    d.band_first = 0;
    d.band_step = 0;
    d.band_height = 0;
    d.rgba = rgba;
    d.depth = depth;
    d.aux = aux;
    d.rgba8 = (uint32_t *)0;
    return sdfv_raymarch_ex(&d, stream);
}
/* material.frag main() over tex0.r in place: the batched raymarch SURVEY 8(b) names */
SDFV_INLINE int sdfv_raymarch(const sdfv_render_params *rp, const float *tex0, const float *tex1, const sdfv_camera *cameras,
                              uint32_t n_cameras, uint32_t width, uint32_t height, uint32_t y0, uint32_t y1, float *rgba,
                              sdfv_march_aux *aux, void *stream) {
    return sdfv_raymarch_volumes(rp, tex0, tex1, (const float *)0, (const float *)0, (const float *)0, cameras, n_cameras, width,
                                 height, y0, y1, rgba, (float *)0, aux, stream);
}
/* ... over the compact distance volume */
SDFV_INLINE int sdfv_raymarch_accel(const sdfv_render_params *rp, const float *tex0, const float *tex1, const float *dist,
                                    const sdfv_camera *cameras, uint32_t n_cameras, uint32_t width, uint32_t height, uint32_t y0,
                                    uint32_t y1, float *rgba, sdfv_march_aux *aux, void *stream) {
    return sdfv_raymarch_volumes(rp, tex0, tex1, dist, (const float *)0, (const float *)0, cameras, n_cameras, width, height, y0,
                                 y1, rgba, (float *)0, aux, stream);
}
/* ... with the depth plane */
SDFV_INLINE int sdfv_raymarch_depth(const sdfv_render_params *rp, const float *tex0, const float *tex1, const float *dist,
                                    const sdfv_camera *cameras, uint32_t n_cameras, uint32_t width, uint32_t height, uint32_t y0,
                                    uint32_t y1, float *rgba, float *depth, sdfv_march_aux *aux, void *stream) {
    return sdfv_raymarch_volumes(rp, tex0, tex1, dist, (const float *)0, (const float *)0, cameras, n_cameras, width, height, y0,
                                 y1, rgba, depth, aux, stream);
}
/* ... over the y-pair volume */
SDFV_INLINE int sdfv_raymarch_pairs(const sdfv_render_params *rp, const float *tex0, const float *tex1, const float *dist,
                                    const float *pairs, const sdfv_camera *cameras, uint32_t n_cameras, uint32_t width,
                                    uint32_t height, uint32_t y0, uint32_t y1, float *rgba, float *depth, sdfv_march_aux *aux,
                                    void *stream) {
    return sdfv_raymarch_volumes(rp, tex0, tex1, dist, pairs, (const float *)0, cameras, n_cameras, width, height, y0, y1, rgba,
                                 depth, aux, stream);
}
/* The balanced image-tile split: the 16-row tile bands band_first, band_first + band_step, ... (band_step >= 1) */
SDFV_INLINE int sdfv_raymarch_bands(const sdfv_render_params *rp, const float *tex0, const float *tex1, const float *dist,
                                    const float *pairs, const float *ilv, const sdfv_camera *cameras, uint32_t n_cameras,
                                    uint32_t width, uint32_t height, uint32_t band_first, uint32_t band_step, float *rgba,
                                    float *depth, sdfv_march_aux *aux, void *stream) {
    sdfv_march_desc d;
    d.size = (uint32_t)sizeof d;
    d.reserved = 0;
    d.rp = rp;
    d.tex0 = tex0;
    d.tex1 = tex1;
    d.dist = dist;
    d.pairs = pairs;
    d.ilv = ilv;
    d.cameras = cameras;
    d.n_cameras = n_cameras;
    d.width = width;
    d.height = height;
    d.y0 = 0;
    d.y1 = height;
    d.band_first = band_first;
    d.band_step = band_step;
    /* in the descriptor band_step 0 means "rows [y0, y1)", not a band set: a band height WITH step 0 is what the library
     * refuses, so the error (and sdfv_last_error's text) comes from there */
    d.band_height = band_step == 0 ? 16u : 0u;
    d.rgba = rgba;
    d.depth = depth;
    d.aux = aux;
    d.rgba8 = (uint32_t *)0;
    return sdfv_raymarch_ex(&d, stream);
}

#ifdef __cplusplus
}
#endif
#endif
