/*
 * sdf_provider.h -- the reference's per-point SDF ABI, exported with the reference's own (unprefixed) symbol
 * names by libsdfdemo_provider.so so that it can stand where `demo_sdf.wasm` / a native SDF library stands.
 *
 * Specification: reference src/sdf/wasm/mod.rs:1-38 (prose), src/sdf/ffi.rs:42-337 (the export side this
 * mirrors, function by function), src/sdf/wasm/native.rs:163-521 (the host side's byte layouts).
 * Differences from the wasm32 build of the reference are only those of the pointer width: pointers and
 * lengths are native (8 bytes) instead of 4, exactly what `#[repr(C)]` gives the same Rust types when
 * ffi.rs is compiled for x86-64.
 *
 * Ownership (ffi.rs:52-55,84-90): every function returns heap memory owned by the library; the caller
 * copies what it needs and hands the SAME pointer to the matching *_free.  Errors (ffi.rs:46-49):
 * unknown ids print "Failed to find SDF with ID <id>" on stderr and return zeros / NULL payloads.
 * The registry is thread-local (ffi.rs:15-17): call init() on the thread that uses the SDF.
 * One deliberate difference: a String payload handed to set_parameter() is COPIED and stays the caller's; the
 * reference reclaims it as a Rust Vec (ffi.rs:228-229), which is only sound when caller and callee share an
 * allocator (its wasm host writes the bytes at a fixed scratch address instead, wasm/native.rs:408-411).
 *
 * All arithmetic of libsdfdemo_provider.so runs on the GPU (sample(): one-point batches of the libsdfgrid kernels;
 * sample_batch(): one batch per call): this ABI is the compatibility path; the hot path is the batched API in sdfgrid.h.
 * The CONSUMER side of this ABI -- any library that exports it becomes an SDFSurface of the host mirror, loaded into the
 * viewer's device textures -- is sdf-viewer_amd/host/provider_sdf.hpp (ProviderSDF) + SDFViewer::update's ingest path.
 */
#ifndef SDF_PROVIDER_H
#define SDF_PROVIDER_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SDFVec3 { float x, y, z; } SDFVec3;                 /* cgmath Vector3<f32>, repr(C) */
typedef struct SDFBoundingBox { SDFVec3 min, max; } SDFBoundingBox; /* [Vector3<f32>; 2], 24 bytes */
typedef struct SDFSample {                                          /* src/sdf/mod.rs:104-118, 28 bytes */
    float distance;
    SDFVec3 color;
    float metallic, roughness, occlusion;
} SDFSample;

/* ffi.rs:72-80 PointerLength<T>: pointer + length IN BYTES */
typedef struct PointerLength { const void *ptr; size_t len_bytes; } PointerLength;

/* ffi.rs:164-186 #[repr(C, u32)] enum SDFParamKindC */
typedef struct SDFParamKindC {
    uint32_t tag; /* 0 Boolean, 1 Int, 2 Float, 3 String */
    union {
        struct { int32_t range_start, range_end, step; } int_;   /* Range<i32> + step */
        struct { float range_start, range_end, step; } float_;   /* Range<f32> + step */
        struct { PointerLength choices; } string_;               /* PointerLength<PointerLength<u8>> */
    } v;
} SDFParamKindC;

/* ffi.rs:202-210 #[repr(C, u32)] enum SDFParamValueC */
typedef struct SDFParamValueC {
    uint32_t tag; /* 0 Boolean, 1 Int, 2 Float, 3 String */
    union {
        bool boolean;
        int32_t int_;
        float float_;
        PointerLength string_; /* PointerLength<u8> */
    } v;
} SDFParamValueC;

/* ffi.rs:148-162 */
typedef struct SDFParamC {
    uint32_t id;
    PointerLength name;        /* utf-8, not NUL terminated */
    SDFParamKindC kind;
    SDFParamValueC value;
    PointerLength description;
} SDFParamC;

/* Box<Result<(), PointerLength<u8>>> as the host reads it (wasm/native.rs:395-445): tag 0 = Ok, 1 = Err */
typedef struct SDFSetParameterResult { uint32_t tag; PointerLength error; } SDFSetParameterResult;
/* Box<Option<[Vector3<f32>; 2]>> as the host reads it (wasm/native.rs:463-489): tag 0 = None, 1 = Some */
typedef struct SDFChangedResult { uint32_t tag; SDFBoundingBox bounds; } SDFChangedResult;

void init(void);                                                     /* demo/ffi.rs:5-8: set_root_sdf(SDFDemo::default()) */
/* extension: (re)build the registry from the demo's CLI flags, e.g. {"-t","normal","-s","0.9"}; 0 on success */
int  init_with_args(int argc, const char *const *argv);

/* extension, OPTIONAL for any provider: how many host threads may call sample() at once (a provider whose registry is
 * thread-local, like ffi.rs:15-17's, does not export it or -- libsdfdemo_provider.so -- answers 1).  The host mirror's SDFViewer::update
 * samples a host-only SDF on that many threads (sdf-viewer_amd/host/provider_sdf.hpp, sdf_viewer_ingest.cpp). */
uint32_t sample_concurrency(void);

/* extension, OPTIONAL for any provider: the "Batched sampling" the trait leaves as a TODO (src/sdf/mod.rs:39).  out[i] =
 * *sample(sdf_id, points[i], distance_only) for i in [0, n); BOTH arrays are the caller's (nothing to free): one call and no
 * allocation per point.  A host that finds the export samples through it (ProviderSDF::sample_batch); an unknown id zeroes out. */
void sample_batch(uint32_t sdf_id, const SDFVec3 *points, size_t n, bool distance_only, SDFSample *out);

SDFBoundingBox *bounding_box(uint32_t sdf_id);                       /* ffi.rs:42-50 */
void bounding_box_free(SDFBoundingBox *ret);                         /* ffi.rs:52-55 */
SDFSample *sample(uint32_t sdf_id, SDFVec3 p, bool distance_only);   /* ffi.rs:57-65 */
void sample_free(SDFSample *ret);                                    /* ffi.rs:67-70 */
PointerLength *children(uint32_t sdf_id);                            /* ffi.rs:110-122: u32 ids */
void children_free(PointerLength *ret);                              /* ffi.rs:124-127 */
PointerLength *name(uint32_t sdf_id);                                /* ffi.rs:131-141 */
void name_free(PointerLength *ret);                                  /* ffi.rs:143-146 */
PointerLength *parameters(uint32_t sdf_id);                          /* ffi.rs:234-256: SDFParamC[] */
void parameters_free(PointerLength *ret);                            /* ffi.rs:258-283 */
SDFSetParameterResult *set_parameter(uint32_t sdf_id, uint32_t param_id, SDFParamValueC value); /* ffi.rs:285-298 */
void set_parameter_free(SDFSetParameterResult *ret);                 /* ffi.rs:300-303 */
SDFChangedResult *changed(uint32_t sdf_id);                          /* ffi.rs:305-315 */
void changed_free(SDFChangedResult *ret);                            /* ffi.rs:317-320 */
SDFVec3 *normal(uint32_t sdf_id, SDFVec3 p, float eps);              /* ffi.rs:322-332: eps <= 0 -> None */
void normal_free(SDFVec3 *ret);                                      /* ffi.rs:334-337 */

#ifdef __cplusplus
}
#endif
#endif
