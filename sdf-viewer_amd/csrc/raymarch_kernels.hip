// raymarch_kernels.hip -- per-pixel sphere tracing of the voxel grid on gfx950:
// the fragment shader of the reference (src/app/scene/sdf/material.frag, whole file) as a HIP kernel.
//
// One thread per pixel.  A 64-lane wave covers an 8x8 pixel tile (neighbouring rays walk neighbouring
// texels, so the gathers of a trilinear fetch land in few cache lines); a 256-thread workgroup covers
// 16x16; blockIdx.z is the camera.  The march loop is wave-synchronous: each iteration ballots the lanes
// still marching and the wave leaves the loop as soon as the ballot is empty (early ray termination)
// instead of running the shader's fixed 255-iteration bound.
//
// texture(sampler3D) is restated in full fp32 (GL texel-centre convention, MirroredRepeat, mix() along
// x, then y, then z): hardware texture filtering uses low-precision fixed-point weights and would not
// hold the 1e-4 RGBA tolerance, so no hipTextureObject is used.  The march reads only tex0.r (or its
// compact copy); the full tex0/tex1 texels are fetched once, at the hit.
//
// Ray set-up: the reference rasterises the bbox cube and the fragment's `pos` is a point on its
// surface (scene/sdf/mod.rs:254-282, material.rs:75-81 Cull::None).  Here `pos` comes from a ray/AABB
// slab test through the pixel centre: the entry point when the camera is outside the box, the exit
// point when it is inside; pixels whose ray misses the box are transparent.
//
// What profiling says (profiles/, tools/wave_timing.py): a frame is latency-bound by its longest wave (one
// grazing ray, up to 255 dependent iterations, ~8 cycles per instruction for a lone wave on its SIMD) and a
// batch of cameras is VALU-issue-bound.  Both regimes pay per INSTRUCTION, so the fast kernels below
// spend their effort on removing instructions from the per-iteration path without changing a single bit
// of the result (see march_fast); the general kernel keeps the shader's structure for every other case.
#include "raymarch_kernels.h"

#include <hip/hip_runtime.h>
#include <string.h>

namespace sdfv {
namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 madd(V3 a, V3 d, float t) { return mk(a.x + d.x * t, a.y + d.y * t, a.z + d.z * t); }
__device__ __forceinline__ float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) {
    float l = length(a);
    return mk(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// GL MIRRORED_REPEAT on a texel index (general form).
__device__ __forceinline__ uint32_t mirror_index(int i, int n) {
    if (i >= 0 && i < n) return (uint32_t)i;
    int period = 2 * n;
    int m = i % period;
    if (m < 0) m += period;
    return (uint32_t)(m < n ? m : period - 1 - m);
}

struct Tex {
    const float4* data;
    int w, h, d;
};

struct Footprint {  // the 8 texel offsets and 3 weights of one LINEAR fetch
    uint32_t o000, o100, o010, o110, o001, o101, o011, o111;
    float ax, ay, az;
};

// FAST: floor(u) is known to be in [-1, N-1] on every axis, where MirroredRepeat == clamp.
template <bool FAST>
__device__ __forceinline__ Footprint footprint(const Tex& t, V3 p01) {
    float u = p01.x * (float)t.w - 0.5f, v = p01.y * (float)t.h - 0.5f, w = p01.z * (float)t.d - 0.5f;
    float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    Footprint f;
    f.ax = u - fu; f.ay = v - fv; f.az = w - fw;
    int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
    uint32_t i0m, i1m, j0m, j1m, k0m, k1m;
    const uint32_t sy = (uint32_t)t.w, sz = (uint32_t)t.w * (uint32_t)t.h;
    if (FAST) {
        i0m = (uint32_t)max(i0, 0); i1m = (uint32_t)min(i0 + 1, t.w - 1);
        j0m = (uint32_t)max(j0, 0) * sy; j1m = (uint32_t)min(j0 + 1, t.h - 1) * sy;
        k0m = (uint32_t)max(k0, 0) * sz; k1m = (uint32_t)min(k0 + 1, t.d - 1) * sz;
    } else {
        i0m = mirror_index(i0, t.w); i1m = mirror_index(i0 + 1, t.w);
        j0m = mirror_index(j0, t.h) * sy; j1m = mirror_index(j0 + 1, t.h) * sy;
        k0m = mirror_index(k0, t.d) * sz; k1m = mirror_index(k0 + 1, t.d) * sz;
    }
    f.o000 = k0m + j0m + i0m; f.o100 = k0m + j0m + i1m;
    f.o010 = k0m + j1m + i0m; f.o110 = k0m + j1m + i1m;
    f.o001 = k1m + j0m + i0m; f.o101 = k1m + j0m + i1m;
    f.o011 = k1m + j1m + i0m; f.o111 = k1m + j1m + i1m;
    return f;
}

__device__ __forceinline__ float trilerp(float t000, float t100, float t010, float t110, float t001, float t101,
                                         float t011, float t111, float ax, float ay, float az) {
    float c00 = mixf(t000, t100, ax), c10 = mixf(t010, t110, ax);
    float c01 = mixf(t001, t101, ax), c11 = mixf(t011, t111, ax);
    return mixf(mixf(c00, c10, ay), mixf(c01, c11, ay), az);
}

// p01 = (p - sdfBoundsMin) / (sdfBoundsMax - sdfBoundsMin), material.frag:44.  XF >= 1: every extent is an
// exact power of two (the demo's [-1,1]^3: 2.0), so dividing equals multiplying by the exact reciprocal,
// bit for bit, and three IEEE divides (~12 instructions each) leave the per-step dependency chain.
template <int XF>
__device__ __forceinline__ V3 to_p01(const RaymarchArgs& a, V3 p) {
    if (XF >= 1)
        return mk((p.x - a.rp.bounds_min[0]) * a.inv_bsize[0], (p.y - a.rp.bounds_min[1]) * a.inv_bsize[1],
                  (p.z - a.rp.bounds_min[2]) * a.inv_bsize[2]);
    return mk((p.x - a.rp.bounds_min[0]) / a.bsize[0], (p.y - a.rp.bounds_min[1]) / a.bsize[1],
              (p.z - a.rp.bounds_min[2]) / a.bsize[2]);
}

// sdfSampleRawNearest's snapped coordinate + NEAREST fetch, material.frag:27-36
__device__ __forceinline__ uint32_t nearest_offset(const RaymarchArgs& a, const Tex& t, V3 p01) {
    float rx = (float)t.w / a.rp.lod_dist_between_samples;
    float ry = (float)t.h / a.rp.lod_dist_between_samples;
    float rz = (float)t.d / a.rp.lod_dist_between_samples;
    float qx = roundf(p01.x * rx) / rx, qy = roundf(p01.y * ry) / ry, qz = roundf(p01.z * rz) / rz;
    int i = (int)floorf(qx * (float)t.w), j = (int)floorf(qy * (float)t.h), k = (int)floorf(qz * (float)t.d);
    return (mirror_index(k, t.d) * (uint32_t)t.h + mirror_index(j, t.h)) * (uint32_t)t.w + mirror_index(i, t.w);
}

// sdfSampleRawInterp(.., p).r only -- what the march and the normal need.  material.frag:42-53
template <bool LINEAR, int XF, bool FAST>
__device__ __forceinline__ float sample_r(const RaymarchArgs& a, const Tex& t, V3 p) {
    V3 q = to_p01<XF>(a, p);
    const float* base = reinterpret_cast<const float*>(t.data);
    if (LINEAR) {
        Footprint f = footprint<FAST>(t, q);
        float t000 = base[(uint64_t)f.o000 * 4], t100 = base[(uint64_t)f.o100 * 4];
        float t010 = base[(uint64_t)f.o010 * 4], t110 = base[(uint64_t)f.o110 * 4];
        float t001 = base[(uint64_t)f.o001 * 4], t101 = base[(uint64_t)f.o101 * 4];
        float t011 = base[(uint64_t)f.o011 * 4], t111 = base[(uint64_t)f.o111 * 4];
        return trilerp(t000, t100, t010, t110, t001, t101, t011, t111, f.ax, f.ay, f.az);
    }
    return base[(uint64_t)nearest_offset(a, t, q) * 4];
}

// LINEAR tex0.r read from either tex0 itself (STRIDE 4) or the compact distance volume (STRIDE 1).
template <int XF, bool FAST, int STRIDE>
__device__ __forceinline__ float sample_r_linear(const RaymarchArgs& a, const float* __restrict__ vol, const Tex& t, V3 p) {
    const Footprint f = footprint<FAST>(t, to_p01<XF>(a, p));
    const float t000 = vol[(uint64_t)f.o000 * STRIDE], t100 = vol[(uint64_t)f.o100 * STRIDE];
    const float t010 = vol[(uint64_t)f.o010 * STRIDE], t110 = vol[(uint64_t)f.o110 * STRIDE];
    const float t001 = vol[(uint64_t)f.o001 * STRIDE], t101 = vol[(uint64_t)f.o101 * STRIDE];
    const float t011 = vol[(uint64_t)f.o011 * STRIDE], t111 = vol[(uint64_t)f.o111 * STRIDE];
    return trilerp(t000, t100, t010, t110, t001, t101, t011, t111, f.ax, f.ay, f.az);
}

template <bool LINEAR, int XF, bool FAST>
__device__ __forceinline__ float4 sample_rgba(const RaymarchArgs& a, const Tex& t, V3 p) {
    V3 q = to_p01<XF>(a, p);
    if (LINEAR) {
        Footprint f = footprint<FAST>(t, q);
        float4 t000 = t.data[f.o000], t100 = t.data[f.o100], t010 = t.data[f.o010], t110 = t.data[f.o110];
        float4 t001 = t.data[f.o001], t101 = t.data[f.o101], t011 = t.data[f.o011], t111 = t.data[f.o111];
        float4 r;
        r.x = trilerp(t000.x, t100.x, t010.x, t110.x, t001.x, t101.x, t011.x, t111.x, f.ax, f.ay, f.az);
        r.y = trilerp(t000.y, t100.y, t010.y, t110.y, t001.y, t101.y, t011.y, t111.y, f.ax, f.ay, f.az);
        r.z = trilerp(t000.z, t100.z, t010.z, t110.z, t001.z, t101.z, t011.z, t111.z, f.ax, f.ay, f.az);
        r.w = trilerp(t000.w, t100.w, t010.w, t110.w, t001.w, t101.w, t011.w, t111.w, f.ax, f.ay, f.az);
        return r;
    }
    return t.data[nearest_offset(a, t, q)];
}

// sdfOutOfBoundsDist, material.frag:83-88.  SYMM: sdfBoundsMin == -sdfBoundsMax (the demo's box); then
// max(min - p, p - max) == |p| - max exactly (the larger operand is always the one equal to |p| - max),
// which is one subtraction with an abs modifier per axis instead of two subtractions and a max.
template <bool SYMM>
__device__ __forceinline__ float oob_dist(const RaymarchArgs& a, V3 p) {
    if (SYMM) {
        return fmaxf(fabsf(p.x) - a.rp.bounds_max[0],
                     fmaxf(fabsf(p.y) - a.rp.bounds_max[1], fabsf(p.z) - a.rp.bounds_max[2]));
    }
    float ox = fmaxf(a.rp.bounds_min[0] - p.x, p.x - a.rp.bounds_max[0]);
    float oy = fmaxf(a.rp.bounds_min[1] - p.y, p.y - a.rp.bounds_max[1]);
    float oz = fmaxf(a.rp.bounds_min[2] - p.z, p.z - a.rp.bounds_max[2]);
    return fmaxf(ox, fmaxf(oy, oz));
}

// sdfRaycast's loop (material.frag:97-126) for the LINEAR filter, with everything the loop does not need
// taken out of it.  Per iteration and lane: out-of-bounds test, texel coordinates, (re)fetch, trilinear
// mix, hit test, advance.
//  * Predicated straight-line code with ONE branch (the re-fetch): lanes that stopped no longer commit
//    results, so the wave does not pay exec-mask bookkeeping for the shader's nested if/else.
//  * One-cell register cache: sphere tracing takes its smallest steps exactly on the longest rays
//    (grazing a surface), so consecutive samples usually fall in the same texel cell; the 8 corner values
//    live in registers and are re-fetched only when floor(u,v,w) changes.
//  * MirroredRepeat needs no modulo: a marching ray is within 1e-4 of the box, hence (the launcher checks
//    1e-4 * N / size <= 0.25 per axis before selecting this kernel) floor(u) is in [-1, N-1], where
//    mirror(i) == clamp(i, 0, N-1).
//  * XF == 2: extents AND texture sizes are powers of two, so ((p-min)*inv)*N == (p-min)*(inv*N) exactly.
//  * No per-iteration status/step counters: a stopped lane's ray_pos no longer moves, so afterwards
//    "out of bounds" is re-derived from it (oob(ray_pos) > 1e-4 <=> it stopped on that test), and the
//    step count is 1 + the last iteration the lane sampled in.
//  * STRIDE = 4 reads tex0.r in place, STRIDE = 1 the compact distance volume (sdfv_commit_distance), where
//    x-neighbours are adjacent floats and each (y, z) pair of corners is ONE 8-byte load.
//  * T: accumulate distanceFromOrigin (only the aux record consumes it).
// Values and operation order are exactly those of sample_r() / the oracle; only redundant work is skipped.
template <int XF, bool SYMM, int STRIDE, bool T>
__device__ __forceinline__ void march_fast(const RaymarchArgs& a, const float* __restrict__ vol, const Tex& t,
                                           V3 ray_dir, bool covered, V3& ray_pos, float& dist_from_origin,
                                           int& status, int& steps, int& iterations) {
    const float fw_ = (float)t.w, fh_ = (float)t.h, fd_ = (float)t.d;
    const float kx = a.inv_bsize[0] * fw_, ky = a.inv_bsize[1] * fh_, kz = a.inv_bsize[2] * fd_;  // exact if XF == 2
    const int wm1 = t.w - 1, hm1 = t.h - 1, dm1 = t.d - 1;
    const uint32_t sy = (uint32_t)t.w, sz = (uint32_t)t.w * (uint32_t)t.h;
    bool marching = covered;
    int last_i = -1;
    float cfu = -4.0f, cfv = -4.0f, cfw = -4.0f;  // never a valid floor(u) of a marching lane
    float t000 = 0.0f, t100 = 0.0f, t010 = 0.0f, t110 = 0.0f, t001 = 0.0f, t101 = 0.0f, t011 = 0.0f, t111 = 0.0f;
    for (int i = 0; i < 255; ++i) {
        // Stop condition: out of bounds (material.frag:106-109)
        marching = marching && !(oob_dist<SYMM>(a, ray_pos) > 1e-4f);
        if (__ballot(marching) == 0ull) break;  // wave-level early termination
        ++iterations;
        last_i = marching ? i : last_i;
        float u, v, w;
        if (XF == 2) {
            u = (ray_pos.x - a.rp.bounds_min[0]) * kx - 0.5f;
            v = (ray_pos.y - a.rp.bounds_min[1]) * ky - 0.5f;
            w = (ray_pos.z - a.rp.bounds_min[2]) * kz - 0.5f;
        } else {
            const V3 q = to_p01<XF>(a, ray_pos);
            u = q.x * fw_ - 0.5f; v = q.y * fh_ - 0.5f; w = q.z * fd_ - 0.5f;
        }
        const float fu = floorf(u), fv = floorf(v), fw = floorf(w);
        const float ax = u - fu, ay = v - fv, az = w - fw;
        if (marching && (fu != cfu || fv != cfv || fw != cfw)) {
            cfu = fu; cfv = fv; cfw = fw;
            const int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
            const uint32_t i0c = (uint32_t)max(i0, 0), i1c = (uint32_t)min(i0 + 1, wm1);
            const uint32_t j0c = (uint32_t)max(j0, 0) * sy, j1c = (uint32_t)min(j0 + 1, hm1) * sy;
            const uint32_t k0c = (uint32_t)max(k0, 0) * sz, k1c = (uint32_t)min(k0 + 1, dm1) * sz;
            const uint32_t r00 = k0c + j0c, r10 = k0c + j1c, r01 = k1c + j0c, r11 = k1c + j1c;
            if (STRIDE == 1 && wm1 >= 1) {
                // x-neighbours are adjacent floats in the distance volume: one 8-byte load per (y, z) pair.
                // b = clamp(i0, 0, W-2) makes {b, b+1} cover {i0c, i1c} also where the clamp folds them together.
                const uint32_t b = (uint32_t)min(max(i0, 0), wm1 - 1);
                const bool lo_is_x = i0c == b, hi_is_y = i1c == b + 1;
                typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
                const f2 q00 = *reinterpret_cast<const f2*>(vol + (uint64_t)(r00 + b));
                const f2 q10 = *reinterpret_cast<const f2*>(vol + (uint64_t)(r10 + b));
                const f2 q01 = *reinterpret_cast<const f2*>(vol + (uint64_t)(r01 + b));
                const f2 q11 = *reinterpret_cast<const f2*>(vol + (uint64_t)(r11 + b));
                t000 = lo_is_x ? q00.x : q00.y; t100 = hi_is_y ? q00.y : q00.x;
                t010 = lo_is_x ? q10.x : q10.y; t110 = hi_is_y ? q10.y : q10.x;
                t001 = lo_is_x ? q01.x : q01.y; t101 = hi_is_y ? q01.y : q01.x;
                t011 = lo_is_x ? q11.x : q11.y; t111 = hi_is_y ? q11.y : q11.x;
            } else {
                t000 = vol[(uint64_t)(r00 + i0c) * STRIDE]; t100 = vol[(uint64_t)(r00 + i1c) * STRIDE];
                t010 = vol[(uint64_t)(r10 + i0c) * STRIDE]; t110 = vol[(uint64_t)(r10 + i1c) * STRIDE];
                t001 = vol[(uint64_t)(r01 + i0c) * STRIDE]; t101 = vol[(uint64_t)(r01 + i1c) * STRIDE];
                t011 = vol[(uint64_t)(r11 + i0c) * STRIDE]; t111 = vol[(uint64_t)(r11 + i1c) * STRIDE];
            }
        }
        const float sample_dist = trilerp(t000, t100, t010, t110, t001, t101, t011, t111, ax, ay, az) - 1e-1f;
        // Stop condition: actually hit the surface (material.frag:117-121)
        marching = marching && !(sample_dist < 1e-5f);
        // Move the ray forward by the minimum distance to the surface (material.frag:124-125)
        const V3 np = madd(ray_pos, ray_dir, sample_dist);
        ray_pos.x = marching ? np.x : ray_pos.x;
        ray_pos.y = marching ? np.y : ray_pos.y;
        ray_pos.z = marching ? np.z : ray_pos.z;
        if (T) {
            const float nt = dist_from_origin + sample_dist;
            dist_from_origin = marching ? nt : dist_from_origin;
        }
    }
    steps = last_i + 1;
    if (covered) status = marching ? -1 : (oob_dist<SYMM>(a, ray_pos) > 1e-4f ? -2 : 1);
}

// ---- the march loop in gfx950 assembly -------------------------------------------------------------------------------
// A frame is as long as its longest wave: ONE grazing ray doing up to 255 dependent iterations alone on its SIMD, where
// a lone wave issues one instruction every ~6-11 cycles whatever the instruction is.  So the loop is priced per
// INSTRUCTION, and hipcc's lowering of either C++ form (the predicated march_fast above: 78 per iteration; the same loop
// written as a plain divergent loop with breaks: ~83, the structurizer's mask bookkeeping -- tried and removed) leaves
// half of the cost on the table.  This is the same loop written by hand (profiles/r02/raymarch_loop_isa.md):
//   * the set of marching lanes IS the EXEC mask: v_cmpx removes the lanes that stop (out-of-bounds test, hit test), so
//     nothing is predicated and no mask register is maintained; s_cbranch_execz is the wave-level early exit;
//   * the one-cell cache is tested on the interpolation weights themselves: a = u - floor_cached(u) is the weight if the
//     cell is unchanged, and it is in [0, 1) -- as an unsigned integer: below 0x3f800000 -- exactly when it is unchanged;
//     three subtractions the filter needs anyway + v_max3_u32 + one compare replace 3 floors + 3 compares + 2 s_or;
//   * (y, z) components ride in packed-f32 instructions (v_pk_add/mul_f32), the corner values are kept as z-pairs so
//     that the x and y levels of the trilinear filter are packed as well;
//   * no per-iteration status or step counters (derived after the loop as in march_fast; the aux variant counts).
// 40 instructions per iteration on the common path (42 for a non-cubic box).  Every arithmetic instruction is the one the C++ form performs, in
// the same order on the same operands (no fma, no reassociation): results are bit-identical, which the parity tests
// check against the oracle for every pixel.  Covers: LINEAR filter, power-of-two extents and texture sizes (XF == 2),
// symmetric box, clamp-for-mirror (fast_index); STRIDE 1 = compact distance volume (one 8-byte load per corner row),
// STRIDE 4 = tex0.r in place.  Byte offsets are 32-bit: the launcher checks the volume's size.
// Registers: v24-v71 (less v41, v45, v49, v69) and s64-s87 are this block's (declared as clobbers; low enough that the kernel stays at 96
// VGPRs = 5 waves per SIMD); operands stay where hipcc put them.
#ifdef SDFV_TUNING
#define SDFV_MARCH_ASM_TUNE(x) x
#else
#define SDFV_MARCH_ASM_TUNE(x) ""
#endif
#define SDFV_MARCH_ASM_HEAD(ROW_BYTES)                                                                                         \
    "s_mov_b64 s[76:77], exec\n"                                                                                    \
    "s_and_b64 exec, exec, %[cov]\n"                                                                                \
    "s_cbranch_execz .Ldone_%=\n"                                                                                   \
    /* position v40, v[42:43] and direction v44, v[46:47] ARE the operands px..dz (explicit register variables) */      \
    "s_mov_b32 s66, %[miny]\n s_mov_b32 s67, %[minz]\n"      /* (miny, minz) */                                      \
    "s_mov_b32 s70, %[ky]\n s_mov_b32 s71, %[kz]\n"          /* (ky, kz) */                                          \
    "s_mov_b32 s72, 0xbf000000\n s_mov_b32 s73, 0xbf000000\n" /* (-0.5, -0.5) */                                     \
    "s_add_i32 s75, %[wm1], -1\n"                            /* W - 2 */                                             \
    "s_lshl_b32 s82, %[w], 2\n s_lshl_b32 s83, %[w], 4\n"    /* bytes per row: W * 4 (distance volume), W * 16 (tex0) */     \
    "s_lshl_b32 s85, %[w], 3\n"                              /* ... W * 8: the y-pair / y-interleaved volumes */        \
    "s_movk_i32 s74, 254\n"                                  /* 255 iterations */                                    \
    /* interior cells (all eight corners inside the volume, no clamp): the four corner rows are one offset against four  \
     * bases -- b00 = base, b10 = base + a row, b01 = base + a slice, b11 = both (row bytes: s82 for the distance volume, \
     * s83 for tex0, s85 for the pair volume).  Sizes need not be powers of two: rows and slices go by multiplication */ \
    "s_mov_b32 s64, " ROW_BYTES "\n"                                                                                 \
    "s_mul_i32 s86, %[h], " ROW_BYTES "\n"                                                                           \
    "s_add_u32 s68, %[base_lo], s86\n s_addc_u32 s69, %[base_hi], 0\n"   /* b01 */                                   \
    "s_add_u32 s64, %[base_lo], s64\n s_addc_u32 s65, %[base_hi], 0\n"   /* b10 */                                   \
    "s_add_u32 s86, s64, s86\n s_addc_u32 s87, s65, 0\n"                 /* b11 */                                   \
    SDFV_MARCH_ASM_TUNE("s_mov_b32 s84, 0\n")                /* tuning build: iterations that ran the fetch block */  \
    "v_mov_b32 v68, 0x7f800000\n v_mov_b32 v70, 0x7f800000\n v_mov_b32 v71, 0x7f800000\n" /* no cell cached */    \
    ".Lloop_%=:\n"
// One iteration, top half.  Two independent chains are interleaved so that a lone wave rarely issues an instruction
// that waits for the one before it: (A) out of bounds? max(|p| - max) > 1e-4 (material.frag:106-109) -> v_cmpx removes
// the lanes that left the box; (B) u = (p - min) * (N / size) - 0.5, x alone and (y, z) packed.  Then the weights
// relative to the cached cell: all three in [0, 1) <=> the cell is unchanged.
// OOB4 / OOB2: the out-of-bounds distance max(|p| - max) in four instructions, or in two when the box is a cube
// (max_x == max_y == max_z = m): max3(|p|) - m -- identical bits, rounding is monotonic so max and "- m" commute.
#define SDFV_MARCH_ASM_OOB4_A "v_sub_f32_e64 v32, |v40|, %[mx]\n"
#define SDFV_MARCH_ASM_OOB4_B "v_sub_f32_e64 v33, |v42|, %[my]\n"
#define SDFV_MARCH_ASM_OOB4_C "v_sub_f32_e64 v34, |v43|, %[mz]\n"
#define SDFV_MARCH_ASM_OOB4_D "v_max3_f32 v32, v32, v33, v34\n"
#define SDFV_MARCH_ASM_OOB2_A "v_max3_f32 v32, |v40|, |v42|, |v43|\n"
#define SDFV_MARCH_ASM_OOB2_D "v_subrev_f32_e32 v32, %[mx], v32\n"
#define SDFV_MARCH_ASM_TOP(T_STEP, OOB_A, OOB_B, OOB_C, OOB_D)                                                      \
    OOB_A                                                                                                           \
    "v_subrev_f32_e32 v48, %[minx], v40\n"                                                                          \
    OOB_B                                                                                                           \
    "v_pk_add_f32 v[50:51], v[42:43], s[66:67] neg_lo:[0,1] neg_hi:[0,1]\n"                                         \
    OOB_C                                                                                                           \
    "v_mul_f32_e32 v48, %[kx], v48\n"                                                                               \
    "v_pk_mul_f32 v[50:51], v[50:51], s[70:71]\n"                                                                   \
    OOB_D                                                                                                           \
    "v_add_f32_e32 v48, -0.5, v48\n"                                                                                \
    "v_pk_add_f32 v[50:51], v[50:51], s[72:73]\n"                                                                   \
    "v_cmpx_nlt_f32_e32 vcc, 0x38d1b717, v32\n"              /* exec &= !(1e-4 < oob) */                             \
    "s_cbranch_execz .Ldone_%=\n"                                                                                   \
    "v_sub_f32_e32 v52, v48, v68\n"                                                                              \
    "v_sub_f32_e32 v54, v50, v70\n"                                                                              \
    "v_sub_f32_e32 v56, v51, v71\n" T_STEP                                                                       \
    "v_max3_u32 v32, v52, v54, v56\n"                                                                            \
    "v_cmp_gt_u32_e32 vcc, 0x3f800000, v32\n"                                                                       \
    "s_andn1_saveexec_b64 s[78:79], vcc\n"                   /* exec = lanes whose cell changed */                   \
    "s_cbranch_execnz .Lfetch_%=\n"                          /* out of line: the cached case falls through */        \
    ".Lcached_%=:\n"
// The cell fetch, out of line behind the loop (one taken branch less in every iteration that stays in its cells).
#define SDFV_MARCH_ASM_FETCH_PREP(INTERIOR)                                                                         \
    ".Lfetch_%=:\n"                                                                                                 \
    SDFV_MARCH_ASM_TUNE("s_add_u32 s84, s84, 1\n")                                                                  \
    /* new cell: floor, weights, clamped corner indices (MirroredRepeat == clamp here), row numbers by shifts */    \
    "v_floor_f32_e32 v68, v48\n v_floor_f32_e32 v70, v50\n v_floor_f32_e32 v71, v51\n"                         \
    "v_cvt_i32_f32_e32 v32, v68\n v_cvt_i32_f32_e32 v33, v70\n v_cvt_i32_f32_e32 v34, v71\n"                      \
    "v_sub_f32_e32 v52, v48, v68\n v_sub_f32_e32 v54, v50, v70\n v_sub_f32_e32 v56, v51, v71\n"             \
    /* every fetching lane's cell inside the volume (cubic volumes: 0 <= i0, j0, k0 <= N - 2, one unsigned compare)? */ \
    "v_max3_u32 v35, v32, v33, v34\n"                                                                               \
    "v_cmp_gt_u32_e32 vcc, %[thresh], v35\n"                                                                        \
    "s_xor_b64 vcc, vcc, exec\n"                                                                                    \
    "s_cbranch_scc1 .Lborder_%=\n"                                                                                  \
    "v_mad_u32_u24 v39, v34, %[h], v33\n"                    /* (k0 * H + j0) * W + i0: 24-bit factors (the launcher checks) */ \
    "v_mad_u32_u24 v39, v39, %[w], v32\n" INTERIOR                                                               \
    "s_branch .Lcached_%=\n"                                                                                        \
    ".Lborder_%=:\n"                                                                                                \
    "v_max_i32_e32 v35, 0, v32\n v_max_i32_e32 v37, 0, v33\n v_max_i32_e32 v38, 0, v34\n"  /* i0c, j0c, k0c */       \
    "v_add_u32_e32 v32, 1, v32\n v_add_u32_e32 v33, 1, v33\n v_add_u32_e32 v34, 1, v34\n"                            \
    "v_min_i32_e32 v32, %[wm1], v32\n v_min_i32_e32 v33, %[hm1], v33\n v_min_i32_e32 v34, %[dm1], v34\n" /* i1c.. */  \
    "v_mad_u32_u24 v28, v38, %[h], v37\n"                    /* rows: (k0c, j0c) */                                  \
    "v_mad_u32_u24 v29, v38, %[h], v33\n"                    /*       (k0c, j1c) */                                  \
    "v_mad_u32_u24 v30, v34, %[h], v37\n"                    /*       (k1c, j0c) */                                  \
    "v_mad_u32_u24 v31, v34, %[h], v33\n"                    /*       (k1c, j1c) */
// Interior fetch (v39 = texel index of corner (i0, j0, k0)): no clamps, no selects.
#define SDFV_MARCH_ASM_INTERIOR_DIST                                                                                \
    "v_lshlrev_b32_e32 v39, 2, v39\n"                                                                               \
    "global_load_dwordx2 v[24:25], v39, %[base]\n"           /* (z0, y0): t000, t100 */                             \
    "global_load_dwordx2 v[26:27], v39, s[64:65]\n"          /* (z0, y1): t010, t110 */                             \
    "global_load_dwordx2 v[28:29], v39, s[68:69]\n"          /* (z1, y0): t001, t101 */                             \
    "global_load_dwordx2 v[30:31], v39, s[86:87]\n"          /* (z1, y1): t011, t111 */                             \
    "s_waitcnt vmcnt(1)\n"                                                                                          \
    "v_pk_mov_b32 v[60:61], v[24:25], v[28:29] op_sel:[0,0]\n" /* (t000, t001) */                                   \
    "v_pk_mov_b32 v[62:63], v[24:25], v[28:29] op_sel:[1,1]\n" /* (t100, t101) */                                   \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "v_pk_mov_b32 v[64:65], v[26:27], v[30:31] op_sel:[0,0]\n" /* (t010, t011) */                                   \
    "v_pk_mov_b32 v[66:67], v[26:27], v[30:31] op_sel:[1,1]\n" /* (t110, t111) */
#define SDFV_MARCH_ASM_INTERIOR_TEX0                                                                                \
    "v_lshlrev_b32_e32 v39, 4, v39\n"                                                                               \
    "global_load_dword v60, v39, %[base]\n global_load_dword v62, v39, %[base] offset:16\n"                       \
    "global_load_dword v64, v39, s[64:65]\n global_load_dword v66, v39, s[64:65] offset:16\n"                     \
    "global_load_dword v61, v39, s[68:69]\n global_load_dword v63, v39, s[68:69] offset:16\n"                     \
    "global_load_dword v65, v39, s[86:87]\n global_load_dword v67, v39, s[86:87] offset:16\n"                     \
    "s_waitcnt vmcnt(0)\n"
// The y-pair volume (sdfv_commit_pairs): texel (x, y, z) holds (d[y], d[min(y + 1, H - 1)]), 8 bytes, so the four corners of a
// cell's z-level -- (x0, y0), (x0, y1), (x1, y0), (x1, y1) -- are 16 CONTIGUOUS bytes: ONE dwordx4 per z-level, two gathers and
// two cache lines per cell instead of four and four.  The gather count is what a fetch costs (tools: profiles/EXPERIMENTS.md).
#define SDFV_MARCH_ASM_INTERIOR_PAIRS                                                                               \
    "v_lshlrev_b32_e32 v39, 3, v39\n"                                                                               \
    "global_load_dwordx4 v[24:27], v39, %[base]\n"           /* z0: t000, t010, t100, t110 */                       \
    "global_load_dwordx4 v[28:31], v39, s[68:69]\n"          /* z1: t001, t011, t101, t111 */                       \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "v_pk_mov_b32 v[60:61], v[24:25], v[28:29] op_sel:[0,0]\n" /* (t000, t001) */                                   \
    "v_pk_mov_b32 v[64:65], v[24:25], v[28:29] op_sel:[1,1]\n" /* (t010, t011) */                                   \
    "v_pk_mov_b32 v[62:63], v[26:27], v[30:31] op_sel:[0,0]\n" /* (t100, t101) */                                   \
    "v_pk_mov_b32 v[66:67], v[26:27], v[30:31] op_sel:[1,1]\n" /* (t110, t111) */
// Border cells of the pair volume: the first component of eight texels (clamped corners), one dword load each.
#define SDFV_MARCH_ASM_FETCH_PAIRS                                                                                  \
    "v_lshlrev_b32_e32 v35, 3, v35\n v_lshlrev_b32_e32 v32, 3, v32\n"   /* i0c, i1c as byte offsets */               \
    "v_mad_u32_u24 v24, v28, s85, v35\n v_mad_u32_u24 v25, v28, s85, v32\n"                                          \
    "global_load_dword v60, v24, %[base]\n global_load_dword v62, v25, %[base]\n"                                  \
    "v_mad_u32_u24 v26, v29, s85, v35\n v_mad_u32_u24 v27, v29, s85, v32\n"                                          \
    "global_load_dword v64, v26, %[base]\n global_load_dword v66, v27, %[base]\n"                                  \
    "v_mad_u32_u24 v24, v30, s85, v35\n v_mad_u32_u24 v25, v30, s85, v32\n"                                          \
    "global_load_dword v61, v24, %[base]\n global_load_dword v63, v25, %[base]\n"                                  \
    "v_mad_u32_u24 v26, v31, s85, v35\n v_mad_u32_u24 v27, v31, s85, v32\n"                                          \
    "global_load_dword v65, v26, %[base]\n global_load_dword v67, v27, %[base]\n"                                  \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "s_branch .Lcached_%=\n"
// The y-INTERLEAVED volume (sdfv_commit_interleaved): rows 2p and 2p + 1 of a slice share one row of (d[2p][x], d[2p+1][x])
// pairs -- 4 B/voxel like the distance volume, no duplication.  A cell whose j0 is even finds the four corners of a z-level
// in 16 contiguous bytes of pair-row j0/2; an odd j0 takes (x0, x1) of row j0 from pair-row (j0-1)/2 and of row j0+1 from
// pair-row (j0+1)/2.  Branch-free: every lane loads pair-rows j0 >> 1 and (j0 + 1) >> 1 (the same address, hence the same
// line, when j0 is even) and picks by parity -- four dwordx4 gathers over 2 (even) or 4 (odd) lines instead of four dwordx2
// over 4 lines: the line count is what a fetch costs (EXPERIMENTS R3.1, R3.10).
#define SDFV_MARCH_ASM_INTERIOR_ILV                                                                                 \
    "s_lshr_b32 s80, %[h], 1\n"                                                                                     \
    "v_and_b32_e32 v48, 1, v33\n"                            /* parity of j0 */                                      \
    "v_lshrrev_b32_e32 v35, 1, v33\n"                        /* p0 = j0 >> 1 */                                      \
    "v_add_u32_e32 v36, 1, v33\n"                                                                                   \
    "v_lshrrev_b32_e32 v36, 1, v36\n"                        /* p1 = (j0 + 1) >> 1 */                                \
    "v_mad_u32_u24 v35, v34, s80, v35\n v_mad_u32_u24 v36, v34, s80, v36\n"     /* pair rows k0 * H/2 + p */         \
    "v_mad_u32_u24 v35, v35, %[w], v32\n v_mad_u32_u24 v37, v36, %[w], v32\n"                                        \
    "v_lshlrev_b32_e32 v36, 3, v35\n v_lshlrev_b32_e32 v37, 3, v37\n"                                               \
    "v_cmp_eq_u32_e32 vcc, 1, v48\n"                                                                                \
    "global_load_dwordx4 v[24:27], v36, %[base]\n"           /* z0, pair-row p0: (y_lo, y_hi) at x0, x1 */           \
    "global_load_dwordx4 v[32:35], v36, s[68:69]\n"          /* z1, p0 */                                            \
    "s_and_saveexec_b64 s[80:81], vcc\n"                     /* odd j0 only: row j0 + 1 lives in the next pair-row */ \
    "global_load_dwordx3 v[28:30], v37, %[base]\n"           /* z0, p1: its y_lo at x0, x1 */                        \
    "global_load_dwordx3 v[36:38], v37, s[68:69]\n"          /* z1, p1 */                                            \
    "s_mov_b64 exec, s[80:81]\n"                                                                                    \
    "s_waitcnt vmcnt(2)\n"                                                                                          \
    "v_cndmask_b32_e32 v60, v24, v25, vcc\n v_cndmask_b32_e32 v62, v26, v27, vcc\n"   /* t000, t100: row j0 */       \
    "v_cndmask_b32_e32 v61, v32, v33, vcc\n v_cndmask_b32_e32 v63, v34, v35, vcc\n"   /* t001, t101 */               \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "v_cndmask_b32_e32 v64, v25, v28, vcc\n v_cndmask_b32_e32 v66, v27, v30, vcc\n"   /* t010, t110: row j0 + 1 */   \
    "v_cndmask_b32_e32 v65, v33, v36, vcc\n v_cndmask_b32_e32 v67, v35, v38, vcc\n"   /* t011, t111 */
// Border cells of the interleaved volume: eight dword loads at (pair-row * W + x) * 8 + (y & 1) * 4.  The clamped row
// numbers of FETCH_PREP are k * H + j with H even: the pair-row is row >> 1, the half row & 1.
#define SDFV_MARCH_ASM_ILV_ROW(R)                            /* v R: row number k * H + j  ->  byte offset of (x = 0) */ \
    "v_and_b32_e32 v36, 1, v" #R "\n"                        /* y & 1 */                                             \
    "v_lshrrev_b32_e32 v" #R ", 1, v" #R "\n"                /* k * H/2 + (j >> 1): H is even */                     \
    "v_mul_u32_u24_e32 v" #R ", s85, v" #R "\n"              /* * W * 8 */                                           \
    "v_lshl_add_u32 v" #R ", v36, 2, v" #R "\n"              /* + (y & 1) * 4 */
#define SDFV_MARCH_ASM_FETCH_ILV                                                                                    \
    "v_lshlrev_b32_e32 v35, 3, v35\n v_lshlrev_b32_e32 v32, 3, v32\n"   /* i0c, i1c as byte offsets (8 per x) */      \
    SDFV_MARCH_ASM_ILV_ROW(28) SDFV_MARCH_ASM_ILV_ROW(29) SDFV_MARCH_ASM_ILV_ROW(30) SDFV_MARCH_ASM_ILV_ROW(31)       \
    "v_add_u32_e32 v24, v28, v35\n v_add_u32_e32 v25, v28, v32\n"                                                    \
    "global_load_dword v60, v24, %[base]\n global_load_dword v62, v25, %[base]\n"                                  \
    "v_add_u32_e32 v26, v29, v35\n v_add_u32_e32 v27, v29, v32\n"                                                    \
    "global_load_dword v64, v26, %[base]\n global_load_dword v66, v27, %[base]\n"                                  \
    "v_add_u32_e32 v24, v30, v35\n v_add_u32_e32 v25, v30, v32\n"                                                    \
    "global_load_dword v61, v24, %[base]\n global_load_dword v63, v25, %[base]\n"                                  \
    "v_add_u32_e32 v26, v31, v35\n v_add_u32_e32 v27, v31, v32\n"                                                    \
    "global_load_dword v65, v26, %[base]\n global_load_dword v67, v27, %[base]\n"                                  \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "s_branch .Lcached_%=\n"
// STRIDE 1: x-neighbours are adjacent floats: one 8-byte load per (y, z) row at b = clamp(i0, 0, W - 2); where the clamp
// folds the two x-corners together both come from the same half (lo_is_x / hi_is_y).
#define SDFV_MARCH_ASM_FETCH_DIST                                                                                   \
    "v_min_i32_e32 v36, s75, v35\n"                          /* b = min(i0c, W - 2) */                               \
    "v_lshlrev_b32_e32 v39, 2, v36\n"                                                                               \
    "v_mad_u32_u24 v28, v28, s82, v39\n v_mad_u32_u24 v29, v29, s82, v39\n"                                          \
    "v_mad_u32_u24 v30, v30, s82, v39\n v_mad_u32_u24 v31, v31, s82, v39\n"                                          \
    "global_load_dwordx2 v[24:25], v28, %[base]\n"           /* (z0, y0) */                                          \
    "global_load_dwordx2 v[26:27], v29, %[base]\n"           /* (z0, y1) */                                          \
    "global_load_dwordx2 v[28:29], v30, %[base]\n"           /* (z1, y0) */                                          \
    "global_load_dwordx2 v[30:31], v31, %[base]\n"           /* (z1, y1) */                                          \
    "v_cmp_eq_u32_e64 s[80:81], v35, v36\n"                  /* lo_is_x */                                           \
    "v_add_u32_e32 v36, 1, v36\n"                                                                                   \
    "v_cmp_eq_u32_e32 vcc, v32, v36\n"                       /* hi_is_y */                                           \
    "s_waitcnt vmcnt(3)\n"                                                                                          \
    "v_cndmask_b32_e64 v60, v25, v24, s[80:81]\n v_cndmask_b32_e32 v62, v24, v25, vcc\n" /* t000, t100 */          \
    "s_waitcnt vmcnt(2)\n"                                                                                          \
    "v_cndmask_b32_e64 v64, v27, v26, s[80:81]\n v_cndmask_b32_e32 v66, v26, v27, vcc\n" /* t010, t110 */          \
    "s_waitcnt vmcnt(1)\n"                                                                                          \
    "v_cndmask_b32_e64 v61, v29, v28, s[80:81]\n v_cndmask_b32_e32 v63, v28, v29, vcc\n" /* t001, t101 */          \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "v_cndmask_b32_e64 v65, v31, v30, s[80:81]\n v_cndmask_b32_e32 v67, v30, v31, vcc\n" /* t011, t111 */          \
    "s_branch .Lcached_%=\n"
// STRIDE 4: tex0.r out of 16-byte texels, one dword load per corner.
#define SDFV_MARCH_ASM_FETCH_TEX0                                                                                   \
    "v_lshlrev_b32_e32 v35, 4, v35\n v_lshlrev_b32_e32 v32, 4, v32\n"   /* i0c, i1c as byte offsets */               \
    "v_mad_u32_u24 v24, v28, s83, v35\n v_mad_u32_u24 v25, v28, s83, v32\n"                                          \
    "global_load_dword v60, v24, %[base]\n global_load_dword v62, v25, %[base]\n"                                  \
    "v_mad_u32_u24 v26, v29, s83, v35\n v_mad_u32_u24 v27, v29, s83, v32\n"                                          \
    "global_load_dword v64, v26, %[base]\n global_load_dword v66, v27, %[base]\n"                                  \
    "v_mad_u32_u24 v24, v30, s83, v35\n v_mad_u32_u24 v25, v30, s83, v32\n"                                          \
    "global_load_dword v61, v24, %[base]\n global_load_dword v63, v25, %[base]\n"                                  \
    "v_mad_u32_u24 v26, v31, s83, v35\n v_mad_u32_u24 v27, v31, s83, v32\n"                                          \
    "global_load_dword v65, v26, %[base]\n global_load_dword v67, v27, %[base]\n"                                  \
    "s_waitcnt vmcnt(0)\n"                                                                                          \
    "s_branch .Lcached_%=\n"
// Corner registers as z-pairs: A0 = v[60:61] = (t000, t001), A1 = v[62:63] = (t100, t101), B0 = v[64:65] =
// (t010, t011), B1 = v[66:67] = (t110, t111); weights (a, 1 - a) as pairs v[52:53], v[54:55], v[56:57].
#define SDFV_MARCH_ASM_FILTER                                                                                       \
    "s_mov_b64 exec, s[78:79]\n"                                                                                    \
    "v_sub_f32_e32 v53, 1.0, v52\n v_sub_f32_e32 v55, 1.0, v54\n v_sub_f32_e32 v57, 1.0, v56\n"                \
    /* mix along x: c = t(x0) * (1 - ax) + t(x1) * ax */                                                            \
    "v_pk_mul_f32 v[34:35], v[62:63], v[52:53] op_sel_hi:[1,0]\n"                                                \
    "v_pk_mul_f32 v[38:39], v[66:67], v[52:53] op_sel_hi:[1,0]\n"                                                \
    "v_pk_mul_f32 v[32:33], v[60:61], v[52:53] op_sel:[0,1] op_sel_hi:[1,1]\n"                                   \
    "v_pk_mul_f32 v[36:37], v[64:65], v[52:53] op_sel:[0,1] op_sel_hi:[1,1]\n"                                   \
    "v_pk_add_f32 v[32:33], v[32:33], v[34:35]\n"            /* (c00, c01) */                                        \
    "v_pk_add_f32 v[36:37], v[36:37], v[38:39]\n"            /* (c10, c11) */                                        \
    /* mix along y */                                                                                               \
    "v_pk_mul_f32 v[32:33], v[32:33], v[54:55] op_sel:[0,1] op_sel_hi:[1,1]\n"                                     \
    "v_pk_mul_f32 v[36:37], v[36:37], v[54:55] op_sel_hi:[1,0]\n"                                                  \
    "v_pk_add_f32 v[32:33], v[32:33], v[36:37]\n"            /* (c0, c1) */                                          \
    /* mix along z, then sample_dist = r - 0.1 */                                                                   \
    "v_pk_mul_f32 v[32:33], v[32:33], v[56:57] op_sel:[0,1] op_sel_hi:[1,0]\n" /* (c0 * (1 - az), c1 * az) */      \
    "v_add_f32_e32 v36, v32, v33\n"                                                                                 \
    "v_add_f32_e32 v36, 0xbdcccccd, v36\n"                                                                          \
    /* hit?  (material.frag:117-121) */                                                                             \
    "v_cmpx_ngt_f32_e32 vcc, 0x3727c5ac, v36\n"              /* exec &= !(1e-5 > sample_dist) */
#define SDFV_MARCH_ASM_ADVANCE                                                                                      \
    /* advance the rays that go on (material.frag:124-125) */                                                       \
    "v_mul_f32_e32 v32, v44, v36\n"                                                                                \
    "v_pk_mul_f32 v[34:35], v[46:47], v[36:37] op_sel_hi:[1,0]\n"                                                  \
    "v_add_f32_e32 v40, v40, v32\n"                                                                                 \
    "v_pk_add_f32 v[42:43], v[42:43], v[34:35]\n"                                                                   \
    "s_add_u32 s74, s74, -1\n"                               /* carry out <=> iterations left */                     \
    "s_cbranch_scc1 .Lloop_%=\n"                                                                                    \
    "s_branch .Ldone_%=\n"
#define SDFV_MARCH_ASM_EPILOGUE                                                                                     \
    ".Ldone_%=:\n"                                                                                                  \
    "s_mov_b64 %[ran], exec\n"                               /* lanes still marching after 255 iterations */         \
    "s_mov_b32 %[left], s74\n"                                                                                      \
    SDFV_MARCH_ASM_TUNE("s_lshl_b32 s84, s84, 16\n s_and_b32 %[left], %[left], 0xffff\n s_or_b32 %[left], %[left], s84\n") \
    "s_and_b64 exec, s[76:77], %[cov]\n"                                                                            \
    "s_nop 0\n"
#define SDFV_MARCH_ASM_END "s_mov_b64 exec, s[76:77]\n"
#define SDFV_MARCH_ASM_OPERANDS                                                                                     \
    [dx] "v"(dirx), [dy] "v"(diry), [dz] "v"(dirz), [cov] "s"(cov), [mx] "s"(a.rp.bounds_max[0]),    \
        [my] "s"(a.rp.bounds_max[1]), [mz] "s"(a.rp.bounds_max[2]), [minx] "s"(a.rp.bounds_min[0]),                 \
        [miny] "s"(a.rp.bounds_min[1]), [minz] "s"(a.rp.bounds_min[2]), [kx] "s"(kx), [ky] "s"(ky), [kz] "s"(kz),   \
        [wm1] "s"(wm1), [hm1] "s"(hm1), [dm1] "s"(dm1), [w] "s"(t.w), [h] "s"(t.h), [base] "s"(vol), [base_lo] "s"(base_lo), [base_hi] "s"(base_hi), [thresh] "s"(thresh)
#define SDFV_MARCH_ASM_CLOBBERS                                                                                     \
    "vcc", "scc", "memory", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75",    \
        "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "v24", "v25", "v26", "v27", "v28",  \
        "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v48", "v50", "v51", "v52",    \
        "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",   \
        "v70", "v71"

template <bool SYMM_UNUSED, int STRIDE, bool T>
__device__ __forceinline__ void march_asm(const RaymarchArgs& a, const float* __restrict__ vol, const Tex& t,
                                          V3 ray_dir, bool covered, V3& ray_pos, float& dist_from_origin,
                                          int& status, int& steps, int& iterations) {
    // float multiplies are VALU work on gfx950: the (uniform) products are moved to scalar registers explicitly
    auto uniform = [](float f) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(f))); };
    const float kx = uniform(a.inv_bsize[0] * (float)t.w), ky = uniform(a.inv_bsize[1] * (float)t.h),
                kz = uniform(a.inv_bsize[2] * (float)t.d);  // exact: a power of two times an integer below 2^24
    const int wm1 = t.w - 1, hm1 = t.h - 1, dm1 = t.d - 1;
    const unsigned long long cov = __ballot(covered);
    const uint32_t base_lo = (uint32_t)(uintptr_t)vol, base_hi = (uint32_t)((uintptr_t)vol >> 32);
    // the interior fetch path tests all three cell indices with ONE compare: cubic volumes only (0: never taken)
    const uint32_t thresh = (t.w == t.h && t.h == t.d && !a.no_interior_fetch) ? (uint32_t)t.w - 1u : 0u;
    unsigned long long ran_out;
    int left = 254;  // the loop's down-counter at exit (tuning build: iterations the wave ran)
    // The block's position and direction registers are the operands themselves (explicit register variables: no copies in
    // or out, and six registers fewer than operands + copies): (py, pz) and (dy, dz) must be aligned pairs for v_pk_*.
    register float px asm("v40") = ray_pos.x;
    register float py asm("v42") = ray_pos.y;
    register float pz asm("v43") = ray_pos.z;
    register float dirx asm("v44") = ray_dir.x;
    register float diry asm("v46") = ray_dir.y;
    register float dirz asm("v47") = ray_dir.z;
#define SDFV_MARCH_ASM_OOB_BOX SDFV_MARCH_ASM_OOB4_A, SDFV_MARCH_ASM_OOB4_B, SDFV_MARCH_ASM_OOB4_C, SDFV_MARCH_ASM_OOB4_D
#define SDFV_MARCH_ASM_OOB_CUBE SDFV_MARCH_ASM_OOB2_A, "", "", SDFV_MARCH_ASM_OOB2_D
#define SDFV_MARCH_ASM_TOP_(T_STEP, ...) SDFV_MARCH_ASM_TOP(T_STEP, __VA_ARGS__)
// aux variant: distanceFromOrigin (v58) and the per-ray fetch count (v59) ride along
#define SDFV_MARCH_ASM_RUN_AUX(SHIFT, INTERIOR, FETCH, OOB)                                                         \
    asm volatile("v_mov_b32 v58, %[tt]\n v_mov_b32 v59, 0\n" SDFV_MARCH_ASM_HEAD(SHIFT)                             \
                 SDFV_MARCH_ASM_TOP_("v_add_u32_e32 v59, 1, v59\n", OOB) SDFV_MARCH_ASM_FILTER                       \
                 "v_add_f32_e32 v58, v58, v36\n" SDFV_MARCH_ASM_ADVANCE SDFV_MARCH_ASM_FETCH_PREP(INTERIOR) FETCH    \
                 SDFV_MARCH_ASM_EPILOGUE "v_mov_b32 %[tt], v58\n v_mov_b32 %[n], v59\n" SDFV_MARCH_ASM_END           \
                 : [px] "+v"(px), [py] "+v"(py), [pz] "+v"(pz), [tt] "+v"(tt), [n] "+v"(n), [ran] "=&s"(ran_out),   \
                   [left] "=&s"(left)                                                                               \
                 : SDFV_MARCH_ASM_OPERANDS                                                                          \
                 : SDFV_MARCH_ASM_CLOBBERS)
#define SDFV_MARCH_ASM_RUN(SHIFT, INTERIOR, FETCH, OOB)                                                             \
    asm volatile(SDFV_MARCH_ASM_HEAD(SHIFT) SDFV_MARCH_ASM_TOP_("", OOB) SDFV_MARCH_ASM_FILTER SDFV_MARCH_ASM_ADVANCE \
                     SDFV_MARCH_ASM_FETCH_PREP(INTERIOR) FETCH SDFV_MARCH_ASM_EPILOGUE SDFV_MARCH_ASM_END            \
                 : [px] "+v"(px), [py] "+v"(py), [pz] "+v"(pz), [ran] "=&s"(ran_out), [left] "=&s"(left)            \
                 : SDFV_MARCH_ASM_OPERANDS                                                                          \
                 : SDFV_MARCH_ASM_CLOBBERS)
    const bool cube = a.cube_box != 0;  // wave-uniform: one scalar branch per kernel
    if (T) {
        float tt = dist_from_origin;
        int n = 0;
        if (STRIDE == 3) {
            if (cube) SDFV_MARCH_ASM_RUN_AUX("s82", SDFV_MARCH_ASM_INTERIOR_ILV, SDFV_MARCH_ASM_FETCH_ILV, SDFV_MARCH_ASM_OOB_CUBE);
            else SDFV_MARCH_ASM_RUN_AUX("s82", SDFV_MARCH_ASM_INTERIOR_ILV, SDFV_MARCH_ASM_FETCH_ILV, SDFV_MARCH_ASM_OOB_BOX);
        } else if (STRIDE == 2) {
            if (cube) SDFV_MARCH_ASM_RUN_AUX("s85", SDFV_MARCH_ASM_INTERIOR_PAIRS, SDFV_MARCH_ASM_FETCH_PAIRS, SDFV_MARCH_ASM_OOB_CUBE);
            else SDFV_MARCH_ASM_RUN_AUX("s85", SDFV_MARCH_ASM_INTERIOR_PAIRS, SDFV_MARCH_ASM_FETCH_PAIRS, SDFV_MARCH_ASM_OOB_BOX);
        } else if (STRIDE == 1) {
            if (cube) SDFV_MARCH_ASM_RUN_AUX("s82", SDFV_MARCH_ASM_INTERIOR_DIST, SDFV_MARCH_ASM_FETCH_DIST, SDFV_MARCH_ASM_OOB_CUBE);
            else SDFV_MARCH_ASM_RUN_AUX("s82", SDFV_MARCH_ASM_INTERIOR_DIST, SDFV_MARCH_ASM_FETCH_DIST, SDFV_MARCH_ASM_OOB_BOX);
        } else {
            if (cube) SDFV_MARCH_ASM_RUN_AUX("s83", SDFV_MARCH_ASM_INTERIOR_TEX0, SDFV_MARCH_ASM_FETCH_TEX0, SDFV_MARCH_ASM_OOB_CUBE);
            else SDFV_MARCH_ASM_RUN_AUX("s83", SDFV_MARCH_ASM_INTERIOR_TEX0, SDFV_MARCH_ASM_FETCH_TEX0, SDFV_MARCH_ASM_OOB_BOX);
        }
        dist_from_origin = tt;
        steps = n;
    } else if (STRIDE == 3) {
        if (cube) SDFV_MARCH_ASM_RUN("s82", SDFV_MARCH_ASM_INTERIOR_ILV, SDFV_MARCH_ASM_FETCH_ILV, SDFV_MARCH_ASM_OOB_CUBE);
        else SDFV_MARCH_ASM_RUN("s82", SDFV_MARCH_ASM_INTERIOR_ILV, SDFV_MARCH_ASM_FETCH_ILV, SDFV_MARCH_ASM_OOB_BOX);
    } else if (STRIDE == 2) {
        if (cube) SDFV_MARCH_ASM_RUN("s85", SDFV_MARCH_ASM_INTERIOR_PAIRS, SDFV_MARCH_ASM_FETCH_PAIRS, SDFV_MARCH_ASM_OOB_CUBE);
        else SDFV_MARCH_ASM_RUN("s85", SDFV_MARCH_ASM_INTERIOR_PAIRS, SDFV_MARCH_ASM_FETCH_PAIRS, SDFV_MARCH_ASM_OOB_BOX);
    } else if (STRIDE == 1) {
        if (cube) SDFV_MARCH_ASM_RUN("s82", SDFV_MARCH_ASM_INTERIOR_DIST, SDFV_MARCH_ASM_FETCH_DIST, SDFV_MARCH_ASM_OOB_CUBE);
        else SDFV_MARCH_ASM_RUN("s82", SDFV_MARCH_ASM_INTERIOR_DIST, SDFV_MARCH_ASM_FETCH_DIST, SDFV_MARCH_ASM_OOB_BOX);
    } else {
        if (cube) SDFV_MARCH_ASM_RUN("s83", SDFV_MARCH_ASM_INTERIOR_TEX0, SDFV_MARCH_ASM_FETCH_TEX0, SDFV_MARCH_ASM_OOB_CUBE);
        else SDFV_MARCH_ASM_RUN("s83", SDFV_MARCH_ASM_INTERIOR_TEX0, SDFV_MARCH_ASM_FETCH_TEX0, SDFV_MARCH_ASM_OOB_BOX);
    }
#ifdef SDFV_TUNING  // left = (iterations that ran the fetch block) << 16 | the down-counter at exit
    iterations = cov ? (min(255, 255 - (int)(short)(left & 0xffff)) | (left & 0xffff0000)) : 0;
#else
    (void)iterations;
    (void)left;
#endif
    if (covered) {
        ray_pos = mk(px, py, pz);
        const bool ran = ((ran_out >> (threadIdx.x & 63)) & 1ull) != 0;
        // a stopped ray's position no longer moves: "out of bounds" is re-derived from it, as in march_fast
        status = ran ? -1 : (oob_dist<true>(a, ray_pos) > 1e-4f ? -2 : 1);
    }
}

// pow and division of the shading tail the way a GLSL compiler emits them for a GPU: exp2(y * log2(x)) and a * rcp(b) on
// the hardware's transcendental unit (v_log_f32 / v_exp_f32 / v_rcp_f32, 1 ulp each).  The results feed outColor only, whose
// gate is 1e-4 against the CPU restatement (libm powf, IEEE divide): measured distance 2.4e-7 at most (1.2e-7 with the
// scene's ACES + sRGB defaults, as with ocml's powf before).  x >= 0 here;
// x == 0 gives exp2(-inf) = 0 like powf.
__device__ __forceinline__ float shader_pow(float x, float y) {
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
__device__ __forceinline__ float shader_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// three-d 0.18.2 tone_mapping / color_mapping (material.frag:167-168) [not vendored in the reference]
__device__ __forceinline__ float tone_map(uint32_t type, float c) {
    if (type == 1) c = shader_div(c, c + 1.0f);
    else if (type == 2) c = shader_div(c * (2.51f * c + 0.03f), c * (2.43f * c + 0.59f) + 0.14f);
    else if (type == 3) {
        float x = fmaxf(0.0f, c - 0.004f);
        c = shader_div(x * (6.2f * x + 0.5f), x * (6.2f * x + 1.7f) + 0.06f);
        c = shader_pow(c, 2.2f);
    }
    return fminf(fmaxf(c, 0.0f), 1.0f);
}
__device__ __forceinline__ float color_map(uint32_t type, float c) {
    if (type != 1) return c;
    float ginv = 1.0f / 2.4f;
    float select = c >= 0.0031308f ? 1.0f : 0.0f;
    float lo = c * 12.92f;
    float hi = 1.055f * shader_pow(c, ginv) - 0.055f;
    return mixf(lo, hi, select);
}

// material.frag:158-173 with the scene's single ambient light (scene/mod.rs:106-110)
__device__ __forceinline__ float4 shade(const RaymarchArgs& a, float4 raw0, float4 raw1) {
    float metallic = raw1.x, occlusion = raw1.z;
    float albedo[3] = {raw0.y * a.rp.tint[0], raw0.z * a.rp.tint[1], raw0.w * a.rp.tint[2]};
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float lit = occlusion * a.rp.ambient[c] * mixf(albedo[c], 0.0f, metallic);
        // further ambient lights of the light list: calculate_lighting sums the lights' contributions
        for (uint32_t l = 0; l < a.rp.n_lights; ++l)
            lit += occlusion * (a.rp.lights[l].intensity * a.rp.lights[l].color[c]) * mixf(albedo[c], 0.0f, metallic);
        lit = tone_map(a.rp.tone_mapping, lit);
        lit = color_map(a.rp.color_mapping, lit);
        if (a.rp.gamma > 0.0f) lit = shader_pow(lit, a.rp.gamma);
        out[c] = lit;
    }
    return make_float4(out[0], out[1], out[2], a.rp.tint[3]);
}

// outColor is written once and never re-read by this kernel: a streaming store keeps it from evicting the
// texels the march is re-reading out of L2.
__device__ __forceinline__ void store_rgba(float4* dst, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(dst));
}
// float -> 8-bit UNORM as a GL framebuffer converts it: clamp to [0, 1], scale by 255, round to nearest (even); NaN -> 0
__device__ __forceinline__ uint32_t unorm8(float c) {
    return (uint32_t)__float2uint_rn(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f);
}
// outColor into whichever output planes the caller asked for (wave-uniform pointers)
__device__ __forceinline__ void store_color(const RaymarchArgs& a, uint64_t out_index, float4 v) {
    if (a.rgba) store_rgba(a.rgba + out_index, v);
    if (a.rgba8)
        __builtin_nontemporal_store(unorm8(v.x) | unorm8(v.y) << 8 | unorm8(v.z) << 16 | unorm8(v.w) << 24, a.rgba8 + out_index);
}

__device__ __forceinline__ void aux_clear(sdfv_march_aux& aux) {
    aux.status = 0; aux.steps = 0;
    aux.hit_pos[0] = aux.hit_pos[1] = aux.hit_pos[2] = 0.0f;
    aux.t = 0.0f;
    aux.raw0[0] = aux.raw0[1] = aux.raw0[2] = aux.raw0[3] = 0.0f;
    aux.raw1[0] = aux.raw1[1] = aux.raw1[2] = aux.raw1[3] = 0.0f;
    aux.normal[0] = aux.normal[1] = aux.normal[2] = 0.0f;
    aux.depth = 1.0f;
}

#ifdef SDFV_TUNING
__device__ __forceinline__ void stamp_wave(const RaymarchArgs& a, uint32_t bx, uint32_t by, uint32_t wave,
                                           unsigned long long t_start, unsigned long long t_start_rt, int iterations,
                                           unsigned long long covered_mask) {
    // indexed by the TILE (whatever workgroup rendered it): stamps of any tile order land in the same slots
    const uint32_t tiles_x = (a.width + 15) / 16, tiles_y = (a.rows_out + 15) / 16;
    const uint64_t wave_id = ((uint64_t)(blockIdx.z * tiles_y + by) * tiles_x + bx) * 4 + wave;
    a.wave_timing[wave_id * 4 + 0] = t_start;
    a.wave_timing[wave_id * 4 + 1] = __builtin_readcyclecounter();
    // bits 32..47 / 48..63: the 100 MHz real-time counter (one clock for the whole device; the cycle counters above are
    // per XCD) at the wave's start / end, modulo 2^16 ticks
    a.wave_timing[wave_id * 4 + 2] = (unsigned long long)(uint32_t)iterations | ((t_start_rt & 0xffffull) << 32) |
                                     ((__builtin_amdgcn_s_memrealtime() & 0xffffull) << 48);
    a.wave_timing[wave_id * 4 + 3] = covered_mask;
}
#endif

// Primary ray through the centre of pixel (px, py) (image row 0 = top), not yet normalised.
__device__ __forceinline__ V3 pixel_ray_raw(const RaymarchArgs& a, const sdfv_camera& cam, uint32_t px, uint32_t py) {
    const float ndc_x = (((float)px + 0.5f) / (float)a.width) * 2.0f - 1.0f;
    const float ndc_y = 1.0f - (((float)py + 0.5f) / (float)a.height) * 2.0f;
    const float sx = ndc_x * cam.aspect * cam.tan_half_fovy;
    const float sy = ndc_y * cam.tan_half_fovy;
    return mk(cam.forward[0] + cam.right[0] * sx + cam.up[0] * sy,
              cam.forward[1] + cam.right[1] * sx + cam.up[1] * sy,
              cam.forward[2] + cam.right[2] * sx + cam.up[2] * sy);
}

// The bbox fragment of that ray (slab test standing in for the rasterised cube, scene/sdf/mod.rs:254-282) and
// main()'s ray set-up, material.frag:133-139.  Returns whether the pixel is covered by the box.
__device__ __forceinline__ bool box_fragment_ray(const RaymarchArgs& a, V3 eye, V3 d_raw, bool in_image,
                                                 V3& ray_origin, V3& ray_dir) {
    const V3 d0 = normalize(d_raw);
    const float tx1 = (a.rp.bounds_min[0] - eye.x) / d0.x, tx2 = (a.rp.bounds_max[0] - eye.x) / d0.x;
    const float ty1 = (a.rp.bounds_min[1] - eye.y) / d0.y, ty2 = (a.rp.bounds_max[1] - eye.y) / d0.y;
    const float tz1 = (a.rp.bounds_min[2] - eye.z) / d0.z, tz2 = (a.rp.bounds_max[2] - eye.z) / d0.z;
    const float tnear = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    const float tfar = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    const bool covered = in_image && (tfar >= tnear && tfar > 0.0f);
    const float tfrag = tnear > 0.0f ? tnear : tfar;
    const V3 pos = madd(eye, d0, tfrag);
    ray_origin = pos;
    ray_dir = normalize(sub(ray_origin, eye));
    if (oob_dist<false>(a, madd(ray_origin, ray_dir, 0.2f)) > 0.0f) ray_origin = madd(eye, ray_dir, 0.2f);
    return covered;
}

// MODE: 0 = general kernel (any filter, any extents: the shader's nested loop with full MirroredRepeat);
//       1 = fast march over tex0.r; 2 = fast march over the compact distance volume (both LINEAR only);
//       3 = the hand-written loop over the y-pair volume (sdfv_commit_pairs; two 16-byte gathers per cell);
//       4 = ... over the y-interleaved volume (sdfv_commit_interleaved; 4 B/voxel, 2 or 4 lines per cell).
// XF:   0 = IEEE divide, 1 = exact power-of-two reciprocal, 2 = power-of-two extents and texture sizes.
// AUX:  the per-pixel march record is stored (and distanceFromOrigin accumulated).
#ifndef SDFV_RM_MIN_WAVES
#define SDFV_RM_MIN_WAVES 1
#endif
// NORMAL: sdfNormal's four taps are compiled in (always with the aux record; without it only for
// SDFV_OPT_RAYMARCH_KEEP_NORMAL).  A compile-time switch: the taps' 32 gathers with 64-bit addresses would otherwise set the
// register count of the kernel that never runs them (79 -> 72 VGPRs = one more wave per SIMD).
template <int MODE, bool LINEAR, int XF, bool SYMM, bool AUX, bool ASM = false, bool NORMAL = AUX>
__global__ __launch_bounds__(256, SDFV_RM_MIN_WAVES) void raymarch_kernel(RaymarchArgs a) {
    constexpr bool FAST = MODE != 0;
    static_assert(NORMAL || !AUX, "the aux record carries the normal");
    // 8x8 pixel tile per wave, 2x2 waves per workgroup
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t bx = blockIdx.x, by = blockIdx.y;
#ifdef SDFV_TUNING
    if (a.tile_order) {  // 1-D launch over the tiles, in the order given
        const uint32_t tiles_x = (a.width + 15) / 16, t = a.tile_order[blockIdx.x];
        bx = t % tiles_x;
        by = t / tiles_x;
    } else
#endif
    if (a.group_shift == 0 && a.rotate_columns) {
        // Launch order (batches of cameras, tile bands).  The workgroups of a launch are dealt to the eight XCDs round robin by
        // their linear number and every XCD works through ITS share at its own pace: the launch is as long as the busiest
        // XCD's share.  With a tile row a multiple of 8 wide (1080p: 120, 4K: 240, 1440p: 160) XCD k would render tile columns
        // k, k + 8, ... of EVERY row of EVERY camera -- and an object 44 columns wide gives six columns to some XCDs and five
        // to others: 24 % more marching on the busiest XCD than on the idlest (profiles/r04_batch_timeline_*.json, EXPERIMENTS R4.12: they finish 290 us
        // apart in a 1.75 ms batch).  Rotating the columns by row and camera deals every XCD every column in turn.
        const uint32_t tiles_x = gridDim.x;
        const uint32_t r = (blockIdx.y + blockIdx.z) % tiles_x;  // scalar
        bx = blockIdx.x + r;
        if (bx >= tiles_x) bx -= tiles_x;
    } else if (a.group_shift) {
        // XCD-aware order (a.group_shift = g): the launch is 1-D over workgroups; workgroup L runs on XCD L % 8 (observed
        // placement, used for speed only).  The image is cut into groups of 2^g x 2^g tiles; group number G goes to XCD
        // G % 8, so that the tiles one XCD's L2 serves are compact patches spread evenly over the image.
        const uint32_t g = a.group_shift, per = 1u << (2 * g);
        const uint32_t L = blockIdx.x, xcd = L & 7u, k = L >> 3;
        const uint32_t G = (k >> (2 * g)) * 8u + xcd, t = k & (per - 1u);
        uint32_t gx, gy;
        if (a.first_w == 0) {
            gx = G % a.groups_x;
            gy = G / a.groups_x;
        } else {
            // box-first: the groups under the projected bounding box start first -- the frame is as long as its longest
            // wave, and those are all there; the groups that only see background fill in behind them
            auto div = [](uint32_t n, uint32_t d, uint32_t m) { return d == 1 ? n : __umulhi(n, m); };
            const uint32_t n_in = a.first_w * a.first_h, top = a.first_gy0 * a.groups_x, rest_w = a.groups_x - a.first_w;
            uint32_t j = G - n_in;
            if (G < n_in) {
                const uint32_t r = div(G, a.first_w, a.m_first_w);
                gx = a.first_gx0 + (G - r * a.first_w);
                gy = a.first_gy0 + r;
            } else if (j < top) {
                gy = div(j, a.groups_x, a.m_groups_x);
                gx = j - gy * a.groups_x;
            } else if ((j -= top) < a.first_h * rest_w) {
                const uint32_t r = div(j, rest_w, a.m_rest_w), c = j - r * rest_w;
                gy = a.first_gy0 + r;
                gx = c < a.first_gx0 ? c : c + a.first_w;
            } else {
                j -= a.first_h * rest_w;
                const uint32_t r = div(j, a.groups_x, a.m_groups_x);
                gy = a.first_gy0 + a.first_h + r;
                gx = j - r * a.groups_x;
            }
        }
        bx = (gx << g) + (t & ((1u << g) - 1u));
        by = (gy << g) + (t >> g);
        if (bx >= a.tiles_x || by >= a.tiles_y) return;  // padding of the last groups
    }
    const uint32_t px = bx * 16 + (wave & 1) * 8 + (lane & 7);
    const uint32_t row = by * 16 + (wave >> 1) * 8 + (lane >> 3);  // row of the output: within [y0, y1), or of the bands
    const uint32_t py = a.y0 + row + (row >> a.band_shift) * a.band_skip;  // band k of the set = rows [k << shift, ...) of the output
    const uint32_t cam_idx = blockIdx.z;
    const bool in_image = px < a.width && py < a.y1;
    // (two scalar-load sequences under a uniform branch; a reference chosen by `?:` would mix the address spaces)
    sdfv_camera cam;
    if (a.camera_list) cam = a.camera_list[cam_idx];
    else cam = a.cameras[cam_idx];
    const uint64_t out_index = ((uint64_t)cam_idx * a.rows_out + row) * a.width + px;
#ifdef SDFV_TUNING
    if (a.priority_map && a.priority_map[by * ((a.width + 15) / 16) + bx]) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t_start = a.wave_timing ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long t_start_rt = a.wave_timing ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif

    const V3 eye = mk(cam.eye[0], cam.eye[1], cam.eye[2]);
    const V3 d_raw = pixel_ray_raw(a, cam, px, py);

    // Conservative tile cull (the reference gets it from rasterising the box).  A ray that misses the box's
    // bounding sphere inflated by 1 % cannot be covered; if no lane of the wave can be covered, the wave
    // writes its transparent pixels and leaves before the divisions of the exact slab test.
    {
        const V3 m = sub(eye, mk(a.cull_center[0], a.cull_center[1], a.cull_center[2]));
        const float dd = d_raw.x * d_raw.x + d_raw.y * d_raw.y + d_raw.z * d_raw.z;
        const float mm = m.x * m.x + m.y * m.y + m.z * m.z;
        const float md = m.x * d_raw.x + m.y * d_raw.y + m.z * d_raw.z;
        const float r2 = a.cull_radius2;
        const bool miss = mm > r2 && (md >= 0.0f || mm * dd - md * md > r2 * dd);
        if (__ballot(in_image && !miss) == 0ull) {
            if (in_image) {
                store_color(a, out_index, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                if (AUX) {
                    sdfv_march_aux aux;
                    aux_clear(aux);
                    a.aux[out_index] = aux;
                }
            }
            if (in_image && a.depth) a.depth[out_index] = 1.0f;
#ifdef SDFV_TUNING
            if (a.wave_timing && lane == 0) stamp_wave(a, bx, by, wave, t_start, t_start_rt, 0, 0ull);
#endif
            return;
        }
    }

    const Tex tex0{a.tex0, (int)a.rp.tex_size[0], (int)a.rp.tex_size[1], (int)a.rp.tex_size[2]};
    const Tex tex1{a.tex1, tex0.w, tex0.h, tex0.d};
    V3 ray_origin, ray_dir;
    const bool covered = box_fragment_ray(a, eye, d_raw, in_image, ray_origin, ray_dir);

    // sdfRaycast(rayOrigin, rayDir, 256), material.frag:92-128
    V3 ray_pos = ray_origin;
    float dist_from_origin = 0.0f;
    int status = covered ? -1 : 0;  // -1 = out of steps unless something else ends the ray
    int steps = 0;
    int iterations = 0;
#ifdef SDFV_TUNING
    const unsigned long long t_loop0 = a.wave_timing ? __builtin_readcyclecounter() : 0ull;
#endif
#define SDFV_MARCH march_fast  // the C++ loop, where the hand-written one's specialisation does not apply
    if (ASM && MODE == 1) {
        march_asm<SYMM, 4, AUX>(a, reinterpret_cast<const float*>(a.tex0), tex0, ray_dir, covered, ray_pos, dist_from_origin,
                                status, steps, iterations);
    } else if (ASM && MODE == 2) {
        march_asm<SYMM, 1, AUX>(a, a.dist, tex0, ray_dir, covered, ray_pos, dist_from_origin, status, steps, iterations);
    } else if (ASM && MODE == 3) {
        march_asm<SYMM, 2, AUX>(a, a.pairs, tex0, ray_dir, covered, ray_pos, dist_from_origin, status, steps, iterations);
    } else if (ASM && MODE == 4) {
        march_asm<SYMM, 3, AUX>(a, a.ilv, tex0, ray_dir, covered, ray_pos, dist_from_origin, status, steps, iterations);
    } else if (MODE == 1) {
        SDFV_MARCH<XF, SYMM, 4, AUX>(a, reinterpret_cast<const float*>(a.tex0), tex0, ray_dir, covered, ray_pos,
                                     dist_from_origin, status, steps, iterations);
    } else if (MODE >= 2) {  // (MODE 3 is only ever launched with the hand-written loop)
        SDFV_MARCH<XF, SYMM, 1, AUX>(a, a.dist, tex0, ray_dir, covered, ray_pos, dist_from_origin, status, steps,
                                     iterations);
    } else {
        bool marching = covered;
        for (int i = 0; i < 255; ++i) {
            if (__ballot(marching) == 0ull) break;  // wave-level early termination
            ++iterations;
            if (marching) {
                if (oob_dist<false>(a, ray_pos) > 1e-4f) {
                    status = -2;
                    marching = false;
                } else {
                    float sample_dist = sample_r<LINEAR, XF, false>(a, tex0, ray_pos) - 1e-1f;
                    ++steps;
                    if (sample_dist < 1e-5f) {
                        status = 1;
                        marching = false;
                    } else {
                        dist_from_origin += sample_dist;
                        ray_pos = madd(ray_pos, ray_dir, sample_dist);
                    }
                }
            }
        }
    }

#ifdef SDFV_TUNING
    const unsigned long long t_loop1 = a.wave_timing ? __builtin_readcyclecounter() : 0ull;
#endif
    float4 rgba = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float frag_depth = 1.0f;  // material.frag:147 (no hit); also where no fragment exists
    sdfv_march_aux aux;
    if (AUX) {
        aux_clear(aux);
        if (covered) {
            aux.status = status;
            aux.steps = steps;
            aux.hit_pos[0] = ray_pos.x; aux.hit_pos[1] = ray_pos.y; aux.hit_pos[2] = ray_pos.z;
            aux.t = dist_from_origin;
        }
    }
    if (status == 1) {
        // The fast kernels know the hit point is within 1e-4 of the box; the normal's taps are h further out,
        // a.fast_normal says whether floor(u) still stays in [-1, N-1] for them.
        const float4 raw0 = sample_rgba<LINEAR, XF, FAST>(a, tex0, ray_pos);  // == the march's last sample
        // one texture's eight 16-byte corners at a time: both sets in flight together cost 6 more VGPRs (78: one wave per
        // SIMD less) and buy nothing even for single frames, which are not occupancy-limited -- measured 1 % SLOWER on
        // every view at 1080p / 256^3 and 4K / 512^3 (round 3 A/B, profiles/r03_hit_path_ab.json)
        V3 pos1 = ray_pos;
        asm volatile("" : "+v"(pos1.x) : "v"(raw0.x), "v"(raw0.y), "v"(raw0.z), "v"(raw0.w));  // same value, after raw0
        const float4 raw1 = sample_rgba<LINEAR, XF, FAST>(a, tex1, pos1);  // material.frag:154
        rgba = shade(a, raw0, raw1);
        if (a.depth) {  // gl_FragDepth, material.frag:180-181
            const float* m = cam.bvp;
            const float hz = m[2] * ray_pos.x + m[6] * ray_pos.y + m[10] * ray_pos.z + m[14];
            const float hw = m[3] * ray_pos.x + m[7] * ray_pos.y + m[11] * ray_pos.z + m[15];
            frag_depth = hz / hw;
        }
        if (NORMAL) {
            // sdfNormal, material.frag:73-80
            const float sxn = (float)tex0.w / a.rp.lod_dist_between_samples;
            const float syn = (float)tex0.h / a.rp.lod_dist_between_samples;
            const float szn = (float)tex0.d / a.rp.lod_dist_between_samples;
            const float h = 1.0f / sqrtf(sxn * sxn + syn * syn + szn * szn);
            const V3 p1 = mk(ray_pos.x + h, ray_pos.y - h, ray_pos.z - h);  // k.xyy
            const V3 p2 = mk(ray_pos.x - h, ray_pos.y - h, ray_pos.z + h);  // k.yyx
            const V3 p3 = mk(ray_pos.x - h, ray_pos.y + h, ray_pos.z - h);  // k.yxy
            const V3 p4 = mk(ray_pos.x + h, ray_pos.y + h, ray_pos.z + h);  // k.xxx
            float d1, d2, d3, d4;
            if (MODE >= 2 && a.dist && a.fast_normal) {  // the taps read the compact distance volume too
                d1 = sample_r_linear<XF, true, 1>(a, a.dist, tex0, p1) - 1e-1f;
                d2 = sample_r_linear<XF, true, 1>(a, a.dist, tex0, p2) - 1e-1f;
                d3 = sample_r_linear<XF, true, 1>(a, a.dist, tex0, p3) - 1e-1f;
                d4 = sample_r_linear<XF, true, 1>(a, a.dist, tex0, p4) - 1e-1f;
            } else if (FAST && a.fast_normal) {
                d1 = sample_r<LINEAR, XF, true>(a, tex0, p1) - 1e-1f;
                d2 = sample_r<LINEAR, XF, true>(a, tex0, p2) - 1e-1f;
                d3 = sample_r<LINEAR, XF, true>(a, tex0, p3) - 1e-1f;
                d4 = sample_r<LINEAR, XF, true>(a, tex0, p4) - 1e-1f;
            } else {
                d1 = sample_r<LINEAR, XF, false>(a, tex0, p1) - 1e-1f;
                d2 = sample_r<LINEAR, XF, false>(a, tex0, p2) - 1e-1f;
                d3 = sample_r<LINEAR, XF, false>(a, tex0, p3) - 1e-1f;
                d4 = sample_r<LINEAR, XF, false>(a, tex0, p4) - 1e-1f;
            }
            const V3 n = normalize(mk(d1 + -d2 + -d3 + d4, -d1 + -d2 + d3 + d4, -d1 + d2 + -d3 + d4));
            if (AUX) {
                // gl_FragDepth, material.frag:180-181
                const float* m = cam.bvp;
                const float hz = m[2] * ray_pos.x + m[6] * ray_pos.y + m[10] * ray_pos.z + m[14];
                const float hw = m[3] * ray_pos.x + m[7] * ray_pos.y + m[11] * ray_pos.z + m[15];
                aux.raw0[0] = raw0.x; aux.raw0[1] = raw0.y; aux.raw0[2] = raw0.z; aux.raw0[3] = raw0.w;
                aux.raw1[0] = raw1.x; aux.raw1[1] = raw1.y; aux.raw1[2] = raw1.z; aux.raw1[3] = raw1.w;
                aux.normal[0] = n.x; aux.normal[1] = n.y; aux.normal[2] = n.z;
                aux.depth = hz / hw;
            } else {
                // keep the normal live when nobody stores it: the shader text computes it per hit
                asm volatile("" ::"v"(n.x), "v"(n.y), "v"(n.z));
            }
        }
    }

#ifdef SDFV_TUNING
    // slot 3: cycles before the march loop (ray set-up) | cycles of the loop << 32; what is left of end - start is the hit's
    // texel gathers and shading
    if (a.wave_timing && lane == 0)
        stamp_wave(a, bx, by, wave, t_start, t_start_rt, iterations, ((t_loop0 - t_start) & 0xffffffffull) | ((t_loop1 - t_loop0) << 32));
#endif
    if (in_image) {
        store_color(a, out_index, rgba);
        if (a.depth) a.depth[out_index] = frag_depth;
        if (AUX) a.aux[out_index] = aux;
    }
}

// ---- march over a z-slab: the grid stays sharded across GPUs and rays are handed between ranks ----------------
// Same arithmetic, same order as the MODE 0 loop above (hence as the oracle); what differs is WHO executes an
// iteration: the rank whose slab holds the ray's cell.  Texel z-indices are clamped (the launcher requires the
// fast_index condition) and rebased to the first resident slice.
struct SlabTex {
    const float4* data;
    int w, h, d;    // GLOBAL texture size
    int z_lo;       // first resident slice
};

__device__ __forceinline__ Footprint footprint_slab(const SlabTex& t, V3 p01, int& k0c) {
    float u = p01.x * (float)t.w - 0.5f, v = p01.y * (float)t.h - 0.5f, w = p01.z * (float)t.d - 0.5f;
    float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    Footprint f;
    f.ax = u - fu; f.ay = v - fv; f.az = w - fw;
    const int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
    const uint32_t sy = (uint32_t)t.w, sz = (uint32_t)t.w * (uint32_t)t.h;
    const uint32_t i0m = (uint32_t)max(i0, 0), i1m = (uint32_t)min(i0 + 1, t.w - 1);
    const uint32_t j0m = (uint32_t)max(j0, 0) * sy, j1m = (uint32_t)min(j0 + 1, t.h - 1) * sy;
    k0c = min(max(k0, 0), t.d - 1);
    const uint32_t k0m = (uint32_t)(max(k0, 0) - t.z_lo) * sz, k1m = (uint32_t)(min(k0 + 1, t.d - 1) - t.z_lo) * sz;
    f.o000 = k0m + j0m + i0m; f.o100 = k0m + j0m + i1m;
    f.o010 = k0m + j1m + i0m; f.o110 = k0m + j1m + i1m;
    f.o001 = k1m + j0m + i0m; f.o101 = k1m + j0m + i1m;
    f.o011 = k1m + j1m + i0m; f.o111 = k1m + j1m + i1m;
    return f;
}

__device__ __forceinline__ float4 fetch_rgba(const float4* data, const Footprint& f) {
    const float4 t000 = data[f.o000], t100 = data[f.o100], t010 = data[f.o010], t110 = data[f.o110];
    const float4 t001 = data[f.o001], t101 = data[f.o101], t011 = data[f.o011], t111 = data[f.o111];
    float4 r;
    r.x = trilerp(t000.x, t100.x, t010.x, t110.x, t001.x, t101.x, t011.x, t111.x, f.ax, f.ay, f.az);
    r.y = trilerp(t000.y, t100.y, t010.y, t110.y, t001.y, t101.y, t011.y, t111.y, f.ax, f.ay, f.az);
    r.z = trilerp(t000.z, t100.z, t010.z, t110.z, t001.z, t101.z, t011.z, t111.z, f.ax, f.ay, f.az);
    r.w = trilerp(t000.w, t100.w, t010.w, t110.w, t001.w, t101.w, t011.w, t111.w, f.ax, f.ay, f.az);
    return r;
}

template <int XF, bool AUX>
__global__ __launch_bounds__(256) void raymarch_slab_kernel(RaymarchArgs a, SlabMarchArgs s) {
    const bool device_counts = s.in_count[0] != nullptr || s.in_count[1] != nullptr;
    const bool first_round = s.in == nullptr && !device_counts;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const sdfv_camera& cam = a.cameras[0];
    uint32_t px, py;
    bool live;            // this thread carries a ray (first round: a pixel of the image)
    int i = 0;
    V3 ray_pos = mk(0.0f, 0.0f, 0.0f);
    float dist_from_origin = 0.0f;
    if (first_round) {
        // 8x8 pixel tiles per wave, like the single-GPU kernel
        const uint32_t tiles_x = (a.width + 7) / 8, tile = gid >> 6, lane = gid & 63;
        px = (tile % tiles_x) * 8 + (lane & 7);
        py = (tile / tiles_x) * 8 + (lane >> 3);
        live = px < a.width && py < a.height;
    } else {
        sdfv_ray_state st{};
        if (device_counts) {
            // the neighbours' ray buffers as they arrived: their counts are read here, on the device
            const uint32_t n0 = s.in_count[0] ? min(*s.in_count[0], s.capacity) : 0u;
            const uint32_t n1 = s.in_count[1] ? min(*s.in_count[1], s.capacity) : 0u;
            live = gid < n0 + n1;
            if (live) st = gid < n0 ? s.in_rays[0][gid] : s.in_rays[1][gid - n0];
        } else {
            live = gid < s.n_in;
            st = s.in[live ? gid : 0];
        }
        // a continuation round is launched for the most rays that COULD arrive (their number is only known on the device):
        // waves beyond the lists leave before the ray set-up
        if (__ballot(live) == 0ull) return;
        px = st.pixel % a.width;
        py = st.pixel / a.width;
        i = (int)st.iteration;
        ray_pos = mk(st.pos[0], st.pos[1], st.pos[2]);
        dist_from_origin = st.t;
    }
    const uint32_t pixel = py * a.width + px;
    const V3 eye = mk(cam.eye[0], cam.eye[1], cam.eye[2]);
    V3 ray_origin, ray_dir;
    const bool covered = box_fragment_ray(a, eye, pixel_ray_raw(a, cam, px, py), live, ray_origin, ray_dir);
    const SlabTex tex{a.tex0, (int)a.rp.tex_size[0], (int)a.rp.tex_size[1], (int)a.rp.tex_size[2], (int)s.z_lo};
    const float* dist_r = reinterpret_cast<const float*>(a.tex0);

    bool marching = live && covered;
    if (first_round) {
        ray_pos = ray_origin;
        // every pixel of this rank's image starts transparent; the rank a ray ends on overwrites it
        if (live) {
            a.rgba[pixel] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (AUX) {
                sdfv_march_aux z;
                aux_clear(z);
                z.depth = 0.0f;  // summed over ranks; the caller restores 1.0 where no rank reports a status
                a.aux[pixel] = z;
            }
        }
        // a ray that starts in another rank's slab is that rank's to begin
        int k0c;
        (void)footprint_slab(tex, to_p01<XF>(a, ray_pos), k0c);
        marching = marching && k0c >= (int)s.own_begin && k0c < (int)s.own_end;
    }
    const bool mine = marching;  // this rank reports the ray's end unless it hands the ray over
    int status = 0, steps = 0;
    bool exported = false;
    int export_dir = 0;
    Footprint hit_fp{};
    // one-cell cache (round 4), as in the single-GPU kernels: the eight corner distances are re-fetched only when the ray
    // enters another cell -- (o000, o111) identify the cell with its clamps -- and the longest rays are the ones that crawl
    uint32_t cell0 = 0xffffffffu, cell7 = 0xffffffffu;
    float c000 = 0.0f, c100 = 0.0f, c010 = 0.0f, c110 = 0.0f, c001 = 0.0f, c101 = 0.0f, c011 = 0.0f, c111 = 0.0f;
    for (;;) {
        marching = marching && i < 255;  // sdfRaycast's `for (i < 255)`; a ray that runs out keeps status 0 for now
        if (__ballot(marching) == 0ull) break;
        if (!marching) continue;
        if (oob_dist<false>(a, ray_pos) > 1e-4f) {  // material.frag:106-109
            status = -2;
            steps = i;
            marching = false;
            continue;
        }
        int k0c;
        const Footprint f = footprint_slab(tex, to_p01<XF>(a, ray_pos), k0c);
        if (k0c < (int)s.own_begin || k0c >= (int)s.own_end) {
            // the cell belongs to a z-neighbour: hand the ray over exactly as it is -- after the loop, a wave at a time
            export_dir = k0c < (int)s.own_begin ? 0 : 1;
            exported = true;
            marching = false;
            continue;
        }
        if (f.o000 != cell0 || f.o111 != cell7) {
            cell0 = f.o000;
            cell7 = f.o111;
            c000 = dist_r[(uint64_t)f.o000 * 4]; c100 = dist_r[(uint64_t)f.o100 * 4];
            c010 = dist_r[(uint64_t)f.o010 * 4]; c110 = dist_r[(uint64_t)f.o110 * 4];
            c001 = dist_r[(uint64_t)f.o001 * 4]; c101 = dist_r[(uint64_t)f.o101 * 4];
            c011 = dist_r[(uint64_t)f.o011 * 4]; c111 = dist_r[(uint64_t)f.o111 * 4];
        }
        const float sample_dist = trilerp(c000, c100, c010, c110, c001, c101, c011, c111, f.ax, f.ay, f.az) - 1e-1f;
        ++i;  // one more tex0 fetch done
        if (sample_dist < 1e-5f) {  // material.frag:117-121
            status = 1;
            steps = i;
            hit_fp = f;
            marching = false;
        } else {  // material.frag:124-125
            dist_from_origin += sample_dist;
            ray_pos = madd(ray_pos, ray_dir, sample_dist);
        }
    }
    // Rays for the neighbours: one counter update per wave and direction (round 4; it was one same-address atomic per ray,
    // two with `leftover`, which serialise in one L2 channel), the states stored side by side.  All lanes of the wave are here:
    // the loop ends on a ballot and every return above is wave-uniform.
    if (__ballot(exported) != 0ull) {
        const uint32_t lane = __lane_id();
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const uint64_t m = __ballot(exported && export_dir == dir);
            if (m == 0ull) continue;
            const uint32_t n = (uint32_t)__popcll(m);
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) {
                base = atomicAdd(dir == 0 ? s.count_down : s.count_up, n);
                if (base + n > s.capacity && s.overflow) *s.overflow = 1u;  // the caller's capacity was too small: the image is incomplete
                if (s.leftover) atomicAdd(s.leftover, n);
            }
            base = __shfl(base, leader);
            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (exported && export_dir == dir && at < s.capacity) {
                sdfv_ray_state st;
                st.pixel = pixel;
                st.iteration = (uint32_t)i;
                st.pos[0] = ray_pos.x; st.pos[1] = ray_pos.y; st.pos[2] = ray_pos.z;
                st.t = dist_from_origin;
                (dir == 0 ? s.out_down : s.out_up)[at] = st;
            }
        }
    }
    if (!mine || exported) return;
    if (status == 0) {  // out of steps, material.frag:99-101
        status = -1;
        steps = i;
    }

    float4 rgba = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 raw0 = rgba, raw1 = rgba;
    if (status == 1) {
        raw0 = fetch_rgba(a.tex0, hit_fp);  // == the march's last sample
        raw1 = fetch_rgba(a.tex1, hit_fp);  // material.frag:154
        rgba = shade(a, raw0, raw1);
    }
    a.rgba[pixel] = rgba;
    if (AUX) {
        sdfv_march_aux aux;
        aux_clear(aux);
        aux.status = status;
        aux.steps = steps;
        aux.hit_pos[0] = ray_pos.x; aux.hit_pos[1] = ray_pos.y; aux.hit_pos[2] = ray_pos.z;
        aux.t = dist_from_origin;
        if (status == 1) {
            const float* m = cam.bvp;  // gl_FragDepth, material.frag:180-181
            const float hz = m[2] * ray_pos.x + m[6] * ray_pos.y + m[10] * ray_pos.z + m[14];
            const float hw = m[3] * ray_pos.x + m[7] * ray_pos.y + m[11] * ray_pos.z + m[15];
            aux.raw0[0] = raw0.x; aux.raw0[1] = raw0.y; aux.raw0[2] = raw0.z; aux.raw0[3] = raw0.w;
            aux.raw1[0] = raw1.x; aux.raw1[1] = raw1.y; aux.raw1[2] = raw1.z; aux.raw1[3] = raw1.w;
            aux.depth = hz / hw;
            // sdfNormal (material.frag:73-80): four taps h away from the hit; each needs the two slices around its own
            // floor(w), which can be one slice further than the march's fetch -- resident only with a second upper ghost
            // slice (or at the ends of the grid).  Without it the normal stays (0, 0, 0).
            if (a.fast_normal) {
                const float sxn = (float)tex.w / a.rp.lod_dist_between_samples;
                const float syn = (float)tex.h / a.rp.lod_dist_between_samples;
                const float szn = (float)tex.d / a.rp.lod_dist_between_samples;
                const float h = 1.0f / sqrtf(sxn * sxn + syn * syn + szn * szn);
                const V3 taps[4] = {mk(ray_pos.x + h, ray_pos.y - h, ray_pos.z - h), mk(ray_pos.x - h, ray_pos.y - h, ray_pos.z + h),
                                    mk(ray_pos.x - h, ray_pos.y + h, ray_pos.z - h), mk(ray_pos.x + h, ray_pos.y + h, ray_pos.z + h)};
                float d[4];
                bool resident = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int kc;
                    const V3 q = to_p01<XF>(a, taps[k]);
                    const float w = q.z * (float)tex.d - 0.5f;
                    const int k0 = (int)floorf(w);
                    const int lo = max(k0, 0), hi = min(k0 + 1, tex.d - 1);
                    resident = resident && lo >= (int)s.z_lo && hi < (int)(s.z_lo + s.z_count);
                    d[k] = 0.0f;
                    if (resident) {  // only then are the eight texels inside this rank's allocation
                        const Footprint f = footprint_slab(tex, q, kc);
                        d[k] = trilerp(dist_r[(uint64_t)f.o000 * 4], dist_r[(uint64_t)f.o100 * 4], dist_r[(uint64_t)f.o010 * 4],
                                       dist_r[(uint64_t)f.o110 * 4], dist_r[(uint64_t)f.o001 * 4], dist_r[(uint64_t)f.o101 * 4],
                                       dist_r[(uint64_t)f.o011 * 4], dist_r[(uint64_t)f.o111 * 4], f.ax, f.ay, f.az) - 1e-1f;
                    }
                }
                if (resident) {
                    const V3 n = normalize(mk(d[0] + -d[1] + -d[2] + d[3], -d[0] + -d[1] + d[2] + d[3], -d[0] + d[1] + -d[2] + d[3]));
                    aux.normal[0] = n.x; aux.normal[1] = n.y; aux.normal[2] = n.z;
                }
            }
        }
        a.aux[pixel] = aux;
    }
}

// unused dynamic LDS per workgroup (SDFV_OPT_RAYMARCH_WAVES_PER_SIMD): 160 KB per CU, one wave of a workgroup per SIMD
#define SDFV_RM_LDS(a) ((a).lds_cap_bytes)
template <int MODE, bool LINEAR, int XF, bool SYMM>
void launch_aux(const RaymarchArgs& a, dim3 grid, hipStream_t stream) {
    if (a.aux)
        hipLaunchKernelGGL((raymarch_kernel<MODE, LINEAR, XF, SYMM, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
    else if (a.compute_normal)
        hipLaunchKernelGGL((raymarch_kernel<MODE, LINEAR, XF, SYMM, false, false, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
    else
        hipLaunchKernelGGL((raymarch_kernel<MODE, LINEAR, XF, SYMM, false>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
}

// The hand-written loop addresses rows and slices by 24-bit multiplies (v_mad_u32_u24: full rate, any size -- not only powers
// of two): row numbers k * H + j and the bytes of a row (16 * W for tex0) must stay below 2^24.
static bool asm_addressing_ok(const RaymarchArgs& a) {
    return a.rp.tex_size[0] < (1u << 20) && (uint64_t)a.rp.tex_size[1] * a.rp.tex_size[2] < (1ull << 24);
}

template <int MODE>
void launch_fast(const RaymarchArgs& a, dim3 grid, hipStream_t stream) {
    const int xf = a.pow2_extent ? (a.pow2_size ? 2 : 1) : 0;
    const bool symm = a.symmetric_box != 0;
    // the hand-written loop: its specialisation (power-of-two extents and sizes, symmetric box) and 32-bit byte offsets
    const uint64_t texels = (uint64_t)a.rp.tex_size[0] * a.rp.tex_size[1] * a.rp.tex_size[2];
    if (xf == 2 && symm && a.asm_loop && a.rp.tex_size[0] >= 2 && asm_addressing_ok(a) && texels <= (MODE == 2 ? (1ull << 30) : (1ull << 28))) {
        if (a.aux) hipLaunchKernelGGL((raymarch_kernel<MODE, true, 2, true, true, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
        else if (a.compute_normal)
            hipLaunchKernelGGL((raymarch_kernel<MODE, true, 2, true, false, true, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
        else hipLaunchKernelGGL((raymarch_kernel<MODE, true, 2, true, false, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
        return;
    }
    if (xf == 2 && symm) launch_aux<MODE, true, 2, true>(a, grid, stream);
    else if (xf == 2) launch_aux<MODE, true, 2, false>(a, grid, stream);
    else if (xf == 1 && symm) launch_aux<MODE, true, 1, true>(a, grid, stream);
    else if (xf == 1) launch_aux<MODE, true, 1, false>(a, grid, stream);
    else launch_aux<MODE, true, 0, false>(a, grid, stream);
}

}  // namespace

namespace {
constexpr uint32_t kStoreChunk = 32;  // 32 x 120 B + the two scalars: 3 856 B of kernel arguments
struct CameraChunk {
    sdfv_camera c[kStoreChunk];
};
static_assert(sizeof(sdfv_camera) % 4 == 0, "cameras are copied word by word");
__global__ __launch_bounds__(256) void store_cameras_kernel(CameraChunk chunk, uint32_t words, uint32_t* __restrict__ out) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&chunk);
    for (uint32_t w = threadIdx.x; w < words; w += 256) out[w] = src[w];
}
}  // namespace

hipError_t launch_store_cameras(const sdfv_camera* host, uint32_t n, sdfv_camera* device, hipStream_t stream) {
    for (uint32_t c0 = 0; c0 < n; c0 += kStoreChunk) {
        const uint32_t nc = n - c0 < kStoreChunk ? n - c0 : kStoreChunk;
        CameraChunk chunk;
        memcpy(chunk.c, host + c0, nc * sizeof(sdfv_camera));
        hipLaunchKernelGGL(store_cameras_kernel, dim3(1), dim3(256), 0, stream, chunk, (uint32_t)(nc * sizeof(sdfv_camera) / 4),
                           reinterpret_cast<uint32_t*>(device + c0));
        if (hipError_t e = hipGetLastError()) return e;
    }
    return hipSuccess;
}

hipError_t launch_raymarch_slab(const RaymarchArgs& a, const SlabMarchArgs& s, hipStream_t stream) {
    const bool device_counts = s.in_count[0] || s.in_count[1];
    const uint64_t threads = device_counts ? (uint64_t)s.max_in
                             : s.in ? (uint64_t)s.n_in
                                    : (uint64_t)((a.width + 7) / 8) * ((a.height + 7) / 8) * 64;  // whole 8x8 tiles
    if (threads == 0) return hipSuccess;
    const dim3 grid((uint32_t)((threads + 255) / 256));
    if (a.pow2_extent) {
        if (a.aux) hipLaunchKernelGGL((raymarch_slab_kernel<1, true>), grid, dim3(256), 0, stream, a, s);
        else hipLaunchKernelGGL((raymarch_slab_kernel<1, false>), grid, dim3(256), 0, stream, a, s);
    } else {
        if (a.aux) hipLaunchKernelGGL((raymarch_slab_kernel<0, true>), grid, dim3(256), 0, stream, a, s);
        else hipLaunchKernelGGL((raymarch_slab_kernel<0, false>), grid, dim3(256), 0, stream, a, s);
    }
    return hipGetLastError();
}

static hipError_t launch_raymarch_grid(const RaymarchArgs& a, dim3 grid, hipStream_t stream);

// Screen rectangle (in groups of tiles) of the bounding box seen from camera 0, for the box-first launch order: the eight
// corners through the same pinhole model as pixel_ray_raw, one pixel of margin.  Order only: any rectangle renders the same
// image, so a corner behind the camera simply switches the ordering off.
static void box_first_rectangle(RaymarchArgs& ag, uint32_t groups_y) {
    const sdfv_camera& cam = ag.cameras[0];
    float x_lo = 3.0e38f, x_hi = -3.0e38f, y_lo = 3.0e38f, y_hi = -3.0e38f;
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {(c & 1 ? ag.rp.bounds_max : ag.rp.bounds_min)[0], (c & 2 ? ag.rp.bounds_max : ag.rp.bounds_min)[1],
                            (c & 4 ? ag.rp.bounds_max : ag.rp.bounds_min)[2]};
        const float v[3] = {p[0] - cam.eye[0], p[1] - cam.eye[1], p[2] - cam.eye[2]};
        const float zc = v[0] * cam.forward[0] + v[1] * cam.forward[1] + v[2] * cam.forward[2];
        if (!(zc > 1e-6f)) return;
        const float xc = v[0] * cam.right[0] + v[1] * cam.right[1] + v[2] * cam.right[2];
        const float yc = v[0] * cam.up[0] + v[1] * cam.up[1] + v[2] * cam.up[2];
        const float px = (xc / (zc * cam.aspect * cam.tan_half_fovy) + 1.0f) * 0.5f * (float)ag.width;
        const float py = (1.0f - yc / (zc * cam.tan_half_fovy)) * 0.5f * (float)ag.height - (float)ag.y0;
        if (!(px == px) || !(py == py)) return;
        x_lo = fminf(x_lo, px), x_hi = fmaxf(x_hi, px), y_lo = fminf(y_lo, py), y_hi = fmaxf(y_hi, py);
    }
    const float edge = (float)(16u << ag.group_shift), gxs = (float)ag.groups_x, gys = (float)groups_y;
    const float fx0 = fminf(fmaxf(floorf((x_lo - 1.0f) / edge), 0.0f), gxs), fx1 = fminf(fmaxf(floorf((x_hi + 1.0f) / edge) + 1.0f, 0.0f), gxs);
    const float fy0 = fminf(fmaxf(floorf((y_lo - 1.0f) / edge), 0.0f), gys), fy1 = fminf(fmaxf(floorf((y_hi + 1.0f) / edge) + 1.0f, 0.0f), gys);
    if (!(fx1 > fx0) || !(fy1 > fy0)) return;  // the box is off screen
    ag.first_gx0 = (uint32_t)fx0, ag.first_gy0 = (uint32_t)fy0;
    ag.first_w = (uint32_t)(fx1 - fx0), ag.first_h = (uint32_t)(fy1 - fy0);
    if (ag.first_w == ag.groups_x && ag.first_h == groups_y) {
        ag.first_w = 0;  // covers the image: plain order
        return;
    }
    auto magic = [](uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); };  // d == 1: the kernel does not divide
    ag.m_groups_x = magic(ag.groups_x), ag.m_first_w = magic(ag.first_w), ag.m_rest_w = magic(ag.groups_x - ag.first_w);
}

// Resident waves per SIMD for this launch (7 = what the register file allows = no cap).
//
// A frame is as long as its longest wave.  While every wave that can be active is resident at once, a long wave shares its
// CU's gather path with up to 27 others and advances at a fraction of its lone speed; with fewer resident waves the ones that
// started first (box-first order: those under the box) finish sooner and the rest fill in behind them.  That only pays when
// (i) the frame is bound by its long waves, not by throughput -- a box covering a small multiple of the machine, not the
// whole image; (ii) a gather that misses L2 is expensive -- a marched volume larger than the Infinity Cache; smaller ones
// (256^3: 64 MB) move by +-3 % either way (profiles/r03_occupancy_rule.json: 13 views x
// 1080p/256^3, 1440p/512^3, 4K/512^3 x every cap).  What the launcher knows: the screen rectangle of the projected bounding
// box (box-first order) in waves, the machine's wave slots, the volume's bytes.  Rule: cap at 4 when the rectangle holds
// between 1x and 3.5x the slots at 7 per SIMD and the volume exceeds the last-level cache; no cap otherwise (camera
// inside, box filling the image, batches of cameras, small volumes, a box of a few tiles).  Speed only.
// Which volume the hand-written loop marches over, of those the caller handed in: 4 the y-interleaved volume, 3 the pair
// volume, 0 neither (the loop's specialisation does not apply, or neither was given).  With both: the pair volume while its
// 8 B/voxel fit the last-level cache (fewest gathers), the interleaved one beyond (half the footprint) --
// profiles/r03_pairs_bench.json.
// Does the hand-written loop's specialisation apply to these render parameters, and may it address a volume of `kind` (3 the
// pair volume, 4 the y-interleaved one)?  Pointers apart, everything march_volume_mode decides on -- sdfv_march_volume_advice
// asks the same question BEFORE a volume exists (the advice once checked less than the launcher, so a host could
// be advised a layout no march would ever read and then gather 16-byte tex0 texels instead of its 4-byte distance volume).
bool march_volume_applicable(const RaymarchArgs& a, int kind) {
    const uint64_t texels = (uint64_t)a.rp.tex_size[0] * a.rp.tex_size[1] * a.rp.tex_size[2];
    const bool loop_ok = a.rp.lod_dist_between_samples == 1.0f && a.fast_index && a.pow2_extent && a.pow2_size && a.symmetric_box &&
                         a.asm_loop && a.rp.tex_size[0] >= 2 && asm_addressing_ok(a);
    // 32-bit byte offsets: 8 B/texel of pairs reach 2^28 texels with the loop's shifts, 4 B/voxel of ilv 2^30 (like dist)
    if (kind == 3) return loop_ok && texels <= (1ull << 28);
    if (kind == 4) return loop_ok && a.rp.tex_size[1] >= 2 && (a.rp.tex_size[1] & 1u) == 0 && texels <= (1ull << 30);
    return false;
}

static int march_volume_mode(const RaymarchArgs& a) {
    const uint64_t texels = (uint64_t)a.rp.tex_size[0] * a.rp.tex_size[1] * a.rp.tex_size[2];
    // a non-cubic grid never takes the loop's interior fetch (one compare for all three cell indices), and the border fetch of
    // these two volumes is eight single loads against the distance volume's four 8-byte ones: with `dist` at hand, use it
    const bool cubic = a.rp.tex_size[0] == a.rp.tex_size[1] && a.rp.tex_size[1] == a.rp.tex_size[2];
    if (!cubic && a.dist) return 0;
    const bool pairs_ok = a.pairs && march_volume_applicable(a, 3);
    const bool ilv_ok = a.ilv && march_volume_applicable(a, 4);
    if (pairs_ok && ilv_ok) return (a.last_level_cache_bytes && texels * 8u > a.last_level_cache_bytes) ? 4 : 3;
    return ilv_ok ? 4 : (pairs_ok ? 3 : 0);
}

static uint32_t occupancy_rule(const RaymarchArgs& ag) {
    if (ag.waves_per_simd >= 2 && ag.waves_per_simd <= 6) return ag.waves_per_simd;  // the caller's cap
    if (ag.waves_per_simd == 7) return 7;
    if (ag.n_cameras != 1 || ag.first_w == 0 || ag.group_shift == 0 || ag.wave_slots_per_simd_unit == 0) return 7;
    const uint64_t texels = (uint64_t)ag.rp.tex_size[0] * ag.rp.tex_size[1] * ag.rp.tex_size[2];
    const int mode = march_volume_mode(ag);
    const uint64_t volume_bytes = texels * (mode == 3 ? 8u : ((mode == 4 || ag.dist) ? 4u : 16u));
    if (ag.last_level_cache_bytes == 0 || volume_bytes <= ag.last_level_cache_bytes) return 7;
    const uint64_t rect_waves = (uint64_t)ag.first_w * ag.first_h * (4ull << (2 * ag.group_shift));  // 4 waves per 16 x 16 tile
    const uint64_t slots7 = 7ull * ag.wave_slots_per_simd_unit;
    if (rect_waves < slots7 || 2 * rect_waves > 7 * slots7) return 7;
    return 4;
}

// w waves per SIMD = w workgroups per CU: each asks for a w-th of the CU's 160 KB of LDS (less a little for rounding)
static uint32_t lds_cap_for(uint32_t w) {
    uint32_t bytes = (w >= 2 && w <= 6) ? (160u * 1024u) / w - 1024u : 0u;
    if (bytes > 65536u) bytes = 65536u;  // the launch limit without an opt-in: 2 waves per SIMD
    return bytes;
}

hipError_t launch_raymarch(const RaymarchArgs& a, hipStream_t stream) {
    const uint32_t rows = a.rows_out;
    if (a.width == 0 || rows == 0 || a.n_cameras == 0) return hipSuccess;
    dim3 grid((a.width + 15) / 16, (rows + 15) / 16, a.n_cameras);
    RaymarchArgs ag = a;
    const dim3 tiles = grid;
#ifdef SDFV_TUNING
    if (a.tile_order) {
        ag.group_shift = 0;
        ag.lds_cap_bytes = lds_cap_for(occupancy_rule(ag));  // no rectangle here: the caller's cap or none
        return launch_raymarch_grid(ag, dim3(tiles.x * tiles.y, 1, 1), stream);
    }
#endif
    // the 1-D launch over groups of 2^shift x 2^shift tiles (padded to whole groups, a multiple of 8 of them), with the
    // box-first rectangle where one exists
    auto grouped = [&](uint32_t shift) {
        ag.group_shift = shift;
        ag.tiles_x = tiles.x;
        ag.tiles_y = tiles.y;
        ag.groups_x = (tiles.x + (1u << shift) - 1) >> shift;
        const uint32_t groups_y = (tiles.y + (1u << shift) - 1) >> shift;
        const uint32_t groups = ((ag.groups_x * groups_y + 7u) / 8u) * 8u;
        ag.first_w = 0;
        if (a.box_first && a.n_cameras == 1 && groups < 65536u) box_first_rectangle(ag, groups_y);  // 16-bit quotients
        grid = dim3(groups << (2 * shift), 1, a.n_cameras);
    };
    if (a.band_skip) {
        ag.group_shift = 0;  // interleaved bands: launch order (the tile orders below assume a contiguous range of rows)
    } else if (a.group_shift == kGroupAuto) {
        // one camera: groups of 2 x 2 tiles with the box-first order where the bounding box projects to a proper part of the
        // image, otherwise (camera inside the box, box filling the image, order switched off) groups of 4 x 4
        grouped(1);
        if (ag.first_w == 0) grouped(2);
    } else if (a.group_shift) {
        grouped(a.group_shift);
    }
    ag.lds_cap_bytes = lds_cap_for(occupancy_rule(ag));
    return launch_raymarch_grid(ag, grid, stream);
}

static hipError_t launch_raymarch_grid(const RaymarchArgs& a, dim3 grid, hipStream_t stream) {
    const bool linear = a.rp.lod_dist_between_samples == 1.0f;
    if (linear && a.fast_index) {
        // the pair / interleaved volumes are acceleration structures of the hand-written loop only: wherever that loop's
        // specialisation does not apply the march reads the distance volume / tex0.r as before -- the same bits either way
        const int mode = march_volume_mode(a);
        const bool ilv_ok = mode == 4, pairs_ok = mode == 3;
        if (ilv_ok) {
            if (a.aux) hipLaunchKernelGGL((raymarch_kernel<4, true, 2, true, true, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
            else if (a.compute_normal)
                hipLaunchKernelGGL((raymarch_kernel<4, true, 2, true, false, true, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
            else hipLaunchKernelGGL((raymarch_kernel<4, true, 2, true, false, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
        } else if (pairs_ok) {
            if (a.aux) hipLaunchKernelGGL((raymarch_kernel<3, true, 2, true, true, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
            else if (a.compute_normal)
                hipLaunchKernelGGL((raymarch_kernel<3, true, 2, true, false, true, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
            else hipLaunchKernelGGL((raymarch_kernel<3, true, 2, true, false, true>), grid, dim3(256), SDFV_RM_LDS(a), stream, a);
        } else if (a.dist) launch_fast<2>(a, grid, stream);
        else launch_fast<1>(a, grid, stream);
    } else if (linear) {
        if (a.pow2_extent) launch_aux<0, true, 1, false>(a, grid, stream);
        else launch_aux<0, true, 0, false>(a, grid, stream);
    } else {
        if (a.pow2_extent) launch_aux<0, false, 1, false>(a, grid, stream);
        else launch_aux<0, false, 0, false>(a, grid, stream);
    }
    return hipGetLastError();
}

}  // namespace sdfv
