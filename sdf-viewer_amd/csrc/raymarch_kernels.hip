// raymarch_kernels.hip -- per-pixel sphere tracing of the voxel grid on gfx950:
// the fragment shader of the reference (src/app/scene/sdf/material.frag, whole file) as a HIP kernel.
//
// One thread per pixel.  A 64-lane wave covers an 8x8 pixel tile (neighbouring rays walk neighbouring
// texels, so the 8 dword gathers of a trilinear fetch land in few cache lines); a 256-thread workgroup
// covers 16x16.  The march loop is wave-synchronous: each iteration ballots the lanes still
// marching and the wave leaves the loop as soon as the ballot is empty (early ray termination),
// instead of running the shader's fixed 255-iteration bound.
//
// texture(sampler3D) is restated in full fp32 (GL texel-centre convention, MirroredRepeat, mix() along
// x, then y, then z): hardware texture filtering uses low-precision fixed-point weights and would not
// hold the 1e-4 RGBA tolerance, so no hipTextureObject is used.  The march reads only tex0.r (4 of
// every 16 bytes); the full tex0/tex1 texels are fetched once, at the hit.
//
// Ray set-up: the reference rasterises the bbox cube and the fragment's `pos` is a point on its
// surface (scene/sdf/mod.rs:254-282, material.rs:75-81 Cull::None).  Here `pos` comes from a ray/AABB
// slab test through the pixel centre: the entry point when the camera is outside the box, the exit
// point when it is inside; pixels whose ray misses the box are transparent.
#include "raymarch_kernels.h"

#include <hip/hip_runtime.h>

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

// GL MIRRORED_REPEAT on a texel index.  The march keeps p within 1e-4 of the box, so i is in
// [-1, n] in practice; the general case is kept for arbitrary callers.
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
    uint64_t o000, o100, o010, o110, o001, o101, o011, o111;
    float ax, ay, az;
};

__device__ __forceinline__ Footprint footprint(const Tex& t, float p01x, float p01y, float p01z) {
    float u = p01x * (float)t.w - 0.5f, v = p01y * (float)t.h - 0.5f, w = p01z * (float)t.d - 0.5f;
    float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    Footprint f;
    f.ax = u - fu; f.ay = v - fv; f.az = w - fw;
    int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
    uint64_t i0m = mirror_index(i0, t.w), i1m = mirror_index(i0 + 1, t.w);
    uint64_t j0m = (uint64_t)mirror_index(j0, t.h) * t.w, j1m = (uint64_t)mirror_index(j0 + 1, t.h) * t.w;
    uint64_t slice = (uint64_t)t.w * t.h;
    uint64_t k0m = mirror_index(k0, t.d) * slice, k1m = mirror_index(k0 + 1, t.d) * slice;
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

// p01 = (p - sdfBoundsMin) / (sdfBoundsMax - sdfBoundsMin), material.frag:44.  When every extent is an exact
// power of two (the demo's [-1,1]^3: 2.0) dividing equals multiplying by the exact reciprocal, bit for
// bit, and the three IEEE divides (~12 instructions each) leave the per-step dependency chain.
template <bool POW2>
__device__ __forceinline__ V3 to_p01(const RaymarchArgs& a, V3 p) {
    if (POW2)
        return mk((p.x - a.rp.bounds_min[0]) * a.inv_bsize[0], (p.y - a.rp.bounds_min[1]) * a.inv_bsize[1],
                  (p.z - a.rp.bounds_min[2]) * a.inv_bsize[2]);
    return mk((p.x - a.rp.bounds_min[0]) / a.bsize[0], (p.y - a.rp.bounds_min[1]) / a.bsize[1],
              (p.z - a.rp.bounds_min[2]) / a.bsize[2]);
}

// sdfSampleRawNearest's snapped coordinate + NEAREST fetch, material.frag:27-36
__device__ __forceinline__ uint64_t nearest_offset(const RaymarchArgs& a, const Tex& t, V3 p01) {
    float rx = (float)t.w / a.rp.lod_dist_between_samples;
    float ry = (float)t.h / a.rp.lod_dist_between_samples;
    float rz = (float)t.d / a.rp.lod_dist_between_samples;
    float qx = roundf(p01.x * rx) / rx, qy = roundf(p01.y * ry) / ry, qz = roundf(p01.z * rz) / rz;
    int i = (int)floorf(qx * (float)t.w), j = (int)floorf(qy * (float)t.h), k = (int)floorf(qz * (float)t.d);
    return ((uint64_t)mirror_index(k, t.d) * t.h + mirror_index(j, t.h)) * t.w + mirror_index(i, t.w);
}

// sdfSampleRawInterp(.., p).r only -- what the march and the normal need.  material.frag:42-53
template <bool LINEAR, bool POW2>
__device__ __forceinline__ float sample_r(const RaymarchArgs& a, const Tex& t, V3 p) {
    V3 q = to_p01<POW2>(a, p);
    const float* base = reinterpret_cast<const float*>(t.data);
    if (LINEAR) {
        Footprint f = footprint(t, q.x, q.y, q.z);
        float t000 = base[f.o000 * 4], t100 = base[f.o100 * 4], t010 = base[f.o010 * 4], t110 = base[f.o110 * 4];
        float t001 = base[f.o001 * 4], t101 = base[f.o101 * 4], t011 = base[f.o011 * 4], t111 = base[f.o111 * 4];
        return trilerp(t000, t100, t010, t110, t001, t101, t011, t111, f.ax, f.ay, f.az);
    }
    return base[nearest_offset(a, t, q) * 4];
}

template <bool LINEAR, bool POW2>
__device__ __forceinline__ float4 sample_rgba(const RaymarchArgs& a, const Tex& t, V3 p) {
    V3 q = to_p01<POW2>(a, p);
    if (LINEAR) {
        Footprint f = footprint(t, q.x, q.y, q.z);
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

// The march's sampler: tex0.r with a one-cell register cache.  Sphere tracing takes its smallest steps
// exactly on the longest rays (grazing a surface), so consecutive samples usually fall in the same
// texel cell: the 8 corner values are kept in registers and re-fetched only when floor(u,v,w) changes.
// Values and operation order are those of sample_r(); only redundant loads are skipped.
struct CellCache {
    float fu, fv, fw;  // floor of the unnormalised texel coordinates of the cached cell
    float t000, t100, t010, t110, t001, t101, t011, t111;
};

template <bool POW2>
__device__ __forceinline__ float march_sample(const RaymarchArgs& a, const Tex& t, V3 p, CellCache& c) {
    V3 q = to_p01<POW2>(a, p);
    float u = q.x * (float)t.w - 0.5f, v = q.y * (float)t.h - 0.5f, w = q.z * (float)t.d - 0.5f;
    float fu = floorf(u), fv = floorf(v), fw = floorf(w);
    float ax = u - fu, ay = v - fv, az = w - fw;
    if (fu != c.fu || fv != c.fv || fw != c.fw) {
        c.fu = fu; c.fv = fv; c.fw = fw;
        int i0 = (int)fu, j0 = (int)fv, k0 = (int)fw;
        uint32_t i0m = mirror_index(i0, t.w), i1m = mirror_index(i0 + 1, t.w);
        uint32_t j0m = mirror_index(j0, t.h) * (uint32_t)t.w, j1m = mirror_index(j0 + 1, t.h) * (uint32_t)t.w;
        uint64_t slice = (uint64_t)t.w * t.h;
        const float* b0 = reinterpret_cast<const float*>(t.data + mirror_index(k0, t.d) * slice);
        const float* b1 = reinterpret_cast<const float*>(t.data + mirror_index(k0 + 1, t.d) * slice);
        c.t000 = b0[(j0m + i0m) * 4u]; c.t100 = b0[(j0m + i1m) * 4u];
        c.t010 = b0[(j1m + i0m) * 4u]; c.t110 = b0[(j1m + i1m) * 4u];
        c.t001 = b1[(j0m + i0m) * 4u]; c.t101 = b1[(j0m + i1m) * 4u];
        c.t011 = b1[(j1m + i0m) * 4u]; c.t111 = b1[(j1m + i1m) * 4u];
    }
    return trilerp(c.t000, c.t100, c.t010, c.t110, c.t001, c.t101, c.t011, c.t111, ax, ay, az);
}

// sdfOutOfBoundsDist, material.frag:83-88
__device__ __forceinline__ float oob_dist(const RaymarchArgs& a, V3 p) {
    float ox = fmaxf(a.rp.bounds_min[0] - p.x, p.x - a.rp.bounds_max[0]);
    float oy = fmaxf(a.rp.bounds_min[1] - p.y, p.y - a.rp.bounds_max[1]);
    float oz = fmaxf(a.rp.bounds_min[2] - p.z, p.z - a.rp.bounds_max[2]);
    return fmaxf(ox, fmaxf(oy, oz));
}

// three-d 0.18.2 tone_mapping / color_mapping (material.frag:167-168) [not vendored in the reference]
__device__ __forceinline__ float tone_map(uint32_t type, float c) {
    if (type == 1) c = c / (c + 1.0f);
    else if (type == 2) c = (c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f);
    else if (type == 3) {
        float x = fmaxf(0.0f, c - 0.004f);
        c = (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
        c = powf(c, 2.2f);
    }
    return fminf(fmaxf(c, 0.0f), 1.0f);
}
__device__ __forceinline__ float color_map(uint32_t type, float c) {
    if (type != 1) return c;
    float ginv = 1.0f / 2.4f;
    float select = c >= 0.0031308f ? 1.0f : 0.0f;
    float lo = c * 12.92f;
    float hi = 1.055f * powf(c, ginv) - 0.055f;
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
        lit = tone_map(a.rp.tone_mapping, lit);
        lit = color_map(a.rp.color_mapping, lit);
        if (a.rp.gamma > 0.0f) lit = powf(lit, a.rp.gamma);
        out[c] = lit;
    }
    return make_float4(out[0], out[1], out[2], a.rp.tint[3]);
}

template <bool LINEAR, bool POW2>
__global__ __launch_bounds__(256) void raymarch_kernel(RaymarchArgs a) {
    // 8x8 pixel tile per wave, 2x2 waves per workgroup
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const uint32_t row = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);  // row within [y0, y1)
    const uint32_t py = a.y0 + row;
    const uint32_t cam_idx = blockIdx.z;
    const bool in_image = px < a.width && py < a.y1;
    const sdfv_camera& cam = a.cameras[cam_idx];
    const Tex tex0{a.tex0, (int)a.rp.tex_size[0], (int)a.rp.tex_size[1], (int)a.rp.tex_size[2]};
    const Tex tex1{a.tex1, tex0.w, tex0.h, tex0.d};

    const unsigned long long t_start = a.wave_timing ? __builtin_readcyclecounter() : 0ull;
    sdfv_march_aux aux;
    aux.status = 0; aux.steps = 0;
    aux.hit_pos[0] = aux.hit_pos[1] = aux.hit_pos[2] = 0.0f;
    aux.t = 0.0f;
    aux.raw0[0] = aux.raw0[1] = aux.raw0[2] = aux.raw0[3] = 0.0f;
    aux.raw1[0] = aux.raw1[1] = aux.raw1[2] = aux.raw1[3] = 0.0f;
    aux.normal[0] = aux.normal[1] = aux.normal[2] = 0.0f;
    aux.depth = 1.0f;
    float4 rgba = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    // primary ray through the pixel centre (image row 0 = top)
    float ndc_x = (((float)px + 0.5f) / (float)a.width) * 2.0f - 1.0f;
    float ndc_y = 1.0f - (((float)py + 0.5f) / (float)a.height) * 2.0f;
    float sx = ndc_x * cam.aspect * cam.tan_half_fovy;
    float sy = ndc_y * cam.tan_half_fovy;
    V3 eye = mk(cam.eye[0], cam.eye[1], cam.eye[2]);
    V3 d0 = normalize(mk(cam.forward[0] + cam.right[0] * sx + cam.up[0] * sy,
                         cam.forward[1] + cam.right[1] * sx + cam.up[1] * sy,
                         cam.forward[2] + cam.right[2] * sx + cam.up[2] * sy));

    // bbox fragment via slab test
    float tx1 = (a.rp.bounds_min[0] - eye.x) / d0.x, tx2 = (a.rp.bounds_max[0] - eye.x) / d0.x;
    float ty1 = (a.rp.bounds_min[1] - eye.y) / d0.y, ty2 = (a.rp.bounds_max[1] - eye.y) / d0.y;
    float tz1 = (a.rp.bounds_min[2] - eye.z) / d0.z, tz2 = (a.rp.bounds_max[2] - eye.z) / d0.z;
    float tnear = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    float tfar = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    const bool covered = in_image && (tfar >= tnear && tfar > 0.0f);
    float tfrag = tnear > 0.0f ? tnear : tfar;
    V3 pos = madd(eye, d0, tfrag);

    // main(), material.frag:133-139
    V3 ray_origin = pos;
    V3 ray_dir = normalize(sub(ray_origin, eye));
    if (oob_dist(a, madd(ray_origin, ray_dir, 0.2f)) > 0.0f) ray_origin = madd(eye, ray_dir, 0.2f);

    // sdfRaycast(rayOrigin, rayDir, 256), material.frag:92-128
    V3 ray_pos = ray_origin;
    float dist_from_origin = 0.0f;
    int status = covered ? -1 : 0;  // -1 = out of steps unless something else ends the ray
    int steps = 0;
    bool marching = covered;
    int iterations = 0;
    CellCache cell;
    cell.fu = cell.fv = cell.fw = -4.0f;  // never a valid floor(u): u >= -0.5 - 1e-4 * N while marching
    cell.t000 = cell.t100 = cell.t010 = cell.t110 = cell.t001 = cell.t101 = cell.t011 = cell.t111 = 0.0f;
    for (int i = 0; i < 255; ++i) {
        if (__ballot(marching) == 0ull) break;  // wave-level early termination
        ++iterations;
        if (marching) {
            if (oob_dist(a, ray_pos) > 1e-4f) {
                status = -2;
                marching = false;
            } else {
                float sample_dist =
                    (LINEAR ? march_sample<POW2>(a, tex0, ray_pos, cell) : sample_r<false, POW2>(a, tex0, ray_pos)) - 1e-1f;
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

    if (covered) {
        aux.status = status;
        aux.steps = steps;
        aux.hit_pos[0] = ray_pos.x; aux.hit_pos[1] = ray_pos.y; aux.hit_pos[2] = ray_pos.z;
        aux.t = dist_from_origin;
    }
    if (status == 1) {
        float4 raw0 = sample_rgba<LINEAR, POW2>(a, tex0, ray_pos);  // == the march's last sample
        float4 raw1 = sample_rgba<LINEAR, POW2>(a, tex1, ray_pos);  // material.frag:154
        rgba = shade(a, raw0, raw1);
        if (a.aux || a.compute_normal) {
            // sdfNormal, material.frag:73-80
            float sxn = (float)tex0.w / a.rp.lod_dist_between_samples;
            float syn = (float)tex0.h / a.rp.lod_dist_between_samples;
            float szn = (float)tex0.d / a.rp.lod_dist_between_samples;
            float h = 1.0f / sqrtf(sxn * sxn + syn * syn + szn * szn);
            float d1 = sample_r<LINEAR, POW2>(a, tex0, mk(ray_pos.x + h, ray_pos.y - h, ray_pos.z - h)) - 1e-1f;  // k.xyy
            float d2 = sample_r<LINEAR, POW2>(a, tex0, mk(ray_pos.x - h, ray_pos.y - h, ray_pos.z + h)) - 1e-1f;  // k.yyx
            float d3 = sample_r<LINEAR, POW2>(a, tex0, mk(ray_pos.x - h, ray_pos.y + h, ray_pos.z - h)) - 1e-1f;  // k.yxy
            float d4 = sample_r<LINEAR, POW2>(a, tex0, mk(ray_pos.x + h, ray_pos.y + h, ray_pos.z + h)) - 1e-1f;  // k.xxx
            V3 n = normalize(mk(d1 + -d2 + -d3 + d4, -d1 + -d2 + d3 + d4, -d1 + d2 + -d3 + d4));
            // gl_FragDepth, material.frag:180-181
            const float* m = cam.bvp;
            float hz = m[2] * ray_pos.x + m[6] * ray_pos.y + m[10] * ray_pos.z + m[14];
            float hw = m[3] * ray_pos.x + m[7] * ray_pos.y + m[11] * ray_pos.z + m[15];
            aux.raw0[0] = raw0.x; aux.raw0[1] = raw0.y; aux.raw0[2] = raw0.z; aux.raw0[3] = raw0.w;
            aux.raw1[0] = raw1.x; aux.raw1[1] = raw1.y; aux.raw1[2] = raw1.z; aux.raw1[3] = raw1.w;
            aux.normal[0] = n.x; aux.normal[1] = n.y; aux.normal[2] = n.z;
            aux.depth = hz / hw;
            if (a.compute_normal && !a.aux) {
                // keep the normal live when nobody stores it, as the shader text computes it per hit
                asm volatile("" ::"v"(n.x), "v"(n.y), "v"(n.z));
            }
        }
    }

    if (a.wave_timing && lane == 0) {
        const uint64_t wave_id = ((uint64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
        a.wave_timing[wave_id * 4 + 0] = t_start;
        a.wave_timing[wave_id * 4 + 1] = __builtin_readcyclecounter();
        a.wave_timing[wave_id * 4 + 2] = (unsigned long long)iterations;
        a.wave_timing[wave_id * 4 + 3] = (unsigned long long)__ballot(covered);
    }
    if (in_image) {
        const uint64_t o = ((uint64_t)cam_idx * (a.y1 - a.y0) + row) * a.width + px;
        a.rgba[o] = rgba;
        if (a.aux) a.aux[o] = aux;
    }
}

}  // namespace

hipError_t launch_raymarch(const RaymarchArgs& a, hipStream_t stream) {
    const uint32_t rows = a.y1 - a.y0;
    if (a.width == 0 || rows == 0 || a.n_cameras == 0) return hipSuccess;
    dim3 grid((a.width + 15) / 16, (rows + 15) / 16, a.n_cameras);
    const bool linear = a.rp.lod_dist_between_samples == 1.0f;
    if (linear && a.pow2_extent)
        hipLaunchKernelGGL((raymarch_kernel<true, true>), grid, dim3(256), 0, stream, a);
    else if (linear)
        hipLaunchKernelGGL((raymarch_kernel<true, false>), grid, dim3(256), 0, stream, a);
    else if (a.pow2_extent)
        hipLaunchKernelGGL((raymarch_kernel<false, true>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((raymarch_kernel<false, false>), grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace sdfv
