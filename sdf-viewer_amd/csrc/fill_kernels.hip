// fill_kernels.hip -- grid fill on gfx950: SDFViewer::update's per-voxel loop
// (reference src/app/scene/sdf/mod.rs:173-215) as store-bound HIP kernels.
//
// Dense kernel (fill_dense_kernel): the final state of a fresh grid.  One thread per voxel, x fastest, so
// a 64-lane wave emits two contiguous 1 KiB bursts (tex0, tex1) of global_store_dwordx4.  No global
// loads at all: the position derives from the index.  Per-workgroup LDS staging holds
//   - the 256-entry sRGB->linear table (colour passes through u8, scene/sdf/mod.rs:201),
//   - the (y, z) coordinates of the workgroup's rows (idx/(dim-1)*size+min costs an IEEE divide per
//     axis: x is fixed per thread and computed once, y/z once per row instead of once per voxel);
// the demo SDF's parameter block (the "CSG-tree params") rides in the kernel arguments, i.e. SGPRs.
//
// Pass kernel (fill_pass_kernel): one LoadingManager pass with stride `step` and the update_required
// test (reads tex0.r, 4 B per visited voxel) -- the progressive / changed_box path.
#include "fill_kernels.h"

#include "demo_sdf_device.h"

namespace sdfv {

__constant__ float c_srgb_lut[256] = {
#include "srgb_lut.inc"
};

namespace {

constexpr int kBlock = 256;

struct LdsLut {
    const float* p;
    __device__ __forceinline__ float operator[](uint32_t i) const { return p[i]; }
};

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void store_texel(float4* dst, const float4& v) {
    if (NT) {  // global_store_dwordx4 ... nt: write-once stream, nothing re-reads it from L2
        v4f t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(dst));
    } else {
        *dst = v;
    }
}

// A load of data this kernel looks at ONCE (the distance volume under a pass's update_required test, tex0 under a commit):
// nontemporal -- global_load ... nt does not allocate in L2 / the Infinity Cache, so a scan that follows a fill does not have to
// push the fill's dirty lines out of the way first.  tools/ubench/read_stream.hip, 512 MiB read once behind 1 GiB of stores:
// plain loads 0.128 ms (4.2 TB/s), nt loads 0.080 ms (6.7 TB/s); eight reads in a row: 6.7 vs 7.0 TB/s.  The step-1 no-op pass
// of a loaded 512^3 grid went 0.144 -> see EXPERIMENTS R6.2.
__device__ __forceinline__ float4 load_once(const float4* p) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float load_once(const float* p) { return __builtin_nontemporal_load(p); }
// ... chosen per launch (PassArgs::stream_loads, wave-uniform)
template <typename T>
__device__ __forceinline__ T load_scan(const T* p, uint32_t stream) {
    return stream ? load_once(p) : *p;
}

// Entry of the distance volume for voxel x of slab row `row` (= z_local * H + y) in either layout (FillArgs::dist_ilv).
__device__ __forceinline__ uint64_t vol_index(uint32_t ilv, uint64_t row, uint32_t x, uint32_t W) {
    return ilv ? ((row >> 1) * W + x) * 2 + (row & 1) : row * W + x;
}

// TX = lanes along x per row segment (64, 128 or 256); a workgroup owns TY = 256 / TX consecutive rows x TX
// voxels and does ONE voxel per thread, so the grid walks memory front to back in dispatch order exactly
// like a memset.  Measured on MI355X (profiles/r01/v1_fill_sweep.json, r01/v2_fill_sweep.json): persistent
// strided workgroups lose 25-40 % of the store rate and 2/4/8 rows per thread lose 7/11/14 %.
// Boundary-first order: logical workgroup `b` of a launch -> the workgroup of the memory-order grid whose voxels it
// fills.  The first order_lead * bps workgroups are the slab's leading slices, the next bps its LAST slice, then the
// interior in memory order.  All operands are wave-uniform (SGPRs).
struct OrderedBlock {
    uint32_t block;     // memory-order workgroup index
    bool boundary;      // one of the slices a z-neighbour waits for
    bool last_slice;    // ... the slab's last one (goes to the upper neighbour)
};
__device__ __forceinline__ OrderedBlock ordered_block(const FillArgs& a, uint32_t b) {
    OrderedBlock r;
    const uint32_t lead = a.order_lead * a.order_bps;
    r.boundary = b < lead + a.order_bps;
    r.last_slice = r.boundary && b >= lead;
    r.block = r.last_slice ? (a.slab_d - 1) * a.order_bps + (b - lead) : (r.boundary ? b : b - a.order_bps);
    return r;
}

// The boundary workgroups' packed copies (one message per neighbour and direction instead of one per texture).
__device__ __forceinline__ void store_staged(const FillArgs& a, const OrderedBlock& ob, uint64_t o, const float4& v0,
                                             const float4& v1) {
    const uint64_t slice = (uint64_t)a.W * a.H;
    float4 *d0 = nullptr, *d1 = nullptr;
    if (ob.last_slice) {
        if (a.stage_hi) {
            const uint64_t w = o - (uint64_t)(a.slab_d - 1) * slice;
            d0 = a.stage_hi + w;
            d1 = a.stage_hi + slice + w;
        }
    } else if (a.stage_lo) {
        d0 = a.stage_lo + o;  // o < order_lead * slice
        d1 = a.stage_lo + a.order_lead * slice + o;
    }
    if (!d0) return;
    *d0 = v0;
    *d1 = v1;
}

template <int TX, bool NT, typename Cfg, bool ORDERED = false>
__global__ __launch_bounds__(kBlock) void fill_dense_kernel(FillArgs a) {
    constexpr int TY = kBlock / TX;
    __shared__ float s_lut[256];
    __shared__ float2 s_yz[TY];

    const uint32_t tid = threadIdx.x;
    const uint32_t n_rows = a.H * a.slab_d;  // rows of the slab: row = z_local * H + y
    // "this launch has started" = everything enqueued before it on its stream has finished: the multi-GPU fill step lets
    // the communicator's stream wait on this word (hipStreamWaitValue32) instead of on an event recorded before the fill
    if (!ORDERED && a.signal && blockIdx.x == 0 && tid == 0)
        __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    OrderedBlock ob{blockIdx.x, false, false};
    if (ORDERED) ob = ordered_block(a, blockIdx.x + a.block_base);
    // 1-D grid, x-chunk fastest: workgroup id -> (row group, x chunk); both uniform (SGPRs)
    const uint32_t row_group = a.x_chunks == 1 ? ob.block : ob.block / a.x_chunks;
    const uint32_t chunk = ob.block - row_group * a.x_chunks;
    const uint32_t row0 = row_group * TY;
    s_lut[tid] = c_srgb_lut[tid];
    if (tid < TY && row0 + tid < n_rows) {
        const uint32_t row = row0 + tid;
        const uint32_t zl = row / a.H, y = row - zl * a.H;
        s_yz[tid] = make_float2(voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]),
                                voxel_coord(a.z_begin + zl, a.dm1[2], a.bb_size[2], a.bb_min[2]));
    }
    __syncthreads();

    const LdsLut lut{s_lut};
    const uint32_t tx = tid % TX, ty = tid / TX;
    const uint32_t x = chunk * TX + tx;
    const uint32_t row = row0 + ty;
    const bool in_range = ORDERED || (x < a.W && row < n_rows);  // ordered launches cover whole workgroups only
    const bool ilv = !ORDERED && TY >= 2 && a.dist_ilv;          // block-uniform; the launcher picks TY >= 2 for this layout
    if (!ilv && !in_range) return;
    float4 v0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v1 = v0;
    const uint64_t o = (uint64_t)row * a.W + x;
    if (in_range) {
        const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
        const float2 yz = s_yz[ty];
        fill_voxel<Cfg>(a.prm, a.sdf_id, px, yz.x, yz.y, lut, a.air_dist, v0, v1);
        if (!ORDERED || !a.stage_only) {
            store_texel<NT>(a.tex0 + o, v0);
            store_texel<NT>(a.tex1 + o, v1);
            if (a.dist && !ilv) a.dist[o] = v0.x;  // wave-uniform: +4 B/voxel instead of a second pass over tex0
        }
    }
    if (ilv) {
        // y-interleaved volume: rows 2p and 2p + 1 of this workgroup meet in LDS and leave as ONE row of pairs -- 8-byte
        // stores, whole lines, from the workgroup that computed both (two workgroups writing the halves of a line would make
        // the memory side merge partial lines).  row0 is even (TY even), and so is the slab's row count (H even).
        __shared__ float s_d[kBlock];
        s_d[tid] = v0.x;
        __syncthreads();
        if ((ty & 1u) == 0 && in_range)
            reinterpret_cast<float2*>(a.dist)[(uint64_t)(row >> 1) * a.W + x] = make_float2(s_d[tid], s_d[tid + TX]);
    }
    if (ORDERED && ob.boundary) store_staged(a, ob, o, v0, v1);  // wave-uniform
}

// The dense fused fill that writes the y-INTERLEAVED volume, for widths that are multiples of 256: a thread owns voxel x of
// BOTH rows of a pair (2p, 2p + 1), so the pair's distances meet in its registers -- no LDS exchange, no second barrier --
// and a workgroup stores whole 4 KiB row segments like the plain kernel (the TX = 128 / TY = 2 form above stores 2 KiB ones
// and makes its even-row waves wait for the odd-row ones).  512^3: 0.734 against 0.800 ms (the plain-volume fused fill: 0.686).
template <bool NT, typename Cfg>
__global__ __launch_bounds__(kBlock) void fill_dense_pairrows_kernel(FillArgs a) {
    __shared__ float s_lut[256];
    __shared__ float2 s_yz[2];
    const uint32_t tid = threadIdx.x;
    if (a.signal && blockIdx.x == 0 && tid == 0)  // see fill_dense_kernel
        __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t pair = a.x_chunks == 1 ? blockIdx.x : blockIdx.x / a.x_chunks;  // pair-row p of the slab: rows 2p, 2p + 1
    const uint32_t chunk = blockIdx.x - pair * a.x_chunks;
    s_lut[tid] = c_srgb_lut[tid];
    if (tid < 2) {
        const uint32_t row = 2 * pair + tid;
        const uint32_t zl = row / a.H, y = row - zl * a.H;  // (H is even: both rows lie in one slice)
        s_yz[tid] = make_float2(voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]),
                                voxel_coord(a.z_begin + zl, a.dm1[2], a.bb_size[2], a.bb_min[2]));
    }
    __syncthreads();
    const LdsLut lut{s_lut};
    const uint32_t x = chunk * kBlock + tid;  // W is a multiple of kBlock: always inside
    const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
    const uint64_t o = (uint64_t)(2 * pair) * a.W + x;
    float4 v0, v1, w0, w1;
    const float2 yz0 = s_yz[0], yz1 = s_yz[1];
    fill_voxel<Cfg>(a.prm, a.sdf_id, px, yz0.x, yz0.y, lut, a.air_dist, v0, v1);
    store_texel<NT>(a.tex0 + o, v0);
    store_texel<NT>(a.tex1 + o, v1);
    fill_voxel<Cfg>(a.prm, a.sdf_id, px, yz1.x, yz1.y, lut, a.air_dist, w0, w1);
    store_texel<NT>(a.tex0 + o + a.W, w0);
    store_texel<NT>(a.tex1 + o + a.W, w1);
    reinterpret_cast<float2*>(a.dist)[(uint64_t)pair * a.W + x] = make_float2(v0.x, w0.x);
}

// The interleaved-volume fill in the PLAIN kernel's shape (EXPERIMENTS R5.3): (one voxel per thread, a workgroup = 256 x of ONE
// row: three store segments per workgroup, like the fused fill that writes the plain volume), the two rows of a pair filled by
// two workgroups of the SAME XCD one dispatch slot apart, so that the halves of every volume line (4 bytes in every 8) meet in
// that XCD's write-back L2 instead of at the memory side.  Workgroup b runs on XCD b % 8 (the placement the march's tile orders
// rest on): slot = b / 8, unit = (slot / 2) * 8 + b % 8 = (pair-row, x chunk), half = slot & 1 = which row of the pair.
template <bool NT, typename Cfg>
__global__ __launch_bounds__(kBlock) void fill_dense_ilv_paired_kernel(FillArgs a, uint32_t n_units) {
    __shared__ float s_lut[256];
    __shared__ float2 s_yz;
    const uint32_t tid = threadIdx.x;
    if (a.signal && blockIdx.x == 0 && tid == 0)  // see fill_dense_kernel
        __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t slot = blockIdx.x >> 3, unit = (slot >> 1) * 8u + (blockIdx.x & 7u), half = slot & 1u;
    if (unit >= n_units) return;  // (the grid is padded to whole groups of 16)
    const uint32_t pair = a.x_chunks == 1 ? unit : unit / a.x_chunks;
    const uint32_t chunk = unit - pair * a.x_chunks;
    const uint32_t row = 2u * pair + half;
    s_lut[tid] = c_srgb_lut[tid];
    if (tid == 0) {
        const uint32_t zl = row / a.H, y = row - zl * a.H;
        s_yz = make_float2(voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]),
                           voxel_coord(a.z_begin + zl, a.dm1[2], a.bb_size[2], a.bb_min[2]));
    }
    __syncthreads();
    const LdsLut lut{s_lut};
    const uint32_t x = chunk * kBlock + tid;  // W is a multiple of kBlock: always inside
    const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
    const uint64_t o = (uint64_t)row * a.W + x;
    float4 v0, v1;
    const float2 yz = s_yz;
    fill_voxel<Cfg>(a.prm, a.sdf_id, px, yz.x, yz.y, lut, a.air_dist, v0, v1);
    store_texel<NT>(a.tex0 + o, v0);
    store_texel<NT>(a.tex1 + o, v1);
    a.dist[((uint64_t)pair * a.W + x) * 2 + half] = v0.x;
}

// Flat form of the dense kernel for widths that do not fill the row-chunk form's lanes (W not a multiple of the
// 64 / 128 / 256 chunk): thread <-> voxel over the slab's flat index, so every wave is full and every store burst
// is 1 KiB whatever W is.  x and row come from an exact division by W done as a 64-bit multiply-high with
// M = floor(2^64 / W) + 1 (exact for dividends below 2^32); the (y, z) coordinates of the few rows a workgroup
// touches are staged in LDS by its first lanes.
template <bool NT, typename Cfg, bool STRIDED = false, bool ORDERED = false>
__global__ __launch_bounds__(kBlock) void fill_dense_flat_kernel(FillArgs a) {
    __shared__ float s_lut[256];
    __shared__ float2 s_yz[kBlock + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_vox = a.W * a.H * a.slab_d;  // < 2^32, checked by the launcher
    if (!ORDERED && a.signal && blockIdx.x == 0 && tid == 0)  // see fill_dense_kernel
        __hip_atomic_store(a.signal, a.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    OrderedBlock ob{blockIdx.x, false, false};
    if (ORDERED) ob = ordered_block(a, blockIdx.x + a.block_base);  // slices are whole numbers of workgroups here
    const uint32_t v0 = ob.block * kBlock;
    auto div_w = [&](uint32_t v) -> uint32_t {
        if (a.W == 1) return v;
        const unsigned long long t = ((unsigned long long)v * (uint32_t)a.w_magic) >> 32;
        return (uint32_t)(((unsigned long long)v * (uint32_t)(a.w_magic >> 32) + t) >> 32);
    };
    const uint32_t row_first = div_w(v0);
    const uint32_t v_last = min(v0 + kBlock - 1, n_vox - 1);
    const uint32_t n_rows_here = div_w(v_last) - row_first + 1;  // <= kBlock + 1
    s_lut[tid] = c_srgb_lut[tid];
    for (uint32_t r = tid; r < n_rows_here; r += kBlock) {
        const uint32_t row = row_first + r;
        const uint32_t zl = row / a.H, y = row - zl * a.H;
        const uint32_t z = a.z_begin + (STRIDED ? zl * a.z_step : zl);
        s_yz[r] = make_float2(voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]),
                              voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]));
    }
    __syncthreads();
    const uint32_t v = v0 + tid;
    if (!ORDERED && v >= n_vox) return;  // ordered launches cover whole workgroups only
    const uint32_t row = div_w(v);
    const uint32_t x = v - row * a.W;
    const LdsLut lut{s_lut};
    const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
    const float2 yz = s_yz[row - row_first];
    float4 v0t, v1t;
    fill_voxel<Cfg>(a.prm, a.sdf_id, px, yz.x, yz.y, lut, a.air_dist, v0t, v1t);
    size_t at = v;
    if (STRIDED) at += (size_t)(row / a.H) * (a.z_step - 1) * a.H * a.W;  // the slices skipped in between
    if (!ORDERED || !a.stage_only) {
        store_texel<NT>(a.tex0 + at, v0t);
        store_texel<NT>(a.tex1 + at, v1t);
        if (a.dist) a.dist[at] = v0t.x;
    }
    if (ORDERED && ob.boundary) store_staged(a, ob, at, v0t, v1t);  // wave-uniform
}

// Ghost slices out of the packed receive buffers (one launch for up to four contiguous copies).
__global__ __launch_bounds__(kBlock) void copy_segments_kernel(CopySegments c) {
    const uint32_t seg = blockIdx.y;
    const uint32_t n = c.n[seg];
    const float4* __restrict__ src = c.src[seg];
    float4* __restrict__ dst = c.dst[seg];
    float* __restrict__ r_out = c.r_out[seg];  // block-uniform
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 t = src[i];
        dst[i] = t;
        if (r_out) r_out[i] = t.x;
    }
}

// Exact v / d for v < 2^32 (DivU32 made by the launcher): a shift for powers of two, otherwise the 64-bit multiply-high
// with M = floor((2^64 - 1) / d) + 1 the flat dense kernel uses.  All fields are wave-uniform (SGPRs): one scalar branch.
__device__ __forceinline__ uint32_t div_u32(uint32_t v, const DivU32& d) {
    if (d.pow2) return v >> d.shift;
    const unsigned long long t = ((unsigned long long)v * (uint32_t)d.magic) >> 32;
    return (uint32_t)(((unsigned long long)v * (uint32_t)(d.magic >> 32) + t) >> 32);
}

// update_required (scene/sdf/mod.rs:184-190) for one visited voxel whose stored tex0.r is not AIR_DIST: inside the
// changed box?  A cheap estimate of the coordinates (one multiply-add per axis; the exact ones differ from it by less than
// approx_margin) rejects what is clearly outside before anyone pays for three correctly rounded divides.
__device__ __forceinline__ bool maybe_in_box(const FillArgs& a, const PassArgs& p, uint32_t x, uint32_t y, uint32_t z) {
    const float ex = (float)x * p.approx_scale[0] + a.bb_min[0];
    const float ey = (float)y * p.approx_scale[1] + a.bb_min[1];
    const float ez = (float)z * p.approx_scale[2] + a.bb_min[2];
    return !(ex < p.box[0] - p.approx_margin[0] || ex > p.box[3] + p.approx_margin[0] ||
             ey < p.box[1] - p.approx_margin[1] || ey > p.box[4] + p.approx_margin[1] ||
             ez < p.box[2] - p.approx_margin[2] || ez > p.box[5] + p.approx_margin[2]);
}

// The rewrite of one voxel that update_required holds for.  No LDS staging of the colour table in the pass kernels: most
// workgroups of a pass over a loaded grid have nothing to update and would pay the staging and its barrier for nothing; the
// few lookups (the "normal" material only) read the 1 KiB table through the caches.
template <typename Cfg>
__device__ __forceinline__ void pass_store(const FillArgs& a, const PassArgs& p, float px, float py, float pz, uint64_t flat,
                                           uint64_t row, uint32_t x) {
    const LdsLut lut{c_srgb_lut};
    float4 v0, v1;
    fill_voxel<Cfg>(a.prm, a.sdf_id, px, py, pz, lut, a.air_dist, v0, v1);
    // Texture stores stream past L2 (nt), like the fused dense fill's: nothing re-reads them before the march's few texels
    // under the hits, while the volume -- which this very pass READS -- keeps the cache.  tools/ubench/rw_mix.hip: a
    // 4 B/voxel read stream next to the 36 B/voxel store stream costs 0.108 ms with plain stores and 0.094 with nt at 256^3.
    store_texel<true>(a.tex0 + flat, v0);
    if (p.dist) {
        // The volume's contract: the textures were initialised / filled by this library, so tex1.a holds AIR_DIST
        // everywhere (update() never writes it, scene/sdf/mod.rs:205-208) and need not be read back.
        p.dist[vol_index(a.dist_ilv, row, x, a.W)] = v0.x;
        v1.w = a.air_dist;
    } else {
        // tex1.a is not update()'s to touch: carry the stored value through a full 16-byte store (12-byte partial
        // stores make the memory side read-modify-write the line; ~10 % slower on the step-1 pass)
        v1.w = reinterpret_cast<const float*>(a.tex1 + flat)[3];
    }
    store_texel<true>(a.tex1 + flat, v1);
}

// One LoadingManager pass, any step: one thread per visited voxel (x, y, z multiples of `step`; z is GLOBAL), workgroups in
// memory order like the dense kernel.  Reads tex0.r (or the distance volume's entry) for update_required; with
// p.all_required (the launcher knows that it holds for every visited voxel) nothing is read at all.
template <typename Cfg>
__global__ __launch_bounds__(kBlock) void fill_pass_kernel(FillArgs a, PassArgs p) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // < 2^32 visited voxels: checked by the launcher
    if (i >= p.n_visited) return;
    const uint32_t r = div_u32(i, p.div_nx), ix = i - r * p.nx;
    const uint32_t iz = div_u32(r, p.div_ny), iy = r - iz * p.ny;
    const uint32_t x = ix * p.step, y = iy * p.step, z = p.z_first + iz * p.step;  // global z
    const uint64_t row = (uint64_t)(z - a.z_begin) * a.H + y;
    const uint64_t flat = row * a.W + x;
    if (!p.all_required) {
        const bool is_air = (p.dist ? load_scan(p.dist + vol_index(a.dist_ilv, row, x, a.W), p.stream_loads) : load_scan(&a.tex0[flat].x, p.stream_loads)) == a.air_dist;
        // most visited voxels of a loaded grid need nothing: leave before paying three correctly rounded divides
        if (!is_air && (!p.has_box || !maybe_in_box(a, p, x, y, z))) return;
        const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
        const float py = voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]);
        const float pz = voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]);
        if (!is_air && !(px >= p.box[0] && px <= p.box[3] && py >= p.box[1] && py <= p.box[4] && pz >= p.box[2] && pz <= p.box[5]))
            return;
        pass_store<Cfg>(a, p, px, py, pz, flat, row, x);
        return;
    }
    pass_store<Cfg>(a, p, voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]), voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]),
                    voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]), flat, row, x);
}

// The step-1 pass over a grid with its distance volume.  A pass over a loaded grid is bound by per-wave latency (one
// load, then exit), not by bytes, so each lane answers update_required for FOUR x-neighbours with one 16-byte load of
// the volume and a wave with nothing to do leaves after that single load (a quarter of the waves, a quarter of the
// time).  Where there is work, the wave's 256 voxels are processed in four rounds of 64 CONSECUTIVE voxels (lane l
// takes voxel 64*j + l of the span, its flag fetched from the lane that loaded it), so texel stores stay coalesced.
// A pass that updates most of what it visits (a fresh grid) is bound by VALU issue before it is bound by bytes, so the
// per-voxel path is the dense kernel's: index arithmetic by shifts / multiply-high (no integer division), the row's
// (y, z) coordinates -- two correctly rounded divides -- recomputed only when a round enters a new row, compile-time
// packed materials for the default configuration.  Needs W % 4 == 0, a 16-byte aligned volume, < 2^32 voxels.
template <typename Cfg>
__global__ __launch_bounds__(kBlock) void fill_pass_quad_kernel(FillArgs a, PassArgs p) {
    const uint32_t n_vox = p.n_visited;  // visited voxels of the slab (step 1: all of them)
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;  // this lane's quad of voxels [4q, 4q + 4)
    const uint32_t lane = threadIdx.x & 63;
    uint32_t air_bits = 0;
    if (4 * (uint64_t)q < n_vox) {
        float4 d;
        if (a.dist_ilv) {  // the quad's four x-neighbours are every other float of 32 contiguous bytes of its pair-row
            const uint32_t qr = div_u32(4u * q, p.div_nx), qx = 4u * q - qr * a.W;
            const float* b = p.dist + ((uint64_t)(qr >> 1) * a.W + qx) * 2;
            const float4 lo = load_scan(reinterpret_cast<const float4*>(b), p.stream_loads), hi = load_scan(reinterpret_cast<const float4*>(b + 4), p.stream_loads);
            d = (qr & 1u) ? make_float4(lo.y, lo.w, hi.y, hi.w) : make_float4(lo.x, lo.z, hi.x, hi.z);
        } else {
            d = load_scan(reinterpret_cast<const float4*>(p.dist + (size_t)q * 4), p.stream_loads);
        }
        air_bits = (d.x == a.air_dist ? 1u : 0u) | (d.y == a.air_dist ? 2u : 0u) | (d.z == a.air_dist ? 4u : 0u) |
                   (d.w == a.air_dist ? 8u : 0u);
    }
    if (!p.has_box && __ballot(air_bits != 0) == 0ull) return;  // wave-uniform: nothing to update in these 256 voxels
    if (p.has_box) {
        // ... and with a box: a wave none of whose quads holds AIR or can reach into the box leaves here too, on ONE coordinate
        // estimate per lane (its quad's row and x range against the box, maybe_in_box's margins) instead of four rounds of
        // per-voxel estimates -- seven waves in eight of an edit that touches an eighth of the grid (round 5)
        bool reach = false;
        if (4 * (uint64_t)q < n_vox) {
            const uint32_t qr = div_u32(4u * q, p.div_nx), qx = 4u * q - qr * a.W;
            const uint32_t qz = div_u32(qr, p.div_ny), qy = qr - qz * a.H;
            // either end of the quad may be in the box, or the box's x range lies strictly between the quad's ends (whichever way
            // the coordinates run)
            const float e0 = (float)qx * p.approx_scale[0] + a.bb_min[0], e3 = (float)(qx + 3u) * p.approx_scale[0] + a.bb_min[0];
            reach = maybe_in_box(a, p, qx, qy, p.z_first + qz) || maybe_in_box(a, p, qx + 3u, qy, p.z_first + qz) ||
                    (fminf(e0, e3) < p.box[0] && fmaxf(e0, e3) > p.box[3]);
        }
        if (__ballot(air_bits != 0 || reach) == 0ull) return;
    }
    const uint32_t span0 = (q - lane) * 4;  // first voxel of the wave's span (flat index within the slab)
    uint32_t row_cached = 0xffffffffu;
    float py = 0.0f, pz = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t bits = (uint32_t)__shfl((int)air_bits, (int)(j * 16 + (lane >> 2)));
        const uint32_t v = span0 + j * 64 + lane;
        const bool is_air = ((bits >> (lane & 3)) & 1u) != 0;
        const uint32_t r = div_u32(v, p.div_nx), x = v - r * a.W;   // step 1: nx == W
        const uint32_t zl = div_u32(r, p.div_ny), y = r - zl * a.H;  //         ny == H
        const uint32_t z = p.z_first + zl;
        bool work = v < n_vox && (is_air || (p.has_box && maybe_in_box(a, p, x, y, z)));
        if (__ballot(work) == 0ull) continue;  // wave-uniform
        if (__ballot(work && r != row_cached) != 0ull) {  // wave-uniform: some lane entered a new row
            row_cached = r;
            py = voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]);
            pz = voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]);
        }
        if (!work) continue;
        const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
        if (!is_air && !(px >= p.box[0] && px <= p.box[3] && py >= p.box[1] && py <= p.box[4] && pz >= p.box[2] && pz <= p.box[5]))
            continue;
        pass_store<Cfg>(a, p, px, py, pz, v, r, x);
    }
}

// A pass with step >= 2 in which update_required holds for every visited voxel (nothing to decide, see PassArgs::all_required),
// written as WHOLE ROWS: one thread per voxel of a visited row (rows: y and global z multiples of step), memory order.  Lanes
// on the visited lattice store their sample; the lanes between them re-store what the voxel holds -- over a FRESH grid
// (every voxel is new_voxels' [AIR_DIST; 4], scene/sdf/mod.rs:76-77) that is a constant and nothing is read at all,
// otherwise the texel is loaded and written back.  Either way the memory side sees whole 128-byte lines instead of one
// 16-byte texel in every `step`: a strided pass of single texels makes it read-modify-write every line it touches, one
// texel at a time (step 2 over a 256^3 grid: 0.087 ms for 75 MB of samples -- as long as the dense fill of the whole grid).
template <typename Cfg, bool FRESH>
__global__ __launch_bounds__(kBlock) void fill_pass_rows_kernel(FillArgs a, PassArgs p) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;  // < 2^32: checked by the launcher
    if (t >= a.W * p.ny * p.nz) return;
    const uint32_t r = div_u32(t, p.div_w), x = t - r * a.W;
    const uint32_t iz = div_u32(r, p.div_ny), iy = r - iz * p.ny;
    const uint32_t y = iy * p.step, z = p.z_first + iz * p.step;
    const uint64_t flat = ((uint64_t)(z - a.z_begin) * a.H + y) * a.W + x;
    float4 v0 = make_float4(a.air_dist, a.air_dist, a.air_dist, a.air_dist), v1 = v0;
    const bool visited = (x & (p.step - 1)) == 0;
    if (visited) {
        const LdsLut lut{c_srgb_lut};
        fill_voxel<Cfg>(a.prm, a.sdf_id, voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]),
                        voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]), voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]),
                        lut, a.air_dist, v0, v1);
        // tex1.a is not update()'s to touch (scene/sdf/mod.rs:205-208): AIR_DIST in a fresh grid and under the volume's
        // contract, otherwise whatever the voxel holds
        if (!FRESH && !p.dist) v1.w = reinterpret_cast<const float*>(a.tex1 + flat)[3];
    } else if (!FRESH) {
        v0 = a.tex0[flat];
        v1 = a.tex1[flat];
    }
    store_texel<true>(a.tex0 + flat, v0);
    store_texel<true>(a.tex1 + flat, v1);
    if (!p.dist) return;
    if (!a.dist_ilv) {
        p.dist[flat] = v0.x;  // every lane: the volume equals tex0.r (its contract), so whole lines here too
        return;
    }
    // y-interleaved volume: a visited row is an EVEN row (step >= 2, H even) and shares its pair-row with the odd row after
    // it, which this pass does not visit.  Over a fresh / virgin grid that neighbour is (logically) AIR: whole pairs are
    // written; otherwise its half of the pair is carried through.
    float2* pr = reinterpret_cast<float2*>(p.dist) + ((uint64_t)(z - a.z_begin) * a.H + y) / 2 * a.W + x;
    *pr = make_float2(v0.x, FRESH ? a.air_dist : pr->y);
}

// A pass with step >= 2 over a grid the caller says NOTHING about (no flags, no box), with its distance volume: the whole
// visited rows again, but each wave decides for itself what it is looking at ("a caller that says nothing
// stops paying 3x").  Each lane answers for FOUR x-neighbours of a visited row with one 16-byte load of the volume (the same
// cache lines a load of the lattice points alone would touch, a quarter of the threads); a wave none of whose 256 voxels has a
// lattice point holding AIR_DIST leaves after that load (a pass over a loaded grid).  Where there is work, four rounds of 64
// consecutive voxels as in the quad kernel: a round ALL of whose voxels hold AIR_DIST -- the fresh grid, or the rows a
// coarser pass has not visited -- is written WHOLE, samples on the lattice and new_voxels' [AIR_DIST; 4] between them, no
// texel read and no partial line (the memory side read-modify-writes a line that receives one 16-byte texel in every
// `step`: 2.2x the bytes at step 2); any other round stores its lattice samples alone, as fill_pass_kernel does.
// "tex0.r == AIR_DIST means the voxel holds new_voxels' state" is the reference's own reading of that value
// (update_required, scene/sdf/mod.rs:184-190) and, with a volume, this library's contract for the textures behind it.
// Needs W % 4 == 0, a 16-byte aligned volume (either layout), < 2^32 voxels in the visited rows.
template <typename Cfg, int Q>
__global__ __launch_bounds__(kBlock) void fill_pass_rows_adaptive_kernel(FillArgs a, PassArgs p) {
    // Q voxels per lane = ONE 16-byte load of the volume: four x-neighbours of a plain volume (Q = 4), two of the interleaved
    // one, whose 16 bytes hold (even row, odd row) of two x (Q = 2; two loads per lane ran a no-op pass at 0.57 of this rate)
    static_assert(Q == 4 || Q == 2, "one 16-byte load per lane");
    const uint32_t n_row_vox = a.W * p.ny * p.nz;  // voxels of the visited rows (< 2^32: checked by the launcher)
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;  // this lane's voxels [Q q, Q q + Q) of that index space
    const uint32_t lane = threadIdx.x & 63;
    uint32_t bits = 0;  // bit k: voxel k of the lane holds AIR_DIST; bit 4 + k: it is on the visited lattice
    float odd0 = 0.0f, odd1 = 0.0f;  // interleaved volume: the lane's partners in the odd row of the pair
    if (Q * (uint64_t)q < n_row_vox) {
        const uint32_t rr = div_u32(Q * q, p.div_w), x = Q * q - rr * a.W;
        const uint32_t iz = div_u32(rr, p.div_ny), iy = rr - iz * p.ny;
        const uint64_t row = (uint64_t)(p.z_first - a.z_begin + iz * p.step) * a.H + iy * p.step;
        const uint32_t m = p.step - 1u;
        if (Q == 2) {  // a visited row is an even row (step >= 2, H even): the .x halves of its pair-row
            const float4 d = load_scan(reinterpret_cast<const float4*>(p.dist + ((row >> 1) * a.W + x) * 2), p.stream_loads);
            odd0 = d.y, odd1 = d.w;
            bits = (d.x == a.air_dist ? 1u : 0u) | (d.z == a.air_dist ? 2u : 0u) | ((x & m) == 0 ? 16u : 0u) | (((x + 1u) & m) == 0 ? 32u : 0u);
        } else {
            const float4 d = load_scan(reinterpret_cast<const float4*>(p.dist + row * a.W + x), p.stream_loads);
            bits = (d.x == a.air_dist ? 1u : 0u) | (d.y == a.air_dist ? 2u : 0u) | (d.z == a.air_dist ? 4u : 0u) |
                   (d.w == a.air_dist ? 8u : 0u) | ((x & m) == 0 ? 16u : 0u) | (((x + 1u) & m) == 0 ? 32u : 0u) |
                   (((x + 2u) & m) == 0 ? 64u : 0u) | (((x + 3u) & m) == 0 ? 128u : 0u);
        }
    }
    if (__ballot((bits & (bits >> 4)) != 0) == 0ull) return;  // wave-uniform: no lattice point of these 64 Q voxels is AIR
    const uint32_t span0 = (q - lane) * Q;
    const uint32_t k = lane & (Q - 1u);
    uint32_t row_cached = 0xffffffffu;
    float py = 0.0f, pz = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < Q; ++j) {
        const int src = (int)(j * (64 / Q) + lane / Q);
        const uint32_t b = (uint32_t)__shfl((int)bits, src);
        const uint32_t v = span0 + j * 64 + lane;
        const bool in = v < n_row_vox;
        const bool is_air = ((b >> k) & 1u) != 0, is_lat = ((b >> (4u + k)) & 1u) != 0;
        const bool need = in && is_air && is_lat;  // update_required without a box: the stored distance is AIR_DIST
        if (__ballot(need) == 0ull) continue;      // wave-uniform
        const bool whole = __ballot(in && !is_air) == 0ull;  // wave-uniform: every voxel of the round is still new_voxels'
        float partner = 0.0f;
        if (Q == 2 && whole) {  // (wave-uniform) the odd row's half of this voxel's pair, from the lane that loaded it
            const float o0 = __shfl(odd0, src), o1 = __shfl(odd1, src);
            partner = k == 0 ? o0 : o1;
        }
        const uint32_t rr = div_u32(v, p.div_w), x = v - rr * a.W;
        const uint32_t iz = div_u32(rr, p.div_ny), iy = rr - iz * p.ny;
        const uint32_t y = iy * p.step, z = p.z_first + iz * p.step;
        if (__ballot(need && rr != row_cached) != 0ull) {  // wave-uniform: some lane entered a new row
            row_cached = rr;
            py = voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]);
            pz = voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]);
        }
        if (!in) continue;
        const uint64_t row = (uint64_t)(z - a.z_begin) * a.H + y, flat = row * a.W + x;
        if (!whole) {
            if (need) pass_store<Cfg>(a, p, voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]), py, pz, flat, row, x);
            continue;
        }
        float4 v0 = make_float4(a.air_dist, a.air_dist, a.air_dist, a.air_dist), v1 = v0;
        if (is_lat) {
            const LdsLut lut{c_srgb_lut};
            fill_voxel<Cfg>(a.prm, a.sdf_id, voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]), py, pz, lut, a.air_dist, v0, v1);
            v1.w = a.air_dist;  // the volume's contract (pass_store)
        }
        store_texel<true>(a.tex0 + flat, v0);
        store_texel<true>(a.tex1 + flat, v1);
        if (Q == 2) reinterpret_cast<float2*>(p.dist)[(row >> 1) * a.W + x] = make_float2(v0.x, partner);
        else p.dist[flat] = v0.x;  // every lane: whole lines here too
    }
}

__global__ __launch_bounds__(kBlock) void grid_init_kernel(float4* tex0, float4* tex1, uint64_t n, float air) {
    const float4 v = make_float4(air, air, air, air);
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;  // memory order, like the dense fill
    if (i < n) {
        tex0[i] = v;
        tex1[i] = v;
    }
}

// Lazy half of new_voxels' initial state for a grid whose first passes ran with SDFV_PASS_VIRGIN_GRID: [AIR_DIST; 4] into
// every row those passes did NOT write (y or global z not a multiple of `step`; step 0 = every row), and AIR_DIST into the
// distance volume's entries of those rows.  One workgroup per row chunk (x chunks, y, local z): no index division, the
// workgroups of written rows leave at once.
__global__ __launch_bounds__(kBlock) void grid_init_unvisited_kernel(float4* tex0, float4* tex1, float* dist, uint32_t W,
                                                                     uint32_t H, uint32_t z_begin, uint32_t step, float air,
                                                                     uint32_t dist_ilv) {
    const uint32_t y = blockIdx.y, zl = blockIdx.z;
    if (step != 0 && (y & (step - 1)) == 0 && ((z_begin + zl) & (step - 1)) == 0) return;  // a row the passes wrote whole
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const uint64_t i = ((uint64_t)zl * H + y) * W + x;
    const float4 v = make_float4(air, air, air, air);
    store_texel<true>(tex0 + i, v);
    store_texel<true>(tex1 + i, v);
    if (dist) dist[vol_index(dist_ilv, (uint64_t)zl * H + y, x, W)] = air;
}

bool is_default_config(const FillArgs& a) {
    return a.sdf_id == SDFV_SDF_DEMO && a.prm.cube_material == SDFV_MATERIAL_BRICK &&
           a.prm.sphere_material == SDFV_MATERIAL_NORMAL && a.prm.disable_sphere == 0;
}

// Calls launch(Cfg{}) with the kernels' configuration policy for these arguments: the static default configuration or the
// one read from the parameter block, each with either Srgba::from policy (FillArgs::srgb_round, SDFV_OPT_EXT_SRGB_QUANT).
template <typename Launch>
void with_cfg(const FillArgs& a, Launch&& launch) {
    const bool dflt = is_default_config(a);
    if (a.srgb_round) {
        if (dflt) launch(DefaultCfgT<true>{});
        else launch(RuntimeCfgT<true>{});
    } else {
        if (dflt) launch(DefaultCfgT<false>{});
        else launch(RuntimeCfgT<false>{});
    }
}
#define SDFV_LAUNCH_CFG(a, KERNEL, ...) \
    with_cfg(a, [&](auto cfg_) {        \
        using Cfg = decltype(cfg_);     \
        hipLaunchKernelGGL(KERNEL, __VA_ARGS__); \
    })

// SDFViewer::commit's device-side analogue: nothing to upload, but the raymarch likes a compact copy of
// tex0.r (4 B/voxel instead of one dword in every 16 B).  Reads whole texels (coalesced dwordx4) and writes
// one dword per voxel: 20 B/voxel of traffic, once per commit.
__global__ __launch_bounds__(kBlock) void commit_distance_kernel(const float4* __restrict__ tex0,
                                                                 float* __restrict__ dist, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
        dist[i] = load_once(tex0 + i).x;
}

// The y-pair volume the hand-written march loop gathers from (raymarch_kernels.hip SDFV_MARCH_ASM_INTERIOR_PAIRS): texel
// (x, y, z) = (d[z][y][x], d[z][min(y + 1, H - 1)][x]).  Reads the compact distance volume twice (the second time one row
// on: served by the caches), writes 8 B/voxel; once per load, pays for itself from the second frame on.
// Launched over (x chunks, y, z): no index division; the stores stream past L2 (nt), the reads keep it.
__global__ __launch_bounds__(kBlock) void commit_pairs_kernel(const float* __restrict__ dist, float2* __restrict__ pairs,
                                                              uint32_t W, uint32_t H) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= W) return;
    const uint64_t i = ((uint64_t)z * H + y) * W + x;
    const float d0 = dist[i];
    const float d1 = y + 1 < H ? dist[i + W] : d0;
    v2f v;
    v.x = d0;
    v.y = d1;
    __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(pairs) + i);
}

// The y-interleaved volume (raymarch_kernels.hip SDFV_MARCH_ASM_INTERIOR_ILV): rows 2p and 2p + 1 of a slice stored as one
// row of (d[2p][x], d[2p+1][x]) pairs; 4 B/voxel, H even.  One thread per PAIR: two coalesced 4-byte reads, one 8-byte store.
// (Flat index, cached accesses: the (x, p, z) launch with streamed stores that serves commit_pairs_kernel is 10 % slower here.)
__global__ __launch_bounds__(kBlock) void commit_interleaved_kernel(const float* __restrict__ dist, float2* __restrict__ ilv,
                                                                    uint32_t W, uint64_t n_pairs) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;  // pair (x, p, z): i = (z * H/2 + p) * W + x
    if (i >= n_pairs) return;
    const uint64_t prow = i / W;
    const uint32_t x = (uint32_t)(i - prow * W);
    const uint64_t src = prow * 2 * W + x;  // row 2p of that slice: (z * H + 2p) * W + x
    ilv[i] = make_float2(dist[src], dist[src + W]);
}

template <int TX, bool NT>
hipError_t launch_dense_cfg(const FillArgs& args, hipStream_t stream) {
    constexpr int TY = kBlock / TX;
    FillArgs a = args;
    a.x_chunks = (a.W + TX - 1) / TX;
    const uint64_t n_rows = (uint64_t)a.H * a.slab_d;
    const uint64_t blocks = a.x_chunks * ((n_rows + TY - 1) / TY);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    dim3 grid((uint32_t)blocks, 1, 1);
    SDFV_LAUNCH_CFG(a, (fill_dense_kernel<TX, NT, Cfg>), grid, dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

template <bool NT>
hipError_t launch_dense_pairrows(const FillArgs& args, hipStream_t stream) {
    FillArgs a = args;
    a.x_chunks = a.W / kBlock;
    const uint64_t blocks = a.x_chunks * ((uint64_t)a.H * a.slab_d / 2);
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    SDFV_LAUNCH_CFG(a, (fill_dense_pairrows_kernel<NT, Cfg>), dim3((uint32_t)blocks), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

template <bool NT>
hipError_t launch_dense_ilv_paired(const FillArgs& args, hipStream_t stream) {
    FillArgs a = args;
    a.x_chunks = a.W / kBlock;
    const uint64_t units = a.x_chunks * ((uint64_t)a.H * a.slab_d / 2);
    const uint64_t blocks = ((units + 7) / 8) * 16;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    SDFV_LAUNCH_CFG(a, (fill_dense_ilv_paired_kernel<NT, Cfg>), dim3((uint32_t)blocks), dim3(kBlock), 0, stream, a, (uint32_t)units);
    return hipGetLastError();
}

template <int TX>
hipError_t launch_dense_tx(const FillArgs& a, const FillLaunch& cfg, hipStream_t stream) {
    return cfg.nontemporal ? launch_dense_cfg<TX, true>(a, stream) : launch_dense_cfg<TX, false>(a, stream);
}

}  // namespace

template <bool NT>
hipError_t launch_dense_flat(const FillArgs& args, hipStream_t stream) {
    FillArgs a = args;
    const uint64_t n_vox = (uint64_t)a.W * a.H * a.slab_d;
    // floor((2^64 - 1) / W) + 1: floor(2^64 / W) + 1 when W does not divide 2^64, exactly 2^64 / W when it does
    a.w_magic = a.W > 1 ? (~0ull / a.W) + 1ull : 0ull;
    const uint32_t blocks = (uint32_t)((n_vox + kBlock - 1) / kBlock);
    SDFV_LAUNCH_CFG(a, (fill_dense_flat_kernel<NT, Cfg>), dim3(blocks), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

namespace {
// Which index form an ordered launch uses, and its workgroup counts: the row-chunk form when rows fill whole
// workgroups in both directions, else the flat form when a slice is a whole number of workgroups, else none.
struct OrderedPlan {
    int tx;  // 64 / 128 / 256 = row-chunk form, 0 = flat form, -1 = not possible
    uint32_t per_slice, total;
};
OrderedPlan plan_ordered(const FillArgs& a) {
    OrderedPlan p{-1, 0, 0};
    if (a.W == 0 || a.H == 0 || a.slab_d == 0) return p;
    const uint64_t slice = (uint64_t)a.W * a.H, n_vox = slice * a.slab_d;
    if (n_vox >= (1ull << 32)) return p;
    const uint32_t chunk = a.W <= 64 ? 64 : (a.W <= 128 ? 128 : 256);
    const uint32_t ty = kBlock / chunk;
    if (a.W % chunk == 0 && a.H % ty == 0) {
        p.tx = (int)chunk;
        p.per_slice = (a.W / chunk) * (a.H / ty);
    } else if (slice % kBlock == 0) {
        p.tx = 0;
        p.per_slice = (uint32_t)(slice / kBlock);
    } else {
        return p;
    }
    p.total = p.per_slice * a.slab_d;
    return p;
}

template <int TX>
void launch_ordered_rows(const FillArgs& args, uint32_t blocks, hipStream_t stream) {
    FillArgs a = args;
    a.x_chunks = (a.W + TX - 1) / TX;
    SDFV_LAUNCH_CFG(a, (fill_dense_kernel<TX, false, Cfg, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
}
}  // namespace

OrderedBlocks ordered_blocks(const FillArgs& a) {
    const OrderedPlan p = plan_ordered(a);
    return p.tx < 0 ? OrderedBlocks{0, 0} : OrderedBlocks{p.per_slice, p.total};
}

hipError_t launch_fill_dense_ordered(const FillArgs& args, uint32_t block_begin, uint32_t block_end, hipStream_t stream) {
    const OrderedPlan p = plan_ordered(args);
    if (p.tx < 0 || block_begin > block_end || block_end > p.total || args.order_lead == 0 ||
        args.order_lead + 1 >= args.slab_d || args.dist_ilv)  // (the boundary-first order writes the plain volume only)
        return hipErrorInvalidValue;
    if (block_begin == block_end) return hipSuccess;
    FillArgs a = args;
    a.order_bps = p.per_slice;
    a.block_base = block_begin;
    const uint32_t blocks = block_end - block_begin;
    if (p.tx == 64) launch_ordered_rows<64>(a, blocks, stream);
    else if (p.tx == 128) launch_ordered_rows<128>(a, blocks, stream);
    else if (p.tx == 256) launch_ordered_rows<256>(a, blocks, stream);
    else {
        a.w_magic = a.W > 1 ? (~0ull / a.W) + 1ull : 0ull;
        SDFV_LAUNCH_CFG(a, (fill_dense_flat_kernel<false, Cfg, false, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_copy_segments(const CopySegments& c, hipStream_t stream) {
    uint32_t n_max = 0, segs = 0;
    for (int i = 0; i < 4; ++i)
        if (c.n[i]) {
            segs = i + 1;
            if (c.n[i] > n_max) n_max = c.n[i];
        }
    if (segs == 0) return hipSuccess;
    const uint32_t bx = (n_max + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(copy_segments_kernel, dim3(bx < 4096 ? bx : 4096, segs), dim3(kBlock), 0, stream, c);
    return hipGetLastError();
}

hipError_t launch_fill_slices(const FillArgs& args, hipStream_t stream) {
    FillArgs a = args;
    const uint64_t n_vox = (uint64_t)a.W * a.H * a.slab_d;
    if (n_vox == 0) return hipSuccess;
    if (n_vox >= (1ull << 32) || a.z_step == 0) return hipErrorInvalidValue;
    a.w_magic = a.W > 1 ? (~0ull / a.W) + 1ull : 0ull;
    const uint32_t blocks = (uint32_t)((n_vox + kBlock - 1) / kBlock);
    SDFV_LAUNCH_CFG(a, (fill_dense_flat_kernel<false, Cfg, true>), dim3(blocks), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_fill_dense(const FillArgs& a, const FillLaunch& cfg, hipStream_t stream) {
    if (a.W == 0 || a.H == 0 || a.slab_d == 0) return hipSuccess;
    if (a.dist && a.dist_ilv) {
        // the interleaved volume leaves from workgroups that hold BOTH rows of a pair: the row-chunk form with two or four
        // rows per workgroup (TX 128 or 64), whatever the width
        if ((a.H & 1u) || ((uintptr_t)a.dist & 7)) return hipErrorInvalidValue;
        // Rows of two or more workgroups (W a multiple of 256, >= 512): ONE row per workgroup like the plain kernel, the rows of a
        // pair on the same XCD one dispatch slot apart -- the halves of every volume line merge in that XCD's L2 (R5.3: 512^3
        // 0.763 ms against 0.803 for the thread-per-pair form of R4.10 and 0.745 for the fused fill with the plain volume;
        // 1024^3 6.25 / 6.62 / 6.05).  Without the eight-XCD placement: a thread per x of both rows of the pair (R4.10).  One
        // workgroup per row (W = 256) keeps the LDS form (paired: +8 %, thread-per-pair: +14 %).  force_*: A/B runs, tests.
        if (a.W % kBlock == 0 && (cfg.force_paired || (cfg.xcd_pairing && a.W >= 2 * kBlock && !cfg.force_rows && !cfg.force_pairrows)))
            return cfg.nontemporal ? launch_dense_ilv_paired<true>(a, stream) : launch_dense_ilv_paired<false>(a, stream);
        if (a.W % kBlock == 0 && (cfg.force_pairrows || (a.W >= 2 * kBlock && !cfg.force_rows)))
            return cfg.nontemporal ? launch_dense_pairrows<true>(a, stream) : launch_dense_pairrows<false>(a, stream);
        return a.W <= 64 ? launch_dense_tx<64>(a, cfg, stream) : launch_dense_tx<128>(a, cfg, stream);
    }
    // Row-chunk form when the width fills its lanes, flat form otherwise (and always when forced for A/B runs).
    const uint32_t chunk = a.W <= 64 ? 64 : (a.W <= 128 ? 128 : 256);
    const bool lanes_full = a.W % chunk == 0;
    const uint64_t n_vox = (uint64_t)a.W * a.H * a.slab_d;
    if (n_vox < (1ull << 32) && (cfg.force_flat || (!lanes_full && !cfg.force_rows)))
        return cfg.nontemporal ? launch_dense_flat<true>(a, stream) : launch_dense_flat<false>(a, stream);
    if (a.W <= 64) return launch_dense_tx<64>(a, cfg, stream);
    if (a.W <= 128) return launch_dense_tx<128>(a, cfg, stream);
    return launch_dense_tx<256>(a, cfg, stream);
}

static DivU32 make_div(uint32_t d) {
    DivU32 v;
    v.pow2 = d != 0 && (d & (d - 1)) == 0;
    v.shift = 0;
    while (v.pow2 && (1u << v.shift) < d) ++v.shift;
    v.magic = d > 1 ? (~0ull / d) + 1ull : 0ull;  // d == 1 is a power of two: the multiply is never taken
    return v;
}

hipError_t launch_fill_pass(const FillArgs& a, const PassArgs& pass, const FillLaunch& dense_cfg, hipStream_t stream) {
    PassArgs p = pass;
    if (p.virgin) p.fresh = p.all_required = 1;  // nothing is read: every voxel no pass of this load has written is LOGICALLY AIR
    for (int i = 0; i < 3; ++i) {
        // |idx * (size / dm1) + min  -  ((idx / dm1) * size + min)| is a few ulp of the largest intermediate
        // (<= |size| + |min|); 64 ulp of that is a safe bound, and tiny next to a voxel's extent.
        p.approx_scale[i] = a.dm1[i] > 0.0f ? a.bb_size[i] / a.dm1[i] : 0.0f;
        p.approx_margin[i] = 64.0f * 1.1920929e-7f * (fabsf(a.bb_size[i]) + fabsf(a.bb_min[i]));
        if (!(a.dm1[i] > 0.0f) || !(p.approx_margin[i] >= 0.0f)) p.approx_margin[i] = INFINITY;  // never filter
    }
    const uint64_t n = (uint64_t)p.nx * p.ny * p.nz;
    if (n == 0) return hipSuccess;
    if (p.all_required && p.step == 1 && (p.dist || p.fresh)) {
        // every voxel of the slab is rewritten and nothing needs reading: that IS the dense fill (+ its distance volume).
        // (The dense kernel writes tex1.a = AIR_DIST: what a fresh grid holds and what the volume's contract guarantees;
        // without either the general kernel below carries the stored alpha through, as update() leaves it alone.)
        FillArgs d = a;
        d.dist = p.dist;
        return launch_fill_dense(d, dense_cfg, stream);
    }
    // The kernels below index the slab's voxels, the visited lattice and the visited rows with 32 bits.  A slab beyond that
    // (>= ~1626^3 voxels: fits in 288 GB) is passed over in pieces of whole slices, each below the limit.
    const uint64_t limit = p.index_limit ? p.index_limit : (1ull << 32);
    const uint64_t slice = (uint64_t)a.W * a.H;
    if (slice * a.slab_d >= limit) {
        const uint64_t per_piece = (limit - 1) / slice;  // slices per piece
        if (per_piece == 0) return hipErrorInvalidValue;  // one slice alone exceeds 32-bit indexing
        for (uint64_t z0 = 0; z0 < a.slab_d; z0 += per_piece) {
            FillArgs a2 = a;
            PassArgs p2 = pass;
            a2.z_begin = a.z_begin + (uint32_t)z0;
            a2.slab_d = (uint32_t)(a.slab_d - z0 < per_piece ? a.slab_d - z0 : per_piece);
            a2.tex0 = a.tex0 + z0 * slice;
            a2.tex1 = a.tex1 + z0 * slice;
            if (a.dist) a2.dist = a.dist + z0 * slice;
            if (pass.dist) p2.dist = pass.dist + z0 * slice;
            const uint32_t z_end = a2.z_begin + a2.slab_d;
            p2.z_first = ((a2.z_begin + p.step - 1) / p.step) * p.step;
            p2.nz = p2.z_first < z_end ? (z_end - p2.z_first + p.step - 1) / p.step : 0;
            if (hipError_t e = launch_fill_pass(a2, p2, dense_cfg, stream)) return e;
        }
        return hipSuccess;
    }
    p.n_visited = (uint32_t)n;
    p.div_nx = make_div(p.nx);
    p.div_ny = make_div(p.ny);
    p.div_w = make_div(a.W);
    if (p.all_required && (p.step <= 8 || p.virgin)) {  // (a virgin grid has nothing to read back: whole rows at any step)
        // whole visited rows: one of every `step` texels computed, the rest re-stored (a constant over a fresh grid)
        const uint32_t blocks = (uint32_t)(((uint64_t)a.W * p.ny * p.nz + kBlock - 1) / kBlock);
        if (p.fresh) {
            SDFV_LAUNCH_CFG(a, (fill_pass_rows_kernel<Cfg, true>), dim3(blocks), dim3(kBlock), 0, stream, a, p);
        } else {
            SDFV_LAUNCH_CFG(a, (fill_pass_rows_kernel<Cfg, false>), dim3(blocks), dim3(kBlock), 0, stream, a, p);
        }
        return hipGetLastError();
    }
    const bool vol16 = p.dist && a.W % 4 == 0 && ((uintptr_t)p.dist & 15) == 0;  // (either layout: 16-byte loads of the volume)
    if (!p.all_required && !p.has_box && p.step >= 2 && p.step <= 8 && vol16 && !p.no_adaptive && (uint64_t)a.W * p.ny * p.nz < (1ull << 32)) {
        // nothing known, nothing boxed: whole visited rows, every wave deciding on what it reads (rows_adaptive above).  Up to
        // step 8, like the all_required rows form: the rows hold W * ny * nz voxels against the lattice's nx * ny * nz, so at
        // larger steps a no-op pass would scan step / 4 times the lanes of the per-voxel kernel below
        const uint64_t lanes = ((uint64_t)a.W * p.ny * p.nz + (a.dist_ilv ? 1 : 3)) / (a.dist_ilv ? 2 : 4);
        const uint32_t blocks = (uint32_t)((lanes + kBlock - 1) / kBlock);
        if (a.dist_ilv) {
            SDFV_LAUNCH_CFG(a, (fill_pass_rows_adaptive_kernel<Cfg, 2>), dim3(blocks), dim3(kBlock), 0, stream, a, p);
        } else {
            SDFV_LAUNCH_CFG(a, (fill_pass_rows_adaptive_kernel<Cfg, 4>), dim3(blocks), dim3(kBlock), 0, stream, a, p);
        }
        return hipGetLastError();
    }
    const bool quad = p.step == 1 && vol16;
    const uint64_t threads = quad ? (n + 3) / 4 : n;
    const uint32_t blocks = (uint32_t)((threads + kBlock - 1) / kBlock);
    if (quad) {
        SDFV_LAUNCH_CFG(a, (fill_pass_quad_kernel<Cfg>), dim3(blocks), dim3(kBlock), 0, stream, a, p);
    } else {
        SDFV_LAUNCH_CFG(a, (fill_pass_kernel<Cfg>), dim3(blocks), dim3(kBlock), 0, stream, a, p);
    }
    return hipGetLastError();
}

hipError_t launch_commit_distance(const float* tex0, float* dist, uint64_t n_voxels, hipStream_t stream) {
    if (n_voxels == 0) return hipSuccess;
    uint64_t blocks = (n_voxels + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(commit_distance_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream,
                       reinterpret_cast<const float4*>(tex0), dist, n_voxels);
    return hipGetLastError();
}

// (x chunks, rows, slices): workgroups of min(256, W rounded up to a wave) threads along x
static bool commit_grid(uint32_t W, uint32_t rows, uint32_t D, dim3& grid, dim3& block) {
    if (rows > 65535u || D > 65535u) return false;
    const uint32_t threads = W >= (uint32_t)kBlock ? (uint32_t)kBlock : ((W + 63u) / 64u) * 64u;
    block = dim3(threads);
    grid = dim3((W + threads - 1) / threads, rows, D);
    return true;
}

hipError_t launch_commit_pairs(const float* dist, float* pairs, uint32_t W, uint32_t H, uint64_t n_voxels, hipStream_t stream) {
    if (n_voxels == 0) return hipSuccess;
    dim3 grid, block;
    if (!commit_grid(W, H, (uint32_t)(n_voxels / ((uint64_t)W * H)), grid, block)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(commit_pairs_kernel, grid, block, 0, stream, dist, reinterpret_cast<float2*>(pairs), W, H);
    return hipGetLastError();
}

hipError_t launch_commit_interleaved(const float* dist, float* ilv, uint32_t W, uint64_t n_voxels, hipStream_t stream) {
    const uint64_t n_pairs = n_voxels / 2;
    if (n_pairs == 0) return hipSuccess;
    const uint64_t blocks = (n_pairs + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(commit_interleaved_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, dist, reinterpret_cast<float2*>(ilv),
                       W, n_pairs);
    return hipGetLastError();
}

hipError_t launch_grid_init_unvisited(float* tex0, float* tex1, float* dist, uint32_t W, uint32_t H, uint32_t z_begin,
                                      uint32_t slab_d, uint32_t step, float air, uint32_t dist_ilv, hipStream_t stream) {
    if ((uint64_t)W * H * slab_d == 0) return hipSuccess;
    dim3 grid, block;
    if (!commit_grid(W, H, slab_d, grid, block)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(grid_init_unvisited_kernel, grid, block, 0, stream, reinterpret_cast<float4*>(tex0),
                       reinterpret_cast<float4*>(tex1), dist, W, H, z_begin, step, air, dist_ilv);
    return hipGetLastError();
}

hipError_t launch_grid_init(float* tex0, float* tex1, uint64_t n_voxels, float air, hipStream_t stream) {
    if (n_voxels == 0) return hipSuccess;
    const uint64_t blocks = (n_voxels + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(grid_init_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream,
                       reinterpret_cast<float4*>(tex0), reinterpret_cast<float4*>(tex1), n_voxels, air);
    return hipGetLastError();
}

}  // namespace sdfv
