// fill_kernels.hip -- grid fill on gfx950: SDFViewer::update's per-voxel loop
// (reference src/app/scene/sdf/mod.rs:173-215) as store-bound HIP kernels.
//
// Dense kernel (fill_dense_kernel): the final state of a fresh grid.  One thread per voxel, x fastest, so
// a 64-lane wave emits two contiguous 1 KiB bursts (tex0, tex1) of global_store_dwordx4.  No global
// loads at all: the position derives from the index.  Per-workgroup LDS staging holds what is O(1)
// or O(N) and shared by every voxel the workgroup touches:
//   - the demo SDF's parameter block (the "CSG-tree params"),
//   - the 256-entry sRGB->linear table (colour passes through u8, scene/sdf/mod.rs:201),
//   - the y and z coordinate tables (idx/(dim-1)*size+min costs an IEEE divide per axis; x is fixed
//     per thread and hoisted, y/z are staged once per workgroup instead of per voxel).
// Workgroups are persistent: a (x-chunk, row-phase) pair walks rows with a fixed stride, so the
// staging cost is amortised over thousands of voxels per thread.
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

// TX = lanes along x per row segment (64, 128 or 256); TY = 256 / TX rows per workgroup step.
template <int TX, bool NT>
__global__ __launch_bounds__(kBlock) void fill_dense_kernel(FillArgs a) {
    constexpr int TY = kBlock / TX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_lut = reinterpret_cast<float*>(smem);                 // 256
    float* s_y = s_lut + 256;                                       // H
    float* s_z = s_y + a.H;                                         // slab depth
    sdfv_demo_params* s_prm = reinterpret_cast<sdfv_demo_params*>(s_z + a.slab_d);

    const uint32_t tid = threadIdx.x;
    s_lut[tid] = c_srgb_lut[tid];
    for (uint32_t i = tid; i < a.H; i += kBlock) s_y[i] = voxel_coord(i, a.dm1[1], a.bb_size[1], a.bb_min[1]);
    for (uint32_t i = tid; i < a.slab_d; i += kBlock)
        s_z[i] = voxel_coord(a.z_begin + i, a.dm1[2], a.bb_size[2], a.bb_min[2]);
    if (tid == 0) *s_prm = a.prm;
    __syncthreads();

    const sdfv_demo_params prm = *s_prm;
    const LdsLut lut{s_lut};

    const uint32_t tx = tid % TX, ty = tid / TX;
    const uint32_t x = blockIdx.x * TX + tx;
    if (x >= a.W) return;
    const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);

    // rows of the slab: row = z_local * H + y.  This thread walks row0, row0 + stride, ...
    const uint64_t n_rows = (uint64_t)a.H * a.slab_d;
    uint64_t row = (uint64_t)blockIdx.y * TY + ty;
    if (row >= n_rows) return;
    uint32_t y = (uint32_t)(row % a.H), zl = (uint32_t)(row / a.H);
    const uint32_t dy = a.row_stride_y, dz = a.row_stride_z;  // (gridDim.y * TY) % H, / H

    float4* t0 = a.tex0 + row * a.W + x;
    float4* t1 = a.tex1 + row * a.W + x;
    const uint64_t step_elems = (uint64_t)gridDim.y * TY * a.W;

    while (zl < a.slab_d) {
        const float py = s_y[y], pz = s_z[zl];
        Sample s = demo_sample(prm, a.sdf_id, px, py, pz, false);
        float4 v0, v1;
        pack_sample(s, lut, a.air_dist, v0, v1);
        store_texel<NT>(t0, v0);
        store_texel<NT>(t1, v1);
        t0 += step_elems;
        t1 += step_elems;
        y += dy;
        zl += dz;
        if (y >= a.H) { y -= a.H; zl += 1; }
    }
}

struct GlobalLut {
    __device__ __forceinline__ float operator[](uint32_t i) const { return c_srgb_lut[i]; }
};

// One LoadingManager pass: thread per visited voxel (x, y, z multiples of `step`; z is GLOBAL).
__global__ __launch_bounds__(kBlock) void fill_pass_kernel(FillArgs a, PassArgs p) {
    const uint64_t n = (uint64_t)p.nx * p.ny * p.nz;
    const GlobalLut lut{};
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t ix = (uint32_t)(i % p.nx);
        const uint64_t r = i / p.nx;
        const uint32_t iy = (uint32_t)(r % p.ny), iz = (uint32_t)(r / p.ny);
        const uint32_t x = ix * p.step, y = iy * p.step, z = p.z_first + iz * p.step;  // global z
        const float px = voxel_coord(x, a.dm1[0], a.bb_size[0], a.bb_min[0]);
        const float py = voxel_coord(y, a.dm1[1], a.bb_size[1], a.bb_min[1]);
        const float pz = voxel_coord(z, a.dm1[2], a.bb_size[2], a.bb_min[2]);
        const uint64_t flat = ((uint64_t)(z - a.z_begin) * a.H + y) * a.W + x;
        // update_required, scene/sdf/mod.rs:184-190
        bool update_required = a.tex0[flat].x == a.air_dist;
        if (p.has_box) {
            update_required = update_required ||
                              (px >= p.box[0] && px <= p.box[3] && py >= p.box[1] && py <= p.box[4] &&
                               pz >= p.box[2] && pz <= p.box[5]);
        }
        if (!update_required) continue;
        Sample s = demo_sample(a.prm, a.sdf_id, px, py, pz, false);
        float4 v0, v1;
        pack_sample(s, lut, a.air_dist, v0, v1);
        a.tex0[flat] = v0;
        // tex1.a is not written by update(): store 12 bytes only
        float* t1 = reinterpret_cast<float*>(a.tex1 + flat);
        t1[0] = v1.x;
        t1[1] = v1.y;
        t1[2] = v1.z;
    }
}

__global__ __launch_bounds__(kBlock) void grid_init_kernel(float4* tex0, float4* tex1, uint64_t n, float air) {
    const float4 v = make_float4(air, air, air, air);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        tex0[i] = v;
        tex1[i] = v;
    }
}

template <int TX>
hipError_t launch_dense_tx(const FillArgs& args, const FillLaunch& cfg, hipStream_t stream) {
    FillArgs a = args;
    constexpr int TY = kBlock / TX;
    const uint32_t cx = (a.W + TX - 1) / TX;
    const uint64_t n_rows = (uint64_t)a.H * a.slab_d;
    uint64_t row_blocks = (n_rows + TY - 1) / TY;
    uint64_t want = cfg.target_blocks / cx;
    if (want < 1) want = 1;
    if (row_blocks > want) row_blocks = want;
    if (row_blocks > 65535) row_blocks = 65535;
    const uint64_t stride_rows = row_blocks * TY;
    a.row_stride_y = (uint32_t)(stride_rows % a.H);
    a.row_stride_z = (uint32_t)(stride_rows / a.H);
    const size_t lds = (256 + (size_t)a.H + a.slab_d) * sizeof(float) + sizeof(sdfv_demo_params);
    dim3 grid(cx, (uint32_t)row_blocks, 1);
    if (cfg.nontemporal)
        hipLaunchKernelGGL((fill_dense_kernel<TX, true>), grid, dim3(kBlock), lds, stream, a);
    else
        hipLaunchKernelGGL((fill_dense_kernel<TX, false>), grid, dim3(kBlock), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

size_t fill_dense_lds_bytes(const FillArgs& a) {
    return (256 + (size_t)a.H + a.slab_d) * sizeof(float) + sizeof(sdfv_demo_params);
}

hipError_t launch_fill_dense(const FillArgs& a, const FillLaunch& cfg, hipStream_t stream) {
    if (a.W == 0 || a.H == 0 || a.slab_d == 0) return hipSuccess;
    if (a.W <= 64) return launch_dense_tx<64>(a, cfg, stream);
    if (a.W <= 128) return launch_dense_tx<128>(a, cfg, stream);
    return launch_dense_tx<256>(a, cfg, stream);
}

hipError_t launch_fill_pass(const FillArgs& a, const PassArgs& p, hipStream_t stream) {
    const uint64_t n = (uint64_t)p.nx * p.ny * p.nz;
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fill_pass_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, a, p);
    return hipGetLastError();
}

hipError_t launch_grid_init(float* tex0, float* tex1, uint64_t n_voxels, float air, hipStream_t stream) {
    if (n_voxels == 0) return hipSuccess;
    uint64_t blocks = (n_voxels + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(grid_init_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream,
                       reinterpret_cast<float4*>(tex0), reinterpret_cast<float4*>(tex1), n_voxels, air);
    return hipGetLastError();
}

}  // namespace sdfv
