// Single-wave latency microbenchmarks for gfx950 (tuning aid for the raymarch loop): cycles per instruction of
// dependent chains, measured with s_memtime around N repetitions.  hipcc --offload-arch=gfx950 -O2 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

__global__ void k_valu_add(float* out, unsigned long long* t, float a) {
    float v = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(a));)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_valu_indep(float* out, unsigned long long* t, float a) {
    float v0 = a, v1 = a, v2 = a, v3 = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v0 + v1 + v2 + v3; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_pk_mul(float* out, unsigned long long* t, float a) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f v = {a, a}, b = {a, a};
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(b));)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v.x + v.y; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_cmp_cndmask(float* out, unsigned long long* t, float a) {
    float v = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(a) : "vcc");)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_cmp_salu_cndmask(float* out, unsigned long long* t, float a) {
    float v = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v) : "v"(a) : "vcc", "s20", "s21");)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_salu(float* out, unsigned long long* t, int a) {
    int s = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("s_add_i32 %0, %0, %1" : "+s"(s) : "s"(a));)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)s; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_branch_loop(float* out, unsigned long long* t, float a, int n) {
    float v = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {  // v_add + (compiler's loop counter + s_cbranch)
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(a));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_ballot_loop(float* out, unsigned long long* t, float a, int n) {
    float v = a;
    bool go = true;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {  // v op -> v_cmp -> ballot -> scalar branch, like the march loop's exit test
        if (__ballot(go) == 0ull) break;
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(a));
        go = v < 1e30f;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_floor_cvt(float* out, unsigned long long* t, float a) {
    float v = a;
    unsigned long long t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_floor_f32 %0, %0\n v_sub_f32 %0, %0, %1" : "+v"(v) : "v"(a));)
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void k_load_chain(const unsigned* __restrict__ idx, float* out, unsigned long long* t, int n) {
    unsigned j = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) j = idx[j];  // dependent L1/L2-hit loads
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)j; if (threadIdx.x == 0) t[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* t; unsigned* idx;
    hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMalloc(&idx, 4096 * 4);
    std::vector<unsigned> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i * 67 + 13) % 4096;
    hipMemcpy(idx, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    auto report = [&](const char* name, double per) {
        unsigned long long c; hipDeviceSynchronize(); hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost);
        printf("%-34s %8llu cycles  %.2f per %s\n", name, c, c / per, "unit");
    };
    for (int lanes : {64, 1}) {
        printf("== %d active lane(s)\n", lanes);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_valu_add, 1, lanes, 0, 0, out, t, 1.0f); if (rep) report("dependent v_add_f32 x256", 256);
            hipLaunchKernelGGL(k_valu_indep, 1, lanes, 0, 0, out, t, 1.0f); if (rep) report("independent v_add_f32 x1024", 1024);
            hipLaunchKernelGGL(k_pk_mul, 1, lanes, 0, 0, out, t, 1.0f); if (rep) report("dependent v_pk_mul_f32 x256", 256);
            hipLaunchKernelGGL(k_floor_cvt, 1, lanes, 0, 0, out, t, 1.5f); if (rep) report("dependent floor+sub pairs x256", 512);
            hipLaunchKernelGGL(k_cmp_cndmask, 1, lanes, 0, 0, out, t, 1.0f); if (rep) report("v_cmp->vcc->v_cndmask x256", 512);
            hipLaunchKernelGGL(k_cmp_salu_cndmask, 1, lanes, 0, 0, out, t, 1.0f); if (rep) report("v_cmp->s_and->v_cndmask x256", 768);
            hipLaunchKernelGGL(k_salu, 1, lanes, 0, 0, out, t, 1); if (rep) report("dependent s_add_i32 x256", 256);
            hipLaunchKernelGGL(k_branch_loop, 1, lanes, 0, 0, out, t, 1.0f, 1000); if (rep) report("loop{v_add; counter; branch} x1000", 1000);
            hipLaunchKernelGGL(k_ballot_loop, 1, lanes, 0, 0, out, t, 1e-3f, 1000); if (rep) report("loop{ballot-exit; v_add; v_cmp} x1000", 1000);
            hipLaunchKernelGGL(k_load_chain, 1, lanes, 0, 0, idx, out, t, 1000); if (rep) report("dependent global_load (L1 hit) x1000", 1000);
        }
    }
    return 0;
}
