// v_pk_mov_b32 operand selection on gfx950, printed: which halves of the two 64-bit sources land in the result for each
// op_sel / op_sel_hi.  hipcc --offload-arch=gfx950 -O2 tools/ubench/pkmov.hip -o tools/ubench/pkmov && tools/ubench/pkmov
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a0 = 0xA0, a1 = 0xA1, b0 = 0xB0, b1 = 0xB1;
    unsigned long long a = ((unsigned long long)a1 << 32) | a0, b = ((unsigned long long)b1 << 32) | b0, d;
#define T(i, sel) asm volatile("v_pk_mov_b32 %0, %1, %2 " sel : "=v"(d) : "v"(a), "v"(b)); out[2*i] = (unsigned)d; out[2*i+1] = (unsigned)(d >> 32);
    T(0, "op_sel:[0,0]") T(1, "op_sel:[1,1]") T(2, "op_sel:[0,1]") T(3, "op_sel:[1,0]")
    T(4, "op_sel:[0,0] op_sel_hi:[0,0]") T(5, "op_sel:[1,1] op_sel_hi:[1,1]") T(6, "op_sel:[0,0] op_sel_hi:[1,1]") T(7, "op_sel:[1,1] op_sel_hi:[0,0]")
    T(8, "op_sel:[0,1] op_sel_hi:[0,1]") T(9, "op_sel:[1,0] op_sel_hi:[1,0]")
}
int main() {
    unsigned* d; hipMalloc(&d, 80); k<<<1, 1>>>(d); unsigned h[20]; hipMemcpy(h, d, 80, hipMemcpyDeviceToHost);
    const char* n[] = {"op_sel:[0,0]", "op_sel:[1,1]", "op_sel:[0,1]", "op_sel:[1,0]", "[0,0] hi[0,0]", "[1,1] hi[1,1]", "[0,0] hi[1,1]", "[1,1] hi[0,0]", "[0,1] hi[0,1]", "[1,0] hi[1,0]"};
    for (int i = 0; i < 10; ++i) printf("%-16s lo=%02X hi=%02X\n", n[i], h[2*i], h[2*i+1]);
}
