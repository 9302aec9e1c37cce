// semantics check of __builtin_amdgcn_permlane32_swap on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* p, float* a, float* b) {
    float v = p[threadIdx.x];
    unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    // the builtin's second result is lowered wrongly by this hipcc (both results read vdst): use the instruction itself
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(u2));
    unsigned r[2] = {u, u2};
    a[threadIdx.x] = __builtin_bit_cast(float, r[0]);
    b[threadIdx.x] = __builtin_bit_cast(float, r[1]);
}
int main() {
    float h[64], ha[64], hb[64], *d, *da, *db;
    for (int i = 0; i < 64; i++) h[i] = i;
    hipMalloc(&d, 256); hipMalloc(&da, 256); hipMalloc(&db, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, da, db);
    hipMemcpy(ha, da, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, db, 256, hipMemcpyDeviceToHost);
    for (int i : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 = %2.0f  r1 = %2.0f\n", i, ha[i], hb[i]);
    return 0;
}
