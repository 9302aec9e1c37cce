// VALU / transcendental / MFMA issue rates per SIMD by waves per SIMD (gfx950): what bounds the D = 64 attention kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/bin/valu_issue && tools/ubench/bin/valu_issue
// One workgroup per CU of W waves (W/4 per SIMD), each wave runs N iterations of a fixed block of independent instructions;
// reported: cycles (s_memtime) per wave-instruction per SIMD = elapsed / (N * block * waves_per_simd).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int n, int mfma_waves) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    f32x16 acc, acc2;
    for (int i = 0; i < 16; i++) { acc[i] = 0.f; acc2[i] = 0.f; }
    bf16x8 fa, fb;
    for (int i = 0; i < 8; i++) { fa[i] = (__bf16)(0.01f * (float)(threadIdx.x & 7)); fb[i] = (__bf16)0.5f; }
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (MODE == 3) || (MODE == 4 && wave < mfma_waves);
    const bool do_valu = (MODE == 0 || MODE == 1 || MODE == 2) || (MODE == 4 && wave >= mfma_waves);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it++) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 8; j++) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc2, 0, 0, 0); }
        }
        if (do_valu) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (MODE == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
                    else if (MODE == 2) { float x = a[i], y = a[(i + 1) & 15]; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&a[(i + 2) & 14])); (void)x; (void)y; }
                    else a[i] = fmaf(a[i], 1.0001f, 0.5f);
                }
        }
        if (MODE == 5) {          // ONE wave interleaves: per MFMA, FILL independent fmas
#pragma unroll
            for (int j = 0; j < 16; j++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 6; i++) a[(j + i) & 15] = fmaf(a[(j + i) & 15], 1.0001f, 0.5f);
            }
        }
        if (MODE == 6 || MODE == 7 || MODE == 8) {   // the same work with the order PINNED: 1 MFMA, then its fillers (sched_group_barrier)
            constexpr int NF = MODE == 6 ? 5 : (MODE == 7 ? 3 : 8);      // plain fillers per gap; MODE 7 adds one v_exp_f32 per gap
#pragma unroll
            for (int j = 0; j < 16; j++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NF; i++) a[(j + i) & 15] = fmaf(a[(j + i) & 15], 1.0001f, 0.5f);
                if (MODE == 7) a[(j + 9) & 15] = __builtin_amdgcn_exp2f(a[(j + 9) & 15]);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NF, 0);
                if (MODE == 7) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += a[i] + acc[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // the workgroup's span: earliest start to latest end over its waves (the oldest wave wins every arbitration: timing wave 0
    // alone reads the single-wave rate whatever its partners do)
    __shared__ unsigned long long lo, hi;
    if (threadIdx.x == 0) { lo = ~0ull; hi = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&lo, t0); atomicMax(&hi, t1); }
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = hi - lo;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    const int n = 2000;
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "mfma 32x32x16 (2 chains)", "MFMA waves + fma waves", "1 MFMA + 6 fma per gap, every wave",
                           "pinned: 1 MFMA + 5 fma per gap", "pinned: 1 MFMA + 3 fma + 1 exp per gap", "pinned: 1 MFMA + 8 fma per gap"};
    for (int mode = 0; mode < 9; mode++)
        for (int waves = 4; waves <= 16; waves *= 2) {
            if (mode == 4 && waves < 8) continue;
            std::vector<unsigned long long> h(256);
            for (int rep = 0; rep < 2; rep++) {
                switch (mode) {
                    case 0: k<0><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    case 1: k<1><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    case 2: k<2><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    case 3: k<3><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    case 4: k<4><<<256, waves * 64>>>(out, cyc, n, waves / 2); break;
                    case 5: k<5><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    case 6: k<6><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    case 7: k<7><<<256, waves * 64>>>(out, cyc, n, 0); break;
                    default: k<8><<<256, waves * 64>>>(out, cyc, n, 0); break;
                }
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
            double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
            const int wps = waves / 4;
            if (mode <= 2) printf("%-36s %d wave(s)/SIMD: %6.2f cycles per wave-instruction per SIMD\n", names[mode], wps, avg / (n * 64.0 * wps));
            else if (mode == 3) printf("%-36s %d wave(s)/SIMD: %6.2f cycles per MFMA per SIMD\n", names[mode], wps, avg / (n * 16.0 * wps));
            else if (mode == 4) printf("%-36s %d wave(s)/SIMD (half MFMA, half fma): %8.0f cycles per iteration (16 MFMA | 64 fma per wave)\n", names[mode], wps, avg / n);
            else printf("%-40s %d wave(s)/SIMD: %6.2f cycles per (MFMA + fillers) per SIMD\n", names[mode], wps, avg / (n * 16.0 * wps));
        }
    return 0;
}
