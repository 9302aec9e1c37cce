// Micro-benchmark: how fast can a CU pull L2-resident data (a) straight into LDS with buffer_load ... lds (LDS-DMA, 1 KiB per
// wave-instruction) and (b) into registers with buffer_load_dwordx4 (+ optional ds_write_b128 into LDS)?  One 512-thread
// workgroup per CU, each streaming its own 64 KiB window of a buffer over and over (L2 hits), 8 instructions in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill tools/ubench/fill_rate.hip && timeout 60 /tmp/fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ v4i32 make_srd(const void* base, int bytes) {
    const unsigned long long a = (unsigned long long)base;
    v4i32 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xFFFFu));
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000;
    return d;
}

template <int MODE>   // 0: LDS-DMA, 1: registers only, 2: registers + ds_write_b128
__global__ __launch_bounds__(512) void k(const char* buf, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* win = buf + (size_t)blockIdx.x * 65536;
    const v4i32 srd = make_srd(win, 65536);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_char*)smem;
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        // 64 pieces of 1 KiB per pass over the window; 8 per wave
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int piece = wave * 8 + p;
                const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + piece * 1024));
                const int voff = piece * 1024 + lane * 16;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(srd) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 r[8];
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int piece = wave * 8 + p;
                r[p] = *reinterpret_cast<const f32x4*>(win + piece * 1024 + lane * 16 + (it & 1) * 0);
                asm volatile("" : "+v"(r[p]));
            }
#pragma unroll
            for (int p = 0; p < 8; p++) {
                if (MODE == 2) *reinterpret_cast<f32x4*>(smem + (wave * 8 + p) * 1024 + lane * 16) = r[p];
                else acc += r[p];
            }
            if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 0 || MODE == 2) acc[0] += *reinterpret_cast<float*>(smem + tid * 4);
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] == 12345.f) sink[0] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    char* buf; unsigned long long* cyc; float* sink;
    hipMalloc(&buf, 256 * 65536); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 16);
    hipMemset(buf, 1, 256 * 65536);
    const int iters = 2000;
    for (int mode = 0; mode < 3; mode++)
        for (int grid : {32, 256}) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); k<0><<<grid, 512, 65536>>>(buf, iters, cyc, sink); }
                if (mode == 1) { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); k<1><<<grid, 512, 65536>>>(buf, iters, cyc, sink); }
                if (mode == 2) { hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); k<2><<<grid, 512, 65536>>>(buf, iters, cyc, sink); }
                hipDeviceSynchronize();
            }
            unsigned long long h[256];
            hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < grid; i++) avg += h[i]; avg /= grid;
            printf("%-28s %3d CUs: %6.1f B/clk/CU\n", mode == 0 ? "LDS-DMA (buffer_load lds)" : mode == 1 ? "registers (buffer_load x4)" : "registers + ds_write_b128", grid,
                   65536.0 * iters / avg);
        }
    return 0;
}
