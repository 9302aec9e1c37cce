// Micro-benchmark: how fast can ONE workgroup (8 waves) push a 256x256 bf16 tile (128 KiB) to global memory, as a function
// of how many CUs do it at once and of the store shape.  Build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/ubench/store_rate.hip && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: each lane stores 16 B, 8 lanes cover one 128-B row segment, rows ldc apart (the GEMM epilogue shape)
// mode 1: fully linear: wave-instruction writes 1 KiB contiguous
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(char* out, int ldc_bytes, int items, unsigned long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 v = {1.f * tid, 2.f, 3.f, 4.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < items; it++) {
        const int item = blockIdx.x + it * gridDim.x;
        char* base = out + (size_t)item * 256 * ldc_bytes;          // tile rows [item*256, +256), 512 B of columns
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int row = wm * 128 + i * 16 + h * 8 + (lane >> 3);
                    *reinterpret_cast<f32x4*>(base + (size_t)row * ldc_bytes + wn * 128 + (lane & 7) * 16) = v;
                }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) *reinterpret_cast<f32x4*>(base + (size_t)(wave * 16 + i) * 1024 + lane * 16) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int items = 12;
    const size_t bytes = (size_t)256 * items * 256 * 3072;     // room for ldc = 3072 B rows
    char* out; unsigned long long* cyc;
    hipMalloc(&out, bytes); hipMalloc(&cyc, 256 * 8);
    hipMemset(out, 0, bytes);
    for (int mode = 0; mode < 2; mode++)
        for (int ldc : {512, 3072})
            for (int grid : {8, 32, 64, 128, 256}) {
                if (mode == 1 && ldc != 512) continue;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) store_kernel<0><<<grid, 512>>>(out, ldc, items, cyc);
                    else store_kernel<1><<<grid, 512>>>(out, 1024, items, cyc);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> h(grid);
                hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                const double per_item = (double)h[grid / 2] / items;
                printf("mode %d ldc %4d B  grid %3d: %7.0f cycles per 128-KiB tile (%.1f B/cyc/CU), kernel %.1f us, aggregate %.2f TB/s\n", mode, ldc, grid,
                       per_item, 131072.0 / per_item, ms * 1e3, (double)grid * items * 131072 / (ms * 1e-3) / 1e12);
            }
    return 0;
}
