// Micro-benchmark: what ONE dependent kernel of a batch-1 decode chain costs on this box, replayed from a hipGraph
// (the structure of decode.hip: 5L+2 short kernels per token, each consuming the vector the previous one wrote).
//   a: empty kernel, 1 workgroup            -> the bare dependent-launch boundary
//   b: empty kernel, 512 workgroups
//   c: every workgroup reads the 512-float vector of the previous kernel, block-reduces it, writes its 4 outputs
//   d: c + each wave streams one 2 KiB weight row (fp32 [N][K], N = 1536 -> 3 MB per kernel) and dots it with the vector
//   e: d with K = 2048 rows (N = 512: 4 MB, 8 loads in flight per lane)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/graph_chain tools/ubench/graph_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1.f; }

__device__ __forceinline__ float wsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int KMAX>
__global__ __launch_bounds__(256) void k_vec(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ W, int K, int N) {
    __shared__ float xs[4096 + 8];
    float* red = xs + 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x * 4 + wave;
    float4 wv[KMAX];
    if (W) {
        const float* wr = W + (size_t)(n < N ? n : N - 1) * K;
#pragma unroll
        for (int i = 0; i < KMAX; i++) {
            const int k = (lane + 64 * i) * 4;
            wv[i] = k < K ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = tid; k < K; k += 256) { float v = x[k]; xs[k] = v; s += v; }
    s = wsum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / K;
    float acc = 0.f;
    if (W) {
#pragma unroll
        for (int i = 0; i < KMAX; i++) {
            const int k = (lane + 64 * i) * 4;
            if (k < K) {
                float4 xv = *reinterpret_cast<const float4*>(xs + k);
                acc += xv.x * wv[i].x + xv.y * wv[i].y + xv.z * wv[i].z + xv.w * wv[i].w;
            }
        }
        acc = wsum(acc);
    }
    if (lane == 0 && n < N) y[n] = mean * 1e-3f + acc * 1e-3f + 1.0f;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *v0, *v1, *W;
    CK(hipMalloc(&v0, 4096 * 4)); CK(hipMalloc(&v1, 4096 * 4)); CK(hipMalloc(&W, (size_t)6 * 4 * 1024 * 1024));
    CK(hipMemset(v0, 0, 4096 * 4)); CK(hipMemset(v1, 0, 4096 * 4)); CK(hipMemset(W, 0, (size_t)6 * 4 * 1024 * 1024));
    const int NK = 32, REPLAY = 400;
    for (int variant = 0; variant < 6; variant++) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NK; i++) {
            float* in = (i & 1) ? v1 : v0; float* out = (i & 1) ? v0 : v1;
            const float* w = W + (size_t)(i % 6) * 1024 * 1024;       // a different 4 MB region per "layer" (24 MB set)
            switch (variant) {
                case 0: k_empty<<<1, 256, 0, s>>>(out); break;
                case 1: k_empty<<<512, 256, 0, s>>>(out); break;
                case 2: k_vec<2><<<128, 256, 0, s>>>(in, out, nullptr, 512, 512); break;
                case 3: k_vec<2><<<384, 256, 0, s>>>(in, out, w, 512, 1536); break;
                case 4: k_vec<8><<<128, 256, 0, s>>>(in, out, w, 2048, 512); break;
                case 5: k_vec<2><<<512, 256, 0, s>>>(in, out, w, 512, 2048); break;
            }
        }
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ex, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < REPLAY; i++) CK(hipGraphLaunch(ex, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* names[] = {"a empty 1 WG", "b empty 512 WG", "c vector pass 128 WG", "d vector + 3 MB weights (K=512, 384 WG)",
                               "e vector + 4 MB weights (K=2048, 128 WG)", "f vector + 4 MB weights (K=512, 512 WG)"};
        printf("%-44s %.2f us per kernel (%d-kernel graph, %.1f us per replay)\n", names[variant], ms * 1e3 / (REPLAY * NK), NK, ms * 1e3 / REPLAY);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    }
    return 0;
}
