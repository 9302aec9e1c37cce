// decode.hip -- autoregressive decode: the loop of `composer generate` (cli.py:659-676) on one GPU.
//
// Two modes behind one per-token step:
//   CMP_DECODE_LITERAL : cli.py as written -- `past` is never passed back, so every step after the first
//                        feeds ONE token at position 0 with no context.
//   CMP_DECODE_KV      : model(x, past=presents) (transformer.py:735-765, 423-426): one new token per step at
//                        position P+i attending to a preallocated KV cache (the reference's tf.concat
//                        re-allocates the cache every step).
// The prompt goes through the batched forward of model.hip (prefill); each later token is a fixed chain of
// 5L+2 small kernels whose only varying inputs (position, current token, RNG counter, output slot) live in
// device memory, so the chain is captured ONCE into a hipGraph and replayed per token.
//
// Arithmetic is fp32 in both model dtypes (weights fp32 master, transposed once per decode_begin to
// [N][K] so that a wave reads one output column as a contiguous, 16-B-per-lane stream).
// Sampling: temperature <= 0 -> argmax, lowest index on ties (tf.argmax); otherwise Gumbel-max over
// logits/temperature == a draw from softmax(logits/temperature) (tf.random.categorical, cli.py:671-673).
#include "model.h"

struct DecState {        // device-resident loop state
    int pos;             // position id of the token about to be consumed
    int token;           // that token
    int produced;        // number of ids written to ids[]
    int advance;         // 1: kv mode (pos += 1 per step), 0: literal (pos stays 0)
    unsigned rng;        // sampling counter
    int cap;             // capacity of ids[]
    int W;               // wpe rows
};

struct DecLayerW {
    float *attn_wT, *proj_wT, *fc_wT, *pr_wT;
    float *kc, *vc;      // [H][W][D]
};

struct DecodeState {
    DecState* st = nullptr;
    int32_t* ids = nullptr;
    float *x = nullptr, *u = nullptr, *qkv = nullptr, *att = nullptr, *r = nullptr, *g = nullptr, *logits = nullptr;
    std::vector<DecLayerW> lw;
    std::vector<void*> allocs;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int mode = 0;
    bool v1 = false;            // COMPOSER_DECODE_V1=1: first-generation per-token kernels
    float temperature = 0.f;
    uint64_t seed = 0;
    int produced = 0, returned = 0, cap = 0, pos = 0;
    bool begun = false;
};

void decode_state_free(DecodeState* d) {
    if (!d) return;
    if (d->exec) hipGraphExecDestroy(d->exec);
    if (d->graph) hipGraphDestroy(d->graph);
    for (void* p : d->allocs) hipFree(p);
    delete d;
}

template <typename Tp> static int dalloc(DecodeState* d, Tp** p, size_t bytes) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, bytes ? bytes : 16));
    d->allocs.push_back(q);
    *p = (Tp*)q;
    return CMP_OK;
}

// out[n][k] = in[k][n]
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int K, int N) {
    __shared__ float tile[32][33];
    int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int k = k0 + j, n = n0 + tx;
        tile[j][tx] = (k < K && n < N) ? in[(int64_t)k * N + n] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int n = n0 + j, k = k0 + tx;
        if (n < N && k < K) out[(int64_t)n * K + k] = tile[tx][j];
    }
}

// K/V of the prompt from the prefill's c_attn output [P][3E] (activation dtype) into the cache [H][W][D]
template <typename T>
__global__ void cache_fill_kernel(const T* __restrict__ qkv, float* __restrict__ kc, float* __restrict__ vc, int P, int E,
                                  int H, int D, int W) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * E) return;
    int t = i / E, e = i % E, h = e / D, d = e % D;
    kc[((int64_t)h * W + t) * D + d] = to_f32<T>(qkv[(int64_t)t * 3 * E + E + e]);
    vc[((int64_t)h * W + t) * D + d] = to_f32<T>(qkv[(int64_t)t * 3 * E + 2 * E + e]);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += red[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s = fmaxf(s, red[w]);
    return s;
}

// y[n] = act( IN(x) . Wt[n,:] + bias[n] ) + resid[n].  One WAVE per output column (4 columns per workgroup): the
// column's K weights are one contiguous row of the transposed matrix, read 16 bytes per lane with every load of the row
// in flight at once (K <= 64*4*GV_MAXI).  IN: 0 plain copy, 1 LayerNorm (every workgroup recomputes the row statistics of
// the 2-8 KiB input: cheaper than another launch), 2 combine of the split-key attention partials.
#define GV_MAXI 12
#define ATT_SPLITS 8        // 4: 206 us/token, 8: 194, 16: 222 (the combine in the next GEMV grows), same-box
template <int ACT, int IN>
__global__ __launch_bounds__(256) void dec_gemv_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, float eps,
                                                       const float* __restrict__ Wt, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ y,
                                                       float* __restrict__ u_out, int K, int N, int D) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [K] + 8
    float* red = xs + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this wave's weight row is requested FIRST: it does not depend on the input, and the LayerNorm / combine prologue
    // below (two block reductions) then runs under the HBM/L2 latency instead of in front of it
    const int n = blockIdx.x * 4 + wave;
    const float* wr = Wt + (int64_t)min(n, N - 1) * K;
    f32x4 wv[GV_MAXI];
#pragma unroll
    for (int i = 0; i < GV_MAXI; i++) {
        const int k = (lane + 64 * i) * 4;
        wv[i] = (k < K) ? *reinterpret_cast<const f32x4*>(wr + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (IN == 1) {
        float s = 0.f;
        for (int k = tid; k < K; k += 256) s += x[k];
        float mu = block_sum(s, red) / (float)K;
        float q = 0.f;
        for (int k = tid; k < K; k += 256) { float dd = x[k] - mu; q += dd * dd; }
        float var = block_sum(q, red) / (float)K;
        float rs = 1.0f / sqrtf(var + eps);
        for (int k = tid; k < K; k += 256) {
            float v = (x[k] - mu) * rs * ln_g[k] + ln_b[k];
            xs[k] = v;
            if (u_out && blockIdx.x == 0) u_out[k] = v;
        }
    } else if (IN == 2) {
        // x: attention partials [H][ATT_SPLITS][D+2] = {o[D] (unnormalised), running max, sum}; K = H*D.
        // First the H*ATT_SPLITS combine weights exp(m_s - m) / sum (one exponential each), then the weighted sums.
        float* cw = red + 8;                              // [H * ATT_SPLITS] (the launcher sizes the LDS for it)
        const int H = K / D;
        for (int t = tid; t < H * ATT_SPLITS; t += 256) {
            const int h = t / ATT_SPLITS;
            const float* p = x + (size_t)h * ATT_SPLITS * (D + 2);
            float mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < ATT_SPLITS; s++) mx = fmaxf(mx, p[s * (D + 2) + D]);
            float den = 0.f;
#pragma unroll
            for (int s = 0; s < ATT_SPLITS; s++) den += expf(p[s * (D + 2) + D] - mx) * p[s * (D + 2) + D + 1];
            cw[t] = expf(p[(t % ATT_SPLITS) * (D + 2) + D] - mx) / den;      // exp(-inf) = 0 for empty splits
        }
        __syncthreads();
        for (int k = tid; k < K; k += 256) {
            const int h = k / D, dd = k % D;
            const float* p = x + (size_t)h * ATT_SPLITS * (D + 2);
            float num = 0.f;
#pragma unroll
            for (int s = 0; s < ATT_SPLITS; s++) num += cw[h * ATT_SPLITS + s] * p[s * (D + 2) + dd];
            xs[k] = num;
        }
    } else {
        for (int k = tid; k < K; k += 256) {
            float v = x[k];
            xs[k] = v;
            if (u_out && blockIdx.x == 0) u_out[k] = v;
        }
    }
    __syncthreads();
    if (n >= N) return;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < GV_MAXI; i++) {
        const int k = (lane + 64 * i) * 4;
        if (k < K) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(xs + k);
            acc += xv[0] * wv[i][0] + xv[1] * wv[i][1] + xv[2] * wv[i][2] + xv[3] * wv[i][3];
        }
    }
    float v = wave_sum(acc);
    if (lane == 0) {
        if (bias) v += bias[n];
        if (ACT == 1) v = gelu_f<true>(v);
        if (resid) v += resid[n];
        y[n] = v;
    }
}

// Split-key single-query attention: grid (H, ATT_SPLITS).  Workgroup (h, s) owns keys [s*chunk, (s+1)*chunk) of head h
// (chunk = W/ATT_SPLITS); the one that owns position `pos` appends this token's k,v to the cache.  Scores: 16 lanes
// cooperate on one key (16-byte coalesced reads of the [W][D] cache rows, D <= 64... 128 via two passes), 4 keys per
// wave-instruction.  Output: unnormalised o[D], max, sum per (h, s); the next kernel (c_proj GEMV, IN=2) combines them.
__global__ __launch_bounds__(256) void dec_attn_kernel(const float* __restrict__ qkv, float* __restrict__ kc,
                                                       float* __restrict__ vc, float* __restrict__ part,
                                                       const DecState* __restrict__ st, int E, int D, int W, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // q[D] | red[8] | scores[cap] | opart[256]
    const int cap = (W + ATT_SPLITS - 1) / ATT_SPLITS + 4;
    float* qs = sm;
    float* red = sm + D;
    float* sc = red + 8;
    float* opart = sc + cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.x, sp = blockIdx.y;
    const int pos = st->pos;
    // the pos+1 live keys are split evenly (multiples of 4) over the ATT_SPLITS workgroups of this head
    const int chunk = ((pos + 1 + ATT_SPLITS - 1) / ATT_SPLITS + 3) & ~3;
    const int j0 = sp * chunk, j1 = min(pos + 1, j0 + chunk);       // keys [j0, j1) of this split (may be empty)
    float* kh = kc + (int64_t)h * W * D;
    float* vh = vc + (int64_t)h * W * D;
    float* out = part + ((size_t)h * ATT_SPLITS + sp) * (D + 2);
    if (tid < D) {
        qs[tid] = qkv[h * D + tid];
        if (pos >= j0 && pos < j1) {
            kh[(int64_t)pos * D + tid] = qkv[E + h * D + tid];
            vh[(int64_t)pos * D + tid] = qkv[2 * E + h * D + tid];
        }
    }
    __threadfence_block();
    __syncthreads();
    const int nk = j1 - j0;
    if (nk <= 0) {
        if (tid < D) out[tid] = 0.f;
        if (tid == 0) { out[D] = -INFINITY; out[D + 1] = 0.f; }
        return;
    }
    // scores: LPK lanes per key, each lane owns 4 consecutive d; 8 keys in flight per lane group
    const int LPK = D / 4;                       // 4, 8, 16 or 32 lanes per key
    const int kpw = 64 / LPK;                    // keys per wave-instruction
    const int c = lane % LPK, kk = lane / LPK;
    const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + 4 * c);
    float mx = -INFINITY;
    const int stride = 4 * kpw;
    for (int jb = wave * kpw + kk; jb < nk; jb += 8 * stride) {
        f32x4 kv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = jb + u * stride;
            kv[u] = (j < nk) ? *reinterpret_cast<const f32x4*>(kh + (int64_t)(j0 + j) * D + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = jb + u * stride;
            float a = qv[0] * kv[u][0] + qv[1] * kv[u][1] + qv[2] * kv[u][2] + qv[3] * kv[u][3];
            for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o);
            a *= scale;
            if (j < nk) {
                if (c == 0) sc[j] = a;
                mx = fmaxf(mx, a);
            }
        }
    }
    mx = block_max(mx, red);
    float s = 0.f;
    for (int j = tid; j < nk; j += 256) {
        float p = expf(sc[j] - mx);
        sc[j] = p;
        s += p;
    }
    s = block_sum(s, red);
    __syncthreads();
    const int groups = 256 / D;
    const int g = tid / D, dd = tid % D;
    float o = 0.f;
    for (int jb = g; jb < nk; jb += 8 * groups) {
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = jb + u * groups;
            vv[u] = (j < nk) ? vh[(int64_t)(j0 + j) * D + dd] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int j = jb + u * groups;
            if (j < nk) o += sc[j] * vv[u];
        }
    }
    opart[tid] = o;
    __syncthreads();
    if (tid < D) {
        float t = 0.f;
        for (int gg = 0; gg < groups; gg++) t += opart[gg * D + tid];
        out[tid] = t;
    }
    if (tid == 0) { out[D] = mx; out[D + 1] = s; }
}

// choose the next id from logits[V]; record it; build the next input embedding x = wte[id] + wpe[pos']
__global__ __launch_bounds__(256) void dec_sample_kernel(const float* __restrict__ logits, int ldz_row_off, int V,
                                                         float temperature, unsigned seed, DecState* __restrict__ st,
                                                         int32_t* __restrict__ ids, const float* __restrict__ wte,
                                                         const float* __restrict__ wpe, float* __restrict__ x, int E,
                                                         int first) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ int chosen;
    const int tid = threadIdx.x;
    const float* z = logits + ldz_row_off;
    const unsigned ctr = st->rng;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = tid; c < V; c += 256) {
        float v = z[c];
        if (temperature > 0.f) {
            unsigned hsh = drop_hash(seed, 0xC0FFEEu + ctr, (uint64_t)c);
            // 23 bits + 0.5: every value is exact in fp32 and strictly inside (0,1) (24 bits + 0.5 rounds up to 1.0 for the
            // top value: -log(-log(1)) = +inf, one token in ~43 000 would ignore its logit)
            float u = ((float)(hsh >> 9) + 0.5f) * (1.0f / 8388608.0f);
            v = v / temperature - logf(-logf(u));
        }
        if (v > best) { best = v; arg = c; }
    }
    bv[tid] = best;
    bi[tid] = arg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            float ov = bv[tid + s];
            int oi = bi[tid + s];
            if (ov > bv[tid] || (ov == bv[tid] && oi < bi[tid])) { bv[tid] = ov; bi[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        int id = bi[0];
        chosen = id;
        int n = st->produced;
        if (n < st->cap) ids[n] = id;
        st->produced = n + 1;
        st->rng = ctr + 1;
        st->token = id;
        if (!first) st->pos = st->advance ? st->pos + 1 : 0;
    }
    __syncthreads();
    const int id = chosen;
    int pos = st->pos;             // written by tid 0 above, visible after the barrier
    if (pos >= st->W) pos = st->W - 1;   // host refuses to step past the table; never index outside it
    for (int e = tid; e < E; e += 256) x[e] = wte[(int64_t)id * E + e] + wpe[(int64_t)pos * E + e];
}


// =================================================================================================
// v2 per-token kernels (default).  Same arithmetic, shorter dependent chains inside each kernel:
//   * reductions by DPP row operations + v_readlane (no ds_bpermute round trips, one workgroup barrier per block reduction)
//   * the LayerNorm prologue keeps its input in registers (one global read of x instead of three)
//   * a column can be split over 2 or 4 waves so that the narrow GEMVs (N = E) still fill all 256 CUs
//   * attention: K cache stored [H][D/4][W][4] so that TWO lanes own a key (no 16-lane shuffle per score), the current
//     token's k/v are taken from the c_attn output instead of a write -> barrier -> read through the cache, and the V rows
//     are requested before the softmax so their latency hides under it.
// COMPOSER_DECODE_V1=1 selects the first-generation kernels above (A/B timing on one box).
// =================================================================================================
#define DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
#define DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define DPP_XOR1 0xB1          // quad_perm(1,0,3,2)
#define DPP_XOR2 0x4E          // quad_perm(2,3,0,1)
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
__device__ __forceinline__ float rl_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
// sum / max of the 64 lanes, same value (and same summation order) in every lane
__device__ __forceinline__ float wave_sum2(float v) {
    v += DPP_F(v, DPP_XOR1);
    v += DPP_F(v, DPP_XOR2);
    v += DPP_F(v, DPP_HALF_MIRROR);
    v += DPP_F(v, DPP_MIRROR);
    return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}
__device__ __forceinline__ float wave_max2(float v) {
    v = fmaxf(v, DPP_F(v, DPP_XOR1));
    v = fmaxf(v, DPP_F(v, DPP_XOR2));
    v = fmaxf(v, DPP_F(v, DPP_HALF_MIRROR));
    v = fmaxf(v, DPP_F(v, DPP_MIRROR));
    return fmaxf(fmaxf(rl_f(v, 0), rl_f(v, 16)), fmaxf(rl_f(v, 32), rl_f(v, 48)));
}
// block (4 waves) reductions: one barrier; red must not be in use by a previous reduction that other waves may still read
__device__ __forceinline__ float block_sum2(float v, float* red) {
    v = wave_sum2(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max2(float v, float* red) {
    v = wave_max2(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

#define GV2_XR 8            // LayerNorm input values per thread held in registers: K <= 256 * GV2_XR
template <int ACT, int IN>
__global__ __launch_bounds__(256) void dec_gemv2_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, float eps,
                                                        const float* __restrict__ Wt, const float* __restrict__ bias,
                                                        const float* __restrict__ resid, float* __restrict__ y,
                                                        float* __restrict__ u_out, int K, int N, int D, int wpc) {
    extern __shared__ __attribute__((aligned(16))) float xs[];   // [K] | red[16] | combine weights
    float* red = xs + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave -> (column, K part): wpc waves share one column
    const int cpw = 4 / wpc;                       // columns per workgroup
    const int n = blockIdx.x * cpw + wave / wpc;
    const int part = wave % wpc;
    const int Kp = K / wpc;
    const float* wr = Wt + (int64_t)min(n, N - 1) * K + part * Kp;
    f32x4 wv[GV_MAXI];
#pragma unroll
    for (int i = 0; i < GV_MAXI; i++) {
        const int k = (lane + 64 * i) * 4;
        wv[i] = (k < Kp) ? *reinterpret_cast<const f32x4*>(wr + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (IN == 1) {
        float xr[GV2_XR], gr[GV2_XR], br[GV2_XR];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < GV2_XR; i++) {
            const int k = tid + 256 * i;
            xr[i] = k < K ? x[k] : 0.f;
            gr[i] = k < K ? ln_g[k] : 0.f;
            br[i] = k < K ? ln_b[k] : 0.f;
            s += xr[i];
        }
        const float mu = block_sum2(s, red) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < GV2_XR; i++) {
            const int k = tid + 256 * i;
            const float dd = k < K ? xr[i] - mu : 0.f;
            q += dd * dd;
        }
        const float var = block_sum2(q, red + 4) / (float)K;
        const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < GV2_XR; i++) {
            const int k = tid + 256 * i;
            if (k < K) {
                const float v = (xr[i] - mu) * rs * gr[i] + br[i];
                xs[k] = v;
                if (u_out && blockIdx.x == 0) u_out[k] = v;
            }
        }
    } else if (IN == 2) {
        float* cw = red + 16;                             // [H * ATT_SPLITS]
        const int H = K / D;
        for (int t = tid; t < H * ATT_SPLITS; t += 256) {
            const int h = t / ATT_SPLITS;
            const float* p = x + (size_t)h * ATT_SPLITS * (D + 2);
            float mv[ATT_SPLITS], sv[ATT_SPLITS];
            float mx = -INFINITY;
#pragma unroll
            for (int sI = 0; sI < ATT_SPLITS; sI++) { mv[sI] = p[sI * (D + 2) + D]; sv[sI] = p[sI * (D + 2) + D + 1]; mx = fmaxf(mx, mv[sI]); }
            float den = 0.f;
#pragma unroll
            for (int sI = 0; sI < ATT_SPLITS; sI++) den += expf(mv[sI] - mx) * sv[sI];
            cw[t] = expf(mv[t % ATT_SPLITS] - mx) / den;                    // exp(-inf) = 0 for empty splits
        }
        // the partial outputs are requested before the barrier that publishes the weights
        float pv[2][ATT_SPLITS];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int k = tid + 256 * i;
            const int h = k < K ? k / D : 0, dd = k < K ? k % D : 0;
            const float* p = x + (size_t)h * ATT_SPLITS * (D + 2);
#pragma unroll
            for (int sI = 0; sI < ATT_SPLITS; sI++) pv[i][sI] = p[sI * (D + 2) + dd];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int k = tid + 256 * i;
            if (k < K) {
                const int h = k / D;
                float num = 0.f;
#pragma unroll
                for (int sI = 0; sI < ATT_SPLITS; sI++) num += cw[h * ATT_SPLITS + sI] * pv[i][sI];
                xs[k] = num;
            }
        }
        for (int k = tid + 512; k < K; k += 256) {                          // K > 512 (wider models): the plain loop
            const int h = k / D, dd = k % D;
            const float* p = x + (size_t)h * ATT_SPLITS * (D + 2);
            float num = 0.f;
#pragma unroll
            for (int sI = 0; sI < ATT_SPLITS; sI++) num += cw[h * ATT_SPLITS + sI] * p[sI * (D + 2) + dd];
            xs[k] = num;
        }
    } else {
        for (int k = tid * 4; k < K; k += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + k);
            *reinterpret_cast<f32x4*>(xs + k) = v;
            if (u_out && blockIdx.x == 0) *reinterpret_cast<f32x4*>(u_out + k) = v;
        }
    }
    __syncthreads();
    float acc = 0.f;
    const float* xp = xs + part * Kp;
#pragma unroll
    for (int i = 0; i < GV_MAXI; i++) {
        const int k = (lane + 64 * i) * 4;
        if (k < Kp) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(xp + k);
            acc += xv[0] * wv[i][0] + xv[1] * wv[i][1] + xv[2] * wv[i][2] + xv[3] * wv[i][3];
        }
    }
    float v = wave_sum2(acc);
    if (wpc > 1) {                                   // fixed-order sum of the column's K parts
        float* r2 = red + 8;
        if (lane == 0) r2[wave] = v;
        __syncthreads();
        if (part != 0) return;
        v = r2[wave];
        for (int pI = 1; pI < wpc; pI++) v += r2[wave + pI];
    }
    if (n >= N) return;
    if (lane == 0) {
        if (bias) v += bias[n];
        if (ACT == 1) v = gelu_f<true>(v);
        if (resid) v += resid[n];
        y[n] = v;
    }
}

// Split-key single-query attention, grid (H, ATT_SPLITS).  K cache layout [H][D/4][W][4] (chunk-major): the two lanes that
// own key j read chunk c of it at ((h*D/4 + c)*W + j)*4 -- 32 consecutive keys per half wave are one contiguous 512 bytes.
// V cache [H][W][D].  The current token (key == pos) is read from the c_attn output and appended to both caches by the
// workgroup whose key range holds it.
__host__ __device__ static inline int attn2_cap(int W) { return ((W + ATT_SPLITS - 1) / ATT_SPLITS + 4 + 3) & ~3; }   // score slots (multiple of 4)
template <int D>
__global__ __launch_bounds__(256) void dec_attn2_kernel(const float* __restrict__ qkv, float* __restrict__ kcT,
                                                        float* __restrict__ vc, float* __restrict__ part,
                                                        const DecState* __restrict__ st, int E, int W, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // scores[cap] | red[16] | opart[groups][D]
    constexpr int CH = D / 4;                       // 16-byte chunks per row
    constexpr int CPL = CH / 2;                     // chunks per lane of a key pair
    constexpr int GROUPS = 256 / CH;                // V phase: key groups
    constexpr int VMAX = 8;
    const int cap = attn2_cap(W);
    float* sc = sm;
    float* red = sm + cap;
    float* opart = red + 16;
    const int tid = threadIdx.x, h = blockIdx.x, sp = blockIdx.y;
    const int pos = st->pos;
    const int chunk = ((pos + 1 + ATT_SPLITS - 1) / ATT_SPLITS + 3) & ~3;
    const int j0 = sp * chunk, j1 = min(pos + 1, j0 + chunk);
    const int nk = j1 - j0;
    float* out = part + ((size_t)h * ATT_SPLITS + sp) * (D + 2);
    if (nk <= 0) {
        if (tid < D) out[tid] = 0.f;
        if (tid == 0) { out[D] = -INFINITY; out[D + 1] = 0.f; }
        return;
    }
    const float* qh = qkv + h * D;
    const float* kcur = qkv + E + h * D;
    const float* vcur = qkv + 2 * E + h * D;
    float* kh = kcT + (int64_t)h * CH * W * 4;
    float* vh = vc + (int64_t)h * W * D;
    // ---- scores
    const int pair = tid >> 1, half = tid & 1;
    f32x4 qv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) qv[c] = *reinterpret_cast<const f32x4*>(qh + (half * CPL + c) * 4);
    // V phase mapping and its first pass of loads (no dependence on the scores)
    const int g = tid / CH, vcI = tid % CH;
    f32x4 vv[VMAX];
#pragma unroll
    for (int u = 0; u < VMAX; u++) {
        const int j = g + u * GROUPS;
        const int key = j0 + j;
        const float* src = (key == pos) ? vcur + vcI * 4 : vh + (int64_t)key * D + vcI * 4;
        vv[u] = (j < nk) ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float mx = -INFINITY;
    for (int jb = 0; jb < nk; jb += 128) {
        const int j = jb + pair;
        const int key = j0 + j;
        const bool valid = j < nk;
        const bool cur = key == pos;
        f32x4 kv[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int cc = half * CPL + c;
            const float* src = cur ? kcur + cc * 4 : kh + ((int64_t)cc * W + key) * 4;
            kv[c] = valid ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; c++) a += qv[c][0] * kv[c][0] + qv[c][1] * kv[c][1] + qv[c][2] * kv[c][2] + qv[c][3] * kv[c][3];
        a += DPP_F(a, DPP_XOR1);
        a *= scale;
        if (valid) {
            if (half == 0) sc[j] = a;
            mx = fmaxf(mx, a);
        }
    }
    // append the current token to the caches (read by later tokens only)
    if (pos >= j0 && pos < j1 && tid < D) {
        kh[((int64_t)(tid >> 2) * W + pos) * 4 + (tid & 3)] = kcur[tid];
        vh[(int64_t)pos * D + tid] = vcur[tid];
    }
    mx = block_max2(mx, red);            // the barrier inside also publishes sc[]
    float s = 0.f;
    for (int j = tid; j < nk; j += 256) {
        const float p = expf(sc[j] - mx);
        sc[j] = p;
        s += p;
    }
    s = block_sum2(s, red + 4);          // barrier: p values visible
    // ---- P.V
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < VMAX; u++) {
        const int j = g + u * GROUPS;
        if (j < nk) { const float p = sc[j]; o += vv[u] * p; }
    }
    for (int jb = VMAX * GROUPS; jb < nk; jb += VMAX * GROUPS) {           // long key ranges: further passes
#pragma unroll
        for (int u = 0; u < VMAX; u++) {
            const int j = jb + g + u * GROUPS;
            const int key = j0 + j;
            const float* src = (key == pos) ? vcur + vcI * 4 : vh + (int64_t)key * D + vcI * 4;
            vv[u] = (j < nk) ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < VMAX; u++) {
            const int j = jb + g + u * GROUPS;
            if (j < nk) { const float p = sc[j]; o += vv[u] * p; }
        }
    }
    *reinterpret_cast<f32x4*>(opart + g * D + vcI * 4) = o;
    __syncthreads();
    if (tid < D) {
        float t = 0.f;
#pragma unroll
        for (int gg = 0; gg < GROUPS; gg++) t += opart[gg * D + tid];
        out[tid] = t;
    }
    if (tid == 0) { out[D] = mx; out[D + 1] = s; }
}

// K rows of the prompt into the chunk-major cache of dec_attn2_kernel
template <typename T>
__global__ void cache_fill2_kernel(const T* __restrict__ qkv, float* __restrict__ kcT, float* __restrict__ vc, int P, int E,
                                   int H, int D, int W) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * E) return;
    int t = i / E, e = i % E, h = e / D, d = e % D;
    kcT[(((int64_t)h * (D / 4) + (d >> 2)) * W + t) * 4 + (d & 3)] = to_f32<T>(qkv[(int64_t)t * 3 * E + E + e]);
    vc[((int64_t)h * W + t) * D + d] = to_f32<T>(qkv[(int64_t)t * 3 * E + 2 * E + e]);
}

// next id from logits[V] (argmax with lowest index on ties, or Gumbel-max), then the next input embedding
__global__ __launch_bounds__(256) void dec_sample2_kernel(const float* __restrict__ logits, int ldz_row_off, int V,
                                                          float temperature, unsigned seed, DecState* __restrict__ st,
                                                          int32_t* __restrict__ ids, const float* __restrict__ wte,
                                                          const float* __restrict__ wpe, float* __restrict__ x, int E,
                                                          int first) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* z = logits + ldz_row_off;
    const unsigned ctr = st->rng;
    const int pos0 = st->pos, adv = st->advance, nprod = st->produced, capI = st->cap, Wn = st->W;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = tid; c < V; c += 256) {
        float v = z[c];
        if (temperature > 0.f) {
            unsigned hsh = drop_hash(seed, 0xC0FFEEu + ctr, (uint64_t)c);
            float u = ((float)(hsh >> 9) + 0.5f) * (1.0f / 8388608.0f);       // 23 bits + 0.5: exact, strictly inside (0,1)
            v = v / temperature - logf(-logf(u));
        }
        if (v > best) { best = v; arg = c; }
    }
#define ARGMAX_STEP(ctrl)                                                        \
    {                                                                            \
        const float ov = DPP_F(best, ctrl);                                      \
        const int oi = DPP_I(arg, ctrl);                                         \
        if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }      \
    }
    ARGMAX_STEP(DPP_XOR1) ARGMAX_STEP(DPP_XOR2) ARGMAX_STEP(DPP_HALF_MIRROR) ARGMAX_STEP(DPP_MIRROR)
#undef ARGMAX_STEP
    for (int r = 16; r < 64; r += 16) {
        const float ov = rl_f(best, r);
        const int oi = __builtin_amdgcn_readlane(arg, r);
        if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
    }
    // lane 0 of every wave now holds the wave's winner (rows merged into row 0's lanes)
    if (lane == 0) { bv[wave] = best; bi[wave] = arg; }
    __syncthreads();
    float fb = bv[0];
    int id = bi[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (bv[w] > fb || (bv[w] == fb && bi[w] < id)) { fb = bv[w]; id = bi[w]; }
    int pos = first ? pos0 : (adv ? pos0 + 1 : 0);
    if (tid == 0) {
        if (nprod < capI) ids[nprod] = id;
        st->produced = nprod + 1;
        st->rng = ctr + 1;
        st->token = id;
        st->pos = pos;
    }
    if (pos >= Wn) pos = Wn - 1;         // host refuses to step past the table; never index outside it
    for (int e = tid; e < E; e += 256) x[e] = wte[(int64_t)id * E + e] + wpe[(int64_t)pos * E + e];
}

// -------------------------------------------------------------------------------------------------
static int launch_gemv(hipStream_t s, int act, int in_mode, const float* x, const float* g, const float* b, float eps,
                       const float* Wt, const float* bias, const float* resid, float* y, float* u_out, int K, int N, int D) {
    CMP_REQUIRE(K % 4 == 0 && K <= 256 * GV_MAXI, "decode gemv: K=%d unsupported (max %d)", K, 256 * GV_MAXI);
    int grid = cdiv(N, 4);
    size_t smem = (size_t)(K + 8 + (in_mode == 2 ? (K / D) * ATT_SPLITS : 0)) * 4;      // input | reduction scratch | combine weights
#define GV(A, I) dec_gemv_kernel<A, I><<<grid, 256, smem, s>>>(x, g, b, eps, Wt, bias, resid, y, u_out, K, N, D)
    if (act == 1) { if (in_mode == 1) GV(1, 1); else if (in_mode == 2) GV(1, 2); else GV(1, 0); }
    else { if (in_mode == 1) GV(0, 1); else if (in_mode == 2) GV(0, 2); else GV(0, 0); }
#undef GV
    KERNEL_CHECK();
    return CMP_OK;
}

static int launch_gemv2(hipStream_t s, int act, int in_mode, const float* x, const float* g, const float* b, float eps,
                        const float* Wt, const float* bias, const float* resid, float* y, float* u_out, int K, int N, int D) {
    // waves per column: the narrow outputs are split over 2 or 4 waves so that >= 256 workgroups exist
    int wpc = 1;
    while (wpc < 4 && cdiv(N * wpc, 4) < 256 && K % (8 * wpc) == 0) wpc *= 2;
    CMP_REQUIRE(K % 4 == 0 && K / wpc <= 256 * GV_MAXI, "decode gemv: K=%d unsupported (max %d)", K, 256 * GV_MAXI);
    CMP_REQUIRE(in_mode != 1 || K <= 256 * GV2_XR, "decode gemv: LayerNorm width %d unsupported (max %d)", K, 256 * GV2_XR);
    const int grid = cdiv(N * wpc, 4);
    size_t smem = (size_t)(K + 16 + (in_mode == 2 ? (K / D) * ATT_SPLITS : 0)) * 4;
#define GV(A, I) dec_gemv2_kernel<A, I><<<grid, 256, smem, s>>>(x, g, b, eps, Wt, bias, resid, y, u_out, K, N, D, wpc)
    if (act == 1) { if (in_mode == 1) GV(1, 1); else if (in_mode == 2) GV(1, 2); else GV(1, 0); }
    else { if (in_mode == 1) GV(0, 1); else if (in_mode == 2) GV(0, 2); else GV(0, 0); }
#undef GV
    KERNEL_CHECK();
    return CMP_OK;
}

static int launch_attn2(hipStream_t s, cmp_model* m, DecodeState* d, const DecLayerW& w, float scale) {
    const size_t smem = (size_t)(attn2_cap(m->W) + 16 + 1024) * 4;
    dim3 grid(m->H, ATT_SPLITS);
    switch (m->D) {
        case 16: dec_attn2_kernel<16><<<grid, 256, smem, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->E, m->W, scale); break;
        case 32: dec_attn2_kernel<32><<<grid, 256, smem, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->E, m->W, scale); break;
        case 64: dec_attn2_kernel<64><<<grid, 256, smem, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->E, m->W, scale); break;
        default: dec_attn2_kernel<128><<<grid, 256, smem, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->E, m->W, scale); break;
    }
    KERNEL_CHECK();
    return CMP_OK;
}

static int enqueue_token_step2(cmp_model* m, DecodeState* d) {
    hipStream_t s = m->ctx->stream;
    const int E = m->E, L = m->L;
    const bool ln = m->cfg.use_layer_norm != 0;
    const float eps = m->cfg.ln_eps;
    const float scale = m->cfg.scale_attention ? 1.0f / sqrtf((float)m->D) : 1.0f;
    for (int i = 0; i < L; i++) {
        const LayerOff& o = m->lo[i];
        const DecLayerW& w = d->lw[i];
        CHECK_RC(launch_gemv2(s, 0, ln ? 1 : 0, d->x, m->P + o.ln1_g, m->P + o.ln1_b, eps, w.attn_wT, m->P + o.attn_b, nullptr,
                              d->qkv, d->u, E, 3 * E, m->D));
        CHECK_RC(launch_attn2(s, m, d, w, scale));
        CHECK_RC(launch_gemv2(s, 0, 2, d->att, nullptr, nullptr, eps, w.proj_wT, m->P + o.proj_b, d->u, d->r, nullptr, E, E, m->D));
        CHECK_RC(launch_gemv2(s, 1, ln ? 1 : 0, d->r, m->P + o.ln2_g, m->P + o.ln2_b, eps, w.fc_wT, m->P + o.fc_b, nullptr, d->g,
                              nullptr, E, 4 * E, m->D));
        CHECK_RC(launch_gemv2(s, 0, 0, d->g, nullptr, nullptr, eps, w.pr_wT, m->P + o.pr_b, d->r, d->x, nullptr, 4 * E, E, m->D));
    }
    CHECK_RC(launch_gemv2(s, 0, 1, d->x, m->P + m->off_lnf_g, m->P + m->off_lnf_b, eps, m->P + m->off_wte, nullptr, nullptr,
                          d->logits, nullptr, E, m->V, m->D));
    dec_sample2_kernel<<<1, 256, 0, s>>>(d->logits, 0, m->V, d->temperature, (unsigned)d->seed, d->st, d->ids, m->P + m->off_wte,
                                         m->P + m->off_wpe, d->x, E, 0);
    KERNEL_CHECK();
    return CMP_OK;
}

// one token: consumes d->x (embedding of st->token at st->pos), produces the next id and the next d->x
static int enqueue_token_step(cmp_model* m, DecodeState* d) {
    if (!d->v1) return enqueue_token_step2(m, d);
    hipStream_t s = m->ctx->stream;
    const int E = m->E, L = m->L;
    const bool ln = m->cfg.use_layer_norm != 0;
    const float eps = m->cfg.ln_eps;
    const float scale = m->cfg.scale_attention ? 1.0f / sqrtf((float)m->D) : 1.0f;
    for (int i = 0; i < L; i++) {
        const LayerOff& o = m->lo[i];
        const DecLayerW& w = d->lw[i];
        CHECK_RC(launch_gemv(s, 0, ln ? 1 : 0, d->x, m->P + o.ln1_g, m->P + o.ln1_b, eps, w.attn_wT, m->P + o.attn_b, nullptr,
                             d->qkv, d->u, E, 3 * E, m->D));
        size_t smem = (size_t)(m->D + 8 + (m->W + ATT_SPLITS - 1) / ATT_SPLITS + 4 + 256) * 4;
        dec_attn_kernel<<<dim3(m->H, ATT_SPLITS), 256, smem, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, E, m->D, m->W, scale);
        KERNEL_CHECK();
        CHECK_RC(launch_gemv(s, 0, 2, d->att, nullptr, nullptr, eps, w.proj_wT, m->P + o.proj_b, d->u, d->r, nullptr, E, E, m->D));
        CHECK_RC(launch_gemv(s, 1, ln ? 1 : 0, d->r, m->P + o.ln2_g, m->P + o.ln2_b, eps, w.fc_wT, m->P + o.fc_b, nullptr, d->g,
                             nullptr, E, 4 * E, m->D));
        CHECK_RC(launch_gemv(s, 0, 0, d->g, nullptr, nullptr, eps, w.pr_wT, m->P + o.pr_b, d->r, d->x, nullptr, 4 * E, E, m->D));
    }
    CHECK_RC(launch_gemv(s, 0, 1, d->x, m->P + m->off_lnf_g, m->P + m->off_lnf_b, eps, m->P + m->off_wte, nullptr, nullptr,
                         d->logits, nullptr, E, m->V, m->D));
    dec_sample_kernel<<<1, 256, 0, s>>>(d->logits, 0, m->V, d->temperature, (unsigned)d->seed, d->st, d->ids, m->P + m->off_wte,
                                        m->P + m->off_wpe, d->x, E, 0);
    KERNEL_CHECK();
    return CMP_OK;
}

extern "C" int cmp_decode_begin(cmp_model* m, const int32_t* prompt, int P, int mode, float temperature, uint64_t seed) {
    CMP_REQUIRE(m && prompt && P > 0, "decode_begin: prompt must hold at least one id");
    CMP_REQUIRE(mode == CMP_DECODE_LITERAL || mode == CMP_DECODE_KV, "decode_begin: bad mode %d", mode);
    CMP_REQUIRE(P <= m->W, "decode_begin: prompt length %d exceeds window_size %d", P, m->W);
    for (int i = 0; i < P; i++)
        CMP_REQUIRE(prompt[i] >= 0 && prompt[i] < m->V, "decode_begin: prompt id %d out of range [0,%d)", prompt[i], m->V);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    hipStream_t s = m->ctx->stream;
    if (m->dec) { HIP_CHECK(hipStreamSynchronize(s)); decode_state_free(m->dec); m->dec = nullptr; }
    DecodeState* d = new DecodeState();
    m->dec = d;
    d->mode = mode;
    { const char* e = getenv("COMPOSER_DECODE_V1"); d->v1 = e && e[0] == '1'; }
    d->temperature = temperature;
    d->seed = seed;
    d->cap = 1 << 16;
    const int E = m->E, L = m->L, W = m->W;
    CHECK_RC(dalloc(d, &d->st, sizeof(DecState)));
    CHECK_RC(dalloc(d, &d->ids, (size_t)d->cap * 4));
    CHECK_RC(dalloc(d, &d->x, (size_t)E * 4));
    CHECK_RC(dalloc(d, &d->u, (size_t)E * 4));
    CHECK_RC(dalloc(d, &d->qkv, (size_t)3 * E * 4));
    CHECK_RC(dalloc(d, &d->att, (size_t)m->H * ATT_SPLITS * (m->D + 2) * 4));     // split-key attention partials
    CHECK_RC(dalloc(d, &d->r, (size_t)E * 4));
    CHECK_RC(dalloc(d, &d->g, (size_t)4 * E * 4));
    CHECK_RC(dalloc(d, &d->logits, (size_t)m->ldz * 4));
    d->lw.resize(L);
    for (int i = 0; i < L; i++) {
        const LayerOff& o = m->lo[i];
        DecLayerW& w = d->lw[i];
        CHECK_RC(dalloc(d, &w.attn_wT, (size_t)3 * E * E * 4));
        CHECK_RC(dalloc(d, &w.proj_wT, (size_t)E * E * 4));
        CHECK_RC(dalloc(d, &w.fc_wT, (size_t)4 * E * E * 4));
        CHECK_RC(dalloc(d, &w.pr_wT, (size_t)4 * E * E * 4));
        CHECK_RC(dalloc(d, &w.kc, (size_t)W * E * 4));
        CHECK_RC(dalloc(d, &w.vc, (size_t)W * E * 4));
        auto tr = [&](const float* in, float* out, int K, int N) {
            dim3 grid(cdiv(N, 32), cdiv(K, 32));
            transpose_kernel<<<grid, 256, 0, s>>>(in, out, K, N);
        };
        tr(m->P + o.attn_w, w.attn_wT, E, 3 * E);
        tr(m->P + o.proj_w, w.proj_wT, E, E);
        tr(m->P + o.fc_w, w.fc_wT, E, 4 * E);
        tr(m->P + o.pr_w, w.pr_wT, 4 * E, E);
        KERNEL_CHECK();
    }
    // prefill: the whole prompt through the batched forward (Transformer.call with past=None)
    CHECK_RC(ensure_workspace(m, 1, P));
    HIP_CHECK(hipMemcpyAsync(m->x_dev, prompt, (size_t)P * 4, hipMemcpyHostToDevice, s));
    CHECK_RC(model_forward(m, m->x_dev, 1, P, false, 0));
    if (mode == CMP_DECODE_KV) {
        for (int i = 0; i < L; i++) {
            int grid = cdiv(P * E, 256);
            if (d->v1) {
                if (m->dtype == CMP_BF16)
                    cache_fill_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, E, m->H, m->D, W);
                else
                    cache_fill_kernel<float><<<grid, 256, 0, s>>>((const float*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, E, m->H, m->D, W);
            } else {
                if (m->dtype == CMP_BF16)
                    cache_fill2_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, E, m->H, m->D, W);
                else
                    cache_fill2_kernel<float><<<grid, 256, 0, s>>>((const float*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, E, m->H, m->D, W);
            }
            KERNEL_CHECK();
        }
    }
    DecState h;
    h.pos = (mode == CMP_DECODE_KV) ? P : 0;     // position of the first generated token when it is fed back
    h.token = 0;
    h.produced = 0;
    h.advance = (mode == CMP_DECODE_KV) ? 1 : 0;
    h.rng = 0;
    h.cap = d->cap;
    h.W = W;
    HIP_CHECK(hipMemcpyAsync(d->st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    // first id from the last prompt row (cli.py:673 `[-1, 0]`)
    if (d->v1)
        dec_sample_kernel<<<1, 256, 0, s>>>(m->logits, (P - 1) * m->ldz, m->V, temperature, (unsigned)seed, d->st, d->ids,
                                            m->P + m->off_wte, m->P + m->off_wpe, d->x, E, 1);
    else
        dec_sample2_kernel<<<1, 256, 0, s>>>(m->logits, (P - 1) * m->ldz, m->V, temperature, (unsigned)seed, d->st, d->ids,
                                             m->P + m->off_wte, m->P + m->off_wpe, d->x, E, 1);
    KERNEL_CHECK();
    HIP_CHECK(hipStreamSynchronize(s));
    d->produced = 1;
    d->returned = 0;
    d->pos = h.pos;
    // capture the per-token chain once
    const char* nog = getenv("COMPOSER_NO_GRAPH");
    if (!(nog && nog[0] == '1')) {
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_token_step(m, d);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != CMP_OK) return rc;
        HIP_CHECK(e);
        d->graph = g;
        HIP_CHECK(hipGraphInstantiate(&d->exec, g, nullptr, nullptr, 0));
    }
    d->begun = true;
    return CMP_OK;
}

extern "C" int cmp_decode_steps(cmp_model* m, int n, int32_t* ids_out) {
    CMP_REQUIRE(m && ids_out && n >= 0, "decode_steps: bad arguments");
    DecodeState* d = m->dec;
    if (!d || !d->begun) {
        cmp_set_error("decode_steps: call cmp_decode_begin first");
        return CMP_ERR_STATE;
    }
    HIP_CHECK(hipSetDevice(m->ctx->device));
    hipStream_t s = m->ctx->stream;
    const int need = d->returned + n;
    CMP_REQUIRE(need <= d->cap, "decode_steps: more than %d ids per decode_begin", d->cap);
    while (d->produced < need) {
        if (d->mode == CMP_DECODE_KV)
            CMP_REQUIRE(d->pos < m->W, "decode_steps: position %d outside the wpe table (window_size %d): "
                        "prompt_len + length - 1 must be <= window_size in kv-cache mode", d->pos, m->W);
        if (d->exec) HIP_CHECK(hipGraphLaunch(d->exec, s));
        else CHECK_RC(enqueue_token_step(m, d));
        d->produced++;
        if (d->mode == CMP_DECODE_KV) d->pos++;
    }
    HIP_CHECK(hipMemcpyAsync(ids_out, d->ids + d->returned, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    d->returned = need;
    return CMP_OK;
}
