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
#include "../model.h"

struct DecState {        // device-resident loop state
    int pos;             // position id of the token about to be consumed
    int token;           // that token
    int produced;        // number of ids written to ids[]
    int advance;         // 1: kv mode (pos += 1 per step), 0: literal (pos stays 0)
    unsigned rng;        // sampling counter
    int cap;             // capacity of ids[]
    int W;               // wpe rows
    float temperature;   // <= 0: greedy.  Temperature and seed live here (not in kernel arguments) so that the captured per-token
    unsigned seed;       // graph does not depend on them and survives from one cmp_decode_begin to the next
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
    bool built = false;                 // buffers allocated, chain captured (reused by later cmp_decode_begin calls)
    int64_t weights_version = -1;       // cmp_model::param_version the transposed decode weights were made from
    bool graph_is_v1 = false, graph_on = false;
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

#ifdef COMPOSER_EXPERIMENTS      // first-generation decode kernels (COMPOSER_DECODE_V1=1 in an experiments build): K cache [H][W][D]
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
#endif

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
#ifndef ATT_SPLITS
#define ATT_SPLITS 4        // round-2 kernels, same box: 4: 124.4 us/token, 8: 128-129, 16: 128.6 (round-1 kernels: 206 / 194 / 222)
#endif
template <int ACT, int IN>
__global__ __launch_bounds__(256) void dec_gemv_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, float eps,
                                                       const float* __restrict__ Wt, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ y,
                                                       float* __restrict__ u_out, int K, int N, int D, int PS) {
    // PS: floats per attention-partial record (IN == 2): D + 2 from dec_attn_kernel, PSTRIDE(D) from dec_attn2_kernel
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
            const float* p = x + (size_t)h * ATT_SPLITS * PS;
            float mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < ATT_SPLITS; s++) mx = fmaxf(mx, p[s * PS + D]);
            float den = 0.f;
#pragma unroll
            for (int s = 0; s < ATT_SPLITS; s++) den += expf(p[s * PS + D] - mx) * p[s * PS + D + 1];
            cw[t] = expf(p[(t % ATT_SPLITS) * PS + D] - mx) / den;      // exp(-inf) = 0 for empty splits
        }
        __syncthreads();
        for (int k = tid; k < K; k += 256) {
            const int h = k / D, dd = k % D;
            const float* p = x + (size_t)h * ATT_SPLITS * PS;
            float num = 0.f;
#pragma unroll
            for (int s = 0; s < ATT_SPLITS; s++) num += cw[h * ATT_SPLITS + s] * p[s * PS + dd];
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
    // rows longer than one register-resident pass (K > 3072: the mlp c_proj of a model wider than 768): further passes
    for (int k0 = 256 * GV_MAXI; k0 < K; k0 += 256 * GV_MAXI) {
#pragma unroll
        for (int i = 0; i < GV_MAXI; i++) {
            const int k = k0 + (lane + 64 * i) * 4;
            wv[i] = (k < K) ? *reinterpret_cast<const f32x4*>(wr + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < GV_MAXI; i++) {
            const int k = k0 + (lane + 64 * i) * 4;
            if (k < K) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(xs + k);
                acc += xv[0] * wv[i][0] + xv[1] * wv[i][1] + xv[2] * wv[i][2] + xv[3] * wv[i][3];
            }
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

#ifdef COMPOSER_EXPERIMENTS      // first-generation attention and sampler kernels
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
                                                         DecState* __restrict__ st,
                                                         int32_t* __restrict__ ids, const float* __restrict__ wte,
                                                         const float* __restrict__ wpe, float* __restrict__ x, int E,
                                                         int first) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ int chosen;
    const int tid = threadIdx.x;
    const float* z = logits + ldz_row_off;
    const unsigned ctr = st->rng;
    const float temperature = st->temperature;
    const unsigned seed = st->seed;
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
        int id = min(max(bi[0], 0), V - 1);         // all-NaN logits leave the sentinel index: never address outside wte
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
#endif  // COMPOSER_EXPERIMENTS


// =================================================================================================
// The per-token kernels (second generation; the first one is kept in experiments builds for A/B timing).  Same arithmetic; every kernel's dependent chain is as short as the
// data flow allows, because a batch-1 token is 5L+2 dependent launches of ~1.8 us boundary each (tools/ubench/graph_chain.hip)
// and what is left to win is inside the kernels:
//   * GEMV: a workgroup is 1-4 INDEPENDENT waves, one output column per wave: no LDS, no barrier.  Each lane loads the
//     slices of the input vector that face its weight slices straight from global memory (L2), so the LayerNorm statistics
//     / the split-key attention combine are wave-local (DPP row operations + v_readlane) and recomputed by every wave.
//   * attention: the K cache is stored [H][D/4][W][4] so that TWO lanes own a key (one DPP add per score instead of a
//     16-lane shuffle tree); each wave runs its own online softmax over its keys and the four waves meet at ONE barrier;
//     the current token's k/v come from the c_attn output instead of a write -> barrier -> read through the cache.
// In an experiments build (-DCOMPOSER_EXPERIMENTS) COMPOSER_DECODE_V1=1 selects the first-generation kernels above.
// =================================================================================================
#define DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
#define DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define DPP_XOR1 0xB1          // quad_perm(1,0,3,2)
#define DPP_XOR2 0x4E          // quad_perm(2,3,0,1)
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
__device__ __forceinline__ float rl_f(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
// sum / max of the 64 lanes, same value (and the same summation order) in every lane
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

#define PSTRIDE(D) ((D) + 4)          // attention partial record: o[D], running max, sum, 2 pad floats (16-byte aligned rows)

// y[n] = act( IN(x) . Wt[n,:] + bias[n] ) + resid[n]; a wave owns CW adjacent output columns, blockDim.x / 64 waves per workgroup.
// KI = 16-byte chunks of a row per lane (K <= 256 * KI): the loads are unrolled KI times, so the narrow model widths do not carry
// twelve predicated slots.  CW = 2 halves the number of waves to dispatch for the wide outputs and shares the input prologue.
template <int ACT, int IN, int KI, int CW>
__global__ __launch_bounds__(256) void dec_gemv2_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, float eps,
                                                        const float* __restrict__ Wt, const float* __restrict__ bias,
                                                        const float* __restrict__ resid, float* __restrict__ y,
                                                        float* __restrict__ u_out, int K, int N, int D) {
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * CW;
    if (n0 >= N) return;                                           // whole wave; nothing below synchronises waves
    f32x4 wv[CW][KI], xv[KI];
#pragma unroll
    for (int c = 0; c < CW; c++) {
        const float* wr = Wt + (int64_t)min(n0 + c, N - 1) * K;
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int k = (lane + 64 * i) * 4;
            // streamed once per token by one wave: non-temporal (MI355X_MICROARCH "nt-weights": shorter issue-to-landed time)
#ifdef DEC_HALF_W      // timing experiment only (wrong results): what a half-size (bf16) weight stream would cost
            wv[c][i] = (k < K && (i & 1) == 0) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wr + k / 2)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#else
            wv[c][i] = (k < K) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wr + k)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
        }
    }
    if (IN == 2) {
        // x: attention partials [H][ATT_SPLITS][PSTRIDE]; K = H*D.  Chunk k..k+3 lies in head k/D.
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int k = (lane + 64 * i) * 4;
            f32x4 num = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                const int h = k / D, dd = k % D;
                const float* p = x + (size_t)h * ATT_SPLITS * PSTRIDE(D);
                f32x4 ov[ATT_SPLITS];
                float mv[ATT_SPLITS], sv[ATT_SPLITS];
                float mx = -INFINITY;
#pragma unroll
                for (int sI = 0; sI < ATT_SPLITS; sI++) {
                    ov[sI] = *reinterpret_cast<const f32x4*>(p + sI * PSTRIDE(D) + dd);
                    const f32x2 ms = *reinterpret_cast<const f32x2*>(p + sI * PSTRIDE(D) + D);
                    mv[sI] = ms[0]; sv[sI] = ms[1];
                    mx = fmaxf(mx, mv[sI]);
                }
                float den = 0.f;
#pragma unroll
                for (int sI = 0; sI < ATT_SPLITS; sI++) { mv[sI] = expf(mv[sI] - mx); den += mv[sI] * sv[sI]; }   // exp(-inf) = 0: empty splits
                const float inv = 1.0f / den;
#pragma unroll
                for (int sI = 0; sI < ATT_SPLITS; sI++) num += ov[sI] * (mv[sI] * inv);
            }
            xv[i] = num;
        }
    } else {
#pragma unroll
        for (int i = 0; i < KI; i++) {
            const int k = (lane + 64 * i) * 4;
            xv[i] = (k < K) ? *reinterpret_cast<const f32x4*>(x + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (IN == 1) {
            f32x4 gv[KI], bv[KI];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < KI; i++) {
                const int k = (lane + 64 * i) * 4;
                gv[i] = (k < K) ? *reinterpret_cast<const f32x4*>(ln_g + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
                bv[i] = (k < K) ? *reinterpret_cast<const f32x4*>(ln_b + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
                s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
            }
            const float mu = wave_sum2(s) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < KI; i++) {
                const int k = (lane + 64 * i) * 4;
                if (k < K) {
                    const f32x4 dv = xv[i] - mu;
                    q += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
                }
            }
            const float var = wave_sum2(q) / (float)K;
            const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
            for (int i = 0; i < KI; i++) xv[i] = (xv[i] - mu) * rs * gv[i] + bv[i];      // padding lanes: gamma = beta = 0
        }
        if (u_out && n0 == 0) {
#pragma unroll
            for (int i = 0; i < KI; i++) {
                const int k = (lane + 64 * i) * 4;
                if (k < K) *reinterpret_cast<f32x4*>(u_out + k) = xv[i];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CW; c++) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KI; i++) acc += (xv[i][0] * wv[c][i][0] + xv[i][1] * wv[c][i][1]) + (xv[i][2] * wv[c][i][2] + xv[i][3] * wv[c][i][3]);
        float v = wave_sum2(acc);
        const int n = n0 + c;
        if (lane == 0 && n < N) {
            if (bias) v += bias[n];
            if (ACT == 1) v = gelu_f<true>(v);
            if (resid) v += resid[n];
            y[n] = v;
        }
    }
}

// Split-key single-query attention, grid (H, ATT_SPLITS), 4 waves.  K cache [H][D/4][W][4] (chunk-major): the two lanes
// that own key j read chunk c of it at ((h*D/4 + c)*W + j)*4 -- 32 consecutive keys are one contiguous 512 bytes per chunk.
// V cache [H][W][D].  Wave w takes keys w*32 .. w*32+31 of every 128-key pass and keeps its own running max / sum / output;
// the four waves' partials meet in LDS behind one barrier.  The current token (key == pos) is read from the c_attn output
// and appended to both caches by the workgroup whose key range holds it.
template <int D>
__global__ __launch_bounds__(256) void dec_attn2_kernel(const float* __restrict__ qkv, float* __restrict__ kcT,
                                                        float* __restrict__ vc, float* __restrict__ part,
                                                        const DecState* __restrict__ st, int E, int W, float scale) {
    constexpr int CH = D / 4;                       // 16-byte chunks per row
    constexpr int CPL = CH / 2;                     // K chunks per lane of a key pair
    constexpr int KPI = 64 / CH;                    // V rows per wave-instruction
    constexpr int VL = 32 / KPI;                    // V loads per lane per 32-key pass
    __shared__ __attribute__((aligned(16))) float opart[4][KPI][D];
    __shared__ float pw[4][32];
    __shared__ float mw[4], lw[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.x, sp = blockIdx.y;
    const int pos = st->pos;
    const int chunk = ((pos + 1 + ATT_SPLITS - 1) / ATT_SPLITS + 3) & ~3;
    const int j0 = sp * chunk, j1 = min(pos + 1, j0 + chunk);
    const int nk = j1 - j0;
    float* out = part + ((size_t)h * ATT_SPLITS + sp) * PSTRIDE(D);
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
    if (pos >= j0 && pos < j1 && tid < D) {        // append (read by later tokens only)
        kh[((int64_t)(tid >> 2) * W + pos) * 4 + (tid & 3)] = kcur[tid];
        vh[(int64_t)pos * D + tid] = vcur[tid];
    }
    const int pairI = lane >> 1, half = lane & 1;
    const int kg = lane / CH, vcI = lane % CH;
    f32x4 qv[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) qv[c] = *reinterpret_cast<const f32x4*>(qh + (half * CPL + c) * 4);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int jb = wave * 32; jb < nk; jb += 128) {       // wave-uniform: this wave's 32 keys of the pass
        const int j = jb + pairI;
        const int key = j0 + j;
        const bool valid = j < nk;
        f32x4 kv[CPL], vv[VL];
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int cc = half * CPL + c;
            const float* src = (key == pos) ? kcur + cc * 4 : kh + ((int64_t)cc * W + key) * 4;
            kv[c] = valid ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < VL; u++) {
            const int jv = jb + u * KPI + kg;
            const int keyv = j0 + jv;
            const float* src = (keyv == pos) ? vcur + vcI * 4 : vh + (int64_t)keyv * D + vcI * 4;
            vv[u] = (jv < nk) ? *reinterpret_cast<const f32x4*>(src) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; c++) a += (qv[c][0] * kv[c][0] + qv[c][1] * kv[c][1]) + (qv[c][2] * kv[c][2] + qv[c][3] * kv[c][3]);
        a += DPP_F(a, DPP_XOR1);
        a = valid ? a * scale : -INFINITY;
        const float m_new = fmaxf(m_run, wave_max2(a));          // finite: key jb of this wave is valid
        const float p = valid ? expf(a - m_new) : 0.f;
        const float alpha = expf(m_run - m_new);                 // exp(-inf) = 0 on the first pass
        l_run = l_run * alpha + wave_sum2(half == 0 ? p : 0.f);
        m_run = m_new;
        if (half == 0) pw[wave][pairI] = p;
        // (LDS operations of one wave execute in order: the reads below see the writes above without a barrier)
        o *= alpha;
#pragma unroll
        for (int u = 0; u < VL; u++) o += vv[u] * pw[wave][u * KPI + kg];
    }
    *reinterpret_cast<f32x4*>(&opart[wave][kg][vcI * 4]) = o;
    if (lane == 0) { mw[wave] = m_run; lw[wave] = l_run; }
    __syncthreads();
    if (tid < D) {
        const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
        float t = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float f = expf(mw[w] - M);                     // a wave without keys: exp(-inf) = 0
            float ow = 0.f;
#pragma unroll
            for (int r = 0; r < KPI; r++) ow += opart[w][r][tid];
            t += f * ow;
            l += f * lw[w];
        }
        out[tid] = t;
        if (tid == 0) { out[D] = M; out[D + 1] = l; }
    }
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

// The draw itself, shared by the per-token sampler and the kernel-level test entry (cmp_k_sample): every thread of a 256-thread
// workgroup returns the chosen id.  temperature <= 0: argmax, lowest index on ties (tf.argmax).  Otherwise Gumbel-max:
// argmax_c(z[c]/temperature + G_c), G_c = -log(-log(u_c)) with u_c a counter hash of (seed, draw counter, column) -- one
// draw from softmax(z / temperature) (tf.random.categorical, cli.py:671-673).  bv/bi: 4-entry LDS scratch.
__device__ __forceinline__ int sample_block(const float* __restrict__ z, int V, float temperature, unsigned seed, unsigned ctr,
                                            float* bv, int* bi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    const float inv_t = temperature > 0.f ? 1.0f / temperature : 0.f;
    for (int c = tid; c < V; c += 256) {
        float v = z[c];
        if (temperature > 0.f) {
            unsigned hsh = drop_hash(seed, 0xC0FFEEu + ctr, (uint64_t)c);
            float u = ((float)(hsh >> 9) + 0.5f) * (1.0f / 8388608.0f);       // 23 bits + 0.5: exact, strictly inside (0,1)
            v = v * inv_t - __logf(-__logf(u));
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
    for (int r = 16; r < 64; r += 16) {              // rows 1..3 into every lane (lane 0 ends with the wave's winner)
        const float ov = rl_f(best, r);
        const int oi = __builtin_amdgcn_readlane(arg, r);
        if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = arg; }
    __syncthreads();
    float fb = bv[0];
    int id = bi[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (bv[w] > fb || (bv[w] == fb && bi[w] < id)) { fb = bv[w]; id = bi[w]; }
    return min(max(id, 0), V - 1);       // all-NaN logits leave the sentinel index: never address outside wte
}

// next id from logits[V], then the next input embedding
__global__ __launch_bounds__(256) void dec_sample2_kernel(const float* __restrict__ logits, int ldz_row_off, int V,
                                                          DecState* __restrict__ st,
                                                          int32_t* __restrict__ ids, const float* __restrict__ wte,
                                                          const float* __restrict__ wpe, float* __restrict__ x, int E,
                                                          int first) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int tid = threadIdx.x;
    const float* z = logits + ldz_row_off;
    const unsigned ctr = st->rng;
    const float temperature = st->temperature;
    const unsigned seed = st->seed;
    const int pos0 = st->pos, adv = st->advance, nprod = st->produced, capI = st->cap, Wn = st->W;
    int pos = first ? pos0 : (adv ? pos0 + 1 : 0);
    const int posc = min(pos, Wn - 1);     // host refuses to step past the table; never index outside it
    // the position row does not depend on the sampled id: requested now, consumed after the argmax
    float pe[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = tid + 256 * i; pe[i] = e < E ? wpe[(int64_t)posc * E + e] : 0.f; }
    const int id = sample_block(z, V, temperature, seed, ctr, bv, bi);
    if (tid == 0) {
        if (nprod < capI) ids[nprod] = id;
        st->produced = nprod + 1;
        st->rng = ctr + 1;
        st->token = id;
        st->pos = pos;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = tid + 256 * i; if (e < E) x[e] = wte[(int64_t)id * E + e] + pe[i]; }
    for (int e = tid + 1024; e < E; e += 256) x[e] = wte[(int64_t)id * E + e] + wpe[(int64_t)posc * E + e];
}

// n independent draws from ONE logits row with counters counter0 .. counter0+n-1: the sampler of the decode chain exposed for
// the distribution test against the oracle's softmax(z / temperature)
__global__ __launch_bounds__(256) void sample_many_kernel(const float* __restrict__ z, int V, float temperature, unsigned seed,
                                                          unsigned counter0, int32_t* __restrict__ ids) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int id = sample_block(z, V, temperature, seed, counter0 + blockIdx.x, bv, bi);
    if (threadIdx.x == 0) ids[blockIdx.x] = id;
}
extern "C" int cmp_k_sample(void* stream, const float* logits, int V, float temperature, uint64_t seed, uint32_t counter0, int n,
                            int32_t* ids_out) {
    CMP_REQUIRE(logits && ids_out && V > 0 && n >= 0, "k_sample: bad arguments");
    if (n == 0) return CMP_OK;
    sample_many_kernel<<<n, 256, 0, (hipStream_t)stream>>>(logits, V, temperature, (unsigned)seed, counter0, ids_out);
    KERNEL_CHECK();
    return CMP_OK;
}

// -------------------------------------------------------------------------------------------------
static int launch_gemv(hipStream_t s, int act, int in_mode, const float* x, const float* g, const float* b, float eps,
                       const float* Wt, const float* bias, const float* resid, float* y, float* u_out, int K, int N, int D,
                       int PS = 0) {
    CMP_REQUIRE(K % 4 == 0 && K <= 8192, "decode gemv: K=%d unsupported (a multiple of 4, at most 8192)", K);
    if (!PS) PS = D + 2;
    int grid = cdiv(N, 4);
    size_t smem = (size_t)(K + 8 + (in_mode == 2 ? (K / D) * ATT_SPLITS : 0)) * 4;      // input | reduction scratch | combine weights
#define GV(A, I) dec_gemv_kernel<A, I><<<grid, 256, smem, s>>>(x, g, b, eps, Wt, bias, resid, y, u_out, K, N, D, PS)
    if (act == 1) { if (in_mode == 1) GV(1, 1); else if (in_mode == 2) GV(1, 2); else GV(1, 0); }
    else { if (in_mode == 1) GV(0, 1); else if (in_mode == 2) GV(0, 2); else GV(0, 0); }
#undef GV
    KERNEL_CHECK();
    return CMP_OK;
}

static int launch_gemv2(hipStream_t s, int act, int in_mode, const float* x, const float* g, const float* b, float eps,
                        const float* Wt, const float* bias, const float* resid, float* y, float* u_out, int K, int N, int D) {
    // rows longer than the register-resident kernel holds: the workgroup-staged kernel in several passes (same partial records)
    if (K > 256 * GV_MAXI) return launch_gemv(s, act, in_mode, x, g, b, eps, Wt, bias, resid, y, u_out, K, N, D, PSTRIDE(D));
    CMP_REQUIRE(K % 4 == 0, "decode gemv: K=%d must be a multiple of 4", K);
    // CW = 2 columns per wave for the wide outputs of the narrow-K launches; 4, 2 or 1 waves per workgroup so that the narrow
    // outputs still give every CU a workgroup
#ifdef DEC_NO_CW2
    const int cw = 1;
#else
    const int cw = (K <= 512 && N >= 1024) ? 2 : 1;
#endif
    const int waves = cdiv(N, cw);
    int block = 256;
    while (block > 64 && cdiv(waves, block / 64) < 256) block >>= 1;
    const int grid = cdiv(waves, block / 64);
#define GV3(A, I, KI, CW) dec_gemv2_kernel<A, I, KI, CW><<<grid, block, 0, s>>>(x, g, b, eps, Wt, bias, resid, y, u_out, K, N, D)
#define GV2(A, I) do { if (K <= 512) { if (cw == 2) GV3(A, I, 2, 2); else GV3(A, I, 2, 1); } else GV3(A, I, GV_MAXI, 1); } while (0)
    if (act == 1) { if (in_mode == 1) GV2(1, 1); else if (in_mode == 2) GV2(1, 2); else GV2(1, 0); }
    else { if (in_mode == 1) GV2(0, 1); else if (in_mode == 2) GV2(0, 2); else GV2(0, 0); }
#undef GV2
#undef GV3
    KERNEL_CHECK();
    return CMP_OK;
}

static int launch_attn2(hipStream_t s, cmp_model* m, DecodeState* d, const DecLayerW& w, float scale) {
    dim3 grid(m->H, ATT_SPLITS);
    switch (m->D) {
        case 16: dec_attn2_kernel<16><<<grid, 256, 0, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->Ea, m->W, scale); break;
        case 32: dec_attn2_kernel<32><<<grid, 256, 0, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->Ea, m->W, scale); break;
        case 64: dec_attn2_kernel<64><<<grid, 256, 0, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->Ea, m->W, scale); break;
        default: dec_attn2_kernel<128><<<grid, 256, 0, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, m->Ea, m->W, scale); break;
    }
    KERNEL_CHECK();
    return CMP_OK;
}

static int enqueue_token_step2(cmp_model* m, DecodeState* d) {
    hipStream_t s = m->ctx->stream;
    // COMPOSER_DECODE_DIAG_SKIP (experiments builds only; timing diagnosis -- the ids are wrong): bit 0 LN1+c_attn, 1 attention,
    // 2 c_proj, 3 LN2+c_fc, 4 mlp c_proj, 5 logits, 6 sampler are left out of the captured chain
    int skip = 0;
#ifdef COMPOSER_EXPERIMENTS
    { const char* e = getenv("COMPOSER_DECODE_DIAG_SKIP"); if (e) skip = atoi(e); }
#endif
    const int E = m->E, Ea = m->Ea, L = m->L;
    const bool ln = m->cfg.use_layer_norm != 0;
    const float eps = m->cfg.ln_eps;
    const float scale = m->cfg.scale_attention ? 1.0f / sqrtf((float)m->Dl) : 1.0f;
    for (int i = 0; i < L; i++) {
        const LayerOff& o = m->lo[i];
#ifdef DEC_ALIAS_L0     // timing experiment only (wrong results): every block reads block 0's weights, which then stay in L2
        const DecLayerW w = [&] { DecLayerW t = d->lw[0]; t.kc = d->lw[i].kc; t.vc = d->lw[i].vc; return t; }();
#else
        const DecLayerW& w = d->lw[i];
#endif
        if (!(skip & 1)) CHECK_RC(launch_gemv2(s, 0, ln ? 1 : 0, d->x, m->P + o.ln1_g, m->P + o.ln1_b, eps, w.attn_wT, m->P + o.attn_b, nullptr,
                              d->qkv, d->u, E, 3 * Ea, m->D));
        if (!(skip & 2)) CHECK_RC(launch_attn2(s, m, d, w, scale));
        if (!(skip & 4)) CHECK_RC(launch_gemv2(s, 0, 2, d->att, nullptr, nullptr, eps, w.proj_wT, m->P + o.proj_b, d->u, d->r, nullptr, Ea, E, m->D));
        if (!(skip & 8)) CHECK_RC(launch_gemv2(s, 1, ln ? 1 : 0, d->r, m->P + o.ln2_g, m->P + o.ln2_b, eps, w.fc_wT, m->P + o.fc_b, nullptr, d->g,
                              nullptr, E, 4 * E, m->D));
        if (!(skip & 16)) CHECK_RC(launch_gemv2(s, 0, 0, d->g, nullptr, nullptr, eps, w.pr_wT, m->P + o.pr_b, d->r, d->x, nullptr, 4 * E, E, m->D));
    }
    if (!(skip & 32)) CHECK_RC(launch_gemv2(s, 0, 1, d->x, m->P + m->off_lnf_g, m->P + m->off_lnf_b, eps, m->P + m->off_wte, nullptr, nullptr,
                          d->logits, nullptr, E, m->V, m->D));
    if (!(skip & 64)) dec_sample2_kernel<<<1, 256, 0, s>>>(d->logits, 0, m->V, d->st, d->ids, m->P + m->off_wte,
                                         m->P + m->off_wpe, d->x, E, 0);
    KERNEL_CHECK();
    return CMP_OK;
}

// one token: consumes d->x (embedding of st->token at st->pos), produces the next id and the next d->x
static int enqueue_token_step(cmp_model* m, DecodeState* d) {
    if (!d->v1) return enqueue_token_step2(m, d);
#ifndef COMPOSER_EXPERIMENTS
    cmp_set_error("decode: the first-generation kernels exist only in an experiments build");
    return CMP_ERR_STATE;
#else
    hipStream_t s = m->ctx->stream;
    const int E = m->E, Ea = m->Ea, L = m->L;
    const bool ln = m->cfg.use_layer_norm != 0;
    const float eps = m->cfg.ln_eps;
    const float scale = m->cfg.scale_attention ? 1.0f / sqrtf((float)m->Dl) : 1.0f;
    for (int i = 0; i < L; i++) {
        const LayerOff& o = m->lo[i];
        const DecLayerW& w = d->lw[i];
        CHECK_RC(launch_gemv(s, 0, ln ? 1 : 0, d->x, m->P + o.ln1_g, m->P + o.ln1_b, eps, w.attn_wT, m->P + o.attn_b, nullptr,
                             d->qkv, d->u, E, 3 * Ea, m->D));
        size_t smem = (size_t)(m->D + 8 + (m->W + ATT_SPLITS - 1) / ATT_SPLITS + 4 + 256) * 4;
        dec_attn_kernel<<<dim3(m->H, ATT_SPLITS), 256, smem, s>>>(d->qkv, w.kc, w.vc, d->att, d->st, Ea, m->D, m->W, scale);
        KERNEL_CHECK();
        CHECK_RC(launch_gemv(s, 0, 2, d->att, nullptr, nullptr, eps, w.proj_wT, m->P + o.proj_b, d->u, d->r, nullptr, Ea, E, m->D));
        CHECK_RC(launch_gemv(s, 1, ln ? 1 : 0, d->r, m->P + o.ln2_g, m->P + o.ln2_b, eps, w.fc_wT, m->P + o.fc_b, nullptr, d->g,
                             nullptr, E, 4 * E, m->D));
        CHECK_RC(launch_gemv(s, 0, 0, d->g, nullptr, nullptr, eps, w.pr_wT, m->P + o.pr_b, d->r, d->x, nullptr, 4 * E, E, m->D));
    }
    CHECK_RC(launch_gemv(s, 0, 1, d->x, m->P + m->off_lnf_g, m->P + m->off_lnf_b, eps, m->P + m->off_wte, nullptr, nullptr,
                         d->logits, nullptr, E, m->V, m->D));
    dec_sample_kernel<<<1, 256, 0, s>>>(d->logits, 0, m->V, d->st, d->ids, m->P + m->off_wte,
                                        m->P + m->off_wpe, d->x, E, 0);
    KERNEL_CHECK();
    return CMP_OK;
#endif
}

extern "C" int cmp_decode_begin(cmp_model* m, const int32_t* prompt, int P, int mode, float temperature, uint64_t seed) {
    CMP_REQUIRE(m && prompt && P > 0, "decode_begin: prompt must hold at least one id");
    CMP_REQUIRE(mode == CMP_DECODE_LITERAL || mode == CMP_DECODE_KV, "decode_begin: bad mode %d", mode);
    CMP_REQUIRE(P <= m->W, "decode_begin: prompt length %d exceeds window_size %d", P, m->W);
    for (int i = 0; i < P; i++)
        CMP_REQUIRE(prompt[i] >= 0 && prompt[i] < m->V, "decode_begin: prompt id %d out of range [0,%d)", prompt[i], m->V);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    hipStream_t s = m->ctx->stream;
    // The decode state (buffers, KV caches, transposed fp32 weights, the captured per-token chain) is built once per model
    // and reused by later calls: of the 7.6 ms a call used to spend here before the first token (45 hipMalloc / hipFree, 24
    // transposes, capture + instantiate: 6 % of a 1024-token generate) what is left is the prefill.  The transposes are
    // redone when a parameter has changed since (cmp_model::param_version); the chain is re-captured only when the kernel
    // generation or the graph switch changes (temperature, seed and mode live in the device-side state).
#ifdef COMPOSER_EXPERIMENTS
    const bool v1 = [] { const char* e = getenv("COMPOSER_DECODE_V1"); return e && e[0] == '1'; }();
#else
    const bool v1 = false;
#endif
    const bool graph_on = [] { const char* e = getenv("COMPOSER_NO_GRAPH"); return !(e && e[0] == '1'); }();
    DecodeState* d = m->dec;
    if (d && !d->built) {                       // an earlier call failed part-way (an allocation, the capture): start over
        HIP_CHECK(hipStreamSynchronize(s));
        decode_state_free(d);
        m->dec = d = nullptr;
    }
    if (d && d->built && (d->graph_is_v1 != v1 || d->graph_on != graph_on)) {
        HIP_CHECK(hipStreamSynchronize(s));
        decode_state_free(d);
        m->dec = d = nullptr;
    }
    if (!d) {
        d = new DecodeState();
        m->dec = d;
    }
    d->begun = false;
    d->mode = mode;
    d->v1 = v1;
    d->temperature = temperature;
    d->seed = seed;
    d->cap = 1 << 16;
    const int E = m->E, Ea = m->Ea, L = m->L, W = m->W;
    if (!d->built) {
        CHECK_RC(dalloc(d, &d->st, sizeof(DecState)));
        CHECK_RC(dalloc(d, &d->ids, (size_t)d->cap * 4));
        CHECK_RC(dalloc(d, &d->x, (size_t)E * 4));
        CHECK_RC(dalloc(d, &d->u, (size_t)E * 4));
        CHECK_RC(dalloc(d, &d->qkv, (size_t)3 * Ea * 4));
        CHECK_RC(dalloc(d, &d->att, (size_t)m->H * ATT_SPLITS * PSTRIDE(m->D) * 4));  // split-key attention partials
        CHECK_RC(dalloc(d, &d->r, (size_t)E * 4));
        CHECK_RC(dalloc(d, &d->g, (size_t)4 * E * 4));
        CHECK_RC(dalloc(d, &d->logits, (size_t)m->ldz * 4));
        d->lw.resize(L);
        for (int i = 0; i < L; i++) {
            DecLayerW& w = d->lw[i];
            CHECK_RC(dalloc(d, &w.attn_wT, (size_t)3 * Ea * E * 4));
            CHECK_RC(dalloc(d, &w.proj_wT, (size_t)Ea * E * 4));
            CHECK_RC(dalloc(d, &w.fc_wT, (size_t)4 * E * E * 4));
            CHECK_RC(dalloc(d, &w.pr_wT, (size_t)4 * E * E * 4));
            CHECK_RC(dalloc(d, &w.kc, (size_t)W * Ea * 4));
            CHECK_RC(dalloc(d, &w.vc, (size_t)W * Ea * 4));
        }
    }
    if (d->weights_version != m->param_version) {
        for (int i = 0; i < L; i++) {
            const LayerOff& o = m->lo[i];
            DecLayerW& w = d->lw[i];
            auto tr = [&](const float* in, float* out, int K, int N) {
                dim3 grid(cdiv(N, 32), cdiv(K, 32));
                transpose_kernel<<<grid, 256, 0, s>>>(in, out, K, N);
            };
            tr(m->P + o.attn_w, w.attn_wT, E, 3 * Ea);
            tr(m->P + o.proj_w, w.proj_wT, Ea, E);
            tr(m->P + o.fc_w, w.fc_wT, E, 4 * E);
            tr(m->P + o.pr_w, w.pr_wT, 4 * E, E);
            KERNEL_CHECK();
        }
        d->weights_version = m->param_version;
    }
    // prefill: the whole prompt through the batched forward (Transformer.call with past=None)
    CHECK_RC(ensure_workspace(m, 1, P));
    HIP_CHECK(hipMemcpyAsync(m->x_dev, prompt, (size_t)P * 4, hipMemcpyHostToDevice, s));
    CHECK_RC(model_forward(m, m->x_dev, 1, P, false, 0));
    if (mode == CMP_DECODE_KV) {
        for (int i = 0; i < L; i++) {
            int grid = cdiv(P * Ea, 256);
#ifdef COMPOSER_EXPERIMENTS
            if (d->v1) {
                if (m->dtype == CMP_BF16)
                    cache_fill_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, Ea, m->H, m->D, W);
                else
                    cache_fill_kernel<float><<<grid, 256, 0, s>>>((const float*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, Ea, m->H, m->D, W);
            } else
#endif
            {
                if (m->dtype == CMP_BF16)
                    cache_fill2_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, Ea, m->H, m->D, W);
                else
                    cache_fill2_kernel<float><<<grid, 256, 0, s>>>((const float*)m->act[i].qkv, d->lw[i].kc, d->lw[i].vc, P, Ea, m->H, m->D, W);
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
    h.temperature = temperature;
    h.seed = (unsigned)seed;
    HIP_CHECK(hipMemcpyAsync(d->st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    // first id from the last prompt row (cli.py:673 `[-1, 0]`)
#ifdef COMPOSER_EXPERIMENTS
    if (d->v1) dec_sample_kernel<<<1, 256, 0, s>>>(m->logits, (P - 1) * m->ldz, m->V, d->st, d->ids, m->P + m->off_wte, m->P + m->off_wpe, d->x, E, 1);
    else
#endif
    dec_sample2_kernel<<<1, 256, 0, s>>>(m->logits, (P - 1) * m->ldz, m->V, d->st, d->ids, m->P + m->off_wte, m->P + m->off_wpe, d->x, E, 1);
    KERNEL_CHECK();
    HIP_CHECK(hipStreamSynchronize(s));
    d->produced = 1;
    d->returned = 0;
    d->pos = h.pos;
    // capture the per-token chain once per decode state
    if (!d->built) {
        if (graph_on) {
            HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int rc = enqueue_token_step(m, d);
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamEndCapture(s, &g);
            if (rc != CMP_OK) return rc;
            HIP_CHECK(e);
            d->graph = g;
            HIP_CHECK(hipGraphInstantiate(&d->exec, g, nullptr, nullptr, 0));
        }
        d->graph_is_v1 = v1;
        d->graph_on = graph_on;
        d->built = true;
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
