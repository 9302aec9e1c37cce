// elementwise.hip -- the HBM-bound kernels of the Transformer hot path (gfx950):
// token+position embedding fwd/bwd, LayerNorm fwd/bwd, fused softmax-cross-entropy fwd+bwd,
// Keras-formulation Adam, column sums (bias gradients).
// Each kernel is one pass over its operands with 16-byte-per-lane coalesced accesses.
#include "model.h"

// =================================================================================================
// Embedding: h[b,t,:] = wte[x[b,t],:] + wpe[pos0+t,:]  (+ dropout)      transformer.py:137-138,786,793-794
// =================================================================================================
// pos_ids / type_ids (per token, or null): Transformer.call's position_ids (:770-773, 786) and token_type_ids (:787-791:
// a second row of wte added to the sum) -- forward passes only; the train loop never passes them (:916-917)
template <typename T>
__global__ void embed_fwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ wte,
                                 const float* __restrict__ wpe, T* __restrict__ out, int ntok, int T_,
                                 int E, int pos0, DropCfg drop, const int32_t* __restrict__ pos_ids,
                                 const int32_t* __restrict__ type_ids) {
    constexpr int VN = Vec16<T>::N;
    const int chunks = E / VN;
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)ntok * chunks;
    for (; gid < total; gid += (int64_t)gridDim.x * blockDim.x) {
        int tok = (int)(gid / chunks);
        int e0 = (int)(gid % chunks) * VN;
        int id = ids[tok];
        int t = tok % T_;
        const float* a = wte + (int64_t)id * E + e0;
        const float* p = wpe + (int64_t)(pos_ids ? pos_ids[tok] : pos0 + t) * E + e0;
        const float* ty = type_ids ? wte + (int64_t)type_ids[tok] * E + e0 : nullptr;
        Vec16<T> r;
        const uint32_t rowh = drop_row_hash(drop, (uint32_t)tok);
#pragma unroll
        for (int i = 0; i < VN; i += 4) {
            f32x4 av = *reinterpret_cast<const f32x4*>(a + i);
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
            f32x4 tv = {0.f, 0.f, 0.f, 0.f};
            if (ty) tv = *reinterpret_cast<const f32x4*>(ty + i);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = ty ? (av[j] + pv[j]) + tv[j] : av[j] + pv[j];        // :793 input + position + token-type
                if (drop.thr) v = apply_drop_rc(drop, rowh, (uint32_t)(e0 + i + j), v);
                r.set(i + j, v);
            }
        }
        st16(out + (int64_t)tok * E + e0, r);
    }
}

// The same sum, one wave per token row, for the LayerNorm-fused block path (model.hip: model_forward): besides the row it
// leaves the row's partial statistics -- (mean, M2) of every 256-column segment of the STORED bf16 values -- for the c_attn
// epilogue of block 0, which applies ln_1 itself (common.h: LnEpi).  A lane owns the 16-byte chunks lane + 64 i; 32 chunks
// make a segment, so a segment is one half of the wave at one i.
template <int MAXI>
__global__ __launch_bounds__(256) void embed_fwd_stats_kernel(const int32_t* __restrict__ ids, const float* __restrict__ wte,
                                                               const float* __restrict__ wpe, bf16_t* __restrict__ out,
                                                               float* __restrict__ part, int ntok, int T_, int E, int pos0, DropCfg drop) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int chunks = E >> 3, np = E >> 8;
    for (int tok = blockIdx.x * wpb + (threadIdx.x >> 6); tok < ntok; tok += gridDim.x * wpb) {
        const int id = ids[tok], t = tok % T_;
        const uint32_t rowh = drop_row_hash(drop, (uint32_t)tok);
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            const int c = lane + 64 * i;
            const bool live = c < chunks;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] = 0.f;
            if (live) {
                const float* a = wte + (int64_t)id * E + c * 8;
                const float* p = wpe + (int64_t)(pos0 + t) * E + c * 8;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(p), p1 = *reinterpret_cast<const f32x4*>(p + 4);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float v = j < 4 ? a0[j] + p0[j] : a1[j - 4] + p1[j - 4];
                    if (drop.thr) v = apply_drop_rc(drop, rowh, (uint32_t)(c * 8 + j), v);
                    o[j] = (bf16_t)v;
                    f[j] = (float)o[j];
                }
                *reinterpret_cast<bf16x8*>(out + (int64_t)tok * E + c * 8) = o;
            }
            // the segment's mean, then its squared deviations (two passes over registers), each summed over the half-wave
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) sm += f[j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            const float mu = sm * (1.0f / 256.0f);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) { const float d = f[j] - mu; q = fmaf(d, d, q); }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const int seg = c >> 5;
            if (live && (lane & 31) == 0) *reinterpret_cast<f32x2*>(part + ((int64_t)tok * np + seg) * 2) = (f32x2){mu, q};
        }
    }
}

// dwte[x[b,t],:] += dh[b,t,:] (f32 atomics, 256 contiguous bytes per wave-instruction);
// dwpe[pos0+t,:] += sum_b dh[b,t,:].   One thread per (t, e); loops over b.
template <typename T>
__global__ void embed_bwd_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dh,
                                 float* __restrict__ dwte, float* __restrict__ dwpe, int B, int T_, int E,
                                 int pos0, DropCfg drop) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T_ * E) return;
    int t = (int)(gid / E), e = (int)(gid % E);
    float acc = 0.f;
    for (int b = 0; b < B; b++) {
        int tok = b * T_ + t;
        float v = to_f32<T>(dh[(int64_t)tok * E + e]);
        v = apply_drop(drop, (uint32_t)tok, (uint32_t)e, v);
        acc += v;
        atomicAdd(dwte + (int64_t)ids[tok] * E + e, v);
    }
    dwpe[(int64_t)(pos0 + t) * E + e] += acc;
}

// Deterministic form of the dwte scatter-add (COMPOSER_DETERMINISTIC=1): no atomics.  Pass 1, grid (V, SEG): workgroup
// (v, g) walks the tokens of segment g in order and sums the rows whose id is v into part[v][g][:]; pass 2 adds a row's
// SEG partials in segment order.  The ids are re-read V times (they stay in L2); every gradient row is read once.
#define EMB_SEG 64
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_det_part_kernel(const int32_t* __restrict__ ids, const T* __restrict__ dh,
                                                                 float* __restrict__ part, int ntok, int E, DropCfg drop) {
    const int v = blockIdx.x, g = blockIdx.y;
    const int per = cdiv(ntok, EMB_SEG);
    const int t0 = g * per, t1 = min(ntok, t0 + per);
    float* out = part + ((size_t)v * EMB_SEG + g) * E;
    for (int e0 = 0; e0 < E; e0 += 256) {
        const int e = e0 + threadIdx.x;
        float acc = 0.f;
        for (int tok = t0; tok < t1; tok++) {
            if (ids[tok] != v) continue;                   // uniform over the workgroup
            if (e < E) acc += apply_drop(drop, (uint32_t)tok, (uint32_t)e, to_f32<T>(dh[(int64_t)tok * E + e]));
        }
        if (e < E) out[e] = acc;
    }
}
__global__ void embed_bwd_det_reduce_kernel(const float* __restrict__ part, float* __restrict__ dwte, int V, int E) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)V * E) return;
    const int v = (int)(i / E), e = (int)(i % E);
    float a = 0.f;
    for (int g = 0; g < EMB_SEG; g++) a += part[((size_t)v * EMB_SEG + g) * E + e];
    dwte[i] += a;
}
// dwpe part of embed_bwd_kernel alone (already a fixed-order loop over the batch)
template <typename T>
__global__ void embed_bwd_wpe_kernel(const T* __restrict__ dh, float* __restrict__ dwpe, int B, int T_, int E, int pos0, DropCfg drop) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T_ * E) return;
    int t = (int)(gid / E), e = (int)(gid % E);
    float acc = 0.f;
    for (int b = 0; b < B; b++) {
        int tok = b * T_ + t;
        acc += apply_drop(drop, (uint32_t)tok, (uint32_t)e, to_f32<T>(dh[(int64_t)tok * E + e]));
    }
    dwpe[(int64_t)(pos0 + t) * E + e] += acc;
}

// ---- sorted form of the dwte scatter-add (default; the atomic kernel above remains for vocabularies too wide for the LDS
// histogram).  The atomic form adds every gradient element on its own: B*T*E f32 atomics = 268 MB through the memory-side
// atomic unit (~1.3 TB/s, MI355X_MICROARCH "Global float atomics") = 225 us at C2, B = 128 -- ten times the time the 134 MB
// of gradient rows take to read.  Here the tokens are counting-sorted by id (three small kernels: per-block histogram in LDS,
// scan, scatter with LDS ranks), a workgroup sums up to EMB_CH rows of ONE id (eight rows per thread in flight) and issues E atomics for them, and the position
// gradient is a plain per-position sum over the batch.  Workspace (int32): [0] chunk count | hist V*NB | base V*NB | total V |
// start V+1 | cstart V+1 | chunk ids B*T/32 + V + 1 | order B*T.
#define EMB_CH 128          // rows of one id per workgroup: 32 / 64 / 128 / 256 measured 64 / 43 / 35 / 40 us at B = 128 (23 / 16 / 15 / 15 at B = 32):
                            // fewer, larger chunks mean fewer same-row atomics into the 390-row table
#define EMB_NB 64
// hist[v][b]: tokens with id v in token block b (EMB_NB blocks)
__global__ __launch_bounds__(256) void embed_hist_kernel(const int32_t* __restrict__ ids, int* __restrict__ hist, int ntok, int V) {
    extern __shared__ int emb_lds[];
    for (int v = threadIdx.x; v < V; v += 256) emb_lds[v] = 0;
    __syncthreads();
    const int per = cdiv(ntok, (int)gridDim.x), t0 = blockIdx.x * per, t1 = min(ntok, t0 + per);
    for (int t = t0 + threadIdx.x; t < t1; t += 256) atomicAdd(&emb_lds[ids[t]], 1);
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += 256) hist[(size_t)v * EMB_NB + blockIdx.x] = emb_lds[v];
}
// thread v: base[v][b] = tokens with id v in blocks before b (its EMB_NB counts are one contiguous 256 bytes), total[v]
__global__ __launch_bounds__(256) void embed_scan1_kernel(const int* __restrict__ hist, int* __restrict__ base, int* __restrict__ total, int V) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4* h = reinterpret_cast<const i4*>(hist + (size_t)v * EMB_NB);
    i4* o = reinterpret_cast<i4*>(base + (size_t)v * EMB_NB);
    i4 c[EMB_NB / 4];
#pragma unroll
    for (int i = 0; i < EMB_NB / 4; i++) c[i] = h[i];
    int run = 0;
#pragma unroll
    for (int i = 0; i < EMB_NB / 4; i++) {
        i4 e;
#pragma unroll
        for (int k = 0; k < 4; k++) { e[k] = run; run += c[i][k]; }
        o[i] = e;
    }
    total[v] = run;
}
// one workgroup: start[v], cstart[v] = exclusive scans over v of total[v] and of its chunk count; chunk_id[j] = the id chunk j belongs to
__global__ __launch_bounds__(1024) void embed_scan2_kernel(const int* __restrict__ total, int* __restrict__ start, int* __restrict__ cstart,
                                                           int* __restrict__ nchunks, int* __restrict__ chunk_id, int V) {
    __shared__ int part[1024], partc[1024];
    const int per = cdiv(V, 1024), v0 = min(V, (int)threadIdx.x * per), v1 = min(V, v0 + per);
    int sum = 0, sumc = 0;
    for (int v = v0; v < v1; v++) { sum += total[v]; sumc += (total[v] + EMB_CH - 1) / EMB_CH; }
    part[threadIdx.x] = sum; partc[threadIdx.x] = sumc;
    __syncthreads();
    // exclusive scan of the 1024 partials (Hillis-Steele on a copy: 10 rounds)
    for (int off = 1; off < 1024; off <<= 1) {
        const int a = threadIdx.x >= off ? part[threadIdx.x - off] : 0, c = threadIdx.x >= off ? partc[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += a; partc[threadIdx.x] += c;
        __syncthreads();
    }
    int a = part[threadIdx.x] - sum, c = partc[threadIdx.x] - sumc;            // inclusive -> exclusive
    if (threadIdx.x == 1023) { *nchunks = partc[1023]; start[V] = part[1023]; cstart[V] = partc[1023]; }
    for (int v = v0; v < v1; v++) {
        start[v] = a; cstart[v] = c;
        const int nc = (total[v] + EMB_CH - 1) / EMB_CH;
        for (int k = 0; k < nc; k++) chunk_id[c + k] = v;
        a += total[v]; c += nc;
    }
}
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int32_t* __restrict__ ids, const int* __restrict__ base,
                                                            const int* __restrict__ start, int* __restrict__ order, int ntok, int V) {
    extern __shared__ int emb_lds[];
    for (int v = threadIdx.x; v < V; v += 256) emb_lds[v] = start[v] + base[(size_t)v * EMB_NB + blockIdx.x];
    __syncthreads();
    const int per = cdiv(ntok, (int)gridDim.x), t0 = blockIdx.x * per, t1 = min(ntok, t0 + per);
    for (int t = t0 + threadIdx.x; t < t1; t += 256) order[atomicAdd(&emb_lds[ids[t]], 1)] = t;
}
// workgroup j: chunk j of the sorted token list (up to EMB_CH rows of one id): dwte[id] += their sum.  RP rows side by side.
template <typename T>
__global__ __launch_bounds__(256) void embed_segsum_kernel(const T* __restrict__ dh, const int* __restrict__ order, const int* __restrict__ total,
                                                           const int* __restrict__ start, const int* __restrict__ cstart,
                                                           const int* __restrict__ nchunks, const int* __restrict__ chunk_id,
                                                           float* __restrict__ dwte, int V, int E, DropCfg drop) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float red[256 * VN];
    const int j = blockIdx.x;
    if (j >= *nchunks) return;
    const int v = chunk_id[j];
    const int first = (j - cstart[v]) * EMB_CH, n = min(EMB_CH, total[v] - first);
    const int* ord = order + start[v] + first;
    const int CPR = E / VN;                        // 16-byte chunks per row
    const int RP = CPR >= 256 ? 1 : 256 / CPR;     // rows in flight side by side
    for (int c0 = 0; c0 < CPR; c0 += 256) {
        const int rs = RP > 1 ? (int)threadIdx.x / CPR : 0, c = RP > 1 ? (int)threadIdx.x % CPR : c0 + (int)threadIdx.x;
        float acc[VN];
#pragma unroll
        for (int q = 0; q < VN; q++) acc[q] = 0.f;
        if (c < CPR && rs < RP) {
            // eight rows per thread in flight: the row indices first, then all eight 16-byte row pieces, then the sums (one row at a
            // time the kernel is a chain of dependent loads -- index, row, index, row -- and read 2 TB/s)
            for (int r0 = rs; r0 < n; r0 += 8 * RP) {
                int tok[8];
#pragma unroll
                for (int u = 0; u < 8; u++) tok[u] = ord[min(r0 + u * RP, n - 1)];
                Vec16<T> row[8];
#pragma unroll
                for (int u = 0; u < 8; u++) row[u] = ld16(dh + (int64_t)tok[u] * E + c * VN);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (r0 + u * RP < n) {
                        const uint32_t hh = drop.thr ? drop_row_hash(drop, (uint32_t)tok[u]) : 0u;
#pragma unroll
                        for (int q = 0; q < VN; q++) {
                            float x0 = row[u].get(q);
                            if (drop.thr) x0 = apply_drop_rc(drop, hh, (uint32_t)(c * VN + q), x0);
                            acc[q] += x0;
                        }
                    }
                }
            }
        }
        if (RP > 1) {                              // fold the RP row groups through LDS
#pragma unroll
            for (int q = 0; q < VN; q++) red[threadIdx.x * VN + q] = acc[q];
            __syncthreads();
            if ((int)threadIdx.x < CPR) {
                for (int g = 1; g < RP; g++)
#pragma unroll
                    for (int q = 0; q < VN; q++) acc[q] += red[(g * CPR + threadIdx.x) * VN + q];
#pragma unroll
                for (int q = 0; q < VN; q++) atomicAdd(dwte + (int64_t)v * E + threadIdx.x * VN + q, acc[q]);
            }
            __syncthreads();
        } else if (c < CPR) {
#pragma unroll
            for (int q = 0; q < VN; q++) atomicAdd(dwte + (int64_t)v * E + c * VN + q, acc[q]);
        }
    }
}
// dwpe[pos0 + t, :] += sum_b dh[b, t, :]: one workgroup per position, RP batch rows side by side, no atomics
template <typename T>
__global__ __launch_bounds__(256) void embed_wpe_sum_kernel(const T* __restrict__ dh, float* __restrict__ dwpe, int B, int T_, int E, int pos0,
                                                            DropCfg drop) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float red[256 * VN];
    const int t = blockIdx.x;
    const int CPR = E / VN, RP = CPR >= 256 ? 1 : 256 / CPR;
    for (int c0 = 0; c0 < CPR; c0 += 256) {
        const int rs = RP > 1 ? (int)threadIdx.x / CPR : 0, c = RP > 1 ? (int)threadIdx.x % CPR : c0 + (int)threadIdx.x;
        float acc[VN];
#pragma unroll
        for (int q = 0; q < VN; q++) acc[q] = 0.f;
        if (c < CPR && rs < RP) {
            for (int b = rs; b < B; b += 2 * RP) {
                const int b1 = b + RP, tok0 = b * T_ + t, tok1 = (b1 < B ? b1 : b) * T_ + t;
                const Vec16<T> a = ld16(dh + (int64_t)tok0 * E + c * VN);
                const Vec16<T> bb = ld16(dh + (int64_t)tok1 * E + c * VN);
                const uint32_t h0 = drop_row_hash(drop, (uint32_t)tok0), h1 = drop_row_hash(drop, (uint32_t)tok1);
#pragma unroll
                for (int q = 0; q < VN; q++) {
                    float x0 = a.get(q), x1 = b1 < B ? bb.get(q) : 0.f;
                    if (drop.thr) {
                        x0 = apply_drop_rc(drop, h0, (uint32_t)(c * VN + q), x0);
                        x1 = apply_drop_rc(drop, h1, (uint32_t)(c * VN + q), x1);
                    }
                    acc[q] += x0 + x1;
                }
            }
        }
        if (RP > 1) {
#pragma unroll
            for (int q = 0; q < VN; q++) red[threadIdx.x * VN + q] = acc[q];
            __syncthreads();
            if ((int)threadIdx.x < CPR) {
                for (int g = 1; g < RP; g++)
#pragma unroll
                    for (int q = 0; q < VN; q++) acc[q] += red[(g * CPR + threadIdx.x) * VN + q];
#pragma unroll
                for (int q = 0; q < VN; q++) dwpe[(int64_t)(pos0 + t) * E + threadIdx.x * VN + q] += acc[q];
            }
            __syncthreads();
        } else if (c < CPR) {
#pragma unroll
            for (int q = 0; q < VN; q++) dwpe[(int64_t)(pos0 + t) * E + c * VN + q] += acc[q];
        }
    }
}

// =================================================================================================
// LayerNorm (Keras non-fused path: biased variance, eps inside rsqrt)   transformer.py:551,563,694
// one wave per row; a lane owns chunks (lane + 64*i) of 16 bytes.
// =================================================================================================
#define LN_MAXI 8
template <typename T, int MAXI>
__global__ void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, T* __restrict__ y,
                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int E,
                                     float eps) {
    constexpr int VN = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int chunks = E / VN;
    // the NEXT row of this wave is requested before the current one is reduced and written (two rows in flight per wave: with one,
    // the ~8 000 resident waves keep 8 MB in flight, which is what 5 TB/s needs -- the kernel sat at 0.62 of the HBM peak)
    const int stride = gridDim.x * wpb;
    int row = blockIdx.x * wpb + (threadIdx.x >> 6);
    Vec16<T> v[MAXI], nx[MAXI];
    auto fetch = [&](int r, Vec16<T> (&dst)[MAXI]) {
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            const int c = lane + 64 * i;
            if (c < chunks) dst[i] = ld16(x + (int64_t)r * E + c * VN);
        }
    };
    if (row < rows) fetch(row, v);
    for (; row < rows; row += stride) {
        const bool more = row + stride < rows;
        if (more) fetch(row + stride, nx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            int c = lane + 64 * i;
            if (c < chunks) {
#pragma unroll
                for (int j = 0; j < VN; j++) s += v[i].get(j);
            }
        }
        float mu = wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            int c = lane + 64 * i;
            if (c < chunks) {
#pragma unroll
                for (int j = 0; j < VN; j++) {
                    float d = v[i].get(j) - mu;
                    q += d * d;
                }
            }
        }
        float var = wave_sum(q) / (float)E;
        float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            int c = lane + 64 * i;
            if (c < chunks) {
                Vec16<T> o;
#pragma unroll
                for (int j = 0; j < VN; j++) {
                    int e = c * VN + j;
                    o.set(j, (v[i].get(j) - mu) * rs * gamma[e] + beta[e]);
                }
                st16(y + (int64_t)row * E + c * VN, o);
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < MAXI; i++) v[i] = nx[i];
        }
    }
}

// dx = resid + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma;  partial dgamma/dbeta per workgroup
// into ws[wg][2][E]; ln_param_reduce_kernel folds them into dgamma/dbeta.
// FUSED (the LayerNorm-fused block path, bf16): the forward pass never ran this LayerNorm as a kernel -- its statistics arrive as
// the partials the producing GEMM epilogue left (`part`, np segments of 256 columns, merged here), and its OUTPUT, which only
// the weight-gradient GEMM of the consuming Conv1D needs, is written here (`yout` = xhat * gamma + beta) beside dx.
// PRE (with mean / rstd arrays, round 6): dy holds rstd o gradient, like LnBwdFused::prescaled of the FUSED form
template <typename T, int MAXI, bool FUSED = false, bool PRE = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const T* __restrict__ resid,
                                     T* __restrict__ dx, float* __restrict__ ws, int rows, int E,
                                     T* __restrict__ dmask, int want_colsum, DropCfg drop,
                                     float* __restrict__ direct_g, float* __restrict__ direct_b, float* __restrict__ direct_cs,
                                     LnBwdFused fz = LnBwdFused()) {
    // optional fused consumer prologue: the output dx is the gradient of a residual branch's dropout output
    // (x + dropout(proj(..))): dmask = dx * mask/(1-p) feeds that projection's wgrad/dgrad and its column sums are the
    // projection's bias gradient (third partial, ws[wg][2] -> cs)
    constexpr int VN = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float ln_smem[];   // [waves][3][E]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int chunks = E / VN;
    float dg[MAXI][VN], db[MAXI][VN], gm[MAXI][VN], cs[MAXI][VN], be[FUSED ? MAXI : 1][VN];
#pragma unroll
    for (int i = 0; i < MAXI; i++) {
        int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < VN; j++) {
            dg[i][j] = 0.f;
            db[i][j] = 0.f;
            cs[i][j] = 0.f;
            gm[i][j] = (c < chunks) ? gamma[c * VN + j] : 0.f;
            if (FUSED) be[i][j] = (c < chunks) ? fz.beta[c * VN + j] : 0.f;
        }
    }
    constexpr int NPMAX = 2 * MAXI;             // segments of 256 columns in a row of at most 64 * MAXI chunks of 8
    // Software pipeline over the rows of this wave: the three input rows (dy, x, residual) and the statistics of the NEXT
    // row are requested before the current row is reduced and written, so ~6 KiB per wave (~96 KiB per CU at 4 waves
    // per SIMD) stay in flight.  (Before: dy/x, two wave reductions, THEN the residual row -- two exposed HBM latencies
    // per row and 3 KiB per wave in flight: 4.7 TB/s.)
    struct RowIn {
        Vec16<T> dy[MAXI], x[MAXI], r[MAXI];
        float mu, rs;
        f32x2 pt[FUSED ? NPMAX : 1];           // the row's partial statistics, merged when the row is reduced
    };
    auto fetch = [&](int row, RowIn& in) {
        if constexpr (FUSED) {
#pragma unroll
            for (int sg = 0; sg < NPMAX; sg++)
                in.pt[sg] = sg < fz.np ? *reinterpret_cast<const f32x2*>(fz.part + ((int64_t)row * fz.np + sg) * 2) : (f32x2){0.f, 0.f};
        } else {
            in.mu = mean[row];
            in.rs = rstd[row];
        }
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            const int c = lane + 64 * i;
            if (c < chunks) {
                in.dy[i] = ld16(dy + (int64_t)row * E + c * VN);
                in.x[i] = ld16(x + (int64_t)row * E + c * VN);
                if (resid) in.r[i] = ld16(resid + (int64_t)row * E + c * VN);
            }
        }
    };
    const int stride = gridDim.x * wpb;
    int row = blockIdx.x * wpb + wave;
    RowIn cur, nxt;
    if (row < rows) fetch(row, cur);
    for (; row < rows; row += stride) {
        const bool more = row + stride < rows;
        if (more) fetch(row + stride, nxt);
        float mu = cur.mu, rs = cur.rs;
        if constexpr (FUSED) {
            float a = 0.f;
#pragma unroll
            for (int sg = 0; sg < NPMAX; sg++) a += cur.pt[sg][0];                 // (absent segments hold zeros)
            mu = a / (float)fz.np;
            float m2 = 0.f;
#pragma unroll
            for (int sg = 0; sg < NPMAX; sg++) {
                const float d = cur.pt[sg][0] - mu;
                if (sg < fz.np) m2 += cur.pt[sg][1] + 256.0f * d * d;
            }
            rs = __builtin_amdgcn_rsqf(m2 / (256.0f * (float)fz.np) + fz.eps);
        }
        // prescaled: dy holds rstd o gradient (the A operand of the dgrad GEMM in front was stored rstd-scaled, model.hip: backward);
        // the row's 1 / rstd gives the gradient back -- everything below is the formula of the plain form
        float dsc = 1.0f;
        if constexpr (FUSED) { if (fz.prescaled) dsc = 1.0f / rs; }
        if constexpr (PRE) dsc = 1.0f / rs;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            int c = lane + 64 * i;
            if (c < chunks) {
#pragma unroll
                for (int j = 0; j < VN; j++) {
                    float d = cur.dy[i].get(j) * dsc;
                    float xh = (cur.x[i].get(j) - mu) * rs;
                    float g = d * gm[i][j];
                    s1 += g;
                    s2 += g * xh;
                    dg[i][j] += d * xh;
                    db[i][j] += d;
                }
            }
        }
        s1 = wave_sum(s1) / (float)E;
        s2 = wave_sum(s2) / (float)E;
        const uint32_t rowh = drop_row_hash(drop, (uint32_t)row);        // one full hash per row
#pragma unroll
        for (int i = 0; i < MAXI; i++) {
            int c = lane + 64 * i;
            if (c < chunks) {
                Vec16<T> o, om, oy;
#pragma unroll
                for (int j = 0; j < VN; j++) {
                    float xh = (cur.x[i].get(j) - mu) * rs;
                    float g = cur.dy[i].get(j) * dsc * gm[i][j];
                    float v = rs * (g - s1 - xh * s2);
                    if (resid) v += cur.r[i].get(j);
                    o.set(j, v);
                    if (FUSED) oy.set(j, xh * gm[i][j] + be[i][j]);
                    if (want_colsum || dmask) {
                        // the consumer sees the STORED (rounded) value
                        float vm = drop.thr ? apply_drop_rc(drop, rowh, (uint32_t)(c * VN + j), o.get(j)) : o.get(j);
                        om.set(j, vm);
                        cs[i][j] += om.get(j);
                    }
                }
                st16(dx + (int64_t)row * E + c * VN, o);
                if (dmask) st16(dmask + (int64_t)row * E + c * VN, om);
                if constexpr (FUSED) { if (fz.yout) st16((T*)fz.yout + (int64_t)row * E + c * VN, oy); }
            }
        }
        if (more) cur = nxt;
    }
    // cross-wave reduction of the parameter-gradient partials
    float* sm = ln_smem + (size_t)wave * 3 * E;
#pragma unroll
    for (int i = 0; i < MAXI; i++) {
        int c = lane + 64 * i;
        if (c < chunks) {
#pragma unroll
            for (int j = 0; j < VN; j++) {
                sm[c * VN + j] = dg[i][j];
                sm[E + c * VN + j] = db[i][j];
                sm[2 * E + c * VN + j] = cs[i][j];
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * E; e += blockDim.x) {
        float a = 0.f;
        for (int w = 0; w < wpb; w++) a += ln_smem[(size_t)w * 3 * E + e];
        if (direct_g) {
            // few workgroups (small batches): straight into the gradients, no second launch
            if (e < E) atomicAdd(direct_g + e, a);
            else if (e < 2 * E) atomicAdd(direct_b + (e - E), a);
            else if (direct_cs) atomicAdd(direct_cs + (e - 2 * E), a);
        } else {
            ws[(size_t)blockIdx.x * 3 * E + e] = a;
        }
    }
}

// grid (ceil(2E/256), slices): each thread folds nparts/slices partials, then one atomic per slice
__global__ void ln_param_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ colsum, int nparts, int E) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (colsum ? 3 : 2) * E) return;
    float a = 0.f;
    const int stride = gridDim.y;
    int p = blockIdx.y;
    for (; p + 7 * stride < nparts; p += 8 * stride) {       // 8 independent loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = ws[(size_t)(p + u * stride) * 3 * E + e];
#pragma unroll
        for (int u = 0; u < 8; u++) a += v[u];
    }
    for (; p < nparts; p += stride) a += ws[(size_t)p * 3 * E + e];
    if (e < E) atomicAdd(dgamma + e, a);
    else if (e < 2 * E) atomicAdd(dbeta + (e - E), a);
    else atomicAdd(colsum + (e - 2 * E), a);
}

// =================================================================================================
// softmax cross-entropy forward + backward in one pass     transformer.py:888,918,924-926
// one wave per row.  loss_row = logsumexp(z) - z[y];  dz = (softmax(z) - onehot(y)) * inv_n
// =================================================================================================
#define XENT_MAXI 8   // register-resident rows: vocab <= 512 columns (incl. padding); wider rows take softmax_xent_wide_kernel
template <typename T>
__global__ void softmax_xent_kernel(const float* __restrict__ z, int ldz, const int32_t* __restrict__ y,
                                    T* __restrict__ dz, float* __restrict__ row_loss,
                                    int32_t* __restrict__ row_correct, int rows, int V, float inv_n) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const float* zr = z + (int64_t)row * ldz;
        float v[XENT_MAXI];
        float mx = -INFINITY;
        int arg = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < XENT_MAXI; i++) {
            int c = lane + 64 * i;
            v[i] = (c < V) ? zr[c] : -INFINITY;
            if (v[i] > mx) {   // strict > keeps the lowest index within the lane (columns ascend with i)
                mx = v[i];
                arg = c;
            }
        }
        // wave argmax: larger value wins, ties -> lower index (tf.argmax)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float om = __shfl_xor(mx, o);
            int oa = __shfl_xor(arg, o);
            if (om > mx || (om == mx && oa < arg)) {
                mx = om;
                arg = oa;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < XENT_MAXI; i++) {
            int c = lane + 64 * i;
            if (c < V) s += expf(v[i] - mx);
        }
        s = wave_sum(s);
        float lse = mx + logf(s);
        int yy = y[row];
        if (dz) {
            T* dr = dz + (int64_t)row * ldz;
#pragma unroll
            for (int i = 0; i < XENT_MAXI; i++) {
                int c = lane + 64 * i;
                if (c < ldz) {
                    float g = 0.f;
                    if (c < V) g = (expf(v[i] - lse) - (c == yy ? 1.0f : 0.0f)) * inv_n;
                    dr[c] = from_f32<T>(g);
                }
            }
        }
        // z[y] lives in lane y&63, slot y>>6; select with static indices (a runtime-indexed v[] would go to scratch)
        float zy2 = 0.f;
#pragma unroll
        for (int i = 0; i < XENT_MAXI; i++) {
            float cand = __shfl(v[i], yy & 63);
            if (i == (yy >> 6)) zy2 = cand;
        }
        if (lane == 0) {
            row_loss[row] = lse - zy2;
            row_correct[row] = (arg == yy) ? 1 : 0;
        }
    }
}

// The same for rows of at most 512 columns whose stride is a multiple of 8: a lane owns EIGHT CONSECUTIVE columns -- two 16-byte
// loads and one 16-byte (bf16) gradient store per row and lane where the kernel above issues seven 4-byte loads and seven
// 2-byte stores; one exponential per element (the gradient is e * (1 / sum)); the next row is requested before the current one
// is reduced.  Ties: the lowest column wins (ascending inside a lane, lower lane = lower columns).
template <typename T>
__global__ __launch_bounds__(256) void softmax_xent8_kernel(const float* __restrict__ z, int ldz, const int32_t* __restrict__ y,
                                                            T* __restrict__ dz, float* __restrict__ row_loss,
                                                            int32_t* __restrict__ row_correct, int rows, int V, float inv_n) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int c0 = 8 * lane;
    const bool active = c0 < ldz;
    const int stride = gridDim.x * wpb;
    int row = blockIdx.x * wpb + (threadIdx.x >> 6);
    f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0;
    int ny = 0;
    auto fetch = [&](int r) {
        if (active) {
            n0 = *reinterpret_cast<const f32x4*>(z + (int64_t)r * ldz + c0);
            n1 = *reinterpret_cast<const f32x4*>(z + (int64_t)r * ldz + c0 + 4);
        }
        ny = y[r];
    };
    if (row < rows) fetch(row);
    for (; row < rows; row += stride) {
        float v[8] = {n0[0], n0[1], n0[2], n0[3], n1[0], n1[1], n1[2], n1[3]};
        const int yy = ny;
        if (row + stride < rows) fetch(row + stride);
        float mx = -INFINITY;
        int arg = 0x7fffff;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!active || c0 + j >= V) v[j] = -INFINITY;
            if (v[j] > mx) {       // strict >: the lowest column of the lane
                mx = v[j];
                arg = c0 + j;
            }
        }
        const float gmx = wave_max(mx);
        // lowest column among the lanes that hold the maximum (column numbers are exact in fp32)
        const int garg = (int)(-wave_max(mx == gmx ? -(float)arg : -8388607.0f));
        float e[8], ssum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            e[j] = expf(v[j] - gmx);                 // exp(-inf) = 0 for the padding columns
            ssum += e[j];
        }
        ssum = wave_sum(ssum);
        const float lse = gmx + logf(ssum);
        if (dz && active) {
            const float inv = 1.0f / ssum;
            Vec16<T> o0, o1;                          // fp32: two 16-byte stores; bf16: one
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; j++) g[j] = (c0 + j < V) ? (e[j] * inv - (c0 + j == yy ? 1.0f : 0.0f)) * inv_n : 0.f;
            T* dr = dz + (int64_t)row * ldz + c0;
            if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                for (int j = 0; j < 4; j++) { o0.set(j, g[j]); o1.set(j, g[4 + j]); }
                st16(dr, o0);
                st16(dr + 4, o1);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) o0.set(j, g[j]);
                st16(dr, o0);
            }
        }
        float zy = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j == (yy & 7)) zy = v[j];
        zy = __shfl(zy, yy >> 3);
        if (lane == 0) {
            row_loss[row] = lse - zy;
            row_correct[row] = (garg == yy) ? 1 : 0;
        }
    }
}

// the same for any vocabulary size (ldz > 64 * XENT_MAXI): three passes over the row, which stays in L1/L2 between them
template <typename T>
__global__ void softmax_xent_wide_kernel(const float* __restrict__ z, int ldz, const int32_t* __restrict__ y,
                                         T* __restrict__ dz, float* __restrict__ row_loss,
                                         int32_t* __restrict__ row_correct, int rows, int V, float inv_n) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const float* zr = z + (int64_t)row * ldz;
        float mx = -INFINITY;
        int arg = 0x7fffffff;
        for (int c = lane; c < V; c += 64) {
            const float v = zr[c];
            if (v > mx) {      // strict >: the lowest index within the lane (columns ascend)
                mx = v;
                arg = c;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float om = __shfl_xor(mx, o);
            int oa = __shfl_xor(arg, o);
            if (om > mx || (om == mx && oa < arg)) {
                mx = om;
                arg = oa;
            }
        }
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += expf(zr[c] - mx);
        s = wave_sum(s);
        const float lse = mx + logf(s);
        const int yy = y[row];
        if (dz) {
            T* dr = dz + (int64_t)row * ldz;
            for (int c = lane; c < ldz; c += 64) {
                float g = 0.f;
                if (c < V) g = (expf(zr[c] - lse) - (c == yy ? 1.0f : 0.0f)) * inv_n;
                dr[c] = from_f32<T>(g);
            }
        }
        if (lane == 0) {
            row_loss[row] = lse - zr[yy];
            row_correct[row] = (arg == yy) ? 1 : 0;
        }
    }
}

// deterministic single-workgroup reduction of the per-row losses (fixed summation order: 1024 threads, 16-byte loads,
// four independent partial sums per thread; the 256-thread scalar loop it replaces took 142 us at 131072 rows)
__global__ __launch_bounds__(1024) void metrics_reduce_kernel(const float* __restrict__ row_loss, const int32_t* __restrict__ row_correct,
                                                               int rows, Metrics* __restrict__ out) {
    __shared__ double sl[1024];
    __shared__ long long sc[1024];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = 0;
    const int rows4 = rows >> 2;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int i = threadIdx.x; i < rows4; i += 1024) {
        const f32x4 l = reinterpret_cast<const f32x4*>(row_loss)[i];
        const i32x4 k = reinterpret_cast<const i32x4*>(row_correct)[i];
        a0 += (double)l[0];
        a1 += (double)l[1];
        a2 += (double)l[2];
        a3 += (double)l[3];
        c += k[0] + k[1] + k[2] + k[3];
    }
    for (int i = (rows4 << 2) + threadIdx.x; i < rows; i += 1024) {
        a0 += (double)row_loss[i];
        c += row_correct[i];
    }
    sl[threadIdx.x] = (a0 + a1) + (a2 + a3);
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sl[threadIdx.x] += sl[threadIdx.x + s];
            sc[threadIdx.x] += sc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->loss_sum = sl[0];
        out->correct = sc[0];
        out->loss_mean = (float)(sl[0] / (double)rows);
        out->acc = (float)((double)sc[0] / (double)rows);
    }
}

// =================================================================================================
// Adam, Keras OptimizerV2 formulation (transformer.py:887,921):
//   m += (g-m)(1-b1); v += (g^2-v)(1-b2); theta -= alpha*m/(sqrt(v)+eps), alpha = lr*sqrt(1-b2^t)/(1-b1^t)
// one flat pass over all parameters: reads theta,g,m,v (16 B/param) writes theta,m,v (12) [+2 bf16 shadow]
// =================================================================================================
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, bf16_t* __restrict__ shadow, int64_t n4, int64_t n,
                            float alpha, float beta1, float beta2, float eps, float gscale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float gg = gv[j] * gscale;
            mv[j] = beta1 * mv[j] + (1.0f - beta1) * gg;
            vv[j] = beta2 * vv[j] + (1.0f - beta2) * gg * gg;
            pv[j] = pv[j] - alpha * mv[j] / (sqrtf(vv[j]) + eps);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (shadow) {
            bf16x4 s;
#pragma unroll
            for (int j = 0; j < 4; j++) s[j] = (bf16_t)pv[j];
            reinterpret_cast<bf16x4*>(shadow)[i] = s;
        }
    }
}

__global__ void cast_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = reinterpret_cast<const f32x4*>(in)[i];
        bf16x4 s;
#pragma unroll
        for (int j = 0; j < 4; j++) s[j] = (bf16_t)a[j];
        reinterpret_cast<bf16x4*>(out)[i] = s;
    }
}

// =================================================================================================
// column sums (bias gradients): out[c] += sum_r X[r, c]
// grid (ceil(cols/(64*VN)), row splits); a lane owns one 16-byte chunk of columns, the 4 waves take rows
// r, r+4, ... (4 loads in flight per lane); cross-wave fold through LDS, one f32 atomic per column per split.
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int ldx, float* __restrict__ out, int rows, int cols,
                                                     float* __restrict__ part) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float sm[4][64 * VN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * VN;
    const int rows_per = cdiv(rows, (int)gridDim.y);
    const int r0 = blockIdx.y * rows_per;
    const int r1 = min(rows, r0 + rows_per);
    float acc[VN];
#pragma unroll
    for (int j = 0; j < VN; j++) acc[j] = 0.f;
    if (c0 < cols) {
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            Vec16<T> v0 = ld16(X + (int64_t)r * ldx + c0), v1 = ld16(X + (int64_t)(r + 4) * ldx + c0);
            Vec16<T> v2 = ld16(X + (int64_t)(r + 8) * ldx + c0), v3 = ld16(X + (int64_t)(r + 12) * ldx + c0);
#pragma unroll
            for (int j = 0; j < VN; j++) acc[j] += (v0.get(j) + v1.get(j)) + (v2.get(j) + v3.get(j));
        }
        for (; r < r1; r += 4) {
            Vec16<T> v0 = ld16(X + (int64_t)r * ldx + c0);
#pragma unroll
            for (int j = 0; j < VN; j++) acc[j] += v0.get(j);
        }
    }
#pragma unroll
    for (int j = 0; j < VN; j++) sm[wave][lane * VN + j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * VN; i += 256) {
        int c = blockIdx.x * 64 * VN + i;
        const float v = sm[0][i] + sm[1][i] + sm[2][i] + sm[3][i];
        if (c < cols) {
            if (part) part[(size_t)blockIdx.y * cols + c] = v;       // deterministic mode: fixed-order fold afterwards
            else atomicAdd(out + c, v);
        }
    }
}
__global__ void colsum_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int splits, int cols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float a = 0.f;
    for (int sI = 0; sI < splits; sI++) a += part[(size_t)sI * cols + c];
    out[c] += a;
}

// =================================================================================================
// host launchers (C ABI)
// =================================================================================================
extern "C" int cmp_k_embed_fwd(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out,
                               int B, int T, int E, int pos0, int dtype, float p_drop, uint64_t seed,
                               uint32_t rng_stream) {
    return embed_fwd_run(stream, ids, wte, wpe, out, B, T, E, pos0, dtype, p_drop, seed, rng_stream, nullptr, nullptr);
}

int embed_fwd_run(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out, int B, int T, int E, int pos0,
                  int dtype, float p_drop, uint64_t seed, uint32_t rng_stream, const int32_t* pos_ids, const int32_t* type_ids) {
    CMP_REQUIRE(E % 8 == 0, "embed_fwd: E=%d must be a multiple of 8", E);
    hipStream_t s = (hipStream_t)stream;
    DropCfg d = make_drop(p_drop, seed, rng_stream);
    int ntok = B * T;
    if (ntok == 0) return CMP_OK;
    if (dtype == CMP_BF16) {
        int64_t total = (int64_t)ntok * (E / 8);
        int grid = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
        embed_fwd_kernel<bf16_t><<<grid, 256, 0, s>>>(ids, wte, wpe, (bf16_t*)out, ntok, T, E, pos0, d, pos_ids, type_ids);
    } else {
        int64_t total = (int64_t)ntok * (E / 4);
        int grid = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
        embed_fwd_kernel<float><<<grid, 256, 0, s>>>(ids, wte, wpe, (float*)out, ntok, T, E, pos0, d, pos_ids, type_ids);
    }
    KERNEL_CHECK();
    return CMP_OK;
}

int embed_fwd_stats_run(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out, float* part, int B, int T,
                        int E, int pos0, float p_drop, uint64_t seed, uint32_t rng_stream) {
    CMP_REQUIRE(E % 256 == 0 && E <= 2048, "embed_fwd_stats: E=%d must be a multiple of 256, at most 2048", E);
    hipStream_t s = (hipStream_t)stream;
    DropCfg d = make_drop(p_drop, seed, rng_stream);
    const int ntok = B * T;
    if (ntok == 0) return CMP_OK;
    const int grid = std::min(cdiv(ntok, 4), 8192), maxi = cdiv(E / 8, 64);
#define EMB_ST(MI) embed_fwd_stats_kernel<MI><<<grid, 256, 0, s>>>(ids, wte, wpe, (bf16_t*)out, part, ntok, T, E, pos0, d)
    if (maxi == 1) EMB_ST(1); else if (maxi == 2) EMB_ST(2); else EMB_ST(4);
#undef EMB_ST
    KERNEL_CHECK();
    return CMP_OK;
}
extern "C" int cmp_k_embed_fwd_stats(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out, float* part,
                                     int B, int T, int E, int pos0, float p_drop, uint64_t seed, uint32_t rng_stream) {
    return embed_fwd_stats_run(stream, ids, wte, wpe, out, part, B, T, E, pos0, p_drop, seed, rng_stream);
}

// ---- LayerNorm fold, weight side (common.h: LnEpi).  For every Conv1D that consumes a LayerNorm output (c_attn <- ln_1, c_fc <-
// ln_2; transformer.py:583-584,591) one pass over the fp32 master weight W [rows = E][cols = N] writes
//   WT[n][k] = bf16(gamma[k] * W[k][n])      the K-contiguous operand of the forward GEMM on the RAW rows
//   cs[n]    = sum_k float(WT[n][k])         (of the ROUNDED values: what the matrix cores multiply by)
//   bias'[n] = bias[n] + sum_k beta[k] * W[k][n]
// A workgroup owns 32 columns and walks the rows in 32x32 tiles through LDS (transposed store); no atomics: reproducible.
__global__ __launch_bounds__(256) void ln_fold_prep_kernel(const float* __restrict__ P, bf16_t* __restrict__ ST, float* __restrict__ fold,
                                                           const FoldDesc* __restrict__ desc) {
    __shared__ bf16_t tile[32][34];
    __shared__ float red[2][8][33];
    const FoldDesc d = desc[blockIdx.y];
    const int c0 = blockIdx.x * 32;
    if (c0 >= d.cols) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const float* __restrict__ W = P + d.w_off;
    const float* __restrict__ g = P + d.g_off;
    const float* __restrict__ be = P + d.be_off;
    bf16_t* __restrict__ dst = ST + d.w_off;
    float acs = 0.f, abb = 0.f;
    for (int r0 = 0; r0 < d.rows; r0 += 32) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + ty + 8 * j;
            const float w = W[(int64_t)r * d.cols + c0 + tx];
            const bf16_t ws = (bf16_t)(g[r] * w);
            acs += (float)ws;
            abb = fmaf(be[r], w, abb);
            tile[ty + 8 * j][tx] = ws;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) dst[(int64_t)(c0 + ty + 8 * j) * d.rows + r0 + tx] = tile[tx][ty + 8 * j];
        __syncthreads();
    }
    red[0][ty][tx] = acs;
    red[1][ty][tx] = abb;
    __syncthreads();
    if (ty == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) { a += red[0][k][tx]; b += red[1][k][tx]; }
        fold[d.out_off + c0 + tx] = a;
        fold[d.out_off + d.cols + c0 + tx] = P[d.b_off + c0 + tx] + b;
    }
}
// ... and for ln_f in front of the tied-logits matmul (transformer.py:811, 818; the weight is wte itself, [V][E], K-contiguous as it
// is): one wave per vocabulary row v writes out[v][k] = bf16(gamma[k] * wte[v][k]), cs[v] = sum of the rounded row, bias'[v] =
// sum_k beta[k] * wte[v][k].  cs / bias' hold `npad` entries (the tile columns beyond V read zeros).
__global__ __launch_bounds__(256) void lnf_fold_prep_kernel(const float* __restrict__ wte, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ out, float* __restrict__ cs,
                                                            float* __restrict__ bias, int V, int E, int npad) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int v = blockIdx.x * wpb + (threadIdx.x >> 6); v < npad; v += gridDim.x * wpb) {
        float a = 0.f, b = 0.f;
        if (v < V) {
            for (int k = lane * 4; k < E; k += 256) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wte + (int64_t)v * E + k);
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + k), be = *reinterpret_cast<const f32x4*>(beta + k);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    o[j] = (bf16_t)(g[j] * w[j]);
                    a += (float)o[j];
                    b = fmaf(be[j], w[j], b);
                }
                *reinterpret_cast<bf16x4*>(out + (int64_t)v * E + k) = o;
            }
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (lane == 0) { cs[v] = a; bias[v] = b; }
    }
}
int lnf_fold_prep_run(void* stream, const float* wte, const float* gamma, const float* beta, void* out, float* cs, float* bias, int V, int E,
                      int npad) {
    CMP_REQUIRE(E % 4 == 0 && npad >= V, "lnf_fold_prep: E=%d npad=%d", E, npad);
    lnf_fold_prep_kernel<<<cdiv(npad, 4), 256, 0, (hipStream_t)stream>>>(wte, gamma, beta, (bf16_t*)out, cs, bias, V, E, npad);
    KERNEL_CHECK();
    return CMP_OK;
}
int ln_fold_prep_run(void* stream, const float* P, void* ST, float* fold, const void* desc_dev, int ndesc, int max_cols) {
    if (ndesc <= 0) return CMP_OK;
    ln_fold_prep_kernel<<<dim3(cdiv(max_cols, 32), ndesc), 256, 0, (hipStream_t)stream>>>(P, (bf16_t*)ST, fold, (const FoldDesc*)desc_dev);
    KERNEL_CHECK();
    return CMP_OK;
}
// kernel-level entry point (tests): ONE matrix, operands given separately.  W fp32 [E, N], bias [N], gamma / beta [E] ->
// WT bf16 [N, E], cs [N], bias_out [N].  E and N multiples of 32.
extern "C" int cmp_k_ln_fold_prep(void* stream, const float* W, const float* bias, const float* gamma, const float* beta, void* WT,
                                  float* cs, float* bias_out, int E, int N) {
    CMP_REQUIRE(E > 0 && N > 0 && E % 32 == 0 && N % 32 == 0, "ln_fold_prep: E=%d N=%d must be multiples of 32", E, N);
    // one flat staging buffer laid out like the model's (P: W | bias | gamma | beta; fold: cs | bias'), copied in and out on the stream
    hipStream_t s = (hipStream_t)stream;
    const int64_t nW = (int64_t)E * N;
    float* Pb = nullptr; float* fold = nullptr; FoldDesc* dd = nullptr;
    HIP_CHECK(hipMalloc((void**)&Pb, (size_t)(nW + N + 2 * E) * 4));
    hipError_t e = hipMalloc((void**)&fold, (size_t)2 * N * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dd, sizeof(FoldDesc));
    FoldDesc h{0, nW, nW + N, nW + N + E, 0, E, N};
    if (e == hipSuccess) e = hipMemcpyAsync(Pb, W, (size_t)nW * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(Pb + nW, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(Pb + nW + N, gamma, (size_t)E * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(Pb + nW + N + E, beta, (size_t)E * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dd, &h, sizeof(h), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int rc = CMP_OK;
    if (e == hipSuccess) {
        // (the kernel writes ST + w_off with w_off = 0: WT itself)
        rc = ln_fold_prep_run(stream, Pb, WT, fold, dd, 1, N);
        if (rc == CMP_OK) e = hipMemcpyAsync(cs, fold, (size_t)N * 4, hipMemcpyDeviceToDevice, s);
        if (rc == CMP_OK && e == hipSuccess) e = hipMemcpyAsync(bias_out, fold + N, (size_t)N * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(Pb); (void)hipFree(fold); (void)hipFree(dd);
    if (rc != CMP_OK) return rc;
    HIP_CHECK(e);
    return CMP_OK;
}

// int32 words of the sorted form's workspace for ntok tokens and V ids (0: the vocabulary does not fit the LDS histogram)
int64_t embed_bwd_sort_ws_words(int64_t ntok, int V) {
    if (V <= 0 || V > 8192) return 0;
    return 16 + 2ll * EMB_NB * V + V + 2ll * (V + 1) + (ntok / EMB_CH + V + 1) + ntok + 16;
}

extern "C" int cmp_k_embed_bwd_v(void* stream, const int32_t* ids, const void* dh, float* dwte, float* dwpe,
                                 int B, int T, int E, int pos0, int dtype, float p_drop, uint64_t seed,
                                 uint32_t rng_stream, int V) {
    // the sorted form with a workspace of its own for this call (kernel-level tests; the model owns one)
    const int64_t words = embed_bwd_sort_ws_words((int64_t)B * T, V);
    int* ws = nullptr;
    if (words > 0) HIP_CHECK(hipMalloc((void**)&ws, (size_t)words * 4));
    const int rc = embed_bwd_run(stream, ids, dh, dwte, dwpe, B, T, E, pos0, dtype, p_drop, seed, rng_stream, 0, nullptr, 0, V, ws, words);
    if (ws) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(ws); }
    return rc;
}
extern "C" int cmp_k_embed_bwd(void* stream, const int32_t* ids, const void* dh, float* dwte, float* dwpe,
                               int B, int T, int E, int pos0, int dtype, float p_drop, uint64_t seed,
                               uint32_t rng_stream) {
    return embed_bwd_run(stream, ids, dh, dwte, dwpe, B, T, E, pos0, dtype, p_drop, seed, rng_stream, 0, nullptr, 0, 0, nullptr, 0);
}

// V > 0 with a workspace of V * EMB_SEG * E floats selects the atomic-free (bitwise reproducible) form; sort_V > 0 with a
// workspace of embed_bwd_sort_ws_words() int32 words the sorted form; otherwise one f32 atomic per gradient element
int embed_bwd_run(void* stream, const int32_t* ids, const void* dh, float* dwte, float* dwpe, int B, int T, int E, int pos0,
                  int dtype, float p_drop, uint64_t seed, uint32_t rng_stream, int V, float* det_ws, size_t det_ws_bytes,
                  int sort_V, int* sort_ws, int64_t sort_ws_words) {
    hipStream_t s = (hipStream_t)stream;
    DropCfg d = make_drop(p_drop, seed, rng_stream);
    if (B * T == 0) return CMP_OK;
    if (V > 0 && det_ws) {
        CMP_REQUIRE((size_t)V * EMB_SEG * E * 4 <= det_ws_bytes, "embed_bwd: deterministic workspace too small");
        dim3 g1(V, EMB_SEG);
        const int g2 = (int)cdiv64((int64_t)V * E, 256), g3 = (int)cdiv64((int64_t)T * E, 256);
        if (dtype == CMP_BF16) {
            embed_bwd_det_part_kernel<bf16_t><<<g1, 256, 0, s>>>(ids, (const bf16_t*)dh, det_ws, B * T, E, d);
            embed_bwd_wpe_kernel<bf16_t><<<g3, 256, 0, s>>>((const bf16_t*)dh, dwpe, B, T, E, pos0, d);
        } else {
            embed_bwd_det_part_kernel<float><<<g1, 256, 0, s>>>(ids, (const float*)dh, det_ws, B * T, E, d);
            embed_bwd_wpe_kernel<float><<<g3, 256, 0, s>>>((const float*)dh, dwpe, B, T, E, pos0, d);
        }
        embed_bwd_det_reduce_kernel<<<g2, 256, 0, s>>>(det_ws, dwte, V, E);
        KERNEL_CHECK();
        return CMP_OK;
    }
    const int64_t ntok = (int64_t)B * T;
    const int vn = dtype == CMP_BF16 ? 8 : 4;
    if (sort_V > 0 && sort_ws && embed_bwd_sort_ws_words(ntok, sort_V) > 0 && sort_ws_words >= embed_bwd_sort_ws_words(ntok, sort_V) &&
        ntok >= 4096 && E % vn == 0 && ntok < (1ll << 31)) {
        const int Vv = sort_V, NB = EMB_NB;
        int* nchunks = sort_ws;
        int* hist = sort_ws + 16;                       // 16-byte aligned rows of EMB_NB counts
        int* base = hist + (size_t)NB * Vv;
        int* total = base + (size_t)NB * Vv;
        int* start = total + Vv;
        int* cstart = start + Vv + 1;
        int* chunk_id = cstart + Vv + 1;
        const int maxchunks = (int)(ntok / EMB_CH) + Vv + 1;
        int* order = chunk_id + maxchunks;
        const size_t lds = (size_t)Vv * 4;
        embed_hist_kernel<<<NB, 256, lds, s>>>(ids, hist, (int)ntok, Vv);
        embed_scan1_kernel<<<cdiv(Vv, 256), 256, 0, s>>>(hist, base, total, Vv);
        embed_scan2_kernel<<<1, 1024, 0, s>>>(total, start, cstart, nchunks, chunk_id, Vv);
        embed_scatter_kernel<<<NB, 256, lds, s>>>(ids, base, start, order, (int)ntok, Vv);
        if (dtype == CMP_BF16) {
            embed_segsum_kernel<bf16_t><<<maxchunks, 256, 0, s>>>((const bf16_t*)dh, order, total, start, cstart, nchunks, chunk_id, dwte, Vv, E, d);
            embed_wpe_sum_kernel<bf16_t><<<T, 256, 0, s>>>((const bf16_t*)dh, dwpe, B, T, E, pos0, d);
        } else {
            embed_segsum_kernel<float><<<maxchunks, 256, 0, s>>>((const float*)dh, order, total, start, cstart, nchunks, chunk_id, dwte, Vv, E, d);
            embed_wpe_sum_kernel<float><<<T, 256, 0, s>>>((const float*)dh, dwpe, B, T, E, pos0, d);
        }
        KERNEL_CHECK();
        return CMP_OK;
    }
    int grid = (int)cdiv64((int64_t)T * E, 256);
    if (dtype == CMP_BF16)
        embed_bwd_kernel<bf16_t><<<grid, 256, 0, s>>>(ids, (const bf16_t*)dh, dwte, dwpe, B, T, E, pos0, d);
    else
        embed_bwd_kernel<float><<<grid, 256, 0, s>>>(ids, (const float*)dh, dwte, dwpe, B, T, E, pos0, d);
    KERNEL_CHECK();
    return CMP_OK;
}

static int ln_check(int E, int dtype) {
    int vn = dtype == CMP_BF16 ? 8 : 4;
    CMP_REQUIRE(E % vn == 0 && E / vn <= 64 * LN_MAXI && E <= 2048, "layernorm: E=%d unsupported for dtype %d (a multiple of %d, at most 2048)",
                E, dtype, vn);
    return CMP_OK;
}

extern "C" int cmp_k_layernorm_fwd(void* stream, const void* x, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, int rows, int E, float eps, int dtype) {
    int rc = ln_check(E, dtype);
    if (rc) return rc;
    if (rows == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = std::min(cdiv(rows, 4), 8192);
    const int vn = dtype == CMP_BF16 ? 8 : 4;
    const int maxi = cdiv(E / vn, 64);          // 16-byte chunks per lane: 1 for E <= 512 (bf16) / 256 (fp32)
    PROF_START(6, s);
#define LN_FWD(TT, MI) layernorm_fwd_kernel<TT, MI><<<grid, 256, 0, s>>>((const TT*)x, gamma, beta, (TT*)y, mean, rstd, rows, E, eps)
    if (dtype == CMP_BF16) { if (maxi == 1) LN_FWD(bf16_t, 1); else if (maxi == 2) LN_FWD(bf16_t, 2); else if (maxi <= 4) LN_FWD(bf16_t, 4); else LN_FWD(bf16_t, 8); }
    else { if (maxi == 1) LN_FWD(float, 1); else if (maxi == 2) LN_FWD(float, 2); else if (maxi <= 4) LN_FWD(float, 4); else LN_FWD(float, 8); }
#undef LN_FWD
    PROF_STOP(6, s, (double)rows * (2.0 * E * dtype_size(dtype) + 8.0), (double)rows * (2.0 * E * dtype_size(dtype) + 8.0));
    KERNEL_CHECK();
    return CMP_OK;
}

// workgroups of the backward launch: 8 rows per workgroup up to a cap.  Every workgroup leaves 3E partial sums for
// ln_param_reduce_kernel, so the cap also sets that traffic: same-box whole-step A/B, cap 512 / 1024 / 2048 / 4096:
// 7.40 / 7.38 / 7.45 / 7.57 ms at 32768 rows (C2, B=32), 26.70 / 26.54 / 26.54 / 26.69 ms at 131072 rows.
static int ln_bwd_grid(int rows) {
    int cap = 1024;
    return std::max(1, std::min(cdiv(rows, 8), cap));
}

extern "C" int64_t cmp_k_layernorm_bwd_ws(int rows, int E) { return (int64_t)ln_bwd_grid(rows) * 3 * E * sizeof(float); }

extern "C" int cmp_k_layernorm_bwd(void* stream, const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, const void* resid, void* dx, float* dgamma, float* dbeta,
                                   void* ws, int rows, int E, int dtype) {
    return cmp_k_layernorm_bwd_fused(stream, dy, x, gamma, mean, rstd, resid, dx, dgamma, dbeta, ws, rows, E, dtype,
                                     nullptr, nullptr, 0.f, 0, 0);
}

extern "C" int cmp_k_layernorm_bwd_fused(void* stream, const void* dy, const void* x, const float* gamma, const float* mean,
                                         const float* rstd, const void* resid, void* dx, float* dgamma, float* dbeta,
                                         void* ws, int rows, int E, int dtype, void* dmask, float* colsum, float p_drop,
                                         uint64_t seed, uint32_t rng_stream) {
    return layernorm_bwd_run(stream, dy, x, gamma, mean, rstd, resid, dx, dgamma, dbeta, ws, rows, E, dtype, dmask, colsum, p_drop,
                             seed, rng_stream, false, nullptr, false);
}
// The form the LayerNorm-fused block path uses (bf16, E a multiple of 256): the row statistics come as partials [rows][E/256][2]
// (mean, M2 per 256-column segment, as the producing GEMM epilogue / embedding kernel leaves them) and the LayerNorm OUTPUT
// yout = xhat * gamma + beta is written beside dx (null: not wanted).  dmask (null: not wanted) is written whatever p_drop is.
// Round 6's backward pass: (mean, rstd) of every row from its partial statistics, for any number of LayerNorm sites in ONE launch
// (grid.y = site): the backward pass of the fused block path merges all 2L sites of a step up front, and every consumer (the
// LayerNorm backward kernels, the attention backward kernels) reads plain per-row arrays.
struct LnMergeSite { const float* part; float* mean; float* rstd; };
__global__ void ln_stats_merge_kernel(const LnMergeSite* __restrict__ sites, int np, float eps, int rows) {
    const LnMergeSite st = sites[blockIdx.y];
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    float mu, rs;
    ln_merge_parts(st.part + (int64_t)row * np * 2, np, eps, mu, rs);
    st.mean[row] = mu;
    st.rstd[row] = rs;
}
int ln_stats_merge_run(void* stream, const void* sites_dev, int nsites, int np, float eps, int rows) {
    if (rows == 0 || nsites == 0) return CMP_OK;
    ln_stats_merge_kernel<<<dim3(cdiv(rows, 256), nsites), 256, 0, (hipStream_t)stream>>>((const LnMergeSite*)sites_dev, np, eps, rows);
    KERNEL_CHECK();
    return CMP_OK;
}
// kernel-level forms (tests): one site; and the LayerNorm backward on dy = rstd o gradient with the merged statistics
extern "C" int cmp_k_ln_stats_merge(void* stream, const float* part, int np, float eps, float* mean, float* rstd, int rows) {
    CMP_REQUIRE(part && mean && rstd && np >= 1 && np <= 8, "ln_stats_merge: bad arguments");
    LnMergeSite h{part, mean, rstd};
    LnMergeSite* d = nullptr;
    HIP_CHECK(hipMalloc((void**)&d, sizeof(h)));
    hipError_t e = hipMemcpyAsync(d, &h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)stream);
    int rc = e == hipSuccess ? ln_stats_merge_run(stream, d, 1, np, eps, rows) : CMP_ERR_HIP;
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    return rc;
}
extern "C" int cmp_k_layernorm_bwd_prescaled(void* stream, const void* dy_scaled, const void* x, const float* gamma, const float* mean,
                                             const float* rstd, const void* resid, void* dx, float* dgamma, float* dbeta, void* ws,
                                             int rows, int E, void* dmask, float* colsum, float p_drop, uint64_t seed, uint32_t rng_stream) {
    return layernorm_bwd_run(stream, dy_scaled, x, gamma, mean, rstd, resid, dx, dgamma, dbeta, ws, rows, E, CMP_BF16, dmask, colsum,
                             p_drop, seed, rng_stream, false, nullptr, true, true);
}
extern "C" int cmp_k_wgrad_ln_fix(void* stream, float* G, int rows, int cols, const float* gamma, const float* beta, const float* colsum) {
    return wgrad_ln_fix_run(stream, G, rows, cols, gamma, beta, colsum, nullptr, 0, 0, nullptr, nullptr, nullptr);
}
extern "C" int cmp_k_layernorm_bwd_parts(void* stream, const void* dy, const void* x, const float* gamma, const float* beta,
                                         const float* part, float eps, const void* resid, void* dx, void* yout, float* dgamma,
                                         float* dbeta, void* ws, int rows, int E, void* dmask, float* colsum, float p_drop,
                                         uint64_t seed, uint32_t rng_stream) {
    LnBwdFused fz;
    fz.part = part; fz.np = E / 256; fz.eps = eps; fz.beta = beta; fz.yout = yout;
    return layernorm_bwd_run(stream, dy, x, gamma, nullptr, nullptr, resid, dx, dgamma, dbeta, ws, rows, E, CMP_BF16, dmask, colsum, p_drop,
                             seed, rng_stream, false, &fz, true);
}

// Round 6 (the fused block path's backward pass): a Conv1D weight gradient n^T . D with n = LayerNorm(r) is accumulated by the
// grouped launch as R = r^T . D' on the RAW rows r and D' = rstd o D (what the producing epilogue stored), so that n is never written:
//   n^T . D = gamma o (R - 1 (x) (mean^T . D')) + beta (x) colsum(D),   and   mean^T . D' = (1/E) 1^T . r^T . D' = the COLUMN MEANS of R
// (a row's mean is the mean of that row of r as stored -- what the partial statistics were taken from), so the pass needs nothing
// but R itself, gamma, beta and colsum(D) (= the Conv1D's bias gradient, summed by whoever produced D'): per column, the mean over
// the E rows, then G[k, j] = gamma[k] * (R[k, j] - mean_j) + beta[k] * colsum[j].  One launch for the matrices of a block; a
// workgroup owns 32 columns of one matrix (32 row groups x 32 columns).
struct LnFixDesc { float* G; const float* gamma; const float* beta; const float* colsum; int rows, cols, blk0, pad; };
__global__ __launch_bounds__(1024) void wgrad_ln_fix_kernel(LnFixDesc d0, LnFixDesc d1) {
    __shared__ float part[32][33];
    const LnFixDesc d = ((int)blockIdx.x >= d1.blk0 && d1.G) ? d1 : d0;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 columns (128-byte runs) x 32 row groups
    const int c = ((int)blockIdx.x - d.blk0) * 32 + tx;
    const bool ok = c < d.cols;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                       // independent chains: the loads of a thread overlap
    if (ok) {
        int k = ty;
        for (; k + 96 < d.rows; k += 128) {
            s0 += d.G[(int64_t)k * d.cols + c];
            s1 += d.G[(int64_t)(k + 32) * d.cols + c];
            s2 += d.G[(int64_t)(k + 64) * d.cols + c];
            s3 += d.G[(int64_t)(k + 96) * d.cols + c];
        }
        for (; k < d.rows; k += 32) s0 += d.G[(int64_t)k * d.cols + c];
    }
    part[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int t = 0; t < 32; t++) tot += part[t][tx];                    // (every row group adds them in the same order)
    if (!ok) return;
    const float mean = tot / (float)d.rows, bc = d.colsum[c];
    for (int k = ty; k < d.rows; k += 32) {
        const int64_t at = (int64_t)k * d.cols + c;
        d.G[at] = fmaf(d.gamma[k], d.G[at] - mean, d.beta[k] * bc);
    }
}
int wgrad_ln_fix_run(void* stream, float* G0, int rows0, int cols0, const float* gamma0, const float* beta0, const float* colsum0,
                     float* G1, int rows1, int cols1, const float* gamma1, const float* beta1, const float* colsum1) {
    CMP_REQUIRE(G0 && gamma0 && beta0 && colsum0 && rows0 > 0 && cols0 > 0, "wgrad_ln_fix: bad arguments");
    LnFixDesc d0{G0, gamma0, beta0, colsum0, rows0, cols0, 0, 0};
    LnFixDesc d1{G1, gamma1, beta1, colsum1, rows1, cols1, cdiv(cols0, 32), 0};
    const int grid = cdiv(cols0, 32) + (G1 ? cdiv(cols1, 32) : 0);
    wgrad_ln_fix_kernel<<<grid, 1024, 0, (hipStream_t)stream>>>(d0, d1);
    KERNEL_CHECK();
    return CMP_OK;
}
// deterministic: the per-workgroup partials are folded by ONE thread per column in workgroup order (no float atomics)
int layernorm_bwd_run(void* stream, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* resid, void* dx, float* dgamma, float* dbeta, void* ws, int rows, int E, int dtype, void* dmask,
                      float* colsum, float p_drop, uint64_t seed, uint32_t rng_stream, bool deterministic, const LnBwdFused* fz,
                      bool keep_dmask, bool prescaled) {
    int rc = ln_check(E, dtype);
    if (rc) return rc;
    if (rows == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = ln_bwd_grid(rows);
    size_t smem = (size_t)4 * 3 * E * sizeof(float);
    const int vn = dtype == CMP_BF16 ? 8 : 4;
    const int maxi = cdiv(E / vn, 64);
    DropCfg dcfg = make_drop(p_drop, seed, rng_stream);
    if (fz) CMP_REQUIRE(dtype == CMP_BF16 && fz->part && fz->beta && E % 256 == 0 && fz->np == E / 256,
                        "layernorm_bwd: the statistics-from-partials form needs bf16 rows of a multiple of 256 columns (E=%d, np=%d)", E, fz->np);
    // no mask: the consumer reads dx itself (the fused block path keeps the copy: there dx is overwritten before the weight gradients read it)
    if (p_drop <= 0.f && !fz && !keep_dmask) dmask = nullptr;
    PROF_START(8, s);
    const int want_cs = colsum != nullptr;
    // up to 256 workgroups (rows <= 2048) the per-workgroup partials go to the gradients by atomics from the kernel itself; beyond
    // that (and always in deterministic mode) they are folded by ln_param_reduce_kernel
    const bool direct = !deterministic && grid <= 256;
    float* dg_ = direct ? dgamma : nullptr;
    // (the cross-wave parameter-gradient reduction holds [4 waves][3][E] floats: 96 KiB at E = 2048)
#define LN_BWD(TT, MI) do { \
        if (smem > 65536) HIP_CHECK(hipFuncSetAttribute((const void*)layernorm_bwd_kernel<TT, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        layernorm_bwd_kernel<TT, MI><<<grid, 256, smem, s>>>((const TT*)dy, (const TT*)x, gamma, mean, rstd, (const TT*)resid, (TT*)dx, (float*)ws, rows, E, (TT*)dmask, want_cs, dcfg, dg_, dbeta, colsum); \
    } while (0)
#define LN_BWDF(MI) do { \
        if (smem > 65536) HIP_CHECK(hipFuncSetAttribute((const void*)layernorm_bwd_kernel<bf16_t, MI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        layernorm_bwd_kernel<bf16_t, MI, true><<<grid, 256, smem, s>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)resid, (bf16_t*)dx, (float*)ws, rows, E, (bf16_t*)dmask, want_cs, dcfg, dg_, dbeta, colsum, *fz); \
    } while (0)
#define LN_BWDP(MI) do { \
        if (smem > 65536) HIP_CHECK(hipFuncSetAttribute((const void*)layernorm_bwd_kernel<bf16_t, MI, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        layernorm_bwd_kernel<bf16_t, MI, false, true><<<grid, 256, smem, s>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)resid, (bf16_t*)dx, (float*)ws, rows, E, (bf16_t*)dmask, want_cs, dcfg, dg_, dbeta, colsum); \
    } while (0)
    if (prescaled && !fz) {
        CMP_REQUIRE(dtype == CMP_BF16 && maxi <= 4, "layernorm_bwd: the prescaled form exists for bf16 rows of at most 2048 columns");
        if (maxi == 1) LN_BWDP(1); else if (maxi == 2) LN_BWDP(2); else LN_BWDP(4);
    } else
    if (fz) { if (maxi == 1) LN_BWDF(1); else if (maxi == 2) LN_BWDF(2); else if (maxi <= 4) LN_BWDF(4); else LN_BWDF(8); }
    else if (dtype == CMP_BF16) { if (maxi == 1) LN_BWD(bf16_t, 1); else if (maxi == 2) LN_BWD(bf16_t, 2); else if (maxi <= 4) LN_BWD(bf16_t, 4); else LN_BWD(bf16_t, 8); }
    else { if (maxi == 1) LN_BWD(float, 1); else if (maxi == 2) LN_BWD(float, 2); else if (maxi <= 4) LN_BWD(float, 4); else LN_BWD(float, 8); }
#undef LN_BWD
#undef LN_BWDF
#undef LN_BWDP
    KERNEL_CHECK();
    if (!direct)
        ln_param_reduce_kernel<<<dim3(cdiv(3 * E, 256), deterministic ? 1 : std::min(grid, 32)), 256, 0, s>>>((const float*)ws, dgamma, dbeta, colsum, grid, E);
    {   // class 8: dy, x (+ resid) in, dx (+ the dropout-masked copy) out, mean / rstd
        const double lb = (double)rows * ((3.0 + (resid ? 1.0 : 0.0) + (dmask ? 1.0 : 0.0) + (fz && fz->yout ? 1.0 : 0.0)) * E * dtype_size(dtype) + 8.0);
        PROF_STOP(8, s, lb, lb);
    }
    KERNEL_CHECK();
    return CMP_OK;
}

extern "C" int cmp_k_softmax_xent(void* stream, const float* logits, int ldz, const int32_t* y, void* dlogits,
                                  float* row_loss, int32_t* row_correct, int rows, int V, float inv_n, int dtype) {
    CMP_REQUIRE(V > 0 && V <= ldz, "softmax_xent: V=%d ldz=%d", V, ldz);
    if (rows == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = std::min(cdiv(rows, 4), 8192);
    if (ldz > 64 * XENT_MAXI) {
        if (dtype == CMP_BF16)
            softmax_xent_wide_kernel<bf16_t><<<grid, 256, 0, s>>>(logits, ldz, y, (bf16_t*)dlogits, row_loss, row_correct, rows, V, inv_n);
        else
            softmax_xent_wide_kernel<float><<<grid, 256, 0, s>>>(logits, ldz, y, (float*)dlogits, row_loss, row_correct, rows, V, inv_n);
    } else if (ldz % 8 == 0 && ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0) {
        if (dtype == CMP_BF16)
            softmax_xent8_kernel<bf16_t><<<grid, 256, 0, s>>>(logits, ldz, y, (bf16_t*)dlogits, row_loss, row_correct, rows, V, inv_n);
        else
            softmax_xent8_kernel<float><<<grid, 256, 0, s>>>(logits, ldz, y, (float*)dlogits, row_loss, row_correct, rows, V, inv_n);
    } else if (dtype == CMP_BF16)
        softmax_xent_kernel<bf16_t><<<grid, 256, 0, s>>>(logits, ldz, y, (bf16_t*)dlogits, row_loss, row_correct, rows, V, inv_n);
    else
        softmax_xent_kernel<float><<<grid, 256, 0, s>>>(logits, ldz, y, (float*)dlogits, row_loss, row_correct, rows, V, inv_n);
    KERNEL_CHECK();
    return CMP_OK;
}

int launch_metrics_reduce(hipStream_t s, const float* row_loss, const int32_t* row_correct, int rows, void* metrics) {
    metrics_reduce_kernel<<<1, 1024, 0, s>>>(row_loss, row_correct, rows, (Metrics*)metrics);
    KERNEL_CHECK();
    return CMP_OK;
}

extern "C" int cmp_k_adam(void* stream, float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n,
                          float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale) {
    CMP_REQUIRE(n % 4 == 0, "adam: n=%lld must be a multiple of 4", (long long)n);
    if (n == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    double alpha = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    int64_t n4 = n / 4;
    int grid = (int)std::min<int64_t>(cdiv64(n4, 256), 8192);
    PROF_START(7, s);
    adam_kernel<<<grid, 256, 0, s>>>(p, g, m, v, (bf16_t*)shadow_bf16, n4, n, (float)alpha, beta1, beta2, eps, grad_scale);
    PROF_STOP(7, s, (double)n * (28.0 + (shadow_bf16 ? 2.0 : 0.0)), (double)n * (28.0 + (shadow_bf16 ? 2.0 : 0.0)));
    KERNEL_CHECK();
    return CMP_OK;
}

int launch_cast_bf16(hipStream_t s, const float* in, void* out, int64_t n) {
    if (n == 0) return CMP_OK;
    int64_t n4 = n / 4;
    int grid = (int)std::min<int64_t>(cdiv64(n4, 256), 8192);
    cast_f32_to_bf16_kernel<<<grid, 256, 0, s>>>(in, (bf16_t*)out, n4);
    KERNEL_CHECK();
    return CMP_OK;
}

extern "C" int cmp_k_colsum(void* stream, const void* X, int ldx, float* out, int rows, int cols, int dtype) {
    return colsum_run(stream, X, ldx, out, rows, cols, dtype, nullptr, 0);
}

// det_ws (>= 128 * cols floats): per-split partial sums + a fixed-order fold instead of float atomics
int colsum_run(void* stream, const void* X, int ldx, float* out, int rows, int cols, int dtype, float* det_ws, size_t det_ws_bytes) {
    if (rows == 0 || cols == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    const int vn = dtype == CMP_BF16 ? 8 : 4;
    CMP_REQUIRE(cols % vn == 0 && ldx % vn == 0, "colsum: cols=%d and ldx=%d must be multiples of %d", cols, ldx, vn);
    const int gx = cdiv(cols, 64 * vn);
    int splits = std::max(1, std::min(std::min(rows / 64, 128), std::max(1, 1024 / gx)));
    dim3 grid(gx, splits);
    if (det_ws) CMP_REQUIRE((size_t)splits * cols * 4 <= det_ws_bytes, "colsum: deterministic workspace too small");
    if (dtype == CMP_BF16)
        colsum_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)X, ldx, out, rows, cols, det_ws);
    else
        colsum_kernel<float><<<grid, 256, 0, s>>>((const float*)X, ldx, out, rows, cols, det_ws);
    if (det_ws) colsum_fold_kernel<<<cdiv(cols, 256), 256, 0, s>>>(det_ws, out, splits, cols);
    KERNEL_CHECK();
    return CMP_OK;
}
