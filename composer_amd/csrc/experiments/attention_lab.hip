// attention.hip -- causal scaled-dot-product attention of Attention._multihead_attention
// (transformer.py:331-371) and its gradient, flash-style: the [B,H,T,T] score tensor the reference
// materialises ~6 times per layer is never written to HBM.
//
// Semantics kept from the reference: w = (q.k^T) * rsqrt(D) (:339-348); masked scores are EXACTLY -1e4
// (:351-354), which underflows to probability 0 in fp32, so key tiles entirely above the diagonal are
// skipped without changing a bit; softmax (:360); dropout on the probabilities (:361); w.v (:367).
// split_heads/merge_heads (:373-395) are pure addressing here: q,k,v are read straight out of the
// head-merged c_attn output [B,T,3E] and o is written head-merged [B,T,E].
//
// Matrix-core mapping (32x32 MFMA; bf16: 32x32x16, fp32 parity mode: 32x32x2 = exact fma chains):
//   forward / dQ : S^T = K.Q^T with the QUERY on the lane  -> row max/sum are lane-local (+1 cross-half
//                  exchange), and the f32 accumulator tile is directly the B operand of O^T += V^T.P^T
//                  (resp. dQ^T += K^T.dS^T) -- no LDS round trip for P.
//   dK/dV        : S = Q.K^T with the KEY on the lane -> P and dS are directly the B operands of
//                  dV^T += dO^T.P and dK^T += Q^T.dS; each wave keeps dK^T/dV^T of its 32 keys in registers.
//   "transposed" A operands (V^T, K^T, Q^T, dO^T) come from row-major LDS images through
//   ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (fp32).
#include "../common.h"
#if !defined(COMPOSER_EXPERIMENTS) || !defined(ATTN_DIAG)      // measurement ladders exist in experiments builds only
#undef ATTN_DIAG
#define ATTN_DIAG 0
#endif

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// combine a value with the one on lane ^ 32 (the other 32x32 half): v_permlane32_swap is a VALU instruction, the
// ds_bpermute shuffle it replaces was an LDS round trip on the softmax critical path.
__device__ __forceinline__ void both_halves(float v, float& a, float& b) {
    uint32_t u = __builtin_bit_cast(uint32_t, v), u2 = u;
    // v_permlane32_swap swaps lanes 32..63 of its first operand with lanes 0..31 of its second, in place:
    // u -> {lo, lo}, u2 -> {hi, hi}.  (Issued as asm: this hipcc lowers the builtin's SECOND result to the first
    // operand's register -- checked on hardware with tools/ubench/permlane_test.hip.)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(u2));
    const uint32_t r[2] = {u, u2};
    a = __builtin_bit_cast(float, r[0]);
    b = __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float half_max(float v) { float a, b; both_halves(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float v) { float a, b; both_halves(v, a, b); return a + b; }

template <typename T> struct AT;
template <> struct AT<bf16_t> {
    static constexpr int KSTEP = 16;
    static constexpr int VN = 8;
    typedef bf16x8 frag;
    static constexpr bool EXACT = false;
};
template <> struct AT<float> {
    static constexpr int KSTEP = 2;
    static constexpr int VN = 4;
    typedef float frag;
    static constexpr bool EXACT = true;
};

template <typename T> using frag_t = typename AT<T>::frag;

// Requiring 2 waves/SIMD caps the kernel at 256 registers, which makes hipcc keep MFMA accumulators in VGPR form;
// with the 512-register budget of 1 wave/SIMD it parks them in AGPRs and pays a v_accvgpr_read/write pair for
// every softmax/rescale touch of an accumulator (measured: ~40% of the loop's vector instructions).
template <typename T, int D> struct Occ {
    static constexpr int MINW = (std::is_same<T, float>::value && D == 128) ? 1 : 2;
    // forward and dQ, bf16 heads up to 64 wide: 3 waves/SIMD (168 registers; dQ lands on 168-170 by itself, and 170
    // already rounds up to a 2-wave allocation).  Left at 2, hipcc's scheduler spends the spare
    // registers on a longer software pipeline (213) and the lost occupancy costs more than the schedule gains; at 4
    // (128 registers) it spills ~120 registers inside the key loop (measured: 2.2x slower).
#ifndef ATTN_MINW_Q
#define ATTN_MINW_Q 3
#endif
    static constexpr int MINW_Q = (std::is_same<T, bf16_t>::value && D <= 64) ? ATTN_MINW_Q : MINW;
};

template <typename T, int D> struct Geo {
    static constexpr int DP = D < 32 ? 32 : D;                       // padded head width (zero filled)
    static constexpr int DT = DP / 32;                               // 32-row output tiles along d
    // LDS row stride in elements. bf16: 2*S bytes, multiple of 16, odd number of 16-B chunks (b128 row reads
    // conflict-free).  fp32: odd (b32 row reads with the row on the lane conflict-free).
    static constexpr int S = std::is_same<T, float>::value ? DP + 1 : DP + 8;
    static constexpr int NS = D / AT<T>::KSTEP;                      // MFMA steps over the head dimension
};

template <typename T> __device__ __forceinline__ f32x16 mfma32(frag_t<T> a, frag_t<T> b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<float>(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// stage `nrows` rows x D of a head-merged global tensor into an LDS image [nrows][S], zero filled
template <typename T, int D>
__device__ __forceinline__ void stage_rows(T* img, const T* __restrict__ g, int64_t gstride, int row0, int row_end,
                                           int nrows, int tid) {
    constexpr int VN = AT<T>::VN, DP = Geo<T, D>::DP, S = Geo<T, D>::S;
    constexpr int CPR = DP / VN;
    for (int c = tid; c < nrows * CPR; c += 256) {
        int r = c / CPR, cc = c % CPR;
        Vec16<T> v;
#pragma unroll
        for (int j = 0; j < VN; j++) v.set(j, 0.f);
        if (row0 + r < row_end && cc * VN < D) v = ld16(g + (int64_t)(row0 + r) * gstride + cc * VN);
        if (std::is_same<T, float>::value) {
#pragma unroll
            for (int j = 0; j < VN; j++) img[r * S + cc * VN + j] = from_f32<T>(v.get(j));
        } else {
            st16(img + r * S + cc * VN, v);
        }
    }
}

// Software-pipelined staging of a [64 rows][D] tile: load() issues the global loads of the NEXT tile into registers
// before the current tile's MFMA/softmax work (latency hides under it), store() writes them to the other LDS
// buffer afterwards -- one barrier per tile.
template <typename T, int D, int ROWS = 64> struct Stager {
    static constexpr int VN = AT<T>::VN, DP = Geo<T, D>::DP, S = Geo<T, D>::S;
    // 16-byte chunks per row that exist in memory.  Heads narrower than the padded width (D = 16 in bf16: DP = 32) stage only those:
    // the padding columns of the LDS image are never written and feed nothing that is stored (mma_rows reads D columns; the d >= D
    // rows of a transposed output tile, which mma_acc_b computes from them, are dropped by store_t_tile / colsum_t_tiles).
    static constexpr int CPR = (D < DP ? D : DP) / VN;
    static constexpr int NC = (ROWS * CPR + 255) / 256;
    static constexpr bool WHOLE = (ROWS * CPR) % 256 == 0;          // every thread owns NC chunks
    Vec16<T> r[NC];
    __device__ __forceinline__ void load(const T* __restrict__ g, int64_t gstride, int row0, int row_end, int tid) {
        if constexpr (!std::is_same<T, float>::value || D >= 32) {
            // Range-checked buffer loads: rows at or past row_end come back as zeros from the hardware, so the loads are
            // unconditional straight-line code.  (The guarded global loads they replace sat under divergent branches;
            // hipcc's wait-count pass then put s_waitcnt vmcnt(1)/vmcnt(0) in front of the first MFMAs of the CURRENT
            // tile, i.e. every 64-key tile waited for the NEXT tile's HBM round trip: ~3.5k cycles per 32-key
            // sub-tile instead of ~1k.)  A thread without a chunk (ROWS * CPR < 256) reads past the range: zeros, never stored.
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
            const int64_t span = ((int64_t)(row_end - 1) * gstride + D) * (int64_t)sizeof(T);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)span, 0x00020000);
#pragma unroll
            for (int i = 0; i < NC; i++) {
                const int c = tid + 256 * i;
                const int row = c / CPR, cc = c % CPR;
                int off = (int)(((int64_t)(row0 + row) * gstride + cc * VN) * (int64_t)sizeof(T));
                if (!WHOLE && c >= ROWS * CPR) off = 0x7FFFFFF0;
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                r[i].v = __builtin_bit_cast(decltype(r[i].v), w);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const int c = tid + 256 * i;
            const int row = c / CPR, cc = c % CPR;
#pragma unroll
            for (int j = 0; j < VN; j++) r[i].set(j, 0.f);
            if (c < ROWS * CPR && row0 + row < row_end && cc * VN < D) r[i] = ld16(g + (int64_t)(row0 + row) * gstride + cc * VN);
        }
    }
    __device__ __forceinline__ void store(T* img, int tid) const {
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const int c = tid + 256 * i;
            const int row = c / CPR, cc = c % CPR;
            // whole chunks per thread (every D >= 32): unconditional, so the stores are straight-line code that can be scheduled
            // among the tile's last MFMAs instead of four exec-masked blocks behind them
#ifdef ATTN_COND_STORE
            if (c < ROWS * CPR) {
#else
            if (WHOLE || c < ROWS * CPR) {
#endif
                if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                    for (int j = 0; j < VN; j++) img[row * S + cc * VN + j] = r[i].get(j);
                } else {
                    st16(img + row * S + cc * VN, r[i]);
                }
            }
        }
    }
};

// per-lane register fragments of a [32 rows][D] global tile used as the MFMA B operand B[k=d][col=row]
template <typename T, int D>
__device__ __forceinline__ void load_bfrags(frag_t<T>* f, const T* __restrict__ g,
                                            int64_t gstride, int row, bool valid, int h) {
    constexpr int NS = Geo<T, D>::NS;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if constexpr (std::is_same<T, float>::value) {
            f[s] = valid ? g[(int64_t)row * gstride + 2 * s + h] : 0.f;
        } else {
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; j++) z[j] = (bf16_t)0.f;
            f[s] = valid ? *reinterpret_cast<const bf16x8*>(g + (int64_t)row * gstride + 16 * s + 8 * h) : z;
        }
    }
}

// C(32x32) += rows(img[row0..row0+31][0..D)) . Bfrag      (A operand read row-wise from the LDS image)
template <typename T, int D>
__device__ __forceinline__ f32x16 mma_rows(const T* img, int row0, const frag_t<T>* bf,
                                           int lane, f32x16 acc) {
    constexpr int NS = Geo<T, D>::NS, S = Geo<T, D>::S;
    const int r = row0 + (lane & 31), h = lane >> 5;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        frag_t<T> a;
        if constexpr (std::is_same<T, float>::value) a = img[r * S + 2 * s + h];
        else a = *reinterpret_cast<const bf16x8*>(img + r * S + 16 * s + 8 * h);
        acc = mfma32<T>(a, bf[s], acc);
    }
    return acc;
}

// Y^T tile dt (32 d-rows x 32 cols) += img[krow0..krow0+31][32dt..32dt+31]^T . X, X = a 32x32 accumulator
// (rows = the contraction index, in registers; cols on the lanes) used directly as the B operand.
template <typename T, int D>
__device__ __forceinline__ f32x16 mma_acc_b(const T* img, int krow0, int dt, const f32x16& x, int lane, f32x16 acc) {
    constexpr int S = Geo<T, D>::S;
    const int h = lane >> 5;
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float a = img[(krow0 + rho(r, h)) * S + 32 * dt + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[r], acc, 0, 0, 0);
        }
    } else {
        const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int col = 32 * dt + 16 * (G & 1) + 4 * p;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int kr = krow0 + 16 * s + 4 * (G >> 1) + q;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + kr * S + col));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + (kr + 8) * S + col));
            bf16x8 a, b;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a[j] = lo[j];
                a[4 + j] = hi[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = (bf16_t)x[8 * s + j];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    return acc;
}

#define LOG2E_F 1.4426950408889634f
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max16(const f32x16& s) {
    float a = max3(s[0], s[1], s[2]), b = max3(s[3], s[4], s[5]), c = max3(s[6], s[7], s[8]);
    float d = max3(s[9], s[10], s[11]), e = max3(s[12], s[13], s[14]);
    return max3(max3(a, b, c), max3(d, e, s[15]), a);
}

// p = exp2(s * c - m) on all 16 values; returns the sum of the results as two partial sums.  Scalar fma / exp / add on purpose
// (and attention.hip is compiled without SLP vectorisation, build.py): the packed forms v_pk_fma_f32 / v_pk_add_f32 process two
// values per instruction but cost more issue time beside MFMAs than the two scalar instructions (whole-step A/B in build.py).
__device__ __forceinline__ f32x2 exp2_scaled16(f32x16& s, float c, float negm, f32x2 sum) {
    float s0 = sum[0], s1 = sum[1];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(fmaf(s[r], c, negm)), p1 = fast_exp2(fmaf(s[r + 1], c, negm));
        s0 += p0;
        s1 += p1;
        s[r] = p0;
        s[r + 1] = p1;
    }
    return f32x2{s0, s1};
}
// Keeps hipcc from sinking a select below the bf16 conversion: it rewrites cvt(select(c, p, 0)) as
// select(c, cvt(p), 0), which turns 8 two-value v_cvt_pk_bf16_f32 into 16 single conversions plus 8 v_perm_b32.
__device__ __forceinline__ void pin16(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float x = v[r];
        asm("" : "+v"(x));
        v[r] = x;
    }
}

// Dropout keep-test on the 16 values of one 32-key sub-tile with the QUERY on the lane (forward, dQ): the four
// registers 4*g4..4*g4+3 hold consecutive keys, so they share the xor base and use the four fixed 24-bit multipliers
// (v_mul_u32_u24 is full rate; v_mul_lo_u32 is quarter rate).  Dropped entries become 0; the 1/(1-p) scale is applied
// ONCE to the kernel's output instead of per element.
template <bool PIN = false>
__device__ __forceinline__ void mask16_qlane(f32x16& v, uint32_t rowh, int k0, int h, uint32_t thr) {
    const uint32_t gg = ((uint32_t)(k0 >> 2) + (uint32_t)h) * ATTN_G;
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
        const uint32_t x = rowh ^ (gg + (uint32_t)(2 * g4) * ATTN_G);
        v[4 * g4 + 0] = (__umul24(x, ATTN_C0) >= thr) ? v[4 * g4 + 0] : 0.f;
        v[4 * g4 + 1] = (__umul24(x, ATTN_C1) >= thr) ? v[4 * g4 + 1] : 0.f;
        v[4 * g4 + 2] = (__umul24(x, ATTN_C2) >= thr) ? v[4 * g4 + 2] : 0.f;
        v[4 * g4 + 3] = (__umul24(x, ATTN_C3) >= thr) ? v[4 * g4 + 3] : 0.f;
    }
    if (PIN) pin16(v);
}

// store a transposed accumulator tile Y^T[d][row] (d in registers, row on the lane) to out[row][d0 + d]
template <typename T, int D>
__device__ __forceinline__ void store_t_tile(T* __restrict__ out, int64_t ostride, int row, bool valid, int dt,
                                             const f32x16& y, float mul, int h) {
    if constexpr (std::is_same<T, bf16_t>::value && D % 32 == 0) {
        // The two lanes of a row (l, l + 32) own alternating 8-byte groups of it.  One v_permlane32_swap per dword regroups them
        // so that each lane holds 16 contiguous bytes: 2 dwordx4 stores per tile and lane instead of 4 dwordx2 (the store tail of
        // a block is store-ISSUE-bound, MI355X_MICROARCH.md "attention epilogue store tail").
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
            uint32_t w[2][2];                  // [group gp | gp + 1][dword]: 4 bf16 each
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
                    const bf2 t = {(bf16_t)(y[4 * (gp + k) + 2 * j] * mul), (bf16_t)(y[4 * (gp + k) + 2 * j + 1] * mul)};
                    w[k][j] = __builtin_bit_cast(uint32_t, t);
                }
            // swap lanes 32..63 of the first operand with lanes 0..31 of the second: afterwards lane l < 32 holds group gp of
            // both halves (its own 4 values, then its partner's), lane l + 32 group gp + 1 of both halves
#pragma unroll
            for (int j = 0; j < 2; j++) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(w[0][j]), "+v"(w[1][j]));
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
            const u32x4 v = {w[0][0], w[0][1], w[1][0], w[1][1]};
            if (valid) *reinterpret_cast<u32x4*>(out + (int64_t)row * ostride + 32 * dt + 8 * (gp + h)) = v;
        }
        return;
    }
    if (!valid) return;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        int d = 32 * dt + 8 * g + 4 * h;
        if (d < D) {
            if constexpr (std::is_same<T, float>::value) {
                f32x4 v = {y[4 * g] * mul, y[4 * g + 1] * mul, y[4 * g + 2] * mul, y[4 * g + 3] * mul};
                *reinterpret_cast<f32x4*>(out + (int64_t)row * ostride + d) = v;
            } else {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (bf16_t)(y[4 * g + j] * mul);
                *reinterpret_cast<bf16x4*>(out + (int64_t)row * ostride + d) = v;
            }
        }
    }
}

// Column sums of a wave's transposed output tiles (Y^T[d][row], 32 rows on the lanes of each half) added to
// bias_grad[d]: the c_attn bias gradient is the column sum of [dQ | dK | dV] (Conv1D bias under tf.GradientTape,
// transformer.py:205-209); taking it from the f32 accumulators saves a 63 us re-read of the [M,3E] gradient per layer.
// Transpose-reduce: 31 shuffles leave lane l (of each half) with the total of value l, then ONE atomic instruction per
// wave writes 64 distinct columns.  (A first version reduced every value to lane 0 and issued 32 two-lane atomics per
// tile set: 3 M atomic instructions per layer on 1536 addresses cost 0.8 ms.)
template <typename T, int D>
__device__ __forceinline__ void colsum_t_tiles(float* __restrict__ bias_grad, bool valid, const f32x16* y, float mul, int h,
                                               int lane) {
    constexpr int DT = Geo<T, D>::DT;
    const int l = lane & 31;
#pragma unroll
    for (int t0 = 0; t0 < DT; t0 += 2) {
        constexpr int dummy = 0;
        (void)dummy;
        const bool pair = t0 + 1 < DT;                  // two tiles = 32 values, or a single tile = 16
        float v[32];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v[r] = valid ? y[t0][r] * mul : 0.f;
            v[16 + r] = (pair && valid) ? y[pair ? t0 + 1 : t0][r] * mul : 0.f;
        }
        if (pair) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float keep = (l & 16) ? v[k + 16] : v[k], send = (l & 16) ? v[k] : v[k + 16];
                v[k] = keep + __shfl_xor(send, 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] += __shfl_xor(v[k], 16);
        }
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1) {
#pragma unroll
            for (int k = 0; k < s; k++) {
                const float keep = (l & s) ? v[k + s] : v[k], send = (l & s) ? v[k] : v[k + s];
                v[k] = keep + __shfl_xor(send, s);
            }
        }
        // lane l now holds value index l (pair) or l & 15 (single tile; both 16-lane groups hold the same totals)
        const int idx = pair ? l : (l & 15);
        const int d = 32 * (t0 + (idx >> 4)) + 8 * ((idx & 15) >> 2) + 4 * h + (idx & 3);
        if ((pair || l < 16) && d < D) atomicAdd(bias_grad + d, v[0]);
    }
}

// Causal work balance: query block i needs i+1 key tiles.  A workgroup takes the PAIR (nb-1-x, x) -- heavy one
// first -- so every workgroup does nb+1 tiles (the middle block of an odd count runs alone).  grid.x = (nb+1)/2.
// When B*H is small (the default config at batch 1: 16 (batch, head) rows) the launcher gives every block its own workgroup
// instead (gridDim.x == nb, heaviest first): twice the workgroups, each half as long; with many rows the pairing is faster
// (same-box A/B at B*H = 1024: 303-317 us paired, 341-346 us unpaired).
__device__ __forceinline__ int pair_block(int ph, int nb, int x) {
    if ((int)gridDim.x == nb && nb > 1) return ph == 0 ? nb - 1 - x : -1;
    const int hi = nb - 1 - x;
    if (ph == 0) return hi;
    return x < hi ? x : -1;
}

// XCD-aware block mapping: consecutive workgroup ids go to different XCDs (id mod 8), so the gridDim.x workgroups
// that share one (batch, head)'s K/V (Q/dO) would land on different L2s.  Re-deal: XCD x takes (batch, head) rows
// x, x+8, ... and runs the gridDim.x blocks of a row on consecutive slots -> one L2 serves the row's tile re-reads.
// With H a multiple of 8 that alone would give every (batch, head) row of one XCD the SAME head index (by = 8k + xcd), i.e.
// the same 128-byte column slice of the [B,T,3E] tensor for every K/V/Q tile read through that L2 -- rows 3E*2 bytes apart,
// a fixed subset of its channels.  The head index is therefore rotated by the batch index (a bijection inside a batch).
__device__ __forceinline__ void xcd_block(int& bx, int& by, int H) {
    const int gx = gridDim.x, gy = gridDim.y;
    if (gy & 7) { bx = blockIdx.x; by = blockIdx.y; return; }
    const int L = blockIdx.x + gx * blockIdx.y;          // dispatch order: x fastest
    const int xcd = L & 7, slot = L >> 3;
    bx = slot % gx;
    by = (slot / gx) * 8 + xcd;
#ifndef ATTN_NO_HEAD_ROTATE
    const int bb = by / H;
    by = bb * H + (by % H + bb) % H;
#endif
}

// Block plan of the forward and dQ launches (plan_u >= 0; 1-D grid, B*H a multiple of 8).  All paired workgroups cost the same
// (nb+1 tiles), so a grid that is not a whole number of rounds over the resident slots (B*H = 1024: 4096 pairs over 768 slots =
// 5.33 rounds; B*H = 256: 1.33) ends with most of the chip idle.  The plan takes the first plan_u (batch, head) rows of every
// XCD out of the pairing: an XCD runs its paired rows first, then the single blocks of those rows heaviest level first
// (all rows' block nb-1, then nb-2, ...), so the launch ends on the cheapest jobs (longest-processing-time order).
// Workgroup L: xcd = L & 7, slot = L >> 3; gridDim.x = 8 * ((rows_per_xcd - plan_u) * pairs + plan_u * nb).
__device__ __forceinline__ void attn_job(int plan_u, int nb, int H, int& by, int& qb0, int& qb1) {
    if (plan_u < 0) {
        int bx;
        xcd_block(bx, by, H);
        qb0 = pair_block(0, nb, bx);
        qb1 = pair_block(1, nb, bx);
        return;
    }
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int pairs = (nb + 1) >> 1;
    const int n_paired = (int)(gridDim.x >> 3) - plan_u * nb;
    int row;
    if (slot < n_paired) {
        row = plan_u + slot / pairs;
        const int x = slot % pairs, hi = nb - 1 - x;
        qb0 = hi;
        qb1 = x < hi ? x : -1;
    } else {
        const int s2 = slot - n_paired;
        row = s2 % plan_u;
        qb0 = nb - 1 - s2 / plan_u;
        qb1 = -1;
    }
    by = row * 8 + xcd;
#ifndef ATTN_NO_HEAD_ROTATE
    const int bb = by / H;
    by = bb * H + (by % H + bb) % H;
#endif
}

// =================================================================================================
// forward.  grid ((nb+1)/2, B*H), 256 threads: wave w owns query rows [qb*128 + 32w, +32)
// =================================================================================================
// AMASK: Transformer.call's attention_mask (transformer.py:774-779, 356-358): amask[b][key] = (1 - mask) * -1e4 is ADDED to the
// scaled, causally masked scores of every query of batch row b (forward passes only; the train loop never passes a mask).
// KS ("key split", small grids -- attn_key_split below): the workgroup owns ONE 32-row query block, its four waves take the four
// 64-key quarters of every staged 256-key tile and their (m, l, O) partials are merged through LDS at the end: a quarter of the
// serial tile walk per wave and four times the workgroups.  grid (ceil(T/32), B*H), heaviest block first.
template <typename T, int D, bool DROP, bool AMASK = false, bool KS = false>
__global__ __launch_bounds__(256, (KS ? 2 : Occ<T, D>::MINW_Q)) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                                          float* __restrict__ lse, int Tn, int H,
                                                                          float scale, DropCfg drop, int plan_u,
                                                                          const float* __restrict__ amask) {
    using G = Geo<T, D>;
    constexpr bool EXACT = AT<T>::EXACT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Kb = reinterpret_cast<T*>(smem_raw);          // 2 x { K [KT][S] | V [KT][S] }
    static_assert(!KS || (!EXACT && !AMASK), "key split: throughput mode without a mask term");
    constexpr int KT = KS ? 256 : 64;                // keys per staged tile
    constexpr int QR = KS ? 32 : 128;                // query rows per workgroup
    constexpr int IMG = KT * G::S;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = KS ? wave * 64 : 0;               // this wave's keys inside a staged tile
    const int nb = cdiv(Tn, 128);
    int by, qb0, qb1;
    if constexpr (KS) {
        by = blockIdx.y;
        qb0 = (int)gridDim.x - 1 - (int)blockIdx.x;
        qb1 = -1;
    } else {
        attn_job(plan_u, nb, H, by, qb0, qb1);
    }
#if ATTN_DIAG == 9                                   // measurement build: every batch row aliases row 0 or 1 (data set stays in L2 / MALL)
    const int b = (by / H) & 1, hd = by % H;
#else
    const int b = by / H, hd = by % H;
#endif
    const int E = H * D;
    const int64_t rs = 3 * E;                        // row stride of qkv
    const T* qg = qkv + (int64_t)b * Tn * rs + hd * D;
    const T* kg = qg + E;
    const T* vg = qg + 2 * E;
    T* og = o + (int64_t)b * Tn * E + hd * D;
    const float c2 = scale * LOG2E_F, neg_big = -1e4f / scale;
    // Deferred rescale (throughput mode): the running maximum m is only raised -- and O, l rescaled -- when some row's new maximum
    // exceeds it by more than 2^RESCALE_LOG2 in the probability domain; until then the probabilities are taken against the stale
    // maximum (at most 2^RESCALE_LOG2, exact in fp32 sums and as bf16 operands) and O / l stays the same quotient.  With random
    // scores some row of a wave sets a new maximum in nearly every tile, so the exact vote rescaled (34 multiplies and an
    // exponential per lane) almost always; the deferred one does so on the first tile of a row and rarely after.
#ifndef ATTN_RESCALE_LOG2
#define ATTN_RESCALE_LOG2 8.0f
#endif
    const float rescale_raw = ATTN_RESCALE_LOG2 / c2;
    const float* am = AMASK ? amask + (int64_t)b * Tn : nullptr;
    const float inv_scale = 1.0f / scale;           // throughput mode works on raw scores: the mask term goes in divided by the scale

    for (int ph = 0; ph < 2; ph++) {
#ifdef ATTN_LIGHT_FIRST
        // the pair's LIGHT query block first: the four pair-workgroups of a (batch, head) row then start on key tile 0 together and
        // their heavy blocks follow one tile apart -- a sliding window of the row's K/V stays in the XCD's L2 instead of the light
        // blocks re-reading the first tiles long after the heavy ones streamed past them
        const int qb = qb1 < 0 ? (ph == 0 ? qb0 : -1) : (ph == 0 ? qb1 : qb0);
#else
        const int qb = ph == 0 ? qb0 : qb1;
#endif
        if (qb < 0) break;
        const int q0w = KS ? qb * 32 : qb * 128 + wave * 32;
        const int q = q0w + (lane & 31);
        const bool qvalid = q < Tn;
        frag_t<T> qf[G::NS];
        load_bfrags<T, D>(qf, qg, rs, q, qvalid, h);
        f32x16 oacc[G::DT];
#pragma unroll
        for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[dt][r] = 0.f;
        float m = -INFINITY, lsum = 0.f;   // running max: scaled domain (parity mode) / raw domain (throughput mode)
        // With a mask term no key tile is skipped: a causally masked key sits at -1e4 + its mask term, and in a row whose allowed
        // keys are all masked out (a padding query) that is not negligible against the row maximum -- the reference's softmax
        // runs over all of them (transformer.py:351-360).
        const int kv_end = AMASK ? Tn : min(Tn, qb * QR + QR);
        const uint32_t rowh = attn_row_hash(drop.seed, drop.stream, (uint32_t)(by * Tn + q));

        Stager<T, D, KT> sk, sv;
        sk.load(kg, rs, 0, Tn, tid);
        sv.load(vg, rs, 0, Tn, tid);
        sk.store(Kb, tid);
        sv.store(Kb + IMG, tid);
        __syncthreads();
        // One 64-key tile.  `interior` (compile time) = the tile lies below the diagonal for EVERY wave of the workgroup:
        // that instance has no branch around an accumulating MFMA.  (With the skip/mask branches of the general form in
        // the only loop, hipcc carried the O^T accumulators through 64 v_mov_b64 per pair of tiles -- MFMA into a copy,
        // copy back at the join -- and waited out the MFMA pipeline before the copies.)
        auto tile = [&](auto interior, const int kts, const int it) __attribute__((always_inline)) {
            const T* Ks = Kb + (it & 1) * 2 * IMG + wk * G::S;
            const T* Vs = Ks + IMG;
            const bool more = kts + KT < kv_end;
            const int kt0 = kts + wk;                 // first key of this wave's 64
#if ATTN_DIAG == 7          // every tile load re-reads tile 0 (L1 / L2 hits): instruction issue without the traffic
            if (more) {
                sk.load(kg, rs, decltype(interior)::value ? 0 : kts + KT, Tn, tid);
                sv.load(vg, rs, decltype(interior)::value ? 0 : kts + KT, Tn, tid);
            }
#elif ATTN_DIAG != 5
            if (more) {
                sk.load(kg, rs, kts + KT, Tn, tid);
                sv.load(vg, rs, kts + KT, Tn, tid);
            }
#endif
            // Fast path (throughput mode): all 64 keys of the tile are at or below every query of this wave -> no
            // masking; both 32-key sub-tiles go through ONE softmax step: 8 score MFMAs back to back, one row-max exchange
            // and one rescale vote per 64 keys, 32 exponentials, 8 PV MFMAs.
            const bool full64 = decltype(interior)::value || (!EXACT && (kt0 + 63 <= q0w) && (kt0 + 64 <= Tn));   // wave-uniform
            if (full64) {
                f32x16 s0, s1;
#pragma unroll
                for (int r = 0; r < 16; r++) { s0[r] = 0.f; s1[r] = 0.f; }
                // ATTN_DIAG (measurement builds only -- results are wrong; tools/kbench.py attn):
                // 1 no exponentials, 2 no PV MFMAs, 3 no score MFMAs / K fragment reads, 4 no staging stores and no barrier,
                // 5 no global loads either, 6 no row max / rescale
#if ATTN_DIAG == 3
#pragma unroll
                for (int r = 0; r < 16; r++) { s0[r] = (float)(kt0 + r) * 1e-3f; s1[r] = (float)(kt0 - r) * 1e-3f; }
#else
                s0 = mma_rows<T, D>(Ks, 0, qf, lane, s0);
                s1 = mma_rows<T, D>(Ks, 32, qf, lane, s1);
#endif
                if constexpr (AMASK) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        s0[r] += am[kt0 + rho(r, h)] * inv_scale;
                        s1[r] += am[kt0 + 32 + rho(r, h)] * inv_scale;
                    }
                }
#if ATTN_DIAG == 6
                const float mloc = 0.f;
#else
                const float mloc = half_max(fmaxf(max16(s0), max16(s1)));
#endif
                const float mnew = fmaxf(m, mloc);
                if (!__all(mnew <= m + rescale_raw)) {
                    const float alpha = fast_exp2((m - mnew) * c2);
                    lsum *= alpha;
#pragma unroll
                    for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                        for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
                    m = mnew;
                }
                const float mc = m * c2;
                f32x2 ps = {0.f, 0.f};
#if ATTN_DIAG == 1
#pragma unroll
                for (int r = 0; r < 16; r++) { s0[r] = fmaf(s0[r], c2, -mc); s1[r] = fmaf(s1[r], c2, -mc); ps[0] += s0[r]; ps[1] += s1[r]; }
#else
                ps = exp2_scaled16(s0, c2, -mc, ps);
                ps = exp2_scaled16(s1, c2, -mc, ps);
#endif
                lsum += ps[0] + ps[1];
                if constexpr (DROP) {
                    mask16_qlane<true>(s0, rowh, kt0, h, drop.thr);
                    mask16_qlane<true>(s1, rowh, kt0 + 32, h, drop.thr);
                }
#if ATTN_DIAG == 2
#pragma unroll
                for (int r = 0; r < 16; r++) { oacc[0][r] += s0[r]; oacc[G::DT - 1][r] += s1[r]; }
#else
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++) {
                    oacc[dt] = mma_acc_b<T, D>(Vs, 0, dt, s0, lane, oacc[dt]);
                    oacc[dt] = mma_acc_b<T, D>(Vs, 32, dt, s1, lane, oacc[dt]);
                }
#endif
            } else
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const int k0 = kt0 + 32 * sub;
                if ((!AMASK && k0 > q0w + 31) || k0 >= Tn) continue;          // wave-uniform: tile entirely masked
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; r++) s[r] = 0.f;
                s = mma_rows<T, D>(Ks, 32 * sub, qf, lane, s);
                if constexpr (EXACT) {
                    // parity mode: the reference's arithmetic as written (scale, mask to exactly -1e4, exp)
                    float mloc = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        int key = k0 + rho(r, h);
                        float v = s[r] * scale;
                        if (key > q || key >= Tn) v = -1e4f;          // w*b - 1e4*(1-b), transformer.py:354
                        if constexpr (AMASK) v = key < Tn ? v + am[key] : -INFINITY;     // :356-358; keys past the end do not exist
                        s[r] = v;
                        mloc = fmaxf(mloc, v);
                    }
                    mloc = half_max(mloc);
                    const float mnew = fmaxf(m, mloc);
                    const float alpha = expf(m - mnew);
                    float ps = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float p = expf(s[r] - mnew);
                        ps += p;
                        s[r] = p;
                    }
                    lsum = lsum * alpha + ps;
                    m = mnew;
#pragma unroll
                    for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                        for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
                } else {
                    // throughput mode: raw scores, the scale folded into ONE fma feeding v_exp_f32 (base 2), mask only
                    // on tiles touching the diagonal / sequence end, rescale only when some row's max actually grew.
                    const bool edge = (k0 + 31 > q0w) || (k0 + 32 > Tn);      // wave-uniform
                    if (edge) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            int key = k0 + rho(r, h);
                            if (key > q || key >= Tn) s[r] = neg_big;       // == -1e4 after scaling
                        }
                    }
                    if constexpr (AMASK) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int key = k0 + rho(r, h);
                            s[r] = key < Tn ? s[r] + am[key] * inv_scale : -INFINITY;
                        }
                    }
                    float mloc = max16(s);
                    mloc = half_max(mloc);
                    const float mnew = fmaxf(m, mloc);
                    if (!__all(mnew <= m + rescale_raw)) {
                        const float alpha = fast_exp2((m - mnew) * c2);
                        lsum *= alpha;
#pragma unroll
                        for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                            for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
                        m = mnew;
                    }
                    const float mc = m * c2;
                    const f32x2 ps = exp2_scaled16(s, c2, -mc, f32x2{0.f, 0.f});
                    lsum += ps[0] + ps[1];
                }
                if constexpr (DROP) mask16_qlane<true>(s, rowh, k0, h, drop.thr);
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++) oacc[dt] = mma_acc_b<T, D>(Vs, 32 * sub, dt, s, lane, oacc[dt]);
            }
#if ATTN_DIAG == 4 || ATTN_DIAG == 5
            if (!decltype(interior)::value) {
#endif
            if (more) {
                T* nbuf = Kb + ((it & 1) ^ 1) * 2 * IMG;
                sk.store(nbuf, tid);
                sv.store(nbuf + IMG, tid);
            }
            __syncthreads();
#if ATTN_DIAG == 4 || ATTN_DIAG == 5
            }
#endif
        };
        int kt0 = 0, it = 0;
        const int interior_end = EXACT ? 0 : (qb * QR) / KT * KT;      // whole tiles below every query row of the block
        for (; kt0 < interior_end; kt0 += KT, it++) tile(std::true_type{}, kt0, it);
        for (; kt0 < kv_end; kt0 += KT, it++) tile(std::false_type{}, kt0, it);
        float ltot = half_sum(lsum);
        if constexpr (KS) {
            // merge the four waves' partial softmax states (the last tile's barrier has passed: the staging buffers are free)
            constexpr int NV = 2 + 16 * G::DT;
            float* mg = reinterpret_cast<float*>(smem_raw);
            if (wave > 0) {
                float* w = mg + (wave - 1) * NV * 64 + lane;
                w[0] = m;
                w[64] = ltot;
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) w[(2 + 16 * dt + r) * 64] = oacc[dt][r];
            }
            __syncthreads();
            if (wave > 0) continue;               // (single phase: leaves the loop)
            float mstar = m;
#pragma unroll
            for (int w = 0; w < 3; w++) mstar = fmaxf(mstar, mg[w * NV * 64 + lane]);
            const float a0 = fast_exp2((m - mstar) * c2);      // wave 0 always holds key 0: m is finite
            ltot *= a0;
#pragma unroll
            for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) oacc[dt][r] *= a0;
#pragma unroll
            for (int w = 0; w < 3; w++) {
                const float* src = mg + w * NV * 64 + lane;
                const float aw = fast_exp2((src[0] - mstar) * c2);     // a wave without keys left m = -inf: factor 0
                ltot = fmaf(src[64], aw, ltot);
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) oacc[dt][r] = fmaf(src[(2 + 16 * dt + r) * 64], aw, oacc[dt][r]);
            }
            m = mstar;
        }
        const float inv = (DROP ? drop.scale : 1.0f) / ltot;
#pragma unroll
        for (int dt = 0; dt < G::DT; dt++) store_t_tile<T, D>(og, E, q, qvalid, dt, oacc[dt], inv, h);
        if (qvalid && h == 0) lse[(int64_t)by * Tn + q] = (EXACT ? m : m * scale) + logf(ltot);
    }
}

#ifdef COMPOSER_EXPERIMENTS
// Three more forward structures for bf16 / D = 64, built and measured in round 3 (profiles/NEGATIVE_RESULTS.md, "Attention"): none beats
// the kernel above.  They are NOT part of the product library: `python tools/ab_build.py <name> attention.hip -DCOMPOSER_EXPERIMENTS`
// builds a library that has them (COMPOSER_ATTN64=force|pipe|dense selects one), tests/extra/test_gpu_attn64.py runs them.
// =================================================================================================
// forward, bf16 / D = 64, second structure ("fwd64").  What round 2 measured on the kernel above: no pipe is busy (matrix
// 25 %, vector ~30-50 %, LDS 33 %) -- the waves wait: for the register-staged K/V loads (-26 % without them), for the
// staging stores and for a barrier every 16 MFMAs.  Here
//   * a wave owns 64 query rows (two 32-row blocks), so a staged 64-key tile feeds 32 MFMAs per wave between barriers and
//     every K / V fragment read from LDS is used twice;
//   * K and V tiles reach LDS by LDS-DMA (buffer_load ... lds, issued from inline asm so that hipcc's wait-count pass
//     never sees a pending LDS write) into a 3-stage ring: two tiles in flight behind counted s_waitcnt vmcnt, no staging
//     registers, no staging stores, ONE raw s_barrier per tile;
//   * the LDS images are lane-linear (what the DMA writes), 128-byte rows; the bank-conflict swizzles sit on the per-lane
//     SOURCE address and on the read address: K (ds_read_b128 by rows) 16-byte chunk ^ ((row >> 1) & 7), V
//     (ds_read_b64_tr_b16) 64-byte half ^ ((row >> 1) & 1);
//   * 256 threads, 2 workgroups per CU (256 registers per wave, 48 KiB of LDS each): two independent workgroups per SIMD
//     interleave matrix and vector work.
// A workgroup takes a PAIR of 256-row query blocks (heaviest with lightest, as above); inside a block wave w owns the
// 32-row blocks w and 7-w.  Tiles below the block are mask-free; the (up to four) tiles on the diagonal are masked by
// comparison in every wave (masked probabilities are exactly 0), which keeps every accumulating MFMA out of branches.
// =================================================================================================
typedef int a2_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char a2_lds_char;
__device__ __forceinline__ a2_v4i a2_make_srd(const void* base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    a2_v4i d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xFFFFu));
    d[2] = __builtin_amdgcn_readfirstlane((int)min(bytes, (int64_t)0x7FFFFFF0));
    d[3] = 0x00020000;
    return d;
}
// one 1-KiB LDS-DMA piece: LDS [lds_addr, +1024) <- 16 bytes per lane from base + voff + soff (range-checked: zeros past the end)
__device__ __forceinline__ void a2_dma16(a2_v4i srd, uint32_t lds_addr, int voff, int soff) {
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
    for (int i = 0; i < 4; i++) srd[i] = __builtin_amdgcn_readfirstlane(srd[i]);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff)
                 : "memory", "m0");
}
#ifndef A2_STAGES
#define A2_STAGES 3                 // ring stages: A2_STAGES - 1 tiles in flight
#endif
#define A2_IMG 8192                 // one [64 keys][64 d] bf16 image
// A2_DIAG (measurement builds only -- results are wrong; tools/attn_ab.sh): 1 no exponentials, 2 no softmax arithmetic at all,
// 3 no DMA inside the tile loop, 4 no barrier / counted waits, 5 no PV MFMAs, 6 no score MFMAs, 8 one workgroup per CU
#ifndef A2_DIAG
#define A2_DIAG 0
#endif
#if A2_DIAG == 8
#define A2_MINW 1
#else
#define A2_MINW 2
#endif
#define A2_STAGE (2 * A2_IMG)

template <bool DROP>
__global__ __launch_bounds__(256, A2_MINW) void attn_fwd64_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            float* __restrict__ lse, int Tn, int H, float scale, DropCfg drop) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_block(bx, by, H);
    const int b = by / H, hd = by % H;
    const int E = H * 64;
    const int64_t rs = 3 * E;
    const bf16_t* qg = qkv + (int64_t)b * Tn * rs + hd * 64;
    bf16_t* og = o + (int64_t)b * Tn * E + hd * 64;
    const float c2 = scale * LOG2E_F, neg_big = -1e4f / scale;
    const int nb = cdiv(Tn, 256);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(a2_lds_char*)smem_raw;
    const int64_t span = ((int64_t)(Tn - 1) * rs + 64) * 2;
    const a2_v4i srk = a2_make_srd(qg + E, span), srv = a2_make_srd(qg + 2 * E, span);
    const int rsb = (int)(rs * 2);                                       // row stride in bytes
    // this wave's DMA pieces: pieces 2w, 2w+1 of the K image and of the V image (8 rows x 128 B each)
    int kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = 8 * (2 * wave + i) + (lane >> 3), pc = lane & 7;
        kvo[i] = row * rsb + ((pc ^ ((row >> 1) & 7)) << 4);
        vvo[i] = row * rsb + ((pc ^ (((row >> 1) & 1) << 2)) << 4);
    }
    auto issue = [&](int t, int stage) {
        const int soff = t * 64 * rsb;
        const uint32_t kb = lds0 + stage * A2_STAGE + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++) a2_dma16(srk, kb + i * 1024, kvo[i], soff);
#pragma unroll
        for (int i = 0; i < 2; i++) a2_dma16(srv, kb + A2_IMG + i * 1024, vvo[i], soff);
    };
    // fragment read addresses (byte offsets inside an image)
    int koff[4];                                                        // K rows: lane = key row, chunk 2s + h
#pragma unroll
    for (int s = 0; s < 4; s++) koff[s] = (lane & 31) * 128 + (((2 * s + h) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int G = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    int voffr[2];                                                       // V transposed reads: dt = 0, 1
#pragma unroll
    for (int dt = 0; dt < 2; dt++) voffr[dt] = (4 * (G >> 1) + qq) * 128 + ((dt ^ ((qq >> 1) & 1)) << 6) + 32 * (G & 1) + 8 * pp;

    for (int ph = 0; ph < 2; ph++) {
        const int hi = nb - 1 - bx;
        const int qb = ph == 0 ? hi : (bx < hi ? bx : -1);
        if (qb < 0) break;
        int q0[2] = {qb * 256 + 32 * wave, qb * 256 + 32 * (7 - wave)};
        int q[2];
        bool qvalid[2];
        bf16x8 qf[2][4];
        uint32_t rowh[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            q[i] = q0[i] + (lane & 31);
            qvalid[i] = q[i] < Tn;
            load_bfrags<bf16_t, 64>(qf[i], qg, rs, q[i], qvalid[i], h);
            rowh[i] = attn_row_hash(drop.seed, drop.stream, (uint32_t)(by * Tn + q[i]));
        }
        f32x16 oacc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) oacc[i][dt][r] = 0.f;
        float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
        const int kv_end = min(Tn, qb * 256 + 256);
        const int nt = cdiv(kv_end, 64), nint = qb * 4;                  // tiles; the first nint lie below every query row of the block
#pragma unroll
        for (int i = 0; i < A2_STAGES - 1; i++)
            if (i < nt) issue(i, i);

        // one 64-key tile for both 32-row blocks of this wave; MASK (compile time): compare keys with queries / Tn
        auto tile = [&](auto MASKT, const int t, const int st) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(MASKT)::value;
            const char* Ks = smem_raw + st * A2_STAGE;
            const char* Vs = Ks + A2_IMG;
            bf16x8 kf[2][4];
#pragma unroll
            for (int sub = 0; sub < 2; sub++)
#pragma unroll
                for (int s = 0; s < 4; s++) kf[sub][s] = *reinterpret_cast<const bf16x8*>(Ks + sub * 4096 + koff[s]);
            f32x16 sc[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int sub = 0; sub < 2; sub++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) sc[i][sub][r] = 0.f;
#pragma unroll
#if A2_DIAG == 6
                    for (int r = 0; r < 16; r++) sc[i][sub][r] = (float)kf[sub][r & 3][r & 7] * 1e-3f + (float)(t + r);
#else
                    for (int s = 0; s < 4; s++) sc[i][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][s], qf[i][s], sc[i][sub], 0, 0, 0);
#endif
                }
            bf16x8 pb[2][2][2];                                          // [block][sub][k-step]: probabilities as the B operand
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (MASK) {
#pragma unroll
                    for (int sub = 0; sub < 2; sub++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int key = t * 64 + 32 * sub + rho(r, h);
                            if (key > q[i] || key >= Tn) sc[i][sub][r] = neg_big;      // == -1e4 after scaling (transformer.py:354)
                        }
                }
#if A2_DIAG == 2
                if (false) {
#else
                {
#endif
                const float mloc = half_max(fmaxf(max16(sc[i][0]), max16(sc[i][1])));
                const float mnew = fmaxf(m[i], mloc);
                if (!__all(mnew == m[i])) {
                    const float alpha = fast_exp2((m[i] - mnew) * c2);
                    lsum[i] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 2; dt++)
#pragma unroll
                        for (int r = 0; r < 16; r++) oacc[i][dt][r] *= alpha;
                    m[i] = mnew;
                }
                const float mc = m[i] * c2;
                f32x2 ps = {0.f, 0.f};
#if A2_DIAG == 1
#pragma unroll
                for (int r = 0; r < 16; r++) { sc[i][0][r] = fmaf(sc[i][0][r], c2, -mc); sc[i][1][r] = fmaf(sc[i][1][r], c2, -mc); ps[0] += sc[i][0][r]; ps[1] += sc[i][1][r]; }
#else
                ps = exp2_scaled16(sc[i][0], c2, -mc, ps);
                ps = exp2_scaled16(sc[i][1], c2, -mc, ps);
#endif
                lsum[i] += ps[0] + ps[1];
                }
                if constexpr (DROP) {
                    mask16_qlane<true>(sc[i][0], rowh[i], t * 64, h, drop.thr);
                    mask16_qlane<true>(sc[i][1], rowh[i], t * 64 + 32, h, drop.thr);
                }
#pragma unroll
                for (int sub = 0; sub < 2; sub++)
#pragma unroll
                    for (int s = 0; s < 2; s++)
#pragma unroll
                        for (int j = 0; j < 8; j++) pb[i][sub][s][j] = (bf16_t)sc[i][sub][8 * s + j];
            }
#pragma unroll
            for (int sub = 0; sub < 2; sub++)
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int dt = 0; dt < 2; dt++) {
                        const char* va = Vs + (32 * sub + 16 * s) * 128 + voffr[dt];
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(va));
                        const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(va + 8 * 128));
                        bf16x8 a;
#pragma unroll
                        for (int j = 0; j < 4; j++) { a[j] = lo[j]; a[4 + j] = hi4[j]; }
#pragma unroll
                        for (int i = 0; i < 2; i++) {
#if A2_DIAG == 5
                            asm volatile("" ::"v"(a), "v"(pb[i][sub][s]));
                            oacc[i][dt][(sub * 2 + s) & 15] += 1.0f;
#else
                            oacc[i][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[i][sub][s], oacc[i][dt], 0, 0, 0);
#endif
                        }
                    }
        };
        auto step = [&](auto MASKT, const int t) __attribute__((always_inline)) {
            // tile t has landed: this wave's pieces by the counted wait (the younger tiles' pieces may still fly), the others' by the
            // barrier; past the barrier every wave has left tile t-1, whose stage takes tile t + A2_STAGES - 1
#if A2_DIAG != 4
            {
                const int ahead = min(nt - 1 - t, A2_STAGES - 2);       // younger tiles that may stay in flight (4 pieces each)
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
#if A2_DIAG != 3
            if (t + A2_STAGES - 1 < nt) issue(t + A2_STAGES - 1, (t + A2_STAGES - 1) % A2_STAGES);
#endif
            tile(MASKT, t, t % A2_STAGES);
        };
        int t = 0;
        for (; t < nint; t++) step(std::false_type{}, t);
        for (; t < nt; t++) step(std::true_type{}, t);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float ltot = half_sum(lsum[i]);
            const float inv = (DROP ? drop.scale : 1.0f) / ltot;
#pragma unroll
            for (int dt = 0; dt < 2; dt++) store_t_tile<bf16_t, 64>(og, E, q[i], qvalid[i], dt, oacc[i][dt], inv, h);
            if (qvalid[i] && h == 0) lse[(int64_t)by * Tn + q[i]] = m[i] * scale + logf(ltot);
        }
        // the second block's first tiles go into stages the slower waves may still be reading
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// =================================================================================================
// forward, bf16 / D = 64, third structure ("fwd32p"): software-pipelined across 32-key units inside ONE wave.
// The unit of work is a 32-key sub-tile u with three stages -- S_u = K_u Q^T (4 MFMAs), softmax(S_u) -> P_u (vector),
// O += V_u^T P_u (4 MFMAs) -- and the instruction stream of step u is { MFMAs of S_{u+1} and of PV_{u-1} } interleaved with
// { softmax of unit u }: the vector work always has eight independent MFMAs to sit between, and the LDS operands of step u+1
// are read at the end of step u.  A wave owns 32 query rows; K/V tiles come through the LDS-DMA ring of the kernel above with
// FIVE stages (tiles t-1, t, t+1 are read in iteration t; t+2 and t+3 are in flight), one barrier per 64-key tile.
// The online-softmax rescale of O for unit u is applied AFTER PV_{u-1} has been accumulated (P_{u-1} was scaled with the old
// running maximum).  Selected by COMPOSER_ATTN64=pipe; measurements in DESIGN.md "Attention, round 3".
// =================================================================================================
#define A3_STAGES 5
#ifndef A3_PIN
#define A3_PIN 1
#endif
// A3_DIAG (measurement builds only -- results are wrong), a bit mask: 1 no exponentials, 2 no softmax arithmetic, 4 no
// running-maximum head, 8 no barrier / waits / DMA inside the loop, 16 no MFMAs, 32 no V reads from LDS, 64 no K reads from LDS,
// 128 every batch row aliases batch row 0 or 1 (the whole data set stays in L2 / MALL), 256 no output stores, 512 no prologue DMA, 1024 one tile per block
#ifndef A3_DIAG
#define A3_DIAG 0
#endif
#define A3D(bit) ((A3_DIAG & (bit)) != 0)
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd32p_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                             float* __restrict__ lse, int Tn, int H, float scale, DropCfg drop) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_block(bx, by, H);
#if A3D(128)
    const int b = (by / H) & 1, hd = by % H;
#else
    const int b = by / H, hd = by % H;
#endif
    const int E = H * 64;
    const int64_t rs = 3 * E;
    const bf16_t* qg = qkv + (int64_t)b * Tn * rs + hd * 64;
    bf16_t* og = o + (int64_t)b * Tn * E + hd * 64;
    const float c2 = scale * LOG2E_F, neg_big = -1e4f / scale;
    const int nb = cdiv(Tn, 128);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(a2_lds_char*)smem_raw;
    const int64_t span = ((int64_t)(Tn - 1) * rs + 64) * 2;
    const a2_v4i srk = a2_make_srd(qg + E, span), srv = a2_make_srd(qg + 2 * E, span);
    const int rsb = (int)(rs * 2);
    int kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = 8 * (2 * wave + i) + (lane >> 3), pc = lane & 7;
        kvo[i] = row * rsb + ((pc ^ ((row >> 1) & 7)) << 4);
        vvo[i] = row * rsb + ((pc ^ (((row >> 1) & 1) << 2)) << 4);
    }
    auto issue = [&](int t) {
        const int soff = t * 64 * rsb;
        const uint32_t kb = lds0 + (t % A3_STAGES) * A2_STAGE + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++) a2_dma16(srk, kb + i * 1024, kvo[i], soff);
#pragma unroll
        for (int i = 0; i < 2; i++) a2_dma16(srv, kb + A2_IMG + i * 1024, vvo[i], soff);
    };
    int koff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) koff[s] = (lane & 31) * 128 + (((2 * s + h) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int G = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    int voffr[2];
#pragma unroll
    for (int dt = 0; dt < 2; dt++) voffr[dt] = (4 * (G >> 1) + qq) * 128 + ((dt ^ ((qq >> 1) & 1)) << 6) + 32 * (G & 1) + 8 * pp;

    for (int ph = 0; ph < 2; ph++) {
        const int hi = nb - 1 - bx;
        const int qb = ph == 0 ? hi : (bx < hi ? bx : -1);
        if (qb < 0) break;
        const int q = qb * 128 + 32 * wave + (lane & 31);
        const bool qvalid = q < Tn;
        bf16x8 qf[4];
        load_bfrags<bf16_t, 64>(qf, qg, rs, q, qvalid, h);
        const uint32_t rowh = attn_row_hash(drop.seed, drop.stream, (uint32_t)(by * Tn + q));
        f32x16 oacc[2];
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[dt][r] = 0.f;
        float m = -INFINITY, lsum = 0.f;
        const int kv_end = min(Tn, qb * 128 + 128);
#if A3D(1024)
        const int nt = 1, nint = 0;
#else
        const int nt = cdiv(kv_end, 64), nint = qb * 2;                  // tiles; the first nint lie below every query row of the block
#endif
#if !A3D(512)
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (i < nt) issue(i);
#endif

        f32x16 scA, scB;
        bf16x8 pbA[2], pbB[2];
        bf16x8 kf[4], va[2][2];                                             // operands of the NEXT step's MFMAs, read one step ahead
        auto load_k = [&](const char* Kn) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 4; s++) kf[s] = *reinterpret_cast<const bf16x8*>(Kn + koff[s]);
        };
        auto load_v = [&](const char* Vp) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const char* a = Vp + (16 * s) * 128 + voffr[dt];
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a));
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 8 * 128));
#pragma unroll
                    for (int j = 0; j < 4; j++) { va[s][dt][j] = lo[j]; va[s][dt][4 + j] = hi4[j]; }
                }
        };
        // one pipeline step: softmax of `cur` (unit with first key k0) -> pcur, while the matrix pipe computes nxt = K Q^T (kf) and,
        // when PV, O += V^T pprev (va); then the operands of the following step are read (Kn2: its K rows, Vp2: its V rows)
        auto unit = [&](auto MASKT, auto PVT, f32x16& cur, f32x16& nxt, bf16x8 (&pcur)[2], const bf16x8 (&pprev)[2], const char* Kn2,
                        const char* Vp2, const int k0) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(MASKT)::value;
            constexpr bool PV = decltype(PVT)::value;
            if (MASK) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = k0 + rho(r, h);
                    if (key > q || key >= Tn) cur[r] = neg_big;
                }
            }
#if A3D(4)
            const bool need = false;
            float alpha = 1.0f;
            if (m < -1e30f) m = 8.0f;
#else
            const float mloc = half_max(max16(cur));
            const float mnew = fmaxf(m, mloc);
            const bool need = !__all(mnew == m);
            float alpha = 1.0f;
            if (need) {
                alpha = fast_exp2((m - mnew) * c2);
                lsum *= alpha;
                m = mnew;
            }
#endif
            const float mc = m * c2;
            __builtin_amdgcn_sched_barrier(0);
#if A3D(16)
#pragma unroll
            for (int r = 0; r < 16; r++) nxt[r] = (float)kf[r & 3][r & 7] * 1e-3f + (float)(k0 + r);
            if (PV) {
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int dt = 0; dt < 2; dt++) { asm volatile("" ::"v"(va[s][dt]), "v"(pprev[s])); oacc[dt][(2 * s + dt) & 15] += 1.0f; }
            }
#else
#pragma unroll
            for (int r = 0; r < 16; r++) nxt[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; s++) nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], nxt, 0, 0, 0);
            if (PV) {
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int dt = 0; dt < 2; dt++) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[s][dt], pprev[s], oacc[dt], 0, 0, 0);
            }
#endif
#if A3D(2)
            f32x2 ps = {cur[0], cur[1]};
#elif A3D(1)
            f32x2 ps = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r++) { cur[r] = fmaf(cur[r], c2, -mc); ps[r & 1] += cur[r]; }
#else
            f32x2 ps = exp2_scaled16(cur, c2, -mc, f32x2{0.f, 0.f});
#endif
            lsum += ps[0] + ps[1];
#if !A3D(2)
            if constexpr (DROP) mask16_qlane<true>(cur, rowh, k0, h, drop.thr);
#endif
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) pcur[s][j] = (bf16_t)cur[8 * s + j];
#if A3_PIN
            // pinned interleave: one MFMA per slice of the vector work
#pragma unroll
            for (int i = 0; i < (PV ? 8 : 4); i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, DROP ? (PV ? 12 : 24) : (PV ? 5 : 10), 0);
                __builtin_amdgcn_sched_group_barrier(0x400, PV ? 2 : 4, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(pcur[0]), "+v"(pcur[1]), "+v"(lsum));     // keeps the vector work of this unit inside this block
#if !A3D(64)
            load_k(Kn2);
#endif
#if !A3D(32)
            load_v(Vp2);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (need) {                                                     // rescale O AFTER P_{u-1} (old maximum) went in
#pragma unroll
                for (int dt = 0; dt < 2; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
            }
        };
        auto iter = [&](auto MASKT, auto FIRSTT, const int t) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(FIRSTT)::value;
            // tiles <= t + 1 have landed (own pieces: counted wait; the others': the barrier); tile t + 2 may still fly.  The LDS reads
            // in flight here (operands of step A, tiles t-1 and t) do not touch the stage that is re-filled below (tile t-2's).
#if !A3D(8)
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 3 < nt) issue(t + 3);
#endif
            const char* St = smem_raw + (t % A3_STAGES) * A2_STAGE;
            const char* Sn = smem_raw + ((t + 1) % A3_STAGES) * A2_STAGE;
            // step A: unit 2t (S_{2t+1} from K(t) rows 32.., PV of unit 2t-1); then read step B's operands: K(t+1) rows 0.., V(t) rows 0..
            unit(MASKT, std::integral_constant<bool, !FIRST>{}, scA, scB, pbA, pbB, Sn, St + A2_IMG, t * 64);
            // step B: unit 2t+1; then step A(t+1)'s operands: K(t+1) rows 32.., V(t) rows 32..
            unit(MASKT, std::true_type{}, scB, scA, pbB, pbA, Sn + 4096, St + A2_IMG + 32 * 128, t * 64 + 32);
        };
        // prologue: S_0, and the operands of step A(0): K(0) rows 32.. (no PV in the very first step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_k(smem_raw);
#pragma unroll
        for (int r = 0; r < 16; r++) scA[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; s++) scA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], scA, 0, 0, 0);
        load_k(smem_raw + 4096);
        load_v(smem_raw + A2_IMG);                                          // unused by the first step (finite or not: never multiplied)
        if (nint > 0) iter(std::false_type{}, std::true_type{}, 0);
        else iter(std::true_type{}, std::true_type{}, 0);
        int t = 1;
        for (; t < nint; t++) iter(std::false_type{}, std::false_type{}, t);
        for (; t < nt; t++) iter(std::true_type{}, std::false_type{}, t);
        // drain: PV of the last unit (its V rows were read by the last step)
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int dt = 0; dt < 2; dt++) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[s][dt], pbB[s], oacc[dt], 0, 0, 0);
        const float ltot = half_sum(lsum);
        const float inv = (DROP ? drop.scale : 1.0f) / ltot;
#if A3D(256)
        if (inv == 123.456f)
#endif
        {
#pragma unroll
            for (int dt = 0; dt < 2; dt++) store_t_tile<bf16_t, 64>(og, E, q, qvalid, dt, oacc[dt], inv, h);
            if (qvalid && h == 0) lse[(int64_t)by * Tn + q] = m * scale + logf(ltot);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// =================================================================================================
// forward, bf16 / D = 64, fourth structure ("fwd32d"): the occupancy experiment.  The round-2 tile body (a 32-key unit at a time:
// 4 score MFMAs, softmax, 4 PV MFMAs; 32 query rows per wave) fed by the LDS-DMA ring instead of register-staged loads, so that
// the per-lane state fits 128 registers and FOUR workgroups share a CU (4 waves per SIMD; two ring stages = 32 KiB each).
// Selected by COMPOSER_ATTN64=dense.
// =================================================================================================
#ifndef A4_MINW
#define A4_MINW 4
#endif
#ifndef A4_STAGES
#define A4_STAGES 2
#endif
template <bool DROP>
__global__ __launch_bounds__(256, A4_MINW) void attn_fwd32d_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                                  float* __restrict__ lse, int Tn, int H, float scale, DropCfg drop) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    xcd_block(bx, by, H);
    const int b = by / H, hd = by % H;
    const int E = H * 64;
    const int64_t rs = 3 * E;
    const bf16_t* qg = qkv + (int64_t)b * Tn * rs + hd * 64;
    bf16_t* og = o + (int64_t)b * Tn * E + hd * 64;
    const float c2 = scale * LOG2E_F, neg_big = -1e4f / scale;
    const int nb = cdiv(Tn, 128);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(a2_lds_char*)smem_raw;
    const int64_t span = ((int64_t)(Tn - 1) * rs + 64) * 2;
    const a2_v4i srk = a2_make_srd(qg + E, span), srv = a2_make_srd(qg + 2 * E, span);
    const int rsb = (int)(rs * 2);
    int kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int row = 8 * (2 * wave + i) + (lane >> 3), pc = lane & 7;
        kvo[i] = row * rsb + ((pc ^ ((row >> 1) & 7)) << 4);
        vvo[i] = row * rsb + ((pc ^ (((row >> 1) & 1) << 2)) << 4);
    }
    auto issue = [&](int t) {
        const int soff = t * 64 * rsb;
        const uint32_t kb = lds0 + (t % A4_STAGES) * A2_STAGE + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; i++) a2_dma16(srk, kb + i * 1024, kvo[i], soff);
#pragma unroll
        for (int i = 0; i < 2; i++) a2_dma16(srv, kb + A2_IMG + i * 1024, vvo[i], soff);
    };
    int koff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) koff[s] = (lane & 31) * 128 + (((2 * s + h) ^ (((lane & 31) >> 1) & 7)) << 4);
    const int G = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    int voffr[2];
#pragma unroll
    for (int dt = 0; dt < 2; dt++) voffr[dt] = (4 * (G >> 1) + qq) * 128 + ((dt ^ ((qq >> 1) & 1)) << 6) + 32 * (G & 1) + 8 * pp;

    for (int ph = 0; ph < 2; ph++) {
        const int hi = nb - 1 - bx;
        const int qb = ph == 0 ? hi : (bx < hi ? bx : -1);
        if (qb < 0) break;
        const int q0w = qb * 128 + 32 * wave;
        const int q = q0w + (lane & 31);
        const bool qvalid = q < Tn;
        bf16x8 qf[4];
        load_bfrags<bf16_t, 64>(qf, qg, rs, q, qvalid, h);
        const uint32_t rowh = attn_row_hash(drop.seed, drop.stream, (uint32_t)(by * Tn + q));
        f32x16 oacc[2];
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[dt][r] = 0.f;
        float m = -INFINITY, lsum = 0.f;
        const int kv_end = min(Tn, qb * 128 + 128);
        const int nt = cdiv(kv_end, 64), nint = qb * 2;
#pragma unroll
        for (int i = 0; i < A4_STAGES - 1; i++)
            if (i < nt) issue(i);

        // one 32-key unit, complete: scores, online softmax, PV.  MASK (compile time): compare keys with queries / Tn
        auto unit = [&](auto MASKT, const char* Ku, const char* Vu, const int k0) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(MASKT)::value;
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; r++) sc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ku + koff[s]);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc, 0, 0, 0);
            }
            if (MASK) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = k0 + rho(r, h);
                    if (key > q || key >= Tn) sc[r] = neg_big;
                }
            }
            const float mloc = half_max(max16(sc));
            const float mnew = fmaxf(m, mloc);
            if (!__all(mnew == m)) {
                const float alpha = fast_exp2((m - mnew) * c2);
                lsum *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
                m = mnew;
            }
            const float mc = m * c2;
            const f32x2 ps = exp2_scaled16(sc, c2, -mc, f32x2{0.f, 0.f});
            lsum += ps[0] + ps[1];
            if constexpr (DROP) mask16_qlane<true>(sc, rowh, k0, h, drop.thr);
            bf16x8 pb[2];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) pb[s][j] = (bf16_t)sc[8 * s + j];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const char* a = Vu + (16 * s) * 128 + voffr[dt];
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a));
                    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 8 * 128));
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 4; j++) { v[j] = lo[j]; v[4 + j] = hi4[j]; }
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, pb[s], oacc[dt], 0, 0, 0);
                }
        };
        auto step = [&](auto MASKT, const int t) __attribute__((always_inline)) {
            // tile t has landed (own pieces: counted wait; the others': the barrier); every wave has left tile t-1, whose stage is re-filled
            {
                const int ahead = min(nt - 1 - t, A4_STAGES - 2);
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + A4_STAGES - 1 < nt) issue(t + A4_STAGES - 1);
            const char* Ks = smem_raw + (t % A4_STAGES) * A2_STAGE;
            const char* Vs = Ks + A2_IMG;
            unit(MASKT, Ks, Vs, t * 64);
            // the second unit of a diagonal tile lies entirely above the diagonal for the waves whose rows end below it
            if (!decltype(MASKT)::value || t * 64 + 32 <= q0w + 31) unit(MASKT, Ks + 4096, Vs + 32 * 128, t * 64 + 32);
        };
        int t = 0;
        for (; t < nint; t++) step(std::false_type{}, t);
        for (; t < nt; t++) {
            if (t * 64 <= q0w + 31) step(std::true_type{}, t);
            else {                                                          // nothing to do in this tile: keep the barrier and the ring going
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (t + A4_STAGES - 1 < nt) issue(t + A4_STAGES - 1);
            }
        }
        const float ltot = half_sum(lsum);
        const float inv = (DROP ? drop.scale : 1.0f) / ltot;
#pragma unroll
        for (int dt = 0; dt < 2; dt++) store_t_tile<bf16_t, 64>(og, E, q, qvalid, dt, oacc[dt], inv, h);
        if (qvalid && h == 0) lse[(int64_t)by * Tn + q] = m * scale + logf(ltot);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

#endif  // COMPOSER_EXPERIMENTS

// =================================================================================================
// dQ.  same geometry as forward: dQ^T += K^T . dS^T,  dS^T = P^T * (dP^T - delta),  dP^T = V . dO^T
// With dropout (keep-scale f = 1/(1-p)):  dS = f * P * (M*dP~ - delta/f)  -> the f goes to the output scale.
// =================================================================================================
// KS: as in the forward -- one 32-row query block per workgroup, the waves split every staged 256-key tile, dQ partials summed in LDS.
// (the body is shared by attn_dq_kernel and the small-grid backward launch attn_bwd_ks_kernel, which passes its block in ks_blk)
template <typename T, int D, bool DROP, int KS>
__device__ __forceinline__ void attn_dq_body(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                             const float* __restrict__ lse, float* __restrict__ delta, T* __restrict__ dqkv,
                                             float* __restrict__ bias_grad, int Tn, int H, float scale, DropCfg drop, int plan_u,
                                             int ks_blk) {
    using G = Geo<T, D>;
    constexpr bool EXACT = AT<T>::EXACT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Kb = reinterpret_cast<T*>(smem_raw);          // 2 x { K | V }
    static_assert(!KS || !EXACT, "key split: throughput mode");
    constexpr int KT = KS ? 256 : 64, QR = KS ? 32 : 128;
    constexpr int IMG = KT * G::S;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = KS ? wave * 64 : 0;
    const int nb = cdiv(Tn, 128);
    int by, qb0, qb1;
    if constexpr (KS) {
        by = blockIdx.y;
        qb0 = ks_blk;
        qb1 = -1;
    } else {
        attn_job(plan_u, nb, H, by, qb0, qb1);
    }
    const int b = by / H, hd = by % H;
    const int E = H * D;
    const int64_t rs = 3 * E;
    const T* qg = qkv + (int64_t)b * Tn * rs + hd * D;
    const T* kg = qg + E;
    const T* vg = qg + 2 * E;
    const T* dog = d_o + (int64_t)b * Tn * E + hd * D;
    T* dqg = dqkv + (int64_t)b * Tn * rs + hd * D;
    const float c2 = scale * LOG2E_F;
    const float keep_scale = DROP ? drop.scale : 1.0f;

    for (int ph = 0; ph < 2; ph++) {
#ifdef ATTN_LIGHT_FIRST
        // the pair's LIGHT query block first: the four pair-workgroups of a (batch, head) row then start on key tile 0 together and
        // their heavy blocks follow one tile apart -- a sliding window of the row's K/V stays in the XCD's L2 instead of the light
        // blocks re-reading the first tiles long after the heavy ones streamed past them
        const int qb = qb1 < 0 ? (ph == 0 ? qb0 : -1) : (ph == 0 ? qb1 : qb0);
#else
        const int qb = ph == 0 ? qb0 : qb1;
#endif
        if (qb < 0) break;
        const int q0w = KS ? qb * 32 : qb * 128 + wave * 32;
        const int q = q0w + (lane & 31);
        const bool qvalid = q < Tn;
        frag_t<T> qf[G::NS], dof[G::NS];
        load_bfrags<T, D>(qf, qg, rs, q, qvalid, h);
        load_bfrags<T, D>(dof, dog, E, q, qvalid, h);
        const float lse_q = qvalid ? lse[(int64_t)by * Tn + q] : 0.f;
        // delta[q] = rowsum(dO . O) is computed here -- this lane already holds its half of the dO row
        // -- and stored for the dK/dV kernel that runs next on the stream (this replaced a separate 52 us pass per layer).
        // The O row is requested now and consumed after the first K/V tile has been staged, under that latency.
        frag_t<T> of[G::NS];
        load_bfrags<T, D>(of, o + (int64_t)b * Tn * E + hd * D, E, q, qvalid, h);
        const float lse2 = lse_q * LOG2E_F;
        f32x16 dq[G::DT];
#pragma unroll
        for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) dq[dt][r] = 0.f;
        const int kv_end = min(Tn, qb * QR + QR);
        const uint32_t rowh = attn_row_hash(drop.seed, drop.stream, (uint32_t)(by * Tn + q));

        Stager<T, D, KT> sk, sv;
        sk.load(kg, rs, 0, Tn, tid);
        sv.load(vg, rs, 0, Tn, tid);
        sk.store(Kb, tid);
        sv.store(Kb + IMG, tid);
        float dsum = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < G::NS; sidx++) {
            if constexpr (std::is_same<T, float>::value) {
                dsum += of[sidx] * dof[sidx];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) dsum += (float)of[sidx][j] * (float)dof[sidx][j];
            }
        }
        dsum = half_sum(dsum);
        if (qvalid && h == 0 && (!KS || wave == 0)) delta[(int64_t)by * Tn + q] = dsum;
        const float del_q = (qvalid ? dsum : 0.f) / keep_scale;
        __syncthreads();
        // One 64-key tile.  `interior` (compile time) = every key of the tile lies at or below every query row of the block and
        // every row and key exists: that instance carries NO mask code.  (Round 3 had the mask in one loop under a wave-uniform
        // `if (edge)`: hipcc if-converted it -- 32 integer compares + 16 selects + ~17 scalar mask operations in the block of the
        // score MFMAs of EVERY sub-tile, a fifth of the kernel's vector instructions, needed on two tiles in eighteen.)
        auto tile = [&](auto interior, const int kts, const int it) __attribute__((always_inline)) {
            constexpr bool INT = decltype(interior)::value;
            const T* Ks = Kb + (it & 1) * 2 * IMG + wk * G::S;
            const T* Vs = Ks + IMG;
            const bool more = kts + KT < kv_end;
            const int kt0 = kts + wk;                 // first key of this wave's 64
            if (more) {
                sk.load(kg, rs, kts + KT, Tn, tid);
                sv.load(vg, rs, kts + KT, Tn, tid);
            }
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const int k0 = kt0 + 32 * sub;
                if (!INT && (k0 > q0w + 31 || k0 >= Tn)) continue;
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
                s = mma_rows<T, D>(Ks, 32 * sub, qf, lane, s);
                dp = mma_rows<T, D>(Vs, 32 * sub, dof, lane, dp);
                if constexpr (DROP) mask16_qlane(dp, rowh, k0, h, drop.thr);
                if constexpr (EXACT) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        int key = k0 + rho(r, h);
                        bool masked = (key > q) || (key >= Tn) || !qvalid;
                        float p = masked ? 0.f : expf(s[r] * scale - lse_q);
                        s[r] = p * (dp[r] - del_q);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) s[r] = fast_exp2(fmaf(s[r], c2, -lse2));
                    if constexpr (!INT) {
                        const bool edge = (k0 + 31 > q0w) || (k0 + 32 > Tn) || (q0w + 32 > Tn);      // wave-uniform
                        if (edge) {
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int key = k0 + rho(r, h);
                                if ((key > q) || (key >= Tn) || !qvalid) s[r] = 0.f;
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) s[r] *= dp[r] - del_q;
                }
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++) dq[dt] = mma_acc_b<T, D>(Ks, 32 * sub, dt, s, lane, dq[dt]);
            }
            if (more) {
                T* nbuf = Kb + ((it & 1) ^ 1) * 2 * IMG;
                sk.store(nbuf, tid);
                sv.store(nbuf + IMG, tid);
            }
            __syncthreads();
        };
        int kt0 = 0, it = 0;
        const int interior_end = (EXACT || qb * QR + QR > Tn) ? 0 : (qb * QR) / KT * KT;     // whole tiles below every (existing) query row of the block
        for (; kt0 < interior_end; kt0 += KT, it++) tile(std::true_type{}, kt0, it);
        for (; kt0 < kv_end; kt0 += KT, it++) tile(std::false_type{}, kt0, it);
        if constexpr (KS) {
            // sum of the four waves' partial dQ^T tiles (the last tile's barrier has passed: the staging buffers are free)
            float* mg = reinterpret_cast<float*>(smem_raw);
            if (wave > 0) {
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) mg[(((wave - 1) * G::DT + dt) * 16 + r) * 64 + lane] = dq[dt][r];
            }
            __syncthreads();
            if (wave > 0) continue;               // (single phase: leaves the loop)
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) dq[dt][r] += mg[((w * G::DT + dt) * 16 + r) * 64 + lane];
        }
#pragma unroll
        for (int dt = 0; dt < G::DT; dt++) store_t_tile<T, D>(dqg, rs, q, qvalid, dt, dq[dt], scale * keep_scale, h);
        if (bias_grad) colsum_t_tiles<T, D>(bias_grad + hd * D, qvalid, dq, scale * keep_scale, h, lane);
    }
}
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(256, (Occ<T, D>::MINW_Q))
void attn_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o, const float* __restrict__ lse,
                    float* __restrict__ delta, T* __restrict__ dqkv, float* __restrict__ bias_grad, int Tn, int H, float scale,
                    DropCfg drop, int plan_u) {
    attn_dq_body<T, D, DROP, 0>(qkv, o, d_o, lse, delta, dqkv, bias_grad, Tn, H, scale, drop, plan_u, 0);
}

// =================================================================================================
// dK, dV.  grid ((nb+1)/2, B*H): wave w owns keys [kb*128 + 32w, +32); loops over the query tiles at or below the
// diagonal.  Key block j needs nb-j query blocks -> paired (j, nb-1-j) like the forward.
// =================================================================================================
// KS (small grids): one 32-key block per workgroup (grid (ceil(T/32), B*H), key block 0 = the heaviest first), the four waves take
// the four 64-query quarters of every staged 256-query tile, dK / dV partials summed in LDS.
// KS == 2 takes delta[q] = rowsum(dO . O) from `o` itself (a 32- or 64-byte row per thread and staged tile) instead of from the dQ
// kernel: the two roles of a small backward then run side by side in ONE launch (attn_bwd_ks_kernel).
template <typename T, int D, bool DROP, int KS>
__device__ __forceinline__ void attn_dkv_body(const T* __restrict__ qkv, const T* __restrict__ d_o, const float* __restrict__ lse,
                                              const float* __restrict__ delta, T* __restrict__ dqkv, float* __restrict__ bias_grad,
                                              int Tn, int H, float scale, DropCfg drop, const T* __restrict__ o, int ks_blk) {
    using G = Geo<T, D>;
    constexpr bool EXACT = AT<T>::EXACT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* Qb = reinterpret_cast<T*>(smem_raw);          // 2 x { Q [QT][S] | dO [QT][S] }
    static_assert(!KS || !EXACT, "key split: throughput mode");
    constexpr int QT = KS ? 256 : 64;                // queries per staged tile
    constexpr int IMG = QT * G::S;
    float* Lb = reinterpret_cast<float*>(Qb + (KS ? 2 : 4) * IMG);     // 2 x { lse [QT] | delta/f [QT] | dropout row hash [QT] } (KS: 1 x)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = KS ? wave * 64 : 0;               // this wave's queries inside a staged tile
    int bx, by;
    if constexpr (KS) {
        bx = ks_blk;
        by = blockIdx.y;
    } else {
        xcd_block(bx, by, H);
    }
    const int b = by / H, hd = by % H;
    const int E = H * D;
    const int64_t rs = 3 * E;
    const T* qg = qkv + (int64_t)b * Tn * rs + hd * D;
    const T* kg = qg + E;
    const T* vg = qg + 2 * E;
    const T* dog = d_o + (int64_t)b * Tn * E + hd * D;
    T* dkg = dqkv + (int64_t)b * Tn * rs + E + hd * D;
    T* dvg = dkg + E;
    const float c2 = scale * LOG2E_F;
    const float keep_scale = DROP ? drop.scale : 1.0f;
    const int nb = cdiv(Tn, 128);

    for (int ph = 0; ph < 2; ph++) {
        // light/heavy are mirrored w.r.t. the forward: key block 0 is the heavy one
        const int kb = (KS || ((int)gridDim.x == nb && nb > 1)) ? (ph == 0 ? bx : -1) : (ph == 0 ? bx : (bx < nb - 1 - bx ? nb - 1 - bx : -1));
        if (kb < 0) break;
        const int k0w = KS ? kb * 32 : kb * 128 + wave * 32;
        const int key = k0w + (lane & 31);
        const bool kvalid = key < Tn;
        frag_t<T> kf[G::NS], vf[G::NS];
        load_bfrags<T, D>(kf, kg, rs, key, kvalid, h);
        load_bfrags<T, D>(vf, vg, rs, key, kvalid, h);
        const uint32_t kgG = ((uint32_t)key >> 2) * ATTN_G;
        const uint32_t kC = (key & 3) == 0 ? ATTN_C0 : ((key & 3) == 1 ? ATTN_C1 : ((key & 3) == 2 ? ATTN_C2 : ATTN_C3));
        f32x16 dk[G::DT], dv[G::DT];
#pragma unroll
        for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

        Stager<T, D, QT> sq, so;
        float st_l = 0.f, st_d = 0.f;
        constexpr int ORW = KS == 2 ? D / AT<T>::VN : 1;
        Vec16<T> st_o[ORW], st_g[ORW];           // KS == 2: the thread's O and dO row of the next tile (delta is their dot product)
        const T* og_ks = KS == 2 ? o + (int64_t)b * Tn * E + hd * D : nullptr;
        // per-row scalars of tile qt (threads 0..63): raw loads only -- any arithmetic on the loaded value here would make
        // the compiler wait for it (and for the tile loads issued before it) ahead of the current tile's MFMAs; the
        // scale factors are applied in store_rows, after the tile's work.  Rows past Tn read a clamped (finite) row;
        // their probabilities are masked to zero.
        // Every wave issues them (same 256 bytes, L1 hits): with the loads under `if (tid < 64)` the wait-count pass
        // sized the waits in front of the MFMAs for the shorter path and made the other wave stall on a tile load.
        auto load_rows = [&](int qt) {
            const int qq = min(qt + (tid & (QT - 1)), Tn - 1);
            st_l = lse[(int64_t)by * Tn + qq];
            if constexpr (KS == 2) {
#pragma unroll
                for (int i = 0; i < ORW; i++) {
                    st_o[i] = ld16(og_ks + (int64_t)qq * E + i * AT<T>::VN);
                    st_g[i] = ld16(dog + (int64_t)qq * E + i * AT<T>::VN);
                }
            } else {
                st_d = delta[(int64_t)by * Tn + qq];
            }
        };
        auto store_rows = [&](float* dst, int qt) {
            if constexpr (KS == 2) {
                st_d = 0.f;
#pragma unroll
                for (int i = 0; i < ORW; i++)
#pragma unroll
                    for (int j = 0; j < AT<T>::VN; j++) st_d = fmaf(st_o[i].get(j), st_g[i].get(j), st_d);
            }
            if (tid < QT) {
                dst[tid] = st_l * (EXACT ? 1.0f : LOG2E_F);
                dst[QT + tid] = st_d / keep_scale;
                reinterpret_cast<uint32_t*>(dst)[2 * QT + tid] = attn_row_hash(drop.seed, drop.stream, (uint32_t)(by * Tn + qt + tid));
            }
        };
        const int qstart = KS ? (kb * 32) / 64 * 64 : kb * 128;
        sq.load(qg, rs, qstart, Tn, tid);
        so.load(dog, E, qstart, Tn, tid);
        load_rows(qstart);
        sq.store(Qb, tid);
        so.store(Qb + IMG, tid);
        store_rows(Lb, qstart);
        __syncthreads();
        // One 64-query tile.  `interior` (compile time) = every query row of the tile lies below every key of the block and every
        // row and key exists: no mask code in that instance (see the dQ kernel: the if-converted mask was a fifth of the vector
        // instructions of every sub-tile).
        auto tile = [&](auto interior, const int qts, const int it) __attribute__((always_inline)) {
            constexpr bool INT = decltype(interior)::value;
            // KS stages into ONE buffer (a barrier in front of the refill): 43 KiB of LDS at D = 16, three workgroups per CU -- the
            // double-buffered 86 KiB left one per CU and the 512 workgroups of the default configuration took two rounds
            const int sb = KS ? 0 : (it & 1);
            const T* Qs = Qb + sb * 2 * IMG + wq * G::S;
            const T* Os = Qs + IMG;
            const float* Ls = Lb + sb * 3 * QT + wq;
            const bool more = qts + QT < Tn;
            const int qt0 = qts + wq;                 // first query of this wave's 64
            if (more) {
                sq.load(qg, rs, qts + QT, Tn, tid);
                so.load(dog, E, qts + QT, Tn, tid);
                load_rows(qts + QT);
            }
#pragma unroll
            for (int sub = 0; sub < 2; sub++) {
                const int qb0 = qt0 + 32 * sub;
                if (!INT && (qb0 + 31 < k0w || qb0 >= Tn || k0w >= Tn)) continue;     // wave-uniform: all (q, key) pairs masked
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
                s = mma_rows<T, D>(Qs, 32 * sub, kf, lane, s);              // S[q][key]
                dp = mma_rows<T, D>(Os, 32 * sub, vf, lane, dp);            // dP~[q][key]
                // row constants of the 16 query rows this lane sees: rows 8g+4h .. +3 are consecutive -> 16-byte reads
                f32x4 lq[4], dq4[4];
                __attribute__((ext_vector_type(4))) uint32_t rh4[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int ql = 32 * sub + 8 * g + 4 * h;
                    lq[g] = *reinterpret_cast<const f32x4*>(Ls + ql);
                    dq4[g] = *reinterpret_cast<const f32x4*>(Ls + QT + ql);
                    if constexpr (DROP) rh4[g] = *reinterpret_cast<const __attribute__((ext_vector_type(4))) uint32_t*>(Ls + 2 * QT + ql);
                }
                f32x16 pt;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if constexpr (EXACT) pt[r] = expf(s[r] * scale - lq[r >> 2][r & 3]);
                    else pt[r] = fast_exp2(fmaf(s[r], c2, -lq[r >> 2][r & 3]));        // lse*log2(e) in this mode
                }
                if constexpr (!INT) {
                    const bool edge = EXACT || (qb0 < k0w + 32) || (qb0 + 32 > Tn) || (k0w + 32 > Tn);   // wave-uniform
                    if (edge) {
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int qq = qt0 + 32 * sub + rho(r, h);
                            if ((key > qq) || (qq >= Tn) || !kvalid) pt[r] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float p = pt[r];
                    if constexpr (EXACT || !DROP) {
                        float dpv = dp[r];
                        float pd = p;
                        if constexpr (DROP) {
                            const bool keep = __umul24(rh4[r >> 2][r & 3] ^ kgG, kC) >= drop.thr;
                            dpv = keep ? dpv : 0.f;
                            pd = keep ? p : 0.f;
                        }
                        pt[r] = pd;                                   // (dropped) probabilities feed dV
                        s[r] = p * (dpv - dq4[r >> 2][r & 3]);        // dS/f feeds dK
                    } else {
                        // p (M dP~ - delta/f) = (M p) dP~ - p delta/f: ONE select (the masked probability, which dV needs anyway), a
                        // multiply and an fma instead of two selects, a subtract and a multiply
                        const bool keep = __umul24(rh4[r >> 2][r & 3] ^ kgG, kC) >= drop.thr;
                        const float pd = keep ? p : 0.f;
                        pt[r] = pd;
                        s[r] = fmaf(pd, dp[r], -(p * dq4[r >> 2][r & 3]));
                    }
                }
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++) {
                    dv[dt] = mma_acc_b<T, D>(Os, 32 * sub, dt, pt, lane, dv[dt]);
                    dk[dt] = mma_acc_b<T, D>(Qs, 32 * sub, dt, s, lane, dk[dt]);
                }
            }
            if (more) {
                if constexpr (KS) __syncthreads();
                T* nbuf = Qb + (KS ? 0 : (sb ^ 1)) * 2 * IMG;
                sq.store(nbuf, tid);
                so.store(nbuf + IMG, tid);
                store_rows(Lb + (KS ? 0 : (sb ^ 1)) * 3 * QT, qts + QT);
            }
            __syncthreads();
        };
        // the diagonal block's tiles (masked), then the tiles below it (interior), then a ragged last tile (masked)
        int qt0 = qstart, it = 0;
        const int diag_end = EXACT ? Tn : min(Tn, KS ? k0w + 32 : qstart + 128);
        for (; qt0 < diag_end; qt0 += QT, it++) tile(std::false_type{}, qt0, it);
        for (; qt0 + QT <= Tn; qt0 += QT, it++) tile(std::true_type{}, qt0, it);
        for (; qt0 < Tn; qt0 += QT, it++) tile(std::false_type{}, qt0, it);
        if constexpr (KS) {
            // sum of the four waves' partial dK^T / dV^T tiles (the last tile's barrier has passed: the staging buffers are free)
            float* mg = reinterpret_cast<float*>(smem_raw);
            if (wave > 0) {
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        mg[(((wave - 1) * 2 * G::DT + 2 * dt) * 16 + r) * 64 + lane] = dk[dt][r];
                        mg[(((wave - 1) * 2 * G::DT + 2 * dt + 1) * 16 + r) * 64 + lane] = dv[dt][r];
                    }
            }
            __syncthreads();
            if (wave > 0) continue;               // (single phase: leaves the loop)
#pragma unroll
            for (int w = 0; w < 3; w++)
#pragma unroll
                for (int dt = 0; dt < G::DT; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        dk[dt][r] += mg[((w * 2 * G::DT + 2 * dt) * 16 + r) * 64 + lane];
                        dv[dt][r] += mg[((w * 2 * G::DT + 2 * dt + 1) * 16 + r) * 64 + lane];
                    }
        }
#pragma unroll
        for (int dt = 0; dt < G::DT; dt++) {
            store_t_tile<T, D>(dkg, rs, key, kvalid, dt, dk[dt], scale * keep_scale, h);
            store_t_tile<T, D>(dvg, rs, key, kvalid, dt, dv[dt], keep_scale, h);
        }
        if (bias_grad) {
            colsum_t_tiles<T, D>(bias_grad + E + hd * D, kvalid, dk, scale * keep_scale, h, lane);
            colsum_t_tiles<T, D>(bias_grad + 2 * E + hd * D, kvalid, dv, keep_scale, h, lane);
        }
    }
}
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(256, (Occ<T, D>::MINW)) void attn_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                                          const float* __restrict__ lse,
                                                                          const float* __restrict__ delta, T* __restrict__ dqkv,
                                                                          float* __restrict__ bias_grad, int Tn, int H, float scale,
                                                                          DropCfg drop) {
    attn_dkv_body<T, D, DROP, 0>(qkv, d_o, lse, delta, dqkv, bias_grad, Tn, H, scale, drop, nullptr, 0);
}
// The whole backward of a small grid in one launch: grid (2 * ceil(T/32), B*H); even x = dK/dV of key block x/2, odd x = dQ of query
// block ceil(T/32) - 1 - x/2 -- the heaviest blocks of both roles first, the lightest last, so the second round of workgroups (two
// fit a CU) fills the slots the light ones leave.  (Side by side on two streams the pair lost to its event fork / join, NEGATIVE_RESULTS.)
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_ks_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                                             const float* __restrict__ lse, float* __restrict__ delta, T* __restrict__ dqkv,
                                                             float* __restrict__ bias_grad, int Tn, int H, float scale, DropCfg drop) {
    const int x = blockIdx.x, nqb = (int)gridDim.x >> 1;
    if (x & 1) attn_dq_body<T, D, DROP, 1>(qkv, o, d_o, lse, delta, dqkv, bias_grad, Tn, H, scale, drop, -1, nqb - 1 - (x >> 1));
    else attn_dkv_body<T, D, DROP, 2>(qkv, d_o, lse, delta, dqkv, bias_grad, Tn, H, scale, drop, o, x >> 1);
}
// ... and as two launches where the merged grid would not be resident at once (the default configuration at T = 1024: 1024 workgroups on
// 512 slots took 59 us merged against 19 + 29 us apart); dK/dV then reads the delta the dQ launch stored.
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_dq_ks_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ d_o,
                                                            const float* __restrict__ lse, float* __restrict__ delta, T* __restrict__ dqkv,
                                                            float* __restrict__ bias_grad, int Tn, int H, float scale, DropCfg drop) {
    attn_dq_body<T, D, DROP, 1>(qkv, o, d_o, lse, delta, dqkv, bias_grad, Tn, H, scale, drop, -1, (int)gridDim.x - 1 - (int)blockIdx.x);
}
template <typename T, int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_dkv_ks_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv, float* __restrict__ bias_grad,
                                                             int Tn, int H, float scale, DropCfg drop) {
    attn_dkv_body<T, D, DROP, 1>(qkv, d_o, lse, delta, dqkv, bias_grad, Tn, H, scale, drop, nullptr, (int)blockIdx.x);
}

// =================================================================================================
// host launchers
// =================================================================================================
int colsum_run(void* stream, const void* X, int ldx, float* out, int rows, int cols, int dtype, float* det_ws, size_t det_ws_bytes);   // elementwise.hip
// grid.x: block pairs, or single blocks when the paired grid would leave most of the 256 CUs x 2-3 workgroups without work
static int attn_grid_x(int Tn, int BH) {
    const int nb = cdiv(Tn, 128), pairs = (nb + 1) / 2;
    return (nb > 1 && (int64_t)pairs * BH < 512) ? nb : pairs;
}
// The block plan of a forward / dQ launch (attn_job): how many (batch, head) rows per XCD run as single blocks behind the paired
// ones.  An XCD's workgroups start in order on 32 CUs x wgs_per_cu slots; the paired rows fill whole rounds of those slots and
// the rows of the last, partial round go unpaired (with less than one round of pairs: three quarters of the rows).  Measured
// (tools/ubench/attn_plan_probe.py, outputs bit-identical): B*H = 256: forward 73.6 -> 68.7 us, dQ 87.8 -> 82.0; 512: 140 -> 131,
// 171 -> 154; 1024: 286 -> 281, 321 -> 314; 128: 42 -> 38, 50 -> 46.  (A list-scheduling model of the launch predicted three times
// these gains: workgroups of a partly filled round run faster, the slots are not independent machines.)
// -1 = the plain 2-D grid (B*H not a multiple of 8, a single block, few rows, or the pairs already fill whole rounds).
static int attn_plan_u(int Tn, int BH, int wgs_per_cu) {
    const int nb = cdiv(Tn, 128), pairs = (nb + 1) / 2;
    if (nb < 2 || (BH & 7) || (int64_t)pairs * BH < 512 || wgs_per_cu < 1) return -1;
    const int rows_x = BH / 8, slots = 32 * wgs_per_cu;
#if defined(COMPOSER_EXPERIMENTS)
    if (const char* e = getenv("COMPOSER_ATTN_PLAN_U")) return std::min(atoi(e), rows_x);     // measurement override
#endif
    const int rounds = rows_x * pairs / slots;
    const int paired_rows = rounds > 0 ? rounds * slots / pairs : rows_x / 4;
    const int u = rows_x - paired_rows;
    return u > 0 ? u : -1;
}
static dim3 attn_plan_grid(int Tn, int BH, int u) {
    if (u < 0) return dim3(attn_grid_x(Tn, BH), BH);
    const int nb = cdiv(Tn, 128), pairs = (nb + 1) / 2, rows_x = BH / 8;
    return dim3(8 * ((rows_x - u) * pairs + u * nb), 1);
}
// Small grids (the reference's default configuration at batch 1: 16 (batch, head) rows of 1024 tokens = 128 workgroups of the
// 128-row kind, every one a serial walk over up to 16 key tiles at ~1 us each): the key-split forms of the three kernels (KS).
// COMPOSER_ATTN_KS=0 turns them off, =1 forces them wherever they exist (tests).
static bool attn_key_split(int Tn, int BH, int D, bool bf16) {
    if (!bf16 || D > 32 || Tn < 64) return false;       // (at D = 64 the 256-row staging registers spill: not built)
    static const int mode = [] { const char* e = getenv("COMPOSER_ATTN_KS"); return e ? atoi(e) : -1; }();
    if (mode == 0) return false;
    if (mode == 1) return true;
    // one round of 512 key-split workgroups (two per CU); with twice the rows the split already loses (default model at batch 2:
    // 2.17 vs 2.01 ms/step, tools/ks_threshold_probe.py)
    return (int64_t)cdiv(Tn, 128) * BH <= 128;
}
template <typename K> static int attn_wgs_per_cu(K kernel, size_t smem) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, smem) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
#ifdef COMPOSER_EXPERIMENTS
// COMPOSER_ATTN64=force (read per call) takes the 64-rows-per-wave LDS-DMA forward kernel below (tests/test_gpu_attn64.py runs
// every shape through it).  It is NOT the default: measured on one box at the C2 shape (B*H = 1024, T = 1024), 289 us without
// dropout against 264-300 for the kernel above, and 528 us with dropout (its mask arithmetic pushes the 64-row state past 256
// registers: scratch traffic inside a loop that counts vmcnt by hand) -- DESIGN.md "Attention, round 3".
static int attn64_mode() {
    const char* e = getenv("COMPOSER_ATTN64");
    if (!e) return 0;
    return e[0] == 'o' ? -1 : (e[0] == 'f' ? 1 : (e[0] == 'p' ? 2 : (e[0] == 'd' ? 3 : 0)));
}
#endif
template <typename T, int D>
static int launch_fwd(hipStream_t s, const void* qkv, void* o, float* lse, int B, int Tn, int H, float scale, DropCfg d,
                      const float* amask) {
#ifdef COMPOSER_EXPERIMENTS
    if constexpr (std::is_same<T, bf16_t>::value && D == 64) {
        const int nb = cdiv(Tn, 256), pairs = (nb + 1) / 2;
        const int mode = attn64_mode();
        if (mode == 3 && (int64_t)Tn * 3 * H * 64 * 2 < 0x7FFFFFF0ll) {
            const size_t smem4 = A4_STAGES * A2_STAGE;
            static bool attr4 = false;
            if (!attr4) {
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd32d_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd32d_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4));
                attr4 = true;
            }
            const int nb4 = cdiv(Tn, 128);
            dim3 grid((nb4 + 1) / 2, B * H);
            const double flops = 2.0 * B * H * (double)Tn * Tn * D;
            PROF_START(3, s);
            if (d.thr) attn_fwd32d_kernel<true><<<grid, 256, smem4, s>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, H, scale, d);
            else attn_fwd32d_kernel<false><<<grid, 256, smem4, s>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, H, scale, d);
            PROF_STOP(3, s, flops, (double)B * Tn * H * (4.0 * 64 * 2 + 4.0));
            KERNEL_CHECK();
            return CMP_OK;
        }
        if (mode == 2 && (int64_t)Tn * 3 * H * 64 * 2 < 0x7FFFFFF0ll) {
            const size_t smem3 = A3_STAGES * A2_STAGE;
            static bool attr3 = false;
            if (!attr3) {
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd32p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd32p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3));
                attr3 = true;
            }
            const int nb3 = cdiv(Tn, 128);
            dim3 grid((nb3 + 1) / 2, B * H);
            const double flops = 2.0 * B * H * (double)Tn * Tn * D;
            PROF_START(3, s);
            if (d.thr) attn_fwd32p_kernel<true><<<grid, 256, smem3, s>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, H, scale, d);
            else attn_fwd32p_kernel<false><<<grid, 256, smem3, s>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, H, scale, d);
            PROF_STOP(3, s, flops, (double)B * Tn * H * (4.0 * 64 * 2 + 4.0));
            KERNEL_CHECK();
            return CMP_OK;
        }
        if (mode == 1 && (int64_t)Tn * 3 * H * 64 * 2 < 0x7FFFFFF0ll) {
            const size_t smem2 = A2_STAGES * A2_STAGE;
            static bool attr = false;
            if (!attr) {
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
                HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
                attr = true;
            }
            dim3 grid(pairs, B * H);
            const double flops = 2.0 * B * H * (double)Tn * Tn * D;
            PROF_START(3, s);
            if (d.thr) attn_fwd64_kernel<true><<<grid, 256, smem2, s>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, H, scale, d);
            else attn_fwd64_kernel<false><<<grid, 256, smem2, s>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, H, scale, d);
            PROF_STOP(3, s, flops, (double)B * Tn * H * (4.0 * 64 * 2 + 4.0));
            KERNEL_CHECK();
            return CMP_OK;
        }
    }
#endif
    if constexpr (std::is_same<T, bf16_t>::value && D <= 32) {
        if (!amask && attn_key_split(Tn, B * H, D, true)) {
            const size_t smem_ks = 4 * 256 * Geo<T, D>::S * sizeof(T);
            static const bool attr_ks = [&] {
                bool ok = hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ks) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ks) == hipSuccess;
                return ok;
            }();
            CMP_REQUIRE(attr_ks, "attention: the key-split forward kernel does not get %zu bytes of LDS", smem_ks);
            const dim3 grid(cdiv(Tn, 32), B * H);
            PROF_START(3, s);
            if (d.thr) attn_fwd_kernel<T, D, true, false, true><<<grid, 256, smem_ks, s>>>((const T*)qkv, (T*)o, lse, Tn, H, scale, d, -1, nullptr);
            else attn_fwd_kernel<T, D, false, false, true><<<grid, 256, smem_ks, s>>>((const T*)qkv, (T*)o, lse, Tn, H, scale, d, -1, nullptr);
            PROF_STOP(3, s, 2.0 * B * H * (double)Tn * Tn * D, (double)B * Tn * H * (4.0 * D * sizeof(T) + 4.0));
            KERNEL_CHECK();
            return CMP_OK;
        }
    }
    size_t smem = 4 * 64 * Geo<T, D>::S * sizeof(T);
    if (smem > 65536) {
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    static const int wgs_per_cu = attn_wgs_per_cu(attn_fwd_kernel<T, D, true>, smem);
    const int plan_u = attn_plan_u(Tn, B * H, wgs_per_cu);
    const dim3 grid = attn_plan_grid(Tn, B * H, plan_u);
    const double flops = 2.0 * B * H * (double)Tn * Tn * D;      // QK^T + PV on the unmasked half
    if (amask) {      // Transformer.call(attention_mask=...): the instances that add the per-key mask term
        if (smem > 65536) {
            HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        }
        if (d.thr) attn_fwd_kernel<T, D, true, true><<<grid, 256, smem, s>>>((const T*)qkv, (T*)o, lse, Tn, H, scale, d, plan_u, amask);
        else attn_fwd_kernel<T, D, false, true><<<grid, 256, smem, s>>>((const T*)qkv, (T*)o, lse, Tn, H, scale, d, plan_u, amask);
        KERNEL_CHECK();
        return CMP_OK;
    }
    PROF_START(3, s);
    if (d.thr) attn_fwd_kernel<T, D, true><<<grid, 256, smem, s>>>((const T*)qkv, (T*)o, lse, Tn, H, scale, d, plan_u, nullptr);
    else attn_fwd_kernel<T, D, false><<<grid, 256, smem, s>>>((const T*)qkv, (T*)o, lse, Tn, H, scale, d, plan_u, nullptr);
    PROF_STOP(3, s, flops, (double)B * Tn * H * (4.0 * D * sizeof(T) + 4.0));      // q, k, v in; o, lse out
    KERNEL_CHECK();
    return CMP_OK;
}
template <typename T, int D>
static int launch_bwd(hipStream_t s, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                      void* dqkv, int B, int Tn, int H, float scale, DropCfg d, float* bias_grad) {
    if constexpr (std::is_same<T, bf16_t>::value && D <= 32) {
        if (attn_key_split(Tn, B * H, D, true)) {
            // one launch, both roles: LDS = the larger of the two (dQ double-buffers its 256-key tiles)
            const size_t smem_ks = 4 * 256 * Geo<T, D>::S * sizeof(T);
            const size_t smem_kv = smem_ks / 2 + 3 * 256 * sizeof(float);
            static const bool attr_ks = [&] {
                bool ok = hipFuncSetAttribute((const void*)attn_bwd_ks_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ks) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)attn_bwd_ks_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ks) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)attn_dq_ks_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ks) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)attn_dq_ks_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ks) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)attn_dkv_ks_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv) == hipSuccess;
                ok = ok && hipFuncSetAttribute((const void*)attn_dkv_ks_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_kv) == hipSuccess;
                return ok;
            }();
            CMP_REQUIRE(attr_ks, "attention: the key-split backward kernels do not get %zu bytes of LDS", smem_ks);
            const double flk = (double)B * H * (double)Tn * Tn * D;
            // The c_attn bias gradient (column sums of [dQ | dK | dV]) is NOT fused here: a fused sum ends a workgroup on a transpose-reduce
            // and a device-scope float atomic whose round trip the launch has to wait out -- 16.5 of the 29.4 us of the dK/dV launch (two
            // sums) and ~8 of dQ's 19 at the default configuration (measurement builds, profiles/r5_06); only a launch of several rounds
            // hides that tail behind other workgroups.  A column-sum pass over the [tokens, 3E] gradient costs ~5 us at these sizes (63 us
            // at the benchmark's, where the fused sums stay).
            float* const bias_out = bias_grad;
            bias_grad = nullptr;
            auto bias_pass = [&]() -> int {
                return bias_out ? colsum_run(s, dqkv, 3 * H * D, bias_out, B * Tn, 3 * H * D, CMP_BF16, nullptr, 0) : CMP_OK;
            };
            if ((int64_t)2 * cdiv(Tn, 32) * B * H > 512) {         // not resident at once: two launches
                const dim3 g1(cdiv(Tn, 32), B * H);
                PROF_START(4, s);
                if (d.thr) attn_dq_ks_kernel<T, D, true><<<g1, 256, smem_ks, s>>>((const T*)qkv, (const T*)o, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
                else attn_dq_ks_kernel<T, D, false><<<g1, 256, smem_ks, s>>>((const T*)qkv, (const T*)o, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
                PROF_STOP(4, s, 3.0 * flk, (double)B * Tn * H * (6.0 * D * sizeof(T) + 8.0));
                KERNEL_CHECK();
                PROF_START(5, s);
                if (d.thr) attn_dkv_ks_kernel<T, D, true><<<g1, 256, smem_kv, s>>>((const T*)qkv, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
                else attn_dkv_ks_kernel<T, D, false><<<g1, 256, smem_kv, s>>>((const T*)qkv, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
                PROF_STOP(5, s, 4.0 * flk, (double)B * Tn * H * (6.0 * D * sizeof(T) + 8.0));
                KERNEL_CHECK();
                return bias_pass();
            }
            const dim3 kgrid(2 * cdiv(Tn, 32), B * H);
            PROF_START(5, s);                  // (counted with the dK/dV class: one launch carries the seven products)
            if (d.thr) attn_bwd_ks_kernel<T, D, true><<<kgrid, 256, smem_ks, s>>>((const T*)qkv, (const T*)o, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
            else attn_bwd_ks_kernel<T, D, false><<<kgrid, 256, smem_ks, s>>>((const T*)qkv, (const T*)o, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
            PROF_STOP(5, s, 7.0 * flk, (double)B * Tn * H * (9.0 * D * sizeof(T) + 8.0));
            KERNEL_CHECK();
            return bias_pass();
        }
    }
    dim3 grid(attn_grid_x(Tn, B * H), B * H);
    // Short launches take the c_attn bias sums out of the kernels too (see the key-split branch): the fused float atomics leave a ~16 us
    // tail that only a launch of several rounds hides.  Same box, [tokens, 3E] gradient of 1.6 ... 12.6 MB (the default model at batch
    // 1 ... 8): 2.99 -> 2.67 ms/step at batch 8, 1.95 -> 1.76 at batch 2; 25 MB (6L/8H/d512 at batch 8): 3.03 vs 3.04; 100 MB (batch 32):
    // 7.39 vs 7.43 -- the pass re-reads the gradient, so it stops at 16 MiB (COMPOSER_ATTN_BIAS_PASS=<MiB> overrides, 0 = never).
    static const int bias_pass_mib = [] { const char* e = getenv("COMPOSER_ATTN_BIAS_PASS"); return e ? atoi(e) : 16; }();
    float* bias_out = nullptr;
    if (bias_grad && std::is_same<T, bf16_t>::value && (int64_t)B * Tn * 3 * H * D * 2 <= (int64_t)bias_pass_mib << 20) { bias_out = bias_grad; bias_grad = nullptr; }
    size_t smem = 4 * 64 * Geo<T, D>::S * sizeof(T);
    if (smem + 1536 > 65536) {
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_dq_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_dq_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_dkv_kernel<T, D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem + 1536));
        HIP_CHECK(hipFuncSetAttribute((const void*)attn_dkv_kernel<T, D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem + 1536));
    }
    const double fl = (double)B * H * (double)Tn * Tn * D;        // one product over the unmasked half
    PROF_START(4, s);
    static const int wgs_per_cu = attn_wgs_per_cu(attn_dq_kernel<T, D, true>, smem);
    const int plan_u = attn_plan_u(Tn, B * H, wgs_per_cu);
    const dim3 qgrid = attn_plan_grid(Tn, B * H, plan_u);
    if (d.thr) attn_dq_kernel<T, D, true><<<qgrid, 256, smem, s>>>((const T*)qkv, (const T*)o, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d, plan_u);
    else attn_dq_kernel<T, D, false><<<qgrid, 256, smem, s>>>((const T*)qkv, (const T*)o, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d, plan_u);
    PROF_STOP(4, s, 3.0 * fl, (double)B * Tn * H * (6.0 * D * sizeof(T) + 8.0));       // q, k, v, o, dO, lse in; dQ, delta out
    KERNEL_CHECK();
    PROF_START(5, s);
    if (d.thr) attn_dkv_kernel<T, D, true><<<grid, 256, smem + 384 * sizeof(float), s>>>((const T*)qkv, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
    else attn_dkv_kernel<T, D, false><<<grid, 256, smem + 384 * sizeof(float), s>>>((const T*)qkv, (const T*)d_o, lse, delta, (T*)dqkv, bias_grad, Tn, H, scale, d);
    PROF_STOP(5, s, 4.0 * fl, (double)B * Tn * H * (6.0 * D * sizeof(T) + 8.0));       // q, k, v, dO, lse, delta in; dK, dV out
    KERNEL_CHECK();
    if (bias_out) return colsum_run(s, dqkv, 3 * H * D, bias_out, B * Tn, 3 * H * D, CMP_BF16, nullptr, 0);
    return CMP_OK;
}

#define DISPATCH_D(T, fn, ...)                                                         \
    switch (D) {                                                                       \
        case 16: return fn<T, 16>(__VA_ARGS__);                                        \
        case 32: return fn<T, 32>(__VA_ARGS__);                                        \
        case 64: return fn<T, 64>(__VA_ARGS__);                                        \
        case 128: return fn<T, 128>(__VA_ARGS__);                                      \
        default: cmp_set_error("attention: head size %d unsupported (16/32/64/128)", D); \
                 return CMP_ERR_INVALID;                                               \
    }

extern "C" int cmp_k_attn_fwd(void* stream, const void* qkv, void* o, float* lse, int B, int T, int H, int D, int scale,
                              int dtype, float p_drop, uint64_t seed, uint32_t rng_stream) {
    return attn_fwd_run(stream, qkv, o, lse, B, T, H, D, scale ? 1.0f / sqrtf((float)D) : 1.0f, dtype, p_drop, seed, rng_stream, nullptr);
}

// sc: the factor on q.k (the model driver passes 1/sqrt(E/H) of the reference's head size, which may be smaller than D: zero-padded heads)
// amask: device float [B, T] added to the scaled scores of every query of a batch row (null: none)
int attn_fwd_run(void* stream, const void* qkv, void* o, float* lse, int B, int T, int H, int D, float sc, int dtype, float p_drop,
                 uint64_t seed, uint32_t rng_stream, const float* amask) {
    if (B * T == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    DropCfg d = make_drop(p_drop, seed, rng_stream);
    if (dtype == CMP_BF16) { DISPATCH_D(bf16_t, launch_fwd, s, qkv, o, lse, B, T, H, sc, d, amask) }
    else { DISPATCH_D(float, launch_fwd, s, qkv, o, lse, B, T, H, sc, d, amask) }
}

// The attention probabilities themselves (Transformer(..., output_attention_weights=True), transformer.py:360-369, 808-809):
// out[b][h][i - q0][j] = dropout(softmax row i)[j] for the queries i in [q0, Tn), recomputed from q, k and the forward
// kernel's row log-sum-exp.  An inspection path (the [B,H,T,T] tensor the hot kernels never write): one workgroup per
// (batch, head, query), a thread per key, plain dot products.
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(const T* __restrict__ qkv, const float* __restrict__ lse,
                                                         const float* __restrict__ amask, float* __restrict__ out, int q0, int Tn,
                                                         int H, int D, float scale, DropCfg drop) {
    const int bh = blockIdx.y, i = q0 + blockIdx.x, b = bh / H, hd = bh % H;
    const int E = H * D;
    const int64_t rs = 3 * E;
    const T* qrow = qkv + ((int64_t)b * Tn + i) * rs + hd * D;
    const float l = lse[(int64_t)bh * Tn + i];
    const uint32_t rowh = attn_row_hash(drop.seed, drop.stream, (uint32_t)(bh * Tn + i));
    float* orow = out + ((int64_t)bh * (Tn - q0) + (i - q0)) * Tn;
    for (int j = threadIdx.x; j < Tn; j += blockDim.x) {
        const T* krow = qkv + ((int64_t)b * Tn + j) * rs + E + hd * D;
        float sdot = 0.f;
        for (int d = 0; d < D; d++) sdot = fmaf(to_f32<T>(qrow[d]), to_f32<T>(krow[d]), sdot);
        float v = sdot * scale;
        if (j > i) v = -1e4f;                                          // :351-354
        if (amask) v += amask[(int64_t)b * Tn + j];                    // :356-358
        float pr = expf(v - l);                                        // :360
        if (drop.thr) pr = attn_elem_hash(rowh, (uint32_t)j) >= drop.thr ? pr * drop.scale : 0.f;     // :361
        orow[j] = pr;
    }
}

int attn_probs_run(void* stream, const void* qkv, const float* lse, const float* amask, float* out, int B, int q0, int T, int H, int D,
                   float sc, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream) {
    if (B * (T - q0) <= 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, rng_stream);
    const dim3 grid(T - q0, B * H);
    if (dtype == CMP_BF16) attn_probs_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)qkv, lse, amask, out, q0, T, H, D, sc, d);
    else attn_probs_kernel<float><<<grid, 256, 0, s>>>((const float*)qkv, lse, amask, out, q0, T, H, D, sc, d);
    KERNEL_CHECK();
    return CMP_OK;
}

// One-shot for the kernel-level tests that call cmp_k_attn_bwd directly (per calling thread): the next cmp_k_attn_bwd also
// adds the column sums of [dQ | dK | dV] to out[0..3E) (the c_attn bias gradient).  The model driver passes the pointer to
// attn_bwd_run itself.
static thread_local float* t_attn_bias_next = nullptr;
extern "C" int cmp_attn_bwd_bias_next(float* out) {
    t_attn_bias_next = out;
    return CMP_OK;
}

extern "C" int cmp_k_attn_bwd(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse,
                              float* delta_ws, void* dqkv, int B, int T, int H, int D, int scale, int dtype,
                              float p_drop, uint64_t seed, uint32_t rng_stream) {
    float* bias_grad = t_attn_bias_next;
    t_attn_bias_next = nullptr;
    return attn_bwd_run(stream, qkv, o, d_o, lse, delta_ws, dqkv, B, T, H, D, scale ? 1.0f / sqrtf((float)D) : 1.0f, dtype, p_drop, seed,
                        rng_stream, bias_grad);
}

int attn_bwd_run(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dqkv,
                 int B, int T, int H, int D, float sc, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream,
                 float* bias_grad, AttnLnRows lr) {
    // (round-6 signature; the lab copy has no rstd-scaled stores: COMPOSER_LN_FUSED=3 needs the product library)
    CMP_REQUIRE(!lr.rstd, "attention backward (experiments build): LayerNorm row scaling is not in the lab copy");
    if (B * T == 0) return CMP_OK;
    hipStream_t s = (hipStream_t)stream;
    DropCfg d = make_drop(p_drop, seed, rng_stream);
    if (dtype == CMP_BF16) { DISPATCH_D(bf16_t, launch_bwd, s, qkv, o, d_o, lse, delta_ws, dqkv, B, T, H, sc, d, bias_grad) }
    else { DISPATCH_D(float, launch_bwd, s, qkv, o, d_o, lse, delta_ws, dqkv, B, T, H, sc, d, bias_grad) }
}
