// gemm.hip -- dense contractions of the Transformer hot path on the gfx950 matrix cores.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(k,n) )
//
// Replaces Conv1D.call (transformer.py:205-209), the tied-logits matmul (:139-144) and their
// tape gradients (:920).  Operand storage (no data is ever transposed in HBM):
//   ta=0: A stored [M,K] (K contiguous: forward X, dgrad dY)        ta=1: A stored [K,M] (wgrad X^T)
//   tb=0: B stored [K,N] (N contiguous: Conv1D weight, wgrad dY)    tb=1: B stored [N,K] (dgrad W, logits wte)
//
// bf16 path: v_mfma_f32_16x16x32_bf16, 128x128x64 workgroup tile, 4 waves (2x2) of 64x64, LDS double buffer.
//   K-contiguous operands : LDS image [row][64 k] (128-B rows, 16-B chunk XOR-swizzled by (row>>1)&7),
//                           fragments by ds_read_b128 (conflict-free for the b128 lane groups).
//   contraction-slow ones : LDS image [64 k][128 cols] (256-B rows, chunk XOR 2*((k&3)|((k>>3)&1)<<2)),
//                           fragments by two ds_read_b64_tr_b16 (hardware transpose, conflict-free).
// fp32 path (parity mode): v_mfma_f32_16x16x4_f32 = exact fp32 fma chains, 64x64x16 tile.
#include "../common.h"
#include <vector>
#include <mutex>

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct Epilogue {
    const float* bias;     // [N] or null
    int act;               // 0 none, 1 gelu (pre-activation -> aux), 2 multiply by gelu'(aux)
    void* aux;
    int ldaux;
    const void* resid;
    int ldr;
    int out_fp32;
    int atomic;            // split-K: atomicAdd into fp32 C
    int dbg_nostore;       // timing experiment only: run the whole epilogue but skip the global stores
    float* colsum;         // compile-time kinds only: out[col] += sum over rows of the STORED (rounded) C (f32 atomics)
    DropCfg drop;
    uint32_t* sched;       // persistent kernels: item counters of this launch (ItemPuller below); null = static striding
    uint32_t* sched_clear; // ... and the counter set the NEXT launch on this stream will draw from (zeroed by this one)
    LnEpi ln;              // LayerNorm fused into the epilogue (compile-time kinds of the forward layout only, see common.h)
    int rev;               // persistent 256x256 kernel: every XCD group walks its run of tiles from the END (see item_coords)
    int n_cols;            // EPI_PLAIN32: live output columns (the last tile column may be ragged)
};

template <typename T, bool EXACT>
__device__ __forceinline__ void epilogue_store(const Epilogue& ep, void* C, int ldc, int row, int col, float v) {
    if (ep.atomic) {
        atomicAdd((float*)C + (int64_t)row * ldc + col, v);
        return;
    }
    if (ep.bias) v += ep.bias[col];
    if (ep.act == 1) {
        if (ep.aux) ((T*)ep.aux)[(int64_t)row * ep.ldaux + col] = from_f32<T>(v);
        v = gelu_f<EXACT>(v);
    } else if (ep.act == 2) {
        v *= gelu_grad_f<EXACT>(to_f32<T>(((const T*)ep.aux)[(int64_t)row * ep.ldaux + col]));
    }
    v = apply_drop(ep.drop, (uint32_t)row, (uint32_t)col, v);
    if (ep.resid) v += to_f32<T>(((const T*)ep.resid)[(int64_t)row * ep.ldr + col]);
    if (ep.out_fp32) ((float*)C)[(int64_t)row * ldc + col] = v;
    else ((T*)C)[(int64_t)row * ldc + col] = from_f32<T>(v);
}

// =================================================================================================
// fp32 (parity mode)
// =================================================================================================
#define F_BM 64
#define F_BN 64
#define F_BK 16
#define F_LD 80   // LDS row stride in floats: 80 mod 32 = 16 -> the two k-rows of a 32-lane read group hit disjoint banks

__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ A, int64_t sam,
                                                       int64_t sak, const float* __restrict__ B, int64_t sbk,
                                                       int64_t sbn, void* C, int ldc, Epilogue ep, int ktiles_per_split) {
    __shared__ float As[F_BK][F_LD];
    __shared__ float Bs[F_BK][F_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * F_BM, n0 = blockIdx.x * F_BN;
    const int nk = cdiv(K, F_BK);
    const int kt0 = blockIdx.z * ktiles_per_split;
    const int kt1 = min(nk, kt0 + ktiles_per_split);
    const bool a_kc = (sak == 1), b_kc = (sbk == 1);

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float ra[4], rb[4];
    auto gload = [&](int kt) {
        const int k0 = kt * F_BK;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int idx = tid + 256 * i;
            int m, k;
            if (a_kc) { m = idx >> 4; k = idx & 15; } else { k = idx >> 6; m = idx & 63; }
            ra[i] = (m0 + m < M && k0 + k < K) ? A[(int64_t)(m0 + m) * sam + (int64_t)(k0 + k) * sak] : 0.f;
            int n;
            if (b_kc) { n = idx >> 4; k = idx & 15; } else { k = idx >> 6; n = idx & 63; }
            rb[i] = (n0 + n < N && k0 + k < K) ? B[(int64_t)(k0 + k) * sbk + (int64_t)(n0 + n) * sbn] : 0.f;
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int idx = tid + 256 * i;
            int m, k;
            if (a_kc) { m = idx >> 4; k = idx & 15; } else { k = idx >> 6; m = idx & 63; }
            As[k][m] = ra[i];
            int n;
            if (b_kc) { n = idx >> 4; k = idx & 15; } else { k = idx >> 6; n = idx & 63; }
            Bs[k][n] = rb[i];
        }
    };

    if (kt0 < kt1) gload(kt0);
    for (int kt = kt0; kt < kt1; kt++) {
        __syncthreads();
        swrite();
        __syncthreads();
        if (kt + 1 < kt1) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < F_BK / 4; ks++) {
            const int kk = ks * 4 + (lane >> 4);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                a[i] = As[kk][wm * 32 + i * 16 + (lane & 15)];
                b[i] = Bs[kk][wn * 32 + i * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                int col = n0 + wn * 32 + j * 16 + (lane & 15);
                if (row < M && col < N) epilogue_store<float, true>(ep, C, ldc, row, col, acc[i][j][r]);
            }
}

// =================================================================================================
// bf16
// =================================================================================================
#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_IMG (128 * 64 * 2)   // bytes per operand image

// K-major image [128 rows][64 k]: byte offset of 16-B chunk c (0..7) of row r
__device__ __forceinline__ int kmaj_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
// contraction-slow image [64 k][128 cols]: byte offset of 16-B chunk c (0..15) of k-row k
__device__ __forceinline__ int cslow_off(int k, int c) { return k * 256 + ((c ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1)) << 4); }

template <bool KMAJOR>
__device__ __forceinline__ void g_load_tile(bf16x8 (&r)[4], const bf16_t* __restrict__ P, int64_t ld, int row0, int nrows,
                                            int k0, int K, int tid) {
    // KMAJOR: P[(row0+row)*ld + k]; rows = tile's M/N index.   else: P[(k0+k)*ld + row0 + col]
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int q = tid + 256 * i;
        bf16x8 z;
#pragma unroll
        for (int j = 0; j < 8; j++) z[j] = (bf16_t)0.f;
        if (KMAJOR) {
            int row = q >> 3, c = q & 7;
            bool ok = (row0 + row < nrows) && (k0 + c * 8 < K);
            r[i] = ok ? *reinterpret_cast<const bf16x8*>(P + (int64_t)(row0 + row) * ld + k0 + c * 8) : z;
        } else {
            int k = q >> 4, c = q & 15;
            bool ok = (k0 + k < K) && (row0 + c * 8 < nrows);
            r[i] = ok ? *reinterpret_cast<const bf16x8*>(P + (int64_t)(k0 + k) * ld + row0 + c * 8) : z;
        }
    }
}
template <bool KMAJOR>
__device__ __forceinline__ void g_store_tile(const bf16x8 (&r)[4], char* img, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int q = tid + 256 * i;
        int off = KMAJOR ? kmaj_off(q >> 3, q & 7) : cslow_off(q >> 4, q & 15);
        *reinterpret_cast<bf16x8*>(img + off) = r[i];
    }
}
// fragment for one 16-row (or 16-col) tile `t16` (index within the 128-wide image) and 32-deep k-step ks
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 g_frag(const char* img, int t16, int ks, int lane) {
    if (KMAJOR) {
        int row = t16 * 16 + (lane & 15);
        int c = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(img + kmaj_off(row, c));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int k = ks * 32 + 8 * g + q;
        const int c = t16 * 2 + (p >> 1);
        const int sub = (p & 1) * 8;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + cslow_off(k, c) + sub));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + cslow_off(k + 4, c) + sub));
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[j] = lo[j];
            f[4 + j] = hi[j];
        }
        return f;
    }
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(int M, int N, int K, const bf16_t* __restrict__ A, int64_t lda,
                                                        const bf16_t* __restrict__ B, int64_t ldb, void* C, int ldc,
                                                        Epilogue ep, int ktiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A img | B img]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;
    const int nk = cdiv(K, G_BK);
    const int kt0 = blockIdx.z * ktiles_per_split;
    const int kt1 = min(nk, kt0 + ktiles_per_split);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[4], rb[4];
    if (kt0 < kt1) {
        g_load_tile<A_KM>(ra, A, lda, m0, M, kt0 * G_BK, K, tid);
        g_load_tile<B_KM>(rb, B, ldb, n0, N, kt0 * G_BK, K, tid);
        g_store_tile<A_KM>(ra, smem, tid);
        g_store_tile<B_KM>(rb, smem + G_IMG, tid);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt++) {
        const int st = (kt - kt0) & 1;
        const char* ia = smem + st * 2 * G_IMG;
        const char* ib = ia + G_IMG;
        const bool more = kt + 1 < kt1;
        if (more) {
            g_load_tile<A_KM>(ra, A, lda, m0, M, (kt + 1) * G_BK, K, tid);
            g_load_tile<B_KM>(rb, B, ldb, n0, N, (kt + 1) * G_BK, K, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                fa[i] = g_frag<A_KM>(ia, wm * 4 + i, ks, lane);
                fb[i] = g_frag<B_KM>(ib, wn * 4 + i, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            char* na = smem + (st ^ 1) * 2 * G_IMG;
            g_store_tile<A_KM>(ra, na, tid);
            g_store_tile<B_KM>(rb, na + G_IMG, tid);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                int col = n0 + wn * 64 + j * 16 + (lane & 15);
                if (row < M && col < N) epilogue_store<bf16_t, false>(ep, C, ldc, row, col, acc[i][j][r]);
            }
}

// =================================================================================================
// bf16 fast path: same tile and LDS images, but
//   * operands go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction); the LDS
//     destination is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address and the
//     same XOR on the fragment read (rule "both sides or neither");
//   * out-of-range rows come back as zeros from the buffer descriptor's range check (exact in the slow
//     dimension of either layout);
//   * SWAP: mfma(B_frag, A_frag) puts the C ROW on the lane and 4 consecutive C COLUMNS in the 4 accumulator
//     registers -> 8-byte (bf16) / 16-byte (fp32) epilogue stores and vector bias/aux/residual loads;
//     split-K launches keep the un-swapped map (16 consecutive floats per row per atomic instruction);
//   * 1-D grid with a bijective XCD-aware remap: workgroups that share an A row-panel run on one XCD's L2.
// =================================================================================================
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

template <bool KM>
__device__ __forceinline__ void glds_tile(__amdgpu_buffer_rsrc_t rs, char* img, int ld_bytes, int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = wave * 4 + i;
        int voff;
        if (KM) {
            const int r = 8 * p + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voff = r * ld_bytes + (k0 + c * 8) * 2;
        } else {
            const int k = 4 * p + (lane >> 4);
            const int c = (lane & 15) ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1);
            voff = (k0 + k) * ld_bytes + c * 16;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + p * 1024), 16, voff, 0, 0, 0);
    }
}

template <typename T>
__device__ __forceinline__ void epilogue_store4(const Epilogue& ep, void* C, int ldc, int row, int col, f32x4 v) {
    if (ep.bias) {
        f32x4 b = *reinterpret_cast<const f32x4*>(ep.bias + col);
        v += b;
    }
    if (ep.act == 1) {
        if (ep.aux) {
            bf16x4 a;
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x4*>((bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col) = a;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = gelu_f<false>(v[j]);
    } else if (ep.act == 2) {
        bf16x4 a = *reinterpret_cast<const bf16x4*>((const bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] *= gelu_grad_f<false>((float)a[j]);
    }
    if (ep.drop.thr) {
        const uint32_t rh = drop_row_hash(ep.drop, (uint32_t)row);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = apply_drop_rc(ep.drop, rh, (uint32_t)(col + j), v[j]);
    }
    if (ep.resid) {
        bf16x4 r = *reinterpret_cast<const bf16x4*>((const bf16_t*)ep.resid + (int64_t)row * ep.ldr + col);
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] += (float)r[j];
    }
    if (ep.out_fp32) {
        *reinterpret_cast<f32x4*>((float*)C + (int64_t)row * ldc + col) = v;
    } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x4*>((bf16_t*)C + (int64_t)row * ldc + col) = o;
    }
}

// 8 consecutive columns of one row (col % 8 == 0); columns >= N of a ragged last chunk are not stored
__device__ __forceinline__ void epilogue_store8(const Epilogue& ep, void* C, int ldc, int row, int col, int N, f32x4 v0, f32x4 v1) {
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const bool full = col + 8 <= N;
    if (ep.bias) {
        if (full) {
            f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + col), b1 = *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) { v[j] += b0[j]; v[4 + j] += b1[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) if (col + j < N) v[j] += ep.bias[col + j];
        }
    }
    if (ep.act == 1) {
        if (ep.aux) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x8*>((bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col) = a;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = gelu_f<false>(v[j]);
    } else if (ep.act == 2) {
        bf16x8 a = *reinterpret_cast<const bf16x8*>((const bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] *= gelu_grad_f<false>((float)a[j]);
    }
    if (ep.drop.thr) {
        const uint32_t rh = drop_row_hash(ep.drop, (uint32_t)row);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = apply_drop_rc(ep.drop, rh, (uint32_t)(col + j), v[j]);
    }
    if (ep.resid) {
        bf16x8 r = *reinterpret_cast<const bf16x8*>((const bf16_t*)ep.resid + (int64_t)row * ep.ldr + col);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] += (float)r[j];
    }
    if (ep.dbg_nostore && v[0] != 12345.678f) return;
    if (ep.out_fp32) {
        float* o = (float*)C + (int64_t)row * ldc + col;
        if (full) {
            *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) if (col + j < N) o[j] = v[j];
        }
    } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8*>((bf16_t*)C + (int64_t)row * ldc + col) = o;     // ldc is padded to a multiple of 8
    }
}


// ---- split epilogue for the persistent kernels: the memory operands of a chunk (gelu' input, residual) are fetched
// one accumulator row-group AHEAD of their use and the bias once per item -- C may alias them as far as the compiler
// knows, so inside epilogue_store8 every chunk re-loads them after the previous chunk's store (a serial L2 round trip
// per chunk: 16 per wave and item, measured ~25k cycles of a 74k-cycle c_attn item).
struct EpiIn {
    bf16x8 v;          // gelu' input (act == 2) or the residual; no epilogue in the model uses both
};
__device__ __forceinline__ void epi_bias8(const Epilogue& ep, int col, int N, float (&b)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) b[j] = 0.f;
    if (!ep.bias || col >= N) return;
    if (col + 8 <= N) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + col), b1 = *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { b[j] = b0[j]; b[4 + j] = b1[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) if (col + j < N) b[j] = ep.bias[col + j];
    }
}
__device__ __forceinline__ EpiIn epi_fetch8(const Epilogue& ep, int row, int col, bool ok) {
    EpiIn in;
#pragma unroll
    for (int j = 0; j < 8; j++) in.v[j] = (bf16_t)0.f;
    if (ok) {
        if (ep.act == 2) in.v = *reinterpret_cast<const bf16x8*>((const bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col);
        else if (ep.resid) in.v = *reinterpret_cast<const bf16x8*>((const bf16_t*)ep.resid + (int64_t)row * ep.ldr + col);
    }
    return in;
}
__device__ __forceinline__ void epi_finish8(const Epilogue& ep, void* C, int ldc, int row, int col, int N, f32x4 v0, f32x4 v1,
                                            const float (&b)[8], const EpiIn& in) {
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] += b[j];
    if (ep.act == 1) {
        if (ep.aux) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = (bf16_t)v[j];
            *reinterpret_cast<bf16x8*>((bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col) = a;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = gelu_f<false>(v[j]);
    } else if (ep.act == 2) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] *= gelu_grad_f<false>((float)in.v[j]);
    }
    if (ep.drop.thr) {
        const uint32_t rh = drop_row_hash(ep.drop, (uint32_t)row);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = apply_drop_rc(ep.drop, rh, (uint32_t)(col + j), v[j]);
    }
    if (ep.resid) {
        if (ep.act == 2) {      // not used by the model: fall back to an in-place load
            const bf16x8 r = *reinterpret_cast<const bf16x8*>((const bf16_t*)ep.resid + (int64_t)row * ep.ldr + col);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += (float)r[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += (float)in.v[j];
        }
    }
    if (ep.dbg_nostore && v[0] != 12345.678f) return;
    if (ep.out_fp32) {
        float* o = (float*)C + (int64_t)row * ldc + col;
        if (col + 8 <= N) {
            *reinterpret_cast<f32x4*>(o) = (f32x4){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) if (col + j < N) o[j] = v[j];
        }
    } else {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8*>((bf16_t*)C + (int64_t)row * ldc + col) = o;     // ldc is padded to a multiple of 8
    }
}


// ---- compile-time epilogue kinds for the persistent kernels (full 256x256 tiles, bf16 output) -------------------------
// The generic epilogue branches at run time on the Epilogue fields; with global loads under those branches the
// compiler's waitcnt insertion falls back to `s_waitcnt vmcnt(0)` in every chunk, i.e. every chunk waits for the
// previous chunk's STORES to be acknowledged (measured with in-kernel stamps: 18-50k cycles for the 16 chunks of a
// tile, the store round trip serialised 16 times).  A kind fixes what touches memory inside the loop, the loop is
// straight-line code, the operand loads run two chunks ahead of their use and the waits are counted: stores are
// fire-and-forget.
enum { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_GELU_AUX = 2, EPI_RESID = 3, EPI_GELUGRAD = 4, EPI_PLAIN32 = 5 };
//   EPI_PLAIN32  : C (fp32, ragged last tile column) = acc + bias, with the LayerNorm fold only: ln_f into the tied-logits GEMM of
//                  the fused block path (transformer.py:811, 818)
//   EPI_PLAIN    : C = acc (+ bias)                                  c_attn forward; dgrad without epilogue operands
//   EPI_GELU_AUX : aux = acc + bias ; C = gelu(aux)                  c_fc forward
//   EPI_RESID    : C = drop(acc (+ bias)) + resid                    both c_proj forward; dgrad c_attn (+ residual grad)
//   EPI_GELUGRAD : C = acc * gelu'(aux)                              dgrad through the MLP activation
typedef int v4i32 __attribute__((ext_vector_type(4)));
// raw buffer descriptor words (stride 0, range-checked): {base[31:0], base[47:32], num_records, flags}
__device__ __forceinline__ v4i32 make_srd(const void* base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    v4i32 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    d[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xFFFFu));
    d[2] = __builtin_amdgcn_readfirstlane((int)min(bytes, (int64_t)0x7FFFFFF0));
    d[3] = 0x00020000;
    return d;
}
// One 1-KiB LDS-DMA piece issued from inline asm, so hipcc's wait-count pass does not see a pending LDS write (it
// would put s_waitcnt vmcnt(0) in front of the next ds_read and drain the pipeline).  M0 (LDS base) is written in
// the same statement that reads it; completion is tracked by hand with counted s_waitcnt vmcnt(N).
__device__ __forceinline__ void dma16(v4i32 srd, uint32_t lds_addr, int voff) {
    // the "s" operands must BE in SGPRs: after control-flow merges the compiler may carry wave-uniform values in VGPRs
    // and does not legalise inline-asm operands (readfirstlane folds away when the value already lives in an SGPR)
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
#pragma unroll
    for (int i = 0; i < 4; i++) srd[i] = __builtin_amdgcn_readfirstlane(srd[i]);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(lds_addr), "v"(voff), "s"(srd)
                 : "memory", "m0");
}

// LNM (forward layout only): bit 0 = LayerNorm on the way IN (EPI_PLAIN / EPI_GELU_AUX: the fold of common.h's LnEpi -- the GEMM ran
// on the raw rows and the gamma-scaled weight; EPI_RESID: the residual operand is LN(resid), rebuilt from the raw rows), NP =
// 256-column segments of the LayerNorm input's rows; bit 1 = this tile's partial statistics of the OUTPUT rows go out (EPI_RESID):
// per chunk the eight lanes that share a row merge their eight-column partials (three DPP steps), lane 0 of the eight parks the
// wave's 64-column partial in `lstat` (LDS, [4 wave columns][256 tile rows]); the caller folds the four behind a barrier.
//
// What an epilogue READS from memory is requested by epi_prefetch(), which the 256x256 kernel calls BEFORE it issues the next
// item's first k-slab: memory reads return in issue order, so a load issued behind the eight LDS-DMA pieces of a wave is only
// delivered once they have landed (~2 us with the whole chip loading) -- the epilogue then starts after the slab it was meant
// to overlap.  NPRE = operand chunks requested up front (16: all of them, the rest of the epilogue issues no load; 2: the rolling
// form of the deep-pipeline kernels, which keep their stages in flight across the epilogue anyway).
// operand chunks of the 256x256 kernel's epilogues requested ahead of their use (rolling)
#ifndef EPI_AHEAD
#define EPI_AHEAD 2
#endif
// experiment switches (tools/ab_build.py): default-policy stores for more kinds
#ifndef XC_PLAIN_RESID
#define XC_PLAIN_RESID 0
#endif
#ifndef XC_PLAIN_PLAINKIND
#define XC_PLAIN_PLAINKIND 0
#endif
template <int KIND, int LNM, int NP>
struct EpiPre {
    float b[8];                 // bias
    float lnA[8], lnB[8];       // fold: column sums of the gamma-scaled weight; residual rebuild: gamma, beta
    bf16x8 opnd[16];            // residual / gelu' operand of every chunk
    f32x2 pt[2][NP];            // LayerNorm on the way in: partial statistics of rows lane and lane + 64 of the wave's 128
};
template <int KIND, int LNM = 0>
__device__ __forceinline__ bf16x8 epi_opnd(const Epilogue& ep, int row0, int col, int c, int rl) {
    const bf16_t* __restrict__ src = KIND == EPI_RESID ? (const bf16_t*)ep.resid : (const bf16_t*)ep.aux;
    const int lds = KIND == EPI_RESID ? ep.ldr : ep.ldaux;
    const int row = row0 + c * 8 + rl;
    // the 4E-wide pre-activation stream (0.5 GB per launch at B=128) is read non-temporally: 408 -> 380 us same-box;
    // for the E-wide residual the hint measured neutral
    // fused block path: the residual operand of a c_proj GEMM is read for the last time -- non-temporal, so that it does not
    // take Infinity Cache room from the rows this GEMM writes for the next one (profiles/r5_01_ln_fused.txt)
    if (LNM != 0) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src + (int64_t)row * lds + col));
    if (KIND == EPI_GELUGRAD) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src + (int64_t)row * lds + col));
    return *reinterpret_cast<const bf16x8*>(src + (int64_t)row * lds + col);
}
// Fold kinds (LayerNorm into c_attn / c_fc): what the epilogue reads -- the partial statistics of the wave's 128 rows (NP KiB,
// contiguous), 64 floats of bias' and 64 of the column sums -- goes HBM -> LDS by DMA (no registers), issued by fold_dma() BEFORE
// the next item's first k-slab: memory reads are delivered in issue order, so a load issued behind the eight slab pieces of a wave
// only arrives once they have landed (~2 us with the whole chip loading).  (Measured: the same reads as ordinary loads inside the
// epilogue, i.e. behind the slab, cost c_attn +43 us and c_fc +60 us per launch; as register loads in front of the slab they pushed
// the kernel over its 256 registers and the slab's own address registers were spilled and reloaded between its pieces.)
// LDS image per wave at `img`: [128 rows][NP] f32x2, then bias'[64], cs[64].
#define FOLD_IMG_BYTES(NP) ((NP) * 1024 + 512)
template <int NP>
__device__ __forceinline__ void fold_dma(const Epilogue& ep, int row0, int col0, int lane, uint32_t img_lds) {
    const v4i32 sp = make_srd(ep.ln.in_part + (int64_t)row0 * NP * 2, (int64_t)NP * 1024);
#pragma unroll
    for (int p = 0; p < NP; p++) dma16(sp, img_lds + p * 1024, p * 1024 + lane * 16);
    const v4i32 sb = make_srd(ep.bias + col0, 256), sc = make_srd(ep.ln.cs + col0, 256);
    if (lane < 16) {            // 16 lanes x 16 bytes each; the other lanes must not write (the next wave's image follows)
        dma16(sb, img_lds + NP * 1024, lane * 16);
        dma16(sc, img_lds + NP * 1024 + 256, lane * 16);
    }
}
template <int KIND, int LNM, int NP, int NPRE>
__device__ __forceinline__ void epi_prefetch(const Epilogue& ep, int row0, int col, int lane, EpiPre<KIND, LNM, NP>& pre,
                                             const char* fold_img = nullptr) {
    constexpr bool LOADS = KIND == EPI_RESID || KIND == EPI_GELUGRAD;
    constexpr bool LN_IN = (LNM & 1) != 0;
    if constexpr (LN_IN && KIND != EPI_RESID) {
        // from the wave's DMA image (fold_dma); the caller has waited for it
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int s = 0; s < NP; s++) pre.pt[h][s] = *reinterpret_cast<const f32x2*>(fold_img + ((h * 64 + lane) * NP + s) * 8);
        if constexpr (KIND != EPI_GELU_AUX) {          // (the GELU kind re-reads them per chunk, see epi_tile)
            const float* vb = reinterpret_cast<const float*>(fold_img + NP * 1024) + (lane & 7) * 8;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(vb), b1 = *reinterpret_cast<const f32x4*>(vb + 4);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(vb + 64), a1 = *reinterpret_cast<const f32x4*>(vb + 68);
#pragma unroll
            for (int j = 0; j < 4; j++) { pre.b[j] = b0[j]; pre.b[4 + j] = b1[j]; pre.lnA[j] = a0[j]; pre.lnA[4 + j] = a1[j]; }
        }
        return;
    }
    if constexpr (LN_IN) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const f32x2* __restrict__ pp = reinterpret_cast<const f32x2*>(ep.ln.in_part) + (int64_t)(row0 + h * 64 + lane) * NP;
#pragma unroll
            for (int s = 0; s < NP; s++) pre.pt[h][s] = pp[s];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) pre.b[j] = 0.f;
    if (KIND != EPI_GELUGRAD && ep.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.bias + col), b1 = *reinterpret_cast<const f32x4*>(ep.bias + col + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { pre.b[j] = b0[j]; pre.b[4 + j] = b1[j]; }
    }
    if constexpr (LN_IN) {
        const float* pa = KIND == EPI_RESID ? ep.ln.gamma : ep.ln.cs;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + col), a1 = *reinterpret_cast<const f32x4*>(pa + col + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { pre.lnA[j] = a0[j]; pre.lnA[4 + j] = a1[j]; }
        if (KIND == EPI_RESID) {
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(ep.ln.beta + col), c1 = *reinterpret_cast<const f32x4*>(ep.ln.beta + col + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) { pre.lnB[j] = c0[j]; pre.lnB[4 + j] = c1[j]; }
        }
    }
    if constexpr (LOADS) {
        const int rl = lane >> 3;
#pragma unroll
        for (int c = 0; c < NPRE; c++) pre.opnd[c] = epi_opnd<KIND, LNM>(ep, row0, col, c, rl);
    }
}
// NI = 16-row groups of the wave's tile: 8 (128 rows: the 256x256 kernels), 4 (64 rows: the 128x128 kernel)
template <int KIND, bool XOR_STG, int LNM = 0, int NP = 1, int NPRE = 2, int NI = 8>
__device__ __forceinline__ void epi_tile(const Epilogue& ep, bf16_t* __restrict__ C, int ldc, int row0, int col, float* stg,
                                         int lane, const f32x4 (&acc)[NI][4], EpiPre<KIND, LNM, NP>& pre, f32x2* lstat = nullptr,
                                         f32x2* lrow = nullptr, const char* fold_vec = nullptr) {
    constexpr bool LOADS = KIND == EPI_RESID || KIND == EPI_GELUGRAD;
    constexpr bool LN_IN = (LNM & 1) != 0, LN_OUT = (LNM & 2) != 0;
    constexpr bool FOLD_VEC_LDS = LN_IN && KIND == EPI_GELU_AUX;
    const int rl = lane >> 3;
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; j++) cs[j] = 0.f;
    // row r of the wave's 128 keeps its (mean, rstd) at lrow[r * RS]: a strip of its own (residual kinds), or -- fold kinds -- in
    // place over the first partial of the row in the DMA image (a lane overwrites only the two rows it has read itself)
    constexpr int RS = (LN_IN && KIND != EPI_RESID) ? NP : 1;
    if constexpr (LN_IN) {
        // (mean, rstd) of the wave's 128 rows, once per item: lane l merges the partials of rows l and l + 64 (epi_prefetch) and
        // parks the pair in LDS; every chunk then takes its row's pair with one ds_read_b64.
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float mu = 0.f;
#pragma unroll
            for (int s = 0; s < NP; s++) mu += pre.pt[h][s][0];
            mu *= 1.0f / (float)NP;
            float m2 = 0.f;
#pragma unroll
            for (int s = 0; s < NP; s++) {
                const float d = pre.pt[h][s][0] - mu;
                m2 += pre.pt[h][s][1] + 256.0f * d * d;
            }
            lrow[(h * 64 + lane) * RS] = (f32x2){mu, __builtin_amdgcn_rsqf(m2 * (1.0f / (256.0f * (float)NP)) + ep.ln.eps)};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < 2 * NI; c++) {
        const int i = c >> 1, it = c & 1;
        if (it == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (XOR_STG) {
                    const int r = lane & 15, cc = j * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4*>(stg + r * 64 + ((cc ^ r) << 2)) = acc[i][j];
                } else {
                    *reinterpret_cast<f32x4*>(stg + (lane & 15) * 68 + j * 16 + (lane >> 4) * 4) = acc[i][j];
                }
            }
        }
        if (LOADS && c + NPRE < 2 * NI) pre.opnd[c + NPRE] = epi_opnd<KIND, LNM>(ep, row0, col, c + NPRE, rl);
        const int r16 = it * 8 + rl;
        f32x4 v0, v1;
        if (XOR_STG) {
            const int c0 = (lane & 7) * 2;
            v0 = *reinterpret_cast<const f32x4*>(stg + r16 * 64 + ((c0 ^ r16) << 2));
            v1 = *reinterpret_cast<const f32x4*>(stg + r16 * 64 + (((c0 + 1) ^ r16) << 2));
        } else {
            v0 = *reinterpret_cast<const f32x4*>(stg + r16 * 68 + (lane & 7) * 8);
            v1 = *reinterpret_cast<const f32x4*>(stg + r16 * 68 + (lane & 7) * 8 + 4);
        }
        const int row = row0 + i * 16 + r16;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        float ln_mu = 0.f, ln_rs = 1.f;
        if constexpr (LN_IN) {
            const f32x2 st = lrow[(c * 8 + rl) * RS];   // the chunk's row inside the wave's 128: 8 c + rl
            ln_mu = st[0];
            ln_rs = st[1];
        }
        if (LN_IN && KIND != EPI_RESID) {
            // rstd * acc - rstd*mean * colsum(gamma o W) + (beta.W + b)
            const float mrs = ln_mu * ln_rs;
            if constexpr (FOLD_VEC_LDS) {
                // bias' and the column sums are re-read from the wave's DMA image for every chunk (sixteen registers less across the
                // epilogue: with them held, the GELU fold kind at 3 segments went over 256 registers and hipcc put a full
                // `s_waitcnt vmcnt(0)` at the top of its k-step -- c_fc of C4 372 us against 318 unfused)
                // (the offset is made opaque per chunk so that the sixteen chunks' reads are not merged back into registers; a
                // `volatile` read would do that too, but hipcc puts `s_waitcnt vmcnt(0)` in front of every volatile access -- 64 full
                // drains of the store queue per item)
                int voff = (lane & 7) * 32;
                asm volatile("" : "+v"(voff));
                const f32x4* vb = reinterpret_cast<const f32x4*>(fold_vec + voff);
                const f32x4 b0 = vb[0], b1 = vb[1], a0 = vb[16], a1 = vb[17];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    v[j] = fmaf(ln_rs, v[j], fmaf(-mrs, a0[j], b0[j]));
                    v[4 + j] = fmaf(ln_rs, v[4 + j], fmaf(-mrs, a1[j], b1[j]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = fmaf(ln_rs, v[j], fmaf(-mrs, pre.lnA[j], pre.b[j]));
            }
        } else if (KIND != EPI_GELUGRAD) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += pre.b[j];
        }
        if (KIND == EPI_GELU_AUX) {
            if (ep.aux) {           // (inference passes do not keep the pre-activation: one output stream less)
                bf16x8 a;
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = (bf16_t)v[j];
                // streamed once, consumed by a later kernel: non-temporal, so the tile does not evict the A panels and
                // the weight matrix from this XCD's 4 MiB L2 (same-box A/B: c_fc 470 -> 408 us, c_attn 306 -> 279 us)
                __builtin_nontemporal_store(a, reinterpret_cast<bf16x8*>((bf16_t*)ep.aux + (int64_t)row * ep.ldaux + col));
            }
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = gelu_f<false>(v[j]);
        } else if (KIND == EPI_GELUGRAD) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] *= gelu_grad_f<false>((float)pre.opnd[c][j]);
        } else if (KIND == EPI_RESID) {
            if (ep.drop.thr) {
                const uint32_t rh = drop_row_hash(ep.drop, (uint32_t)row);
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = apply_drop_rc(ep.drop, rh, (uint32_t)(col + j), v[j]);
            }
            if constexpr (LN_IN) {
                // the residual operand is LayerNorm(raw row), rounded to bf16 like the stored tensor it replaces
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += (float)(bf16_t)(((float)pre.opnd[c][j] - ln_mu) * ln_rs * pre.lnA[j] + pre.lnB[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += (float)pre.opnd[c][j];
            }
        }
        if constexpr (KIND == EPI_PLAIN32) {
            float* o32 = reinterpret_cast<float*>(C) + (int64_t)row * ldc + col;
            if (col + 8 <= ep.n_cols) {
                __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(o32));
                __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(o32 + 4));
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) if (col + j < ep.n_cols) o32[j] = v[j];
            }
            continue;
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (bf16_t)v[j];
        // Residual-stream rows of the fused block path are read next by a GEMM (the fold), not streamed by a LayerNorm kernel:
        // default-policy stores there (inference forward of C2 7.88 -> 7.67 ms same-box; non-temporal everywhere else)
#ifndef EPI_LNOUT_NT_STORE
        if ((LNM != 0 || XC_PLAIN_RESID) && KIND == EPI_RESID) *reinterpret_cast<bf16x8*>(C + (int64_t)row * ldc + col) = o;
        else if (XC_PLAIN_PLAINKIND && KIND == EPI_PLAIN && LNM == 0) *reinterpret_cast<bf16x8*>(C + (int64_t)row * ldc + col) = o;
        else
#endif

        __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(C + (int64_t)row * ldc + col));
        if (ep.colsum) {
#pragma unroll
            for (int j = 0; j < 8; j++) cs[j] += (float)o[j];
        }
        if constexpr (LN_OUT) {
            // statistics of the STORED (rounded) values: what a LayerNorm kernel reading the tensor back would see
            float f[8], sm = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) { f[j] = (float)o[j]; sm += f[j]; }
            float mn = sm * 0.125f, q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) { const float d = f[j] - mn; q = fmaf(d, d, q); }
            // Chan's update for equal counts n: mean = (ma + mb) / 2, M2 = M2a + M2b + (ma - mb)^2 * n / 2
            {
                const float pm = CMP_DPP_F(mn, 0xB1), pq = CMP_DPP_F(q, 0xB1), d = pm - mn;        // lane ^ 1: n = 8
                mn = 0.5f * (mn + pm); q = (q + pq) + d * d * 4.0f;
            }
            {
                const float pm = CMP_DPP_F(mn, 0x4E), pq = CMP_DPP_F(q, 0x4E), d = pm - mn;        // lane ^ 2: n = 16
                mn = 0.5f * (mn + pm); q = (q + pq) + d * d * 8.0f;
            }
            {
                const float pm = CMP_DPP_F(mn, 0x141), pq = CMP_DPP_F(q, 0x141), d = pm - mn;      // the other quad of the eight: n = 32
                mn = 0.5f * (mn + pm); q = (q + pq) + d * d * 16.0f;
            }
            if ((lane & 7) == 0) lstat[i * 16 + r16] = (f32x2){mn, q};
        }
    }
    if (ep.colsum) {
        // this wave's 128 rows: 16 chunks per lane above, then the 8 lanes that share the columns (lane bits 3..5)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float t = cs[j];
            t += __shfl_xor(t, 8);
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            if (lane < 8) atomicAdd(ep.colsum + col + j, t);
        }
    }
}
// the kind a launch may use (full tiles only; everything else takes the generic run-time epilogue)
static int epi_kind_of(const Epilogue& ep, int M, int N, bool swap, bool slabs, int tile = 256) {
    if (!swap || slabs || ep.out_fp32 || ep.dbg_nostore || ep.atomic || M % tile || N % tile) return EPI_GENERIC;
    if (ep.act == 1) return (!ep.resid && !ep.drop.thr) ? EPI_GELU_AUX : EPI_GENERIC;
    if (ep.act == 2) return (!ep.resid && !ep.drop.thr && !ep.bias) ? EPI_GELUGRAD : EPI_GENERIC;
    if (ep.resid) return EPI_RESID;
    return ep.drop.thr ? EPI_GENERIC : EPI_PLAIN;
}

// RING (launches of at most one workgroup per CU -- the reference's default configuration runs GEMMs of 16-64 tiles on 256 CUs): a ring of
// four k-stages, three in flight, LDS-DMA issued from inline asm behind hand-counted s_waitcnt vmcnt.  The two-stage loop below drains
// every stage at its barrier (hipcc's wait-count pass sees the pending LDS write of the builtin and puts vmcnt(0) in front of the next
// fragment read), so a k-step of such a launch costs one operand round trip, 0.5 us from a warm L2 and ~1 us in the model, where the
// A operand was written by the previous kernel; with 128 KiB of stages in flight the round trips overlap.
template <bool A_KM, bool B_KM, bool SWAP, int EPI = EPI_GENERIC, bool RING = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_fast_kernel(int M, int N, int K, const bf16_t* __restrict__ A, int lda,
                                                                const bf16_t* __restrict__ B, int ldb, void* C, int ldc,
                                                                Epilogue ep, int ktiles_per_split, int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A img | B img]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // bijective XCD remap (blocks b and b+8 share an XCD): XCD x works on a contiguous run of tiles
    int tile;
    {
        const int orig = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, x = orig & 7;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (orig >> 3);
    }
    const int m0 = (tile / tiles_n) * G_BM, n0 = (tile % tiles_n) * G_BN;
    const int nk = cdiv(K, G_BK);
    const int kt0 = blockIdx.y * ktiles_per_split;
    const int kt1 = min(nk, kt0 + ktiles_per_split);

    // buffer descriptors: tile-relative bases, range = the rest of the tensor (clamped): slow-dim overflow reads 0
    const int64_t a_rows = A_KM ? (int64_t)M : (int64_t)K, b_rows = B_KM ? (int64_t)N : (int64_t)K;
    const bf16_t* abase = A_KM ? A + (int64_t)m0 * lda : A + m0;
    const bf16_t* bbase = B_KM ? B + (int64_t)n0 * ldb : B + n0;
    const int64_t a_bytes = (A_KM ? (a_rows - m0) * lda : a_rows * lda - m0) * 2;
    const int64_t b_bytes = (B_KM ? (b_rows - n0) * ldb : b_rows * ldb - n0) * 2;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, (int)min(a_bytes, (int64_t)0x7FFFFFF0), 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, (int)min(b_bytes, (int64_t)0x7FFFFFF0), 0x00020000);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (RING) {
        static_assert(A_KM && B_KM, "ring: both operands K-contiguous");
        constexpr int NST = 4;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
        const v4i32 sa = make_srd(abase, a_bytes), sb = make_srd(bbase, b_bytes);
        auto issue = [&](int kt, int slot) {              // this wave's 4 + 4 one-KiB pieces of stage kt (the layout of glds_tile<true>)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int p = wave * 4 + i, r = 8 * p + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
                dma16(sa, lds0 + slot * 2 * G_IMG + p * 1024, r * lda * 2 + (kt * G_BK + c * 8) * 2);
                dma16(sb, lds0 + slot * 2 * G_IMG + G_IMG + p * 1024, r * ldb * 2 + (kt * G_BK + c * 8) * 2);
            }
        };
#pragma unroll
        for (int i = 0; i < NST - 1; i++)
            if (kt0 + i < kt1) issue(kt0 + i, i);
        for (int kt = kt0; kt < kt1; kt++) {
            const int rel = kt - kt0, slot = rel & (NST - 1);
            // stage kt has landed once at most the (up to two) younger stages of this wave are outstanding, eight pieces each; the
            // barrier then vouches for the other waves' pieces -- and for their having finished with the slot that is refilled next
            const int younger = min(NST - 2, kt1 - 1 - kt);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NST - 1 < kt1) issue(kt + NST - 1, (rel + NST - 1) & (NST - 1));
            const char* ia = smem + slot * 2 * G_IMG;
            const char* ib = ia + G_IMG;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    fa[i] = g_frag<A_KM>(ia, wm * 4 + i, ks, lane);
                    fb[i] = g_frag<B_KM>(ib, wn * 4 + i, ks, lane);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // the epilogue's staging areas lie over slot 0
        asm volatile("" ::: "memory");
    } else {
    if (kt0 < kt1) {
        glds_tile<A_KM>(ra, smem, lda * 2, kt0 * G_BK, wave, lane);
        glds_tile<B_KM>(rb, smem + G_IMG, ldb * 2, kt0 * G_BK, wave, lane);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt++) {
        const int st = (kt - kt0) & 1;
        const char* ia = smem + st * 2 * G_IMG;
        const char* ib = ia + G_IMG;
        if (kt + 1 < kt1) {
            char* na = smem + (st ^ 1) * 2 * G_IMG;
            glds_tile<A_KM>(ra, na, lda * 2, (kt + 1) * G_BK, wave, lane);
            glds_tile<B_KM>(rb, na + G_IMG, ldb * 2, (kt + 1) * G_BK, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                fa[i] = g_frag<A_KM>(ia, wm * 4 + i, ks, lane);
                fb[i] = g_frag<B_KM>(ib, wn * 4 + i, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();     // also drains the LDS-DMA of the next tile (vmcnt(0) is emitted with the barrier)
    }
    }
    if (SWAP) {
        // Epilogue through LDS (the pipeline buffers are free after the loop's last barrier): each wave parks its
        // 64x64 fp32 tile, 32 rows at a time, in a private [32][68]-float region (b128 writes and reads conflict-free),
        // then every lane finishes 8 consecutive columns of one row: bias/aux/residual/output are all 16-byte,
        // row-contiguous accesses (8 lanes = one full 128-B line of bf16).
        if constexpr (EPI != EPI_GENERIC) {
            // whole 128x128 tiles: the compile-time kinds of the persistent kernels (straight-line chunks, stores fire-and-forget,
            // fused bias-gradient column sums) -- the run-time epilogue below waits out a store round trip per chunk, which at the
            // reference's default configuration (1 024 tokens: 16-64 tiles per GEMM) was a third of a 12 us launch
            float* stg16 = reinterpret_cast<float*>(smem) + wave * (16 * 68);
            const int col = n0 + wn * 64 + (lane & 7) * 8;
            EpiPre<EPI, 0, 1> pre;
            epi_prefetch<EPI, 0, 1, 2>(ep, m0 + wm * 64, col, lane, pre);
            epi_tile<EPI, false, 0, 1, 2, 4>(ep, (bf16_t*)C, ldc, m0 + wm * 64, col, stg16, lane, acc, pre);
            return;
        }
        float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 68);
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int ii = 0; ii < 2; ii++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<f32x4*>(stg + (ii * 16 + (lane & 15)) * 68 + j * 16 + (lane >> 4) * 4) = acc[half * 2 + ii][j];
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int rl = it * 8 + (lane >> 3);
                const int row = m0 + wm * 64 + half * 32 + rl;
                const int col = n0 + wn * 64 + (lane & 7) * 8;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + rl * 68 + (lane & 7) * 8);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + rl * 68 + (lane & 7) * 8 + 4);
                if (row < M && col < N) epilogue_store8(ep, C, ldc, row, col, N, v0, v1);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                    int col = n0 + wn * 64 + j * 16 + (lane & 15);
                    if (row < M && col < N) epilogue_store<bf16_t, false>(ep, C, ldc, row, col, acc[i][j][r]);
                }
    }
}

// =================================================================================================
// Item scheduling of the persistent kernels.  Items are dealt in eight groups (item & 7 = group = the label of the
// workgroups that share an XCD under round-robin placement, item >> 3 = index inside the group; item_coords() maps a group
// to a consecutive range of tiles so that an XCD's L2 serves one A row panel).  Round 2 strode statically
// (next = item + gridDim.x): a workgroup that starts late, shares its CU with an RCCL kernel or is missing altogether
// (CU cap) then keeps its whole share while the others idle.  Now every workgroup takes its FIRST item statically
// (no latency in front of the first load) and CLAIMS each later one from its group's counter -- one returning atomic,
// issued a whole item ahead of its use -- and steals from the other groups once its own is exhausted, so a slow
// workgroup costs what it cannot do, not what it was assigned.  Eight counters on eight 128-byte lines; two counter
// sets alternate between consecutive launches on a stream (a launch zeroes the set of the next one: no memset node).
// Speed only: which workgroup runs an item never changes a result.
// =================================================================================================
#define SCHED_SET_WORDS (8 * 32 + 32 + 512)     // 8 group counters on their own 128-byte lines, `started`, 512 first-item flags
// Only wave-uniform values live in the struct; what thread 0 draws (the pending claim, the answer of the exchange on its own
// first-item flag, the set of exhausted groups) is PARKED IN LDS (slot[1..3]) before the k-loop starts: a single vector
// register kept alive across that loop, or one LDS write inside it, changed hipcc's schedule of the fragment reads and cost
// the forward / dgrad GEMMs 10-30 % (same-box A/B against the round-2 kernel).
// slot[0] (256-kernel: slot[0] / slot[4] alternately) next item   slot[1] own first-item flag as found (non-zero: taken by another
// workgroup)   slot[2] parked claim (deep-pipeline kernel) / stolen item   slot[3] exhausted groups
struct ItemPuller {
    uint32_t* ctr;          // this launch's counter set; null = static striding
    int nitems, grp;
    __device__ __forceinline__ int gcount(int x) const { return (nitems >> 3) + (x < (nitems & 7) ? 1 : 0); }
    __device__ __forceinline__ int nstatic(int x) const { const int g = (int)gridDim.x; return x < g ? (g - x + 7) >> 3 : 0; }   // workgroups labelled x
    __device__ __forceinline__ uint32_t* started() const { return ctr + 8 * 32; }
    __device__ __forceinline__ uint32_t* taken() const { return ctr + 8 * 32 + 32; }
    __device__ __forceinline__ void init(const Epilogue& ep, int nitems_, int tid) {
        ctr = ep.sched;
        nitems = nitems_;
        grp = blockIdx.x & 7;
        if (ep.sched_clear && blockIdx.x == 0) {
            for (int i = tid; i < SCHED_SET_WORDS; i += blockDim.x)
                if (i >= 8 * 32 || (i & 31) == 0) __hip_atomic_store(ep.sched_clear + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // thread 0, kernel start, BEFORE the first loads: take this workgroup's own first item (an exchange on its flag, in flight
    // while the item's first k-slab is loaded speculatively -- a workgroup that got its CU late, the chip shared with an RCCL
    // kernel, finds its item done by a workgroup that had run out of work), count it as started, and claim the item after.
    __device__ __forceinline__ void start(int tid, uint32_t& first_old, uint32_t& pend) {
        first_old = 0;
        pend = 0;
        if (ctr && tid == 0) {
            first_old = __hip_atomic_exchange(taken() + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(started(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend = __hip_atomic_fetch_add(ctr + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // thread 0: draw the next index of the current group (the value is looked at a whole item later).  A compiler-counted
    // atomic on purpose: beside the hand-counted LDS-DMA of the deep-pipeline kernel a compiler-sized wait can only be stronger
    // than needed (operations the compiler cannot see are YOUNGER entries of the same in-order counter), never weaker.
    __device__ __forceinline__ uint32_t claim(int tid) const {
        uint32_t pend = 0x7FFFFFFFu;                                   // (a group whose items are all first items has nothing to claim)
        if (ctr && tid == 0)
            pend = __hip_atomic_fetch_add(ctr + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return pend;
    }
    // thread 0: the item a claim stands for, or -1 when the group has run dry
    __device__ __forceinline__ int resolve(uint32_t pend, int* slot) const {
        if (pend < 0x7FFFFFFFu) {
            const int idx = nstatic(grp) + (int)pend;
            if (idx < gcount(grp)) return idx * 8 + grp;
        }
        slot[3] |= 1 << grp;
        return -1;
    }
    // thread 0: the item the parked claim stands for, or -1 when the group has run dry
    __device__ __forceinline__ int claimed(int* slot) const { return resolve((uint32_t)slot[2], slot); }
    // WAVE 0 (all 64 lanes), BEFORE the epilogue of a workgroup whose own group has run dry: one wave-wide look at the eight
    // group counters (lanes 0-7) and `started` (lane 8); the round trip hides under the epilogue, steal() consumes it.
    __device__ __forceinline__ uint32_t peek(int lane) const {
        uint32_t v = 0;
        if (lane < 9) v = __hip_atomic_load(ctr + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    }
    // WAVE 0 (all 64 lanes), after that epilogue -- only the end of a launch gets here: take an item from another group
    // that peek() saw work in, else the first item of a workgroup that has not started yet.  Every lane returns the same
    // value; in the common case (nothing left anywhere) no memory operation is issued here.
    __device__ __forceinline__ int steal(int lane, uint32_t seen, int* slot) const {
        const unsigned dd = (unsigned)__builtin_amdgcn_readfirstlane(slot[3]);
        const bool cand = lane < 8 && !((dd >> lane) & 1u) && (gcount(lane) - nstatic(lane)) > (int)seen;
        unsigned m = (unsigned)__ballot(cand) & 0xFFu;
        m = ((m >> grp) | (m << (8 - grp))) & 0xFFu;          // bit g: group (grp + g) & 7
        while (m) {
            const int g = __builtin_ctz(m);
            m &= m - 1;
            const int x = (grp + g) & 7;
            const int left = gcount(x) - nstatic(x);
            uint32_t c = 0xFFFFFFFFu;
            if (lane == 0) c = __hip_atomic_fetch_add(ctr + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c < (uint32_t)left) return (nstatic(x) + (int)c) * 8 + x;
            if (lane == 0) slot[3] |= 1 << x;
        }
        const int g = (int)gridDim.x;
        const int nstart = __builtin_amdgcn_readlane((int)seen, 8);
        if (nstart >= g) return -1;
        for (int base = 0; base < g; base += 64) {
            const int w = base + lane;
            const uint32_t f = w < g ? __hip_atomic_load(taken() + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            unsigned long long mm = __ballot(f == 0u);
            while (mm) {
                const int bit = __builtin_ctzll(mm);
                mm &= mm - 1;
                uint32_t old = 1u;
                if (lane == 0) old = __hip_atomic_exchange(taken() + base + bit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_readfirstlane((int)old) == 0) return base + bit;      // that workgroup's static item
            }
        }
        return -1;
    }
    // every thread, after a stolen item came back through the slot: later claims draw from that item's group
    __device__ __forceinline__ void follow(int item) { grp = item & 7; }
};

// =================================================================================================
// bf16 large-tile path: 256x256x64 workgroup tile, 8 waves (2 x 4) of 128x64, PERSISTENT workgroups (one per CU).
//   * 128 FLOP per byte staged into LDS (the 128x128 tile: 64) -- the 128^2 kernel saturates L2->LDS bandwidth at
//     ~1 PFLOP/s (measured on the K=2048 shapes);
//   * a workgroup walks its list of (tile, k-split) items; right after an item's last k-step it issues the NEXT
//     item's first k-slab into stage 0, so that load flies under the epilogue (which parks accumulators in stage 1);
//   * same LDS images / swizzles / direct-to-LDS staging / range-checked descriptors as the 128^2 fast path.
// LDS: 2 stages x (A 32 KiB + B 32 KiB) = 128 KiB.
// =================================================================================================
#if !defined(COMPOSER_EXPERIMENTS) || !defined(GEMM_DIAG)      // measurement ladders exist in experiments builds only
#undef GEMM_DIAG
#define GEMM_DIAG 0
#endif
#define H_BM 256
#define H_BN 256
#define H_IMG (256 * 64 * 2)
__device__ __forceinline__ int cslow_off512(int k, int c) { return k * 512 + ((c ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1)) << 4); }

template <bool KM, int AUX = 0>
__device__ __forceinline__ void glds_tile256(__amdgpu_buffer_rsrc_t rs, char* img, int ld_bytes, int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = wave * 4 + i;           // 32 pieces of 1 KiB
        int voff;
        if (KM) {
            const int r = 8 * p + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voff = r * ld_bytes + (k0 + c * 8) * 2;
        } else {
            const int k = 2 * p + (lane >> 5);
            const int c = (lane & 31) ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1);
            voff = (k0 + k) * ld_bytes + c * 16;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(img + p * 1024), 16, voff, 0, 0, AUX);
    }
}
template <bool KM>
__device__ __forceinline__ bf16x8 h_frag(const char* img, int t16, int ks, int lane) {
    if (KM) {
        int row = t16 * 16 + (lane & 15);
        int c = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(img + kmaj_off(row, c));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int k = ks * 32 + 8 * g + q;
        const int c = t16 * 2 + (p >> 1);
        const int sub = (p & 1) * 8;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + cslow_off512(k, c) + sub));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + cslow_off512(k + 4, c) + sub));
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[j] = lo[j];
            f[4 + j] = hi[j];
        }
        return f;
    }
}

// LayerNorm epilogues, in stage 1 behind the waves' staging areas: [4 wave columns][256 rows] f32x2 (the producers' 64-column
// partials), then [8 waves][128 rows] f32x2 (the consumers' merged (mean, rstd) of their rows)
#define LN_STAT_OFF (2 * H_IMG + 40960)
template <bool A_KM, bool B_KM, bool SWAP, int EPI = EPI_GENERIC, int LNM = 0, int NP = 1>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(int M, int N, int K, const bf16_t* __restrict__ A, int lda,
                                                               const bf16_t* __restrict__ B, int ldb, void* C, int ldc,
                                                               Epilogue ep, int ktiles_per_split, int nsplit, int tiles_n,
                                                               int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A img | B img]
    // Cache policy of the A-operand DMA.  The two c_proj GEMMs of the fused block path read att / g for the last time and write the
    // rows the NEXT GEMM folds a LayerNorm into: their A stream is non-temporal (aux = 2) and their output stores default-policy,
    // so the consumer finds its A operand in the Infinity Cache instead of HBM (a fold GEMM on a cold A: +30...+100 us per launch
    // at C2, tools/chain_bench.py; inference forward of C2 7.51 -> 7.22 ms same-box with both, profiles/r5_01_ln_fused.txt).
    constexpr int A_AUX = (EPI == EPI_RESID && LNM != 0) ? 2 : 0;     // (either c_proj kind alone measured worse than both: r5_01)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nitems = ntiles * nsplit;
    const int nk = cdiv(K, G_BK);

    auto item_coords = [&](int item, int& m0, int& n0, int& kt0, int& kt1) {
        // bijective XCD remap over items: the items one XCD works on are consecutive (shared A row panel in its L2)
        const int q = nitems >> 3, r = nitems & 7, x = item & 7;
        // ep.rev: the group's run is walked from its END.  A kernel whose A operand is the tensor the PREVIOUS kernel wrote finds
        // the rows that kernel wrote last still in this XCD's L2 / the Infinity Cache, and the rows it wrote first long evicted:
        // reading in the producer's order misses everywhere (LRU under a stream larger than the cache), reading backwards hits on
        // the tail.  The model alternates the direction from GEMM to GEMM (model.hip).
        const int idx = ep.rev ? (q + (x < r ? 1 : 0)) - 1 - (item >> 3) : (item >> 3);
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
        // split-major: consecutive items (= one XCD's group) are the TILES of one K range, which share that range's A and B
        // panels through the XCD's L2.  (Tile-major -- the splits of one tile side by side -- gave an XCD sixteen disjoint K
        // ranges: no operand byte shared, 2.4x the algorithmic bytes at the L2's memory side in the wgrad launches,
        // profiles/r4_01_hbm_traffic.json.)
        const int sp = lin / ntiles, tile = lin - sp * ntiles;
        m0 = (tile / tiles_n) * H_BM;
        n0 = (tile % tiles_n) * H_BN;
        kt0 = sp * ktiles_per_split;
        kt1 = min(nk, kt0 + ktiles_per_split);
    };
    auto make_rsrc = [&](const bf16_t* P, bool km, int64_t rows, int ld, int t0) {
        const bf16_t* base = km ? P + (int64_t)t0 * ld : P + t0;
        const int64_t bytes = (km ? (rows - t0) * ld : rows * ld - t0) * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)min(bytes, (int64_t)0x7FFFFFF0), 0x00020000);
    };

    int item = blockIdx.x;
    if (item >= nitems) return;
    ItemPuller pl;
    pl.init(ep, nitems, tid);
    int* slot = reinterpret_cast<int*>(smem + 4 * H_IMG);      // scheduler words of thread 0 (see ItemPuller)
    uint32_t fo, pend;
    pl.start(tid, fo, pend);                                   // in flight under the first loads
    int m0, n0, kt0, kt1;
    item_coords(item, m0, n0, kt0, kt1);
    __amdgpu_buffer_rsrc_t ra = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0);
    __amdgpu_buffer_rsrc_t rb = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0);
    if (kt0 < kt1) {
        glds_tile256<A_KM, A_AUX>(ra, smem, lda * 2, kt0 * G_BK, wave, lane);
        glds_tile256<B_KM>(rb, smem + H_IMG, ldb * 2, kt0 * G_BK, wave, lane);
    }
    bool run = true;      // false: this workgroup's first item was taken by another one before it got to run
    if (pl.ctr) {
        if (tid == 0) { slot[1] = (int)fo; slot[2] = (int)pend; slot[3] = 0; }
        __syncthreads();
        run = __builtin_amdgcn_readfirstlane(slot[1]) == 0;
    }
    while (true) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();      // stage 0 of this item has landed (vmcnt(0) rides on the barrier); previous epilogue left stage 1
        bf16x8 diag_a[2][2][4], diag_b[2][4];       // GEMM_DIAG 2 / 4 only
        if (!run) kt1 = kt0;                        // (nothing of the scheduler inside the k-loop, see ItemPuller)
#ifdef GEMM_EARLY_SLAB
        // Static striding, an even number of k-steps (the last one computes out of stage 1, stage 0 is free by then): the NEXT
        // item's first k-slab is issued at the top of this item's LAST k-step, like any other stage, instead of behind the loop.
        // Memory reads are delivered in issue order, so with the slab issued just in front of the epilogue every operand load of
        // the epilogue (residual, gelu' input, bias) waited until the slab had landed; a k-step earlier it has landed by then.
        // Not for the fold kinds: their LDS-DMA image must be requested in front of the slab and lives in stage 1.
        constexpr bool EARLY_OK = SWAP && !((LNM & 1) != 0 && EPI != EPI_RESID && EPI != EPI_GENERIC);
        const bool early = EARLY_OK && !pl.ctr && run && ((kt1 - kt0) & 1) == 0 && kt1 > kt0 && item + (int)gridDim.x < nitems;
        int m0n = 0, n0n = 0, kt0n = 0, kt1n = 0;
        __amdgpu_buffer_rsrc_t ran = ra, rbn = rb;
        if (early) {
            item_coords(item + gridDim.x, m0n, n0n, kt0n, kt1n);
            ran = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0n);
            rbn = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0n);
        }
#else
        constexpr bool early = false;
#endif
        for (int kt = kt0; kt < kt1; kt++) {
            const int st = (kt - kt0) & 1;
            const char* ia = smem + st * 2 * H_IMG;
            const char* ib = ia + H_IMG;
#ifdef GEMM_EARLY_SLAB
            if (early && kt + 1 == kt1) {
                glds_tile256<A_KM, A_AUX>(ran, smem, lda * 2, kt0n * G_BK, wave, lane);
                glds_tile256<B_KM>(rbn, smem + H_IMG, ldb * 2, kt0n * G_BK, wave, lane);
            }
#endif
            // GEMM_DIAG (measurement builds only -- the results are wrong; tools/ubench/gemm_latency_probe.py):
            // 1 = no MFMA (fragment reads kept alive), 2 = no LDS fragment reads (the first step's fragments reused),
            // 3 = no DMA inside the k-loop, 4 = neither reads nor DMA (MFMA + barrier only), 5 = as 1 with TWO stage loads per step
            if (kt + 1 < kt1 && GEMM_DIAG != 3 && GEMM_DIAG != 4) {
                char* na = smem + (st ^ 1) * 2 * H_IMG;
                glds_tile256<A_KM, A_AUX>(ra, na, lda * 2, (kt + 1) * G_BK, wave, lane);
                glds_tile256<B_KM>(rb, na + H_IMG, ldb * 2, (kt + 1) * G_BK, wave, lane);
#if GEMM_DIAG == 5
                // bandwidth probe: a second, different slab per step into the same buffer -> 128 KiB in flight per CU
                const int k2 = (kt + 1 + (kt1 - kt0) / 2) % (kt1 - kt0) + kt0;
                glds_tile256<A_KM, A_AUX>(ra, na, lda * 2, k2 * G_BK, wave, lane);
                glds_tile256<B_KM>(rb, na + H_IMG, ldb * 2, k2 * G_BK, wave, lane);
#endif
            }
#if GEMM_DIAG == 0
            {
                // The k-step as an explicit software pipeline, its order pinned: the fragments of MFMA group g+1 are requested in
                // front of group g's sixteen MFMAs.  hipcc finds this order by itself in some surroundings and not in others (the
                // same loop came out with every fragment read waited for at once when the item scheduler's code was added behind
                // it: forward / dgrad GEMMs -10...-30 %), so it is written down.
                constexpr int RA = A_KM ? 4 : 8, RB = B_KM ? 4 : 8;        // LDS reads per fragment group (transposed reads come in pairs)
                bf16x8 fb0[4], fb1[4], fa0[4], fa1[4];
                auto mfma16 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], const int ih) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (SWAP) acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[ih * 4 + i][j], 0, 0, 0);
                            else acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[ih * 4 + i][j], 0, 0, 0);
                        }
                };
#pragma unroll
                for (int j = 0; j < 4; j++) fb0[j] = h_frag<B_KM>(ib, wn * 4 + j, 0, lane);
#pragma unroll
                for (int i = 0; i < 4; i++) fa0[i] = h_frag<A_KM>(ia, wm * 8 + i, 0, lane);
#pragma unroll
                for (int i = 0; i < 4; i++) fa1[i] = h_frag<A_KM>(ia, wm * 8 + 4 + i, 0, lane);
                mfma16(fa0, fb0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) fb1[j] = h_frag<B_KM>(ib, wn * 4 + j, 1, lane);
#pragma unroll
                for (int i = 0; i < 4; i++) fa0[i] = h_frag<A_KM>(ia, wm * 8 + i, 1, lane);
                mfma16(fa1, fb0, 1);
#pragma unroll
                for (int i = 0; i < 4; i++) fa1[i] = h_frag<A_KM>(ia, wm * 8 + 4 + i, 1, lane);
                mfma16(fa0, fb1, 0);
                mfma16(fa1, fb1, 1);
                // (spreading each prefetch through the MFMAs in front of it -- one fragment per 4 or 2 MFMAs, or ending 4 MFMAs before
                //  its consumer -- measured the same: 27.3-27.5 ms per step for all three orders)
                __builtin_amdgcn_sched_group_barrier(0x100, RB + RA, 0);       // head: B(0), A(0,0)
                __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);            // A(0,1) under group (0,0)
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RB + RA, 0);       // B(1), A(1,0) under group (0,1)
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);            // A(1,1) under group (1,0)
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
#else
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 fb[4];
                const bool rd = (GEMM_DIAG != 2 && GEMM_DIAG != 4) || kt == kt0;
#pragma unroll
                for (int j = 0; j < 4; j++) fb[j] = rd ? h_frag<B_KM>(ib, wn * 4 + j, ks, lane) : diag_b[ks][j];
                if (GEMM_DIAG == 2 || GEMM_DIAG == 4) {
#pragma unroll
                    for (int j = 0; j < 4; j++) diag_b[ks][j] = fb[j];
                }
#pragma unroll
                for (int ih = 0; ih < 2; ih++) {
                    bf16x8 fa[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) fa[i] = rd ? h_frag<A_KM>(ia, wm * 8 + ih * 4 + i, ks, lane) : diag_a[ks][ih][i];
                    if (GEMM_DIAG == 2 || GEMM_DIAG == 4) {
#pragma unroll
                        for (int i = 0; i < 4; i++) diag_a[ks][ih][i] = fa[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
#if GEMM_DIAG == 1 || GEMM_DIAG == 5
                            asm volatile("" ::"v"(fa[i]), "v"(fb[j]));
#else
                            if (SWAP) acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[ih * 4 + i][j], 0, 0, 0);
                            else acc[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[ih * 4 + i][j], 0, 0, 0);
#endif
                        }
                }
            }
#endif
            __syncthreads();
        }
        if (pl.ctr) {                               // the item claimed a whole item ago, through the slot (one more barrier per item)
            if (tid == 0) *slot = pl.claimed(slot);
            __syncthreads();
        }
        // next item's first k-slab goes to stage 0 while this item's epilogue runs out of stage 1
        const int cm0 = m0, cn0 = n0;
        // ... but what a fold epilogue reads from memory is requested FIRST (see fold_dma): in front of the slab, not behind it
        constexpr bool FOLD = SWAP && (LNM & 1) != 0 && EPI != EPI_RESID && EPI != EPI_GENERIC;
        const uint32_t fold_off = 2 * H_IMG + 8 * (16 * 68 * 4) + wave * FOLD_IMG_BYTES(NP);      // behind the waves' staging areas
        if constexpr (FOLD) {
            static_assert(8 * (16 * 68 * 4) + 8 * FOLD_IMG_BYTES(NP) <= 2 * H_IMG, "fold images must fit stage 1");
            if (run) fold_dma<NP>(ep, cm0 + wm * 128, cn0 + wn * 64, lane, (uint32_t)(uintptr_t)(lds_char*)smem + fold_off);
        }
        int next = pl.ctr ? __builtin_amdgcn_readfirstlane(*slot) : (int)(item + gridDim.x < nitems ? item + gridDim.x : -1);
        bool has_next = next >= 0;
        uint32_t pend2 = 0;
        if (has_next) {
            pend2 = pl.claim(tid);
            item_coords(next, m0, n0, kt0, kt1);
            ra = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0);
            rb = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0);
            if (kt0 < kt1 && !early) {          // (early: that slab went out in front of the last k-step)
                glds_tile256<A_KM, A_AUX>(ra, smem, lda * 2, kt0 * G_BK, wave, lane);
                glds_tile256<B_KM>(rb, smem + H_IMG, ldb * 2, kt0 * G_BK, wave, lane);
            }
        }
        uint32_t seen = 0;
        if (!has_next && pl.ctr && wave == 0) seen = pl.peek(lane);
        if (!run) {
        } else if (SWAP) {
            float* stg = reinterpret_cast<float*>(smem + 2 * H_IMG) + wave * (16 * 68);   // stage 1: [16][68] floats per wave
            const int col = cn0 + wn * 64 + (lane & 7) * 8;
            if constexpr (EPI != EPI_GENERIC) {
                EpiPre<EPI, LNM, NP> pre;
                if constexpr (FOLD) {
                    // the wave's own fold image has landed once at most the slab's eight pieces (issued after it) are outstanding
                    if (has_next) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                epi_prefetch<EPI, LNM, NP, EPI_AHEAD>(ep, cm0 + wm * 128, col, lane, pre, smem + fold_off);
                epi_tile<EPI, false, LNM, NP, EPI_AHEAD>(ep, (bf16_t*)C, ldc, cm0 + wm * 128, col, stg, lane, acc, pre,
                                                  reinterpret_cast<f32x2*>(smem + LN_STAT_OFF) + wn * 256 + wm * 128,
                                                  FOLD ? reinterpret_cast<f32x2*>(smem + fold_off)
                                                       : reinterpret_cast<f32x2*>(smem + LN_STAT_OFF + 8192) + wave * 128,
                                                  smem + fold_off + NP * 1024);
            } else {
            const int rbase = cm0 + wm * 128 + (lane >> 3);
            float bias8[8];
            epi_bias8(ep, col, N, bias8);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                EpiIn in[2];          // fetched before the staging round trip through LDS, used after it
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int row = rbase + i * 16 + it * 8;
                    in[it] = epi_fetch8(ep, row, col, row < M && col < N);
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<f32x4*>(stg + (lane & 15) * 68 + j * 16 + (lane >> 4) * 4) = acc[i][j];
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int rl = it * 8 + (lane >> 3);
                    const int row = rbase + i * 16 + it * 8;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + rl * 68 + (lane & 7) * 8);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + rl * 68 + (lane & 7) * 8 + 4);
                    if (row < M && col < N) epi_finish8(ep, C, ldc, row, col, N, v0, v1, bias8, in[it]);
                }
            }
            }
        } else {
            // split-K items: plain f32 atomics, 16 consecutive floats per row per wave-instruction
            float* Cf = (float*)C;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int col = cn0 + wn * 64 + j * 16 + (lane & 15);
                    const int row0 = cm0 + wm * 128 + i * 16 + (lane >> 4) * 4;
                    if (col < N) {
                        if (row0 + 0 < M) atomicAdd(Cf + (int64_t)(row0 + 0) * ldc + col, acc[i][j][0]);
                        if (row0 + 1 < M) atomicAdd(Cf + (int64_t)(row0 + 1) * ldc + col, acc[i][j][1]);
                        if (row0 + 2 < M) atomicAdd(Cf + (int64_t)(row0 + 2) * ldc + col, acc[i][j][2]);
                        if (row0 + 3 < M) atomicAdd(Cf + (int64_t)(row0 + 3) * ldc + col, acc[i][j][3]);
                    }
                }
        }
        if constexpr ((LNM & 2) != 0) {
            // the four wave columns' 64-column partials of every tile row -> the tile's 256-column partial (mean, M2) of that row.
            // Read before the barrier at the top of the next item: its k-loop refills stage 1 only behind that one.
            __syncthreads();
            if (run && tid < 256) {
                const f32x2* ls = reinterpret_cast<const f32x2*>(smem + LN_STAT_OFF);
                const f32x2 p0 = ls[tid], p1 = ls[256 + tid], p2 = ls[512 + tid], p3 = ls[768 + tid];
                const float mu = 0.25f * ((p0[0] + p1[0]) + (p2[0] + p3[0]));
                const float d0 = p0[0] - mu, d1 = p1[0] - mu, d2 = p2[0] - mu, d3 = p3[0] - mu;
                const float m2 = ((p0[1] + p1[1]) + (p2[1] + p3[1])) + 64.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                reinterpret_cast<f32x2*>(ep.ln.out_part)[(int64_t)(cm0 + tid) * (N >> 8) + (cn0 >> 8)] = (f32x2){mu, m2};
            }
        }
        if (!has_next) {
            // own group exhausted: look at the other groups only now, behind this item's stores (a launch's tail)
            if (!pl.ctr) break;
            __syncthreads();                                   // every wave has read the slot and left the staging area
            if (wave == 0) {
                const int got = pl.steal(lane, seen, slot);
                if (lane == 0) *slot = got;
            }
            __syncthreads();
            next = __builtin_amdgcn_readfirstlane(*slot);
            if (next < 0) break;
            pl.follow(next);
            pend2 = pl.claim(tid);
            item_coords(next, m0, n0, kt0, kt1);
            ra = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0);
            rb = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0);
            if (kt0 < kt1) {
                glds_tile256<A_KM, A_AUX>(ra, smem, lda * 2, kt0 * G_BK, wave, lane);
                glds_tile256<B_KM>(rb, smem + H_IMG, ldb * 2, kt0 * G_BK, wave, lane);
            }
        }
        run = true;
        if (pl.ctr && tid == 0) slot[2] = (int)pend2;       // parked before the next item's first barrier
        item = next;
    }
}

// =================================================================================================
// bf16 deep-pipeline path ("p4"): the 256x256 persistent kernel above is latency-bound -- only ONE 64 KiB stage is in
// flight per CU while L2->LDS needs ~100 KiB in flight to stream at its per-CU rate.  Same tile and wave layout, but
//   * 4 LDS stages of BK = 32 (A 16 KiB + B 16 KiB each), up to 3 stages in flight;
//   * per k-step: counted s_waitcnt vmcnt(8/4/0) (never a full drain in steady state) -> raw s_barrier -> issue the
//     stage 3 steps ahead into the buffer that was computed LAST step (free once every wave passed this barrier)
//     -> 32 MFMAs per wave from the current stage;
//   * the next item's first three stages are issued before the epilogue, which parks accumulators in stage 3.
// =================================================================================================
#define P_BK 32
#define P_IMG (256 * 32 * 2)      // 16 KiB per operand image
#define P_STAGE (2 * P_IMG)

// K-major image [256 rows][32 k] (64-byte rows): chunk c (0..3) of row r; the xor table makes the ds_read_b128 lane
// groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (rows 0-3,12-15 with chunk g, rows 4-11 with chunk g+1) conflict-free
__device__ __forceinline__ int p_sw(int r) {   // {0,2,3,1}[(r>>2)&3]
    const int q = (r >> 2) & 3;
    return (0x78 >> (2 * q)) & 3;              // 0b01_11_10_00
}
__device__ __forceinline__ int pk_off(int r, int c) { return r * 64 + ((c ^ p_sw(r)) << 4); }

// PIECES 1-KiB pieces of one image, issued by this wave: pieces [first, first+count)
template <bool KM>
__device__ __forceinline__ void p_glds(v4i32 rs, uint32_t img_lds, int ld_bytes, int k0, int first, int count, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= count) break;
        const int p = first + i;
        int voff;
        if (KM) {
            const int r = 16 * p + (lane >> 2);
            const int c = (lane & 3) ^ p_sw(r);
            voff = r * ld_bytes + (k0 + c * 8) * 2;
        } else {
            const int k = 2 * p + (lane >> 5);
            const int c = (lane & 31) ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1);
            voff = (k0 + k) * ld_bytes + c * 16;
        }
        dma16(rs, img_lds + p * 1024, voff);
    }
}
template <int ROWB> __device__ __forceinline__ int cslow_offw(int k, int c) { return k * ROWB + ((c ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1)) << 4); }
template <bool KM, int ROWB = 512>
__device__ __forceinline__ bf16x8 p_frag(const char* img, int t16, int lane) {
    if (KM) {
        return *reinterpret_cast<const bf16x8*>(img + pk_off(t16 * 16 + (lane & 15), lane >> 4));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int k = 8 * g + q;
        const int c = t16 * 2 + (p >> 1);
        const int sub = (p & 1) * 8;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + cslow_offw<ROWB>(k, c) + sub));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + cslow_offw<ROWB>(k + 4, c) + sub));
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f[j] = lo[j];
            f[4 + j] = hi[j];
        }
        return f;
    }
}

// ---- grouped weight gradients: ONE launch for several split-K problems that contract over the same K (the four Conv1D
// weight gradients of a decoder block contract over the tokens).  Every launch of a split-K problem pays for 256 partial
// 256x256 f32 tiles going through the memory-side float-atomic unit (64 MB at ~1.3 TB/s = ~50 us, MI355X_MICROARCH "Global float
// atomics": measured as the K-independent part of a wgrad launch, 232 us at K = 131072 against 97 us at K = 32768), whatever its
// depth.  Grouped, the block's 48 output tiles (C2) are cut stream-K fashion into one contiguous (tile, k-range) segment per
// workgroup -- a segment that crosses a tile boundary becomes two items -- so the block pays for ~300 partial tiles once instead of
// 4 x 256, every workgroup contracts the same number of k-steps, and the workgroups of an XCD are given segments of the same K
// octant (its L2 fetches each operand panel once).  The host builds the item table (gemm_wgrad_group_run).
#define WG_MAXP 8
struct WgProb { const bf16_t* A; const bf16_t* B; float* C; int M, N, lda, ldb, ldc, pad; };      // 48 bytes, in device memory
// item table entry: {problem, m0 | n0 << 16, first k-step, end k-step}; first == end: nothing to do

// NWM: wave rows (1: 128x256 tile, 4 waves, 2 workgroups per CU; 2: 256x256 tile, 8 waves, 1 per CU).  NST: LDS stages.
template <bool A_KM, bool B_KM, bool SWAP, int NWM, int NST, bool DIAG, int EPI, bool GROUPED>
__device__ __forceinline__ void p4_body(int M, int N, int K, const bf16_t* __restrict__ A, int lda,
                                        const bf16_t* __restrict__ B, int ldb, void* C, int ldc,
                                        const Epilogue& ep, int ksteps_per_split, int nsplit, int tiles_n,
                                        int ntiles, int64_t slab_stride,
                                        unsigned long long* stamps, const WgProb* __restrict__ gprobs, const int4* __restrict__ gitems) {
    // DIAG instantiation only (tools/gemm_timeline.py): workgroup 17, wave 0 records s_memtime at fixed points of its
    // first items into a buffer nothing else reads.  The product instantiation compiles the stamps away.
    constexpr int BM = 128 * NWM;
    constexpr int A_IMG = BM * P_BK * 2, B_IMG = 256 * P_BK * 2, STAGE = A_IMG + B_IMG;
    constexpr int A_ROWB = BM * 2;                    // row bytes of a contraction-slow A image
    constexpr int A_PIECES = A_IMG / 1024, NWAVES = 4 * NWM;
    constexpr int A_PER = A_PIECES / NWAVES, B_PER = 16 / NWAVES;          // pieces per wave: 2 and 4 (NWM=1) / 2 (NWM=2)
    constexpr int DPS = A_PER + B_PER;                // DMA instructions per wave per stage
    constexpr int AHEAD = NST - 1;                    // stages issued ahead at the top of an item
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NST stages x [A img | B img]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nitems = GROUPED ? ntiles : ntiles * nsplit;          // grouped: `ntiles` carries the length of the item table
    const int nk = cdiv(K, P_BK);

    int cur_split = 0;
    auto item_coords = [&](int item, int& m0, int& n0, int& kt0, int& kt1) {
        if constexpr (GROUPED) {
            const int4 d = gitems[item];                  // wave-uniform index: scalar loads
            const WgProb pr = gprobs[d.x];
            A = pr.A; B = pr.B; C = pr.C; M = pr.M; N = pr.N; lda = pr.lda; ldb = pr.ldb; ldc = pr.ldc;
            m0 = d.y & 0xFFFF;
            n0 = (int)((unsigned)d.y >> 16);
            kt0 = d.z;
            kt1 = d.w;
            return;
        }
        const int q = nitems >> 3, r = nitems & 7, x = item & 7;
        const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (item >> 3);
        // split-major: consecutive items (= one XCD's group) are the TILES of one K range, which share that range's A and B
        // panels through the XCD's L2.  (Tile-major -- the splits of one tile side by side -- gave an XCD sixteen disjoint K
        // ranges: no operand byte shared, 2.4x the algorithmic bytes at the L2's memory side in the wgrad launches,
        // profiles/r4_01_hbm_traffic.json.)
        const int sp = lin / ntiles, tile = lin - sp * ntiles;
        m0 = (tile / tiles_n) * BM;
        n0 = (tile % tiles_n) * H_BN;
        kt0 = sp * ksteps_per_split;
        kt1 = min(nk, kt0 + ksteps_per_split);
        cur_split = sp;
    };
    auto make_rsrc = [&](const bf16_t* P, bool km, int64_t rows, int ld, int t0) {
        const bf16_t* base = km ? P + (int64_t)t0 * ld : P + t0;
        const int64_t bytes = (km ? (rows - t0) * ld : rows * ld - t0) * 2;
        return make_srd(base, bytes);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;     // LDS byte address of the staging area

    int item = blockIdx.x;
    if (item >= nitems) return;
    ItemPuller pl;
    pl.init(ep, nitems, tid);
    int* slot = reinterpret_cast<int*>(smem + NST * STAGE);   // scheduler words of thread 0 (see ItemPuller)
    uint32_t fo, pend;
    pl.start(tid, fo, pend);   // OLDER than every DMA below: the counted vmcnt waits stay conservative
    int nstamp = 0;
    auto stamp = [&](int id) {
        if (DIAG && stamps && blockIdx.x == 17 && tid == 0 && nstamp < 250) {
            stamps[2 * nstamp] = (unsigned long long)id;
            stamps[2 * nstamp + 1] = __builtin_amdgcn_s_memtime();
            nstamp++;
        }
    };
    int m0, n0, kt0, kt1;
    item_coords(item, m0, n0, kt0, kt1);
    stamp(1);
    v4i32 ra = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0);
    v4i32 rb = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0);
    auto issue = [&](int kt, int buf) {
        // A image: K-major pieces are 16 rows x 64 B; contraction-slow pieces are (1024 / A_ROWB) k-rows
        if constexpr (A_KM) {
            p_glds<true>(ra, lds0 + buf * STAGE, lda * 2, kt * P_BK, wave * A_PER, A_PER, lane);
        } else {
#pragma unroll
            for (int i = 0; i < A_PER; i++) {
                const int p = wave * A_PER + i;
                constexpr int KPP = 1024 / A_ROWB;            // k-rows per piece: 4 (BM=128) / 2 (BM=256)
                constexpr int CPR = A_ROWB / 16;              // 16-byte chunks per row
                const int k = KPP * p + lane / CPR;
                const int c = (lane % CPR) ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1);
                dma16(ra, lds0 + buf * STAGE + p * 1024, (kt * P_BK + k) * lda * 2 + c * 16);
            }
        }
        p_glds<B_KM>(rb, lds0 + buf * STAGE + A_IMG, ldb * 2, kt * P_BK, wave * B_PER, B_PER, lane);
    };
    // one of this wave's DPS pieces of a stage (the main loop spreads them over the MFMA groups after the barrier: all
    // 32 pieces of a stage issued back to back right behind the barrier cost ~470 cycles of idle MFMA pipe per k-step --
    // 32 KiB through the 64 B/clk address path -- measured with the in-kernel stamps)
    auto issue_piece = [&](int kt, int buf, int j) {
        if (j < A_PER) {
            if constexpr (A_KM) {
                p_glds<true>(ra, lds0 + buf * STAGE, lda * 2, kt * P_BK, wave * A_PER + j, 1, lane);
            } else {
                const int p = wave * A_PER + j;
                constexpr int KPP = 1024 / A_ROWB;
                constexpr int CPR = A_ROWB / 16;
                const int k = KPP * p + lane / CPR;
                const int c = (lane % CPR) ^ (((k & 3) | (((k >> 3) & 1) << 2)) << 1);
                dma16(ra, lds0 + buf * STAGE + p * 1024, (kt * P_BK + k) * lda * 2 + c * 16);
            }
        } else {
            p_glds<B_KM>(rb, lds0 + buf * STAGE + A_IMG, ldb * 2, kt * P_BK, wave * B_PER + (j - A_PER), 1, lane);
        }
    };
#pragma unroll
    for (int i = 0; i < AHEAD; i++)
        if (kt0 + i < kt1) issue(kt0 + i, i);

    bool run = true;      // false: this workgroup's first item was taken by another one before it got to run
    if (pl.ctr) {
        if (tid == 0) { slot[1] = (int)fo; slot[2] = (int)pend; slot[3] = 0; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        run = __builtin_amdgcn_readfirstlane(slot[1]) == 0;
        if (!run) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the speculative stages have landed: their buffers may be refilled
    }
    if (GROUPED && kt1 <= kt0) run = false;                // an empty table entry: no k-loop, no epilogue
    while (true) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int n = run ? kt1 - kt0 : 0;
        // Software pipeline ACROSS the barrier: the wait+barrier that publishes stage t+1 sits inside stage t's 32
        // MFMAs and stage t+1's B fragments + first A fragment are read right after it, under the remaining MFMAs.
        bf16x8 fb[4], fa0;
        auto wait_younger = [&](int y) {        // y stages (DPS instructions each) may remain in flight
            if (y >= 2 && AHEAD >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
            else if (y >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if (n > 0) {
            stamp(2);
            wait_younger(min(n, AHEAD) - 1);
            stamp(3);
            __builtin_amdgcn_s_barrier();
            stamp(4);
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = p_frag<B_KM, 512>(smem + A_IMG, wn * 4 + j, lane);
            fa0 = p_frag<A_KM, A_ROWB>(smem, wm * 8, lane);
        }
        // Stagger (8-wave tile only): waves w and w+4 share a SIMD; waves 0-3 take the stage barrier after MFMA group
        // 3, waves 4-7 before group 0 (same barrier count) so the partners run half a stage apart.  With NWM == 1 the
        // SIMD partner belongs to the other, independent workgroup.
        auto run_stage = [&](int t, auto BAR) {
            constexpr int bar_at = decltype(BAR)::value;     // barrier after this MFMA group (-1: before group 0)
            const char* ia = smem + (t % NST) * STAGE;
            const char* nia = smem + ((t + 1) % NST) * STAGE;
            bf16x8 fa[8], fbn[4], fa0n;
            fa[0] = fa0;
            bool do_issue = false;
            auto sync_point = [&]() {
                // stage t+1 must have landed; stages t+2 .. t+AHEAD-1 may still fly.  Past this barrier every wave has
                // finished stage t-1, so its buffer takes stage t+AHEAD.
                stamp(10);
                wait_younger(min(n - t - 2, AHEAD - 2));
                stamp(11);
                __builtin_amdgcn_s_barrier();
                stamp(12);
#if !defined(COMPOSER_EXPERIMENTS) || !defined(P4_DIAG)      // measurement ladders exist in experiments builds only
#undef P4_DIAG
#define P4_DIAG 0
#endif
                do_issue = t + AHEAD < n && P4_DIAG != 3;       // P4_DIAG (measurement builds, wrong results): 1 no MFMA, 3 no DMA in the k-loop
                stamp(13);
                if (t + 1 < n) {
#pragma unroll
                    for (int j = 0; j < 4; j++) fbn[j] = p_frag<B_KM, 512>(nia + A_IMG, wn * 4 + j, lane);
                    fa0n = p_frag<A_KM, A_ROWB>(nia, wm * 8, lane);
                }
            };
            if constexpr (bar_at < 0) sync_point();
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i < 7) fa[i + 1] = p_frag<A_KM, A_ROWB>(ia, wm * 8 + i + 1, lane);
                if (i == bar_at) sync_point();
                {
                    // the stage-(t+AHEAD) pieces ride on the MFMA groups behind the barrier
                    constexpr int NG = 7 - bar_at > 4 ? 4 : 7 - bar_at;          // groups used: 4
                    constexpr int PPG = (DPS + NG - 1) / NG;
                    const int g = i - (bar_at + 1);
                    if (g >= 0 && g < NG && do_issue) {
#pragma unroll
                        for (int q = 0; q < PPG; q++)
                            if (g * PPG + q < DPS) issue_piece(kt0 + t + AHEAD, (t + AHEAD) % NST, g * PPG + q);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
#if P4_DIAG == 1
                    asm volatile("" ::"v"(fa[i]), "v"(fb[j]));
#else
                    if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
#endif
                }
            }
            if (t + 1 < n) {
#pragma unroll
                for (int j = 0; j < 4; j++) fb[j] = fbn[j];
                fa0 = fa0n;
            }
        };
        if (NWM == 1 || wave < 4) {
            for (int t = 0; t < n; t++) run_stage(t, std::integral_constant<int, 3>());
        } else {
            for (int t = 0; t < n; t++) run_stage(t, std::integral_constant<int, -1>());
        }
        stamp(20);
        if (pl.ctr && tid == 0) *slot = pl.claimed(slot);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave has finished reading the stages of this item
        stamp(21);
        const int cm0 = m0, cn0 = n0;
        void* const Ccur = C;                  // (grouped: item_coords() below moves C / M / N / ldc on to the NEXT item's problem)
        const int Mcur = M, Ncur = N, ldccur = ldc;
        // split-K with a slab workspace: every split writes its own fp32 partial tile with the ordinary full-line
        // epilogue (plain stores run ~4-5x the f32-atomic rate and the sum is reproducible); gemm_slab_reduce folds them
        void* Cit = slab_stride ? (void*)((float*)C + (int64_t)cur_split * slab_stride) : C;
        int next = pl.ctr ? __builtin_amdgcn_readfirstlane(*slot) : (int)(item + gridDim.x < nitems ? item + gridDim.x : -1);
        bool has_next = next >= 0;
        uint32_t pend2 = 0;
        if (has_next) {
            pend2 = pl.claim(tid);             // before the stage DMAs (older in the vmcnt queue)
            item_coords(next, m0, n0, kt0, kt1);
            ra = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0);
            rb = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0);
#pragma unroll
            for (int i = 0; i < AHEAD; i++)
                if (kt0 + i < kt1) issue(kt0 + i, i);              // the epilogue owns the B image of the last stage
        }
        stamp(22);
        uint32_t seen = 0;
        if (!has_next && pl.ctr && wave == 0) seen = pl.peek(lane);
        if (!run) {
        } else if (SWAP) {
            // per wave [16 rows][64 floats] = 4 KiB, 16-byte chunks xor-swizzled by the row
            float* stg = reinterpret_cast<float*>(smem + (NST - 1) * STAGE + A_IMG) + wave * (16 * 64);
            static_assert(NWAVES * 4096 <= B_IMG + (NWM == 2 ? A_IMG : 0), "epilogue staging must fit the last stage");
            if (NWM == 2) stg = reinterpret_cast<float*>(smem + (NST - 1) * STAGE) + wave * (16 * 64);
            const int col = cn0 + wn * 64 + (lane & 7) * 8;
            if constexpr (EPI != EPI_GENERIC) {
                EpiPre<EPI, 0, 1> pre;
                epi_prefetch<EPI, 0, 1, 2>(ep, cm0 + wm * 128, col, lane, pre);
                epi_tile<EPI, true, 0, 1, 2>(ep, (bf16_t*)Cit, ldc, cm0 + wm * 128, col, stg, lane, acc, pre);
            } else {
            const int rbase = cm0 + wm * 128 + (lane >> 3);
            float bias8[8];
            epi_bias8(ep, col, N, bias8);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                stamp(40 + i);
                EpiIn in[2];          // fetched before the staging round trip through LDS, used after it
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int row = rbase + i * 16 + it * 8;
                    in[it] = epi_fetch8(ep, row, col, row < M && col < N);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int r = lane & 15, c = j * 4 + (lane >> 4);
                    *reinterpret_cast<f32x4*>(stg + r * 64 + ((c ^ r) << 2)) = acc[i][j];
                }
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int rl = it * 8 + (lane >> 3);
                    const int row = rbase + i * 16 + it * 8;
                    const int c0 = (lane & 7) * 2;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + rl * 64 + ((c0 ^ rl) << 2));
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + rl * 64 + (((c0 + 1) ^ rl) << 2));
                    if (row < M && col < N) epi_finish8(ep, Cit, ldc, row, col, N, v0, v1, bias8, in[it]);
                }
            }
            }
        } else {
            float* Cf = (float*)Ccur;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int col = cn0 + wn * 64 + j * 16 + (lane & 15);
                    const int row0 = cm0 + wm * 128 + i * 16 + (lane >> 4) * 4;
                    if (col < Ncur) {
                        if (row0 + 0 < Mcur) atomicAdd(Cf + (int64_t)(row0 + 0) * ldccur + col, acc[i][j][0]);
                        if (row0 + 1 < Mcur) atomicAdd(Cf + (int64_t)(row0 + 1) * ldccur + col, acc[i][j][1]);
                        if (row0 + 2 < Mcur) atomicAdd(Cf + (int64_t)(row0 + 2) * ldccur + col, acc[i][j][2]);
                        if (row0 + 3 < Mcur) atomicAdd(Cf + (int64_t)(row0 + 3) * ldccur + col, acc[i][j][3]);
                    }
                }
        }
        stamp(30);
        if (!has_next) {
            // own group exhausted: look at the other groups only now, behind this item's stores (a launch's tail)
            if (!pl.ctr) break;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wave == 0) {
                const int got = pl.steal(lane, seen, slot);
                if (lane == 0) *slot = got;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            next = __builtin_amdgcn_readfirstlane(*slot);
            if (next < 0) break;
            pl.follow(next);
            pend2 = pl.claim(tid);
            item_coords(next, m0, n0, kt0, kt1);
            ra = make_rsrc(A, A_KM, A_KM ? M : K, lda, m0);
            rb = make_rsrc(B, B_KM, B_KM ? N : K, ldb, n0);
#pragma unroll
            for (int i = 0; i < AHEAD; i++)
                if (kt0 + i < kt1) issue(kt0 + i, i);
        }
        run = !(GROUPED && kt1 <= kt0);
        if (pl.ctr && tid == 0) slot[2] = (int)pend2;       // parked before the next item's k-loop
        item = next;
        // the epilogue's stores/atomics sit in the same vmcnt queue behind the prefetched stages: drain them so the
        // counted waits of the next item see only its own DMA (the other workgroup on the CU keeps the pipes busy)
        // (leaving the stores in flight and allowing for them in the first counted waits measured 2-5 % SLOWER in a
        // same-box A/B: the epilogue is ~4.5k of a ~55k-cycle item and the drain ~1k)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(31);
    }
}

template <bool A_KM, bool B_KM, bool SWAP, int NWM, int NST, bool DIAG = false, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(256 * NWM, 2) void gemm_bf16_p4_kernel(int M, int N, int K, const bf16_t* __restrict__ A, int lda,
                                                                    const bf16_t* __restrict__ B, int ldb, void* C, int ldc,
                                                                    Epilogue ep, int ksteps_per_split, int nsplit, int tiles_n,
                                                                    int ntiles, int64_t slab_stride,
                                                                    unsigned long long* stamps) {
    p4_body<A_KM, B_KM, SWAP, NWM, NST, DIAG, EPI, false>(M, N, K, A, lda, B, ldb, C, ldc, ep, ksteps_per_split, nsplit, tiles_n, ntiles,
                                                          slab_stride, stamps, nullptr, nullptr);
}
// the grouped launch: operands stored [K, M_p] / [K, N_p] (contraction over rows), f32 atomics into C_p; nitems table entries
__global__ __launch_bounds__(512, 2) void gemm_wgrad_group_kernel(int K, const WgProb* __restrict__ gprobs, const int4* __restrict__ gitems,
                                                                  int nitems, Epilogue ep) {
    p4_body<false, false, false, 2, 4, false, EPI_GENERIC, true>(0, 0, K, nullptr, 0, nullptr, 0, nullptr, 0, ep, 0, 1, 1, nitems, (int64_t)0,
                                                                 nullptr, gprobs, gitems);
}

// dynamic-LDS opt-in, once per kernel
static void allow_smem(const void* kern, size_t smem) {
    static std::mutex mu;
    static std::vector<const void*> done;
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(done.begin(), done.end(), kern) != done.end()) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    done.push_back(kern);
}
static unsigned long long* g_gemm_stamps = nullptr;    // diagnostic only (cmp_gemm_set_stamps)
extern "C" int cmp_gemm_set_stamps(void* dev_buf) {
    g_gemm_stamps = (unsigned long long*)dev_buf;
    return CMP_OK;
}
template <bool A_KM, bool B_KM, int NWM, int NST>
static bool launch_p4_cfg(hipStream_t s, bool swap, int M, int N, int K, const bf16_t* a, int lda, const bf16_t* b, int ldb,
                          void* C, int ldc, const Epilogue& ep, int per, int nsplit, int64_t slab_stride, int max_wgs) {
    constexpr int BM = 128 * NWM;
    const size_t smem = (size_t)NST * (BM * P_BK * 2 + 256 * P_BK * 2) + 16;      // + the item slot
    {
        static std::once_flag attr_once;         // (per instantiation; two threads may launch through one library)
        std::call_once(attr_once, [smem]() {
            (void)hipFuncSetAttribute((const void*)gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            (void)hipFuncSetAttribute((const void*)gemm_bf16_p4_kernel<A_KM, B_KM, false, NWM, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        });
    }
    const int tiles_n = cdiv(N, H_BN), ntiles = tiles_n * cdiv(M, BM);
    const int grid = std::min(ntiles * nsplit, NWM == 1 ? 2 * max_wgs : max_wgs);   // persistent: one (two) workgroups per CU in use
    if constexpr (A_KM && !B_KM) {
        // the forward layout carries the compile-time epilogue kinds (and the diagnostic timeline build)
        const int kind = epi_kind_of(ep, M, N, swap, slab_stride != 0);
        auto go = [&](auto kern) {
            allow_smem((const void*)kern, smem);
            kern<<<grid, 256 * NWM, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles, slab_stride, g_gemm_stamps);
        };
        if (g_gemm_stamps && swap) {
            if constexpr (NWM == 2) {
                if (kind == EPI_PLAIN) go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, true, EPI_PLAIN>);
                else if (kind == EPI_GELU_AUX) go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, true, EPI_GELU_AUX>);
                else if (kind == EPI_RESID) go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, true, EPI_RESID>);
                else go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, true, EPI_GENERIC>);
                return kind == EPI_PLAIN || kind == EPI_GELU_AUX || kind == EPI_RESID;
            }
        }
        if (kind == EPI_PLAIN) { go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, false, EPI_PLAIN>); return true; }
        if (kind == EPI_GELU_AUX) { go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, false, EPI_GELU_AUX>); return true; }
        if (kind == EPI_RESID) { go(gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST, false, EPI_RESID>); return true; }
    }
    if constexpr (!A_KM && !B_KM && NWM == 2) {
        if (g_gemm_stamps && !swap) {        // diagnostic timeline of the wgrad layout (split-K atomics epilogue)
            allow_smem((const void*)gemm_bf16_p4_kernel<A_KM, B_KM, false, NWM, NST, true>, smem);
            gemm_bf16_p4_kernel<A_KM, B_KM, false, NWM, NST, true><<<grid, 256 * NWM, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles, (int64_t)0, g_gemm_stamps);
            return false;
        }
    }
    if (swap)
        gemm_bf16_p4_kernel<A_KM, B_KM, true, NWM, NST><<<grid, 256 * NWM, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles, slab_stride, g_gemm_stamps);
    else
        gemm_bf16_p4_kernel<A_KM, B_KM, false, NWM, NST><<<grid, 256 * NWM, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles, (int64_t)0, g_gemm_stamps);
    return false;
}
template <bool A_KM, bool B_KM>
static bool launch_p4(hipStream_t s, int cfg, bool swap, int M, int N, int K, const bf16_t* a, int lda, const bf16_t* b, int ldb,
                      void* C, int ldc, const Epilogue& ep, int per, int nsplit, int64_t slab_stride, int max_wgs) {
    if (cfg == 1) return launch_p4_cfg<A_KM, B_KM, 1, 3>(s, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, slab_stride, max_wgs);
    return launch_p4_cfg<A_KM, B_KM, 2, 4>(s, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, slab_stride, max_wgs);
}

// C[i] += sum_s slab[s][i]  (fixed order: reproducible); 16 bytes per lane
__global__ void gemm_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, int64_t n4, int64_t stride4, int nsplit) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = reinterpret_cast<f32x4*>(C)[i];
        for (int sp = 0; sp < nsplit; sp++) a += reinterpret_cast<const f32x4*>(slab)[sp * stride4 + i];
        reinterpret_cast<f32x4*>(C)[i] = a;
    }
}
// One-shot: the NEXT cmp_k_gemm also adds the column sums of its output to out[0..N) (bias gradient of the layer that
// produced the GEMM's input gradient).  Fused into the epilogue where a compile-time kind runs (saves re-reading the
// output: 84 us per layer for the [M,4E] MLP gradient at B=128), otherwise a cmp_k_colsum pass after the launch.
// The model driver (model.hip) passes these per launch in a GemmExtra (model.h) -- no state outlives a call.  The two
// C-ABI arming calls below exist for the kernel-level tests / micro-benchmarks that drive cmp_k_gemm directly; their state is
// per calling thread and consumed by that thread's next cmp_k_gemm.
static thread_local float* t_colsum_next = nullptr;
extern "C" int cmp_gemm_colsum_next(float* out) {
    t_colsum_next = out;
    return CMP_OK;
}
// One-shot, like cmp_gemm_colsum_next: the NEXT cmp_k_gemm of this thread carries a LayerNorm epilogue (common.h: LnEpi; the kernel-
// level parity tests of the fused block path).  in_part / out_part: partial row statistics [rows][segments][2]; cs: fold; gamma, beta:
// residual rebuild.  A launch that cannot carry it fails with CMP_ERR_INVALID.
static thread_local LnEpi t_ln_next;
extern "C" int cmp_gemm_ln_next(const float* in_part, int np, float eps, const float* cs, const float* gamma, const float* beta,
                                float* out_part) {
    t_ln_next = LnEpi{in_part, np, eps, cs, gamma, beta, out_part};
    return CMP_OK;
}
static thread_local float* t_slab_ws = nullptr;       // split-K workspace for this thread's cmp_k_gemm calls
static thread_local size_t t_slab_bytes = 0;
extern "C" int cmp_gemm_set_workspace(void* ws, int64_t bytes) {
    t_slab_ws = (float*)ws;
    t_slab_bytes = ws ? (size_t)bytes : 0;
    return CMP_OK;
}

template <bool A_KM, bool B_KM>
static bool launch_256(hipStream_t s, int grid, bool swap, int M, int N, int K, const bf16_t* a, int lda, const bf16_t* b, int ldb,
                       void* C, int ldc, const Epilogue& ep, int per, int nsplit, int tiles_n, int ntiles) {
    const size_t smem = 4 * H_IMG + 32;                                             // + the scheduler words
    {
        static std::once_flag attr_once;
        std::call_once(attr_once, [smem]() {
            (void)hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<A_KM, B_KM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            (void)hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<A_KM, B_KM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        });
    }
    if constexpr (A_KM && B_KM) {        // the dgrad layout carries the compile-time epilogue kinds
        const int kind = epi_kind_of(ep, M, N, swap, false);
        auto go = [&](auto kern) {
            allow_smem((const void*)kern, smem);
            kern<<<grid, 512, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
        };
        const int lnm = (ep.ln.in_part ? 1 : 0) | (ep.ln.out_part ? 2 : 0);
        if (lnm) {
            // LayerNorm-fused kinds (gemm_run has checked that the launch fits one): fold into c_attn / c_fc, statistics out of
            // (and the rebuilt residual into) both c_proj
#define LN_GO(KIND, LNM_, NP_) do { go(gemm_bf16_256_kernel<A_KM, B_KM, true, KIND, LNM_, NP_>); return true; } while (0)
#define LN_NP(KIND, LNM_) do { if (ep.ln.np == 2) LN_GO(KIND, LNM_, 2); if (ep.ln.np == 3) LN_GO(KIND, LNM_, 3); } while (0)
            if (ep.out_fp32 && lnm == 1) LN_NP(EPI_PLAIN32, 1);
            if (kind == EPI_PLAIN && lnm == 1) LN_NP(EPI_PLAIN, 1);
            if (kind == EPI_GELU_AUX && lnm == 1) LN_NP(EPI_GELU_AUX, 1);
            if (kind == EPI_RESID && lnm == 3) LN_NP(EPI_RESID, 3);
            if (kind == EPI_RESID && lnm == 2) LN_GO(EPI_RESID, 2, 1);
#undef LN_NP
#undef LN_GO
            return false;       // unreachable after gemm_run's check
        }
        if (kind == EPI_PLAIN) { go(gemm_bf16_256_kernel<A_KM, B_KM, true, EPI_PLAIN>); return true; }
        if (kind == EPI_RESID) { go(gemm_bf16_256_kernel<A_KM, B_KM, true, EPI_RESID>); return true; }
        if (kind == EPI_GELUGRAD) { go(gemm_bf16_256_kernel<A_KM, B_KM, true, EPI_GELUGRAD>); return true; }
        if (kind == EPI_GELU_AUX) { go(gemm_bf16_256_kernel<A_KM, B_KM, true, EPI_GELU_AUX>); return true; }
    }
    if (swap)
        gemm_bf16_256_kernel<A_KM, B_KM, true><<<grid, 512, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
    else
        gemm_bf16_256_kernel<A_KM, B_KM, false><<<grid, 512, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
    return false;
}

template <bool A_KM, bool B_KM>
static bool launch_fast(hipStream_t s, dim3 grid, size_t smem, bool swap, int M, int N, int K, const bf16_t* a, int lda,
                        const bf16_t* b, int ldb, void* C, int ldc, const Epilogue& ep, int per, int tiles_n, int ntiles) {
    if constexpr (A_KM && B_KM) {        // forward (transposed weight shadow) and dgrad layout
        static const bool kinds_on = [] { const char* e = getenv("COMPOSER_GEMM_FAST_KINDS"); return !(e && e[0] == '0'); }();
        const int kind = (kinds_on && grid.y == 1) ? epi_kind_of(ep, M, N, swap, false, 128) : EPI_GENERIC;
        auto go = [&](auto kern) { kern<<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles); };
        // at most one workgroup per CU and whole k-steps: the four-stage ring (COMPOSER_GEMM_RING=0 off, =<n> from n k-steps on)
        static const int ring_min = [] { const char* e = getenv("COMPOSER_GEMM_RING"); return e ? atoi(e) : 2; }();       // fewest k-steps that take it
        if (kind != EPI_GENERIC && ring_min > 0 && ntiles <= 256 && K % G_BK == 0 && K >= std::max(2, ring_min) * G_BK) {
            constexpr int smem_ring = 4 * 2 * G_IMG;
            static const bool attr = [] {
                bool ok = true;
                auto set = [&](auto kern) { ok = ok && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem_ring) == hipSuccess; };
                set(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_PLAIN, true>);
                set(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_RESID, true>);
                set(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_GELUGRAD, true>);
                set(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_GELU_AUX, true>);
                return ok;
            }();
            if (attr) {
                auto gor = [&](auto kern) { kern<<<grid, 256, smem_ring, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles); };
                if (kind == EPI_PLAIN) { gor(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_PLAIN, true>); return true; }
                if (kind == EPI_RESID) { gor(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_RESID, true>); return true; }
                if (kind == EPI_GELUGRAD) { gor(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_GELUGRAD, true>); return true; }
                if (kind == EPI_GELU_AUX) { gor(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_GELU_AUX, true>); return true; }
            }
        }
        if (kind == EPI_PLAIN) { go(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_PLAIN>); return true; }
        if (kind == EPI_RESID) { go(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_RESID>); return true; }
        if (kind == EPI_GELUGRAD) { go(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_GELUGRAD>); return true; }
        if (kind == EPI_GELU_AUX) { go(gemm_bf16_fast_kernel<A_KM, B_KM, true, EPI_GELU_AUX>); return true; }
    }
    if (swap)
        gemm_bf16_fast_kernel<A_KM, B_KM, true><<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles);
    else
        gemm_bf16_fast_kernel<A_KM, B_KM, false><<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles);
    return false;
}

// =================================================================================================
// host launcher
// =================================================================================================
// Item-counter workspace of the persistent kernels (ItemPuller): per stream, two sets of eight counters, one 128-byte line
// each; `started` and one first-item flag per workgroup; consecutive launches on a stream alternate sets and every launch zeroes the other one.
#include <mutex>
#include <map>
#define CHECK_SCHED(expr) do { int rc_ = (expr); if (rc_ != CMP_OK) return rc_; } while (0)
// `held` keeps the process-wide table's lock until the launch has been issued (two threads driving one stream through
// cmp_k_gemm must launch in the order of their parity draws); a context's own workspace (ex.sched) needs no lock.
static int sched_next(hipStream_t s, Epilogue& ep, const GemmExtra& ex, std::unique_lock<std::mutex>& held, SchedWs** used) {
    // default: counters when the launch belongs to a data-parallel job (a communicator exists: RCCL kernels may hold CUs), static
    // striding otherwise (undisturbed, the counters cost 0-1.3 % -- an extra barrier and LDS word per item);
    // COMPOSER_GEMM_ITEMS=dynamic / static (read per call) overrides
    const char* e = getenv("COMPOSER_GEMM_ITEMS");
    const bool dyn = e ? e[0] == 'd' : ex.dp;
    ep.sched = ep.sched_clear = nullptr;
    *used = nullptr;
    if (!dyn) return CMP_OK;
    // inside a stream capture the launch is replayed with frozen arguments: the alternation of the two counter sets would not
    // survive a replay, so captured launches stride statically
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return CMP_OK;
    SchedWs* w = ex.sched;
    if (!w) {
        static std::mutex mu;
        static std::map<std::pair<int, hipStream_t>, SchedWs> tab;
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        held = std::unique_lock<std::mutex>(mu);
        w = &tab[std::make_pair(dev, s)];
    }
    if (!w->dev) {
        HIP_CHECK(hipMalloc((void**)&w->dev, 2 * SCHED_SET_WORDS * 4));
        w->dirty = true;
    }
    if (w->dirty) {     // first use, or the previous launch on these counters failed before it could zero the other set
        HIP_CHECK(hipMemsetAsync(w->dev, 0, 2 * SCHED_SET_WORDS * 4, s));      // on the launch's own stream: no legacy-stream call (another thread may be capturing)
        w->parity = 0;
        w->dirty = false;
    }
    ep.sched = w->dev + w->parity * SCHED_SET_WORDS;
    ep.sched_clear = w->dev + (w->parity ^ 1) * SCHED_SET_WORDS;
    w->parity ^= 1;
    *used = w;
    return CMP_OK;
}
void sched_ws_free(SchedWs* w) {
    if (w && w->dev) { (void)hipFree(w->dev); w->dev = nullptr; }
}

// ---- grouped weight gradients: the item table ---------------------------------------------------------------------------
// All T output tiles (of all problems) are cut into the SAME s = floor(G / T) K ranges; item lin = range * T + tile, dealt to the
// XCD groups in consecutive runs like every persistent launch here: the workgroups of an XCD contract the same rows of
// different tiles in step, so its L2 fetches each operand panel once.  (A stream-K cut -- one contiguous range of exactly
// T * nk / G k-steps per workgroup -- was built first and measured: every workgroup equally long, but the ranges of an XCD start
// at unrelated rows, nothing is shared through L2, and the launch took 890 us at K = 131072 where this form is predicted at
// ~800 and four split-K launches take 902-954; at C4's 108 tiles it lost 3.6 % of the step.)
// Cost model in k-steps of 32 (tools/kbench.py wgradgroup: 0.91 us per k-step, 54 us = 59 k-steps of atomic epilogue per launch):
// grouped = ceil(nk / s) * rounds + 59 against the sum over problems of nk / split_p + 59 -- the grouped launch is used when it wins.
// (round-6 entry points the model driver links against; the lab copy keeps the float-atomic grouped launch only)
void wgrad_ws_free(WgradWs* w) {
    if (w && w->ptr) { (void)hipFree(w->ptr); w->ptr = nullptr; w->bytes = 0; }
}
void wgrad_group_free(WgradGroup* g) {
    if (g && g->dev) { (void)hipFree(g->dev); g->dev = nullptr; g->dev_bytes = 0; g->key.clear(); }
}
int wgrad_group_run(void* stream, WgradGroup* g, const WgradProblem* probs, int nprob, int K, const GemmExtra& ex, bool* handled) {
    *handled = false;
    hipStream_t s = (hipStream_t)stream;
    const int G = (ex.max_wgs > 0 ? std::min(ex.max_wgs, 256) : 256) & ~7;
    if (nprob < 1 || nprob > WG_MAXP || K < P_BK || K % P_BK != 0 || G < 8 || ex.slab_ws) return CMP_OK;
    for (int i = 0; i < nprob; i++) {
        const WgradProblem& q = probs[i];
        if (q.M <= 0 || q.N <= 0 || q.M >= 65536 - 256 || q.N >= 65536 - 256 || q.lda % 8 || q.ldb % 8 || ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) ||
            (int64_t)K * q.lda * 2 >= 0x7FFFFFF0ll || (int64_t)K * q.ldb * 2 >= 0x7FFFFFF0ll)
            return CMP_OK;
    }
    // the key is built field by field: WgradProblem has padding holes whose bytes are whatever the caller's stack held (a raw
    // byte copy made every step look like new shapes: a rebuild, two stream syncs and an upload per block per step)
    std::string key;
    key.reserve((size_t)nprob * 48 + 8);
    auto put = [&key](const void* p, size_t n) { key.append((const char*)p, n); };
    for (int i = 0; i < nprob; i++) {
        const WgradProblem& q = probs[i];
        put(&q.A, sizeof(q.A)); put(&q.B, sizeof(q.B)); put(&q.C, sizeof(q.C));
        const int v[5] = {q.lda, q.ldb, q.ldc, q.M, q.N};
        put(v, sizeof(v));
    }
    put(&K, sizeof(K));
    put(&G, sizeof(G));
    if (key != g->key) {
        struct Seg { int prob, m0, n0; };
        std::vector<Seg> tiles;
        for (int i = 0; i < nprob; i++)
            for (int tm = 0; tm < cdiv(probs[i].M, 256); tm++)
                for (int tn = 0; tn < cdiv(probs[i].N, 256); tn++) tiles.push_back({i, tm * 256, tn * 256});
        const int T = (int)tiles.size(), nk = K / P_BK;
        // Split count: every split of every tile is one more partial 256x256 f32 tile through the float-atomic unit (~1.3 TB/s),
        // every split less is a longer k-loop per workgroup.  With few tokens the atomics win: the reference's default configuration
        // (1 024 tokens, 12 tiles) ran 21 splits = 64 MB of atomics for 32 k-steps of work, 44 us per block; the minimum of
        //   ceil(nk / s) * 0.91 us + T * s * 256 KiB / 1.3 TB/s          (tools/kbench.py wgradgroup, same constants as below)
        // over s <= G / T is 4 splits there (~17 us) and stays at G / T for the benchmark shapes (K = 32 768 ... 131 072 tokens).
        int sp = 1;
        {
            const int smax = std::max(1, std::min(G / T, nk));
            double best = 1e30;
            for (int c = 1; c <= smax; c++) {
                const double t = (double)cdiv(nk, c) * 0.91 + (double)T * c * 262144.0 / 1.3e6;
                if (t < best * 0.999) { best = t; sp = c; }
            }
        }
        // Workgroups left over by the common split count (C2: 256 - 48 * 5 = 16) go to whole PROBLEMS, smallest first, as one
        // more split each (tiles that share operand panels keep the same K ranges): their items are shorter, finish early and
        // put their share of the atomic traffic out before the final burst; the longest item is unchanged.
        std::vector<int> tiles_of(nprob, 0), split_of(nprob, sp);
        for (const Seg& t : tiles) tiles_of[t.prob]++;
        {
            int spare = G - T * sp;
            std::vector<int> byt(nprob);
            for (int i = 0; i < nprob; i++) byt[i] = i;
            std::stable_sort(byt.begin(), byt.end(), [&](int a, int b) { return tiles_of[a] < tiles_of[b]; });
            for (int i : byt)
                if (T * sp <= G && sp + 1 <= nk && tiles_of[i] <= spare) { split_of[i] = sp + 1; spare -= tiles_of[i]; }
        }
        // items in K-range-major order (range j of every problem that has one, tile by tile): consecutive items = one XCD's group
        struct Item { int tile, kt0, kt1; };
        std::vector<Item> order;
        int max_split = 0, longest = 0;
        for (int i = 0; i < nprob; i++) max_split = std::max(max_split, split_of[i]);
        for (int j = 0; j < max_split; j++)
            for (int t = 0; t < T; t++) {
                const int spt = split_of[tiles[t].prob], chunk = cdiv(nk, spt);
                if (j * chunk >= nk) continue;
                order.push_back({t, j * chunk, std::min(nk, (j + 1) * chunk)});
                longest = std::max(longest, std::min(nk, (j + 1) * chunk) - j * chunk);
            }
        const int nitems = (int)order.size();
        {   // is one grouped launch cheaper than one split-K launch per problem?  (model above; K-independent overhead in k-steps)
            const double ov = 59.0;
            const double grouped = (double)longest * cdiv(nitems, G) + ov;
            double separate = 0.0;
            for (int i = 0; i < nprob; i++) {
                const int t = cdiv(probs[i].M, 256) * cdiv(probs[i].N, 256);
                const int si = std::max(1, std::min(G / std::max(1, t), nk));
                separate += (double)cdiv(nk, si) * cdiv(t * si, G) + ov;
            }
            if (grouped >= separate && !g->force) {      // remembered: the next call with these shapes returns at once
                g->key = key;
                g->nitems = 0;
                return CMP_OK;
            }
        }
        std::string host(WG_MAXP * sizeof(WgProb) + (size_t)nitems * sizeof(int4), '\0');
        WgProb* hp = (WgProb*)&host[0];
        for (int i = 0; i < nprob; i++)
            hp[i] = WgProb{(const bf16_t*)probs[i].A, (const bf16_t*)probs[i].B, probs[i].C, probs[i].M, probs[i].N, probs[i].lda, probs[i].ldb,
                           probs[i].ldc, 0};
        int4* hi = (int4*)&host[WG_MAXP * sizeof(WgProb)];
        const int q = nitems >> 3, r = nitems & 7;
        for (int item = 0; item < nitems; item++) {
            const int x = item & 7;                                        // the XCD group of the item (ItemPuller): consecutive lin per group
            const int lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (item >> 3);
            const Item& it = order[lin];
            hi[item] = make_int4(tiles[it.tile].prob, tiles[it.tile].m0 | (tiles[it.tile].n0 << 16), it.kt0, it.kt1);
        }
        HIP_CHECK(hipStreamSynchronize(s));            // the previous table of this group may still be in use / in flight (a rebuild is rare)
        if (g->dev_bytes < host.size()) {
            wgrad_group_free(g);
            HIP_CHECK(hipMalloc(&g->dev, host.size()));
            g->dev_bytes = host.size();
        }
        g->host.swap(host);
        HIP_CHECK(hipMemcpyAsync(g->dev, g->host.data(), g->host.size(), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
        g->key = key;
        g->nitems = nitems;
        g->grid = std::min(nitems, G);
        g->rebuilds += 1;
    }
    if (g->nitems == 0) return CMP_OK;                 // the cost model chose one launch per problem for these shapes
    Epilogue ep;
    memset(&ep, 0, sizeof(ep));
    ep.atomic = 1;
    ep.out_fp32 = 1;
    ep.drop = make_drop(0.f, 0, 0);
    std::unique_lock<std::mutex> sched_lock;
    SchedWs* sched_used = nullptr;
    CHECK_SCHED(sched_next(s, ep, ex, sched_lock, &sched_used));
    const size_t smem = (size_t)4 * (256 * P_BK * 2 + 256 * P_BK * 2) + 16;
    {
        static std::once_flag attr_once;
        std::call_once(attr_once, [smem]() {
            (void)hipFuncSetAttribute((const void*)gemm_wgrad_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        });
    }
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < nprob; i++) {
        flops += 2.0 * probs[i].M * probs[i].N * K;
        bytes += 2.0 * ((double)probs[i].M * K + (double)K * probs[i].N) + 4.0 * probs[i].M * probs[i].N;
    }
    PROF_START(2, s);
    gemm_wgrad_group_kernel<<<g->grid, 512, smem, s>>>(K, (const WgProb*)g->dev, (const int4*)((const char*)g->dev + WG_MAXP * sizeof(WgProb)),
                                                        g->nitems, ep);
    PROF_STOP(2, s, flops, bytes);
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            if (sched_used) sched_used->dirty = true;
            cmp_set_error("%s:%d: grouped wgrad launch failed: %s", __FILE__, __LINE__, hipGetErrorString(le));
            return CMP_ERR_HIP;
        }
    }
    *handled = true;
    return CMP_OK;
}
// kernel-level entry point of the grouped launch (tests, micro-benchmarks): nprob problems given as parallel arrays
extern "C" int cmp_k_wgrad_group(void* stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                 float* const* C, const int* ldc, const int* M, const int* N, int K) {
    CMP_REQUIRE(nprob >= 1 && nprob <= WG_MAXP && A && B && C && lda && ldb && ldc && M && N, "wgrad_group: bad arguments");
    static thread_local WgradGroup grp;           // rebuilt whenever the problems change
    grp.force = true;                             // this entry point IS the grouped launch, whatever the cost model says
    WgradProblem pr[WG_MAXP];
    for (int i = 0; i < nprob; i++) pr[i] = WgradProblem{A[i], lda[i], B[i], ldb[i], C[i], ldc[i], M[i], N[i]};
    GemmExtra ex;
    bool handled = false;
    CHECK_SCHED(wgrad_group_run(stream, &grp, pr, nprob, K, ex, &handled));
    CMP_REQUIRE(handled, "wgrad_group: these shapes do not fit the grouped kernel (K %% 32, leading dimensions %% 8, 16-byte alignment, < 2 GiB operands)");
    return CMP_OK;
}

extern "C" int cmp_k_gemm(void* stream, int dtype, int ta, int tb, int M, int N, int K, const void* A, int lda,
                          const void* Bm, int ldb, void* C, int ldc, const float* bias, int act, void* aux, int ldaux,
                          const void* resid, int ldr, int out_fp32, int splitk, float p_drop, uint64_t seed,
                          uint32_t rng_stream, int flags) {
    GemmExtra ex;
    ex.colsum = t_colsum_next;
    t_colsum_next = nullptr;
    ex.ln = t_ln_next;
    t_ln_next = LnEpi{};
    ex.slab_ws = t_slab_ws;
    ex.slab_bytes = t_slab_bytes;
    return gemm_run(stream, dtype, ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, bias, act, aux, ldaux, resid, ldr, out_fp32, splitk,
                    p_drop, seed, rng_stream, flags, ex);
}

int gemm_run(void* stream, int dtype, int ta, int tb, int M, int N, int K, const void* A, int lda, const void* Bm, int ldb,
             void* C, int ldc, const float* bias, int act, void* aux, int ldaux, const void* resid, int ldr, int out_fp32,
             int splitk, float p_drop, uint64_t seed, uint32_t rng_stream, int flags, const GemmExtra& ex) {
    if (M == 0 || N == 0) return CMP_OK;
    CMP_REQUIRE(K > 0, "gemm: K must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int max_wgs = ex.max_wgs > 0 ? std::min(ex.max_wgs, 256) : 256;
    float* const g_slab_ws = ex.slab_ws;
    const size_t g_slab_bytes = ex.slab_ws ? ex.slab_bytes : 0;
    const int g_gemm_role = ex.role;
    std::unique_lock<std::mutex> sched_lock;         // held from the counter-set draw to the launch (process-wide table only)
    SchedWs* sched_used = nullptr;
    Epilogue ep;
    ep.bias = bias;
    ep.act = act;
    ep.aux = aux;
    ep.ldaux = ldaux;
    ep.resid = resid;
    ep.ldr = ldr;
    ep.out_fp32 = out_fp32;
    ep.atomic = splitk > 1 ? 1 : 0;
    ep.dbg_nostore = (flags & 64) ? 1 : 0;
    ep.colsum = nullptr;
    ep.sched = ep.sched_clear = nullptr;
    ep.ln = ex.ln;
    ep.rev = ex.rev ? 1 : 0;
    ep.n_cols = N;

    const int lnm = (ex.ln.in_part ? 1 : 0) | (ex.ln.out_part ? 2 : 0);
    bool ln_done = lnm == 0;         // a launch that asks for a LayerNorm epilogue must reach a kernel that has one
    float* colsum_out = ex.colsum;
    if (colsum_out) CMP_REQUIRE(!out_fp32 && splitk <= 1, "gemm: column sums need a plain (non split-K) output in the compute dtype");
    bool colsum_fused = false;
    ep.drop = make_drop(p_drop, seed, rng_stream);
    if (splitk > 1)
        CMP_REQUIRE(out_fp32 && !bias && act == 0 && !resid && p_drop == 0.f,
                    "gemm: split-K needs a plain fp32 accumulate epilogue");
    CMP_REQUIRE(act == 0 || aux != nullptr || act == 1, "gemm: act=2 needs aux");
    if (dtype == CMP_FP32) {
        int nk = cdiv(K, F_BK);
        splitk = std::max(1, std::min(splitk, nk));
        int per = cdiv(nk, splitk);
        dim3 grid(cdiv(N, F_BN), cdiv(M, F_BM), cdiv(nk, per));
        int64_t sam = ta ? 1 : lda, sak = ta ? lda : 1;
        int64_t sbk = tb ? 1 : ldb, sbn = tb ? ldb : 1;
        gemm_f32_kernel<<<grid, 256, 0, s>>>(M, N, K, (const float*)A, sam, sak, (const float*)Bm, sbk, sbn, C, ldc, ep, per);
    } else {
        CMP_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm(bf16): leading dimensions must be multiples of 8 (lda=%d ldb=%d)", lda, ldb);
        CMP_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)Bm & 15) == 0, "gemm(bf16): operands must be 16-byte aligned");
        int nk = cdiv(K, G_BK);
        const int splitk_req = splitk;
        splitk = std::max(1, std::min(splitk, nk));
        int per = cdiv(nk, splitk);
        dim3 grid(cdiv(N, G_BN), cdiv(M, G_BM), cdiv(nk, per));
        size_t smem = 4 * G_IMG;
        const bf16_t* a = (const bf16_t*)A;
        const bf16_t* b = (const bf16_t*)Bm;
        // timing class of cmp_prof_*: by role when the caller announced one (the model: 0 forward, 1 dgrad, 2 wgrad),
        // otherwise by layout (the forward GEMMs read a transposed weight copy, i.e. the dgrad layout)
        const int cls = g_gemm_role >= 0 ? g_gemm_role : (ta ? 2 : (tb ? 1 : 0));
        // fast path: direct-to-LDS staging needs every 64-deep k-step of a K-contiguous operand inside its row
        // (K % 64 == 0, or the caller vouches for zero padding up to a multiple of 64 with CMP_GEMM_KPAD_ZERO) and
        // 32-bit byte offsets.
        const bool kpad = (K % 64 == 0) || ((flags & 1) && (ta || lda >= (K + 63) / 64 * 64) && (!tb || ldb >= (K + 63) / 64 * 64));
        const bool km_ok = (ta && !tb) || kpad;
        const int64_t a_span = (int64_t)(ta ? K : M) * lda * 2, b_span = (int64_t)(tb ? N : K) * ldb * 2;
        const bool fast = km_ok && a_span < 0x7FFFFFF0ll && b_span < 0x7FFFFFF0ll && ldc % 8 == 0 && (out_fp32 || N % 8 == 0) &&
                          (!aux || (ldaux % 8 == 0 && N % 8 == 0)) && (!resid || (ldr % 8 == 0 && N % 8 == 0)) && !(flags & 2);
        PROF_START(cls, s);
        // 256x256 persistent tiles once they give most of the chip a tile (or a split-K launch sizes its own item count); the
        // 128x128 kernel (2 workgroups per CU) below that: at the default config (E=256, B=1: M=1024) the 256-tile kernels ran 4-16
        // workgroups on 256 CUs (27 us for a 4-tile launch)
        const int64_t t256 = (int64_t)cdiv(M, 256) * cdiv(N, 256);
        const bool big = fast && !(flags & 4) && ((flags & (8 | 16)) || ((int64_t)M * N >= 512ll * 512 && (t256 >= 192 || splitk_req > 1)));
        // measured at the C2 shapes (tools/kbench.py): both-K-contiguous (dgrad) is fastest on the 2-stage BK=64 kernel
        // (its DMA pieces are whole 128-byte lines); forward and wgrad on the 4-stage BK=32 deep pipeline.
        const bool prefer_p4 = !(!ta && tb);
        if ((big && !(flags & 8) && K % 32 == 0 && prefer_p4) || (big && (flags & 16))) {
            // deep-pipeline kernels: split granularity is a 32-deep k-step.  cfg 1: 128x256 tiles, 2 workgroups per CU
            // (one's epilogue/store drain overlaps the other's main loop); cfg 2: 256x256, 1 per CU.
            const int nk32 = cdiv(K, P_BK);
            const int want = std::max(1, std::min(splitk_req, nk32));
            const int per32 = cdiv(nk32, want);
            const int nsplit = cdiv(nk32, per32);
            const int cfg = (flags & 32) ? 1 : 2;
            // split-K: partial slabs + reduce when the registered workspace is large enough, else f32 atomics
            const bool slabs = ep.atomic && nsplit > 1 && ldc == N && (N % 4 == 0) && g_slab_ws &&
                               (size_t)nsplit * M * N * 4 <= g_slab_bytes && !(flags & 128);
            CHECK_SCHED(sched_next(s, ep, ex, sched_lock, &sched_used));
            Epilogue ep2 = ep;
            if (slabs) ep2.atomic = 0;
            const bool swap = !ep2.atomic;
            void* Cdst = slabs ? (void*)g_slab_ws : C;
            const int64_t sstride = slabs ? (int64_t)M * N : 0;
            ep2.colsum = colsum_out;
            if (!ta && !tb) colsum_fused = launch_p4<true, false>(s, cfg, swap, M, N, K, a, lda, b, ldb, Cdst, ldc, ep2, per32, nsplit, sstride, max_wgs);
            else if (!ta && tb) colsum_fused = launch_p4<true, true>(s, cfg, swap, M, N, K, a, lda, b, ldb, Cdst, ldc, ep2, per32, nsplit, sstride, max_wgs);
            else if (ta && !tb) colsum_fused = launch_p4<false, false>(s, cfg, swap, M, N, K, a, lda, b, ldb, Cdst, ldc, ep2, per32, nsplit, sstride, max_wgs);
            else colsum_fused = launch_p4<false, true>(s, cfg, swap, M, N, K, a, lda, b, ldb, Cdst, ldc, ep2, per32, nsplit, sstride, max_wgs);
            if (slabs) {
                const int64_t n4 = (int64_t)M * N / 4;
                gemm_slab_reduce_kernel<<<(int)std::min<int64_t>(cdiv64(n4, 256), 2048), 256, 0, s>>>(g_slab_ws, (float*)C, n4, n4, nsplit);
            }
        } else if (big) {
            const int tiles_n = cdiv(N, H_BN), ntiles = tiles_n * cdiv(M, H_BM);
            const int nsplit = cdiv(nk, per);
            const int g1 = std::min(ntiles * nsplit, max_wgs);
            const bool swap = !ep.atomic;
            ep.colsum = colsum_out;
            if (lnm == 1 && out_fp32) {
                // ln_f into the tied-logits GEMM: fp32 output, the last tile column may be ragged (EPI_PLAIN32)
                ln_done = !ta && tb && swap && act == 0 && !resid && !ep.drop.thr && M % 256 == 0 && nsplit == 1 && !colsum_out && bias && ex.ln.cs &&
                          (ex.ln.np == 2 || ex.ln.np == 3) && K == 256 * ex.ln.np;
            } else if (lnm) {
                const int kind = (!ta && tb) ? epi_kind_of(ep, M, N, swap, false) : EPI_GENERIC;
                const bool in_ok = !(lnm & 1) || (ex.ln.np >= 2 && ex.ln.np <= 3 && (kind == EPI_RESID ? (ex.ln.gamma && ex.ln.beta && N == 256 * ex.ln.np)
                                                                                                     : (ex.ln.cs && bias && K == 256 * ex.ln.np)));
                const bool kind_ok = (lnm == 1 && (kind == EPI_PLAIN || kind == EPI_GELU_AUX)) || ((lnm & 2) && kind == EPI_RESID);
                ln_done = kind_ok && in_ok && nsplit == 1 && !colsum_out;
            }
            if (!ln_done) {
                cmp_set_error("gemm: this launch cannot carry the LayerNorm epilogue that was asked for (M=%d N=%d K=%d ta=%d tb=%d act=%d): "
                              "bf16, A[M,K] . W^T[N,K], whole 256x256 tiles, 2 or 3 segments of 256 columns", M, N, K, ta, tb, act);
                return CMP_ERR_INVALID;
            }
            CHECK_SCHED(sched_next(s, ep, ex, sched_lock, &sched_used));
            if (!ta && !tb) colsum_fused = launch_256<true, false>(s, g1, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
            else if (!ta && tb) colsum_fused = launch_256<true, true>(s, g1, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
            else if (ta && !tb) colsum_fused = launch_256<false, false>(s, g1, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
            else colsum_fused = launch_256<false, true>(s, g1, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, nsplit, tiles_n, ntiles);
        } else if (fast) {
            const int tiles_n = cdiv(N, G_BN), ntiles = tiles_n * cdiv(M, G_BM);
            dim3 g1(ntiles, cdiv(nk, per));
            const bool swap = !ep.atomic;
            ep.colsum = colsum_out;           // (fused by the compile-time kinds only; the run-time epilogue ignores it)
            if (!ta && !tb) launch_fast<true, false>(s, g1, smem, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles);
            else if (!ta && tb) colsum_fused = launch_fast<true, true>(s, g1, smem, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles);
            else if (ta && !tb) launch_fast<false, false>(s, g1, smem, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles);
            else launch_fast<false, true>(s, g1, smem, swap, M, N, K, a, lda, b, ldb, C, ldc, ep, per, tiles_n, ntiles);
        } else
        // A_KM = !ta ; B_KM = tb
        if (!ta && !tb) gemm_bf16_kernel<true, false><<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per);
        else if (!ta && tb) gemm_bf16_kernel<true, true><<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per);
        else if (ta && !tb) gemm_bf16_kernel<false, false><<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per);
        else gemm_bf16_kernel<false, true><<<grid, 256, smem, s>>>(M, N, K, a, lda, b, ldb, C, ldc, ep, per);
        {   // algorithmic HBM bytes: A, B (+ bias, gelu' operand, residual) in; C (+ the stored pre-activation) out, once each
            const double oe = out_fp32 ? 4.0 : 2.0;
            const double gb = 2.0 * ((double)M * K + (double)K * N) + oe * M * N + (bias ? 4.0 * N : 0.0) + (aux ? 2.0 * M * N : 0.0) +
                              (resid ? 2.0 * M * N : 0.0);
            PROF_STOP(cls, s, 2.0 * M * N * K, gb);
        }
    }
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            if (sched_used) sched_used->dirty = true;      // the launch never ran: it did not zero the other counter set
            cmp_set_error("%s:%d: gemm launch failed: %s", __FILE__, __LINE__, hipGetErrorString(le));
            return CMP_ERR_HIP;
        }
    }
    if (sched_lock.owns_lock()) sched_lock.unlock();
    CMP_REQUIRE(ln_done, "gemm: a LayerNorm epilogue was asked of a launch that went to a kernel without one (M=%d N=%d K=%d dtype=%d flags=%d)", M, N, K, dtype, flags);
    if (colsum_out && !colsum_fused) return cmp_k_colsum(stream, C, ldc, colsum_out, M, N, dtype);
    return CMP_OK;
}
