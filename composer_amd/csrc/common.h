// common.h -- shared device/host helpers for libcomposer_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <math.h>
#include <type_traits>

#include "../../include/composer_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define WAVE 64

// ---- error plumbing -----------------------------------------------------------------------------
void cmp_set_error(const char* fmt, ...);

#define HIP_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            cmp_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            (void)hipGetLastError();   /* reported: do not leave it for the next KERNEL_CHECK of this thread to find */ \
            return CMP_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define CMP_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            cmp_set_error(__VA_ARGS__);             \
            return CMP_ERR_INVALID;                 \
        }                                           \
    } while (0)

#define KERNEL_CHECK() HIP_CHECK(hipGetLastError())

// live per-kernel-class timing (cmp_prof_begin / cmp_prof_end, model.hip)
extern thread_local int g_prof_cls;
void prof_start(int cls, hipStream_t s);
void prof_stop(int cls, hipStream_t s, double work, double bytes);
// work: the launch's algorithmic flops (matrix-core classes) or bytes (memory-bound classes); bytes: its algorithmic HBM
// bytes -- every operand read once, every result written once (what the PMC traffic of profiles/ is compared with)
#define PROF_START(cls, s) do { if (g_prof_cls == (cls)) prof_start((cls), (s)); } while (0)
#define PROF_STOP(cls, s, work, bytes) do { if (g_prof_cls == (cls)) prof_stop((cls), (s), (work), (bytes)); } while (0)

// Per-launch extras of the GEMM launcher (gemm.hip: gemm_run) that the C ABI's cmp_k_gemm does not carry.  The model driver
// fills one per call, so nothing about a launch lives in process-wide state.
// Item-counter workspace of the persistent GEMM kernels (gemm.hip: ItemPuller / sched_next): two alternating counter sets on
// the device.  One per cmp_ctx (allocated on first use, freed by cmp_ctx_destroy); launches that arrive without a context
// (cmp_k_gemm) use a process-wide table keyed by (device, stream) under a lock that is held across the launch.
struct SchedWs {
    uint32_t* dev = nullptr;
    int parity = 0;
    bool dirty = false;          // a launch that used the counters failed: both sets are re-zeroed before the next use
};
// LayerNorm folded into the epilogues of the forward GEMMs (gemm.hip: epi_tile<.., LNM, NP>; full 256x256 tiles, bf16).  The
// statistics of a row travel as PARTIALS: (mean, M2 = sum of squared deviations from that mean) of every 256-column segment of
// the row, written by the epilogue of the GEMM (or the embedding kernel) that produced the row and merged by whoever consumes
// it (Chan's parallel update -- never a sum of squares).
//   consumer, fold (c_attn / c_fc: transformer.py:583-584,591 followed by Conv1D :205-209):
//       LN(x).W + b = rstd * (x.(gamma o W)) - rstd*mean * colsum(gamma o W) + (beta.W + b)
//     the GEMM runs on the RAW rows x and the gamma-scaled weight shadow; `cs` = colsum of that shadow, bias = beta.W + b.
//   consumer, residual (attention c_proj, :587 r = LN1(x) + dropout(proj)): the residual operand is rebuilt from the raw
//     row, its statistics and gamma / beta.
//   producer (both c_proj): the epilogue owns a 256-column tile of every output row and emits that segment's partial.
struct LnEpi {
    const float* in_part = nullptr;   // [rows][np][2] partial statistics of the LayerNorm input's rows (null: no LayerNorm on the way in)
    int np = 0;                       // segments per row = E / 256
    float eps = 0.f;
    const float* cs = nullptr;        // fold: [N] column sums of the gamma-scaled bf16 weight
    const float* gamma = nullptr;     // residual rebuild: [N] gamma, beta of the LayerNorm whose output is the residual operand
    const float* beta = nullptr;
    float* out_part = nullptr;        // [rows][N / 256][2] partial statistics of the OUTPUT rows (null: not wanted)
    // Round 6, the backward pass of the fused block path ("scale" mode, with in_part): nothing is normalised, the row's rstd is a
    // FACTOR -- EPI_GELUGRAD writes rstd o (acc * gelu'(aux)) (the column sums of the unscaled product still go to the launch's colsum);
    // EPI_RESID adds rstd o resid instead of resid (model.hip: backward).
    int scale = 0;
};
// (mean, rstd) of a row from its np partials of 256 columns each
__device__ __forceinline__ void ln_merge_parts(const float* __restrict__ p, int np, float eps, float& mean, float& rstd) {
    float mu = 0.f;
    for (int s = 0; s < np; s++) mu += p[2 * s];
    mu /= (float)np;
    float m2 = 0.f;
    for (int s = 0; s < np; s++) {
        const float d = p[2 * s] - mu;
        m2 += p[2 * s + 1] + 256.0f * d * d;
    }
    mean = mu;
    rstd = __builtin_amdgcn_rsqf(m2 / (256.0f * (float)np) + eps);
}

struct GemmExtra {
    LnEpi ln;                    // LayerNorm fused into this launch's epilogue (forward GEMMs of the fused block path, model.hip)
    float* colsum = nullptr;     // also add the column sums of the output to colsum[0..N) (the bias gradient that goes with an
                                 // input-gradient GEMM); fused into the epilogue where possible, else a colsum pass after it
    float* slab_ws = nullptr;    // split-K: per-split fp32 partial tiles + fixed-order reduce instead of float atomics
    size_t slab_bytes = 0;
    int role = -1;               // cmp_prof_* timing class: 0 forward, 1 dgrad, 2 wgrad, -1 = by operand layout
    int max_wgs = 0;             // cap on the persistent kernels' grid (CUs left to a concurrent RCCL kernel); 0 = all 256
    bool rev = false;            // persistent 256x256 kernel: walk every XCD group's run of tiles from its end (gemm.hip: item_coords)
    bool dp = false;             // the launch belongs to a data-parallel job: persistent kernels hand their items out dynamically
    SchedWs* sched = nullptr;    // the calling context's item-counter workspace (a cmp_ctx is single-threaded by contract: no lock)
    struct WgradWs* wws = nullptr;     // grouped weight gradients: the caller's partial-tile workspace selects the last-arriver form
                                       // (no float atomics: deterministic mode); null: the float-atomic form (default, faster)
};
// Partial-tile workspace of the grouped weight-gradient launches of ONE stream (launches on a stream run one after the other, so
// every decoder block's launch uses the same slots); grown by wgrad_group_run when an item table needs more, freed by the owner.
struct WgradWs {
    float* ptr = nullptr;
    size_t bytes = 0;
};
void wgrad_ws_free(WgradWs* w);
// One launch for several split-K weight gradients that contract over the same K rows (gemm.hip: gemm_wgrad_group_kernel):
// problem i is C_i[M_i, N_i] (fp32, accumulated into) += A_i^T . B_i with A_i stored [K, M_i] and B_i stored [K, N_i], bf16.
struct WgradProblem { const void* A; int lda; const void* B; int ldb; float* C; int ldc; int M, N; };
struct WgradGroup {              // owned by the caller, one per call site whose problems keep their pointers (a decoder block)
    void* dev = nullptr;         // device copy of the problem descriptors + the item table
    size_t dev_bytes = 0;
    std::string key;             // what that copy was built from (problems, K, workgroup count)
    std::string host;            // the bytes uploaded (kept until the next rebuild: the upload is stream-ordered)
    int nitems = 0, grid = 0;    // nitems == 0 with a key: the cost model chose one launch per problem for these shapes
    int nslots = 0, ntiles = 0;  // last-arriver form: workspace slots (one per K range of a multi-range tile) and tiles (counters)
    size_t la_off = 0;           // ... byte offset, in `dev`, of {int cnt[ntiles rounded up to 4], int slot0[ntiles]}
    bool la = false;             // the table was cut for the last-arriver form (finer K ranges than the float-atomic form affords)
    bool cnt_dirty = false;      // a launch failed after it may have drawn tickets: the counters are zeroed before the next one
    bool force = false;          // use the grouped launch even where the cost model prefers separate launches (kernel-level tests)
    int64_t rebuilds = 0;        // item tables built so far (a steady train loop builds each block's once: cmp_wgrad_group_rebuilds)
};
// returns CMP_OK with *handled = false when the shapes do not fit the grouped kernel (the caller then runs the problems one by one)
int wgrad_group_run(void* stream, WgradGroup* g, const WgradProblem* probs, int nprob, int K, const GemmExtra& ex, bool* handled);
void wgrad_group_free(WgradGroup* g);
int gemm_run(void* stream, int dtype, int ta, int tb, int M, int N, int K, const void* A, int lda, const void* Bm, int ldb,
             void* C, int ldc, const float* bias, int act, void* aux, int ldaux, const void* resid, int ldr, int out_fp32,
             int splitk, float p_drop, uint64_t seed, uint32_t rng_stream, int flags, const GemmExtra& ex);
// Round 6 (backward pass of the fused block path): the attention backward kernels store rstd o [dQ | dK | dV], rstd being that of the
// token's row in the LayerNorm whose output fed c_attn (ln_stats_merge_kernel) -- the c_attn weight gradient then
// runs on the raw rows (elementwise.hip: wgrad_ln_fix_kernel).  The bias gradient stays the column sums of the unscaled gradient.
struct AttnLnRows {
    const float* rstd = nullptr;      // [tokens]; null: nothing of this
};
int attn_bwd_run(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dqkv,
                 int B, int T, int H, int D, float sc, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream,
                 float* bias_grad, AttnLnRows lr = AttnLnRows());
int attn_fwd_run(void* stream, const void* qkv, void* o, float* lse, int B, int T, int H, int D, float sc, int dtype, float p_drop,
                 uint64_t seed, uint32_t rng_stream, const float* amask = nullptr);
int attn_probs_run(void* stream, const void* qkv, const float* lse, const float* amask, float* out, int B, int q0, int T, int H, int D,
                   float sc, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream);

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- dropout mask: bit-identical to oracle/transformer_oracle.py::dropout_keep -------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t stream, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t h = mix32(lo ^ seed);
    return mix32(h ^ (hi * 0x9E3779B1u) ^ stream);
}
static inline uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    if (t < 0) t = 0;
    return (uint32_t)t;
}
// site: 0 embed, 1 attention probabilities, 2 attn c_proj output, 3 mlp output
static inline uint32_t drop_stream(int64_t step, int layer, int site) {
    return (uint32_t)(((step * 64 + layer) * 4 + site) & 0xFFFFFFFFll);
}

// attention-probability dropout (site 1): one full-quality hash per (batch*head, query) row, then a light
// per-key mix -- the mask is a pure function of (row, key) so forward, dQ and dK/dV regenerate it in any layout.
// oracle: transformer_oracle.py::dropout_keep_attn
__host__ __device__ __forceinline__ uint32_t attn_row_hash(uint32_t seed, uint32_t stream, uint32_t row) {
    return mix32(mix32(row ^ seed) ^ stream);
}
#define ATTN_G 0x9E3779B1u
#define ATTN_C0 0xEBCA6Bu     // 24-bit odd multipliers: v_mul_u32_u24 is a full-rate instruction
#define ATTN_C1 0xB2AE35u
#define ATTN_C2 0xD4EB2Fu
#define ATTN_C3 0x5667B1u
// keep(row, key) = (((rowhash ^ ((key>>2)*G)) & 0xFFFFFF) * C24[key&3]) mod 2^32 >= thr : per element ONE multiply + compare;
// the xor base is shared by 4 consecutive keys (statistics checked in tests/test_oracle.py).
__host__ __device__ __forceinline__ uint32_t attn_elem_hash(uint32_t rowhash, uint32_t key) {
    const uint32_t x = rowhash ^ ((key >> 2) * ATTN_G);
    const uint32_t k3 = key & 3u;
    const uint32_t cc = k3 == 0 ? ATTN_C0 : (k3 == 1 ? ATTN_C1 : (k3 == 2 ? ATTN_C2 : ATTN_C3));
    return (x & 0xFFFFFFu) * cc;
}

struct DropCfg {
    uint32_t thr;     // keep iff hash >= thr ; thr == 0 -> dropout disabled
    uint32_t seed;
    uint32_t stream;
    float scale;      // 1/(1-p)
};
static inline DropCfg make_drop(float p, uint64_t seed, uint32_t stream) {
    DropCfg d;
    d.thr = p > 0.f ? drop_threshold(p) : 0u;
    d.seed = (uint32_t)seed;
    d.stream = stream;
    d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}
// Flat sites (0 embedding sum, 2 attention c_proj output, 3 MLP output) are [token row, feature column] matrices and
// use the same two-level mask as the attention probabilities: one full hash per row (amortised over the 8 columns a
// lane handles), one 24-bit multiply + compare per element.  (The first version hashed the flat index with two
// mix32 rounds per ELEMENT: 5 quarter-rate 32-bit multiplies, ~110 cycles per wave-element -- more epilogue time per
// c_proj tile than its K=512 main loop.)   oracle: transformer_oracle.py::dropout_keep_rows
__device__ __forceinline__ uint32_t drop_row_hash(const DropCfg& d, uint32_t row) { return attn_row_hash(d.seed, d.stream, row); }
__device__ __forceinline__ float apply_drop_rc(const DropCfg& d, uint32_t rowhash, uint32_t col, float v) {
    return attn_elem_hash(rowhash, col) >= d.thr ? v * d.scale : 0.0f;
}
__device__ __forceinline__ float apply_drop(const DropCfg& d, uint32_t row, uint32_t col, float v) {
    if (d.thr == 0u) return v;
    return apply_drop_rc(d, drop_row_hash(d, row), col, v);
}

// ---- numeric helpers ----------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

#define GELU_C 0.7978845608028654f   // sqrt(2/pi), transformer.py:40
#define GELU_K 0.044715f

template <bool EXACT> __device__ __forceinline__ float tanh_f(float x) {
    if (EXACT) return tanhf(x);
    // tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly for |x| large
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(e + 1.0f);
}
// gelu(x) = 0.5 x (1 + tanh(u)), u = C (x + K x^3).  With 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)) =: s(x) the fast (bf16
// kernels) form needs one v_exp_f32 and one v_rcp_f32 and no IEEE division sequence (__frcp_rn expands to ~10
// instructions; the epilogue of a 256x256 c_fc tile was VALU-bound on it: ~25 instructions per element):
//   gelu  = x s                      gelu' = s (1 + x (1 - s) (2C + 6CK x^2))
// EXACT (fp32 kernels, decode) keeps the textbook form on tanhf for parity with the oracle.
#define GELU_W1 (-2.0f * GELU_C * 1.4426950408889634f)      // exp(-2u) = exp2(x (W1 + W2 x^2))
#define GELU_W2 (GELU_W1 * GELU_K)
__device__ __forceinline__ float gelu_sig(float x) {
    const float w = x * (GELU_W1 + GELU_W2 * (x * x));
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}
template <bool EXACT> __device__ __forceinline__ float gelu_f(float x) {
    if (EXACT) {
        float t = tanh_f<true>(GELU_C * (x + GELU_K * x * x * x));
        return 0.5f * x * (1.0f + t);
    }
    return x * gelu_sig(x);
}
template <bool EXACT> __device__ __forceinline__ float gelu_grad_f(float x) {
    if (EXACT) {
        float t = tanh_f<true>(GELU_C * (x + GELU_K * x * x * x));
        return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * GELU_C * (1.0f + 3.0f * GELU_K * x * x);
    }
    const float sg = gelu_sig(x);
    return sg * (1.0f + x * (1.0f - sg) * (2.0f * GELU_C + 6.0f * GELU_C * GELU_K * (x * x)));
}

// Wave-wide sum / max, the same value (and summation order) in every lane.  DPP row operations (quad_perm, row_half_mirror,
// row_mirror: four vector instructions reduce each row of 16 lanes) + v_readlane of the four row totals -- no ds_bpermute
// round trips through the LDS crossbar (the __shfl_xor form is six of them per reduction).  WAVE_REDUCE_SHFL selects that
// form for A/B timing.
#define CMP_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float cmp_readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef WAVE_REDUCE_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#else
    v += CMP_DPP_F(v, 0xB1);       // quad_perm(1,0,3,2)
    v += CMP_DPP_F(v, 0x4E);       // quad_perm(2,3,0,1)
    v += CMP_DPP_F(v, 0x141);      // row_half_mirror
    v += CMP_DPP_F(v, 0x140);      // row_mirror
    return (cmp_readlane_f(v, 0) + cmp_readlane_f(v, 16)) + (cmp_readlane_f(v, 32) + cmp_readlane_f(v, 48));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef WAVE_REDUCE_SHFL
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
#else
    v = fmaxf(v, CMP_DPP_F(v, 0xB1));
    v = fmaxf(v, CMP_DPP_F(v, 0x4E));
    v = fmaxf(v, CMP_DPP_F(v, 0x141));
    v = fmaxf(v, CMP_DPP_F(v, 0x140));
    return fmaxf(fmaxf(cmp_readlane_f(v, 0), cmp_readlane_f(v, 16)), fmaxf(cmp_readlane_f(v, 32), cmp_readlane_f(v, 48)));
#endif
}

// vector load/store of 8 (bf16) or 4 (fp32) contiguous elements = 16 bytes
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(Vec16<T>::v)*>(p) = r.v;
}

static inline size_t dtype_size(int dtype) { return dtype == CMP_BF16 ? 2 : 4; }
