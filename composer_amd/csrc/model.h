// model.h -- shared definitions of the device-resident model state (see model.hip for the layout notes)
#pragma once
#include "common.h"
#include <rccl/rccl.h>
#include <vector>
#include <string>
#include <map>

struct DecodeState;

struct Metrics {
    double loss_sum;
    long long correct;
    float loss_mean;
    float acc;
    int bad_ids;        // ids outside [0, V) seen (and clamped) by the device-pointer entry points
    int pad;
};

#define NCCL_CHECK(expr)                                                                         \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess) {                                                                 \
            cmp_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
            return CMP_ERR_RCCL;                                                                 \
        }                                                                                        \
    } while (0)

#define CHECK_RC(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != CMP_OK) return _rc; \
    } while (0)

struct cmp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // compute
    hipStream_t comm_stream = nullptr;  // RCCL
    ncclComm_t comm = nullptr;
    // cmp_dp_init_exchange: the sum over ranks is delegated to the caller (another transport; two ranks sharing one device in tests)
    int (*xfn)(void*, void*, int64_t, void*) = nullptr;
    void* xuser = nullptr;
    bool dp_on() const { return comm != nullptr || xfn != nullptr; }
    int rank = 0, nranks = 1;
    uint32_t seed_mix = 0;              // mix32(rank), xor-ed into every dropout seed: replicas draw independent masks
    int gemm_max_wgs = 0;               // cap on the persistent GEMM grids while a communicator exists (0 = all CUs)
    hipStream_t copy_stream = nullptr;  // host -> device id uploads of the pipelined train loop
    SchedWs gemm_sched;                 // item counters of the persistent GEMMs launched on `stream` (gemm.hip: sched_next)
};
void sched_ws_free(SchedWs* w);

struct ParamInfo {
    std::string name;
    int rank;
    int64_t shape[4];          // the reference's (logical) shape
    int64_t numel;             // logical element count
    int64_t offset;
    // head-size padding (cmp_model::Dl < D): 0 stored as is; 1 the last dimension is [3][H][Dl] stored as [3][H][D] (c_attn weight
    // and bias); 2 the first dimension is [H][Dl] stored as [H][D] (attention c_proj weight).  Padding entries are zero, stay zero.
    int pad = 0;
    int64_t store = 0;         // stored element count
};

struct LayerOff {   // element offsets into the flat buffers
    int64_t ln1_g, ln1_b, attn_w, attn_b, proj_w, proj_b, ln2_g, ln2_b, fc_w, fc_b, pr_w, pr_b;
    int64_t begin, end;
};

struct WDesc {      // one Conv1D weight in the flat buffers: [rows = in][cols = out]
    int64_t off;
    int rows, cols;
};

struct FoldDesc {   // one LayerNorm -> Conv1D pair of the fused block path (elementwise.hip: ln_fold_prep_kernel); offsets into the flat buffers
    int64_t w_off, b_off, g_off, be_off;   // weight [rows][cols], bias [cols], gamma [rows], beta [rows]
    int64_t out_off;                       // in the fold buffer: cs [cols], then bias' [cols]
    int rows, cols;
};

struct LayerAct {
    void *u, *qkv, *att, *r, *n, *fc, *g;
    float *ln1_mean, *ln1_rstd, *ln2_mean, *ln2_rstd, *lse;
    float *ln1_part = nullptr, *ln2_part = nullptr;    // fused block path: partial row statistics [tokens][E/256][2] of xs[i] and of r
};

struct cmp_model {
    cmp_ctx* ctx = nullptr;
    cmp_model_cfg cfg;
    int V, E, W, L, H, D, ldz;
    // Head sizes other than 16 / 32 / 64 / 128 (E / H = Dl) run on the next supported size D with zero-filled columns: the c_attn
    // output is [.., 3 * Ea], Ea = H * D, head h of q / k / v at h * D with Dl live columns; the attention kernels see (H, D) and
    // the scale 1/sqrt(Dl); zero columns of q, k, v leave every score and every output value unchanged and their gradients are
    // exactly zero.  Parameters, presents and `past` keep the reference's shapes at the ABI.  Dl == D: Ea == E, nothing changes.
    int Dl = 0, Ea = 0;
    int dtype;
    size_t es;                 // activation element size
    std::vector<ParamInfo> params;
    std::map<std::string, int> index;
    int64_t total = 0;         // elements in the flat buffers
    int64_t off_wte = 0, off_wpe = 0, off_lnf_g = 0, off_lnf_b = 0;
    std::vector<LayerOff> lo;
    float *P = nullptr, *G = nullptr, *Am = nullptr, *Av = nullptr;
    bf16_t* S = nullptr;       // bf16 shadow (bf16 mode)
    // Transposed bf16 copies of the four Conv1D weights of every block ([in,out] -> [out,in], same flat offsets), refreshed
    // at the top of every forward pass: with the weight K-contiguous the forward GEMMs run on the kernel whose both
    // operands are read with ds_read_b128 (c_attn 275 -> 209 us, c_fc 393 -> 350, c_proj 104 -> 93, mlp c_proj 298 -> 272
    // at B=128); the refresh moves 38 MB (~12 us).
    bf16_t* ST = nullptr;
    void* wdesc = nullptr;     // device table of (offset, rows, cols) for the 4L matrices
    // LayerNorm-fused block path (model.hip: ln_fused_ok): c_attn and c_fc run on the RAW residual rows and a gamma-scaled
    // transposed shadow (in ST, at the weight's own offset), with the fold vectors cs / bias' of common.h's LnEpi in `lnfold`
    // ([L] x {c_attn: cs[3E], bias'[3E]; c_fc: cs[4E], bias'[4E]}); refreshed with the transposed copies when a parameter moved.
    void* wdesc_plain = nullptr;       // the 2L matrices that stay unscaled in that mode (both c_proj)
    void* fdesc = nullptr;             // 2L FoldDesc
    void* ln_sites = nullptr;          // 2L {partials, mean, rstd} pointer triples of the blocks' LayerNorm sites (ln_stats_merge_kernel); rebuilt with the workspace
    float* lnfold = nullptr;
    int64_t fold_stride = 0;           // floats per block in lnfold; c_attn vectors at 0, c_fc vectors at 6E
    int st_state = 0;                  // what ST holds: 0 nothing, 1 plain transposes, 2 the fused path's set
    int64_t st_version = -1;           // ... of which param_version
    bool fused_last = false;           // the forward pass whose activations are held took the fused path (its backward must too)
    void* dmask3 = nullptr;            // second MLP-branch masked gradient (fused path: alternates with dmask from block to block)
    // ln_f folded into the tied-logits GEMM (inference passes on the fused path): the gamma-scaled copy of wte, its fold vectors
    // (cs then bias', `lnf_npad` entries each) and the partial statistics of the last block's output rows
    bf16_t* wte_lnf = nullptr;
    float* lnf_fold = nullptr;
    int lnf_npad = 0;
    float* lnf_part = nullptr;
    bool hf_valid = false;             // hf = ln_f(xs[L]) of the pass held has been written (cmp_hidden_get_at computes it on demand)
    int64_t iterations = 0;
    int64_t param_version = 0;         // bumped whenever a parameter value changes (cmp_param_set, Adam): the decode state's
                                       // transposed weight copies are refreshed when it has moved
    // workspace
    int capB = 0, capT = 0;
    std::vector<void*> allocs;
    std::vector<void*> xs;     // L+1 residual-stream tensors
    const int32_t* fwd_pos_ids = nullptr;      // cmp_forward_ex only: per-token position / token-type ids of the pass being run
    const int32_t* fwd_type_ids = nullptr;
    const float* fwd_amask = nullptr;          // ... and its additive attention mask term, float [B, past + T]
    float* const* fwd_probs_out = nullptr;     // ... and, when asked for, L host tensors [B, H, T, past + T] for the attention weights
    float* fwd_probs_dev = nullptr;            //     (one layer's worth of device staging)
    void* io_buf[3] = {nullptr, nullptr, nullptr};     // device staging of the inspection entry points (model.hip: io_scratch)
    size_t io_buf_bytes[3] = {0, 0, 0};
    std::vector<LayerAct> act;
    void *hf = nullptr, *dlogits = nullptr;
    float* logits_pack = nullptr;      // [tokens, V] contiguous copy of the logits for cmp_forward's host transfer (allocated on first use)
    float *logits = nullptr, *lnf_mean = nullptr, *lnf_rstd = nullptr, *row_loss = nullptr, *delta = nullptr;
    int32_t *row_correct = nullptr, *x_dev = nullptr, *y_dev = nullptr;
    void *dx = nullptr, *dr = nullptr, *tmpE = nullptr, *dmask = nullptr, *dfc = nullptr, *dqkv = nullptr;
    int* embed_ws = nullptr;           // workspace of the sorted embedding backward (elementwise.hip: embed_bwd_sort_ws_words)
    int64_t embed_ws_words = 0;
    void* dmask2 = nullptr;            // the attention branch's masked gradient: both masked copies of a block stay live until its grouped wgrad launch
    std::vector<WgradGroup> wgrad_groups;      // per decoder block: item table + problem descriptors of that launch
    WgradWs wgrad_ws;                          // ... and the partial-tile workspace they share (one launch at a time on the stream)
    void* ln_ws = nullptr;
    void* slab = nullptr;              // split-K slab workspace (deterministic mode only)
    int64_t slab_bytes = 0;
    Metrics* metrics = nullptr;        // device
    Metrics* metrics_host = nullptr;   // pinned
    float* dp_metrics = nullptr;       // device [4]: {loss mean, accuracy, 1, 0} summed over ranks (3-float all-reduce)
    // pipelined train loop (cmp_train_step_async): STAGES pinned id buffers + device copies, one metrics slot and two
    // events per in-flight step
    static constexpr int STAGES = 3;
    int32_t* stage_host[STAGES] = {nullptr, nullptr, nullptr};     // pinned [2][cap tokens]: x then y
    int32_t* stage_dev[STAGES] = {nullptr, nullptr, nullptr};
    Metrics* stage_metrics = nullptr;                              // pinned [STAGES]
    hipEvent_t stage_uploaded[STAGES] = {nullptr, nullptr, nullptr};
    hipEvent_t stage_done[STAGES] = {nullptr, nullptr, nullptr};
    int64_t stage_ticket[STAGES] = {-1, -1, -1};
    int64_t next_ticket = 0;
    int64_t stage_cap = 0;             // tokens per staging buffer
    int lastB = 0, lastT = 0, last_past = 0;   // shape of the forward pass whose activations are held (cmp_present_get)
    int64_t fwd_gen = 0;               // bumped by every forward pass (train steps and decode prefill included): a Presents
                                       // object remembers the pass it came from and cmp_present_get_at refuses any other
    std::vector<hipEvent_t> bucket_ev; // L+2 events
    hipEvent_t comm_done = nullptr, metrics_ev = nullptr;
    // data-parallel telemetry (cmp_dp_stats): a ring of timed event pairs around the compute stream's wait for the
    // communication stream at the end of a step = the communication time the backward pass did NOT hide
    static constexpr int DP_RING = 32;
    hipEvent_t dp_wait_a[DP_RING] = {}, dp_wait_b[DP_RING] = {};
    bool dp_wait_used[DP_RING] = {};
    int64_t dp_steps = 0;              // steps recorded since the last reset
    int64_t dp_folded = 0;             // ... of which already summed into dp_exposed_ms
    double dp_exposed_ms = 0.0;
    int64_t dp_bytes_step = 0;         // bytes handed to ncclAllReduce by the last step (gradients + the 3-float metrics message)
    int dp_msgs_step = 0;              // all-reduce calls of the last step
    int dp_buckets_updated = 0;        // buckets of the step being enqueued whose Adam update is already on the communication stream
    bool poisoned = false;             // a data-parallel step failed after some of them: parameters partially stepped (train steps refuse)
    std::vector<char> reload_seen;     // while poisoned: which parameters cmp_param_set has replaced since (all of them clears the flag)
    int ln_fused_mode = -1;            // COMPOSER_LN_FUSED as read when the model was created (-1 unset, 0 off, 2 training passes too)
    DecodeState* dec = nullptr;
    int gemm_role = -1;                // profiler class of the GEMMs being enqueued (0 while the forward pass is)

    // dropout seed of this replica: the model seed with the data-parallel rank folded in (rank 0: the seed itself)
    uint64_t drop_seed() const { return (uint64_t)((uint32_t)cfg.seed ^ ctx->seed_mix); }
    const void* w(int64_t off) const { return dtype == CMP_BF16 ? (const void*)(S + off) : (const void*)(P + off); }
};


template <typename Tp> static int dev_alloc(cmp_model* m, Tp** p, size_t bytes) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, bytes ? bytes : 16));
    m->allocs.push_back(q);
    *p = (Tp*)q;
    return CMP_OK;
}

// elementwise.hip
int embed_fwd_run(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out, int B, int T, int E, int pos0,
                  int dtype, float p_drop, uint64_t seed, uint32_t rng_stream, const int32_t* pos_ids, const int32_t* type_ids);
int embed_bwd_run(void* stream, const int32_t* ids, const void* dh, float* dwte, float* dwpe, int B, int T, int E, int pos0,
                  int dtype, float p_drop, uint64_t seed, uint32_t rng_stream, int V, float* det_ws, size_t det_ws_bytes,
                  int sort_V = 0, int* sort_ws = nullptr, int64_t sort_ws_words = 0);
int64_t embed_bwd_sort_ws_words(int64_t ntok, int V);
struct LnBwdFused {     // layernorm_bwd_kernel<.., FUSED>: statistics from partials, the LayerNorm output written beside dx
    const float* part = nullptr;   // [rows][np][2]
    int np = 0;
    float eps = 0.f;
    const float* beta = nullptr;
    void* yout = nullptr;          // xhat * gamma + beta (null: not wanted)
    int prescaled = 0;             // dy holds rstd o (the gradient): what a dgrad GEMM yields when its A operand was stored rstd-scaled
                                   // (round 6: the weight gradients of the fused block path run on the raw LayerNorm input rows)
};
// G[k, j] = gamma[k] * (G[k, j] - mean_k' G[k', j]) + beta[k] * colsum[j] over one or two [rows, cols] weight gradients that were
// accumulated as (raw rows)^T . (rstd o D): turns them into LN(raw rows)^T . D (elementwise.hip: wgrad_ln_fix_kernel)
int wgrad_ln_fix_run(void* stream, float* G0, int rows0, int cols0, const float* gamma0, const float* beta0, const float* colsum0,
                     float* G1, int rows1, int cols1, const float* gamma1, const float* beta1, const float* colsum1);
int layernorm_bwd_run(void* stream, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* resid, void* dx, float* dgamma, float* dbeta, void* ws, int rows, int E, int dtype, void* dmask,
                      float* colsum, float p_drop, uint64_t seed, uint32_t rng_stream, bool deterministic, const LnBwdFused* fz = nullptr,
                      bool keep_dmask = false, bool prescaled = false);
int ln_stats_merge_run(void* stream, const void* sites_dev, int nsites, int np, float eps, int rows);      // sites: {part, mean, rstd} triples of pointers
int embed_fwd_stats_run(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out, float* part, int B, int T,
                        int E, int pos0, float p_drop, uint64_t seed, uint32_t rng_stream);
int ln_fold_prep_run(void* stream, const float* P, void* ST, float* fold, const void* desc_dev, int ndesc, int max_cols);
int lnf_fold_prep_run(void* stream, const float* wte, const float* gamma, const float* beta, void* out, float* cs, float* bias, int V, int E,
                      int npad);
int colsum_run(void* stream, const void* X, int ldx, float* out, int rows, int cols, int dtype, float* det_ws, size_t det_ws_bytes);
int launch_metrics_reduce(hipStream_t s, const float* row_loss, const int32_t* row_correct, int rows, void* metrics);
int launch_cast_bf16(hipStream_t s, const float* in, void* out, int64_t n);
// model.hip
int ensure_workspace(cmp_model* m, int B, int T);
int model_forward(cmp_model* m, const int32_t* x_dev, int B, int T, bool training, int64_t step, int past_len = 0);
// decode.hip
void decode_state_free(DecodeState* d);
