/*
 * composer_hip.h -- C ABI of libcomposer_hip.so: the MI355X (gfx950) implementation of
 * galacticglum/composer's Transformer hot path (teacher-forced train step, evaluation step,
 * autoregressive decode).
 *
 * The reference has no FFI: its seam is the Python class `composer.models.Transformer`
 * (reference composer/models/transformer.py:599-960) called from composer/cli.py:95-141,516-680.
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success, a negative cmp_status otherwise;
 * cmp_last_error() returns a thread-local, library-owned message for the last failure.
 * Plain pointers and sizes only.  "host" pointers are ordinary CPU memory owned by the caller for
 * the duration of the call; "dev" pointers are HIP device pointers (the cmp_k_* kernel-level entry
 * points, used by the parity tests and micro-benchmarks, take dev pointers and a hipStream_t passed
 * as void*).  A cmp_ctx is bound to one device and is not thread-safe.
 */
#ifndef COMPOSER_HIP_H
#define COMPOSER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmp_ctx cmp_ctx;
typedef struct cmp_model cmp_model;

enum cmp_status {
    CMP_OK = 0,
    CMP_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    CMP_ERR_HIP = -2,       /* a HIP runtime call failed */
    CMP_ERR_RCCL = -3,      /* an RCCL call failed */
    CMP_ERR_STATE = -4      /* call sequence error (e.g. decode_steps before decode_begin) */
};

enum cmp_dtype {
    CMP_FP32 = 0,           /* fp32 storage + f32-input MFMA: the parity mode */
    CMP_BF16 = 1            /* bf16 activations / weight shadow, fp32 accumulate + master: throughput */
};

enum cmp_decode_mode {
    CMP_DECODE_LITERAL = 0, /* composer/cli.py:663-676 as written: `past` never fed back */
    CMP_DECODE_KV = 1       /* model(x, past=presents), transformer.py:735-765,423-426 */
};

/* Constructor arguments of Transformer.__init__ (transformer.py:610-614) plus device-side sizing. */
typedef struct cmp_model_cfg {
    int32_t vocab_size;        /* V  */
    int32_t embedding_size;    /* E, multiple of 8 */
    int32_t window_size;       /* W = rows of wpe */
    int32_t layers;            /* L  */
    int32_t heads;             /* H, E % H == 0 (transformer.py:255), head size E/H <= 128 (see cmp_model_create) */
    float   ln_eps;            /* layer_normalization_epsilon */
    int32_t scale_attention;   /* `scale` */
    int32_t use_layer_norm;    /* `use_layer_normalization` */
    float   attn_dropout;      /* attention_dropout_rate */
    float   resid_dropout;     /* residual_dropout_rate */
    int32_t dtype;             /* cmp_dtype */
    int32_t max_batch;         /* largest B of any later call */
    int32_t max_seq;           /* largest T of any later call (<= window_size) */
    uint64_t seed;             /* dropout / sampling seed */
} cmp_model_cfg;

const char* cmp_last_error(void);
int cmp_version(void);
/* key (SHA-256 over sources, headers, flags, compiler) this library was linked from: composer_amd/build.py verify() */
const char* cmp_build_key(void);
/* number of visible HIP devices (0 when none); never initialises a device context beyond the count */
int cmp_device_count(void);

/* ---- context: one per (process, device) ------------------------------------------------------ */
int cmp_ctx_create(int device, cmp_ctx** out);
int cmp_ctx_destroy(cmp_ctx* ctx);
int cmp_sync(cmp_ctx* ctx);
/* hipStream_t of the compute stream (for HIP-event timing by the caller) */
void* cmp_ctx_stream(cmp_ctx* ctx);

/* ---- data parallel: RCCL communicator, one rank per process (new; the reference is single-device) */
int cmp_dp_unique_id(void* id128);                                   /* rank 0: fills 128 bytes */
int cmp_dp_init(cmp_ctx* ctx, int rank, int nranks, const void* id128);
int cmp_dp_allreduce_test(cmp_ctx* ctx, float* host_inout, int n);    /* sum over ranks, for tests */
/* The same data-parallel step over a transport of the caller's instead of RCCL (a fabric RCCL does not drive; in tests/ two ranks
 * that share the one GPU of a box and sum over gloo -- this image's RCCL refuses two ranks on one device).  Everything else is the
 * path cmp_dp_init selects: the same gradient buckets in the same order, the bucket's event on the compute stream, Adam on the
 * bucket right behind its sum on the communication stream, 1/nranks folded into the update, seed ^ mix32(rank) dropout masks,
 * the 3-float metrics message, dynamic GEMM item scheduling.  `fn` is called on the enqueuing thread once per message, every rank
 * in the same order: it must leave the element-wise sum over all ranks of the `count` floats at `dev_f32` (device memory) in place,
 * ORDERED ON `hip_stream` (the communication stream, which already waits for the producers of the buffer; the function may
 * synchronise it), and return 0; any other value fails the step (CMP_ERR_INVALID, cmp_last_error names the message size). */
typedef int (*cmp_exchange_fn)(void* user, void* dev_f32, int64_t count, void* hip_stream);
int cmp_dp_init_exchange(cmp_ctx* ctx, int rank, int nranks, cmp_exchange_fn fn, void* user);
/* measurement aid: `wgs` workgroups that no persistent GEMM workgroup can share a CU with spin for `usec` microseconds on the
 * communication stream -- a stand-in for a concurrent RCCL kernel on a 1-GPU box (tools/ab_sched.sh) */
int cmp_dp_test_hog(cmp_ctx* ctx, int wgs, int usec);
/* While a communicator exists the persistent GEMM kernels launch at most `cus` workgroups (0 = all 256 CUs), leaving the
 * rest of the chip to the RCCL kernels of the overlapped gradient all-reduce.  Also settable with COMPOSER_DP_GEMM_CUS. */
int cmp_dp_set_gemm_cus(cmp_ctx* ctx, int cus);
/* Dropout masks are drawn from seed ^ mix32(rank) so that the replicas of a data-parallel job draw independent masks (the
 * reference is single-device).  cmp_dp_init sets the rank of the communicator; this call overrides it (tests, or a
 * launcher that shards without RCCL).  Parameter initialisation does NOT depend on it: replicas start identical. */
int cmp_dp_set_mask_rank(cmp_ctx* ctx, int rank);
/* Telemetry of the overlapped gradient exchange ("report exposed (non-overlapped) comm time per step").  Inside a
 * train step every gradient bucket is all-reduced and Adam-updated on the communication stream while the backward pass goes
 * on; at the end of the step the compute stream waits for that stream between two timed HIP events.  This call returns, over
 * the steps since the last reset: their count, the SUM of those waits in milliseconds (0 per step = everything was hidden
 * behind the backward pass), and the last step's bytes and number of ncclAllReduce calls (L+2 gradient buckets + the 3-float
 * metrics message).  All zeros without a communicator.  Synchronises with the steps it reports. */
int cmp_dp_stats(cmp_model* m, int reset, int64_t* steps, double* exposed_ms, int64_t* bytes_per_step, int* msgs_per_step);
/* ncclGetVersion of the RCCL this process is bound to (major * 10000 + minor * 100 + patch); needs no communicator. */
int cmp_dp_rccl_version(int* version);
/* The gradient exchange of one train step ALONE: the step's message pattern (3-float metrics message, then the L + 2 gradient buckets
 * in backward order, each its own range of the gradient buffer) issued back to back on the communication stream, `reps` times between
 * two HIP events after one untimed repetition; nothing runs on the compute stream.  ms = the reps' total; bytes / messages per
 * repetition.  The gradient buffer is left zeroed.  Every rank must call it with the same reps (bench.py --allreduce-only). */
int cmp_dp_allreduce_pattern(cmp_model* m, int reps, double* ms, int64_t* bytes_per_rep, int* msgs_per_rep);

/* ---- model: replaces models.Transformer(...) construction (cli.py:123-132) --------------------- */
/* Accepted configurations (anything else: CMP_ERR_INVALID with the reason in cmp_last_error): any vocabulary size; embedding_size
 * a multiple of 8, at most 2048 (LayerNorm keeps a row in registers); embedding_size divisible by heads
 * (transformer.py:255) with a head size of at most 128 -- 16 / 32 / 64 / 128 run natively, every other size runs zero-padded on
 * the next of those with the reference's shapes kept at this ABI (parameters, presents, past); at most 63 blocks; fp32 or bf16. */
int cmp_model_create(cmp_ctx* ctx, const cmp_model_cfg* cfg, cmp_model** out);
int cmp_model_destroy(cmp_model* m);

/* parameter / optimizer-state surface == checkpoint surface (transformer.py:890, models/__init__.py:75-80).
 * Names: "wte/weight" [V,E], "wpe/embeddings" [W,E], "decoder_blocks/<i>/{ln_1,ln_2}/{gamma,beta}" [E],
 * ".../attn/{c_attn,c_proj}/{weight,bias}", ".../mlp/{c_fc,c_proj}/{weight,bias}" (bias is [1,N],
 * transformer.py:190), "ln_f/{gamma,beta}".  kind: 0 value, 1 Adam m, 2 Adam v, 3 last gradient. */
int cmp_param_count(cmp_model* m, int* n);
int cmp_param_info(cmp_model* m, int i, const char** name, int* rank, int64_t shape[4], int64_t* numel);
int cmp_param_get(cmp_model* m, const char* name, int kind, float* host, int64_t numel);
int cmp_param_set(cmp_model* m, const char* name, int kind, const float* host, int64_t numel);
int cmp_adam_iter_get(cmp_model* m, int64_t* iterations);            /* Keras optimizer.iterations */
int cmp_adam_iter_set(cmp_model* m, int64_t iterations);

/* ---- training: one iteration of the loop body at transformer.py:914-930 ------------------------
 * x, y: host int32 [B,T].  Forward (training=True) + sparse-CE + backward + (DP all-reduce) + Adam.
 * Ids outside [0, V) fail the call before anything is enqueued; in a data-parallel job that is fatal for the whole job (the
 * peer ranks are already waiting in the all-reduce): validate the dataset up front, abort every rank on an error.
 * loss/acc (host, may be NULL) are the batch mean loss and accuracy -- of this rank's shard, or, once cmp_dp_init has
 * run, the mean over all ranks (one 3-float all-reduce per step; the reference logs one loss per step, :929-939). */
int cmp_train_step(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, float lr,
                   float* loss, float* acc);
/* Device-resident variant: x_dev/y_dev int32 [B,T] already in HBM; loss/acc are fetched later with
 * cmp_train_metrics (no host sync in the step itself). */
int cmp_train_step_dev(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, float lr);
int cmp_train_metrics(cmp_model* m, float* loss, float* acc);        /* syncs; last step's values */
/* Diagnostic (bench.py `default_cfg.launches`): the kernel launches (and memset / copy nodes) ONE train step of this shape enqueues,
 * counted on a stream capture of the step that is then dropped -- nothing executes, no state changes.  Single-process models only. */
int cmp_train_step_launches(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, int* kernels, int* others);
/* The same capture, instantiated and replayed `replay_reps` times back to back: milliseconds per replay (measurement only -- every
 * replay repeats the captured step's dropout masks and Adam iteration, the model is not a training run afterwards).  What a hipGraph of
 * the whole step would cost against stream launches (tools/default_config_graph_probe.py). */
int cmp_train_step_graph_probe(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, int* kernels, int* others,
                               int replay_reps, float* replay_ms);
/* Pipelined host-buffer step for the train loop (transformer.py:914-946): x/y are copied to pinned staging and uploaded on a
 * copy stream while earlier steps compute; returns at once with a ticket.  cmp_train_metrics_wait blocks until THAT step has
 * finished and returns its loss/accuracy.  At most 3 steps are in flight (a 4th submit waits for the oldest). */
int cmp_train_step_async(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, float lr, int64_t* ticket);
int cmp_train_metrics_wait(cmp_model* m, int64_t ticket, float* loss, float* acc);
/* forward+backward only (no all-reduce, no Adam): gradients readable with cmp_param_get(kind=3) */
int cmp_loss_and_grads(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T,
                       float* loss, float* acc);

/* ---- evaluation: model.evaluate (cli.py:613) --------------------------------------------------- */
int cmp_eval_step(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T,
                  double* loss_sum, int64_t* correct, int64_t* count);

/* ---- inference forward: Transformer.call(inputs, training=False) (transformer.py:696-833) -------
 * logits_out: host fp32 [B,T,V]. */
int cmp_forward_logits(cmp_model* m, const int32_t* x, int B, int T, float* logits_out);
/* The general form, Transformer.call(inputs, past=presents, training=...) (transformer.py:696-833):
 *   past_len > 0: `past` points to L host tensors fp32 [2, B, H, past_len, D] (an earlier call's presents, :735-765); the T
 *   ids of x are the tokens at positions past_len .. past_len+T-1 (the reference passes the last one, :735-737), their keys
 *   and values are appended to `past` (:423-426) and logits_out is [B,T,V]; cmp_present_get then takes T' = past_len + T.
 *   training != 0: dropout active (transformer.py:916-917 calls self(x, training=True)), masks from the model's seed and
 *   optimizer iteration like a train step's; with past, the attention-probability mask of a new token is its row of the
 *   mask over all past_len + T positions. */
int cmp_forward(cmp_model* m, const int32_t* x, int B, int T, int past_len, const float* const* past, int training,
                float* logits_out);
/* ... with Transformer.call's position_ids, token_type_ids (transformer.py:770-773, 786-793: host int32 [B*T] each, or null) and
 * attention_mask (:774-779, 356-358: host int32 [B*(past_len+T)], 1 = attend, 0 = masked; or null).
 *   position_ids: the wpe row of every token (default past_len + t); token_type_ids: a second wte row added to every token's
 *   embedding; attention_mask: (1 - mask) * -1e4 is added to the scaled, causally masked scores of every query of the batch
 *   row, in every layer and head.  Forward passes only (the reference's train loop passes none of them, :916-917).
 *   attention_weights_out (Transformer(..., output_attention_weights=True), :360-369, 808-809, 827-831): null, or L host
 *   tensors fp32 [B, H, T, past_len + T] that receive every layer's attention probabilities after their dropout. */
int cmp_forward_ex(cmp_model* m, const int32_t* x, int B, int T, int past_len, const float* const* past, int training,
                   const int32_t* position_ids, const int32_t* token_type_ids, const int32_t* attention_mask,
                   float* const* attention_weights_out, float* logits_out);
/* presents[layer] of the LAST forward pass (Transformer.call's second result, transformer.py:797-806, 820-821):
 * host fp32 [2, B, H, T, D] = stack([key, value]) after split_heads.  B, T must be that pass's shape (T = past + new). */
int cmp_present_get(cmp_model* m, int layer, int B, int T, float* host_out);
/* The activations a `presents` is read from live until the NEXT forward pass of the model (any of: cmp_forward*, a train or
 * eval step, cmp_decode_begin's prefill).  cmp_forward_generation returns the id of the pass just run; cmp_present_get_at
 * refuses (CMP_ERR_INVALID) when a later pass has replaced it -- the reference returns real tensors (transformer.py:820-821),
 * so a stale read must be an error, never another pass's keys/values. */
int cmp_forward_generation(cmp_model* m, int64_t* generation);
int cmp_present_get_at(cmp_model* m, int layer, int B, int T, int64_t generation, float* host_out);
/* all_hidden_states[index] of the forward pass `generation` (Transformer(..., output_hidden_states=True), transformer.py:610-614,
 * 800-816, 824-825): host fp32 [B, T, E] -- the input of decoder block `index` for index < L (index 0 = token + position
 * embedding after the embedding dropout), the ln_f output for index = L.  T = the pass's NEW positions (1 with `past`). */
int cmp_hidden_get_at(cmp_model* m, int index, int B, int T, int64_t generation, float* host_out);

/* ---- decode: the loop of cli.py:659-676 -------------------------------------------------------
 * temperature <= 0 => argmax with lowest-index tie-break (the tau->0 limit; cli.py:671 divides). */
int cmp_decode_begin(cmp_model* m, const int32_t* prompt, int P, int mode, float temperature, uint64_t seed);
int cmp_decode_steps(cmp_model* m, int n, int32_t* ids_out);
/* The sampler of the decode chain on its own (dev pointers): n independent draws from ONE logits row [V] with draw counters
 * counter0 .. counter0+n-1 -> ids_out[n].  tf.random.categorical(logits / temperature), cli.py:671-673. */
int cmp_k_sample(void* stream, const float* logits, int V, float temperature, uint64_t seed, uint32_t counter0, int n,
                 int32_t* ids_out);

/* ---- live kernel timing (bench.py roofline): HIP events around every launch of ONE kernel class on the
 * stream it is launched on.  cls: 0 gemm forward (A[M,K].B[K,N]), 1 gemm dgrad (B stored [N,K]), 2 gemm wgrad
 * (A stored [K,M]), 3 attention forward, 4 attention dQ, 5 attention dK/dV, 6 layernorm fwd, 7 adam.
 * cmp_prof_end syncs the device and returns summed milliseconds, launch count and summed algorithmic work
 * (flops for 0-5, bytes for 6-7). */
int cmp_prof_begin(int cls);
int cmp_prof_end(double* total_ms, int64_t* launches, double* work);
/* the same, and the summed ALGORITHMIC HBM bytes of those launches (every operand read once, every result written once):
 * what bench.py's per-class `algorithmic_bytes` is, next to the PMC-measured `traffic`.  Class 8 = layernorm backward.
 * Class 9 = the decoder-block stack of a forward pass as ONE span (work = L * (24 E^2 + 2 E T) flops per token). */
int cmp_prof_end2(double* total_ms, int64_t* launches, double* work, double* bytes);
/* between begin and end: stop / continue recording (what has been recorded stays); bench.py times a subset of its timed steps */
int cmp_prof_pause(void);
int cmp_prof_resume(void);

/* ---- kernel-level entry points (dev pointers; dtype = cmp_dtype of activations) ----------------
 * Used by tests/ and bench.py to check and time single kernels against the oracle/roofline. */
int cmp_k_embed_fwd(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out,
                    int B, int T, int E, int pos0, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream);
int cmp_k_embed_bwd(void* stream, const int32_t* ids, const void* dh, float* dwte, float* dwpe,
                    int B, int T, int E, int pos0, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream);
/* the same through the sorted form the model uses for batches of 4096 tokens or more (tokens counting-sorted by id, rows of one id
 * summed before they are added: E f32 atomics per 32 rows instead of one per element); V = vocabulary size (<= 8192) */
int cmp_k_embed_bwd_v(void* stream, const int32_t* ids, const void* dh, float* dwte, float* dwpe,
                      int B, int T, int E, int pos0, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream, int V);
int cmp_k_layernorm_fwd(void* stream, const void* x, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, int rows, int E, float eps, int dtype);
/* dx = resid(optional) + LN_bwd(dy); dgamma/dbeta fp32 [E] are ACCUMULATED into; ws >= cmp_k_layernorm_bwd_ws bytes */
int cmp_k_layernorm_bwd(void* stream, const void* dy, const void* x, const float* gamma, const float* mean,
                        const float* rstd, const void* resid, void* dx, float* dgamma, float* dbeta,
                        void* ws, int rows, int E, int dtype);
int64_t cmp_k_layernorm_bwd_ws(int rows, int E);
/* same, plus the prologue of the projection that consumes dx as the gradient of `x + dropout(proj)`: colsum[E] +=
 * column sums of dropout_grad(dx) (that projection's bias gradient) and, when p_drop > 0, dmask = dx * mask/(1-p). */
int cmp_k_layernorm_bwd_fused(void* stream, const void* dy, const void* x, const float* gamma, const float* mean,
                              const float* rstd, const void* resid, void* dx, float* dgamma, float* dbeta,
                              void* ws, int rows, int E, int dtype, void* dmask, float* colsum, float p_drop,
                              uint64_t seed, uint32_t rng_stream);
/* C[M,N] = epilogue(A.B): ta=0: A is [M,K] (lda); ta=1: A stored [K,M].  tb=0: B stored [K,N]; tb=1: B stored [N,K].
 * epilogue: +bias[N] (fp32, may be NULL); act: 0 none, 1 gelu (pre-activation stored to aux if aux!=NULL),
 * 2 multiply by gelu'(aux[m,n]); dropout (p>0) then +resid[m,n] (may be NULL).  out_fp32: C is fp32 regardless of
 * dtype.  splitk>1: fp32 atomic accumulation into C (C must be pre-zeroed or hold the value to add to).
 * dropout index = row*ldc + col (ldc == N for the model's outputs). */
int cmp_k_gemm(void* stream, int dtype, int ta, int tb, int M, int N, int K,
               const void* A, int lda, const void* Bm, int ldb, void* C, int ldc,
               const float* bias, int act, void* aux, int ldaux, const void* resid, int ldr,
               int out_fp32, int splitk, float p_drop, uint64_t seed, uint32_t rng_stream, int flags);
/* flags: CMP_GEMM_KPAD_ZERO -- K-contiguous bf16 operands are stored with their rows zero-padded to a multiple of
 * 64 elements (lets a ragged K, e.g. the vocabulary, use the direct-to-LDS path); CMP_GEMM_GENERIC forces the
 * register-staged generic kernel (tests). */
#define CMP_GEMM_KPAD_ZERO 1
#define CMP_GEMM_GENERIC 2
#define CMP_GEMM_TILE128 4      /* tests/bench: force the 128x128 direct-to-LDS kernel */
#define CMP_GEMM_TILE256 8      /* tests/bench: force the persistent 256x256 2-stage kernel */
#define CMP_GEMM_P4 16          /* tests/bench: force the persistent deep-pipeline (BK=32) kernel */
#define CMP_GEMM_P4_128 32      /* with CMP_GEMM_P4: 128x256 tile / 4 waves / 3 stages / 2 workgroups per CU instead of 256x256 / 8 waves / 4 stages */
/* ONE launch for up to 8 split-K weight gradients that contract over the same K rows -- the four Conv1D weight gradients of a
 * decoder block (tf.GradientTape through transformer.py:205-209) contract over the tokens: C_i[M_i, N_i] (fp32, ACCUMULATED into
 * with f32 atomics) += A_i^T . B_i, A_i bf16 stored [K, M_i] (lda_i), B_i bf16 stored [K, N_i] (ldb_i).  Requirements
 * (CMP_ERR_INVALID otherwise): K a multiple of 32, leading dimensions multiples of 8, 16-byte aligned operands, every operand
 * below 2 GiB.  The model's backward pass uses this launch in bf16 mode. */
int cmp_k_wgrad_group(void* stream, int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                      float* const* C, const int* ldc, const int* M, const int* N, int K);
/* ---- LayerNorm folded into the block's GEMM epilogues (the fused block path of the bf16 train / inference forward) -----------
 * Replaces the stand-alone tf.keras LayerNormalization calls of DecoderBlock.call (transformer.py:583-584 ln_1, :591 ln_2; layers
 * built at :551,563) for large batches: no LayerNorm kernel runs in the forward pass.  Row statistics travel as PARTIALS, fp32
 * [rows][E/256][2] = (mean, M2 = sum of squared deviations from that mean) of every 256-column segment of a row, merged with
 * Chan's update by the consumer (never a sum of squares).
 *   cmp_k_embed_fwd_stats: cmp_k_embed_fwd (bf16) that also leaves the partials of the rows it wrote (SharedTokenEmbedding +
 *     position embedding, transformer.py:786-794, feeding block 0's ln_1).  E a multiple of 256.
 *   cmp_k_ln_fold_prep: weight side of  LN(x).W + b = rstd*(x.(gamma o W)) - rstd*mean*colsum(gamma o W) + (beta.W + b):
 *     W fp32 [E,N] (Conv1D weight, transformer.py:184-189), bias [N], gamma / beta [E] -> WT bf16 [N,E] = (gamma o W)^T,
 *     cs[N] = column sums of the ROUNDED WT, bias_out[N] = bias + beta.W.  E, N multiples of 32.
 *   cmp_gemm_ln_next: one-shot, the NEXT cmp_k_gemm of this thread carries a LayerNorm epilogue -- in_part (+ np = E/256 in
 *     2..3, eps): statistics of the rows of A (fold: cs given; the GEMM's B operand is WT and its bias is bias_out) or of the
 *     rows of `resid` (gamma, beta given: the residual operand becomes LN(resid), transformer.py:587); out_part: the partials of
 *     the OUTPUT rows are written (residual epilogues).  bf16, ta=0, tb=1, M and N multiples of 256, and a shape that reaches the
 *     persistent 256x256 kernel (CMP_GEMM_TILE256 forces it); anything else fails with CMP_ERR_INVALID.  With out_fp32 the fold also
 *     takes a ragged N (ln_f folded into the tied-logits matmul, transformer.py:811, 818: N = vocabulary size; cs and bias must then
 *     hold N rounded up to 256 entries, zero beyond N).
 *   cmp_k_layernorm_bwd_parts: cmp_k_layernorm_bwd_fused (bf16) with the statistics taken from partials and the LayerNorm OUTPUT
 *     yout = xhat*gamma + beta written beside dx (the weight-gradient GEMM of the consuming Conv1D is its only reader); dmask
 *     (when given) is written whatever p_drop is. */
int cmp_k_embed_fwd_stats(void* stream, const int32_t* ids, const float* wte, const float* wpe, void* out, float* part,
                          int B, int T, int E, int pos0, float p_drop, uint64_t seed, uint32_t rng_stream);
int cmp_k_ln_fold_prep(void* stream, const float* W, const float* bias, const float* gamma, const float* beta, void* WT,
                       float* cs, float* bias_out, int E, int N);
int cmp_gemm_ln_next(const float* in_part, int np, float eps, const float* cs, const float* gamma, const float* beta,
                     float* out_part);
/* Round 6, the backward pass of the LayerNorm-fused block path (weight gradients on the RAW LayerNorm input rows, no LayerNorm output
 * ever written).  One-shot arming calls like cmp_gemm_ln_next, consumed by this thread's next cmp_k_gemm / cmp_k_attn_bwd:
 *   cmp_gemm_ln_scale_next: the rows' rstd (merged from in_part) is a FACTOR -- with act = 2 (gelu' epilogue) the launch stores
 *     C = rstd o (acc * gelu'(aux)) and still sends the column sums of the UNSCALED product to cmp_gemm_colsum_next's vector; with
 *     a residual operand it stores C = acc + rstd o resid.
 *   cmp_k_ln_stats_merge: (mean, rstd) per row from the rows' partial statistics [rows][np][2] (the model merges all its sites in one launch).
 *   cmp_attn_bwd_ln_next: cmp_k_attn_bwd stores rstd[token] o [dQ | dK | dV].
 *   cmp_k_layernorm_bwd_prescaled: the LayerNorm backward (mean / rstd arrays) on a dy that holds rstd o (the gradient).
 *   cmp_k_wgrad_ln_fix: G[k, j] = gamma[k] * (G[k, j] - mean over k' of G[k', j]) + beta[k] * colsum[j] -- turns R = (raw rows)^T .
 *     (rstd o D) into LN(raw rows)^T . D: the mean term mean^T . (rstd o D) IS the column mean of R, a row's mean being the mean of
 *     that row as stored. */
int cmp_gemm_ln_scale_next(const float* in_part, int np, float eps);
int cmp_k_ln_stats_merge(void* stream, const float* part, int np, float eps, float* mean, float* rstd, int rows);
int cmp_attn_bwd_ln_next(const float* rstd);
int cmp_k_layernorm_bwd_prescaled(void* stream, const void* dy_scaled, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, const void* resid, void* dx, float* dgamma, float* dbeta, void* ws,
                                  int rows, int E, void* dmask, float* colsum, float p_drop, uint64_t seed, uint32_t rng_stream);
int cmp_k_wgrad_ln_fix(void* stream, float* G, int rows, int cols, const float* gamma, const float* beta, const float* colsum);
int cmp_k_layernorm_bwd_parts(void* stream, const void* dy, const void* x, const float* gamma, const float* beta,
                              const float* part, float eps, const void* resid, void* dx, void* yout, float* dgamma,
                              float* dbeta, void* ws, int rows, int E, void* dmask, float* colsum, float p_drop,
                              uint64_t seed, uint32_t rng_stream);
/* Diagnostics of the model's last passes: fused = 1 when the last forward pass took the LayerNorm-fused block path;
 * wgrad_table_builds = item tables of the grouped weight-gradient launches built so far (a steady train loop builds one per
 * decoder block, once). */
int cmp_model_path_info(cmp_model* m, int* fused, int64_t* wgrad_table_builds);
/* Registers a device workspace for split-K reductions: with it, split-K launches write per-split fp32 partial tiles and
 * fold them in a fixed order (reproducible, no float atomics); without it (or if too small: splitk*M*N*4 bytes) they
 * accumulate with f32 atomics.  flags & 128 forces the atomic path. */
int cmp_gemm_set_workspace(void* ws_dev, int64_t bytes);
/* One-shot: the next cmp_k_gemm (plain output in the compute dtype) also adds the column sums of its output to
 * out[0..N) -- the bias gradient that goes with an input-gradient GEMM (transformer.py:916-920 via tf.GradientTape).
 * Fused into the GEMM epilogue where possible, otherwise a cmp_k_colsum pass after it. */
int cmp_gemm_colsum_next(float* out);
/* diagnostic only: a device buffer of 500 uint64 receives (id, s_memtime) pairs from one workgroup of the next
 * deep-pipeline GEMM launches (tools/gemm_timeline.py); pass NULL to switch it off. */
int cmp_gemm_set_stamps(void* dev_buf);
int cmp_k_colsum(void* stream, const void* X, int ldx, float* out, int rows, int cols, int dtype);
/* causal attention on qkv [B,T,3E] (head-merged, transformer.py:417): o [B,T,E], lse fp32 [B,H,T] */
int cmp_k_attn_fwd(void* stream, const void* qkv, void* o, float* lse, int B, int T, int H, int D,
                   int scale, int dtype, float p_drop, uint64_t seed, uint32_t rng_stream);
/* One-shot: the next cmp_k_attn_bwd also adds the column sums of [dQ | dK | dV] to out[0..3E): the c_attn bias gradient
 * (Conv1D bias under tf.GradientTape, transformer.py:205-209, 916-920), taken from the f32 accumulators. */
int cmp_attn_bwd_bias_next(float* out);
int cmp_k_attn_bwd(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse,
                   float* delta_ws, void* dqkv, int B, int T, int H, int D, int scale, int dtype,
                   float p_drop, uint64_t seed, uint32_t rng_stream);
/* fused softmax cross-entropy forward+backward: logits fp32 [rows, ldz]; dlogits (dtype) [rows, ldz] with
 * zeroed padding; row_loss fp32 [rows]; row_correct int32 [rows]; inv_n = 1/(B*T) */
int cmp_k_softmax_xent(void* stream, const float* logits, int ldz, const int32_t* y, void* dlogits,
                       float* row_loss, int32_t* row_correct, int rows, int V, float inv_n, int dtype);
/* Keras Adam (transformer.py:887,921): eps outside the bias correction; step = optimizer.iterations+1 */
int cmp_k_adam(void* stream, float* p, const float* g, float* m, float* v, void* shadow_bf16,
               int64_t n, float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale);

#ifdef __cplusplus
}
#endif
#endif /* COMPOSER_HIP_H */
