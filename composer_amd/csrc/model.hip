// model.hip -- cmp_ctx / cmp_model: device-resident state of the Transformer hot path and the native
// train / eval / forward drivers that chain the HIP kernels on one stream (no Python between kernels).
//
// Reference path: Transformer.call (transformer.py:696-833), DecoderBlock.call (:574-597),
// the train-loop body (:914-930) and Keras Adam (:887,921).
//
// HBM layout
//   params / grads / adam_m / adam_v : ONE flat fp32 buffer each, tensors in checkpoint order
//     (wte, wpe, block 0..L-1, ln_f), every tensor offset a multiple of 8 elements; in bf16 mode a flat bf16
//     shadow with identical offsets is rewritten by the Adam kernel.  A decoder block is one contiguous
//     range => one RCCL all-reduce bucket per block, issued on the side stream as soon as that block's
//     backward has been enqueued.
//   activations: row-major [B*T, width] in the activation dtype; saved per layer for backward:
//     x_in, u=LN1(x_in), qkv, att, r=u+proj, n=LN2(r), fc (pre-GELU), g=gelu(fc); LN stats and LSE in fp32.
#include "model.h"
#include <stdarg.h>
#include <dlfcn.h>

static thread_local char g_err[1024] = "";
void cmp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cmp_last_error(void) { return g_err; }
extern "C" int cmp_version(void) { return 1; }
extern "C" int cmp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// -------------------------------------------------------------------------------------------------
// live kernel timing
// -------------------------------------------------------------------------------------------------
// (benchmark instrumentation: one timed class at a time, armed and read by the thread that runs the steps)
thread_local int g_prof_cls = -1;
static thread_local std::vector<hipEvent_t> g_prof_ev;     // start/stop pairs
static thread_local size_t g_prof_used = 0;
static thread_local double g_prof_work = 0.0, g_prof_bytes = 0.0;
void prof_start(int, hipStream_t s) {
    if (g_prof_used + 2 > g_prof_ev.size()) {
        for (int i = 0; i < 256; i++) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            g_prof_ev.push_back(e);
        }
    }
    (void)hipEventRecord(g_prof_ev[g_prof_used], s);
}
void prof_stop(int, hipStream_t s, double work, double bytes) {
    if (g_prof_used + 2 > g_prof_ev.size()) return;
    (void)hipEventRecord(g_prof_ev[g_prof_used + 1], s);
    g_prof_used += 2;
    g_prof_work += work;
    g_prof_bytes += bytes;
}
extern "C" int cmp_prof_begin(int cls) {
    g_prof_cls = cls;
    g_prof_used = 0;
    g_prof_work = 0.0;
    g_prof_bytes = 0.0;
    return CMP_OK;
}
// between cmp_prof_begin and cmp_prof_end: stop / continue recording without touching what has been recorded (bench.py times
// every n-th step of its timed region: two events per launch are not free)
static thread_local int g_prof_armed = -1;
extern "C" int cmp_prof_pause(void) {
    if (g_prof_cls >= 0) { g_prof_armed = g_prof_cls; g_prof_cls = -1; }
    return CMP_OK;
}
extern "C" int cmp_prof_resume(void) {
    if (g_prof_armed >= 0) { g_prof_cls = g_prof_armed; g_prof_armed = -1; }
    return CMP_OK;
}
extern "C" int cmp_prof_end2(double* total_ms, int64_t* launches, double* work, double* bytes);
extern "C" int cmp_prof_end(double* total_ms, int64_t* launches, double* work) { return cmp_prof_end2(total_ms, launches, work, nullptr); }
extern "C" int cmp_prof_end2(double* total_ms, int64_t* launches, double* work, double* bytes) {
    g_prof_cls = -1;
    g_prof_armed = -1;
    if (g_prof_used >= 2) HIP_CHECK(hipEventSynchronize(g_prof_ev[g_prof_used - 1]));       // the last recorded stop event (same stream order)
    double t = 0.0;
    for (size_t i = 0; i + 1 < g_prof_used; i += 2) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, g_prof_ev[i], g_prof_ev[i + 1]));
        t += ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = (int64_t)(g_prof_used / 2);
    if (work) *work = g_prof_work;
    if (bytes) *bytes = g_prof_bytes;
    g_prof_used = 0;
    return CMP_OK;
}

// -------------------------------------------------------------------------------------------------
// roctx ranges (COMPOSER_ROCTX=1): forward / loss / backward (per block) / adam show up as named ranges in
// `rocprofv3 --marker-trace --kernel-trace`.  libroctx64 is looked up at run time; absent or disabled = no-ops.
// -------------------------------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("COMPOSER_ROCTX");
        if (!e || e[0] != '1') return;
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr;
    }
};
Roctx& roctx() { static Roctx r; return r; }
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
};
}  // namespace

// -------------------------------------------------------------------------------------------------
// context
// -------------------------------------------------------------------------------------------------
extern "C" int cmp_ctx_create(int device, cmp_ctx** out) {
    CMP_REQUIRE(out != nullptr, "ctx_create: out is null");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    CMP_REQUIRE(device >= 0 && device < n, "ctx_create: device %d not present (%d visible)", device, n);
    HIP_CHECK(hipSetDevice(device));
    cmp_ctx* c = new cmp_ctx();
    c->device = device;
    // the RCCL stream gets the highest priority the device offers: its kernels are short and latency-bound, and every
    // compute kernel is a persistent launch that would otherwise keep them waiting for a full kernel
    int prio_lo = 0, prio_hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_hi));
    HIP_CHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    {
        const char* e = getenv("COMPOSER_DP_GEMM_CUS");
        if (e && atoi(e) > 0) c->gemm_max_wgs = atoi(e);
    }
    *out = c;
    return CMP_OK;
}
extern "C" int cmp_ctx_destroy(cmp_ctx* c) {
    if (!c) return CMP_OK;
    hipSetDevice(c->device);
    // this context's own streams only: a device-wide (or legacy-stream) wait is an error while ANOTHER thread's context captures a graph
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->comm_stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    if (c->comm) ncclCommDestroy(c->comm);
    sched_ws_free(&c->gemm_sched);
    hipStreamDestroy(c->stream);
    hipStreamDestroy(c->comm_stream);
    hipStreamDestroy(c->copy_stream);
    delete c;
    return CMP_OK;
}
extern "C" int cmp_sync(cmp_ctx* c) {
    CMP_REQUIRE(c, "sync: ctx is null");
    HIP_CHECK(hipStreamSynchronize(c->copy_stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipStreamSynchronize(c->comm_stream));
    return CMP_OK;
}
extern "C" void* cmp_ctx_stream(cmp_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int cmp_dp_unique_id(void* id128) {
    CMP_REQUIRE(id128, "dp_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id128, &id, 128);
    return CMP_OK;
}
extern "C" int cmp_dp_init(cmp_ctx* c, int rank, int nranks, const void* id128) {
    CMP_REQUIRE(c && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "dp_init: bad arguments");
    CMP_REQUIRE(!c->dp_on(), "dp_init: communicator already initialised");
    HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    NCCL_CHECK(ncclCommInitRank(&c->comm, nranks, id, rank));
    c->rank = rank;
    c->nranks = nranks;
    c->seed_mix = mix32((uint32_t)rank);      // seed ^ mix32(rank): rank 0 keeps the model seed, the others draw their own masks
    return CMP_OK;
}
extern "C" int cmp_dp_init_exchange(cmp_ctx* c, int rank, int nranks, int (*fn)(void*, void*, int64_t, void*), void* user) {
    CMP_REQUIRE(c && fn && nranks >= 1 && rank >= 0 && rank < nranks, "dp_init_exchange: bad arguments");
    CMP_REQUIRE(!c->dp_on(), "dp_init_exchange: communicator already initialised");
    c->xfn = fn;
    c->xuser = user;
    c->rank = rank;
    c->nranks = nranks;
    c->seed_mix = mix32((uint32_t)rank);
    return CMP_OK;
}
// sum of p[0, n) over the ranks, in place, ordered on the communication stream: RCCL, or the caller's exchange function
static int dp_allreduce(cmp_ctx* c, float* p, size_t n) {
    if (c->xfn) {
        const int rc = c->xfn(c->xuser, p, (int64_t)n, (void*)c->comm_stream);
        if (rc != 0) {
            cmp_set_error("data-parallel exchange function failed with %d on %zu floats", rc, n);
            return CMP_ERR_INVALID;
        }
        return CMP_OK;
    }
    NCCL_CHECK(ncclAllReduce(p, p, n, ncclFloat, ncclSum, c->comm, c->comm_stream));
    return CMP_OK;
}
extern "C" int cmp_dp_set_mask_rank(cmp_ctx* c, int rank) {
    CMP_REQUIRE(c && rank >= 0, "dp_set_mask_rank: bad arguments");
    c->seed_mix = mix32((uint32_t)rank);
    return CMP_OK;
}
extern "C" int cmp_dp_set_gemm_cus(cmp_ctx* c, int cus) {
    CMP_REQUIRE(c && cus >= 0 && cus <= 256, "dp_set_gemm_cus: 0 (all) .. 256");
    c->gemm_max_wgs = cus;
    return CMP_OK;
}
// Measurement aid for the 1-GPU box: what a concurrent RCCL kernel does to the compute stream, without a second GPU.  `wgs`
// workgroups of 1024 threads and 64 KiB of LDS each (a persistent GEMM workgroup cannot share their CU) spin for `usec`
// microseconds on the communication stream (highest priority) -- tools/ab_sched.sh times train steps beside it.
__global__ __launch_bounds__(1024) void cu_hog_kernel(unsigned long long ticks, int* sink) {
    __shared__ int big[16384];
    big[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && big[(threadIdx.x * 7) & 16383] == -1) *sink = 1;      // (keeps the LDS array alive; sink is null)
}
extern "C" int cmp_dp_test_hog(cmp_ctx* c, int wgs, int usec) {
    CMP_REQUIRE(c && wgs > 0 && wgs <= 256 && usec > 0 && usec <= 1000000, "dp_test_hog: bad arguments");
    HIP_CHECK(hipSetDevice(c->device));
    cu_hog_kernel<<<wgs, 1024, 0, c->comm_stream>>>((unsigned long long)usec * 100ull, nullptr);
    KERNEL_CHECK();
    return CMP_OK;
}
extern "C" int cmp_dp_allreduce_test(cmp_ctx* c, float* host_inout, int n) {
    CMP_REQUIRE(c && c->dp_on(), "dp_allreduce_test: communicator not initialised");
    float* d = nullptr;
    HIP_CHECK(hipMalloc(&d, (size_t)n * 4));
    HIP_CHECK(hipMemcpyAsync(d, host_inout, (size_t)n * 4, hipMemcpyHostToDevice, c->comm_stream));
    if (const int rc = dp_allreduce(c, d, (size_t)n)) {
        (void)hipStreamSynchronize(c->comm_stream);
        (void)hipFree(d);
        return rc;
    }
    HIP_CHECK(hipMemcpyAsync(host_inout, d, (size_t)n * 4, hipMemcpyDeviceToHost, c->comm_stream));
    HIP_CHECK(hipStreamSynchronize(c->comm_stream));
    HIP_CHECK(hipFree(d));
    return CMP_OK;
}

// -------------------------------------------------------------------------------------------------
// model construction
// -------------------------------------------------------------------------------------------------
static int64_t add_param(cmp_model* m, const std::string& name, int rank, int64_t d0, int64_t d1, int pad = 0) {
    ParamInfo p;
    p.name = name;
    p.rank = rank;
    p.shape[0] = d0;
    p.shape[1] = d1;
    p.shape[2] = p.shape[3] = 1;
    p.numel = d0 * (rank > 1 ? d1 : 1);
    p.pad = (m->Dl == m->D) ? 0 : pad;
    p.store = p.numel;
    if (p.pad == 1) p.store = d0 * 3 * m->Ea;          // [d0][3][H][D]
    if (p.pad == 2) p.store = (int64_t)m->Ea * d1;     // [H][D][d1]
    p.offset = m->total;
    m->total += (p.store + 7) / 8 * 8;
    m->index[name] = (int)m->params.size();
    m->params.push_back(p);
    return p.offset;
}


static int model_create_fill(cmp_model* m, cmp_ctx* ctx, const cmp_model_cfg* cfg, int V, int E, int W, int L, int H, int D, int Dl);
extern "C" int cmp_model_destroy(cmp_model* m);

extern "C" int cmp_model_create(cmp_ctx* ctx, const cmp_model_cfg* cfg, cmp_model** out) {
    CMP_REQUIRE(ctx && cfg && out, "model_create: null argument");
    const int V = cfg->vocab_size, E = cfg->embedding_size, W = cfg->window_size, L = cfg->layers, H = cfg->heads;
    CMP_REQUIRE(V > 0 && E > 0 && W > 0 && L > 0 && H > 0, "model_create: sizes must be positive");
    CMP_REQUIRE(E % H == 0, "model_create: embedding_size %d not divisible by heads %d (transformer.py:255)", E, H);
    CMP_REQUIRE(E % 8 == 0, "model_create: embedding_size %d must be a multiple of 8", E);
    CMP_REQUIRE(E <= 2048, "model_create: embedding_size %d unsupported (at most 2048: LayerNorm rows, per-token decode rows of 4E)", E);
    const int Dl = E / H;
    CMP_REQUIRE(Dl <= 128, "model_create: head size %d unsupported (at most 128)", Dl);
    const int D = Dl <= 16 ? 16 : (Dl <= 32 ? 32 : (Dl <= 64 ? 64 : 128));     // the attention kernels' head sizes; Dl < D: zero-padded columns
    CMP_REQUIRE(cfg->dtype == CMP_FP32 || cfg->dtype == CMP_BF16, "model_create: bad dtype %d", cfg->dtype);
    CMP_REQUIRE(L < 64, "model_create: at most 63 layers");
    HIP_CHECK(hipSetDevice(ctx->device));
    cmp_model* m = new cmp_model();
    m->ctx = ctx;
    {   // COMPOSER_LN_FUSED is read ONCE per model, here (tests switch it between models): 0 off, 2 / 3 training passes too
        const char* e = getenv("COMPOSER_LN_FUSED");
        m->ln_fused_mode = e ? atoi(e) : -1;
    }
    // a failure part-way (an allocation, an event) must not leave the buffers allocated so far behind
    const int rc = model_create_fill(m, ctx, cfg, V, E, W, L, H, D, Dl);
    if (rc != CMP_OK) { cmp_model_destroy(m); return rc; }
    *out = m;
    return CMP_OK;
}

static int model_create_fill(cmp_model* m, cmp_ctx* ctx, const cmp_model_cfg* cfg, int V, int E, int W, int L, int H, int D, int Dl) {
    m->cfg = *cfg;
    m->V = V; m->E = E; m->W = W; m->L = L; m->H = H; m->D = D; m->Dl = Dl; m->Ea = H * D;
    m->ldz = (V + 63) / 64 * 64;
    m->dtype = cfg->dtype;
    m->es = dtype_size(cfg->dtype);
    m->off_wte = add_param(m, "wte/weight", 2, V, E);
    m->off_wpe = add_param(m, "wpe/embeddings", 2, W, E);
    m->lo.resize(L);
    for (int i = 0; i < L; i++) {
        std::string p = "decoder_blocks/" + std::to_string(i) + "/";
        LayerOff& o = m->lo[i];
        o.begin = m->total;
        o.ln1_g = add_param(m, p + "ln_1/gamma", 1, E, 1);
        o.ln1_b = add_param(m, p + "ln_1/beta", 1, E, 1);
        o.attn_w = add_param(m, p + "attn/c_attn/weight", 2, E, 3 * E, 1);
        o.attn_b = add_param(m, p + "attn/c_attn/bias", 2, 1, 3 * E, 1);
        o.proj_w = add_param(m, p + "attn/c_proj/weight", 2, E, E, 2);
        o.proj_b = add_param(m, p + "attn/c_proj/bias", 2, 1, E);
        o.ln2_g = add_param(m, p + "ln_2/gamma", 1, E, 1);
        o.ln2_b = add_param(m, p + "ln_2/beta", 1, E, 1);
        o.fc_w = add_param(m, p + "mlp/c_fc/weight", 2, E, 4 * E);
        o.fc_b = add_param(m, p + "mlp/c_fc/bias", 2, 1, 4 * E);
        o.pr_w = add_param(m, p + "mlp/c_proj/weight", 2, 4 * E, E);
        o.pr_b = add_param(m, p + "mlp/c_proj/bias", 2, 1, E);
        o.end = m->total;
    }
    m->off_lnf_g = add_param(m, "ln_f/gamma", 1, E, 1);
    m->off_lnf_b = add_param(m, "ln_f/beta", 1, E, 1);
    size_t bytes = (size_t)m->total * 4;
    CHECK_RC(dev_alloc(m, &m->P, bytes));
    CHECK_RC(dev_alloc(m, &m->G, bytes));
    CHECK_RC(dev_alloc(m, &m->Am, bytes));
    CHECK_RC(dev_alloc(m, &m->Av, bytes));
    HIP_CHECK(hipMemsetAsync(m->P, 0, bytes, ctx->stream));
    HIP_CHECK(hipMemsetAsync(m->G, 0, bytes, ctx->stream));
    HIP_CHECK(hipMemsetAsync(m->Am, 0, bytes, ctx->stream));
    HIP_CHECK(hipMemsetAsync(m->Av, 0, bytes, ctx->stream));
    if (m->dtype == CMP_BF16) {
        CHECK_RC(dev_alloc(m, &m->S, (size_t)m->total * 2));
        HIP_CHECK(hipMemsetAsync(m->S, 0, (size_t)m->total * 2, ctx->stream));
        CHECK_RC(dev_alloc(m, &m->ST, (size_t)m->total * 2));
        HIP_CHECK(hipMemsetAsync(m->ST, 0, (size_t)m->total * 2, ctx->stream));
        std::vector<WDesc> wd;
        for (int i = 0; i < m->L; i++) {
            const LayerOff& o = m->lo[i];
            wd.push_back({o.attn_w, m->E, 3 * m->Ea});
            wd.push_back({o.proj_w, m->Ea, m->E});
            wd.push_back({o.fc_w, m->E, 4 * m->E});
            wd.push_back({o.pr_w, 4 * m->E, m->E});
        }
        CHECK_RC(dev_alloc(m, &m->wdesc, wd.size() * sizeof(WDesc)));
        HIP_CHECK(hipMemcpyAsync(m->wdesc, wd.data(), wd.size() * sizeof(WDesc), hipMemcpyHostToDevice, ctx->stream));
        // the LayerNorm-fused block path's tables (ln_fused_ok): the two c_proj weights keep plain transposes, c_attn / c_fc get the
        // gamma-scaled ones + fold vectors
        std::vector<WDesc> wp;
        std::vector<FoldDesc> fd;
        m->fold_stride = 2 * (int64_t)(3 * m->Ea + 4 * m->E);
        for (int i = 0; i < m->L; i++) {
            const LayerOff& o = m->lo[i];
            wp.push_back({o.proj_w, m->Ea, m->E});
            wp.push_back({o.pr_w, 4 * m->E, m->E});
            fd.push_back({o.attn_w, o.attn_b, o.ln1_g, o.ln1_b, i * m->fold_stride, m->E, 3 * m->Ea});
            fd.push_back({o.fc_w, o.fc_b, o.ln2_g, o.ln2_b, i * m->fold_stride + 6 * m->Ea, m->E, 4 * m->E});
        }
        CHECK_RC(dev_alloc(m, &m->wdesc_plain, wp.size() * sizeof(WDesc)));
        CHECK_RC(dev_alloc(m, &m->fdesc, fd.size() * sizeof(FoldDesc)));
        CHECK_RC(dev_alloc(m, &m->lnfold, (size_t)m->L * m->fold_stride * 4));
        m->lnf_npad = cdiv(m->V, 256) * 256;
        CHECK_RC(dev_alloc(m, &m->wte_lnf, (size_t)m->V * m->E * 2));
        CHECK_RC(dev_alloc(m, &m->lnf_fold, (size_t)2 * m->lnf_npad * 4));
        HIP_CHECK(hipMemcpyAsync(m->wdesc_plain, wp.data(), wp.size() * sizeof(WDesc), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(m->fdesc, fd.data(), fd.size() * sizeof(FoldDesc), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));      // wd, wp, fd are locals
    }
    CHECK_RC(dev_alloc(m, &m->metrics, sizeof(Metrics)));
    CHECK_RC(dev_alloc(m, &m->dp_metrics, 16));
    HIP_CHECK(hipHostMalloc((void**)&m->metrics_host, sizeof(Metrics), hipHostMallocDefault));
    memset(m->metrics_host, 0, sizeof(Metrics));
    m->bucket_ev.resize(L + 2);
    for (auto& e : m->bucket_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&m->comm_done, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&m->metrics_ev, hipEventDisableTiming));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return CMP_OK;
}

extern "C" int cmp_model_destroy(cmp_model* m) {
    if (!m) return CMP_OK;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    hipStreamSynchronize(m->ctx->comm_stream);
    if (m->ctx->copy_stream) hipStreamSynchronize(m->ctx->copy_stream);
    if (m->dec) decode_state_free(m->dec);
    for (void* p : m->allocs) if (p) hipFree(p);
    for (auto& wg : m->wgrad_groups) wgrad_group_free(&wg);
    wgrad_ws_free(&m->wgrad_ws);
    if (m->metrics_host) hipHostFree(m->metrics_host);
    if (m->stage_metrics) hipHostFree(m->stage_metrics);
    for (int i = 0; i < cmp_model::STAGES; i++) {
        if (m->stage_host[i]) hipHostFree(m->stage_host[i]);
        if (m->stage_uploaded[i]) hipEventDestroy(m->stage_uploaded[i]);
        if (m->stage_done[i]) hipEventDestroy(m->stage_done[i]);
    }
    for (auto& e : m->bucket_ev) hipEventDestroy(e);
    if (m->comm_done) hipEventDestroy(m->comm_done);
    if (m->metrics_ev) hipEventDestroy(m->metrics_ev);
    for (int i = 0; i < cmp_model::DP_RING; i++) {
        if (m->dp_wait_a[i]) hipEventDestroy(m->dp_wait_a[i]);
        if (m->dp_wait_b[i]) hipEventDestroy(m->dp_wait_b[i]);
    }
    delete m;
    (void)hipGetLastError();        // whatever a teardown call reported must not surface in this thread's next launch check
    return CMP_OK;
}

extern "C" int cmp_param_count(cmp_model* m, int* n) {
    CMP_REQUIRE(m && n, "param_count: null");
    *n = (int)m->params.size();
    return CMP_OK;
}
extern "C" int cmp_param_info(cmp_model* m, int i, const char** name, int* rank, int64_t shape[4], int64_t* numel) {
    CMP_REQUIRE(m && i >= 0 && i < (int)m->params.size(), "param_info: index %d out of range", i);
    const ParamInfo& p = m->params[i];
    if (name) *name = p.name.c_str();
    if (rank) *rank = p.rank;
    if (shape) for (int k = 0; k < 4; k++) shape[k] = p.shape[k];
    if (numel) *numel = p.numel;
    return CMP_OK;
}
// logical <-> stored layout of the three head-padded tensors of a block (ParamInfo::pad); to_store: logical -> stored (the
// caller zero-fills `st`), else stored -> logical
static void pad_copy(const cmp_model* m, const ParamInfo& p, float* st, float* logical, bool to_store) {
    const int H = m->H, D = m->D, Dl = m->Dl, E = m->E, Ea = m->Ea;
    if (p.pad == 1) {                       // rows x [3][H][Dl]  <->  rows x [3][H][D]
        const int64_t rows = p.numel / (3 * E);
        for (int64_t r = 0; r < rows; r++)
            for (int part = 0; part < 3; part++)
                for (int h = 0; h < H; h++)
                    for (int d = 0; d < Dl; d++) {
                        float& a = st[r * 3 * Ea + (int64_t)part * Ea + h * D + d];
                        float& b = logical[r * 3 * E + (int64_t)part * E + h * Dl + d];
                        if (to_store) a = b; else b = a;
                    }
    } else {                                // [H][Dl] x cols  <->  [H][D] x cols
        const int64_t cols = p.shape[1];
        for (int h = 0; h < H; h++)
            for (int d = 0; d < Dl; d++)
                for (int64_t c = 0; c < cols; c++) {
                    float& a = st[((int64_t)h * D + d) * cols + c];
                    float& b = logical[((int64_t)h * Dl + d) * cols + c];
                    if (to_store) a = b; else b = a;
                }
    }
}
static float* kind_buf(cmp_model* m, int kind) {
    switch (kind) {
        case 0: return m->P;
        case 1: return m->Am;
        case 2: return m->Av;
        case 3: return m->G;
    }
    return nullptr;
}
extern "C" int cmp_param_get(cmp_model* m, const char* name, int kind, float* host, int64_t numel) {
    CMP_REQUIRE(m && name && host, "param_get: null");
    auto it = m->index.find(name);
    CMP_REQUIRE(it != m->index.end(), "param_get: unknown parameter '%s'", name);
    const ParamInfo& p = m->params[it->second];
    CMP_REQUIRE(numel == p.numel, "param_get: '%s' has %lld elements, caller passed %lld", name, (long long)p.numel, (long long)numel);
    float* b = kind_buf(m, kind);
    CMP_REQUIRE(b, "param_get: bad kind %d", kind);
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(m->ctx->comm_stream));
    if (p.pad) {
        std::vector<float> st((size_t)p.store);
        HIP_CHECK(hipMemcpyAsync(st.data(), b + p.offset, (size_t)p.store * 4, hipMemcpyDeviceToHost, m->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        pad_copy(m, p, st.data(), host, false);
        return CMP_OK;
    }
    // (stream-ordered copies on the model's own stream, never the legacy stream: another thread's model may be capturing its decode graph)
    HIP_CHECK(hipMemcpyAsync(host, b + p.offset, (size_t)numel * 4, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    return CMP_OK;
}
extern "C" int cmp_param_set(cmp_model* m, const char* name, int kind, const float* host, int64_t numel) {
    CMP_REQUIRE(m && name && host, "param_set: null");
    auto it = m->index.find(name);
    CMP_REQUIRE(it != m->index.end(), "param_set: unknown parameter '%s'", name);
    const ParamInfo& p = m->params[it->second];
    CMP_REQUIRE(numel == p.numel, "param_set: '%s' has %lld elements, caller passed %lld", name, (long long)p.numel, (long long)numel);
    float* b = kind_buf(m, kind);
    CMP_REQUIRE(b, "param_set: bad kind %d", kind);
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    if (p.pad) {
        std::vector<float> st((size_t)p.store, 0.f);
        pad_copy(m, p, st.data(), const_cast<float*>(host), true);
        HIP_CHECK(hipMemcpyAsync(b + p.offset, st.data(), (size_t)p.store * 4, hipMemcpyHostToDevice, m->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    } else {
        HIP_CHECK(hipMemcpyAsync(b + p.offset, host, (size_t)numel * 4, hipMemcpyHostToDevice, m->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    }
    if (kind == 0) {
        m->param_version += 1;
        if (m->poisoned) {      // a checkpoint reload goes through here for every tensor: the flag falls when ALL of them have been set
            if (m->reload_seen.size() != m->params.size()) m->reload_seen.assign(m->params.size(), 0);
            m->reload_seen[it->second] = 1;
            bool all = true;
            for (char c : m->reload_seen) all = all && c;
            if (all) { m->poisoned = false; m->reload_seen.clear(); }
        }
    }
    if (kind == 0 && m->S) {
        int64_t n8 = (p.store + 7) / 8 * 8;
        CHECK_RC(launch_cast_bf16(m->ctx->stream, m->P + p.offset, m->S + p.offset, n8));
        HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    }
    return CMP_OK;
}
extern "C" int cmp_adam_iter_get(cmp_model* m, int64_t* it) {
    CMP_REQUIRE(m && it, "adam_iter_get: null");
    *it = m->iterations;
    return CMP_OK;
}
extern "C" int cmp_adam_iter_set(cmp_model* m, int64_t it) {
    CMP_REQUIRE(m && it >= 0, "adam_iter_set: bad value");
    m->iterations = it;
    return CMP_OK;
}

// -------------------------------------------------------------------------------------------------
// workspace
// -------------------------------------------------------------------------------------------------
static int ensure_workspace_fill(cmp_model* m, int B, int T);
int ensure_workspace(cmp_model* m, int B, int T) {
    const size_t mark = m->allocs.size();
    const int rc = ensure_workspace_fill(m, B, T);
    if (rc != CMP_OK && m->allocs.size() > mark) {
        // an allocation failed part-way: give back what this attempt took (a retry would otherwise allocate it all again)
        for (size_t i = mark; i < m->allocs.size(); i++) (void)hipFree(m->allocs[i]);
        m->allocs.resize(mark);
        m->xs.clear(); m->act.clear();
        m->capB = 0; m->capT = 0;
        (void)hipGetLastError();
    }
    return rc;
}
static int ensure_workspace_fill(cmp_model* m, int B, int T) {
    CMP_REQUIRE(B > 0 && T > 0, "batch and sequence must be positive (B=%d T=%d)", B, T);
    CMP_REQUIRE(T <= m->W, "sequence length %d exceeds window_size %d (wpe rows, transformer.py:675-679)", T, m->W);
    if (m->capB * m->capT >= B * T && m->capB > 0) return CMP_OK;
    CMP_REQUIRE(m->capB == 0, "workspace was sized for %d tokens; create the model with max_batch/max_seq covering B=%d T=%d",
                m->capB * m->capT, B, T);
    // size for the larger of (this call, cfg.max_batch*max_seq)
    int64_t M = (int64_t)B * T;
    int64_t Mcfg = (int64_t)std::max(1, m->cfg.max_batch) * std::max(1, std::min(m->cfg.max_seq, m->W));
    if (Mcfg > M) { M = Mcfg; }
    m->capB = (int)M; m->capT = 1;
    const int E = m->E, L = m->L;
    const size_t es = m->es;
    m->xs.resize(L + 1);
    for (int i = 0; i <= L; i++) CHECK_RC(dev_alloc(m, &m->xs[i], (size_t)M * E * es));
    m->act.resize(L);
    const bool ln = m->cfg.use_layer_norm != 0;
    for (int i = 0; i < L; i++) {
        LayerAct& a = m->act[i];
        if (ln) CHECK_RC(dev_alloc(m, &a.u, (size_t)M * E * es)); else a.u = m->xs[i];
        CHECK_RC(dev_alloc(m, &a.qkv, (size_t)M * 3 * m->Ea * es));
        CHECK_RC(dev_alloc(m, &a.att, (size_t)M * m->Ea * es));
        CHECK_RC(dev_alloc(m, &a.r, (size_t)M * E * es));
        if (ln) CHECK_RC(dev_alloc(m, &a.n, (size_t)M * E * es)); else a.n = a.r;
        CHECK_RC(dev_alloc(m, &a.fc, (size_t)M * 4 * E * es));
        CHECK_RC(dev_alloc(m, &a.g, (size_t)M * 4 * E * es));
        CHECK_RC(dev_alloc(m, &a.ln1_mean, (size_t)M * 4));
        CHECK_RC(dev_alloc(m, &a.ln1_rstd, (size_t)M * 4));
        CHECK_RC(dev_alloc(m, &a.ln2_mean, (size_t)M * 4));
        CHECK_RC(dev_alloc(m, &a.ln2_rstd, (size_t)M * 4));
        CHECK_RC(dev_alloc(m, &a.lse, (size_t)M * m->H * 4));
        if (ln && m->ST && E % 256 == 0) {          // partial row statistics of the fused block path
            CHECK_RC(dev_alloc(m, &a.ln1_part, (size_t)M * (E / 256) * 8));
            CHECK_RC(dev_alloc(m, &a.ln2_part, (size_t)M * (E / 256) * 8));
        }
    }
    if (ln && m->ST && E % 256 == 0) CHECK_RC(dev_alloc(m, &m->lnf_part, (size_t)M * (E / 256) * 8));
    if (ln && m->ST && E % 256 == 0) {
        // the 2L LayerNorm sites of the blocks as {partials, mean, rstd}: merged in one launch by the backward pass of the fused path
        std::vector<const void*> st;
        for (int i = 0; i < L; i++) {
            const LayerAct& a = m->act[i];
            st.insert(st.end(), {a.ln1_part, a.ln1_mean, a.ln1_rstd, a.ln2_part, a.ln2_mean, a.ln2_rstd});
        }
        CHECK_RC(dev_alloc(m, &m->ln_sites, st.size() * sizeof(void*)));
        HIP_CHECK(hipMemcpyAsync(m->ln_sites, st.data(), st.size() * sizeof(void*), hipMemcpyHostToDevice, m->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    }
    CHECK_RC(dev_alloc(m, &m->hf, (size_t)M * E * es));
    CHECK_RC(dev_alloc(m, &m->logits, (size_t)M * m->ldz * 4));
    CHECK_RC(dev_alloc(m, &m->dlogits, (size_t)M * m->ldz * es));
    CHECK_RC(dev_alloc(m, &m->lnf_mean, (size_t)M * 4));
    CHECK_RC(dev_alloc(m, &m->lnf_rstd, (size_t)M * 4));
    CHECK_RC(dev_alloc(m, &m->row_loss, (size_t)M * 4));
    CHECK_RC(dev_alloc(m, &m->row_correct, (size_t)M * 4));
    CHECK_RC(dev_alloc(m, &m->delta, (size_t)M * m->H * 4));
    CHECK_RC(dev_alloc(m, &m->x_dev, (size_t)M * 4));
    CHECK_RC(dev_alloc(m, &m->y_dev, (size_t)M * 4));
    CHECK_RC(dev_alloc(m, &m->dx, (size_t)M * E * es));
    CHECK_RC(dev_alloc(m, &m->dr, (size_t)M * E * es));
    CHECK_RC(dev_alloc(m, &m->tmpE, (size_t)M * m->Ea * es));          // [M, E] gradients and the [M, Ea] attention-output gradient
    CHECK_RC(dev_alloc(m, &m->dmask, (size_t)M * E * es));
    CHECK_RC(dev_alloc(m, &m->dmask2, (size_t)M * E * es));
    CHECK_RC(dev_alloc(m, &m->dmask3, (size_t)M * E * es));
    CHECK_RC(dev_alloc(m, &m->dfc, (size_t)M * 4 * E * es));
    CHECK_RC(dev_alloc(m, &m->dqkv, (size_t)M * 3 * m->Ea * es));
    {   // COMPOSER_DETERMINISTIC=1: no float atomics anywhere in the step -- split-K wgrads write per-split slabs and fold them
        // in a fixed order, the bias-gradient column sums and the LayerNorm parameter partials are folded by one thread per
        // column, the embedding scatter-add becomes a segmented gather.  Bitwise reproducible steps.
        const char* det = getenv("COMPOSER_DETERMINISTIC");
        if (det && det[0] == '1') {
            m->slab_bytes = (int64_t)16 * E * std::max(4 * E, 3 * m->Ea) * 4 + (int64_t)64 * m->V * E * 4;
            CHECK_RC(dev_alloc(m, &m->slab, (size_t)m->slab_bytes));
        }
    }
    CHECK_RC(dev_alloc(m, &m->ln_ws, (size_t)cmp_k_layernorm_bwd_ws((int)std::min<int64_t>(M, 1 << 30), E)));
    m->embed_ws_words = embed_bwd_sort_ws_words(M, m->V);
    if (m->embed_ws_words > 0) CHECK_RC(dev_alloc(m, &m->embed_ws, (size_t)m->embed_ws_words * 4));
    return CMP_OK;
}

// dropout of a gradient tensor (d(dropout(x)) = dy*mask/(1-p)); only launched when the rate is > 0
template <typename T>
__global__ void drop_apply_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n, int E, DropCfg d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = from_f32<T>(apply_drop(d, (uint32_t)(i / E), (uint32_t)(i % E), to_f32<T>(in[i])));
}
static int drop_apply(cmp_model* m, const void* in, void* out, int64_t n, float p, uint32_t stream_id) {
    DropCfg d = make_drop(p, m->drop_seed(), stream_id);
    int grid = (int)std::min<int64_t>(cdiv64(n, 256), 8192);
    if (m->dtype == CMP_BF16)
        drop_apply_kernel<bf16_t><<<grid, 256, 0, m->ctx->stream>>>((const bf16_t*)in, (bf16_t*)out, n, m->cfg.embedding_size, d);
    else
        drop_apply_kernel<float><<<grid, 256, 0, m->ctx->stream>>>((const float*)in, (float*)out, n, m->cfg.embedding_size, d);
    KERNEL_CHECK();
    return CMP_OK;
}

// transposed weight shadow: 32x32 tiles through LDS, grid (tiles of the largest matrix, 4L matrices)
__global__ __launch_bounds__(256) void transpose_weights_kernel(const bf16_t* __restrict__ S, bf16_t* __restrict__ ST,
                                                                const WDesc* __restrict__ desc) {
    __shared__ bf16_t tile[32][34];
    const WDesc d = desc[blockIdx.y];
    const int tiles_c = (d.cols + 31) >> 5, ntiles = ((d.rows + 31) >> 5) * tiles_c;     // edge tiles are bounds-checked (E % 32 != 0)
    if ((int)blockIdx.x >= ntiles) return;
    const int tr = (blockIdx.x / tiles_c) << 5, tc = (blockIdx.x % tiles_c) << 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const bf16_t* src = S + d.off;
    bf16_t* dst = ST + d.off;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = tr + ty + 8 * j, c = tc + tx;
        tile[ty + 8 * j][tx] = (r < d.rows && c < d.cols) ? src[(int64_t)r * d.cols + c] : (bf16_t)0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = tc + ty + 8 * j, r = tr + tx;
        if (c < d.cols && r < d.rows) dst[(int64_t)c * d.rows + r] = tile[tx][ty + 8 * j];
    }
}
// fold: the LayerNorm-fused path's set (c_attn / c_fc scaled by the LayerNorm gammas, with their fold vectors).  Skipped when ST
// already holds that set for the current parameter values (inference passes between parameter changes).
static int refresh_transposed_weights(cmp_model* m, bool fold) {
    if (!m->ST) return CMP_OK;
    const int want = fold ? 2 : 1;
    if (m->st_state == want && m->st_version == m->param_version) return CMP_OK;
    m->st_state = 0;
    const int maxtiles = std::max(cdiv(4 * m->E, 32) * cdiv(m->E, 32), cdiv(3 * m->Ea, 32) * cdiv(m->E, 32));   // [E,4E] or the head-padded [E,3Ea]
    if (fold) {
        transpose_weights_kernel<<<dim3(maxtiles, 2 * m->L), 256, 0, m->ctx->stream>>>(m->S, m->ST, (const WDesc*)m->wdesc_plain);
        KERNEL_CHECK();
        CHECK_RC(ln_fold_prep_run(m->ctx->stream, m->P, m->ST, m->lnfold, m->fdesc, 2 * m->L, 4 * m->E));
        CHECK_RC(lnf_fold_prep_run(m->ctx->stream, m->P + m->off_wte, m->P + m->off_lnf_g, m->P + m->off_lnf_b, m->wte_lnf, m->lnf_fold,
                                   m->lnf_fold + m->lnf_npad, m->V, m->E, m->lnf_npad));
    } else {
        transpose_weights_kernel<<<dim3(maxtiles, 4 * m->L), 256, 0, m->ctx->stream>>>(m->S, m->ST, (const WDesc*)m->wdesc);
        KERNEL_CHECK();
    }
    m->st_state = want;
    m->st_version = m->param_version;
    return CMP_OK;
}

// The LayerNorm-fused block path (common.h: LnEpi): no ln_1 / ln_2 kernel runs in the forward pass, `u` and `n` are never
// written by it.  Taken when every GEMM of a block reaches the persistent 256x256 kernel with whole tiles (bf16, E = 2 or 3
// segments of 256 columns, no head padding, tokens a multiple of 256 and enough of them) and the call is a plain forward /
// pass (no past, none of cmp_forward_ex's optional inputs).  COMPOSER_LN_FUSED=0 switches it off (A/B timing, tests).
// Measured (profiles/r5_01_ln_fused.txt): the inference forward of C2 7.87 -> 7.65 ms; a TRAIN step loses 1.5-2 % (the two LayerNorm
// backward kernels of a block write u / n instead of the forward kernels, and the fold GEMMs read a colder A operand than the
// one a LayerNorm kernel has just written), so training passes take it only when COMPOSER_LN_FUSED=2 asks for it (tests), or =3: round 6's
// form of the backward pass (weight gradients on the raw LayerNorm input rows, no u / n written at all) -- measured a tie with the
// unfused train step (profiles/r6_02_ln_raw_training.txt).
static bool ln_fused_ok(const cmp_model* m, int M, int past_len, bool training) {
    if (m->ln_fused_mode == 0) return false;
    if (training && m->ln_fused_mode != 2 && m->ln_fused_mode != 3) return false;
    if (!m->cfg.use_layer_norm || m->dtype != CMP_BF16 || m->slab || past_len) return false;
    if (m->Ea != m->E || m->E % 256 || m->E < 512 || m->E > 768 || !m->act[0].ln1_part) return false;   // 2 or 3 segments (the fold images' LDS)
    if (M % 256 || (int64_t)(M / 256) * (m->E / 256) < 192) return false;      // gemm_run's `big`: the N = E GEMMs too
    if (m->fwd_pos_ids || m->fwd_type_ids || m->fwd_amask || m->fwd_probs_out) return false;
    return true;
}

// -------------------------------------------------------------------------------------------------
// forward: Transformer.call (transformer.py:696-833) with past=None
// -------------------------------------------------------------------------------------------------
// column sums with per-split partials folded in a fixed order (deterministic mode; the slab is the scratch space)
static int colsum_det(cmp_model* m, const void* X, int ldx, float* out, int rows, int cols) {
    return colsum_run(m->ctx->stream, X, ldx, out, rows, cols, m->dtype, (float*)m->slab, (size_t)m->slab_bytes);
}
static int colsum_any(cmp_model* m, const void* X, int ldx, float* out, int rows, int cols) {
    if (m->slab) return colsum_det(m, X, ldx, out, rows, cols);
    return cmp_k_colsum(m->ctx->stream, X, ldx, out, rows, cols, m->dtype);
}
static int ln_bwd(cmp_model* m, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                  const void* resid, void* dx, float* dgamma, float* dbeta, int rows, void* dmask, float* colsum, float p_drop,
                  uint32_t rng_stream, const LnBwdFused* fz = nullptr, bool prescaled = false) {
    // (fused block path: every masked copy is written, dropout or not -- dx does not live until the weight gradients there)
    return layernorm_bwd_run(m->ctx->stream, dy, x, gamma, mean, rstd, resid, dx, dgamma, dbeta, m->ln_ws, rows, m->E, m->dtype,
                             dmask, colsum, p_drop, m->drop_seed(), rng_stream, m->slab != nullptr, fz, m->fused_last, prescaled);
}

#define GEMM_REV (1 << 20)      // gemm() flag (model.hip only): GemmExtra::rev
static int gemm(cmp_model* m, int ta, int tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                int ldc, const float* bias, int act, void* aux, int ldaux, const void* resid, int ldr, int out_fp32,
                int splitk, float p_drop, uint32_t rng_stream, int flags = 0, float* colsum = nullptr, const LnEpi* ln = nullptr) {
    const bool det = m->slab != nullptr;                               // COMPOSER_DETERMINISTIC=1
    GemmExtra ex;
    if (ln) ex.ln = *ln;
    ex.rev = (flags & GEMM_REV) != 0;
    flags &= ~GEMM_REV;
    ex.colsum = det ? nullptr : colsum;                                // fused column sums are float atomics
    if (splitk > 1 && det) { ex.slab_ws = (float*)m->slab; ex.slab_bytes = (size_t)m->slab_bytes; }
    ex.role = m->gemm_role >= 0 ? m->gemm_role : (ta ? 2 : 1);         // forward announces 0; backward: A^T = wgrad, else dgrad
    ex.max_wgs = m->ctx->dp_on() ? m->ctx->gemm_max_wgs : 0;              // leave CUs to the concurrent all-reduce kernels
    ex.dp = m->ctx->dp_on();                                   // RCCL kernels may hold CUs: dynamic item scheduling
    ex.sched = &m->ctx->gemm_sched;
    CHECK_RC(gemm_run(m->ctx->stream, m->dtype, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, act, aux, ldaux, resid, ldr,
                      out_fp32, splitk, p_drop, m->drop_seed(), rng_stream, flags, ex));
    if (det && colsum) CHECK_RC(colsum_det(m, C, ldc, colsum, M, N));
    return CMP_OK;
}

// copies rows (b, s0 + t) of src [*, sT rows per batch, src_ld] to rows (b, d0 + t) of dst, w 16-byte vectors per row
__global__ void rows_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int Tn, int w, int sT, int s0,
                                 int src_ld, int dT, int d0, int dst_ld) {
    const int64_t n = (int64_t)B * Tn * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % w);
        const int t = (int)((i / w) % Tn);
        const int b = (int)(i / ((int64_t)w * Tn));
        dst[((int64_t)b * dT + d0 + t) * dst_ld + c] = src[((int64_t)b * sT + s0 + t) * src_ld + c];
    }
}
static int rows_copy(cmp_model* m, const void* src, void* dst, int B, int Tn, int width, int sT, int s0, int src_w, int dT, int d0,
                     int dst_w) {
    const int es = (int)m->es, w = width * es / 16;
    const int64_t n = (int64_t)B * Tn * w;
    rows_copy_kernel<<<(int)std::min<int64_t>(cdiv64(n, 256), 4096), 256, 0, m->ctx->stream>>>(
        (const uint4*)src, (uint4*)dst, B, Tn, w, sT, s0, src_w * es / 16, dT, d0, dst_w * es / 16);
    KERNEL_CHECK();
    return CMP_OK;
}

// past_len > 0 is Transformer.call(inputs, past=presents) (transformer.py:735-765, 423-426): the T new tokens sit at
// positions past_len .. past_len+T-1; every layer's K/V of the first past_len positions were placed into rows [0, past_len)
// of act[i].qkv viewed as [B, past_len + T, 3E] by cmp_forward, the new rows are appended behind them, the causal attention
// kernel runs over all past_len + T rows (the mask of transformer.py:290-301 for nd = T, ns = past_len + T is the last T
// rows of the square one) and the T new output rows are taken out again.
// transformer.py:340-343: w / sqrt(head size) -- the reference's head size E / H, whatever size the kernels run on
static float attn_scale(const cmp_model* m) { return m->cfg.scale_attention ? 1.0f / sqrtf((float)m->Dl) : 1.0f; }

int model_forward(cmp_model* m, const int32_t* x_dev, int B, int T, bool training, int64_t step, int past_len) {
    Range range_("composer.forward");
    hipStream_t s = m->ctx->stream;
    struct RoleGuard {
        cmp_model* m;
        RoleGuard(cmp_model* mm) : m(mm) { m->gemm_role = 0; }
        ~RoleGuard() { m->gemm_role = -1; }
    } role_guard(m);
    const int E = m->E, Ea = m->Ea, M = B * T, dt = m->dtype;
    const float pr = training ? m->cfg.resid_dropout : 0.f;
    const float pa = training ? m->cfg.attn_dropout : 0.f;
    const bool ln = m->cfg.use_layer_norm != 0;
    const int Tp = past_len, Tt = past_len + T;
    m->lastB = B; m->lastT = Tt; m->last_past = Tp;
    m->fwd_gen += 1;
    const bool fused = ln_fused_ok(m, M, Tp, training);
    m->fused_last = fused;
    if (fused)
        CHECK_RC(embed_fwd_stats_run(s, x_dev, m->P + m->off_wte, m->P + m->off_wpe, m->xs[0], m->act[0].ln1_part, B, T, E, 0, pr,
                                     m->drop_seed(), drop_stream(step, 0, 0)));
    else
        CHECK_RC(embed_fwd_run(s, x_dev, m->P + m->off_wte, m->P + m->off_wpe, m->xs[0], B, T, E, Tp, dt, pr, m->drop_seed(),
                               drop_stream(step, 0, 0), m->fwd_pos_ids, m->fwd_type_ids));
    CHECK_RC(refresh_transposed_weights(m, fused));
    // Conv1D weight operand of the forward GEMMs: [in,out] as stored (fp32 mode), or the transposed bf16 copy (tb = 1)
    const bool wt = m->ST != nullptr;
    auto W = [&](int64_t off) { return wt ? (const void*)(m->ST + off) : m->w(off); };
    // tile-walk direction of the four GEMMs of a block (gemm.hip: item_coords, ep.rev): a GEMM that reads what the previous GEMM
    // wrote walks every XCD's run of tiles the other way round, so it starts on the rows still in that XCD's L2 (C2 inference
    // forward 7.22 -> 7.18 ms; COMPOSER_GEMM_ALT=0: all forwards)
    static const int alt_mode = [] { const char* e = getenv("COMPOSER_GEMM_ALT"); return e ? atoi(e) : 1; }();
    const int RV[4] = {alt_mode == 2 ? GEMM_REV : 0, alt_mode == 1 ? GEMM_REV : 0, alt_mode == 2 ? GEMM_REV : 0, alt_mode == 1 ? GEMM_REV : 0};
    // timing class 9: the decoder-block stack of this pass as one span (bench.py `forward.*.blocks_only_ms`: the attention + FFN
    // forward north_star prices, without embedding, ln_f, logits, loss and the host side of the call)
    PROF_START(9, s);
    for (int i = 0; fused && i < m->L; i++) {
        // transformer.py:574-597 with both LayerNorms inside the GEMM epilogues (common.h: LnEpi)
        const LayerOff& o = m->lo[i];
        LayerAct& a = m->act[i];
        const float* fold = m->lnfold + i * m->fold_stride;
        LnEpi l;
        l.in_part = a.ln1_part; l.np = E / 256; l.eps = m->cfg.ln_eps; l.cs = fold;
        CHECK_RC(gemm(m, 0, 1, M, 3 * E, E, m->xs[i], E, W(o.attn_w), E, a.qkv, 3 * E, fold + 3 * E, 0, nullptr, 0, nullptr, 0, 0, 1, 0.f,
                      0, RV[0], nullptr, &l));                                        // qkv = ln_1(x).Wattn + b   :583-584, 417
        CHECK_RC(attn_fwd_run(s, a.qkv, a.att, a.lse, B, Tt, m->H, m->D, attn_scale(m), dt, pa, m->drop_seed(), drop_stream(step, i, 1),
                              nullptr));
        l = LnEpi();
        l.in_part = a.ln1_part; l.np = E / 256; l.eps = m->cfg.ln_eps; l.gamma = m->P + o.ln1_g; l.beta = m->P + o.ln1_b;
        l.out_part = a.ln2_part;
        CHECK_RC(gemm(m, 0, 1, M, E, E, a.att, E, W(o.proj_w), E, a.r, E, m->P + o.proj_b, 0, nullptr, 0, m->xs[i], E, 0, 1, pr,
                      drop_stream(step, i, 2), RV[1], nullptr, &l));                  // r = ln_1(x) + dropout(proj)  :587
        l = LnEpi();
        l.in_part = a.ln2_part; l.np = E / 256; l.eps = m->cfg.ln_eps; l.cs = fold + 6 * E;
        CHECK_RC(gemm(m, 0, 1, M, 4 * E, E, a.r, E, W(o.fc_w), E, a.g, 4 * E, fold + 10 * E, 1, training ? a.fc : nullptr, 4 * E, nullptr,
                      0, 0, 1, 0.f, 0, RV[2], nullptr, &l));                          // g = gelu(ln_2(r).Wfc + b)    :591, 504
        l = LnEpi();
        // (the last block's statistics feed ln_f, folded into the logits GEMM on inference passes; a training pass keeps the ln_f
        //  kernel: its output hf is an operand of the tied weight gradient)
        l.out_part = i + 1 < m->L ? m->act[i + 1].ln1_part : (training ? nullptr : m->lnf_part);
        CHECK_RC(gemm(m, 0, 1, M, E, 4 * E, a.g, 4 * E, W(o.pr_w), 4 * E, m->xs[i + 1], E, m->P + o.pr_b, 0, nullptr, 0, a.r, E, 0, 1, pr,
                      drop_stream(step, i, 3), RV[3], nullptr, l.out_part ? &l : nullptr));   // x = r + dropout(mlp)  :594
    }
    for (int i = 0; !fused && i < m->L; i++) {
        const LayerOff& o = m->lo[i];
        LayerAct& a = m->act[i];
        if (ln)   // transformer.py:583-584 -- the LN output REPLACES the residual stream
            CHECK_RC(cmp_k_layernorm_fwd(s, m->xs[i], m->P + o.ln1_g, m->P + o.ln1_b, a.u, a.ln1_mean, a.ln1_rstd, M, E,
                                         m->cfg.ln_eps, dt));
        CHECK_RC(gemm(m, 0, wt, M, 3 * Ea, E, a.u, E, W(o.attn_w), wt ? E : 3 * Ea, Tp ? m->dqkv : a.qkv, 3 * Ea, m->P + o.attn_b, 0,
                      nullptr, 0, nullptr, 0, 0, 1, 0.f, 0));
        if (Tp) CHECK_RC(rows_copy(m, m->dqkv, a.qkv, B, T, 3 * Ea, T, 0, 3 * Ea, Tt, Tp, 3 * Ea));  // concat([past, new]) :423-426
        CHECK_RC(attn_fwd_run(s, a.qkv, a.att, a.lse, B, Tt, m->H, m->D, attn_scale(m), dt, pa, m->drop_seed(),
                              drop_stream(step, i, 1), m->fwd_amask));
        if (m->fwd_probs_out) {      // output_attention_weights: this layer's [B, H, T, Tt] probabilities, copied out before the next layer
            CHECK_RC(attn_probs_run(s, a.qkv, a.lse, m->fwd_amask, m->fwd_probs_dev, B, Tp, Tt, m->H, m->D, attn_scale(m), dt, pa,
                                    m->drop_seed(), drop_stream(step, i, 1)));
            HIP_CHECK(hipMemcpyAsync(m->fwd_probs_out[i], m->fwd_probs_dev, (size_t)B * m->H * T * Tt * 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        const void* att = a.att;
        if (Tp) { CHECK_RC(rows_copy(m, a.att, m->tmpE, B, T, Ea, Tt, Tp, Ea, T, 0, Ea)); att = m->tmpE; }
        CHECK_RC(gemm(m, 0, wt, M, E, Ea, att, Ea, W(o.proj_w), wt ? Ea : E, a.r, E, m->P + o.proj_b, 0, nullptr, 0, a.u, E, 0, 1, pr,
                      drop_stream(step, i, 2)));                                   // r = u + dropout(proj)  :587
        if (ln)
            CHECK_RC(cmp_k_layernorm_fwd(s, a.r, m->P + o.ln2_g, m->P + o.ln2_b, a.n, a.ln2_mean, a.ln2_rstd, M, E,
                                         m->cfg.ln_eps, dt));
        // the pre-activation (a.fc) is only needed by the backward pass
        CHECK_RC(gemm(m, 0, wt, M, 4 * E, E, a.n, E, W(o.fc_w), wt ? E : 4 * E, a.g, 4 * E, m->P + o.fc_b, 1, training ? a.fc : nullptr,
                      4 * E, nullptr, 0, 0, 1, 0.f, 0));                                              // g = gelu(fc)           :504
        CHECK_RC(gemm(m, 0, wt, M, E, 4 * E, a.g, 4 * E, W(o.pr_w), wt ? 4 * E : E, m->xs[i + 1], E, m->P + o.pr_b, 0, nullptr, 0, a.r, E,
                      0, 1, pr, drop_stream(step, i, 3)));                         // x = r + dropout(mlp)   :594
    }
    PROF_STOP(9, s, (double)M * m->L * (24.0 * E * E + 2.0 * E * Tt), 0.0);        // (causal attention on the unmasked half)
    static const bool lnf_fold_on = [] { const char* e = getenv("COMPOSER_LNF_FOLD"); return !(e && e[0] == '0'); }();
    if (fused && !training && lnf_fold_on) {
        // ln_f folded into the tied-logits GEMM like ln_1 / ln_2 into c_attn / c_fc (:811, 818): the raw rows of the last block
        // against the gamma-scaled copy of wte; hf is not written (cmp_hidden_get_at computes it on demand)
        LnEpi l;
        l.in_part = m->lnf_part; l.np = E / 256; l.eps = m->cfg.ln_eps; l.cs = m->lnf_fold;
        m->hf_valid = false;
        // (CMP_GEMM_TILE256: the fold epilogue exists on the persistent 256x256 kernel only, and [M, V] has fewer tiles than the
        //  blocks' [M, E] launches that ln_fused_ok sized the path by -- E = 768, V = 390 at 16 384 tokens: 192 against 128)
        CHECK_RC(gemm(m, 0, 1, M, m->V, E, m->xs[m->L], E, m->wte_lnf, E, m->logits, m->ldz, m->lnf_fold + m->lnf_npad, 0, nullptr, 0, nullptr,
                      0, 1, 1, 0.f, 0, CMP_GEMM_TILE256, nullptr, &l));
        return CMP_OK;
    }
    CHECK_RC(cmp_k_layernorm_fwd(s, m->xs[m->L], m->P + m->off_lnf_g, m->P + m->off_lnf_b, m->hf, m->lnf_mean, m->lnf_rstd, M,
                                 E, m->cfg.ln_eps, dt));                           // :811 (always applied)
    m->hf_valid = true;
    CHECK_RC(gemm(m, 0, 1, M, m->V, E, m->hf, E, m->w(m->off_wte), E, m->logits, m->ldz, nullptr, 0, nullptr, 0, nullptr, 0, 1,
                  1, 0.f, 0));                                                     // tied logits            :818
    return CMP_OK;
}

static int loss(cmp_model* m, const int32_t* y_dev, int M, bool want_grad) {
    Range range_("composer.loss");
    hipStream_t s = m->ctx->stream;
    CHECK_RC(cmp_k_softmax_xent(s, m->logits, m->ldz, y_dev, want_grad ? m->dlogits : nullptr, m->row_loss, m->row_correct, M,
                                m->V, 1.0f / (float)M, m->dtype));
    CHECK_RC(launch_metrics_reduce(s, m->row_loss, m->row_correct, M, m->metrics));
    return CMP_OK;
}

static int wgrad_splits(int K, int M, int N) {
    // contraction over tokens.  The persistent kernel works on 256x256 tiles x splits: with 8+ tiles, one item per CU
    // (256 CUs) -- every extra split is another round of f32 atomics on the same addresses (measured at K=131072:
    // 16 splits 298 us, 32 splits 369 us for the 2048x512 wgrad).  Few tiles: more splits than CUs/tiles only cost.
    const int tiles256 = cdiv(M, 256) * cdiv(N, 256);
    int s;
    if (tiles256 >= 8) s = std::max(1, 256 / tiles256);
    else s = std::max(1, 768 / std::max(1, cdiv(M, 128) * cdiv(N, 128)));
    return std::min(s, std::max(1, K / 256));
}

// Keras Adam on elements [begin, end) of the flat buffers (transformer.py:887,921); `step` = optimizer.iterations + 1
static int adam_range(cmp_model* m, hipStream_t s, int64_t begin, int64_t end, float lr, int64_t step, float gscale) {
    return cmp_k_adam(s, m->P + begin, m->G + begin, m->Am + begin, m->Av + begin, m->S ? (void*)(m->S + begin) : nullptr, end - begin, lr,
                      0.9f, 0.999f, 1e-7f, step, gscale);
}

// A gradient bucket is complete on the compute stream: all-reduce it on the communication stream behind an event, and
// apply Adam to that bucket's parameters right behind its all-reduce, still on the communication stream -- nothing the
// compute stream enqueues after this point reads them (a block's parameters are last read by its own backward pass, the
// tied embedding by the first GEMMs of the backward pass), so only the LAST bucket's all-reduce + update is left for the
// end-of-step wait instead of every bucket's update.
// Runs whenever a communicator exists, also with ONE rank (RCCL then copies in place): the 1-GPU tests and a 1-rank
// launched bench execute exactly the event / side-stream / ncclAllReduce sequence of the 8-GPU job.
// update == false: gradients only (cmp_loss_and_grads).
static int bucket_ready(cmp_model* m, int ev, int64_t begin, int64_t end, float lr, bool update) {
    cmp_ctx* c = m->ctx;
    if (!c->dp_on()) return CMP_OK;
    HIP_CHECK(hipEventRecord(m->bucket_ev[ev], c->stream));
    HIP_CHECK(hipStreamWaitEvent(c->comm_stream, m->bucket_ev[ev], 0));
    CHECK_RC(dp_allreduce(c, m->G + begin, (size_t)(end - begin)));
    m->dp_bytes_step += (end - begin) * 4;
    m->dp_msgs_step += 1;
    if (update) {
        CHECK_RC(adam_range(m, c->comm_stream, begin, end, lr, m->iterations + 1, 1.0f / (float)c->nranks));
        m->dp_buckets_updated += 1;
    }
    return CMP_OK;
}

// reverse mode of forward() (tf.GradientTape, transformer.py:916-920); formulas in SURVEY appendix A
static int backward(cmp_model* m, const int32_t* x_dev, int B, int T, int64_t step, bool allreduce, float lr = 0.f, bool update = false) {
    Range range_("composer.backward");
    hipStream_t s = m->ctx->stream;
    const int E = m->E, Ea = m->Ea, M = B * T, dt = m->dtype, V = m->V;
    const float pr = m->cfg.resid_dropout, pa = m->cfg.attn_dropout;
    const bool ln = m->cfg.use_layer_norm != 0;
    HIP_CHECK(hipMemsetAsync(m->G, 0, (size_t)m->total * 4, s));
    // tied logits: dwte += dZ^T.hf ; dhf = dZ.wte
    {
        // few output tiles (390 x 512 = 4 of 256x256): on the 128x128 kernel the launch is 768 workgroups of a 48th of the tokens each
        // (1.5 rounds, 48 MB of atomics); on the persistent 256x256 deep-pipeline kernel one item per CU
        const int t256 = cdiv(V, 256) * cdiv(E, 256);
        const bool p4 = dt == CMP_BF16 && t256 <= 64 && M >= 65536 && M % 32 == 0;       // (at 32 768 tokens the two forms tie)
        const int tsplit = p4 ? std::max(2, std::min(256 / t256, M / 1024)) : std::max(2, wgrad_splits(M, V, E));
        const int tflags = p4 ? CMP_GEMM_P4 : 0;
        CHECK_RC(gemm(m, 1, 0, V, E, M, m->dlogits, m->ldz, m->hf, E, m->G + m->off_wte, E, nullptr, 0, nullptr, 0, nullptr, 0, 1,
                      tsplit, 0.f, 0, tflags));
    }
    CHECK_RC(gemm(m, 0, 0, M, E, V, m->dlogits, m->ldz, m->w(m->off_wte), E, m->tmpE, E, nullptr, 0, nullptr, 0, nullptr, 0, 0,
                  1, 0.f, 0, CMP_GEMM_KPAD_ZERO));    // dlogits rows are zero-padded to ldz (softmax_xent kernel)
    // dx of every LayerNorm backward below is the gradient of the previous residual branch's dropout output, so the
    // kernel also emits that branch's masked gradient (dmask) and bias gradient (column sums)
    CHECK_RC(ln_bwd(m, m->tmpE, m->xs[m->L], m->P + m->off_lnf_g, m->lnf_mean, m->lnf_rstd, nullptr, m->dx,
                    m->G + m->off_lnf_g, m->G + m->off_lnf_b, M, m->dmask, m->G + m->lo[m->L - 1].pr_b, pr,
                    drop_stream(step, m->L - 1, 3)));
    bool dmo_ready = true;      // dmask / pr_b of the current layer already produced
    // The LayerNorm-fused block path (model_forward): the forward pass wrote neither u = ln_1(x) nor n = ln_2(r) and kept their
    // statistics as partials.  The two LayerNorm backward kernels of a block merge the partials and write u / n beside dx -- only
    // the block's weight-gradient launch reads them, so that launch moves behind ln_1's backward; by then ln_1's backward has
    // produced the NEXT block's MLP masked gradient, hence two alternating buffers for it.
    const bool fused = m->fused_last;
    // Round 6 (COMPOSER_LN_FUSED=3): the weight gradients of c_fc / c_attn run on the RAW LayerNorm input rows (r / x) --
    //   n^T . D = gamma o (r^T . D' - 1 (x) (mean^T . D')) + beta (x) colsum(D),   D' = rstd o D,   mean^T . D' = column means of r^T . D'
    // -- so the backward pass writes no LayerNorm output either: the GELU' dgrad epilogue stores D' = rstd o dfc,
    // the attention backward kernels store rstd o [dQ | dK | dV], the dgrads that consume them yield rstd o dn / rstd o du (the residual
    // term of du gets its rstd in the epilogue), which the LayerNorm backward kernels take as they are (LnBwdFused::prescaled), and one
    // small pass per block applies gamma / beta and the mean term to the two accumulated gradients (wgrad_ln_fix_kernel).
    const bool raw = fused && m->ln_fused_mode == 3;
    // (the statistics of all 2L sites, merged from their partials in one launch: the plain LayerNorm backward kernel and the attention
    //  backward kernels read per-row arrays; the statistics-from-partials form of the LayerNorm backward runs one wave per SIMD less)
    if (raw) CHECK_RC(ln_stats_merge_run(s, m->ln_sites, 2 * m->L, E / 256, m->cfg.ln_eps, M));
    void* const dmk[2] = {m->dmask, m->dmask3};
    int cur = 0;
    if (allreduce) CHECK_RC(bucket_ready(m, m->L, m->off_lnf_g, m->total, lr, update));
    // The four Conv1D weight gradients of a block contract over the same M tokens: with LayerNorm, bf16 and the atomic
    // (non-deterministic) split-K form they go out as ONE grouped launch behind the block's attention backward (gemm.hip:
    // gemm_wgrad_group_kernel) -- a quarter of the f32-atomic traffic of four split-K launches, equal k-steps per workgroup.
    // Until then both masked gradient copies of the block stay live: the MLP branch's in dmask, the attention branch's in dmask2.
    // (COMPOSER_DETERMINISTIC=1 takes the grouped launch too since round 6, in its last-arriver form: partial tiles summed in the
    //  order of the K ranges, no float atomics -- the deterministic step runs the launch order of the default one)
    const bool grouped = ln && dt == CMP_BF16;
    CMP_REQUIRE(!fused || grouped, "backward: the fused block path needs the grouped weight-gradient order");
    if (grouped && (int)m->wgrad_groups.size() != m->L) m->wgrad_groups.resize(m->L);
    for (int i = m->L - 1; i >= 0; i--) {
        const LayerOff& o = m->lo[i];
        LayerAct& a = m->act[i];
        bool group_now = grouped;
        // ---- MLP: x_out = r + dropout(gelu(n.Wfc+b).Wpr+b)
        const void* dmo = m->dx;
        if (fused) {
            dmo = dmk[cur];             // (also without dropout: dx is overwritten before the weight gradients read it)
        } else if (pr > 0.f) {
            if (!dmo_ready) CHECK_RC(drop_apply(m, m->dx, m->dmask, (int64_t)M * E, pr, drop_stream(step, i, 3)));
            dmo = m->dmask;
        }
        LnBwdFused fz;
        fz.np = E / 256; fz.eps = m->cfg.ln_eps;
        auto wgrad = [&](int Mw, int Nw, const void* A, int lda, const void* Bm, int ldb, float* Cw) {
            return gemm(m, 1, 0, Mw, Nw, M, A, lda, Bm, ldb, Cw, Nw, nullptr, 0, nullptr, 0, nullptr, 0, 1, std::max(2, wgrad_splits(M, Mw, Nw)), 0.f, 0);
        };
        if (!group_now) CHECK_RC(wgrad(4 * E, E, a.g, 4 * E, dmo, E, m->G + o.pr_w));
        if (!dmo_ready) CHECK_RC(colsum_any(m, dmo, E, m->G + o.pr_b, M, E));
        LnEpi ls;
        ls.np = E / 256; ls.eps = m->cfg.ln_eps; ls.scale = 1;
        ls.in_part = a.ln2_part;
        CHECK_RC(gemm(m, 0, 1, M, 4 * E, E, dmo, E, m->w(o.pr_w), E, m->dfc, 4 * E, nullptr, 2, a.fc, 4 * E, nullptr, 0, 0, 1,
                      0.f, 0, 0, m->G + o.fc_b, raw ? &ls : nullptr));      // dfc = (dmo.Wpr^T) * gelu'(fc) [raw: rstd_2 o that]; b_fc grad = column sums of dfc
        if (!group_now) CHECK_RC(wgrad(E, 4 * E, a.n, E, m->dfc, 4 * E, m->G + o.fc_w));
        void* const dao_mask = group_now ? m->dmask2 : m->dmask;
        if (ln) {
            CHECK_RC(gemm(m, 0, 1, M, E, 4 * E, m->dfc, 4 * E, m->w(o.fc_w), 4 * E, m->tmpE, E, nullptr, 0, nullptr, 0, nullptr,
                          0, 0, 1, 0.f, 0));                                       // dn
            fz.part = a.ln2_part; fz.beta = m->P + o.ln2_b; fz.yout = a.n;
            CHECK_RC(ln_bwd(m, m->tmpE, a.r, m->P + o.ln2_g, a.ln2_mean, a.ln2_rstd, m->dx, m->dr, m->G + o.ln2_g, m->G + o.ln2_b, M,
                            dao_mask, m->G + o.proj_b, pr, drop_stream(step, i, 2), (fused && !raw) ? &fz : nullptr, raw));   // dr = dx + LN2'(dn); dao, b_proj grad
        } else {
            CHECK_RC(gemm(m, 0, 1, M, E, 4 * E, m->dfc, 4 * E, m->w(o.fc_w), 4 * E, m->dr, E, nullptr, 0, nullptr, 0, m->dx, E,
                          0, 1, 0.f, 0));                                          // dr = dx + dn
        }
        // ---- attention: r = u + dropout(att.Wproj+b)
        const void* dao = m->dr;
        if (fused) {
            dao = dao_mask;
        } else if (pr > 0.f) {
            if (!ln) CHECK_RC(drop_apply(m, m->dr, m->dmask, (int64_t)M * E, pr, drop_stream(step, i, 2)));
            dao = dao_mask;
        }
        if (!group_now) CHECK_RC(wgrad(Ea, E, a.att, Ea, dao, E, m->G + o.proj_w));
        if (!ln) CHECK_RC(colsum_any(m, dao, E, m->G + o.proj_b, M, E));
        CHECK_RC(gemm(m, 0, 1, M, Ea, E, dao, E, m->w(o.proj_w), E, m->tmpE, Ea, nullptr, 0, nullptr, 0, nullptr, 0, 0, 1, 0.f,
                      0));                                                         // datt
        const bool det = m->slab != nullptr;     // the fused bias sums are float atomics: a separate fixed-order pass instead
        AttnLnRows alr;
        if (raw) alr.rstd = a.ln1_rstd;
        CHECK_RC(attn_bwd_run(s, a.qkv, a.att, m->tmpE, a.lse, m->delta, m->dqkv, B, T, m->H, m->D, attn_scale(m),
                              dt, pa, m->drop_seed(), drop_stream(step, i, 1), det ? nullptr : m->G + o.attn_b, alr));   // b_attn grad = column sums of dqkv
        if (det) CHECK_RC(colsum_det(m, m->dqkv, 3 * Ea, m->G + o.attn_b, M, 3 * Ea));
        auto group_launch = [&]() -> int {
            const WgradProblem wp[4] = {
                {a.g, 4 * E, dmo, E, m->G + o.pr_w, E, 4 * E, E},                  // dWpr   = g^T . dmo
                {raw ? a.r : a.n, E, m->dfc, 4 * E, m->G + o.fc_w, 4 * E, E, 4 * E},           // dWfc   = n^T . dfc        [raw: r^T . (rstd_2 o dfc)]
                {a.att, Ea, dao, E, m->G + o.proj_w, E, Ea, E},                    // dWproj = att^T . dao
                {raw ? m->xs[i] : a.u, E, m->dqkv, 3 * Ea, m->G + o.attn_w, 3 * Ea, E, 3 * Ea}};    // dWattn = u^T . dqkv  [raw: x^T . (rstd_1 o dqkv)]
            GemmExtra ex;
            ex.role = 2;
            ex.max_wgs = m->ctx->dp_on() ? m->ctx->gemm_max_wgs : 0;
            ex.dp = m->ctx->dp_on();
            ex.sched = &m->ctx->gemm_sched;
            if (m->slab) ex.wws = &m->wgrad_ws;        // deterministic mode: partial tiles through workspace slots, summed by each tile's last arriver (gemm.hip: WgLa)
            bool handled = false;
            CHECK_RC(wgrad_group_run(s, &m->wgrad_groups[i], wp, 4, M, ex, &handled));
            if (!handled) {          // shapes outside the grouped kernel's domain (M % 32, ...): one launch each, as without grouping
                CHECK_RC(wgrad(4 * E, E, a.g, 4 * E, dmo, E, m->G + o.pr_w));
                CHECK_RC(wgrad(E, 4 * E, wp[1].A, E, m->dfc, 4 * E, m->G + o.fc_w));
                CHECK_RC(wgrad(Ea, E, a.att, Ea, dao, E, m->G + o.proj_w));
                CHECK_RC(wgrad(E, 3 * Ea, wp[3].A, E, m->dqkv, 3 * Ea, m->G + o.attn_w));
            }
            if (raw) {               // gamma / beta and the mean term on the two gradients that were accumulated on raw rows
                CHECK_RC(wgrad_ln_fix_run(s, m->G + o.fc_w, E, 4 * E, m->P + o.ln2_g, m->P + o.ln2_b, m->G + o.fc_b,
                                          m->G + o.attn_w, E, 3 * E, m->P + o.ln1_g, m->P + o.ln1_b, m->G + o.attn_b));
            }
            return CMP_OK;
        };
        if (group_now && (!fused || raw)) {
            CHECK_RC(group_launch());
        } else if (!group_now) {
            CHECK_RC(wgrad(E, 3 * Ea, a.u, E, m->dqkv, 3 * Ea, m->G + o.attn_w));
        }
        if (ln) {
            ls.in_part = a.ln1_part;
            CHECK_RC(gemm(m, 0, 1, M, E, 3 * Ea, m->dqkv, 3 * Ea, m->w(o.attn_w), 3 * Ea, m->tmpE, E, nullptr, 0, nullptr, 0, m->dr,
                          E, 0, 1, 0.f, 0, 0, nullptr, raw ? &ls : nullptr));      // du = dr + dqkv.Wattn^T  [raw: rstd_1 o du]
            // dx_in = LN1'(du): no skip connection around LN1; feeds layer i-1's MLP branch (or the embedding for i = 0)
            fz.part = a.ln1_part; fz.beta = m->P + o.ln1_b; fz.yout = a.u;
            void* const next_mask = fused ? dmk[cur ^ 1] : m->dmask;
            CHECK_RC(ln_bwd(m, m->tmpE, m->xs[i], m->P + o.ln1_g, a.ln1_mean, a.ln1_rstd, nullptr, m->dx, m->G + o.ln1_g,
                            m->G + o.ln1_b, M, i > 0 ? next_mask : nullptr, i > 0 ? m->G + m->lo[i - 1].pr_b : nullptr, pr,
                            drop_stream(step, i > 0 ? i - 1 : 0, 3), (fused && !raw) ? &fz : nullptr, raw));
            if (fused) {
                if (!raw) CHECK_RC(group_launch());          // g^T.dmo, n^T.dfc, att^T.dao, u^T.dqkv -- u and n as just written
                cur ^= 1;
            }
            dmo_ready = true;
        } else {
            dmo_ready = false;
            CHECK_RC(gemm(m, 0, 1, M, E, 3 * Ea, m->dqkv, 3 * Ea, m->w(o.attn_w), 3 * Ea, m->dx, E, nullptr, 0, nullptr, 0, m->dr, E,
                          0, 1, 0.f, 0));
        }
        if (allreduce) CHECK_RC(bucket_ready(m, i, o.begin, o.end, lr, update));
    }
    CHECK_RC(embed_bwd_run(s, x_dev, m->dx, m->G + m->off_wte, m->G + m->off_wpe, B, T, E, 0, dt, pr, m->drop_seed(),
                           drop_stream(step, 0, 0), m->slab ? V : 0, (float*)m->slab, (size_t)m->slab_bytes, V, m->embed_ws,
                           m->embed_ws_words));
    if (allreduce) CHECK_RC(bucket_ready(m, m->L + 1, 0, m->lo[0].begin, lr, update));
    return CMP_OK;
}

// metrics across ranks (the reference logs ONE loss per step, transformer.py:929-939): {loss mean, accuracy, 1}
// of every rank summed by a 3-float all-reduce on the communication stream, divided by the rank count after it
__global__ void dp_metrics_pack_kernel(const Metrics* __restrict__ mt, float* __restrict__ dp) {
    if (threadIdx.x == 0) { dp[0] = mt->loss_mean; dp[1] = mt->acc; dp[2] = 1.0f; dp[3] = 0.f; }
}
__global__ void dp_metrics_unpack_kernel(Metrics* __restrict__ mt, const float* __restrict__ dp) {
    if (threadIdx.x == 0 && dp[2] > 0.f) { mt->loss_mean = dp[0] / dp[2]; mt->acc = dp[1] / dp[2]; }
}
static int dp_metrics_begin(cmp_model* m) {
    cmp_ctx* c = m->ctx;
    if (!c->dp_on()) return CMP_OK;
    dp_metrics_pack_kernel<<<1, 64, 0, c->stream>>>(m->metrics, m->dp_metrics);
    KERNEL_CHECK();
    HIP_CHECK(hipEventRecord(m->metrics_ev, c->stream));
    HIP_CHECK(hipStreamWaitEvent(c->comm_stream, m->metrics_ev, 0));
    CHECK_RC(dp_allreduce(c, m->dp_metrics, 3));
    m->dp_bytes_step = 12;          // first message of the step: the gradient buckets add theirs (bucket_ready)
    m->dp_msgs_step = 1;
    return CMP_OK;
}

// sums the finished event pairs of the ring into dp_exposed_ms (all == false: only the slot about to be reused)
static int dp_fold(cmp_model* m, int slot, bool all) {
    for (int i = 0; i < cmp_model::DP_RING; i++) {
        if (!(all || i == slot) || !m->dp_wait_used[i]) continue;
        HIP_CHECK(hipEventSynchronize(m->dp_wait_b[i]));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, m->dp_wait_a[i], m->dp_wait_b[i]));
        m->dp_exposed_ms += ms;
        m->dp_folded += 1;
        m->dp_wait_used[i] = false;
    }
    return CMP_OK;
}

static int adam(cmp_model* m, float lr) {
    Range range_("composer.adam");
    cmp_ctx* c = m->ctx;
    m->iterations += 1;
    m->param_version += 1;
    if (c->dp_on()) {
        // every bucket was all-reduced AND updated on the communication stream (bucket_ready); the compute stream waits for the
        // last of them here, between two timed events: what they measure is the communication (+ last update) that the
        // backward pass did not hide -- SURVEY 8d "exposed comm time per step", read with cmp_dp_stats
        const int slot = (int)(m->dp_steps % cmp_model::DP_RING);
        CHECK_RC(dp_fold(m, slot, false));
        if (!m->dp_wait_a[slot]) {
            HIP_CHECK(hipEventCreate(&m->dp_wait_a[slot]));
            HIP_CHECK(hipEventCreate(&m->dp_wait_b[slot]));
        }
        HIP_CHECK(hipEventRecord(m->comm_done, c->comm_stream));
        HIP_CHECK(hipEventRecord(m->dp_wait_a[slot], c->stream));
        HIP_CHECK(hipStreamWaitEvent(c->stream, m->comm_done, 0));
        HIP_CHECK(hipEventRecord(m->dp_wait_b[slot], c->stream));
        m->dp_wait_used[slot] = true;
        m->dp_steps += 1;
        dp_metrics_unpack_kernel<<<1, 64, 0, c->stream>>>(m->metrics, m->dp_metrics);
        KERNEL_CHECK();
        return CMP_OK;
    }
    return adam_range(m, c->stream, 0, m->total, lr, m->iterations, 1.0f);
}

extern "C" int cmp_dp_stats(cmp_model* m, int reset, int64_t* steps, double* exposed_ms, int64_t* bytes_per_step, int* msgs_per_step) {
    CMP_REQUIRE(m, "dp_stats: null model");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(dp_fold(m, -1, true));
    if (steps) *steps = m->dp_folded;
    if (exposed_ms) *exposed_ms = m->dp_exposed_ms;
    if (bytes_per_step) *bytes_per_step = m->ctx->dp_on() ? m->dp_bytes_step : 0;
    if (msgs_per_step) *msgs_per_step = m->ctx->dp_on() ? m->dp_msgs_step : 0;
    if (reset) { m->dp_steps = 0; m->dp_folded = 0; m->dp_exposed_ms = 0.0; }
    return CMP_OK;
}

// Which RCCL this process is running on (ncclGetVersion of the library the dynamic linker bound: with torch imported first that is
// torch's bundled librccl, otherwise /opt/rocm's -- composer_amd/_lib.py reports the path): code = major * 10000 + minor * 100 + patch.
extern "C" int cmp_dp_rccl_version(int* version) {
    CMP_REQUIRE(version, "dp_rccl_version: null");
    NCCL_CHECK(ncclGetVersion(version));
    return CMP_OK;
}

// The gradient exchange of ONE train step alone (bench.py --allreduce-only; the first thing to look at when an 8-GPU run scales badly):
// the step's own message pattern -- the 3-float metrics message, then the L + 2 gradient buckets in the order the backward pass
// completes them (ln_f, block L-1 ... block 0, embeddings), each the bucket's own range of G -- issued back to back on the communication
// stream `reps` times between two events, nothing on the compute stream.  G is zeroed first (sums of zeros stay zero) and is garbage for
// no one: the next train step zeroes it again.  ms = total time of the reps; bytes / msgs = per repetition.
extern "C" int cmp_dp_allreduce_pattern(cmp_model* m, int reps, double* ms, int64_t* bytes_per_rep, int* msgs_per_rep) {
    CMP_REQUIRE(m && reps > 0 && ms, "dp_allreduce_pattern: bad arguments");
    cmp_ctx* c = m->ctx;
    CMP_REQUIRE(c->dp_on(), "dp_allreduce_pattern: communicator not initialised");
    HIP_CHECK(hipSetDevice(c->device));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipMemsetAsync(m->G, 0, (size_t)m->total * 4, c->comm_stream));
    HIP_CHECK(hipMemsetAsync(m->dp_metrics, 0, 16, c->comm_stream));
    hipEvent_t a = nullptr, b = nullptr;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    int64_t bytes = 0;
    int msgs = 0, rc = CMP_OK;
    auto one = [&](float* p, int64_t n) { if (rc == CMP_OK) { rc = dp_allreduce(c, p, (size_t)n); bytes += n * 4; msgs += 1; } };
    for (int r = -1; r < reps && rc == CMP_OK; r++) {       // r = -1: one untimed repetition (connection set-up, first-use allocations)
        if (r == 0) { HIP_CHECK(hipEventRecord(a, c->comm_stream)); bytes = 0; msgs = 0; }
        one(m->dp_metrics, 3);
        one(m->G + m->off_lnf_g, m->total - m->off_lnf_g);
        for (int i = m->L - 1; i >= 0; i--) one(m->G + m->lo[i].begin, m->lo[i].end - m->lo[i].begin);
        one(m->G, m->lo[0].begin);
    }
    if (rc == CMP_OK) {
        HIP_CHECK(hipEventRecord(b, c->comm_stream));
        HIP_CHECK(hipEventSynchronize(b));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, a, b));
        *ms = t;
    } else {
        (void)hipStreamSynchronize(c->comm_stream);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (bytes_per_rep) *bytes_per_rep = bytes / reps;
    if (msgs_per_rep) *msgs_per_rep = msgs / reps;
    return rc;
}

extern "C" int cmp_model_path_info(cmp_model* m, int* fused, int64_t* wgrad_table_builds) {
    CMP_REQUIRE(m, "model_path_info: null model");
    if (fused) *fused = m->fused_last ? 1 : 0;
    if (wgrad_table_builds) {
        int64_t n = 0;
        for (const WgradGroup& g : m->wgrad_groups) n += g.rebuilds;
        *wgrad_table_builds = n;
    }
    return CMP_OK;
}

static int fetch_metrics(cmp_model* m) {
    HIP_CHECK(hipMemcpyAsync(m->metrics_host, m->metrics, sizeof(Metrics), hipMemcpyDeviceToHost, m->ctx->stream));
    return CMP_OK;
}

// ids handed over as DEVICE pointers cannot be checked on the host: one pass clamps them into [0, V) (an id outside the
// table would read / scatter-add outside wte) into the model's own id buffers and counts the offenders; cmp_train_metrics
// reports a non-zero count as CMP_ERR_INVALID.
__global__ void sanitize_ids_kernel(const int32_t* __restrict__ x, const int32_t* __restrict__ y, int32_t* __restrict__ xo,
                                    int32_t* __restrict__ yo, int n, int V, int* __restrict__ bad) {
    int nbad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int a = x[i], b = y[i];
        nbad += (a < 0 || a >= V) + (b < 0 || b >= V);
        xo[i] = min(max(a, 0), V - 1);
        yo[i] = min(max(b, 0), V - 1);
    }
    if (nbad) atomicAdd(bad, nbad);
}

// the step itself on ids already in HBM and already known to be in range
static int train_step_enqueue(cmp_model* m, const int32_t* x_dev, const int32_t* y_dev, int B, int T, float lr) {
    // everything that can be checked is checked before anything is enqueued: in a data-parallel job the peers are already in the
    // metrics all-reduce by the time the backward pass starts
    CMP_REQUIRE(lr >= 0.f && lr == lr, "train step: learning rate %g is not a non-negative number", (double)lr);
    CMP_REQUIRE(!m->poisoned, "train step: an earlier data-parallel step failed after some gradient buckets had been all-reduced and "
                "applied (parameters are partially stepped and may differ between replicas): reload a checkpoint on every rank");
    const int64_t step = m->iterations;
    CHECK_RC(model_forward(m, x_dev, B, T, true, step));
    CHECK_RC(loss(m, y_dev, B * T, true));
    CHECK_RC(dp_metrics_begin(m));
    m->dp_buckets_updated = 0;
    int rc = backward(m, x_dev, B, T, step, true, lr, true);
    if (rc == CMP_OK) rc = adam(m, lr);
    if (rc != CMP_OK && m->ctx->dp_on()) {
        // Buckets already handed to the communication stream keep running (all-reduce + Adam): the compute stream must not touch
        // G / P before they are done (the next step's memset of G would race them), and a model whose buckets were partly
        // updated is no longer the model the caller thinks it has.  The failing call's message is kept.
        cmp_ctx* c = m->ctx;
        if (hipEventRecord(m->comm_done, c->comm_stream) == hipSuccess) (void)hipStreamWaitEvent(c->stream, m->comm_done, 0);
        else (void)hipStreamSynchronize(c->comm_stream);
        (void)hipGetLastError();
        if (m->dp_buckets_updated > 0) {
            m->poisoned = true;
            m->param_version += 1;         // some buckets moved: transposed / folded weight copies and the decode weights are stale
            m->st_state = 0;
            m->reload_seen.assign(m->params.size(), 0);
        }
    }
    return rc;
}

extern "C" int cmp_train_step_dev(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, float lr) {
    CMP_REQUIRE(m && x_dev && y_dev, "train_step_dev: null argument");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(ensure_workspace(m, B, T));
    hipStream_t s = m->ctx->stream;
    HIP_CHECK(hipMemsetAsync(&m->metrics->bad_ids, 0, sizeof(int), s));
    sanitize_ids_kernel<<<std::min(cdiv(B * T, 256), 1024), 256, 0, s>>>((const int32_t*)x_dev, (const int32_t*)y_dev, m->x_dev,
                                                                         m->y_dev, B * T, m->V, &m->metrics->bad_ids);
    KERNEL_CHECK();
    CHECK_RC(train_step_enqueue(m, m->x_dev, m->y_dev, B, T, lr));
    CHECK_RC(fetch_metrics(m));
    return CMP_OK;
}

// Diagnostic: how many launches ONE train step of this shape enqueues.  The step is stream-captured (nothing executes), the nodes of
// the captured graph are counted by type, the graph is dropped and the host-side state the enqueue touched is put back (optimizer
// iteration, parameter / pass generations, what the transposed shadows hold, and the description of the pass whose activations are
// held: shape, path, whether ln_f's output exists -- cmp_present_get / cmp_hidden_get_at keep answering for the last EXECUTED pass).
// Needs one executed train step of the same shape first (item tables of the grouped weight gradients).  Not available once a
// communicator exists (RCCL calls inside a capture).
extern "C" int cmp_train_step_graph_probe(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, int* kernels, int* others,
                                         int replay_reps, float* replay_ms);
extern "C" int cmp_train_step_launches(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, int* kernels, int* others) {
    return cmp_train_step_graph_probe(m, x_dev, y_dev, B, T, kernels, others, 0, nullptr);
}
extern "C" int cmp_train_step_graph_probe(cmp_model* m, const void* x_dev, const void* y_dev, int B, int T, int* kernels, int* others,
                                         int replay_reps, float* replay_ms) {
    CMP_REQUIRE(m && x_dev && y_dev && kernels, "train_step_launches: null argument");
    CMP_REQUIRE(!m->ctx->dp_on(), "train_step_launches: not available with a communicator");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(ensure_workspace(m, B, T));
    hipStream_t s = m->ctx->stream;
    HIP_CHECK(hipStreamSynchronize(s));
    const int64_t it = m->iterations, pv = m->param_version, gen = m->fwd_gen, stv = m->st_version;
    const int sts = m->st_state, lB = m->lastB, lT = m->lastT, lP = m->last_past;
    const bool fl = m->fused_last, hv = m->hf_valid;
    // (precondition: the grouped weight-gradient item tables of this shape exist, i.e. one train step of the shape has EXECUTED --
    //  building them synchronises the stream, which a capture cannot do; wgrad_group_run says so when it happens)
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = train_step_enqueue(m, (const int32_t*)x_dev, (const int32_t*)y_dev, B, T, 0.f);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    m->iterations = it; m->param_version = pv; m->fwd_gen = gen; m->st_version = stv; m->st_state = sts;
    m->lastB = lB; m->lastT = lT; m->last_past = lP; m->fused_last = fl; m->hf_valid = hv;
    if (rc != CMP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    HIP_CHECK(e);
    size_t n = 0;
    HIP_CHECK(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) HIP_CHECK(hipGraphGetNodes(g, nodes.data(), &n));
    int nk = 0, no = 0;
    for (size_t i = 0; i < n; i++) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[i], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel) nk++; else no++;
    }
    // replay_reps > 0 (measurement only): the captured step is instantiated and replayed back to back -- every replay draws the SAME
    // dropout masks and applies Adam with the SAME iteration count (both are launch arguments frozen at capture), so the model is not
    // a training run afterwards; what it answers is what a hipGraph of the step would cost per replay against stream launches.
    if (replay_reps > 0 && replay_ms) {
        hipGraphExec_t ge = nullptr;
        hipError_t e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (e2 == hipSuccess) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            for (int i = 0; i < 3 && e2 == hipSuccess; i++) e2 = hipGraphLaunch(ge, s);
            (void)hipEventRecord(a, s);
            for (int i = 0; i < replay_reps && e2 == hipSuccess; i++) e2 = hipGraphLaunch(ge, s);
            (void)hipEventRecord(b, s);
            if (e2 == hipSuccess) e2 = hipEventSynchronize(b);
            float ms = 0.f;
            if (e2 == hipSuccess) (void)hipEventElapsedTime(&ms, a, b);
            *replay_ms = ms / (float)replay_reps;
            (void)hipEventDestroy(a); (void)hipEventDestroy(b);
            (void)hipGraphExecDestroy(ge);
        }
        if (e2 != hipSuccess) { (void)hipGraphDestroy(g); HIP_CHECK(e2); }
    }
    (void)hipGraphDestroy(g);
    *kernels = nk;
    if (others) *others = no;
    return CMP_OK;
}

extern "C" int cmp_train_metrics(cmp_model* m, float* loss_out, float* acc_out) {
    CMP_REQUIRE(m, "train_metrics: null model");
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    if (loss_out) *loss_out = m->metrics_host->loss_mean;
    if (acc_out) *acc_out = m->metrics_host->acc;
    CMP_REQUIRE(m->metrics_host->bad_ids == 0, "train step: %d token ids outside [0, %d) were passed by device pointer (clamped)",
                m->metrics_host->bad_ids, m->V);
    return CMP_OK;
}

static int check_host_ids(cmp_model* m, const int32_t* ids, int64_t n, const char* what) {
    // one vectorisable pass for the common case (a negative id is a huge unsigned one), the search for the offender only behind it
    uint32_t mx = 0;
    for (int64_t i = 0; i < n; i++) mx = std::max(mx, (uint32_t)ids[i]);
    if (mx < (uint32_t)m->V) return CMP_OK;
    for (int64_t i = 0; i < n; i++)
        CMP_REQUIRE(ids[i] >= 0 && ids[i] < m->V, "%s: token id %d at flat index %lld is outside [0, %d)", what, ids[i], (long long)i, m->V);
    return CMP_OK;
}

static int upload_xy(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, int past_len = 0) {
    CHECK_RC(ensure_workspace(m, B, T + past_len));
    CHECK_RC(check_host_ids(m, x, (int64_t)B * T, "input ids"));
    if (y) CHECK_RC(check_host_ids(m, y, (int64_t)B * T, "target ids"));
    HIP_CHECK(hipMemsetAsync(&m->metrics->bad_ids, 0, sizeof(int), m->ctx->stream));
    HIP_CHECK(hipMemcpyAsync(m->x_dev, x, (size_t)B * T * 4, hipMemcpyHostToDevice, m->ctx->stream));
    if (y) HIP_CHECK(hipMemcpyAsync(m->y_dev, y, (size_t)B * T * 4, hipMemcpyHostToDevice, m->ctx->stream));
    return CMP_OK;
}

// ---- pipelined host-buffer steps (Transformer.train): the ids of step s+1 are staged in pinned memory and uploaded on a
// copy stream while step s computes, and a step's metrics are read one or two steps later, so the Python loop never waits
// for the step it has just submitted (the loop body of transformer.py:914-946 with its logged values unchanged).
static int ensure_stages(cmp_model* m, int64_t tokens) {
    if (m->stage_cap >= tokens) return CMP_OK;
    CMP_REQUIRE(m->stage_cap == 0, "train_step_async: staging was sized for %lld tokens", (long long)m->stage_cap);
    // sized once for the whole workspace (every later batch fits); a failure part-way keeps what was allocated in the model
    // (freed by cmp_model_destroy) and a retry only allocates what is still missing
    const int64_t cap = std::max<int64_t>(tokens, (int64_t)m->capB * m->capT);
    if (!m->stage_metrics) {
        HIP_CHECK(hipHostMalloc((void**)&m->stage_metrics, sizeof(Metrics) * cmp_model::STAGES, hipHostMallocDefault));
        memset(m->stage_metrics, 0, sizeof(Metrics) * cmp_model::STAGES);
    }
    for (int i = 0; i < cmp_model::STAGES; i++) {
        if (!m->stage_host[i]) HIP_CHECK(hipHostMalloc((void**)&m->stage_host[i], (size_t)cap * 8, hipHostMallocDefault));
        if (!m->stage_dev[i]) CHECK_RC(dev_alloc(m, &m->stage_dev[i], (size_t)cap * 8));
        if (!m->stage_uploaded[i]) HIP_CHECK(hipEventCreateWithFlags(&m->stage_uploaded[i], hipEventDisableTiming));
        if (!m->stage_done[i]) HIP_CHECK(hipEventCreateWithFlags(&m->stage_done[i], hipEventDisableTiming));
    }
    m->stage_cap = cap;
    return CMP_OK;
}

extern "C" int cmp_train_step_async(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, float lr, int64_t* ticket) {
    CMP_REQUIRE(m && x && y && ticket, "train_step_async: null argument");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(ensure_workspace(m, B, T));
    const int64_t n = (int64_t)B * T;
    CHECK_RC(ensure_stages(m, n));
    CHECK_RC(check_host_ids(m, x, n, "input ids"));
    CHECK_RC(check_host_ids(m, y, n, "target ids"));
    const int64_t tk = m->next_ticket;
    const int slot = (int)(tk % cmp_model::STAGES);
    if (m->stage_ticket[slot] >= 0) HIP_CHECK(hipEventSynchronize(m->stage_done[slot]));    // slot still owned by step tk - STAGES
    memcpy(m->stage_host[slot], x, (size_t)n * 4);
    memcpy(m->stage_host[slot] + n, y, (size_t)n * 4);
    cmp_ctx* c = m->ctx;
    HIP_CHECK(hipMemcpyAsync(m->stage_dev[slot], m->stage_host[slot], (size_t)n * 8, hipMemcpyHostToDevice, c->copy_stream));
    HIP_CHECK(hipEventRecord(m->stage_uploaded[slot], c->copy_stream));
    HIP_CHECK(hipStreamWaitEvent(c->stream, m->stage_uploaded[slot], 0));
    HIP_CHECK(hipMemsetAsync(&m->metrics->bad_ids, 0, sizeof(int), c->stream));
    CHECK_RC(train_step_enqueue(m, m->stage_dev[slot], m->stage_dev[slot] + n, B, T, lr));
    HIP_CHECK(hipMemcpyAsync(&m->stage_metrics[slot], m->metrics, sizeof(Metrics), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipMemcpyAsync(m->metrics_host, m->metrics, sizeof(Metrics), hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipEventRecord(m->stage_done[slot], c->stream));
    m->stage_ticket[slot] = tk;
    m->next_ticket = tk + 1;
    *ticket = tk;
    return CMP_OK;
}

extern "C" int cmp_train_metrics_wait(cmp_model* m, int64_t ticket, float* loss_out, float* acc_out) {
    CMP_REQUIRE(m, "train_metrics_wait: null model");
    const int slot = (int)(ticket % cmp_model::STAGES);
    CMP_REQUIRE(ticket >= 0 && m->stage_cap > 0 && m->stage_ticket[slot] == ticket,
                "train_metrics_wait: ticket %lld is not in flight (at most %d steps are)", (long long)ticket, cmp_model::STAGES);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    HIP_CHECK(hipEventSynchronize(m->stage_done[slot]));
    if (loss_out) *loss_out = m->stage_metrics[slot].loss_mean;
    if (acc_out) *acc_out = m->stage_metrics[slot].acc;
    return CMP_OK;
}

extern "C" int cmp_train_step(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, float lr, float* loss_out,
                              float* acc_out) {
    CMP_REQUIRE(m && x && y, "train_step: null argument");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(upload_xy(m, x, y, B, T));
    CHECK_RC(train_step_enqueue(m, m->x_dev, m->y_dev, B, T, lr));
    CHECK_RC(fetch_metrics(m));
    if (loss_out || acc_out) CHECK_RC(cmp_train_metrics(m, loss_out, acc_out));
    return CMP_OK;
}

extern "C" int cmp_loss_and_grads(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, float* loss_out,
                                  float* acc_out) {
    CMP_REQUIRE(m && x && y, "loss_and_grads: null argument");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(upload_xy(m, x, y, B, T));
    const int64_t step = m->iterations;
    CHECK_RC(model_forward(m, m->x_dev, B, T, true, step));
    CHECK_RC(loss(m, m->y_dev, B * T, true));
    CHECK_RC(backward(m, m->x_dev, B, T, step, false));
    CHECK_RC(fetch_metrics(m));
    CHECK_RC(cmp_train_metrics(m, loss_out, acc_out));
    return CMP_OK;
}

extern "C" int cmp_eval_step(cmp_model* m, const int32_t* x, const int32_t* y, int B, int T, double* loss_sum,
                             int64_t* correct, int64_t* count) {
    CMP_REQUIRE(m && x && y, "eval_step: null argument");
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(upload_xy(m, x, y, B, T));
    CHECK_RC(model_forward(m, m->x_dev, B, T, false, 0));
    CHECK_RC(loss(m, m->y_dev, B * T, false));
    CHECK_RC(fetch_metrics(m));
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    if (loss_sum) *loss_sum = m->metrics_host->loss_sum;
    if (correct) *correct = m->metrics_host->correct;
    if (count) *count = (int64_t)B * T;
    return CMP_OK;
}

// presents[layer] = stack([key, value]) with key, value [B,H,T,D] (split_heads of the c_attn output; transformer.py:417-435,
// 797-806): gathered from the qkv activation [B*T, 3E] the last forward pass saved
// (Dk: the head size the kernels run on, D <= Dk the reference's: qkv rows are [3][H][Dk])
template <typename T_>
__global__ void present_gather_kernel(const T_* __restrict__ qkv, float* __restrict__ out, int B, int T, int H, int D, int Dk) {
    const int E = H * Dk;
    const int64_t n = (int64_t)2 * B * H * T * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int t = (int)((i / D) % T);
        const int h = (int)((i / ((int64_t)D * T)) % H);
        const int b = (int)((i / ((int64_t)D * T * H)) % B);
        const int kv = (int)(i / ((int64_t)D * T * H * B));
        out[i] = to_f32<T_>(qkv[((int64_t)b * T + t) * 3 * E + (1 + kv) * E + h * Dk + d]);
    }
}
// Device staging of the inspection entry points (presents, hidden states, the optional inputs of cmp_forward_ex): kept in the
// model and grown on demand -- a hipMalloc / hipFree pair per call is two device-wide synchronisations (hipFree waits for every
// stream of the device: it stalled other models sharing the device, tools/thread_probe.py).  slot: 0 / 1 / 2 independent buffers.
static int io_scratch(cmp_model* m, int slot, size_t bytes, void** out) {
    if (m->io_buf_bytes[slot] < bytes) {
        if (m->io_buf[slot]) {
            HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
            for (auto& p : m->allocs) if (p == m->io_buf[slot]) { p = nullptr; break; }
            (void)hipFree(m->io_buf[slot]);
            m->io_buf[slot] = nullptr; m->io_buf_bytes[slot] = 0;
        }
        void* q = nullptr;
        const size_t want = std::max(bytes, (size_t)1 << 16);
        HIP_CHECK(hipMalloc(&q, want));
        m->allocs.push_back(q);
        m->io_buf[slot] = q; m->io_buf_bytes[slot] = want;
    }
    *out = m->io_buf[slot];
    return CMP_OK;
}
extern "C" int cmp_forward_generation(cmp_model* m, int64_t* gen) {
    CMP_REQUIRE(m && gen, "forward_generation: null argument");
    *gen = m->fwd_gen;
    return CMP_OK;
}
extern "C" int cmp_present_get(cmp_model* m, int layer, int B, int T, float* host_out);
extern "C" int cmp_present_get_at(cmp_model* m, int layer, int B, int T, int64_t generation, float* host_out) {
    CMP_REQUIRE(m, "present_get_at: null argument");
    CMP_REQUIRE(generation == m->fwd_gen, "present_get_at: these presents belong to forward pass %lld, the activations held are those of "
                "pass %lld (a later forward / train step / decode prefill overwrote them; read presents before the next pass)",
                (long long)generation, (long long)m->fwd_gen);
    return cmp_present_get(m, layer, B, T, host_out);
}
extern "C" int cmp_present_get(cmp_model* m, int layer, int B, int T, float* host_out) {
    CMP_REQUIRE(m && host_out, "present_get: null argument");
    CMP_REQUIRE(layer >= 0 && layer < m->L, "present_get: layer %d outside [0, %d)", layer, m->L);
    CMP_REQUIRE(B > 0 && T > 0 && !m->act.empty() && B == m->lastB && T == m->lastT,
                "present_get: the forward pass held is [%d,%d] (batch, past + new positions), not [%d,%d]", m->lastB, m->lastT, B, T);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    const int64_t n = (int64_t)2 * B * m->H * T * m->Dl;
    float* tmp = nullptr;
    CHECK_RC(io_scratch(m, 0, (size_t)n * 4, (void**)&tmp));
    const int grid = (int)std::min<int64_t>(cdiv64(n, 256), 4096);
    if (m->dtype == CMP_BF16) present_gather_kernel<bf16_t><<<grid, 256, 0, m->ctx->stream>>>((const bf16_t*)m->act[layer].qkv, tmp, B, T, m->H, m->Dl, m->D);
    else present_gather_kernel<float><<<grid, 256, 0, m->ctx->stream>>>((const float*)m->act[layer].qkv, tmp, B, T, m->H, m->Dl, m->D);
    KERNEL_CHECK();
    HIP_CHECK(hipMemcpyAsync(host_out, tmp, (size_t)n * 4, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    return CMP_OK;
}

// all_hidden_states[index] of the LAST forward pass (Transformer.call with output_hidden_states, transformer.py:800-816): the
// input of decoder block `index` for index < L (index 0 = the embedding sum after its dropout), the ln_f output for index = L.
template <typename T_>
__global__ void hidden_unpack_kernel(const T_* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = to_f32<T_>(in[i]);
}
extern "C" int cmp_hidden_get_at(cmp_model* m, int index, int B, int T, int64_t generation, float* host_out) {
    CMP_REQUIRE(m && host_out, "hidden_get_at: null argument");
    CMP_REQUIRE(index >= 0 && index <= m->L, "hidden_get_at: index %d outside [0, %d]", index, m->L);
    CMP_REQUIRE(generation == m->fwd_gen, "hidden_get_at: these hidden states belong to forward pass %lld, the activations held are "
                "those of pass %lld (a later forward / train step / decode prefill overwrote them)",
                (long long)generation, (long long)m->fwd_gen);
    CMP_REQUIRE(B > 0 && T > 0 && !m->act.empty() && B == m->lastB && T == m->lastT - m->last_past,
                "hidden_get_at: the forward pass held is [%d,%d] (batch, new positions), not [%d,%d]", m->lastB,
                m->lastT - m->last_past, B, T);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    const int64_t n = (int64_t)B * T * m->E;
    if (index == m->L && !m->hf_valid) {        // the pass folded ln_f into its logits GEMM: the ln_f output is produced now
        CHECK_RC(cmp_k_layernorm_fwd(m->ctx->stream, m->xs[m->L], m->P + m->off_lnf_g, m->P + m->off_lnf_b, m->hf, m->lnf_mean, m->lnf_rstd,
                                     B * T, m->E, m->cfg.ln_eps, m->dtype));
        m->hf_valid = true;
    }
    const void* src = index < m->L ? m->xs[index] : m->hf;
    float* tmp = nullptr;
    CHECK_RC(io_scratch(m, 0, (size_t)n * 4, (void**)&tmp));
    const int grid = (int)std::min<int64_t>(cdiv64(n, 256), 4096);
    if (m->dtype == CMP_BF16) hidden_unpack_kernel<bf16_t><<<grid, 256, 0, m->ctx->stream>>>((const bf16_t*)src, tmp, n);
    else hidden_unpack_kernel<float><<<grid, 256, 0, m->ctx->stream>>>((const float*)src, tmp, n);
    KERNEL_CHECK();
    HIP_CHECK(hipMemcpyAsync(host_out, tmp, (size_t)n * 4, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    return CMP_OK;
}

// host `past` [2, B, H, Tp, D] (fp32) -> K and V columns of rows [0, Tp) of qkv viewed as [B, Tt, 3E]; the Q columns of those
// rows are zeroed (their attention outputs are never read)
// (D: the head size the kernels run on, Dl <= D the reference's = the host tensor's; columns Dl..D-1 are zeroed)
template <typename T_>
__global__ void past_scatter_kernel(const float* __restrict__ in, T_* __restrict__ qkv, int B, int Tp, int Tt, int H, int D, int Dl) {
    const int E = H * D;
    const int64_t n = (int64_t)2 * B * H * Tp * D, nq = (int64_t)B * Tp * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + nq; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n) {
            const int d = (int)(i % D);
            const int t = (int)((i / D) % Tp);
            const int h = (int)((i / ((int64_t)D * Tp)) % H);
            const int b = (int)((i / ((int64_t)D * Tp * H)) % B);
            const int kv = (int)(i / ((int64_t)D * Tp * H * B));
            const float v = d < Dl ? in[((((int64_t)kv * B + b) * H + h) * Tp + t) * Dl + d] : 0.f;
            qkv[((int64_t)b * Tt + t) * 3 * E + (1 + kv) * E + h * D + d] = from_f32<T_>(v);
        } else {
            const int64_t j = i - n;
            const int e = (int)(j % E);
            const int t = (int)((j / E) % Tp);
            const int b = (int)(j / ((int64_t)E * Tp));
            qkv[((int64_t)b * Tt + t) * 3 * E + e] = from_f32<T_>(0.f);
        }
    }
}

__global__ void logits_pack_kernel(const float* __restrict__ z, float* __restrict__ out, int64_t n, int V, int ldz) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = z[(i / V) * ldz + (i % V)];
}

extern "C" int cmp_forward(cmp_model* m, const int32_t* x, int B, int T, int past_len, const float* const* past, int training,
                           float* logits_out) {
    return cmp_forward_ex(m, x, B, T, past_len, past, training, nullptr, nullptr, nullptr, nullptr, logits_out);
}

// (1 - mask) * -10000 (transformer.py:774-779)
__global__ void attention_mask_term_kernel(const int32_t* __restrict__ mask, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (1.0f - (float)mask[i]) * -10000.0f;
}

// position_ids / token_type_ids: host int32 [B*T] or null (transformer.py:770-773, 786-793); attention_mask: host int32
// [B*(past_len+T)] or null (:774-779)
extern "C" int cmp_forward_ex(cmp_model* m, const int32_t* x, int B, int T, int past_len, const float* const* past, int training,
                              const int32_t* position_ids, const int32_t* token_type_ids, const int32_t* attention_mask,
                              float* const* attention_weights_out, float* logits_out) {
    CMP_REQUIRE(m && x && logits_out, "forward: null argument");
    CMP_REQUIRE(past_len >= 0 && (past_len == 0 || past != nullptr), "forward: past_len %d without past tensors", past_len);
    CMP_REQUIRE(T > 0 && T + past_len <= m->W, "forward: positions %d..%d exceed window_size %d (wpe rows, transformer.py:675-679,786)",
                past_len, past_len + T - 1, m->W);
    HIP_CHECK(hipSetDevice(m->ctx->device));
    CHECK_RC(upload_xy(m, x, nullptr, B, T, past_len));
    hipStream_t s = m->ctx->stream;
    // the optional id tensors ride in one device buffer that lives for this call
    struct IdBuf {          // (the buffers themselves live in the model: io_scratch)
        cmp_model* m;
        int32_t* dev = nullptr;
        ~IdBuf() {
            m->fwd_pos_ids = m->fwd_type_ids = nullptr;
            m->fwd_amask = nullptr;
            m->fwd_probs_out = nullptr;
            m->fwd_probs_dev = nullptr;
        }
    } idbuf{m};
    if (attention_weights_out) {
        for (int i = 0; i < m->L; i++) CMP_REQUIRE(attention_weights_out[i], "forward: attention_weights_out[%d] is null", i);
        CHECK_RC(io_scratch(m, 1, (size_t)B * m->H * T * (past_len + T) * 4, (void**)&m->fwd_probs_dev));
        m->fwd_probs_out = attention_weights_out;
    }
    if (position_ids || token_type_ids || attention_mask) {
        const int64_t n = (int64_t)B * T, nk = (int64_t)B * (past_len + T);
        if (position_ids)
            for (int64_t i = 0; i < n; i++)
                CMP_REQUIRE(position_ids[i] >= 0 && position_ids[i] < m->W, "forward: position id %d at %lld outside the wpe table [0, %d)",
                            position_ids[i], (long long)i, m->W);
        if (token_type_ids) CHECK_RC(check_host_ids(m, token_type_ids, n, "token type ids"));
        CHECK_RC(io_scratch(m, 2, (size_t)(2 * n + 2 * nk) * 4, (void**)&idbuf.dev));
        if (position_ids) {
            HIP_CHECK(hipMemcpyAsync(idbuf.dev, position_ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
            m->fwd_pos_ids = idbuf.dev;
        }
        if (token_type_ids) {
            HIP_CHECK(hipMemcpyAsync(idbuf.dev + n, token_type_ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
            m->fwd_type_ids = idbuf.dev + n;
        }
        if (attention_mask) {
            int32_t* raw = idbuf.dev + 2 * n;
            float* term = reinterpret_cast<float*>(idbuf.dev + 2 * n + nk);
            HIP_CHECK(hipMemcpyAsync(raw, attention_mask, (size_t)nk * 4, hipMemcpyHostToDevice, s));
            attention_mask_term_kernel<<<(int)std::min<int64_t>(cdiv64(nk, 256), 1024), 256, 0, s>>>(raw, term, nk);
            KERNEL_CHECK();
            m->fwd_amask = term;
        }
    }
    if (past_len > 0) {
        const int64_t n = (int64_t)2 * B * m->H * past_len * m->Dl;         // the host tensors: [2, B, H, past_len, E / H]
        const int64_t ns = (int64_t)2 * B * m->H * past_len * m->D;        // K/V elements written (zero-padded heads included)
        float* tmp = nullptr;
        CHECK_RC(io_scratch(m, 0, (size_t)n * 4, (void**)&tmp));
        int rc = CMP_OK;
        for (int i = 0; i < m->L && rc == CMP_OK; i++) {
            if (!past[i]) { cmp_set_error("forward: past[%d] is null", i); rc = CMP_ERR_INVALID; break; }
            hipError_t e = hipMemcpyAsync(tmp, past[i], (size_t)n * 4, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { cmp_set_error("forward: uploading past[%d]: %s", i, hipGetErrorString(e)); rc = CMP_ERR_HIP; break; }
            const int grid = (int)std::min<int64_t>(cdiv64(ns + (int64_t)B * past_len * m->Ea, 256), 4096);
            if (m->dtype == CMP_BF16) past_scatter_kernel<bf16_t><<<grid, 256, 0, s>>>(tmp, (bf16_t*)m->act[i].qkv, B, past_len, past_len + T, m->H, m->D, m->Dl);
            else past_scatter_kernel<float><<<grid, 256, 0, s>>>(tmp, (float*)m->act[i].qkv, B, past_len, past_len + T, m->H, m->D, m->Dl);
            e = hipStreamSynchronize(s);          // tmp is reused by the next layer's upload
            if (e != hipSuccess) { cmp_set_error("forward: past scatter: %s", hipGetErrorString(e)); rc = CMP_ERR_HIP; }
        }
        CHECK_RC(rc);
    }
    CHECK_RC(model_forward(m, m->x_dev, B, T, training != 0, m->iterations, past_len));
    // [tokens, ldz] -> [tokens, V] on the device, then ONE contiguous copy: hipMemcpy2DAsync to pageable host memory left 2 MiB of
    // device memory behind per context (tools/leak_probe.py)
    if (m->ldz == m->V) {
        HIP_CHECK(hipMemcpyAsync(logits_out, m->logits, (size_t)B * T * m->V * 4, hipMemcpyDeviceToHost, s));
    } else {
        if (!m->logits_pack) CHECK_RC(dev_alloc(m, &m->logits_pack, (size_t)m->capB * m->capT * m->V * 4));
        const int64_t n = (int64_t)B * T * m->V;
        logits_pack_kernel<<<(int)std::min<int64_t>(cdiv64(n, 256), 4096), 256, 0, s>>>(m->logits, m->logits_pack, n, m->V, m->ldz);
        KERNEL_CHECK();
        HIP_CHECK(hipMemcpyAsync(logits_out, m->logits_pack, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_CHECK(hipStreamSynchronize(s));
    return CMP_OK;
}

extern "C" int cmp_forward_logits(cmp_model* m, const int32_t* x, int B, int T, float* logits_out) {
    return cmp_forward(m, x, B, T, 0, nullptr, 0, logits_out);
}
