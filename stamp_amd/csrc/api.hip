// api.hip -- error plumbing, device query and the GEMM entry points of libamdstamp.
#include "gemm_kernel.h"
#include <stdlib.h>
#include <atomic>
#include <mutex>

namespace amds {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return AMDS_ERR_HIP;
}

// ---- per-device context: owns the live profiler and the side stream of the overlapped tile-encoder schedule -------------------------
}  // namespace amds
struct amds_ctx {
    int device = -1;
    int refs = 0;
    std::mutex mu;
    bool prof_on = false;
    amds::ProfRec* recs = nullptr;
    int nrec = 0, nalloc = 0;
    long dropped = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // settings of THIS context (round 6: they were process-wide atomics -- two trainers in one process with different torch flags shared one value)
    std::atomic<int> matmul_precision{AMDS_MATMUL_HIGHEST};      // amds_set_matmul_precision
    std::atomic<int> mil_cls_tail{-1};                           // amds_set_mil_cls_tail; -1 = the AMDS_MIL_CLS_TAIL environment default
};
namespace amds {
std::atomic<int> g_prof_any{0};
namespace {
constexpr int MAX_DEV = 64, PROF_MAX = 1 << 15;
amds_ctx* g_ctx[MAX_DEV] = {};
std::mutex g_tab;
amds_ctx* current_ctx() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_tab);
    return g_ctx[dev];
}
}  // namespace
// A launch is timed when the context of the CALLING THREAD'S CURRENT DEVICE has its profiler on.  Slots are handed out under the
// context's mutex, so concurrent host threads cannot corrupt each other's records.
int prof_begin(int kind, double work, hipStream_t st, amds_ctx** ctx_out) {
    amds_ctx* c = current_ctx();
    if (!c) return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->prof_on) return -1;
    if (c->nrec >= PROF_MAX) { ++c->dropped; return -1; }
    if (!c->recs) c->recs = (ProfRec*)calloc(PROF_MAX, sizeof(ProfRec));
    if (!c->recs) return -1;
    const int slot = c->nrec;
    if (slot >= c->nalloc) {
        if (hipEventCreate(&c->recs[slot].a) != hipSuccess || hipEventCreate(&c->recs[slot].b) != hipSuccess) { ++c->dropped; return -1; }
        c->nalloc = slot + 1;
    }
    c->recs[slot].kind = kind;
    c->recs[slot].work = work;
    c->recs[slot].closed = false;
    (void)hipEventRecord(c->recs[slot].a, st);
    ++c->nrec;
    *ctx_out = c;
    return slot;
}
void prof_end(amds_ctx* c, int slot, hipStream_t st) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (slot >= c->nrec) return;          // the profiler was reset in between
    (void)hipEventRecord(c->recs[slot].b, st);
    c->recs[slot].closed = true;
}

// default tile configuration: AMDS_GEMM_CFG overrides (tuning), else by shape
int default_gemm_cfg(int M, int N, int K) {
    static int env = -2;
    if (env == -2) {
        const char* s = getenv("AMDS_GEMM_CFG");
        env = (s && *s) ? atoi(s) : -1;          // set-but-empty counts as unset
    }
    if (env >= 0) return env;
    // the staggered 256x256 pipeline wins whenever the grid fills the chip; small problems keep 128x128 tiles
    if (N % 256 == 0 && (long)cdiv(M, 256) * (N / 256) >= 192) return 8;
    return 0;
}

}  // namespace amds

using namespace amds;

extern "C" int amds_version(void) { return AMDS_VERSION_MAJOR * 100 + AMDS_VERSION_MINOR; }
extern "C" const char* amds_last_error(void) { return g_err; }

extern "C" int amds_device_info(int device, char* name_host, int n, int* cu_count_host, size_t* hbm_bytes_host) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= device || device < 0) {
        set_error("amds_device_info: no HIP device %d (count %d)", device, cnt);
        return AMDS_ERR_NODEVICE;
    }
    hipDeviceProp_t p;
    AMDS_HIP(hipGetDeviceProperties(&p, device));
    if (name_host && n > 0) {
        strncpy(name_host, p.gcnArchName, (size_t)n - 1);
        name_host[n - 1] = 0;
    }
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    if (hbm_bytes_host) *hbm_bytes_host = p.totalGlobalMem;
    return AMDS_OK;
}

extern "C" amds_ctx* amds_create(int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || device < 0 || device >= cnt || device >= MAX_DEV) {
        set_error("amds_create: no HIP device %d (count %d)", device, cnt);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_tab);
    if (!g_ctx[device]) {
        g_ctx[device] = new amds_ctx();
        g_ctx[device]->device = device;
    }
    ++g_ctx[device]->refs;            // one context per device and process; further calls share it
    return g_ctx[device];
}

extern "C" void amds_destroy(amds_ctx* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(g_tab);
        if (--c->refs > 0) return;
        if (c->device >= 0 && c->device < MAX_DEV && g_ctx[c->device] == c) g_ctx[c->device] = nullptr;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();          // the only place the library synchronises the device
    if (c->prof_on) g_prof_any.fetch_sub(1);
    for (int i = 0; i < c->nalloc; ++i) { (void)hipEventDestroy(c->recs[i].a); (void)hipEventDestroy(c->recs[i].b); }
    free(c->recs);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
}

extern "C" int amds_ctx_device(const amds_ctx* c) { return c ? c->device : -1; }

// the context of the calling thread's current device, if the host created one (amds_create); nullptr otherwise
amds_ctx* amds::ctx_of_current_device() { return current_ctx(); }

// ---- per-context settings: read through the context of the calling thread's current device (the one its streams belong to); a process that never
// created a context for the device gets the defaults
static int cls_tail_env_default() {
    static const int v = (getenv("AMDS_MIL_CLS_TAIL") && atoi(getenv("AMDS_MIL_CLS_TAIL")) == 0) ? 0 : 1;
    return v;
}
int amds::ctx_matmul_precision() {
    amds_ctx* c = current_ctx();
    return c ? c->matmul_precision.load(std::memory_order_relaxed) : AMDS_MATMUL_HIGHEST;
}
int amds::ctx_mil_cls_tail() {
    amds_ctx* c = current_ctx();
    const int v = c ? c->mil_cls_tail.load(std::memory_order_relaxed) : -1;
    return v < 0 ? cls_tail_env_default() : v;
}
// multiprocessor count of the current device, looked up once per DEVICE (round 5 kept one process-wide number per kernel family)
int amds::device_cu_count() {
    static std::atomic<int> cus[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        v = p.multiProcessorCount;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int amds_set_matmul_precision(amds_ctx* ctx, int level) {
    AMDS_REQUIRE(ctx, "amds_set_matmul_precision: null context");
    AMDS_REQUIRE(level == AMDS_MATMUL_HIGHEST || level == AMDS_MATMUL_HIGH, "amds_set_matmul_precision: level must be AMDS_MATMUL_HIGHEST (0) or AMDS_MATMUL_HIGH (1), got %d", level);
    ctx->matmul_precision.store(level, std::memory_order_relaxed);
    return AMDS_OK;
}
extern "C" int amds_get_matmul_precision(amds_ctx* ctx) { return ctx ? ctx->matmul_precision.load(std::memory_order_relaxed) : AMDS_MATMUL_HIGHEST; }
extern "C" int amds_set_mil_cls_tail(amds_ctx* ctx, int on) {
    AMDS_REQUIRE(ctx, "amds_set_mil_cls_tail: null context");
    ctx->mil_cls_tail.store(on ? 1 : 0, std::memory_order_relaxed);
    return AMDS_OK;
}
extern "C" int amds_get_mil_cls_tail(amds_ctx* ctx) {
    const int v = ctx ? ctx->mil_cls_tail.load(std::memory_order_relaxed) : -1;
    return v < 0 ? cls_tail_env_default() : v;
}

// side stream + fork / join events of the overlapped schedule, created on first use on the context's device
int amds::ctx_side_stream(amds_ctx* c, hipStream_t* side, hipEvent_t* ev_in, hipEvent_t* ev_out) {
    AMDS_REQUIRE(c, "null context");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->side) {
        int dev = -1;
        AMDS_HIP(hipGetDevice(&dev));
        AMDS_REQUIRE(dev == c->device, "the calling thread's current device is %d, the context belongs to device %d", dev, c->device);
        AMDS_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        AMDS_HIP(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        AMDS_HIP(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    *side = c->side; *ev_in = c->ev_in; *ev_out = c->ev_out;
    return AMDS_OK;
}

extern "C" int amds_profile_enable(amds_ctx* c, int on) {
    AMDS_REQUIRE(c, "amds_profile_enable: null context");
    std::lock_guard<std::mutex> lk(c->mu);
    const bool want = on != 0;
    if (want != c->prof_on) g_prof_any.fetch_add(want ? 1 : -1);
    c->prof_on = want;
    return AMDS_OK;
}
extern "C" int amds_profile_reset(amds_ctx* c) {
    AMDS_REQUIRE(c, "amds_profile_reset: null context");
    std::lock_guard<std::mutex> lk(c->mu);
    c->nrec = 0; c->dropped = 0;
    return AMDS_OK;
}
extern "C" int amds_profile_read(amds_ctx* c, int kind, double* total_ms_host, long* launches_host, double* total_work_host) {
    AMDS_REQUIRE(c, "amds_profile_read: null context");
    AMDS_REQUIRE(kind >= 0 && kind < PROF_NKINDS, "amds_profile_read: bad kind %d", kind);
    std::lock_guard<std::mutex> lk(c->mu);
    double ms = 0, work = 0;
    long n = 0;
    for (int i = 0; i < c->nrec; ++i) {
        if (c->recs[i].kind != kind || !c->recs[i].closed) continue;
        AMDS_HIP(hipEventSynchronize(c->recs[i].b));
        float t = 0.f;
        AMDS_HIP(hipEventElapsedTime(&t, c->recs[i].a, c->recs[i].b));
        ms += t; work += c->recs[i].work; ++n;
    }
    if (total_ms_host) *total_ms_host = ms;
    if (launches_host) *launches_host = n;
    if (total_work_host) *total_work_host = work;
    return AMDS_OK;
}

static int gemm_impl(int cfg, const void* A, long lda, const void* W, long ldw, int M, int N, int K, int dtype, int epi,
                     void* out, long ldo, const float* bias, const float* scale, const float* pos, int np, int T,
                     int P, float acc_scale, void* stream) {
    AMDS_REQUIRE(A && W && out, "amds_gemm: null pointer");
    AMDS_REQUIRE(M >= 0 && N > 0 && K > 0, "amds_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    AMDS_REQUIRE(K % 64 == 0, "amds_gemm: K=%d must be a multiple of 64 (zero-pad with amds_cast_pad)", K);
    AMDS_REQUIRE(N % 128 == 0 || N % 96 == 0, "amds_gemm: N=%d must be a multiple of 128 or of 96 (zero-pad the weight rows)", N);
    AMDS_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "amds_gemm: lda=%ld/ldw=%ld must be >= K and multiples of 8", lda, ldw);
    AMDS_REQUIRE(ldo % 4 == 0, "amds_gemm: ldo=%ld must be a multiple of 4", ldo);
    AMDS_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0, "amds_gemm: pointers must be 16-byte aligned");
    if (epi == AMDS_EPI_SWIGLU) AMDS_REQUIRE(bias != nullptr, "amds_gemm: SWIGLU epilogue needs a bias");
    if (epi == AMDS_EPI_PATCH) AMDS_REQUIRE(pos && np > 0 && T >= np + P && P >= 0, "amds_gemm: PATCH epilogue needs pos/np/T/P");
    if (M == 0) return AMDS_OK;
    EpiArgs ep;
    ep.out = out; ep.ldo = ldo; ep.bias = bias; ep.scale = scale; ep.pos = pos;
    ep.np = np; ep.T = T; ep.P = P; ep.acc_scale = acc_scale;
    // cfg -2 = -1 (by shape) plus permission to run a ragged last row tile as its own launch (below): the rows of that tile then come from
    // another kernel than their neighbours, so only callers that do not promise bit-identical rows across batch compositions ask for it
    static const bool split_on = getenv("AMDS_GEMM_SPLIT") ? atoi(getenv("AMDS_GEMM_SPLIT")) != 0 : true;       // 0: never split (A/B)
    const bool may_split = cfg == -2 && split_on;
    if (cfg < 0) {
        cfg = default_gemm_cfg(M, N, K);
        // measured (profiles/r01_gemm_vendor_and_power.txt): at the socket power cap the four-wave kernel on 16x16x32 MFMAs (the cheaper
        // instruction per flop) wins on every staged epilogue and on SWIGLU; PATCH keeps the eight-wave kernel
        if (cfg == 8 && epi != AMDS_EPI_PATCH && N % 256 == 0 && !(getenv("AMDS_GEMM_CFG") && *getenv("AMDS_GEMM_CFG"))) cfg = 12;
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * M * (double)N * K, st);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_gemm: bad dtype %d", dtype);
    auto run = [&](int c, const void* a, int m, const EpiArgs& e) {
        return dtype == AMDS_F16 ? gemm_dispatch<f16>(c, epi, a, lda, W, ldw, m, N, K, e, st) : gemm_dispatch<bf16>(c, epi, a, lda, W, ldw, m, N, K, e, st);
    };
    // Ragged M on the 256-row kernel: M = 64 bags x 1025 tokens = 256.25 row tiles puts 2 ... 8 workgroups into a wave of their own (N = 512: 514
    // workgroups = three waves for the work of two).  When dropping the last, partial row tile saves a whole wave, its rows (<= 128) go through the
    // 128 x 128 kernel as a second launch instead.
    if (may_split && cfg == 12 && epi != AMDS_EPI_PATCH) {
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t p;
            AMDS_HIP(hipGetDevice(&dev));
            AMDS_HIP(hipGetDeviceProperties(&p, dev));
            cus = p.multiProcessorCount;
        }
        const int full = M / 256, rem = M - full * 256;
        const long ct = N / 256;
        if (full > 0 && rem > 0 && rem <= 128 && cdiv((full + 1) * ct, (long)cus) > cdiv(full * ct, (long)cus)) {
            const bool f32_out = epi == AMDS_EPI_RESIDUAL || epi == AMDS_EPI_BIAS_F32 || epi == AMDS_EPI_BIAS_GELU_F32 || epi == AMDS_EPI_BIAS_RELU_F32;
            const int rc = run(12, A, full * 256, ep);
            if (rc != AMDS_OK) return rc;
            EpiArgs ep2 = ep;
            ep2.out = (char*)out + (size_t)full * 256 * ldo * (f32_out ? 4 : 2);
            return run(0, (const char*)A + (size_t)full * 256 * lda * 2, rem, ep2);
        }
    }
    return run(cfg, A, M, ep);
}

extern "C" int amds_gemm(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int dtype, int epi,
                         void* out, long ldo, const float* bias, const float* scale, const float* pos, int np, int T,
                         int P, float acc_scale, void* stream) {
    return gemm_impl(-1, A, lda, W, ldw, M, N, K, dtype, epi, out, ldo, bias, scale, pos, np, T, P, acc_scale, stream);
}

// Batched / split-K form of amds_gemm on the production kernel: batch b uses A + b*bsA, W + b*bsW, out + b*bsOut
// (element strides).  Split-K for weight gradients: bsA = bsW = K (advance along the contraction), bsOut = M*N partials.
extern "C" int amds_gemm_batched(const void* A, long lda, long bsA, const void* W, long ldw, long bsW, int M, int N, int K,
                                 int nbatch, int dtype, int epi, void* out, long ldo, long bsOut, const float* bias,
                                 float acc_scale, void* stream) {
    AMDS_REQUIRE(A && W && out, "amds_gemm_batched: null pointer");
    AMDS_REQUIRE(M > 0 && N > 0 && K > 0 && nbatch > 0 && nbatch <= 65535, "amds_gemm_batched: bad shape");
    AMDS_REQUIRE(K % 64 == 0 && N % 256 == 0, "amds_gemm_batched: needs K %% 64 == 0 and N %% 256 == 0 (K=%d N=%d)", K, N);
    AMDS_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && bsA % 8 == 0 && bsW % 8 == 0 && ldo % 4 == 0 && bsOut % 8 == 0, "amds_gemm_batched: strides must keep 16-byte alignment");
    AMDS_REQUIRE(epi == AMDS_EPI_BIAS || epi == AMDS_EPI_BIAS_F32, "amds_gemm_batched: only the BIAS / BIAS_F32 epilogues");
    EpiArgs ep;
    ep.out = out; ep.ldo = ldo; ep.bias = bias; ep.scale = nullptr; ep.pos = nullptr; ep.np = ep.T = ep.P = 0; ep.acc_scale = acc_scale;
    ep.bsA = bsA; ep.bsW = bsW; ep.bsOut = bsOut; ep.nbatch = nbatch;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * nbatch * M * (double)N * K, st);
    static int bcfg = -1;
    if (bcfg < 0) {
        const char* e = getenv("AMDS_GEMM_CFG");
        bcfg = (e && *e && atoi(e) == 8) ? 8 : 12;          // AMDS_GEMM_CFG=8: the eight-wave kernel (A/B)
    }
    if (dtype == AMDS_F16) return gemm_dispatch<f16>(bcfg, epi, A, lda, W, ldw, M, N, K, ep, st);
    if (dtype == AMDS_BF16) return gemm_dispatch<bf16>(bcfg, epi, A, lda, W, ldw, M, N, K, ep, st);
    set_error("amds_gemm_batched: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

// Weight-gradient partials straight from token-major operands (kernel id 15, gemm_4w16.h TN): part[s][n][k] = sum over the tokens t of split s of
// dy[t][n] * x[t][k].  No transposed copies, no padded operands: token rows past `tokens` read as zeros through the buffer descriptors.
extern "C" int amds_wgrad_tn(const void* dy, long ld_dy, const void* x, long ld_x, long tokens, int N, int K, int split_k, int dtype, float* part,
                             void* stream) {
    AMDS_REQUIRE(dy && x && part, "amds_wgrad_tn: null pointer");
    AMDS_REQUIRE(tokens > 0 && N > 0 && K > 0 && split_k > 0 && split_k <= 65535, "amds_wgrad_tn: bad shape");
    AMDS_REQUIRE(N % 256 == 0 && K % 256 == 0, "amds_wgrad_tn: needs N %% 256 == 0 and K %% 256 == 0 (N=%d K=%d)", N, K);
    AMDS_REQUIRE(ld_dy % 8 == 0 && ld_x % 8 == 0 && ld_dy >= N && ld_x >= K, "amds_wgrad_tn: bad pitches");
    AMDS_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)part & 15) == 0, "amds_wgrad_tn: pointers must be 16-byte aligned");
    const long unit = 64L * split_k;
    const long chunk = (tokens + unit - 1) / unit * 64;                 // tokens per split, a multiple of the K tile
    // every split has its own descriptor base (bsA / bsW below): only one split's span has to fit the 2 GB a buffer descriptor addresses
    AMDS_REQUIRE((chunk + 63) * std::max(ld_dy, ld_x) * 2 < (1L << 31), "amds_wgrad_tn: one split's operand span is beyond the 2 GB a buffer descriptor addresses (raise split_k)");
    EpiArgs ep;
    ep.out = part; ep.ldo = K; ep.bias = nullptr; ep.scale = nullptr; ep.pos = nullptr; ep.np = ep.T = ep.P = 0; ep.acc_scale = 1.0f;
    ep.bsA = chunk * ld_dy; ep.bsW = chunk * ld_x; ep.bsOut = (long)N * K; ep.nbatch = split_k; ep.ktot = tokens;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * tokens * (double)N * K, st);
    if (dtype == AMDS_F16) return gemm_dispatch<f16>(15, AMDS_EPI_BIAS_F32, dy, ld_dy, x, ld_x, N, K, (int)chunk, ep, st);
    if (dtype == AMDS_BF16) return gemm_dispatch<bf16>(15, AMDS_EPI_BIAS_F32, dy, ld_dy, x, ld_x, N, K, (int)chunk, ep, st);
    set_error("amds_wgrad_tn: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

// LayerNorm folded into the GEMMs around it (production kernel only): see include/amdstamp.h
extern "C" int amds_gemm_lnfold(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int dtype, int epi, void* out,
                                long ldo, const float* bias, const float* scale, void* xh, float* rowpart, const float* rowstat,
                                const float* colsum, void* stream) {
    AMDS_REQUIRE(A && W && out, "amds_gemm_lnfold: null pointer");
    AMDS_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 64 == 0 && N % 256 == 0, "amds_gemm_lnfold: needs K %% 64 == 0 and N %% 256 == 0 (M=%d N=%d K=%d)", M, N, K);
    AMDS_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ldo % 4 == 0, "amds_gemm_lnfold: bad pitches");
    AMDS_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)xh & 15) == 0,
                 "amds_gemm_lnfold: pointers must be 16-byte aligned");
    const bool producer = xh || rowpart, consumer = rowstat || colsum;
    AMDS_REQUIRE(producer != consumer, "amds_gemm_lnfold: either (xh, rowpart) [RESIDUAL] or (rowstat, colsum) [BIAS / BIAS_GELU / SWIGLU]");
    if (producer) {
        AMDS_REQUIRE(epi == AMDS_EPI_RESIDUAL && xh && rowpart, "amds_gemm_lnfold: the producer form is the RESIDUAL epilogue with xh AND rowpart");
    } else {
        AMDS_REQUIRE((epi == AMDS_EPI_BIAS || epi == AMDS_EPI_BIAS_GELU || epi == AMDS_EPI_SWIGLU) && rowstat && colsum && bias,
                     "amds_gemm_lnfold: the consumer form is BIAS / BIAS_GELU / SWIGLU with rowstat, colsum AND bias");
    }
    if (M == 0) return AMDS_OK;
    EpiArgs ep;
    ep.out = out; ep.ldo = ldo; ep.bias = bias; ep.scale = scale; ep.pos = nullptr; ep.np = ep.T = ep.P = 0; ep.acc_scale = 1.0f;
    ep.xh = xh; ep.rowpart = rowpart; ep.rowstat = rowstat; ep.colsum = colsum;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * M * (double)N * K, st);
    if (dtype == AMDS_F16) return gemm_dispatch<f16>(12, epi, A, lda, W, ldw, M, N, K, ep, st);
    if (dtype == AMDS_BF16) return gemm_dispatch<bf16>(12, epi, A, lda, W, ldw, M, N, K, ep, st);
    set_error("amds_gemm_lnfold: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

// The RESIDUAL producer with the residual stream held as two fp16 planes (include/amdstamp.h): x = hi + lo;  x += scale * (A W^T + bias);
// hi = fp16(x), lo = fp16(x - hi), rowpart = partial (sum, sum of squares) of the fp32 x.  In place; no fp32 rows exist.
extern "C" int amds_gemm_lnfold_planes(const void* A, long lda, const void* W, long ldw, int M, int N, int K, void* hi, void* lo, long ld,
                                       const float* bias, const float* scale, float* rowpart, void* stream) {
    AMDS_REQUIRE(A && W && hi && lo && rowpart, "amds_gemm_lnfold_planes: null pointer");
    AMDS_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 64 == 0 && N % 256 == 0, "amds_gemm_lnfold_planes: needs K %% 64 == 0 and N %% 256 == 0 (M=%d N=%d K=%d)", M, N, K);
    AMDS_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ld % 4 == 0 && ld >= N, "amds_gemm_lnfold_planes: bad pitches");
    AMDS_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)hi & 15) == 0 && ((uintptr_t)lo & 15) == 0,
                 "amds_gemm_lnfold_planes: pointers must be 16-byte aligned");
    AMDS_REQUIRE(hi != lo && A != hi && A != lo, "amds_gemm_lnfold_planes: the planes are updated in place and must not alias each other or A");
    if (M == 0) return AMDS_OK;
    EpiArgs ep;
    ep.out = hi; ep.ldo = ld; ep.bias = bias; ep.scale = scale; ep.pos = nullptr; ep.np = ep.T = ep.P = 0; ep.acc_scale = 1.0f;
    ep.xh = hi; ep.xl = lo; ep.rowpart = rowpart;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * M * (double)N * K, st);
    return gemm_dispatch<f16>(12, AMDS_EPI_RESIDUAL, A, lda, W, ldw, M, N, K, ep, st);
}

// tuning hook: explicit kernel id (see gemm_kernel.h)
extern "C" int amds_gemm_ex(int cfg, const void* A, long lda, const void* W, long ldw, int M, int N, int K, int dtype,
                            int epi, void* out, long ldo, const float* bias, const float* scale, const float* pos,
                            int np, int T, int P, float acc_scale, void* stream) {
    return gemm_impl(cfg, A, lda, W, ldw, M, N, K, dtype, epi, out, ldo, bias, scale, pos, np, T, P, acc_scale, stream);
}
