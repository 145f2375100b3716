// mil_vit_train.hip -- the TRAINING forward and backward of the MIL `vit` head, one call each (amds_mil_vit_fwd / _bwd).
//
// Forward = the train-mode forward of the reference's VisionTransformer (src/stamp/modeling/models/vision_tranformer.py:332-384 with the
// Dropout sites of :157-169, :191, :314-318 live); backward = what autograd derives from it (the reference calls loss.backward() through
// Lightning, src/stamp/modeling/models/__init__.py:239-279).  Both are launch sequences over the kernels the library exposes one by one
// (include/amdstamp.h, "MIL training step"): 16-bit MFMA operands -- bf16 (8 mantissa bits: torch's float32_matmul_precision "medium") or fp16 (11: the TF32 class torch's
// "high" asks for, which the reference sets before training, train.py:519; the caller scales the loss, see stamp_amd/mil_train.py) by cfg.dtype -- fp32 accumulation /
// residual stream / gradients.  The activations the
// backward needs live in ONE caller-owned arena (`saved`), gradients are written in the PADDED weight layout into caller-owned fp32
// buffers (amds_mil_vit_grads); the ALiBi running-mean update (:24-29) happens before the forward and stays with the caller.
// Nothing is allocated and the host never waits for the device.
#include <algorithm>
#include <vector>
#include "common.h"

namespace amds {
namespace {

inline int up(int n, int m) { return (n + m - 1) / m * m; }
inline long upl(long n, long m) { return (n + m - 1) / m * m; }
inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

struct LayerOff {
    size_t h1, mu1, rs1, qkv, att, lse, u_al, osm, x_mid, h2, mu2, rs2, z, u;
};

struct Dims {
    int F, D, H, FF, C, L, alibi;
    int Fp, Dp, FFp, Ha, Da, S, Bb, Tn;
    long M, Mt;
    int dt;          // AMDS_BF16 or AMDS_F16: the type of every 16-bit tensor of the step (weights' operand copies, saved activations, 16-bit gradients)
};

struct SavedPlan {
    size_t a, zp, xp, cc, y, x0, x_bytes, clsn, muf, rsf, total;
    std::vector<LayerOff> layer;
};

int make_dims(const amds_mil_vit_cfg* c, int Bb, int Tn, Dims* d) {
    AMDS_REQUIRE(c, "amds_mil_vit_train: null config");
    AMDS_REQUIRE(c->n_feats > 0 && c->dim > 0 && c->heads > 0 && c->ff > 0 && c->classes > 0 && c->layers >= 0 && c->layers <= 1024,
                 "amds_mil_vit_train: bad config");
    AMDS_REQUIRE(c->dim % c->heads == 0, "amds_mil_vit_train: dim_model=%d has to be divisible by n_heads=%d", c->dim, c->heads);
    AMDS_REQUIRE(c->dim / c->heads <= 64 && c->dim % 4 == 0, "amds_mil_vit_train: needs head_dim <= 64 and dim_model %% 4 == 0 (dim_model=%d, n_heads=%d)",
                 c->dim, c->heads);
    AMDS_REQUIRE(c->dtype == AMDS_BF16 || c->dtype == AMDS_F16, "amds_mil_vit_train: the training step runs on bf16 or fp16 operands (cfg.dtype = AMDS_BF16 / AMDS_F16)");
    d->dt = c->dtype;
    AMDS_REQUIRE(Bb > 0 && Tn > 0, "amds_mil_vit_train: bad shape bags=%d tiles=%d", Bb, Tn);
    d->F = c->n_feats; d->D = c->dim; d->H = c->heads; d->FF = c->ff; d->C = c->classes; d->L = c->layers; d->alibi = c->alibi != 0;
    d->Fp = up(d->F, 256); d->Dp = up(d->D, 256); d->FFp = up(d->FF, 256); d->Ha = up(d->H, 4); d->Da = 64 * d->Ha;
    d->S = Tn + 1; d->Bb = Bb; d->Tn = Tn;
    d->M = (long)Bb * d->S; d->Mt = (long)Bb * Tn;
    AMDS_REQUIRE(d->M < (1L << 31) - 65536, "amds_mil_vit_train: %ld token rows do not fit the 32-bit row index", d->M);
    return AMDS_OK;
}

void plan_saved(const Dims& d, SavedPlan* p) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t M = d.M, Mt = d.Mt;
    p->a = take(Mt * d.Fp * 2);
    p->zp = take(Mt * d.Dp * 2);
    p->xp = take(Mt * d.Dp * 4);
    p->cc = take(M * 2 * 4);
    p->y = take(M * d.Dp * 4);
    p->x_bytes = al(M * d.Dp * 4);
    p->x0 = take(p->x_bytes * (d.L + 1));
    p->layer.resize(d.L);
    for (int l = 0; l < d.L; ++l) {
        LayerOff& o = p->layer[l];
        o.h1 = take(M * d.Dp * 2); o.mu1 = take(M * 4); o.rs1 = take(M * 4);
        o.qkv = take(M * 3 * d.Da * 2); o.att = take(M * d.Da * 2); o.lse = take((size_t)d.Bb * d.Ha * d.S * 4);
        o.u_al = d.alibi ? take(M * d.Da * 2) : 0; o.osm = d.alibi ? take(M * d.Da * 2) : 0;
        o.x_mid = take(M * d.Dp * 4);
        o.h2 = take(M * d.Dp * 2); o.mu2 = take(M * 4); o.rs2 = take(M * 4);
        o.z = take(M * d.FFp * 2); o.u = take(M * d.FFp * 2);
    }
    p->clsn = take((size_t)d.Bb * d.D * 4);
    p->muf = take((size_t)d.Bb * 4);
    p->rsf = take((size_t)d.Bb * 4);
    p->total = off;
}

struct WsPlan {
    long Mp, Mtp;
    size_t dx, dh, g16, du, dz, datt, dqkv, tg, ta, part, part_all, cs, lnb, sums, dqs, dbsp, dbst, gsc, dcls, dlt, dxp, dzp, total;
    size_t cs_bytes, lnb_bytes, sums_bytes;
};

void plan_ws(const Dims& d, int split_k, WsPlan* p) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t M = d.M, Mt = d.Mt;
    const long unit = 64L * split_k;
    p->Mp = upl(d.M, unit);
    p->Mtp = upl(d.Mt, unit);
    const size_t Mpp = (size_t)(p->Mp > p->Mtp ? p->Mp : p->Mtp);
    const int wg = std::max(std::max(3 * d.Da, d.FFp), d.Dp);                       // widest gradient matrix that gets transposed
    const int wa = std::max(std::max(std::max(d.FFp, d.Dp), d.Da), d.Fp);           // widest activation matrix
    p->dx = take(M * d.Dp * 4);
    p->dh = take(M * d.Dp * 4);
    p->g16 = take(M * d.Dp * 2);
    p->du = take(M * d.FFp * 2);
    p->dz = take(M * d.FFp * 2);
    p->datt = take(M * d.Da * 2);
    p->dqkv = take(M * 3 * d.Da * 2);
    p->tg = take((size_t)wg * Mpp * 2);
    p->ta = take((size_t)wa * Mpp * 2);
    const size_t nk = std::max(std::max((size_t)3 * d.Da * d.Dp, (size_t)d.FFp * d.Dp), std::max((size_t)d.Dp * d.Da, (size_t)d.Dp * d.Fp));
    p->part = take(nk * split_k * 4);
    // one region of split-K partials per weight gradient: they are all summed by ONE launch behind the last of them (amds_sum_partials_multi)
    const size_t all_w = (size_t)d.Dp * d.Fp + (size_t)d.L * ((size_t)3 * d.Da * d.Dp + (size_t)d.Dp * d.Da + 2 * (size_t)d.FFp * d.Dp);
    p->part_all = take(all_w * split_k * 4);
    size_t cs = amds_colsum_workspace_bytes(split_k, (int)std::min<size_t>(nk, 0x7fffffff));
    const int widths[] = {d.Dp, d.FFp, 3 * d.Da, d.Ha, d.C};
    for (int w : widths) cs = std::max(cs, amds_colsum_workspace_bytes((int)d.M, w));
    p->cs_bytes = std::max<size_t>(cs, 4);
    p->cs = take(p->cs_bytes);
    p->lnb_bytes = std::max<size_t>(amds_layernorm_bwd_workspace_bytes((int)d.M, d.D), 4);
    p->lnb = take(p->lnb_bytes);
    // the column sums the backward postpones to ONE launch at its end (amds_colsum_multi): per-64-row partials of every LayerNorm's parameter gradients
    // and the chunk partials of every bias gradient, each in its own region
    {
        const size_t nblk = (M + 63) / 64, nch = (M + 1023) / 1024 + 1;
        size_t b = (size_t)(2 * d.L + 1) * 2 * al(nblk * d.D * 4);
        b += (size_t)d.L * (2 * al(nch * d.Dp * 4) + al(nch * d.FFp * 4) + al(nch * 3 * d.Da * 4) + al(nch * d.Ha * 4)) + al(nch * d.Dp * 4);
        p->sums_bytes = b + 4096;
        p->sums = take(p->sums_bytes);
    }
    p->dqs = take((size_t)d.Bb * d.Ha * d.S * 4);
    p->dbsp = take((size_t)d.Bb * d.Ha * d.S * 4);
    p->dbst = take(M * d.Ha * 4);
    p->gsc = take((size_t)2 * d.D * 4);                 // LayerNorm parameter gradients nobody asked for (need_params = false)
    p->dcls = take((size_t)d.Bb * d.D * 4);
    p->dlt = take((size_t)d.Bb * d.C * 4);
    p->dxp = take(Mt * d.Dp * 4);
    p->dzp = take(Mt * d.Dp * 2);
    p->total = off;
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) stage_bags_bf16_kernel(const TI* __restrict__ src, long ld_src, TO* __restrict__ dst, int Fp, long total, int F) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / Fp;
        const int c = (int)(i - r * Fp);
        dst[i] = c < F ? (TO)(float)src[r * ld_src + c] : (TO)0.f;
    }
}

// x rows: class token in front of each bag's projected tiles; coords with the class token at (0, 0) (:347-351).  One block per row.
__global__ void __launch_bounds__(128) train_prefix_cls_kernel(const float* __restrict__ cls, const float* __restrict__ proj, float* __restrict__ x, int Dp,
                                                                const float* __restrict__ coords, float* __restrict__ coords_out, int Tn) {
    const int S = Tn + 1;
    const long row = blockIdx.x;
    const long b = row / S;
    const int s = (int)(row - b * S);
    const f32x4* src = reinterpret_cast<const f32x4*>(s == 0 ? cls : proj + (b * Tn + s - 1) * Dp);
    f32x4* dst = reinterpret_cast<f32x4*>(x + row * Dp);
    for (int c = threadIdx.x; c < Dp / 4; c += 128) dst[c] = src[c];
    if (threadIdx.x == 0 && coords_out) {
        coords_out[2 * row] = s == 0 ? 0.f : coords[2 * (b * Tn + s - 1)];
        coords_out[2 * row + 1] = s == 0 ? 0.f : coords[2 * (b * Tn + s - 1) + 1];
    }
}

// the tile rows of dx (class-token rows dropped): [Bb][S][Dp] -> [Bb*Tn][Dp]
__global__ void __launch_bounds__(128) drop_cls_rows_kernel(const float* __restrict__ dx, float* __restrict__ out, int Dp, int Tn) {
    const long r = blockIdx.x;
    const long b = r / Tn;
    const f32x4* src = reinterpret_cast<const f32x4*>(dx + (r + b + 1) * Dp);
    f32x4* dst = reinterpret_cast<f32x4*>(out + r * Dp);
    for (int c = threadIdx.x; c < Dp / 4; c += 128) dst[c] = src[c];
}

// [R][Cc] fp32 -> [Cc][R]   (tiny: dlogits)
__global__ void transpose_f32_small_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * Cc) {
        const int r = i / Cc, c = i - r * Cc;
        dst[(long)c * R + r] = src[i];
    }
}

// part [B][H][T] -> [B*T][H]   (per-query pieces of the bias_scale gradient, ready for a column sum)
__global__ void __launch_bounds__(256) head_part_rows_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int H, int T) {
    const long n = (long)B * H * T;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long bh = i / T;
        const int t = (int)(i - bh * T);
        const long b = bh / H;
        const int h = (int)(bh - b * H);
        out[(b * T + t) * H + h] = part[i];
    }
}

int gelu_drop_fwd(const void* z, void* u, long n, int zdt, int udt, float p, uint64_t seed, uint32_t sid, void* st) {
    return p > 0.f ? amds_gelu_dropout_fwd(z, u, n, zdt, udt, p, seed, sid, st) : amds_gelu_fwd(z, u, n, zdt, udt, st);
}
int gelu_drop_bwd(const void* z, const void* du, void* dz, long n, int zdt, int dudt, int dzdt, float p, uint64_t seed, uint32_t sid, void* st) {
    return p > 0.f ? amds_gelu_dropout_bwd(z, du, dz, n, zdt, dudt, dzdt, p, seed, sid, st) : amds_gelu_bwd(z, du, dz, n, zdt, dudt, dzdt, st);
}

#define RC(call)                          \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != AMDS_OK) return rc__; \
    } while (0)

constexpr int CFG_TRAIN = -2;      // amds_gemm_ex: by shape, ragged last row tile as its own small launch (M = bags x 1025 is never a multiple of 256)

// the last block on its class rows alone (amds_mil_vit_dropout.cls_tail, else the context's amds_set_mil_cls_tail; with ALiBi the one-query kernel carries the
// distance term: amds_attention_row_alibi_fwd_train).  The pitched rows (a class row every S rows) must fit the 32-bit buffer descriptors of the GEMMs AND of the token-major weight-gradient kernel, which spans
// chunk + 63 rows of pitch S * width 16-bit elements per split (amds_wgrad_tn: chunk = 64 for Bb <= 64 split_k) -- a bag of 8192 tiles with dim_feedforward 2048 passes
// the first bound and not the second: such shapes take the full-block path.
bool cls_tail(const Dims& d, int want) {
    const bool on = want < 0 ? ctx_mil_cls_tail() != 0 : want != 0;
    const long wide = std::max(std::max(d.FFp, 3 * d.Da), d.Dp);
    const long chunk = (d.Bb + 63) / 64 * 64;
    return on && d.L > 0 && d.S <= (d.alibi ? 16384 : 32768) && (long)(d.Bb + 1) * d.S * wide * 4 < (1L << 31) && (chunk + 63) * d.S * wide * 2 < (1L << 31);
}

int gemm_dt(int dt, const void* A, long lda, const void* W, long ldw, long M, int N, int K, int epi, void* out, long ldo, const float* bias, void* st) {
    return amds_gemm_ex(CFG_TRAIN, A, lda, W, ldw, (int)M, N, K, dt, epi, out, ldo, bias, nullptr, nullptr, 0, 0, 0, 1.0f, st);
}

}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_mil_vit_train_saved_bytes(const amds_mil_vit_cfg* cfg_host, int n_bags, int n_tiles) {
    Dims d;
    if (make_dims(cfg_host, n_bags, n_tiles, &d) != AMDS_OK) return 0;
    SavedPlan p;
    plan_saved(d, &p);
    return p.total;
}

extern "C" size_t amds_mil_vit_train_workspace_bytes(const amds_mil_vit_cfg* cfg_host, int n_bags, int n_tiles, int split_k) {
    Dims d;
    if (make_dims(cfg_host, n_bags, n_tiles, &d) != AMDS_OK) return 0;
    if (split_k <= 0 || split_k > 1024) { set_error("amds_mil_vit_train: bad split_k=%d", split_k); return 0; }
    WsPlan p;
    plan_ws(d, split_k, &p);
    return p.total;
}

extern "C" int amds_mil_vit_train_forward(const amds_mil_vit_cfg* cfg_host, const amds_mil_vit_weights* w_host, const void* bags, int bags_dtype,
                                          const float* coords, const amds_mil_vit_dropout* drop_host, float* logits, int n_bags, int n_tiles,
                                          void* saved, size_t saved_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && bags && logits && saved && drop_host, "amds_mil_vit_train_forward: null pointer");
    Dims d;
    RC(make_dims(cfg_host, n_bags, n_tiles, &d));
    const amds_mil_vit_weights& w = *w_host;
    AMDS_REQUIRE(w.class_token && w.proj_w && w.proj_b && w.norm_w && w.norm_b && w.head_w && (d.L == 0 || w.layers_host),
                 "amds_mil_vit_train_forward: incomplete weights");
    AMDS_REQUIRE(!d.alibi || coords, "amds_mil_vit_train_forward: use_alibi=True needs coords");
    AMDS_REQUIRE(bags_dtype == AMDS_F32 || bags_dtype == AMDS_F16 || bags_dtype == AMDS_BF16, "amds_mil_vit_train_forward: bad bags dtype %d", bags_dtype);
    SavedPlan sp;
    plan_saved(d, &sp);
    if (saved_bytes < sp.total) {
        set_error("amds_mil_vit_train_forward: saved-activation arena %zu < required %zu bytes", saved_bytes, sp.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)saved & 255) == 0, "amds_mil_vit_train_forward: arena must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* sv = reinterpret_cast<char*>(saved);
    const int BF = d.dt;                                       // (the name dates from the bf16-only step)
    auto gemm = [&](const void* A, long lda, const void* W, long ldw, long Mr, int N, int K, int epi, void* out, long ldo, const float* bias, void* s2) -> int {
        return gemm_dt(BF, A, lda, W, ldw, Mr, N, K, epi, out, ldo, bias, s2);
    };
    const float p_proj = drop_host->p_proj, p_ff = drop_host->p_ff, p_att = d.alibi ? 0.f : drop_host->p_att;
    const uint64_t seed = drop_host->seed;
    const long M = d.M, Mt = d.Mt;
    const int Dp = d.Dp, Da = d.Da, FFp = d.FFp, Fp = d.Fp, D = d.D, S = d.S, Bb = d.Bb, Ha = d.Ha;

    // ---- project_features: Linear -> GELU -> Dropout (:314-318, :342), bags staged as bf16 operand rows ---------------------------------
    void* a = sv + sp.a;
    if (bags_dtype == AMDS_F16 && BF == AMDS_BF16 && Fp == d.F) RC(amds_convert_f16_bf16(bags, a, Mt * Fp, stream));
    else if (bags_dtype == BF && Fp == d.F) AMDS_HIP(hipMemcpyAsync(a, bags, (size_t)Mt * Fp * 2, hipMemcpyDeviceToDevice, st));       // already the operand type
    else {
        const long total = Mt * Fp;
        const int grid = (int)std::min<long>(8192, (total + 255) / 256);
#define AMDS_STAGE(TI, TO) hipLaunchKernelGGL((stage_bags_bf16_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)bags, (long)d.F, (TO*)a, Fp, total, d.F)
        if (BF == AMDS_BF16) {
            if (bags_dtype == AMDS_F32) AMDS_STAGE(float, bf16);
            else if (bags_dtype == AMDS_F16) AMDS_STAGE(f16, bf16);
            else AMDS_STAGE(bf16, bf16);
        } else {
            if (bags_dtype == AMDS_F32) AMDS_STAGE(float, f16);
            else if (bags_dtype == AMDS_F16) AMDS_STAGE(f16, f16);
            else AMDS_STAGE(bf16, f16);
        }
#undef AMDS_STAGE
        AMDS_LAUNCH_CHECK("stage_bags_bf16_kernel");
    }
    void* zp = sv + sp.zp;
    float* xp = reinterpret_cast<float*>(sv + sp.xp);
    RC(gemm(a, Fp, w.proj_w, Fp, Mt, Dp, Fp, AMDS_EPI_BIAS, zp, Dp, w.proj_b, stream));
    RC(gelu_drop_fwd(zp, xp, Mt * Dp, BF, AMDS_F32, p_proj, seed, 1000, stream));
    float* x = reinterpret_cast<float*>(sv + sp.x0);
    float* cc = d.alibi ? reinterpret_cast<float*>(sv + sp.cc) : nullptr;
    hipLaunchKernelGGL(train_prefix_cls_kernel, dim3((unsigned)M), dim3(128), 0, st, w.class_token, xp, x, Dp, coords, cc, d.Tn);
    AMDS_LAUNCH_CHECK("train_prefix_cls_kernel");

    for (int l = 0; l < d.L; ++l) {
        const amds_mil_vit_layer& Lw = w.layers_host[l];
        const LayerOff& o = sp.layer[l];
        AMDS_REQUIRE(Lw.ln1_w && Lw.ln1_b && Lw.in_w && Lw.in_b && Lw.out_w && Lw.out_b && Lw.ln2_w && Lw.ln2_b && Lw.fc1_w && Lw.fc1_b && Lw.fc2_w &&
                     Lw.fc2_b && (!d.alibi || (Lw.bias_scale && Lw.inv_running_mean)), "amds_mil_vit_train_forward: incomplete weights of layer %d", l);
        float* x_in = reinterpret_cast<float*>(sv + sp.x0 + (size_t)l * sp.x_bytes);
        float* x_out = reinterpret_cast<float*>(sv + sp.x0 + (size_t)(l + 1) * sp.x_bytes);
        float* x_mid = reinterpret_cast<float*>(sv + o.x_mid);
        void *h1 = sv + o.h1, *h2 = sv + o.h2, *qkv = sv + o.qkv, *att = sv + o.att, *z = sv + o.z, *u = sv + o.u;
        float* lse = reinterpret_cast<float*>(sv + o.lse);
        if (Dp != D) {      // LayerNorm writes the first D columns only
            AMDS_HIP(hipMemsetAsync(h1, 0, (size_t)M * Dp * 2, st));
            AMDS_HIP(hipMemsetAsync(h2, 0, (size_t)M * Dp * 2, st));
        }
        // x_mid = x_in + out_proj(attention(in_proj(LayerNorm(x_in))))      (:215-242, :291-292)
        // (the LayerNorm kernel also writes x_mid = x_in, which the out-projection's residual epilogue then updates in place)
        RC(amds_layernorm_train_copy(x_in, Dp, Lw.ln1_w, Lw.ln1_b, h1, Dp, reinterpret_cast<float*>(sv + o.mu1), reinterpret_cast<float*>(sv + o.rs1), (int)M, D,
                                     1e-5f, BF, x_mid, Dp, Dp, stream));
        if (cls_tail(d, drop_host->cls_tail) && l == d.L - 1) {
            // Class-row tail.  The head reads the class row of the last block and nothing else (reference vision_tranformer.py: `self.mlp_head(x[:, 0])`), so this
            // block computes keys | values of every token and -- on the class rows alone, addressed in place by a row pitch of S rows -- the query, its attention
            // (amds_attention_row_fwd_train), the output projection and the MLP.  Dropout draws the bits the full block draws for those rows (row_mul = S).
            // The other rows of att / x_mid's update / h2 / z / u / x_out are never written and never read; they carry no gradient back.
            const long pS = S;
            RC(gemm(h1, Dp, reinterpret_cast<const char*>(Lw.in_w) + (size_t)Da * Dp * 2, Dp, M, 2 * Da, Dp, AMDS_EPI_BIAS, reinterpret_cast<char*>(qkv) + (size_t)Da * 2,
                    3 * Da, Lw.in_b + Da, stream));
            RC(gemm(h1, pS * Dp, Lw.in_w, Dp, Bb, Da, Dp, AMDS_EPI_BIAS, qkv, pS * 3 * Da, Lw.in_b, stream));
            if (d.alibi) RC(amds_attention_row_alibi_fwd_train(qkv, cc, Lw.inv_running_mean, Lw.bias_scale, att, sv + o.u_al, sv + o.osm, lse, Bb, S, Ha, 0, BF, stream));
            else RC(amds_attention_row_fwd_train(qkv, att, lse, Bb, S, Ha, 0, BF, p_att, seed, 10 * l + 1, stream));
            RC(gemm(att, pS * Da, Lw.out_w, Da, Bb, Dp, Da, AMDS_EPI_RESIDUAL, x_mid, pS * Dp, Lw.out_b, stream));
            RC(amds_layernorm_train_copy(x_mid, pS * Dp, Lw.ln2_w, Lw.ln2_b, h2, pS * Dp, reinterpret_cast<float*>(sv + o.mu2), reinterpret_cast<float*>(sv + o.rs2), Bb, D,
                                         1e-5f, BF, p_ff > 0.f ? nullptr : x_out, pS * Dp, Dp, stream));
            RC(gemm(h2, pS * Dp, Lw.fc1_w, Dp, Bb, FFp, Dp, AMDS_EPI_BIAS, z, pS * FFp, Lw.fc1_b, stream));
            RC(gelu_dropout_fwd_rows_dt(z, pS * FFp, u, pS * FFp, Bb, FFp, pS, BF, p_ff, seed, 10 * l + 2, stream));
            if (p_ff > 0.f) {
                float* y = reinterpret_cast<float*>(sv + sp.y);
                RC(gemm(u, pS * FFp, Lw.fc2_w, FFp, Bb, Dp, FFp, AMDS_EPI_BIAS_F32, y, Dp, Lw.fc2_b, stream));
                RC(amds_dropout_add_rows(y, Dp, x_mid, pS * Dp, x_out, pS * Dp, Bb, Dp, pS, p_ff, seed, 10 * l + 3, stream));
            } else {
                RC(gemm(u, pS * FFp, Lw.fc2_w, FFp, Bb, Dp, FFp, AMDS_EPI_RESIDUAL, x_out, pS * Dp, Lw.fc2_b, stream));
            }
            continue;
        }
        RC(gemm(h1, Dp, Lw.in_w, Dp, M, 3 * Da, Dp, AMDS_EPI_BIAS, qkv, 3 * Da, Lw.in_b, stream));
        if (d.alibi)
            RC(attention_alibi_fwd_train_dt(qkv, cc, Lw.inv_running_mean, Lw.bias_scale, att, sv + o.u_al, sv + o.osm, lse, Bb, S, Ha, BF, stream));
        else
            RC(amds_attention_fwd_train(qkv, att, lse, Bb, S, Ha, BF, p_att, seed, 10 * l + 1, stream));
        RC(gemm(att, Da, Lw.out_w, Da, M, Dp, Da, AMDS_EPI_RESIDUAL, x_mid, Dp, Lw.out_b, stream));
        // x_out = x_mid + Dropout(fc2(Dropout(GELU(fc1(LayerNorm(x_mid))))))   (:157-169, :293)
        RC(amds_layernorm_train_copy(x_mid, Dp, Lw.ln2_w, Lw.ln2_b, h2, Dp, reinterpret_cast<float*>(sv + o.mu2), reinterpret_cast<float*>(sv + o.rs2), (int)M, D,
                                     1e-5f, BF, p_ff > 0.f ? nullptr : x_out, Dp, Dp, stream));      // (no dropout: x_out = x_mid here, fc2 adds into it)
        RC(gemm(h2, Dp, Lw.fc1_w, Dp, M, FFp, Dp, AMDS_EPI_BIAS, z, FFp, Lw.fc1_b, stream));
        RC(gelu_drop_fwd(z, u, M * FFp, BF, BF, p_ff, seed, 10 * l + 2, stream));
        if (p_ff > 0.f) {
            float* y = reinterpret_cast<float*>(sv + sp.y);
            RC(gemm(u, FFp, Lw.fc2_w, FFp, M, Dp, FFp, AMDS_EPI_BIAS_F32, y, Dp, Lw.fc2_b, stream));
            RC(amds_dropout_add(y, Dp, x_mid, Dp, x_out, Dp, M, Dp, p_ff, seed, 10 * l + 3, stream));
        } else {
            RC(gemm(u, FFp, Lw.fc2_w, FFp, M, Dp, FFp, AMDS_EPI_RESIDUAL, x_out, Dp, Lw.fc2_b, stream));
        }
    }
    // final LayerNorm on the class-token rows (:294, :382), head in exact fp32 (:384)
    const float* x_last = reinterpret_cast<const float*>(sv + sp.x0 + (size_t)d.L * sp.x_bytes);
    float* clsn = reinterpret_cast<float*>(sv + sp.clsn);
    RC(amds_layernorm_train(x_last, (long)S * Dp, w.norm_w, w.norm_b, clsn, D, reinterpret_cast<float*>(sv + sp.muf), reinterpret_cast<float*>(sv + sp.rsf), Bb, D,
                            1e-5f, AMDS_F32, stream));
    return amds_linear_f32(clsn, w.head_w, w.head_b, logits, Bb, d.C, D, 0, stream);
}

extern "C" int amds_mil_vit_train_backward(const amds_mil_vit_cfg* cfg_host, const amds_mil_vit_weights* w_host, const float* dlogits,
                                           const amds_mil_vit_dropout* drop_host, int n_bags, int n_tiles, const void* saved, size_t saved_bytes,
                                           const amds_mil_vit_grads* grads_host, float* dbags, int split_k, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && dlogits && saved && drop_host && ws, "amds_mil_vit_train_backward: null pointer");
    AMDS_REQUIRE(grads_host || dbags, "amds_mil_vit_train_backward: nothing to compute (no gradient buffers, no dbags)");
    AMDS_REQUIRE(split_k > 0 && split_k <= 1024, "amds_mil_vit_train_backward: bad split_k=%d", split_k);
    Dims d;
    RC(make_dims(cfg_host, n_bags, n_tiles, &d));
    const amds_mil_vit_weights& w = *w_host;
    SavedPlan sp;
    plan_saved(d, &sp);
    WsPlan wp;
    plan_ws(d, split_k, &wp);
    if (saved_bytes < sp.total || ws_bytes < wp.total) {
        set_error("amds_mil_vit_train_backward: arena %zu / workspace %zu < required %zu / %zu bytes", saved_bytes, ws_bytes, sp.total, wp.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE((((uintptr_t)saved | (uintptr_t)ws) & 255) == 0, "amds_mil_vit_train_backward: arena and workspace must be 256-byte aligned");
    const bool need_params = grads_host != nullptr;
    const amds_mil_vit_grads* G = grads_host;
    AMDS_REQUIRE(!need_params || (G->class_token && G->proj_w && G->proj_b && G->norm_w && G->norm_b && G->head_w && G->head_b && (d.L == 0 || G->layers_host)),
                 "amds_mil_vit_train_backward: incomplete gradient buffers");
    AMDS_REQUIRE(!dbags || w.proj_wt, "amds_mil_vit_train_backward: dbags needs the transposed projection weight");
    hipStream_t st = (hipStream_t)stream;
    const char* sv = reinterpret_cast<const char*>(saved);
    char* wk = reinterpret_cast<char*>(ws);
    const int BF = d.dt;
    auto gemm = [&](const void* A, long lda, const void* W, long ldw, long Mr, int N, int K, int epi, void* out, long ldo, const float* bias, void* s2) -> int {
        return gemm_dt(BF, A, lda, W, ldw, Mr, N, K, epi, out, ldo, bias, s2);
    };
    const float p_proj = drop_host->p_proj, p_ff = drop_host->p_ff, p_att = d.alibi ? 0.f : drop_host->p_att;
    const uint64_t seed = drop_host->seed;
    const long M = d.M, Mt = d.Mt, Mp = wp.Mp, Mtp = wp.Mtp;
    const int Dp = d.Dp, Da = d.Da, FFp = d.FFp, Fp = d.Fp, D = d.D, S = d.S, Bb = d.Bb, Ha = d.Ha, C = d.C;
    float* dx = reinterpret_cast<float*>(wk + wp.dx);
    float* dh = reinterpret_cast<float*>(wk + wp.dh);
    void *g16 = wk + wp.g16, *du = wk + wp.du, *dz = wk + wp.dz, *datt = wk + wp.datt, *dqkv = wk + wp.dqkv, *tg = wk + wp.tg, *ta = wk + wp.ta;
    float* part = reinterpret_cast<float*>(wk + wp.part);
    void* cs = wk + wp.cs;
    void* lnb = wk + wp.lnb;

    // Column sums whose results nobody reads before the end of the backward are postponed: first stages (where there is one) on the spot, everything else
    // as entries of ONE amds_colsum_multi launch at the end (21 small launches per step -> 1 for two layers, same bits).  AMDS_COLSUM_DEFER=0: on the spot.
    static const bool defer_sums = !(getenv("AMDS_COLSUM_DEFER") && atoi(getenv("AMDS_COLSUM_DEFER")) == 0);
    amds_colsum_entry sum_e[32];
    int n_sum = 0;
    size_t sums_used = 0;
    auto flush_sums = [&]() -> int {
        sums_used = 0;
        if (n_sum == 0) return AMDS_OK;
        const int n = n_sum;
        n_sum = 0;
        return amds_colsum_multi(sum_e, n, stream);
    };
    auto take_sums = [&](size_t bytes, int entries, float** out) -> int {          // a region that lives until the next flush; nullptr: does not fit at all
        bytes = al(bytes);
        *out = nullptr;
        if (bytes > wp.sums_bytes) return AMDS_OK;
        if (sums_used + bytes > wp.sums_bytes || n_sum + entries > 32) RC(flush_sums());
        *out = reinterpret_cast<float*>(wk + wp.sums + sums_used);
        sums_used += bytes;
        return AMDS_OK;
    };
    // `stable`: x is not written again before the end of the backward (a one-chunk fp32 sum can then read it from the postponed launch)
    auto colsum = [&](const void* x, long ld, float* out, long rows, int cols, int dt, bool stable = false) -> int {
        if (defer_sums) {
            if (rows <= 2048) {
                if (stable && dt == AMDS_F32) {
                    if (n_sum + 1 > 32) RC(flush_sums());
                    sum_e[n_sum++] = amds_colsum_entry{reinterpret_cast<const float*>(x), out, ld, (int)rows, cols, 0};
                    return AMDS_OK;
                }
            } else {
                float* pr;
                RC(take_sums((size_t)((rows + 1023) / 1024) * cols * 4, 1, &pr));
                if (pr) {
                    int nchunk = 0;
                    RC(amds_colsum_partials(x, ld, pr, (int)rows, cols, dt, &nchunk, stream));
                    sum_e[n_sum++] = amds_colsum_entry{pr, out, (long)cols, nchunk, cols, 1};
                    return AMDS_OK;
                }
            }
        }
        return amds_colsum(x, ld, out, (int)rows, cols, dt, 0, cs, wp.cs_bytes, stream);
    };
    // LayerNorm backward; its parameter-gradient partials are summed at the end (or not at all when nobody asked for parameter gradients)
    auto ln_bwd = [&](const float* dy, long dys, const float* x, long xs, const float* mu, const float* rs, const float* gamma, float* dxo, long dxs, int add_skip,
                      float* dgamma, float* dbeta, long rows, void* dx16, long ld16, float p, uint32_t sid) -> int {
        const long nblk = (rows + 63) / 64;
        if (!need_params) {
            float* sc = reinterpret_cast<float*>(lnb);           // (sized for the two partial planes + a reduction workspace)
            return layernorm_bwd_partials_dt(dy, dys, x, xs, mu, rs, gamma, dxo, dxs, add_skip, sc, sc + (size_t)nblk * D, (int)rows, D, dx16, ld16, BF, p, seed, sid, stream);
        }
        if (defer_sums && nblk <= 2048) {
            float* pr;
            RC(take_sums((size_t)2 * al((size_t)nblk * D * 4), 2, &pr));
            if (pr) {
                float* pb = reinterpret_cast<float*>(reinterpret_cast<char*>(pr) + al((size_t)nblk * D * 4));
                RC(layernorm_bwd_partials_dt(dy, dys, x, xs, mu, rs, gamma, dxo, dxs, add_skip, pr, pb, (int)rows, D, dx16, ld16, BF, p, seed, sid, stream));
                sum_e[n_sum++] = amds_colsum_entry{pr, dgamma, (long)D, (int)nblk, D, 0};
                sum_e[n_sum++] = amds_colsum_entry{pb, dbeta, (long)D, (int)nblk, D, 0};
                return AMDS_OK;
            }
        }
        return layernorm_bwd_cast_dt(dy, dys, x, xs, mu, rs, gamma, dxo, dxs, add_skip, dgamma, dbeta, 0, (int)rows, D, lnb, wp.lnb_bytes, dx16, ld16, BF, p, seed, sid, stream);
    };
    // [rows][cols] bf16 -> [cols][pitch]; the columns rows..pitch must read as zeros (the split-K contraction runs over the padded length).  The
    // transposes never write them, so they are zeroed ONCE per pitch for the widest matrix that will use the buffer (`zero_pads`) instead of in
    // front of each of the 8 transposes of a layer (16 memset launches per step -> 4).
    auto zero_pads = [&](void* dst, int max_cols, long rows, long pitch) -> int {
        if (pitch > rows) AMDS_HIP(hipMemset2DAsync((char*)dst + rows * 2, pitch * 2, 0, (size_t)(pitch - rows) * 2, max_cols, st));
        return AMDS_OK;
    };
    auto transpose_pad = [&](const void* src, int cols, void* dst, long rows, long pitch) -> int {
        return amds_transpose16(src, cols, dst, pitch, (int)rows, cols, stream);
    };
    const int wg_cols = std::max(std::max(3 * Da, FFp), Dp), wa_cols = std::max(std::max(std::max(FFp, Dp), Da), Fp);      // as plan_ws sized tg / ta
    // dW straight from the token-major dy / x (amds_wgrad_tn: no transposes, no padded copies); AMDS_WGRAD_TN=0: the transposed form (A/B)
    static const bool use_tn = !(getenv("AMDS_WGRAD_TN") && atoi(getenv("AMDS_WGRAD_TN")) == 0);
    if (need_params && d.L > 0 && !use_tn) {
        RC(zero_pads(tg, wg_cols, M, Mp));
        RC(zero_pads(ta, wa_cols, M, Mp));
    }
    // dW[N][K] = dy^T x: contraction over the padded token dimension in split_k fp32 partials, summed deterministically
    auto wgrad = [&](const void* dyT, const void* xT, int Nn, int Kk, long Mpad, float* out) -> int {
        const long chunk = Mpad / split_k;
        RC(amds_gemm_batched(dyT, Mpad, chunk, xT, Mpad, chunk, Nn, Kk, (int)chunk, split_k, BF, AMDS_EPI_BIAS_F32, part, Kk, (long)Nn * Kk, nullptr, 1.0f, stream));
        return amds_colsum(part, (long)Nn * Kk, out, split_k, Nn * Kk, AMDS_F32, 0, cs, wp.cs_bytes, stream);
    };

    // (TN form: every weight gradient keeps its own region of partials; one launch sums them all at the end -- 9 reduction launches per step -> 1, same bits.
    //  AMDS_WGRAD_DEFER=0: summed one by one, on the spot)
    static const bool defer = !(getenv("AMDS_WGRAD_DEFER") && atoi(getenv("AMDS_WGRAD_DEFER")) == 0);
    const float* def_part[32];
    float* def_out[32];
    long def_count[32];
    int n_def = 0;
    float* part_next = reinterpret_cast<float*>(wk + wp.part_all);
    auto wgrad_tn = [&](const void* dy, long ld_dy, const void* xx, long ld_x, long tokens, int Nn, int Kk, float* out) -> int {
        if (defer && n_def < 32 && ((long)Nn * Kk) % 4 == 0 && ((uintptr_t)out & 15) == 0) {
            RC(amds_wgrad_tn(dy, ld_dy, xx, ld_x, tokens, Nn, Kk, split_k, BF, part_next, stream));
            def_part[n_def] = part_next; def_out[n_def] = out; def_count[n_def] = (long)Nn * Kk; ++n_def;
            part_next += (size_t)Nn * Kk * split_k;
            return AMDS_OK;
        }
        RC(amds_wgrad_tn(dy, ld_dy, xx, ld_x, tokens, Nn, Kk, split_k, BF, part, stream));
        return amds_colsum(part, (long)Nn * Kk, out, split_k, Nn * Kk, AMDS_F32, 0, cs, wp.cs_bytes, stream);
    };

    // ---- head and final LayerNorm -------------------------------------------------------------------------------------------------------
    const float* clsn = reinterpret_cast<const float*>(sv + sp.clsn);
    const float* x_last = reinterpret_cast<const float*>(sv + sp.x0 + (size_t)d.L * sp.x_bytes);
    float* dcls = reinterpret_cast<float*>(wk + wp.dcls);
    if (need_params) {
        float* dlt = reinterpret_cast<float*>(wk + wp.dlt);
        hipLaunchKernelGGL(transpose_f32_small_kernel, dim3((Bb * C + 255) / 256), dim3(256), 0, st, dlogits, dlt, Bb, C);
        AMDS_LAUNCH_CHECK("transpose_f32_small_kernel");
        RC(amds_bgemm_f32(dlt, Bb, 0, 0, clsn, D, 0, 0, 0, G->head_w, D, 0, 0, 1, 1, C, D, Bb, 1.0f, 0.0f, nullptr, 0, stream));      // dW_head = dlogits^T clsn
        RC(colsum(dlogits, C, G->head_b, Bb, C, AMDS_F32, true));                                                      // (the caller's tensor)
    }
    RC(amds_bgemm_f32(dlogits, C, 0, 0, w.head_w, D, 0, 0, 0, dcls, D, 0, 0, 1, 1, Bb, D, C, 1.0f, 0.0f, nullptr, 0, stream));           // dclsn = dlogits W_head
    AMDS_HIP(hipMemsetAsync(dx, 0, (size_t)M * Dp * 4, st));
    float* scratch_g = reinterpret_cast<float*>(wk + wp.gsc);       // LayerNorm's backward always produces dgamma / dbeta
    RC(ln_bwd(dcls, D, x_last, (long)S * Dp, reinterpret_cast<const float*>(sv + sp.muf), reinterpret_cast<const float*>(sv + sp.rsf), w.norm_w,
              dx, (long)S * Dp, 0, need_params ? G->norm_w : scratch_g, need_params ? G->norm_b : scratch_g + D, Bb, nullptr, 0, 0.f, 0));

    const bool fused_cast = (D == Dp) && !(getenv("AMDS_LNBWD_CAST") && atoi(getenv("AMDS_LNBWD_CAST")) == 0);      // (pad columns of g16 would stay unwritten otherwise)
    for (int l = d.L - 1; l >= 0; --l) {
        const amds_mil_vit_layer& Lw = w.layers_host[l];
        const LayerOff& o = sp.layer[l];
        AMDS_REQUIRE(Lw.in_wt && Lw.out_wt && Lw.fc1_wt && Lw.fc2_wt, "amds_mil_vit_train_backward: layer %d has no transposed weights (training pack)", l);
        const amds_mil_vit_layer_grads* Gl = need_params ? &G->layers_host[l] : nullptr;
        AMDS_REQUIRE(!need_params || (Gl->ln1_w && Gl->ln1_b && Gl->in_w && Gl->in_b && Gl->out_w && Gl->out_b && Gl->ln2_w && Gl->ln2_b && Gl->fc1_w &&
                                      Gl->fc1_b && Gl->fc2_w && Gl->fc2_b && (!d.alibi || Gl->bias_scale)),
                     "amds_mil_vit_train_backward: incomplete gradient buffers of layer %d", l);
        const float* x_in = reinterpret_cast<const float*>(sv + sp.x0 + (size_t)l * sp.x_bytes);
        const float* x_mid = reinterpret_cast<const float*>(sv + o.x_mid);
        const void *h1 = sv + o.h1, *h2 = sv + o.h2, *qkv = sv + o.qkv, *att = sv + o.att, *z = sv + o.z, *u = sv + o.u;
        const float* lse = reinterpret_cast<const float*>(sv + o.lse);
        const bool tail = cls_tail(d, drop_host->cls_tail) && l == d.L - 1;
        if (tail) {
            // Class-row tail (see the forward): dx of the last block lives on the class rows; its MLP, second LayerNorm and output projection are differentiated on
            // those Bb rows alone (row pitch S rows, dropout bits of the full tensors' rows b * S).
            const long pS = S;
            RC(dropout_cast_bwd_rows_dt(dx, pS * Dp, g16, pS * Dp, Bb, Dp, pS, BF, p_ff, seed, (uint32_t)(10 * l + 3), stream));
            RC(gemm(g16, pS * Dp, Lw.fc2_wt, Dp, Bb, FFp, Dp, AMDS_EPI_BIAS, du, pS * FFp, nullptr, stream));
            if (need_params) {
                RC(wgrad_tn(g16, pS * Dp, u, pS * FFp, Bb, Dp, FFp, Gl->fc2_w));
                RC(p_ff > 0.f ? colsum(g16, pS * Dp, Gl->fc2_b, Bb, Dp, BF) : colsum(dx, pS * Dp, Gl->fc2_b, Bb, Dp, AMDS_F32));
            }
            RC(gelu_dropout_bwd_rows_dt(z, pS * FFp, du, pS * FFp, dz, pS * FFp, Bb, FFp, pS, BF, p_ff, seed, (uint32_t)(10 * l + 2), stream));
            RC(gemm(dz, pS * FFp, Lw.fc1_wt, FFp, Bb, Dp, FFp, AMDS_EPI_BIAS_F32, dh, pS * Dp, nullptr, stream));
            if (need_params) {
                RC(wgrad_tn(dz, pS * FFp, h2, pS * Dp, Bb, FFp, Dp, Gl->fc1_w));
                RC(colsum(dz, pS * FFp, Gl->fc1_b, Bb, FFp, BF));
            }
            RC(ln_bwd(dh, pS * Dp, x_mid, pS * Dp, reinterpret_cast<const float*>(sv + o.mu2), reinterpret_cast<const float*>(sv + o.rs2), Lw.ln2_w, dx, pS * Dp, 1,
                      need_params ? Gl->ln2_w : scratch_g, need_params ? Gl->ln2_b : scratch_g + D, Bb, fused_cast ? g16 : nullptr, pS * Dp, 0.f, 0));
            if (!fused_cast) RC(dropout_cast_bwd_rows_dt(dx, pS * Dp, g16, pS * Dp, Bb, Dp, pS, BF, 0.f, 0, 0, stream));            // d(x_mid) as bf16
            RC(gemm(g16, pS * Dp, Lw.out_wt, Dp, Bb, Da, Dp, AMDS_EPI_BIAS, datt, pS * Da, nullptr, stream));
            if (need_params) {
                RC(wgrad_tn(g16, pS * Dp, att, pS * Da, Bb, Dp, Da, Gl->out_w));
                RC(colsum(dx, pS * Dp, Gl->out_b, Bb, Dp, AMDS_F32));
            }
        } else {
        // ---- feed-forward branch ----------------------------------------------------------------------------------------------------------
        // g16 = bf16(Dropout'(dx)): written by the LayerNorm backward that produced dx (layers below the top one, D == Dp), else by its own pass
        if (!(fused_cast && l < d.L - 1)) {
            if (p_ff > 0.f) RC(amds_dropout_cast_bwd(dx, Dp, g16, Dp, M, Dp, BF, p_ff, seed, 10 * l + 3, stream));
            else RC(amds_cast_pad(dx, Dp, g16, Dp, (int)M, Dp, BF, stream));
        }
        RC(gemm(g16, Dp, Lw.fc2_wt, Dp, M, FFp, Dp, AMDS_EPI_BIAS, du, FFp, nullptr, stream));                              // du = dy W2
        if (need_params) {
            if (use_tn) RC(wgrad_tn(g16, Dp, u, FFp, M, Dp, FFp, Gl->fc2_w));
            else {
                RC(transpose_pad(g16, Dp, tg, M, Mp));
                RC(transpose_pad(u, FFp, ta, M, Mp));
                RC(wgrad(tg, ta, Dp, FFp, Mp, Gl->fc2_w));
            }
            RC(p_ff > 0.f ? colsum(g16, Dp, Gl->fc2_b, M, Dp, BF) : colsum(dx, Dp, Gl->fc2_b, M, Dp, AMDS_F32));
        }
        RC(gelu_drop_bwd(z, du, dz, M * FFp, BF, BF, BF, p_ff, seed, 10 * l + 2, stream));
        RC(gemm(dz, FFp, Lw.fc1_wt, FFp, M, Dp, FFp, AMDS_EPI_BIAS_F32, dh, Dp, nullptr, stream));                          // dh2 fp32
        if (need_params) {
            if (use_tn) RC(wgrad_tn(dz, FFp, h2, Dp, M, FFp, Dp, Gl->fc1_w));
            else {
                RC(transpose_pad(dz, FFp, tg, M, Mp));
                RC(transpose_pad(h2, Dp, ta, M, Mp));
                RC(wgrad(tg, ta, FFp, Dp, Mp, Gl->fc1_w));
            }
            RC(colsum(dz, FFp, Gl->fc1_b, M, FFp, BF));
        }
        RC(ln_bwd(dh, Dp, x_mid, Dp, reinterpret_cast<const float*>(sv + o.mu2), reinterpret_cast<const float*>(sv + o.rs2), Lw.ln2_w, dx, Dp, 1,
                  need_params ? Gl->ln2_w : scratch_g, need_params ? Gl->ln2_b : scratch_g + D, M, fused_cast ? g16 : nullptr, Dp, 0.f, 0));
        // ---- attention branch -------------------------------------------------------------------------------------------------------------
        if (!fused_cast) RC(amds_cast_pad(dx, Dp, g16, Dp, (int)M, Dp, BF, stream));                                         // d(x_mid) as bf16
        RC(gemm(g16, Dp, Lw.out_wt, Dp, M, Da, Dp, AMDS_EPI_BIAS, datt, Da, nullptr, stream));
        if (need_params) {
            if (use_tn) RC(wgrad_tn(g16, Dp, att, Da, M, Dp, Da, Gl->out_w));
            else {
                RC(transpose_pad(g16, Dp, tg, M, Mp));
                RC(transpose_pad(att, Da, ta, M, Mp));
                RC(wgrad(tg, ta, Dp, Da, Mp, Gl->out_w));
            }
            RC(colsum(dx, Dp, Gl->out_b, M, Dp, AMDS_F32));
        }
        }
        float* dqs = reinterpret_cast<float*>(wk + wp.dqs);
        if (d.alibi && tail) {
            // one query per (bag, head): dbs [Bb][Ha] = -dO . U, summed over the bags = the gradient of bias_scale_h
            float* dbs = reinterpret_cast<float*>(wk + wp.dbsp);
            AMDS_REQUIRE(Lw.inv_running_mean && Lw.bias_scale, "amds_mil_vit_train_backward: layer %d has no ALiBi scales", l);
            RC(amds_attention_row_alibi_bwd_train(qkv, sv + o.osm, sv + o.u_al, datt, lse, reinterpret_cast<const float*>(sv + sp.cc), Lw.bias_scale, Lw.inv_running_mean,
                                                  dqkv, dbs, Bb, S, Ha, 0, BF, stream));
            if (need_params) RC(colsum(dbs, Ha, Gl->bias_scale, Bb, Ha, AMDS_F32));
        } else if (d.alibi) {
            float* dbsp = reinterpret_cast<float*>(wk + wp.dbsp);
            AMDS_REQUIRE(Lw.head_scale && Lw.bias_scale, "amds_mil_vit_train_backward: layer %d has no ALiBi scales", l);     // head_scale = dist_scale
            RC(attention_alibi_bwd_dt(qkv, sv + o.osm, sv + o.u_al, datt, lse, reinterpret_cast<const float*>(sv + sp.cc), Lw.bias_scale, Lw.head_scale, dqs, dbsp,
                                      dqkv, Bb, S, Ha, BF, stream));
            if (need_params) {
                float* dbst = reinterpret_cast<float*>(wk + wp.dbst);
                const long n = (long)Bb * Ha * S;
                hipLaunchKernelGGL(head_part_rows_kernel, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, st, dbsp, dbst, Bb, Ha, S);
                AMDS_LAUNCH_CHECK("head_part_rows_kernel");
                RC(colsum(dbst, Ha, Gl->bias_scale, M, Ha, AMDS_F32));
            }
        } else if (tail) {
            // (only the class query has an output and a gradient: dK / dV are rank-1 in it, dQ is zero elsewhere -- one pass instead of the two blocked kernels)
            RC(amds_attention_row_bwd_train(qkv, att, datt, lse, dqkv, Bb, S, Ha, 0, BF, p_att, seed, 10 * l + 1, stream));
        } else {
            RC(amds_attention_bwd_train(qkv, att, datt, lse, dqs, dqkv, Bb, S, Ha, BF, p_att, seed, 10 * l + 1, stream));
        }
        // Class-row tail: dQ is zero outside the Bb class rows, so the Q third of dqkv contributes to the in-projection's weight gradient and to dh1 through those rows
        // alone -- the two launches over all M rows take the K | V columns (2/3 of their N resp. K), the Q third follows on the class rows (row pitch S rows)
        const bool q_rows = tail && use_tn && Da % 256 == 0;
        const size_t esz = 2;                                               // dqkv / h1 / in_wt: 16-bit
        if (need_params) {
            if (q_rows) {
                RC(wgrad_tn(reinterpret_cast<const char*>(dqkv) + (size_t)Da * esz, 3 * Da, h1, Dp, M, 2 * Da, Dp, Gl->in_w + (size_t)Da * Dp));      // rows Da .. 3 Da of dW_in
                RC(wgrad_tn(dqkv, (long)S * 3 * Da, h1, (long)S * Dp, Bb, Da, Dp, Gl->in_w));                                                         // rows 0 .. Da from the class rows
            } else if (use_tn) RC(wgrad_tn(dqkv, 3 * Da, h1, Dp, M, 3 * Da, Dp, Gl->in_w));
            else {
                RC(transpose_pad(dqkv, 3 * Da, tg, M, Mp));
                RC(transpose_pad(h1, Dp, ta, M, Mp));
                RC(wgrad(tg, ta, 3 * Da, Dp, Mp, Gl->in_w));
            }
            RC(colsum(dqkv, 3 * Da, Gl->in_b, M, 3 * Da, BF));
        }
        if (q_rows) {
            RC(gemm(reinterpret_cast<const char*>(dqkv) + (size_t)Da * esz, 3 * Da, reinterpret_cast<const char*>(Lw.in_wt) + (size_t)Da * esz, 3 * Da, M, Dp, 2 * Da, AMDS_EPI_BIAS_F32,
                    dh, Dp, nullptr, stream));                                                                              // dh1 = dK|dV W_in[K|V rows], every row
            RC(gemm(dqkv, (long)S * 3 * Da, Lw.in_wt, 3 * Da, Bb, Dp, 3 * Da, AMDS_EPI_BIAS_F32, dh, (long)S * Dp, nullptr, stream));     // the class rows again, with their dQ
        } else
        RC(gemm(dqkv, 3 * Da, Lw.in_wt, 3 * Da, M, Dp, 3 * Da, AMDS_EPI_BIAS_F32, dh, Dp, nullptr, stream));                 // dh1 fp32
        // (the layer below starts with g16 = bf16(Dropout'(dx)) at ITS feed-forward dropout site: written here, where dx is made)
        RC(ln_bwd(dh, Dp, x_in, Dp, reinterpret_cast<const float*>(sv + o.mu1), reinterpret_cast<const float*>(sv + o.rs1), Lw.ln1_w, dx, Dp, 1,
                  need_params ? Gl->ln1_w : scratch_g, need_params ? Gl->ln1_b : scratch_g + D, M, (fused_cast && l > 0) ? g16 : nullptr, Dp, p_ff,
                  (uint32_t)(10 * (l - 1) + 3)));
    }
    // ---- class token, project_features ------------------------------------------------------------------------------------------------------
    if (need_params) RC(colsum(dx, (long)S * Dp, G->class_token, Bb, Dp, AMDS_F32, true));                                     // class-token rows (dx is final here)
    float* dxp = reinterpret_cast<float*>(wk + wp.dxp);
    void* dzp = wk + wp.dzp;
    hipLaunchKernelGGL(drop_cls_rows_kernel, dim3((unsigned)Mt), dim3(128), 0, st, dx, dxp, Dp, d.Tn);
    AMDS_LAUNCH_CHECK("drop_cls_rows_kernel");
    RC(gelu_drop_bwd(sv + sp.zp, dxp, dzp, Mt * Dp, BF, AMDS_F32, BF, p_proj, seed, 1000, stream));
    if (need_params) {
        if (use_tn) RC(wgrad_tn(dzp, Dp, sv + sp.a, Fp, Mt, Dp, Fp, G->proj_w));
        else {
            RC(zero_pads(tg, Dp, Mt, Mtp));                   // the tile rows have their own pitch
            RC(zero_pads(ta, Fp, Mt, Mtp));
            RC(transpose_pad(dzp, Dp, tg, Mt, Mtp));
            RC(transpose_pad(sv + sp.a, Fp, ta, Mt, Mtp));
            RC(wgrad(tg, ta, Dp, Fp, Mtp, G->proj_w));
        }
        RC(colsum(dzp, Dp, G->proj_b, Mt, Dp, BF));
    }
    if (dbags) RC(gemm(dzp, Dp, w.proj_wt, Dp, Mt, Fp, Dp, AMDS_EPI_BIAS_F32, dbags, Fp, nullptr, stream));                 // [Mt][Fp] fp32 (padded columns = 0)
    if (n_def > 0) RC(amds_sum_partials_multi(def_part, def_out, def_count, n_def, split_k, stream));
    RC(flush_sums());
    return AMDS_OK;
}
