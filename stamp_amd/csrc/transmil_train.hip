// transmil_train.hip -- the TRAINING forward and backward of the TransMIL head, one call each: the thin loop around amds_nystrom_attn_fwd / _bwd
// (nystrom_train.hip) that used to live in stamp_amd/transmil_core.py.
//
// Forward = the train-mode forward of the reference's TransMIL (src/stamp/modeling/models/trans_mil.py:299-325; one dropout site: `to_out`'s
// Dropout(0.1), :66); backward = what autograd derives from it (loss.backward() through Lightning, models/__init__.py:239-279): _fc2, final LayerNorm on
// the class rows, layer2, PPEG (weight gradients as tap correlations, data gradient = the same convolutions with flipped kernels), layer1, class token,
// the wrap-padded tiles' gradients added onto the first tiles', ReLU, _fc1.  fp32 throughout; launch sequences over caller-owned arenas.
#include <algorithm>
#include "common.h"

namespace amds {
namespace {
inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

struct TtDims { int F, Cd, C, Bb, T, side, n, add; };

int tt_dims(const amds_transmil_cfg* c, int Bb, int T, TtDims* d) {
    AMDS_REQUIRE(c, "amds_transmil_train: null config");
    AMDS_REQUIRE(c->n_feats > 0 && c->dim > 0 && c->dim % 8 == 0 && c->classes > 0, "amds_transmil_train: bad config (dim_hidden must be a multiple of 8)");
    AMDS_REQUIRE(Bb > 0 && T >= 1, "amds_transmil_train: bad shape bags=%d tiles=%d", Bb, T);
    d->F = c->n_feats; d->Cd = c->dim; d->C = c->classes; d->Bb = Bb; d->T = T;
    int side = (int)ceil(sqrt((double)T));
    while ((long)side * side < T) ++side;
    while (side > 1 && (long)(side - 1) * (side - 1) >= T) --side;
    d->side = side;
    d->n = side * side + 1;
    d->add = side * side - T;
    return AMDS_OK;
}

struct TtSaved { size_t a, h, x1, mu1, rs1, ny1, xp, x2, mu2, rs2, ny2, xf, clsn, muf, rsf, y, total, ny_bytes; };

int tt_saved(const TtDims& d, TtSaved* s) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t Mt = (size_t)d.Bb * d.T, M = (size_t)d.Bb * d.n, Cd = d.Cd;
    s->ny_bytes = amds_nystrom_attn_saved_bytes(d.Cd, d.Bb, d.n);
    if (s->ny_bytes == 0) return AMDS_ERR_INVALID;
    s->a = take(Mt * d.F * 4); s->h = take(Mt * Cd * 4);
    s->x1 = take(M * Cd * 4); s->mu1 = take(M * 4); s->rs1 = take(M * 4); s->ny1 = take(s->ny_bytes);
    s->xp = take(M * Cd * 4); s->x2 = take(M * Cd * 4); s->mu2 = take(M * 4); s->rs2 = take(M * 4); s->ny2 = take(s->ny_bytes);
    s->xf = take(M * Cd * 4); s->clsn = take((size_t)d.Bb * Cd * 4); s->muf = take((size_t)d.Bb * 4); s->rsf = take((size_t)d.Bb * 4);
    s->y = take(M * Cd * 4);
    s->total = off;
    return AMDS_OK;
}

struct TtWs { size_t dx, dx2, dy, dcls, dlt, dh, dzh, part, cs, lnb, pw, zb, wflip, ny, gsc, total, cs_bytes, lnb_bytes, pw_bytes, ny_bytes; };

int tt_ws(const TtDims& d, TtWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t Mt = (size_t)d.Bb * d.T, M = (size_t)d.Bb * d.n, Cd = d.Cd;
    w->ny_bytes = amds_nystrom_attn_workspace_bytes(d.Cd, d.Bb, d.n);
    if (w->ny_bytes == 0) return AMDS_ERR_INVALID;
    w->dx = take(M * Cd * 4); w->dx2 = take(M * Cd * 4); w->dy = take(M * Cd * 4);
    w->dcls = take((size_t)d.Bb * Cd * 4); w->dlt = take((size_t)d.Bb * d.C * 4);
    w->dh = take(Mt * Cd * 4); w->dzh = take(Mt * Cd * 4);
    w->part = take((size_t)d.Bb * Cd * d.F * 4);
    size_t cs = amds_colsum_workspace_bytes(d.Bb, d.Cd * d.F);
    cs = std::max(cs, amds_colsum_workspace_bytes((int)Mt, d.Cd));
    cs = std::max(cs, amds_colsum_workspace_bytes(d.Bb, std::max(d.Cd, d.C)));
    w->cs_bytes = std::max<size_t>(cs, 4); w->cs = take(w->cs_bytes);
    w->lnb_bytes = std::max<size_t>(amds_layernorm_bwd_workspace_bytes((int)M, d.Cd), 4); w->lnb = take(w->lnb_bytes);
    w->pw_bytes = std::max<size_t>(amds_ppeg_wgrad_workspace_bytes(d.Bb, d.Cd), 4); w->pw = take(w->pw_bytes);
    w->zb = take(Cd * 4); w->wflip = take(Cd * (49 + 25 + 9) * 4);
    w->ny = take(w->ny_bytes);
    w->gsc = take((size_t)(2 * Cd + 50 * Cd) * 4);
    w->total = off;
    return AMDS_OK;
}

template <typename TI>
__global__ void __launch_bounds__(256) tt_to_f32_kernel(const TI* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}
// x [Bb][n][Cd]: class token, the T projected tiles, then the FIRST tiles again up to side^2 (:306-314)
__global__ void __launch_bounds__(128) tt_wrap_cls_kernel(const float* __restrict__ cls, const float* __restrict__ h, float* __restrict__ x, int Cd, int T, int n, int relu) {
    const long row = blockIdx.x;
    const long b = row / n;
    const int s = (int)(row - b * n);
    const float* src = s == 0 ? cls : h + (b * T + (s - 1 < T ? s - 1 : s - 1 - T)) * Cd;
    float* dst = x + row * Cd;
    if (relu && s != 0) { for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = fmaxf(src[c], 0.f); return; }      // (h holds the pre-activation of _fc1)
    for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = src[c];
}
// dh [Bb][T][Cd] = dx[:, 1 : 1 + T]
__global__ void __launch_bounds__(128) tt_tile_rows_kernel(const float* __restrict__ dx, float* __restrict__ dh, int Cd, int T, int n) {
    const long r = blockIdx.x;
    const long b = r / T;
    const int t = (int)(r - b * T);
    const float* src = dx + (b * n + 1 + t) * Cd;
    float* dst = dh + r * Cd;
    for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = src[c];
}
__global__ void tt_flip_taps_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int taps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)rows * taps) {
        const long r = i / taps;
        const int k = (int)(i - r * taps);
        out[r * taps + (taps - 1 - k)] = w[i];
    }
}
__global__ void tt_transpose_small_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * Cc) {
        const int r = i / Cc, c = i - r * Cc;
        dst[(long)c * R + r] = src[i];
    }
}

#define RC(call)                          \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != AMDS_OK) return rc__; \
    } while (0)
}  // namespace
}  // namespace amds

using namespace amds;

static int tt_cls_tail(const amds_transmil_cfg* c) { return c->train_cls_tail < 0 ? (ctx_mil_cls_tail() != 0) : (c->train_cls_tail != 0); }

extern "C" size_t amds_transmil_train_saved_bytes(const amds_transmil_cfg* cfg_host, int n_bags, int n_tiles) {
    TtDims d; TtSaved s;
    if (tt_dims(cfg_host, n_bags, n_tiles, &d) != AMDS_OK || tt_saved(d, &s) != AMDS_OK) return 0;
    return s.total;
}
extern "C" size_t amds_transmil_train_workspace_bytes(const amds_transmil_cfg* cfg_host, int n_bags, int n_tiles) {
    TtDims d; TtWs w;
    if (tt_dims(cfg_host, n_bags, n_tiles, &d) != AMDS_OK || tt_ws(d, &w) != AMDS_OK) return 0;
    return w.total;
}

extern "C" int amds_transmil_train_forward(const amds_transmil_cfg* cfg_host, const amds_transmil_weights* w_host, const void* bags, int bags_dtype, float p_drop,
                                           uint64_t seed, float* logits, int n_bags, int n_tiles, void* saved, size_t saved_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && bags && logits && saved, "amds_transmil_train_forward: null pointer");
    TtDims d; TtSaved s;
    RC(tt_dims(cfg_host, n_bags, n_tiles, &d));
    RC(tt_saved(d, &s));
    const amds_transmil_weights& w = *w_host;
    AMDS_REQUIRE(w.fc1_w && w.fc1_b && w.cls_token && w.norm_w && w.norm_b && w.fc2_w && w.fc2_b && w.ppeg_w7 && w.ppeg_b7 && w.ppeg_w5 && w.ppeg_b5 && w.ppeg_w3 &&
                 w.ppeg_b3, "amds_transmil_train_forward: incomplete weights");
    AMDS_REQUIRE(bags_dtype == AMDS_F32 || bags_dtype == AMDS_F16 || bags_dtype == AMDS_BF16, "amds_transmil_train_forward: bad bags dtype %d", bags_dtype);
    if (saved_bytes < s.total) { set_error("amds_transmil_train_forward: saved-activation arena %zu < required %zu bytes", saved_bytes, s.total); return AMDS_ERR_WORKSPACE; }
    AMDS_REQUIRE(((uintptr_t)saved & 255) == 0, "amds_transmil_train_forward: arena must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* sv = reinterpret_cast<char*>(saved);
    const int Bb = d.Bb, T = d.T, Cd = d.Cd, n = d.n;
    const long Mt = (long)Bb * T, M = (long)Bb * n;
    float* a = reinterpret_cast<float*>(sv + s.a);
    if (bags_dtype == AMDS_F32) AMDS_HIP(hipMemcpyAsync(a, bags, (size_t)Mt * d.F * 4, hipMemcpyDeviceToDevice, st));
    else {
        const long cnt = Mt * d.F;
        const int grid = (int)std::min<long>(8192, (cnt + 255) / 256);
        if (bags_dtype == AMDS_F16) hipLaunchKernelGGL((tt_to_f32_kernel<f16>), dim3(grid), dim3(256), 0, st, (const f16*)bags, a, cnt);
        else hipLaunchKernelGGL((tt_to_f32_kernel<bf16>), dim3(grid), dim3(256), 0, st, (const bf16*)bags, a, cnt);
        AMDS_LAUNCH_CHECK("tt_to_f32_kernel");
    }
    float* h = reinterpret_cast<float*>(sv + s.h);
    // _fc1: Linear + ReLU (:303).  Below "highest": the product through amds_bgemm_f32 (bf16 x 3 at "high"), h keeps the PRE-activation (amds_relu_bwd tests h > 0:
    // the same mask) and the ReLU rides on the copy into the wrapped sequence (transmil_fwd.hip)
    const bool fc1_x3 = ctx_matmul_precision() != AMDS_MATMUL_HIGHEST && Mt % 128 == 0 && Cd % 128 == 0 && d.F % 32 == 0;
    if (fc1_x3) RC(amds_bgemm_f32(a, d.F, 0, 0, w.fc1_w, d.F, 0, 0, 1, h, Cd, 0, 0, 1, 1, (int)Mt, Cd, d.F, 1.0f, 0.0f, w.fc1_b, 0, stream));
    else RC(amds_linear_f32(a, w.fc1_w, w.fc1_b, h, (int)Mt, Cd, d.F, 1, stream));
    float *x1 = reinterpret_cast<float*>(sv + s.x1), *xp = reinterpret_cast<float*>(sv + s.xp), *x2 = reinterpret_cast<float*>(sv + s.x2);
    float *xf = reinterpret_cast<float*>(sv + s.xf), *y = reinterpret_cast<float*>(sv + s.y);
    hipLaunchKernelGGL(tt_wrap_cls_kernel, dim3((unsigned)M), dim3(128), 0, st, w.cls_token, h, x1, Cd, T, n, fc1_x3 ? 1 : 0);
    AMDS_LAUNCH_CHECK("tt_wrap_cls_kernel");
    // layer1 (:317): xp = x1 + Dropout(to_out(Nystrom(LayerNorm(x1))))
    RC(amds_layernorm_train(x1, Cd, w.layer[0].norm_w, w.layer[0].norm_b, y, Cd, reinterpret_cast<float*>(sv + s.mu1), reinterpret_cast<float*>(sv + s.rs1), (int)M, Cd, 1e-5f,
                            AMDS_F32, stream));
    AMDS_HIP(hipMemcpyAsync(xp, x1, (size_t)M * Cd * 4, hipMemcpyDeviceToDevice, st));
    RC(amds_nystrom_attn_fwd(&w.layer[0], Cd, y, xp, Bb, n, p_drop, seed, 1, sv + s.ny1, s.ny_bytes, stream));
    // PPEG (:318)
    RC(amds_ppeg(xp, x2, w.ppeg_w7, w.ppeg_b7, w.ppeg_w5, w.ppeg_b5, w.ppeg_w3, w.ppeg_b3, Bb, d.side, d.side, Cd, stream));
    // layer2 (:319)
    RC(amds_layernorm_train(x2, Cd, w.layer[1].norm_w, w.layer[1].norm_b, y, Cd, reinterpret_cast<float*>(sv + s.mu2), reinterpret_cast<float*>(sv + s.rs2), (int)M, Cd, 1e-5f,
                            AMDS_F32, stream));
    AMDS_HIP(hipMemcpyAsync(xf, x2, (size_t)M * Cd * 4, hipMemcpyDeviceToDevice, st));
    // layer2's output is read at the class rows alone (:322): the attention output, to_out and Dropout of the other rows are skipped on request
    RC(nystrom_attn_fwd_ex(&w.layer[1], Cd, y, xf, Bb, n, p_drop, seed, 2, sv + s.ny2, s.ny_bytes, tt_cls_tail(cfg_host), stream));
    // final LayerNorm on the class rows, _fc2 (:322-325)
    float* clsn = reinterpret_cast<float*>(sv + s.clsn);
    RC(amds_layernorm_train(xf, (long)n * Cd, w.norm_w, w.norm_b, clsn, Cd, reinterpret_cast<float*>(sv + s.muf), reinterpret_cast<float*>(sv + s.rsf), Bb, Cd, 1e-5f, AMDS_F32,
                            stream));
    return amds_linear_f32(clsn, w.fc2_w, w.fc2_b, logits, Bb, d.C, Cd, 0, stream);
}

extern "C" int amds_transmil_train_backward(const amds_transmil_cfg* cfg_host, const amds_transmil_weights* w_host, const float* dlogits, float p_drop, uint64_t seed,
                                            int n_bags, int n_tiles, const void* saved, size_t saved_bytes, const amds_transmil_grads* grads_host, float* dbags, void* ws,
                                            size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && dlogits && saved && ws, "amds_transmil_train_backward: null pointer");
    AMDS_REQUIRE(grads_host || dbags, "amds_transmil_train_backward: nothing to compute (no gradient buffers, no dbags)");
    TtDims d; TtSaved s; TtWs k;
    RC(tt_dims(cfg_host, n_bags, n_tiles, &d));
    RC(tt_saved(d, &s));
    RC(tt_ws(d, &k));
    if (saved_bytes < s.total || ws_bytes < k.total) {
        set_error("amds_transmil_train_backward: arena %zu / workspace %zu < required %zu / %zu bytes", saved_bytes, ws_bytes, s.total, k.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE((((uintptr_t)saved | (uintptr_t)ws) & 255) == 0, "amds_transmil_train_backward: arena and workspace must be 256-byte aligned");
    const amds_transmil_weights& w = *w_host;
    const amds_transmil_grads* G = grads_host;
    AMDS_REQUIRE(!G || (G->fc1_w && G->fc1_b && G->cls_token && G->ppeg_corr && G->norm_w && G->norm_b && G->fc2_w && G->fc2_b && G->layer[0].norm_w && G->layer[0].norm_b &&
                        G->layer[1].norm_w && G->layer[1].norm_b), "amds_transmil_train_backward: incomplete gradient buffers");
    hipStream_t st = (hipStream_t)stream;
    const char* sv = reinterpret_cast<const char*>(saved);
    char* wk = reinterpret_cast<char*>(ws);
    const int Bb = d.Bb, T = d.T, Cd = d.Cd, n = d.n, C = d.C, F = d.F;
    const long Mt = (long)Bb * T, M = (long)Bb * n;
    auto colsum = [&](const float* x, long ld, float* out, long rows, int cols) { return amds_colsum(x, ld, out, (int)rows, cols, AMDS_F32, 0, wk + k.cs, k.cs_bytes, stream); };
    float* scratch = reinterpret_cast<float*>(wk + k.gsc);                       // parameter gradients nobody asked for
    const float* clsn = reinterpret_cast<const float*>(sv + s.clsn);
    const float* xf = reinterpret_cast<const float*>(sv + s.xf);
    float *dx = reinterpret_cast<float*>(wk + k.dx), *dx2 = reinterpret_cast<float*>(wk + k.dx2), *dy = reinterpret_cast<float*>(wk + k.dy);
    float* dcls = reinterpret_cast<float*>(wk + k.dcls);
    // ---- _fc2, final LayerNorm
    if (G) {
        float* dlt = reinterpret_cast<float*>(wk + k.dlt);
        hipLaunchKernelGGL(tt_transpose_small_kernel, dim3((Bb * C + 255) / 256), dim3(256), 0, st, dlogits, dlt, Bb, C);
        AMDS_LAUNCH_CHECK("tt_transpose_small_kernel");
        RC(amds_bgemm_f32(dlt, Bb, 0, 0, clsn, Cd, 0, 0, 0, G->fc2_w, Cd, 0, 0, 1, 1, C, Cd, Bb, 1.0f, 0.0f, nullptr, 0, stream));
        RC(colsum(dlogits, C, G->fc2_b, Bb, C));
    }
    RC(amds_bgemm_f32(dlogits, C, 0, 0, w.fc2_w, Cd, 0, 0, 0, dcls, Cd, 0, 0, 1, 1, Bb, Cd, C, 1.0f, 0.0f, nullptr, 0, stream));
    AMDS_HIP(hipMemsetAsync(dx, 0, (size_t)M * Cd * 4, st));
    RC(amds_layernorm_bwd(dcls, Cd, xf, (long)n * Cd, reinterpret_cast<const float*>(sv + s.muf), reinterpret_cast<const float*>(sv + s.rsf), w.norm_w, dx, (long)n * Cd, 0,
                          G ? G->norm_w : scratch, G ? G->norm_b : scratch + Cd, 0, Bb, Cd, wk + k.lnb, k.lnb_bytes, stream));
    // ---- layer2
    amds_nystrom_grads g2{nullptr, nullptr, nullptr, nullptr}, g1 = g2;
    if (G) { g2 = amds_nystrom_grads{G->layer[1].qkv_w, G->layer[1].out_w, G->layer[1].out_b, G->layer[1].conv_w}; g1 = amds_nystrom_grads{G->layer[0].qkv_w, G->layer[0].out_w, G->layer[0].out_b, G->layer[0].conv_w}; }
    RC(nystrom_attn_bwd_ex(&w.layer[1], Cd, dx, dy, G ? &g2 : nullptr, Bb, n, p_drop, seed, 2, sv + s.ny2, s.ny_bytes, wk + k.ny, k.ny_bytes, tt_cls_tail(cfg_host), stream));
    RC(amds_layernorm_bwd(dy, Cd, reinterpret_cast<const float*>(sv + s.x2), Cd, reinterpret_cast<const float*>(sv + s.mu2), reinterpret_cast<const float*>(sv + s.rs2),
                          w.layer[1].norm_w, dx, Cd, 1, G ? G->layer[1].norm_w : scratch, G ? G->layer[1].norm_b : scratch + Cd, 0, (int)M, Cd, wk + k.lnb, k.lnb_bytes, stream));
    // ---- PPEG: tap correlations for the weights, the same convolutions with flipped kernels and zero biases for the data
    if (G) RC(amds_ppeg_wgrad(reinterpret_cast<const float*>(sv + s.xp), dx, G->ppeg_corr, Bb, d.side, d.side, Cd, wk + k.pw, k.pw_bytes, stream));
    float* zb = reinterpret_cast<float*>(wk + k.zb);
    AMDS_HIP(hipMemsetAsync(zb, 0, (size_t)Cd * 4, st));
    float* wf7 = reinterpret_cast<float*>(wk + k.wflip);
    float *wf5 = wf7 + (size_t)Cd * 49, *wf3 = wf5 + (size_t)Cd * 25;
    const float* pws[3] = {w.ppeg_w7, w.ppeg_w5, w.ppeg_w3};
    float* pfs[3] = {wf7, wf5, wf3};
    const int taps[3] = {49, 25, 9};
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(tt_flip_taps_kernel, dim3((unsigned)(((long)Cd * taps[i] + 255) / 256)), dim3(256), 0, st, pws[i], pfs[i], Cd, taps[i]);
        AMDS_LAUNCH_CHECK("tt_flip_taps_kernel");
    }
    RC(amds_ppeg(dx, dx2, wf7, zb, wf5, zb, wf3, zb, Bb, d.side, d.side, Cd, stream));
    std::swap(dx, dx2);
    // ---- layer1
    RC(amds_nystrom_attn_bwd(&w.layer[0], Cd, dx, dy, G ? &g1 : nullptr, Bb, n, p_drop, seed, 1, sv + s.ny1, s.ny_bytes, wk + k.ny, k.ny_bytes, stream));
    RC(amds_layernorm_bwd(dy, Cd, reinterpret_cast<const float*>(sv + s.x1), Cd, reinterpret_cast<const float*>(sv + s.mu1), reinterpret_cast<const float*>(sv + s.rs1),
                          w.layer[0].norm_w, dx, Cd, 1, G ? G->layer[0].norm_w : scratch, G ? G->layer[0].norm_b : scratch + Cd, 0, (int)M, Cd, wk + k.lnb, k.lnb_bytes, stream));
    // ---- class token, wrap padding, ReLU, _fc1
    if (G) RC(colsum(dx, (long)n * Cd, G->cls_token, Bb, Cd));
    float *dh = reinterpret_cast<float*>(wk + k.dh), *dzh = reinterpret_cast<float*>(wk + k.dzh);
    hipLaunchKernelGGL(tt_tile_rows_kernel, dim3((unsigned)Mt), dim3(128), 0, st, dx, dh, Cd, T, n);
    AMDS_LAUNCH_CHECK("tt_tile_rows_kernel");
    if (d.add)          // the wrap padding repeats the first tiles (:306-309): their gradients add up (amds_dropout_add with p = 0 is a plain add)
        RC(amds_dropout_add(dx + (size_t)(1 + T) * Cd, (long)n * Cd, dh, (long)T * Cd, dh, (long)T * Cd, Bb, d.add * Cd, 0.0f, 0, 0, stream));
    RC(amds_relu_bwd(reinterpret_cast<const float*>(sv + s.h), dh, dzh, Mt * Cd, stream));
    const float* a = reinterpret_cast<const float*>(sv + s.a);
    if (G) {      // dW1[Cd][F] = sum_b dzh_b^T a_b: one product per bag into partials, then a fixed-order sum over the bags
        float* part = reinterpret_cast<float*>(wk + k.part);
        RC(amds_bgemm_f32(dzh, Cd, (long)T * Cd, 0, a, F, (long)T * F, 0, 2, part, F, (long)Cd * F, 0, Bb, 1, Cd, F, T, 1.0f, 0.0f, nullptr, 0, stream));
        RC(colsum(part, (long)Cd * F, G->fc1_w, Bb, Cd * F));
        RC(colsum(dzh, Cd, G->fc1_b, Mt, Cd));
    }
    if (dbags) RC(amds_bgemm_f32(dzh, Cd, 0, 0, w.fc1_w, F, 0, 0, 0, dbags, F, 0, 0, 1, 1, (int)Mt, F, Cd, 1.0f, 0.0f, nullptr, 0, stream));
    return AMDS_OK;
}
