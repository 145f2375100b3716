// transmil_fwd.hip -- the deploy / validation forward of the TransMIL head as ONE call: bags of tile features -> logits.
//
// Restates the eval-mode forward of the reference's TransMIL (src/stamp/modeling/models/trans_mil.py):
//   _fc1 (Linear + ReLU)                                                                     :290, :303
//   wrap-padding to a square token grid with the FIRST tiles, class token in front           :306-314
//   layer1: x += NystromAttention(LayerNorm(x))                                              :260-263, :317, :81-163 (mask = None)
//   PPEG: x += dw7x7(x) + dw5x5(x) + dw3x3(x) on the grid, class token passed through        :274-283, :318
//   layer2, final LayerNorm, class-token row, _fc2                                           :319-325
// Everything is fp32 like the reference (the pseudo-inverse iteration is numerically touchy): matrix products on the exact-fp32 MFMA
// (amds_bgemm_f32), the rest in the kernels of transmil.hip.  A plain sequence of launches on `stream` over one caller-owned workspace.
#include <algorithm>
#include "common.h"
#include <cstdlib>

namespace amds {
namespace {

inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr int HEADS = 8, ITERS = 6, CONV_K = 33;          // trans_mil.py:252-254 (heads = 8, pinv_iterations = 6), :52 (residual_conv_kernel = 33)

struct TmPlan {
    int Cd, d, m, side, n, pad, np, l;
    size_t hf, x, y, yp, qkv, ql, kl, a1, a2, a3, z, z2, xz, t1, t2, av, a1z, merged, scratch, cls, total;
};

int tm_plan(const amds_transmil_cfg* c, int Bb, int T, TmPlan* p) {
    AMDS_REQUIRE(c, "amds_transmil: null config");
    AMDS_REQUIRE(c->n_feats > 0 && c->dim > 0 && c->dim % 8 == 0 && c->classes > 0, "amds_transmil: bad config (dim_hidden must be a multiple of 8)");
    AMDS_REQUIRE(Bb >= 0 && T >= 1, "amds_transmil: bad shape bags=%d tiles=%d (empty bag)", Bb, T);
    AMDS_REQUIRE((long)Bb * HEADS <= 65535, "amds_transmil: %d bags x 8 heads exceed one launch's batch dimension", Bb);
    p->Cd = c->dim;
    p->d = c->dim / HEADS;
    p->m = c->dim / 2;                                              // num_landmarks = dim // 2 (:253)
    p->side = (int)ceil(sqrt((double)T));
    while ((long)p->side * p->side < T) ++p->side;
    while (p->side > 1 && (long)(p->side - 1) * (p->side - 1) >= T) --p->side;
    p->n = p->side * p->side + 1;
    const int rem = p->n % p->m;
    p->pad = rem > 0 ? p->m - rem : 0;                              // FRONT padding to a multiple of the landmark count (:96-100)
    p->np = p->n + p->pad;
    p->l = (p->n + p->m - 1) / p->m;                                // l = ceil(n / m) (:113)
    const size_t b = Bb, H = HEADS, np = p->np, m = p->m, d = p->d, Cd = p->Cd;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p->hf = take(b * T * (size_t)c->n_feats * 4);
    p->x = take(b * p->n * Cd * 4);
    p->y = take(b * p->n * Cd * 4);
    p->yp = take(b * np * Cd * 4);
    p->qkv = take(b * np * 3 * Cd * 4);
    p->ql = take(b * H * m * d * 4);
    p->kl = take(b * H * m * d * 4);
    p->a1 = take(b * H * np * m * 4);
    p->a2 = take(b * H * m * m * 4);
    p->a3 = take(b * H * m * np * 4);
    p->z = take(b * H * m * m * 4);
    p->z2 = take(b * H * m * m * 4);
    p->xz = take(b * H * m * m * 4);
    p->t1 = take(b * H * m * m * 4);
    p->t2 = take(b * H * m * m * 4);
    p->av = take(b * H * m * d * 4);
    p->a1z = take(b * H * np * m * 4);
    p->merged = take(b * np * Cd * 4);
    p->scratch = take(256);
    p->cls = take(b * Cd * 4);
    p->total = off;
    return AMDS_OK;
}

template <typename TI>
__global__ void __launch_bounds__(256) to_f32_kernel(const TI* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}

// token rows of x [Bb][n][Cd]: row 0 = class token, rows 1..T = the projected tiles, rows T+1..side^2 = the FIRST tiles again (:306-314)
__global__ void __launch_bounds__(128) wrap_cls_kernel(const float* __restrict__ cls, const float* __restrict__ h, float* __restrict__ x, int Cd, int T, int n, int relu) {
    const long row = blockIdx.x;
    const long b = row / n;
    const int s = (int)(row - b * n);
    const float* src = s == 0 ? cls : h + (b * T + (s - 1 < T ? s - 1 : s - 1 - T)) * Cd;
    float* dst = x + row * Cd;
    if (relu && s != 0) { for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = fmaxf(src[c], 0.f); return; }      // (h holds the pre-activation of _fc1)
    for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = src[c];
}

// y [Bb][n][Cd] -> yp [Bb][pad + n][Cd], `pad` zero rows in FRONT of every bag (:100)
__global__ void __launch_bounds__(128) front_pad_kernel(const float* __restrict__ y, float* __restrict__ yp, int Cd, int n, int pad) {
    const int np = n + pad;
    const long row = blockIdx.x;
    const long b = row / np;
    const int s = (int)(row - b * np);
    float* dst = yp + row * Cd;
    if (s < pad) {
        for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = 0.f;
    } else {
        const float* src = y + (b * n + s - pad) * Cd;
        for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = src[c];
    }
}

#define RC(call)                          \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != AMDS_OK) return rc__; \
    } while (0)

int bg(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb, float* Cm, int ldc, long sCo, long sCi, int outer,
       int inner, int M, int N, int K, float alpha, float diag, const float* bias, int accumulate, void* st) {
    return amds_bgemm_f32(A, lda, sAo, sAi, B, ldb, sBo, sBi, transb, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, diag, bias, accumulate, st);
}

// x_res += to_out(NystromAttention(y))   (:81-163 with mask = None, eval mode; residual :263)
// cls_only: the caller reads the class token's row of x_res and nothing else (the second TransLayer: `self.norm(h)[:, 0]`, trans_mil.py:319-323) -- attn1, attn1 z,
// (attn1 z)(attn3 v), the residual convolution and to_out are computed for that one row per bag (row `pad` of the front-padded sequence); landmarks, attn2 and its
// pseudo-inverse, attn3 and attn3 v need every token and stay as they are.
int nystrom(const TmPlan& p, const amds_transmil_layer& L, const float* y, float* x_res, int b, char* wk, void* stream, bool cls_only = false) {
    hipStream_t st = (hipStream_t)stream;
    const int Cd = p.Cd, H = HEADS, m = p.m, d = p.d, n = p.n, np = p.np, pad = p.pad, l = p.l;
    const float* yp = y;
    if (pad) {
        float* ypb = reinterpret_cast<float*>(wk + p.yp);
        hipLaunchKernelGGL(front_pad_kernel, dim3((unsigned)((long)b * np)), dim3(128), 0, st, y, ypb, Cd, n, pad);
        AMDS_LAUNCH_CHECK("front_pad_kernel");
        yp = ypb;
    }
    float* qkv = reinterpret_cast<float*>(wk + p.qkv);
    RC(bg(yp, Cd, 0, 0, L.qkv_w, Cd, 0, 0, 1, qkv, 3 * Cd, 0, 0, 1, 1, b * np, 3 * Cd, Cd, 1.0f, 0.0f, nullptr, 0, stream));      // to_qkv (no bias, :64, :104)
    const float *qp = qkv, *kp = qkv + Cd, *vp = qkv + 2 * Cd;
    const long sb = (long)np * 3 * Cd, sh = d;
    const int ld = 3 * Cd;
    const double scale_d = 1.0 / sqrt((double)d);                                                                                // dim_head ** -0.5 (:61)
    const float scale = (float)scale_d;
    float *ql = reinterpret_cast<float*>(wk + p.ql), *kl = reinterpret_cast<float*>(wk + p.kl);
    RC(amds_landmark_mean(qp, sb, sh, ld, ql, b, H, m, l, d, (float)(scale_d / l), stream));                                        // q scaled (:111), landmarks :113-124
    RC(amds_landmark_mean(kp, sb, sh, ld, kl, b, H, m, l, d, (float)(1.0 / l), stream));
    float *a1 = reinterpret_cast<float*>(wk + p.a1), *a2 = reinterpret_cast<float*>(wk + p.a2), *a3 = reinterpret_cast<float*>(wk + p.a3);
    const long md = (long)m * d, mm = (long)m * m, nm = (long)np * m;
    const long hm = (long)H * m;
    if (cls_only) RC(bg(qp + (size_t)pad * ld, ld, sb, sh, kl, d, H * md, md, 1, a1, m, hm, m, b, H, 1, m, d, scale, 0.0f, nullptr, 0, stream));      // the class row of sim1
    else
    RC(bg(qp, ld, sb, sh, kl, d, H * md, md, 1, a1, m, H * nm, nm, b, H, np, m, d, scale, 0.0f, nullptr, 0, stream));              // sim1 = q kl^T (:126-128)
    RC(bg(ql, d, H * md, md, kl, d, H * md, md, 1, a2, m, H * mm, mm, b, H, m, m, d, 1.0f, 0.0f, nullptr, 0, stream));             // sim2 = ql kl^T
    RC(bg(ql, d, H * md, md, kp, ld, sb, sh, 1, a3, np, H * nm, nm, b, H, m, np, d, 1.0f, 0.0f, nullptr, 0, stream));              // sim3 = ql k^T
    RC(amds_softmax_rows(a1, cls_only ? (long)b * H : (long)b * H * np, m, stream));                                               // :145
    RC(amds_softmax_rows(a2, (long)b * H * m, m, stream));
    RC(amds_softmax_rows(a3, (long)b * H * m, np, stream));
    // Moore-Penrose iteration (:23-37): z <- 0.25 z (13 I - xz (15 I - xz (7 I - xz))),  xz = x z
    float *z = reinterpret_cast<float*>(wk + p.z), *z2 = reinterpret_cast<float*>(wk + p.z2), *xz = reinterpret_cast<float*>(wk + p.xz);
    float *t1 = reinterpret_cast<float*>(wk + p.t1), *t2 = reinterpret_cast<float*>(wk + p.t2);
    AMDS_HIP(hipMemsetAsync(wk + p.scratch, 0, 8, st));
    RC(amds_pinv_init(a2, z, b * H, m, wk + p.scratch, stream));
    // The chain runs chunk by chunk of the batch, 256 matrices (1024 workgroups per launch) at a time: the deploy forward keeps no iterate, so a chunk's five buffers
    // are re-used while parts of them are still in the Infinity Cache -- 6 540 -> 6 720 bags/s at 64 bags (512 matrices: two chunks; four chunks of 128: +1.8 %, eight:
    // -13 %: too few workgroups per launch).  The training forward keeps every iterate for the backward and gains nothing (profiles/r06_transmil_prefetch_ab.txt).
    // AMDS_PINV_CHUNKS=n: n chunks (1 = the whole batch at once), A/B.
    static const int n_chunks = getenv("AMDS_PINV_CHUNKS") ? atoi(getenv("AMDS_PINV_CHUNKS")) : 0;
    const int Zall = b * H, Zc = n_chunks > 0 ? (Zall + n_chunks - 1) / n_chunks : std::min(Zall, 256);
    float *z_in = z, *z2_in = z2;
    for (int c0 = 0; c0 < Zall; c0 += Zc) {
        const int zn = std::min(Zc, Zall - c0);
        const long co = (long)c0 * mm;
        z = z_in; z2 = z2_in;
        auto sq = [&](const float* A, const float* B, float* Cm, float alpha, float diag) {
            return bg(A + co, m, mm, 0, B + co, m, mm, 0, 0, Cm + co, m, mm, 0, zn, 1, m, m, m, alpha, diag, nullptr, 0, stream);
        };
        for (int it = 0; it < ITERS; ++it) {
            RC(amds_bgemm_f32_dual(a2 + co, m, mm, 0, z + co, m, mm, 0, 0, xz + co, t1 + co, m, mm, 0, zn, 1, m, m, m, 1.0f, 0.0f, -1.0f, 7.0f, stream));      // xz and 7 I - xz from one product
            RC(sq(xz, t1, t2, -1.0f, 15.0f));
            RC(sq(xz, t2, t1, -1.0f, 13.0f));
            RC(sq(z, t1, z2, 0.25f, 0.0f));
            std::swap(z, z2);
        }
    }
    float *av = reinterpret_cast<float*>(wk + p.av), *a1z = reinterpret_cast<float*>(wk + p.a1z), *merged = reinterpret_cast<float*>(wk + p.merged);
    RC(bg(a3, np, H * nm, nm, vp, ld, sb, sh, 0, av, d, H * md, md, b, H, m, d, np, 1.0f, 0.0f, nullptr, 0, stream));               // attn3 v
    if (cls_only) {
        RC(bg(a1, m, hm, m, z, m, H * mm, mm, 0, a1z, m, hm, m, b, H, 1, m, m, 1.0f, 0.0f, nullptr, 0, stream));                     // one row of attn1 pinv
        RC(bg(a1z, m, hm, m, av, d, H * md, md, 0, merged, Cd, Cd, d, b, H, 1, d, m, 1.0f, 0.0f, nullptr, 0, stream));               // merged: [b][Cd], the class rows
        RC(amds_dwconv_seq_row(vp, sb, sh, ld, L.conv_w, merged, Cd, d, b, H, np, d, CONV_K, pad, stream));
        return bg(merged, Cd, Cd, 0, L.out_w, Cd, 0, 0, 1, x_res, Cd, (long)n * Cd, 0, b, 1, 1, Cd, Cd, 1.0f, 0.0f, L.out_b, 1, stream);
    }
    RC(bg(a1, m, H * nm, nm, z, m, H * mm, mm, 0, a1z, m, H * nm, nm, b, H, np, m, m, 1.0f, 0.0f, nullptr, 0, stream));             // attn1 pinv
    RC(bg(a1z, m, H * nm, nm, av, d, H * md, md, 0, merged, Cd, (long)np * Cd, d, b, H, np, d, m, 1.0f, 0.0f, nullptr, 0, stream)); // heads merged (:148-153)
    RC(amds_dwconv_seq(vp, sb, sh, ld, L.conv_w, merged, (long)np * Cd, d, Cd, b, H, np, d, CONV_K, stream));                       // + res_conv(v) (:151)
    // to_out on the LAST n rows of every bag (:154-155), accumulated into the residual stream
    return bg(merged + (size_t)pad * Cd, Cd, (long)np * Cd, 0, L.out_w, Cd, 0, 0, 1, x_res, Cd, (long)n * Cd, 0, b, 1, n, Cd, Cd, 1.0f, 0.0f, L.out_b, 1, stream);
}

}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_transmil_workspace_bytes(const amds_transmil_cfg* cfg_host, int n_bags, int n_tiles) {
    TmPlan p;
    if (tm_plan(cfg_host, n_bags, n_tiles, &p) != AMDS_OK) return 0;
    return p.total;
}

extern "C" int amds_transmil_forward(const amds_transmil_cfg* cfg_host, const amds_transmil_weights* w_host, const void* bags, int bags_dtype, float* logits,
                                     int n_bags, int n_tiles, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && bags && logits && ws, "amds_transmil_forward: null pointer");
    TmPlan p;
    RC(tm_plan(cfg_host, n_bags, n_tiles, &p));
    const amds_transmil_weights& w = *w_host;
    AMDS_REQUIRE(w.fc1_w && w.fc1_b && w.cls_token && w.norm_w && w.norm_b && w.fc2_w && w.fc2_b && w.ppeg_w7 && w.ppeg_b7 && w.ppeg_w5 && w.ppeg_b5 &&
                 w.ppeg_w3 && w.ppeg_b3, "amds_transmil_forward: incomplete weights");
    for (int i = 0; i < 2; ++i)
        AMDS_REQUIRE(w.layer[i].norm_w && w.layer[i].norm_b && w.layer[i].qkv_w && w.layer[i].out_w && w.layer[i].out_b && w.layer[i].conv_w,
                     "amds_transmil_forward: incomplete weights of layer %d", i + 1);
    AMDS_REQUIRE(bags_dtype == AMDS_F32 || bags_dtype == AMDS_F16 || bags_dtype == AMDS_BF16, "amds_transmil_forward: bad bags dtype %d", bags_dtype);
    if (ws_bytes < p.total) {
        set_error("amds_transmil_forward: workspace %zu < required %zu bytes", ws_bytes, p.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_transmil_forward: workspace must be 256-byte aligned");
    if (n_bags == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    char* wk = reinterpret_cast<char*>(ws);
    const int Bb = n_bags, T = n_tiles, Cd = p.Cd, F = cfg_host->n_feats, n = p.n;
    const float* hf = reinterpret_cast<const float*>(bags);
    if (bags_dtype != AMDS_F32) {                                                         // the reference casts the bag to float (models/__init__.py:308-313)
        const long cnt = (long)Bb * T * F;
        const int grid = (int)std::min<long>(8192, (cnt + 255) / 256);
        float* dst = reinterpret_cast<float*>(wk + p.hf);
        if (bags_dtype == AMDS_F16) hipLaunchKernelGGL((to_f32_kernel<f16>), dim3(grid), dim3(256), 0, st, (const f16*)bags, dst, cnt);
        else hipLaunchKernelGGL((to_f32_kernel<bf16>), dim3(grid), dim3(256), 0, st, (const bf16*)bags, dst, cnt);
        AMDS_LAUNCH_CHECK("to_f32_kernel");
        hf = dst;
    }
    float *x = reinterpret_cast<float*>(wk + p.x), *y = reinterpret_cast<float*>(wk + p.y);
    float* h1 = reinterpret_cast<float*>(wk + p.qkv);                                    // _fc1 output [Bb*T][Cd]: scratch (the qkv region is larger and free here)
    // _fc1 = Linear + ReLU (:303).  Below torch's "highest" the product is a product like the others (amds_bgemm_f32: bf16 x 3 at "high") and the ReLU rides on the
    // copy into the wrapped sequence; at "highest" the exact-fp32 Linear of the MLP heads (a 64 x 64-tile kernel: 700 us at 65 536 x 1024 x 512 against ~400)
    const bool fc1_x3 = ctx_matmul_precision() != AMDS_MATMUL_HIGHEST && ((long)Bb * T) % 128 == 0 && Cd % 128 == 0 && F % 32 == 0;
    if (fc1_x3) RC(bg(hf, F, 0, 0, w.fc1_w, F, 0, 0, 1, h1, Cd, 0, 0, 1, 1, Bb * T, Cd, F, 1.0f, 0.0f, w.fc1_b, 0, stream));
    else RC(amds_linear_f32(hf, w.fc1_w, w.fc1_b, h1, Bb * T, Cd, F, 1, stream));
    hipLaunchKernelGGL(wrap_cls_kernel, dim3((unsigned)((long)Bb * n)), dim3(128), 0, st, w.cls_token, h1, x, Cd, T, n, fc1_x3 ? 1 : 0);
    AMDS_LAUNCH_CHECK("wrap_cls_kernel");
    // layer1, PPEG, layer2 (:317-319)
    RC(amds_layernorm(x, Cd, w.layer[0].norm_w, w.layer[0].norm_b, y, Cd, Bb * n, Cd, 1e-5f, AMDS_F32, stream));
    RC(nystrom(p, w.layer[0], y, x, Bb, wk, stream));
    RC(amds_ppeg(x, y, w.ppeg_w7, w.ppeg_b7, w.ppeg_w5, w.ppeg_b5, w.ppeg_w3, w.ppeg_b3, Bb, p.side, p.side, Cd, stream));
    std::swap(x, y);
    RC(amds_layernorm(x, Cd, w.layer[1].norm_w, w.layer[1].norm_b, y, Cd, Bb * n, Cd, 1e-5f, AMDS_F32, stream));
    RC(nystrom(p, w.layer[1], y, x, Bb, wk, stream, ctx_mil_cls_tail() != 0));      // (amds_set_mil_cls_tail(0): every row, A/B and the chain tests)
    // final LayerNorm on the class-token rows, _fc2 (:322-325)
    float* cls = reinterpret_cast<float*>(wk + p.cls);
    RC(amds_layernorm(x, (long)n * Cd, w.norm_w, w.norm_b, cls, Cd, Bb, Cd, 1e-5f, AMDS_F32, stream));
    return amds_linear_f32(cls, w.fc2_w, w.fc2_b, logits, Bb, cfg_host->classes, Cd, 0, stream);
}
