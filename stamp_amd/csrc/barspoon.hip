// barspoon.hip -- the deploy / validation forward of the reference's barspoon head (`EncDecTransformer`) as ONE call:
// bags of tile features + tile positions -> one logit vector per target.
//
// Restates src/stamp/modeling/models/barspoon.py in eval mode:
//   projector (Linear + ReLU) :171, sinusoidal encoding of the tile positions added :173-186,
//   pre-norm nn.TransformerEncoder over the tiles :188 (per layer: x += SA(LN1(x)); x += W2 relu(W1 LN2(x))),
//   one learned class token per target decoded against the tile tokens by a pre-norm nn.TransformerDecoder :190-193
//   (per layer: t += SA(LN1(t)); t += MHA(LN2(t), memory = tiles); t += W2 relu(W1 LN3(t))), one Linear head per target :196-203.
// The tile side (projector, encoder, the K / V projections of the cross-attention: everything with a tile dimension) runs on the 16-bit
// MFMA GEMMs and the streaming attention kernel, with the zero-padded weight layout of the MIL `vit` head (amds_mil_vit_layer: widths to
// 256, heads to 64 channels); the class-token side (n_targets rows per bag) runs in exact fp32 (amds_bgemm_f32).  The cross-attention is
// its own kernel: one fp32 query row per (bag, target, head) streamed over the bag's 16-bit keys / values with an online softmax.
#include <algorithm>
#include "common.h"

namespace amds {
namespace {

inline int up(int n, int m) { return (n + m - 1) / m * m; }
inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

struct BsPlan {
    int Fp, Dp, FFp, Ha, Da, Hb, Db, hd_e, hd_d, nt, n_out_total;
    size_t a, x, h, qkv, att, u, kv, tok, th, tqkv, tsc, to, tq, tu, total;
};

int bs_plan(const amds_barspoon_cfg* c, int Bb, int T, BsPlan* p) {
    AMDS_REQUIRE(c, "amds_barspoon: null config");
    AMDS_REQUIRE(c->n_feats > 0 && c->dim > 0 && c->enc_heads > 0 && c->dec_heads > 0 && c->ff > 0 && c->enc_layers >= 0 && c->dec_layers >= 0 && c->n_targets > 0,
                 "amds_barspoon: bad config");
    AMDS_REQUIRE(c->dim % c->enc_heads == 0 && c->dim % c->dec_heads == 0, "amds_barspoon: d_model=%d has to be divisible by the head counts (%d, %d)", c->dim,
                 c->enc_heads, c->dec_heads);
    AMDS_REQUIRE(c->dim / c->enc_heads <= 64 && c->dim / c->dec_heads <= 64 && c->dim % 4 == 0, "amds_barspoon: needs head_dim <= 64 and d_model %% 4 == 0");
    AMDS_REQUIRE(c->n_targets <= 1024, "amds_barspoon: %d targets", c->n_targets);
    AMDS_REQUIRE(c->dtype == AMDS_F16 || c->dtype == AMDS_BF16, "amds_barspoon: operand dtype must be f16 or bf16");
    AMDS_REQUIRE(Bb >= 0 && T > 0, "amds_barspoon: bad shape bags=%d tiles=%d", Bb, T);
    p->Fp = up(c->n_feats, 256); p->Dp = up(c->dim, 256); p->FFp = up(c->ff, 256);
    p->Ha = up(c->enc_heads, 4); p->Da = 64 * p->Ha;                 // encoder self-attention: padded heads (amds_attention wants H % 4 == 0)
    p->Hb = c->dec_heads; p->Db = 64 * p->Hb;                        // cross-attention K / V: decoder heads padded to 64 channels
    p->hd_e = c->dim / c->enc_heads; p->hd_d = c->dim / c->dec_heads;
    p->nt = c->n_targets;
    const size_t M = (size_t)Bb * T, M2 = (size_t)Bb * p->nt, D = c->dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p->a = take(M * p->Fp * 2);
    p->x = take(M * p->Dp * 4);
    p->h = take(M * p->Dp * 2);
    p->qkv = take(M * 3 * p->Da * 2);
    p->att = take(M * p->Da * 2);
    p->u = take(M * p->FFp * 2);
    p->kv = take(M * 2 * p->Db * 2);
    p->tok = take(M2 * D * 4);
    p->th = take(M2 * D * 4);
    p->tqkv = take(M2 * 3 * D * 4);
    p->tsc = take((size_t)Bb * p->Hb * p->nt * p->nt * 4);
    p->to = take(M2 * D * 4);
    p->tq = take(M2 * D * 4);
    p->tu = take(M2 * (size_t)c->ff * 4);
    p->total = off;
    return AMDS_OK;
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) bs_stage_kernel(const TI* __restrict__ src, long ld_src, TO* __restrict__ dst, int Fp, long total, int F) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / Fp;
        const int c = (int)(i - r * Fp);
        dst[i] = c < F ? (TO)(float)src[r * ld_src + c] : (TO)0.f;
    }
}

// x[r][c] += PE(pos[r])[c] for c < D:  [ sin(px / f_i) | sin(py / f_i) | cos(px / f_i) | cos(py / f_i) ],  i < D / 4,  f_i = pe_div[i] (:173-186)
__global__ void __launch_bounds__(256) pos_encoding_add_kernel(float* __restrict__ x, int Dp, int D, const float* __restrict__ pos, const float* __restrict__ pe_div,
                                                                long rows) {
    const int q = D / 4;
    const long total = rows * D;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / D;
        const int c = (int)(i - r * D);
        const int blk = c / q, f = c - blk * q;                // blk: 0 sin x, 1 sin y, 2 cos x, 3 cos y
        const float a = pos[2 * r + (blk & 1)] / pe_div[f];
        x[r * Dp + c] += blk < 2 ? sinf(a) : cosf(a);
    }
}

__global__ void __launch_bounds__(256) broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long per_bag, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) dst[i] = src[i % per_bag];
}

// Cross-attention of the class tokens: out[b][j][h*hd..] = softmax(q[b][j][h*hd..] . K_b,h^T / sqrt(hd)) V_b,h.  q fp32 [B][nt][D]; kv 16-bit
// [B*T][2*Db], K of head h at columns 64h.., V at Db + 64h.. (channels >= hd are zero padding).  One wave per (b, j, h): lane l streams the keys
// l, l + 64, ... with its own running (max, sum, o[64]); the 64 partial states are merged once at the end.
template <typename T>
__global__ void __launch_bounds__(256) cross_attention_kernel(const float* __restrict__ q, const T* __restrict__ kv, float* __restrict__ out, int B, int Tn, int nt,
                                                               int H, int hd, int D, int Db) {
    typedef T vec8 __attribute__((ext_vector_type(8)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)B * nt * H) return;                      // whole waves leave; no workgroup barrier below
    const int h = (int)(item % H);
    const long bj = item / H;
    const int b = (int)(bj / nt);
    const float* qr = q + bj * D + (long)h * hd;
    const float sc = rsqrtf((float)hd) * 1.44269504088896340736f;          // log2 domain
    float qv[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) qv[e] = e < hd ? qr[e] * sc : 0.f;
    float m = -INFINITY, l = 0.f, o[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) o[e] = 0.f;
    const long ld = 2L * Db;
    const T* kb = kv + (long)b * Tn * ld + 64 * h;
    for (int t = lane; t < Tn; t += 64) {
        const T* kr = kb + t * ld;
        float s = 0.f;
#pragma unroll
        for (int pz = 0; pz < 8; ++pz) {
            const vec8 kk = *reinterpret_cast<const vec8*>(kr + 8 * pz);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[8 * pz + e], (float)kk[e], s);
        }
        const float mn = fmaxf(m, s);
        const float corr = exp2f(m - mn), pw = exp2f(s - mn);
        l = l * corr + pw;
        const T* vr = kr + Db;
#pragma unroll
        for (int pz = 0; pz < 8; ++pz) {
            const vec8 vv = *reinterpret_cast<const vec8*>(vr + 8 * pz);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[8 * pz + e] = fmaf(o[8 * pz + e], corr, pw * (float)vv[e]);
        }
        m = mn;
    }
    // merge the 64 lanes' states
    float mg = m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mg = fmaxf(mg, __shfl_xor(mg, off, 64));
    const float w = m == -INFINITY ? 0.f : exp2f(m - mg);
    l *= w;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
    const float inv = 1.0f / l;
    float* orow = out + bj * D + (long)h * hd;
#pragma unroll
    for (int e = 0; e < 64; ++e) {
        float v = o[e] * w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && e < hd) orow[e] = v * inv;
    }
}

#define RC(call)                          \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != AMDS_OK) return rc__; \
    } while (0)

inline int bg(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int tflags, float* Cm, int ldc, long sCo, long sCi,
              int outer, int inner, int M, int N, int K, float alpha, const float* bias, int accumulate, void* st) {
    return bgemm_f32_exact(A, lda, sAo, sAi, B, ldb, sBo, sBi, tflags, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, 0.0f, bias, accumulate, st);
}

}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_barspoon_workspace_bytes(const amds_barspoon_cfg* cfg_host, int n_bags, int n_tiles) {
    BsPlan p;
    if (bs_plan(cfg_host, n_bags, n_tiles, &p) != AMDS_OK) return 0;
    return p.total;
}

extern "C" int amds_barspoon_forward(const amds_barspoon_cfg* cfg_host, const amds_barspoon_weights* w_host, const void* bags, int bags_dtype,
                                     const float* positions, float* logits, int n_bags, int n_tiles, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && bags && logits && ws, "amds_barspoon_forward: null pointer");
    const amds_barspoon_cfg& c = *cfg_host;
    const amds_barspoon_weights& w = *w_host;
    BsPlan p;
    RC(bs_plan(cfg_host, n_bags, n_tiles, &p));
    AMDS_REQUIRE(w.proj_w && w.proj_b && w.class_tokens && w.head_w_host && w.head_b_host && w.n_out_host && (c.enc_layers == 0 || w.enc_layers_host) &&
                 (c.dec_layers == 0 || w.dec_layers_host), "amds_barspoon_forward: incomplete weights");
    AMDS_REQUIRE(!c.positional_encoding || (positions && w.pe_div), "amds_barspoon_forward: positional_encoding=True needs tile positions");
    AMDS_REQUIRE(bags_dtype == AMDS_F32 || bags_dtype == AMDS_F16 || bags_dtype == AMDS_BF16, "amds_barspoon_forward: bad bags dtype %d", bags_dtype);
    if (ws_bytes < p.total) {
        set_error("amds_barspoon_forward: workspace %zu < required %zu bytes", ws_bytes, p.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_barspoon_forward: workspace must be 256-byte aligned");
    if (n_bags == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    char* base = reinterpret_cast<char*>(ws);
    const int Bb = n_bags, T = n_tiles, D = c.dim, Dp = p.Dp, dt = c.dtype, nt = p.nt, Hd = p.Hb, hd = p.hd_d, FF = c.ff;
    const long M = (long)Bb * T, M2 = (long)Bb * nt;
    AMDS_REQUIRE(M < (1L << 31) - 65536, "amds_barspoon_forward: %ld tile rows do not fit the 32-bit row index", M);
    float* x = reinterpret_cast<float*>(base + p.x);
    void *h = base + p.h, *qkv = base + p.qkv, *att = base + p.att, *u = base + p.u, *kv = base + p.kv;

    // ---- projector: Linear + ReLU (:171), positional encodings (:173-186)
    const void* a = bags;
    if (!(bags_dtype == dt && c.n_feats == p.Fp)) {
        const long total = M * p.Fp;
        const int grid = (int)std::min<long>(8192, (total + 255) / 256);
#define STAGE(TI, TO) hipLaunchKernelGGL((bs_stage_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)bags, (long)c.n_feats, (TO*)(base + p.a), p.Fp, total, c.n_feats)
        if (dt == AMDS_F16) {
            if (bags_dtype == AMDS_F32) STAGE(float, f16); else if (bags_dtype == AMDS_F16) STAGE(f16, f16); else STAGE(bf16, f16);
        } else {
            if (bags_dtype == AMDS_F32) STAGE(float, bf16); else if (bags_dtype == AMDS_F16) STAGE(f16, bf16); else STAGE(bf16, bf16);
        }
#undef STAGE
        AMDS_LAUNCH_CHECK("bs_stage_kernel");
        a = base + p.a;
    }
    RC(amds_gemm(a, p.Fp, w.proj_w, p.Fp, (int)M, Dp, p.Fp, dt, AMDS_EPI_BIAS_RELU_F32, x, Dp, w.proj_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream));
    if (c.positional_encoding) {
        const long total = M * D;
        hipLaunchKernelGGL(pos_encoding_add_kernel, dim3((unsigned)std::min<long>(8192, (total + 255) / 256)), dim3(256), 0, st, x, Dp, D, positions, w.pe_div, M);
        AMDS_LAUNCH_CHECK("pos_encoding_add_kernel");
    }
    if (Dp != D) AMDS_HIP(hipMemsetAsync(h, 0, (size_t)M * Dp * 2, st));      // LayerNorm writes the first D columns only

    // ---- encoder (:188): pre-norm layers on the 16-bit MFMA path, the MIL `vit` head's layer with ReLU
    for (int l = 0; l < c.enc_layers; ++l) {
        const amds_mil_vit_layer& L = w.enc_layers_host[l];
        AMDS_REQUIRE(L.ln1_w && L.ln1_b && L.in_w && L.in_b && L.out_w && L.out_b && L.ln2_w && L.ln2_b && L.fc1_w && L.fc1_b && L.fc2_w && L.fc2_b,
                     "amds_barspoon_forward: incomplete weights of encoder layer %d", l);
        RC(amds_layernorm(x, Dp, L.ln1_w, L.ln1_b, h, Dp, (int)M, D, 1e-5f, dt, stream));
        RC(amds_gemm(h, Dp, L.in_w, Dp, (int)M, 3 * p.Da, Dp, dt, AMDS_EPI_BIAS, qkv, 3 * p.Da, L.in_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream));
        RC(amds_attention(qkv, att, Bb, T, p.Ha, dt, stream));
        RC(amds_gemm(att, p.Da, L.out_w, p.Da, (int)M, Dp, p.Da, dt, AMDS_EPI_RESIDUAL, x, Dp, L.out_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream));
        RC(amds_layernorm(x, Dp, L.ln2_w, L.ln2_b, h, Dp, (int)M, D, 1e-5f, dt, stream));
        RC(amds_gemm(h, Dp, L.fc1_w, Dp, (int)M, p.FFp, Dp, dt, AMDS_EPI_BIAS_RELU, u, p.FFp, L.fc1_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream));
        RC(amds_gemm(u, p.FFp, L.fc2_w, p.FFp, (int)M, Dp, p.FFp, dt, AMDS_EPI_RESIDUAL, x, Dp, L.fc2_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream));
    }
    // the encoder's output as the 16-bit A operand of every decoder layer's K / V projection (no final norm: nn.TransformerEncoder(norm=None))
    if (c.dec_layers > 0) RC(amds_cast_pad(x, Dp, h, Dp, (int)M, Dp, dt, stream));

    // ---- decoder (:190-193): the class tokens, fp32
    float *tok = reinterpret_cast<float*>(base + p.tok), *th = reinterpret_cast<float*>(base + p.th), *tqkv = reinterpret_cast<float*>(base + p.tqkv);
    float *tsc = reinterpret_cast<float*>(base + p.tsc), *to = reinterpret_cast<float*>(base + p.to), *tq = reinterpret_cast<float*>(base + p.tq);
    float* tu = reinterpret_cast<float*>(base + p.tu);
    {
        const long total = M2 * D;
        hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)std::min<long>(4096, (total + 255) / 256)), dim3(256), 0, st, w.class_tokens, tok, (long)nt * D, total);
        AMDS_LAUNCH_CHECK("broadcast_rows_kernel");
    }
    const float sa_scale = (float)(1.0 / sqrt((double)hd));
    for (int l = 0; l < c.dec_layers; ++l) {
        const amds_barspoon_dec_layer& L = w.dec_layers_host[l];
        AMDS_REQUIRE(L.ln1_w && L.ln1_b && L.sa_in_w && L.sa_in_b && L.sa_out_w && L.sa_out_b && L.ln2_w && L.ln2_b && L.ca_q_w && L.ca_q_b && L.ca_kv_w && L.ca_kv_b &&
                     L.ca_out_w && L.ca_out_b && L.ln3_w && L.ln3_b && L.fc1_w && L.fc1_b && L.fc2_w && L.fc2_b, "amds_barspoon_forward: incomplete weights of decoder layer %d", l);
        // t += SA(LN1(t)): self-attention among the nt class tokens of a bag
        RC(amds_layernorm(tok, D, L.ln1_w, L.ln1_b, th, D, (int)M2, D, 1e-5f, AMDS_F32, stream));
        RC(amds_linear_f32(th, L.sa_in_w, L.sa_in_b, tqkv, (int)M2, 3 * D, D, 0, stream));
        RC(bg(tqkv, 3 * D, (long)nt * 3 * D, hd, tqkv + D, 3 * D, (long)nt * 3 * D, hd, 1, tsc, nt, (long)Hd * nt * nt, (long)nt * nt, Bb, Hd, nt, nt, hd, sa_scale, nullptr, 0,
              stream));
        RC(amds_softmax_rows(tsc, (long)Bb * Hd * nt, nt, stream));
        RC(bg(tsc, nt, (long)Hd * nt * nt, (long)nt * nt, tqkv + 2 * D, 3 * D, (long)nt * 3 * D, hd, 0, to, D, (long)nt * D, hd, Bb, Hd, nt, hd, nt, 1.0f, nullptr, 0, stream));
        RC(bg(to, D, 0, 0, L.sa_out_w, D, 0, 0, 1, tok, D, 0, 0, 1, 1, (int)M2, D, D, 1.0f, L.sa_out_b, 1, stream));
        // t += MHA(LN2(t), memory): queries from the class tokens, keys / values from the tile tokens
        RC(amds_layernorm(tok, D, L.ln2_w, L.ln2_b, th, D, (int)M2, D, 1e-5f, AMDS_F32, stream));
        RC(amds_linear_f32(th, L.ca_q_w, L.ca_q_b, tq, (int)M2, D, D, 0, stream));
        RC(amds_gemm(h, Dp, L.ca_kv_w, Dp, (int)M, 2 * p.Db, Dp, dt, AMDS_EPI_BIAS, kv, 2 * p.Db, L.ca_kv_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream));
        {
            const long items = M2 * Hd;
            if (dt == AMDS_F16)
                hipLaunchKernelGGL((cross_attention_kernel<f16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, tq, (const f16*)kv, to, Bb, T, nt, Hd, hd, D, p.Db);
            else
                hipLaunchKernelGGL((cross_attention_kernel<bf16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, tq, (const bf16*)kv, to, Bb, T, nt, Hd, hd, D, p.Db);
            AMDS_LAUNCH_CHECK("cross_attention_kernel");
        }
        RC(bg(to, D, 0, 0, L.ca_out_w, D, 0, 0, 1, tok, D, 0, 0, 1, 1, (int)M2, D, D, 1.0f, L.ca_out_b, 1, stream));
        // t += W2 relu(W1 LN3(t))
        RC(amds_layernorm(tok, D, L.ln3_w, L.ln3_b, th, D, (int)M2, D, 1e-5f, AMDS_F32, stream));
        RC(amds_linear_f32(th, L.fc1_w, L.fc1_b, tu, (int)M2, FF, D, 1, stream));
        RC(bg(tu, FF, 0, 0, L.fc2_w, FF, 0, 0, 1, tok, D, 0, 0, 1, 1, (int)M2, D, FF, 1.0f, L.fc2_b, 1, stream));
    }
    // ---- heads (:196-203): target j reads its own class-token row of every bag; logits [Bb][sum n_out], target j at its column offset
    int total_out = 0;
    for (int j = 0; j < nt; ++j) {
        AMDS_REQUIRE(w.n_out_host[j] > 0 && w.head_w_host[j] && w.head_b_host[j], "amds_barspoon_forward: head %d missing", j);
        total_out += w.n_out_host[j];
    }
    int col = 0;
    for (int j = 0; j < nt; ++j) {
        RC(bg(tok + (size_t)j * D, nt * D, 0, 0, w.head_w_host[j], D, 0, 0, 1, logits + col, total_out, 0, 0, 1, 1, Bb, w.n_out_host[j], D, 1.0f, w.head_b_host[j], 0, stream));
        col += w.n_out_host[j];
    }
    return AMDS_OK;
}
