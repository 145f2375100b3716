// mil_vit.hip -- the deploy / validation forward of the MIL `vit` head as ONE call: bags of tile features -> logits.
//
// Restates the eval-mode forward of the reference's VisionTransformer (src/stamp/modeling/models/vision_tranformer.py:332-384):
//   project_features (Linear + GELU, Dropout = identity)                               :342
//   class token prepended, coords (0, 0) for it, padding mask extended by one column   :347-362
//   L x [ x += attention(LayerNorm(x)) ;  x += feed_forward(x) ]                       :290-293 (SelfAttention :194-242, feed_forward :157-169)
//   final LayerNorm, class-token row, mlp_head                                         :294, 382-384
// out of the kernels the rest of the library already exposes one by one (amds_gemm, amds_layernorm, amds_attention*, amds_linear_f32);
// the host only supplies padded device weights (amds_mil_vit_weights) and one workspace.  Nothing is allocated, no host
// synchronisation happens: the call is a plain sequence of launches on `stream` (it can be captured in a hipGraph).
#include "common.h"
#include <atomic>

namespace amds {

namespace {

inline int up(int n, int m) { return (n + m - 1) / m * m; }
inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

struct MilPlan {
    int Fp, Dp, FFp, Ha, Da;
    size_t a, x, h, qkv, att, u, cls, coords, pad, total;
};

int mil_plan(const amds_mil_vit_cfg* c, int Bb, int Tn, MilPlan* p) {
    AMDS_REQUIRE(c->n_feats > 0 && c->dim > 0 && c->heads > 0 && c->ff > 0 && c->classes > 0 && c->layers >= 0, "amds_mil_vit: bad config");
    AMDS_REQUIRE(c->dim % c->heads == 0, "amds_mil_vit: dim_model=%d has to be divisible by n_heads=%d", c->dim, c->heads);
    AMDS_REQUIRE(c->dim / c->heads <= 64 && c->dim % 4 == 0, "amds_mil_vit: needs head_dim <= 64 and dim_model %% 4 == 0 (dim_model=%d, n_heads=%d)",
                 c->dim, c->heads);
    AMDS_REQUIRE(c->dtype == AMDS_F16 || c->dtype == AMDS_BF16, "amds_mil_vit: operand dtype must be f16 or bf16");
    AMDS_REQUIRE(Bb >= 0 && Tn > 0, "amds_mil_vit: bad shape bags=%d tiles=%d (empty bags have no class-token context)", Bb, Tn);
    p->Fp = up(c->n_feats, 256);
    p->Dp = up(c->dim, 256);
    p->FFp = up(c->ff, 256);
    p->Ha = up(c->heads, 4);
    p->Da = 64 * p->Ha;
    const size_t M = (size_t)Bb * (Tn + 1), Mt = (size_t)Bb * Tn;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p->a = take(Mt * p->Fp * 2);                    // staged bags, 16-bit, zero padded columns
    p->x = take(M * p->Dp * 4);                     // residual stream fp32
    p->h = take(M * p->Dp * 2);                     // LayerNorm output
    p->qkv = take(M * 3 * p->Da * 2);
    p->att = take(M * p->Da * 2);
    p->u = take(M * p->FFp * 2);
    p->cls = take((size_t)Bb * c->dim * 4);
    p->coords = take(M * 2 * 4);
    p->pad = take(M);
    p->total = off;
    return AMDS_OK;
}

// bags [Mt][F] (fp32 / f16 / bf16) -> 16-bit operand rows [Mt][Fp], zero padded
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) stage_bags_kernel(const TI* __restrict__ src, long ld_src, TO* __restrict__ dst, int Fp, long total, int F) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / Fp;
        const int c = (int)(i - r * Fp);
        dst[i] = c < F ? (TO)(float)src[r * ld_src + c] : (TO)0.f;
    }
}

// x rows: the class token in front of each bag's projected tiles; coords with the class token at (0, 0); padding mask with a leading 0
// (:347-362).  One block per token row (b, s).
__global__ void __launch_bounds__(128) prefix_cls_kernel(const float* __restrict__ cls, const float* __restrict__ proj, float* __restrict__ x, int Dp,
                                                          const float* __restrict__ coords, float* __restrict__ coords_out,
                                                          const uint8_t* __restrict__ mask, uint8_t* __restrict__ pad, int Tn) {
    const int S = Tn + 1;
    const long row = blockIdx.x;
    const long b = row / S;
    const int s = (int)(row - b * S);
    const f32x4* src = reinterpret_cast<const f32x4*>(s == 0 ? cls : proj + (b * Tn + s - 1) * Dp);
    f32x4* dst = reinterpret_cast<f32x4*>(x + row * Dp);
    for (int c = threadIdx.x; c < Dp / 4; c += 128) dst[c] = src[c];
    if (threadIdx.x == 0) {
        if (coords_out) {
            coords_out[2 * row] = s == 0 ? 0.f : coords[2 * (b * Tn + s - 1)];
            coords_out[2 * row + 1] = s == 0 ? 0.f : coords[2 * (b * Tn + s - 1) + 1];
        }
        if (pad) pad[row] = s == 0 ? (uint8_t)0 : (uint8_t)(mask[b * Tn + s - 1] != 0);
    }
}

}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_mil_vit_workspace_bytes(const amds_mil_vit_cfg* cfg_host, int bags, int tiles) {
    MilPlan p;
    if (!cfg_host || mil_plan(cfg_host, bags, tiles, &p) != AMDS_OK) return 0;
    return p.total;
}

extern "C" int amds_mil_vit_forward(const amds_mil_vit_cfg* cfg_host, const amds_mil_vit_weights* w_host, const void* bags, int bags_dtype,
                                    const float* coords, const uint8_t* mask, float* logits, int n_bags, int n_tiles, void* ws, size_t ws_bytes,
                                    void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && bags && logits && ws, "amds_mil_vit_forward: null pointer");
    const amds_mil_vit_cfg& c = *cfg_host;
    const amds_mil_vit_weights& w = *w_host;
    MilPlan p;
    int rc = mil_plan(cfg_host, n_bags, n_tiles, &p);
    if (rc != AMDS_OK) return rc;
    AMDS_REQUIRE(w.class_token && w.proj_w && w.proj_b && w.norm_w && w.norm_b && w.head_w && (c.layers == 0 || w.layers_host),
                 "amds_mil_vit_forward: incomplete weights");
    AMDS_REQUIRE(!c.alibi || coords, "amds_mil_vit_forward: use_alibi=True needs coords");
    AMDS_REQUIRE(bags_dtype == AMDS_F32 || bags_dtype == AMDS_F16 || bags_dtype == AMDS_BF16, "amds_mil_vit_forward: bad bags dtype %d", bags_dtype);
    if (ws_bytes < p.total) {
        set_error("amds_mil_vit_forward: workspace %zu < required %zu bytes", ws_bytes, p.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_mil_vit_forward: workspace must be 256-byte aligned");
    if (n_bags == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    char* base = reinterpret_cast<char*>(ws);
    const int Bb = n_bags, Tn = n_tiles, S = Tn + 1, D = c.dim, Dp = p.Dp, dt = c.dtype;
    const long M = (long)Bb * S, Mt = (long)Bb * Tn;
    AMDS_REQUIRE(M * (long)p.FFp < (1L << 40) && M < (1L << 31), "amds_mil_vit_forward: %ld token rows do not fit the 32-bit row index", M);
    float* x = reinterpret_cast<float*>(base + p.x);
    void* h = base + p.h;
    void* qkv = base + p.qkv;
    void* att = base + p.att;
    void* u = base + p.u;
    float* cls = reinterpret_cast<float*>(base + p.cls);
    float* cw = c.alibi ? reinterpret_cast<float*>(base + p.coords) : nullptr;
    uint8_t* pad = mask ? reinterpret_cast<uint8_t*>(base + p.pad) : nullptr;

    // project_features: the bags as 16-bit operand rows (already in that form when dtype and pitch agree)
    const void* a = bags;
    if (!(bags_dtype == dt && c.n_feats == p.Fp)) {
        const long total = Mt * p.Fp;
        const int grid = (int)min((long)8192, (total + 255) / 256);
#define STAGE(TI, TO) hipLaunchKernelGGL((stage_bags_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)bags, (long)c.n_feats, (TO*)(base + p.a), \
                                         p.Fp, total, c.n_feats)
        if (dt == AMDS_F16) {
            if (bags_dtype == AMDS_F32) STAGE(float, f16);
            else if (bags_dtype == AMDS_F16) STAGE(f16, f16);
            else STAGE(bf16, f16);
        } else {
            if (bags_dtype == AMDS_F32) STAGE(float, bf16);
            else if (bags_dtype == AMDS_F16) STAGE(f16, bf16);
            else STAGE(bf16, bf16);
        }
#undef STAGE
        AMDS_LAUNCH_CHECK("stage_bags_kernel");
        a = base + p.a;
    }
    // Linear + GELU in fp32 into scratch (the qkv region: 6 Da >= 4 Dp bytes per row always), then the rows move behind the class tokens
    float* proj = reinterpret_cast<float*>(qkv);
    if ((rc = amds_gemm(a, p.Fp, w.proj_w, p.Fp, (int)Mt, Dp, p.Fp, dt, AMDS_EPI_BIAS_GELU_F32, proj, Dp, w.proj_b, nullptr, nullptr, 0, 0, 0, 1.0f,
                        stream)) != AMDS_OK) return rc;
    hipLaunchKernelGGL(prefix_cls_kernel, dim3((unsigned)M), dim3(128), 0, st, w.class_token, proj, x, Dp, coords, cw, mask, pad, Tn);
    AMDS_LAUNCH_CHECK("prefix_cls_kernel");
    if (Dp != D) AMDS_HIP(hipMemsetAsync(h, 0, (size_t)M * Dp * 2, st));      // LayerNorm writes the first D columns only

    // amds_set_mil_cls_tail(0) / AMDS_MIL_CLS_TAIL=0: every row of the last block (A/B, tests).  Not with ALiBi (its attention has no one-query form).  A padding
    // mask changes nothing for the class query: the reference's mask blocks (padded query, padded key) pairs and the class token as a KEY of the tile queries
    // (vision_tranformer.py:356-368) -- the class token is never padded, so its own row attends to every key, exactly the unmasked one-query attention.
    const bool cls_tail = ctx_mil_cls_tail() && !c.alibi && S <= 32768 && (long)(Bb - 1) * S * Dp * 4 < (1L << 31);
    for (int l = 0; l < c.layers && rc == AMDS_OK; ++l) {
        const amds_mil_vit_layer& L = w.layers_host[l];
        AMDS_REQUIRE(L.ln1_w && L.ln1_b && L.in_w && L.in_b && L.out_w && L.out_b && L.ln2_w && L.ln2_b && L.fc1_w && L.fc1_b && L.fc2_w && L.fc2_b &&
                     (!c.alibi || L.head_scale), "amds_mil_vit_forward: incomplete weights of layer %d", l);
        if ((rc = amds_layernorm(x, Dp, L.ln1_w, L.ln1_b, h, Dp, (int)M, D, 1e-5f, dt, stream)) != AMDS_OK) break;
        if (cls_tail && l == c.layers - 1) {
            // Class-row tail: the head reads x[:, 0] behind this block and nothing else (reference vision_tranformer.py: `self.mlp_head(x[:, 0])`), so the block
            // computes keys | values of every token, and query, attention, output projection and MLP of the class rows alone (row b * S of the token-major
            // tensors: a row pitch of S * Dp addresses them in place).  The other rows of x keep the previous block's values; nothing reads them.
            const int esz = 2;
            const char* w_kv = reinterpret_cast<const char*>(L.in_w) + (size_t)p.Da * Dp * esz;
            char* qkv_kv = reinterpret_cast<char*>(qkv) + (size_t)p.Da * esz;
            if ((rc = amds_gemm(h, Dp, w_kv, Dp, (int)M, 2 * p.Da, Dp, dt, AMDS_EPI_BIAS, qkv_kv, 3 * p.Da, L.in_b + p.Da, nullptr, nullptr, 0, 0, 0, 1.0f,
                                stream)) != AMDS_OK) break;
            char* qc = reinterpret_cast<char*>(att);                          // [Bb][Da] queries | [Bb][Da] attention outputs (the att buffer is idle)
            char* oc = qc + (size_t)Bb * p.Da * esz;
            if ((rc = amds_gemm(h, (long)S * Dp, L.in_w, Dp, Bb, p.Da, Dp, dt, AMDS_EPI_BIAS, qc, p.Da, L.in_b, nullptr, nullptr, 0, 0, 0, 1.0f,
                                stream)) != AMDS_OK) break;
            if ((rc = amds_attention_row(qc, p.Da, qkv, oc, p.Da, Bb, S, p.Ha, dt, stream)) != AMDS_OK) break;
            if ((rc = amds_gemm(oc, p.Da, L.out_w, p.Da, Bb, Dp, p.Da, dt, AMDS_EPI_RESIDUAL, x, (long)S * Dp, L.out_b, nullptr, nullptr, 0, 0, 0, 1.0f,
                                stream)) != AMDS_OK) break;
            if ((rc = amds_layernorm(x, (long)S * Dp, L.ln2_w, L.ln2_b, h, Dp, Bb, D, 1e-5f, dt, stream)) != AMDS_OK) break;      // (h's first Bb rows: its LN1 rows are consumed)
            if ((rc = amds_gemm(h, Dp, L.fc1_w, Dp, Bb, p.FFp, Dp, dt, AMDS_EPI_BIAS_GELU, u, p.FFp, L.fc1_b, nullptr, nullptr, 0, 0, 0, 1.0f,
                                stream)) != AMDS_OK) break;
            rc = amds_gemm(u, p.FFp, L.fc2_w, p.FFp, Bb, Dp, p.FFp, dt, AMDS_EPI_RESIDUAL, x, (long)S * Dp, L.fc2_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream);
            continue;
        }
        if ((rc = amds_gemm(h, Dp, L.in_w, Dp, (int)M, 3 * p.Da, Dp, dt, AMDS_EPI_BIAS, qkv, 3 * p.Da, L.in_b, nullptr, nullptr, 0, 0, 0, 1.0f,
                            stream)) != AMDS_OK) break;
        if (c.alibi)        // output bf16 (range of the distance term), so the output projection runs on bf16 operands
            rc = pad ? amds_attention_alibi_masked(qkv, cw, L.head_scale, pad, att, Bb, S, p.Ha, dt, stream)
                     : amds_attention_alibi(qkv, cw, L.head_scale, att, Bb, S, p.Ha, dt, stream);
        else
            rc = pad ? amds_attention_masked(qkv, pad, att, Bb, S, p.Ha, c.heads, dt, stream) : amds_attention(qkv, att, Bb, S, p.Ha, dt, stream);
        if (rc != AMDS_OK) break;
        if ((rc = amds_gemm(att, p.Da, L.out_w, p.Da, (int)M, Dp, p.Da, c.alibi ? AMDS_BF16 : dt, AMDS_EPI_RESIDUAL, x, Dp, L.out_b, nullptr, nullptr, 0,
                            0, 0, 1.0f, stream)) != AMDS_OK) break;                                                       // x = attn(x) + x   (:291-292)
        if ((rc = amds_layernorm(x, Dp, L.ln2_w, L.ln2_b, h, Dp, (int)M, D, 1e-5f, dt, stream)) != AMDS_OK) break;
        if ((rc = amds_gemm(h, Dp, L.fc1_w, Dp, (int)M, p.FFp, Dp, dt, AMDS_EPI_BIAS_GELU, u, p.FFp, L.fc1_b, nullptr, nullptr, 0, 0, 0, 1.0f,
                            stream)) != AMDS_OK) break;
        rc = amds_gemm(u, p.FFp, L.fc2_w, p.FFp, (int)M, Dp, p.FFp, dt, AMDS_EPI_RESIDUAL, x, Dp, L.fc2_b, nullptr, nullptr, 0, 0, 0, 1.0f, stream);   // x = ff(x) + x (:293)
    }
    if (rc != AMDS_OK) return rc;
    // final LayerNorm on the class-token rows only (row stride = one bag), then the head in exact fp32
    if ((rc = amds_layernorm(x, (long)S * Dp, w.norm_w, w.norm_b, cls, D, Bb, D, 1e-5f, AMDS_F32, stream)) != AMDS_OK) return rc;
    return amds_linear_f32(cls, w.head_w, w.head_b, logits, Bb, c.classes, D, 0, stream);
}
