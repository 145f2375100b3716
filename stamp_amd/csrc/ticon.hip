// ticon.hip -- the TICON stage of the reference's H-optimus + TICON extractor as ONE call on a batch of tile embeddings.
//
// src/stamp/preprocessing/extractor/ticon.py: `HOptimusTICON.forward` :691-718 pushes every tile's embedding ALONE (one token, coordinates
// (0, 0)) through `EncoderDecoder.forward` :543-562.  With a single key the ALiBi attention (:183-215) returns its value whatever the
// bias, so a block is  x += g1 * proj(v_proj(LN1(x)));  x += g2 * fc2(silu(x1) * x2), (x1 | x2) = fc1(LN2(x))  (:290-343, :54-77) -- row-wise
// fp32 work on [n_tiles][dim]: exact-fp32 MFMA products (amds_linear_f32 / amds_bgemm_f32), LayerNorm and activation kernels.
#include <algorithm>
#include "common.h"

namespace amds {
namespace {
inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

struct TcPlan { size_t e, x, t, u, total; };

int tc_plan(const amds_ticon_weights* w, int B, TcPlan* p) {
    AMDS_REQUIRE(w, "amds_ticon: null weights");
    AMDS_REQUIRE(w->in_dim > 0 && w->dim > 0 && w->dim % 4 == 0 && w->hidden > 0 && w->hidden % 2 == 0 && w->depth >= 0 && B >= 0, "amds_ticon: bad configuration");
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    p->e = take((size_t)B * w->in_dim * 4);
    p->x = take((size_t)B * w->dim * 4);
    p->t = take((size_t)B * w->dim * 4);
    p->u = take((size_t)B * std::max(w->hidden, w->dim) * 4);
    p->total = off;
    return AMDS_OK;
}

__global__ void __launch_bounds__(256) tc_f16_to_f32_kernel(const f16* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}

#define RC(call)                          \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != AMDS_OK) return rc__; \
    } while (0)
}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_ticon_tile_workspace_bytes(const amds_ticon_weights* w_host, int n_tiles) {
    TcPlan p;
    if (tc_plan(w_host, n_tiles, &p) != AMDS_OK) return 0;
    return p.total;
}

extern "C" int amds_ticon_tile_forward(const amds_ticon_weights* w_host, const void* emb, int emb_dtype, void* out, int out_dtype, int n_tiles, void* ws,
                                       size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(w_host, "amds_ticon_tile_forward: null weights");
    if (n_tiles == 0) return AMDS_OK;
    AMDS_REQUIRE(emb && out && ws, "amds_ticon_tile_forward: null pointer");
    const amds_ticon_weights& w = *w_host;
    TcPlan p;
    RC(tc_plan(w_host, n_tiles, &p));
    AMDS_REQUIRE(w.in_fc1_w && w.in_fc1_b && w.in_fc2_w && w.in_fc2_b && w.in_norm_w && w.in_norm_b && w.norm_w && w.norm_b && (w.depth == 0 || w.blocks_host),
                 "amds_ticon_tile_forward: incomplete weights");
    AMDS_REQUIRE((emb_dtype == AMDS_F32 || emb_dtype == AMDS_F16) && (out_dtype == AMDS_F32 || out_dtype == AMDS_F16), "amds_ticon_tile_forward: bad dtype");
    if (ws_bytes < p.total) {
        set_error("amds_ticon_tile_forward: workspace %zu < required %zu bytes", ws_bytes, p.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_ticon_tile_forward: workspace must be 256-byte aligned");
    if (n_tiles == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    char* base = reinterpret_cast<char*>(ws);
    const int B = n_tiles, D = w.dim, Hh = w.hidden, H2 = w.hidden / 2;
    const float* e = reinterpret_cast<const float*>(emb);
    if (emb_dtype == AMDS_F16) {
        const long n = (long)B * w.in_dim;
        float* ef = reinterpret_cast<float*>(base + p.e);
        hipLaunchKernelGGL(tc_f16_to_f32_kernel, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, st, (const f16*)emb, ef, n);
        AMDS_LAUNCH_CHECK("tc_f16_to_f32_kernel");
        e = ef;
    }
    float *x = reinterpret_cast<float*>(base + p.x), *t = reinterpret_cast<float*>(base + p.t), *u = reinterpret_cast<float*>(base + p.u);
    // input projection: Linear, SiLU, Linear, LayerNorm (:94-98)
    RC(amds_linear_f32(e, w.in_fc1_w, w.in_fc1_b, u, B, D, w.in_dim, 0, stream));
    RC(amds_mlp_act_f32(u, D, B, D, 2, stream));
    RC(amds_linear_f32(u, w.in_fc2_w, w.in_fc2_b, t, B, D, D, 0, stream));
    RC(amds_layernorm(t, D, w.in_norm_w, w.in_norm_b, x, D, B, D, 1e-5f, AMDS_F32, stream));
    for (int l = 0; l < w.depth; ++l) {
        const amds_ticon_block& b = w.blocks_host[l];
        AMDS_REQUIRE(b.ln1_w && b.ln1_b && b.v_w && b.v_b && b.proj_w && b.proj_b && b.ln2_w && b.ln2_b && b.fc1_w && b.fc1_b && b.fc2_w && b.fc2_b,
                     "amds_ticon_tile_forward: incomplete weights of block %d", l);
        RC(amds_layernorm(x, D, b.ln1_w, b.ln1_b, t, D, B, D, 1e-5f, AMDS_F32, stream));
        RC(amds_linear_f32(t, b.v_w, b.v_b, u, B, D, D, 0, stream));                                                       // one key: attention = value
        RC(bgemm_f32_exact(u, D, 0, 0, b.proj_w, D, 0, 0, 1, x, D, 0, 0, 1, 1, B, D, D, 1.0f, 0.0f, b.proj_b, 1, stream));    // x += g1 * proj(v)
        RC(amds_layernorm(x, D, b.ln2_w, b.ln2_b, t, D, B, D, 1e-5f, AMDS_F32, stream));
        RC(amds_linear_f32(t, b.fc1_w, b.fc1_b, u, B, Hh, D, 0, stream));
        RC(amds_mlp_act_f32(u, Hh, B, H2, 1, stream));                                                                     // u[:, :H2] = silu(x1) * x2
        RC(bgemm_f32_exact(u, Hh, 0, 0, b.fc2_w, H2, 0, 0, 1, x, D, 0, 0, 1, 1, B, D, H2, 1.0f, 0.0f, b.fc2_b, 1, stream));   // x += g2 * fc2(.)
    }
    return amds_layernorm(x, D, w.norm_w, w.norm_b, out, D, B, D, 1e-5f, out_dtype, stream);
}
