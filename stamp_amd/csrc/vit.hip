// vit.hip -- tile-encoder forward: u8 tiles -> fp16 CLS features, as a fixed chain of launches on one
// stream.  Arithmetic plan per block (timm VisionTransformer Block, pre-LN, optional LayerScale):
//   h   = LN1(x)                      fp32 -> act dtype        (layernorm_kernel)
//   qkv = h Wqkv^T + b                act -> act               (MFMA GEMM, EPI_BIAS)
//   a   = softmax(q k^T / 8) v        act -> act               (attn_vit_kernel)
//   x  += ls1 * (a Wproj^T + b)       fp32 residual stream     (MFMA GEMM, EPI_RESIDUAL)
//   h   = LN2(x)
//   u   = gelu(h W1^T + b1) | silu(g)*v                         (MFMA GEMM, EPI_BIAS_GELU / EPI_SWIGLU)
//   x  += ls2 * (u W2^T + b2)                                   (MFMA GEMM, EPI_RESIDUAL)
// With LayerNorm folded into the GEMMs (blocks carry qkv_colsum / fc1_colsum; include/amdstamp.h, amds_gemm_lnfold) the two
// layernorm launches of a block disappear: the proj / fc2 epilogues also write a 16-bit copy of the rows they update plus partial
// row sums, amds_ln_rowstat turns those into (rstd, -mean*rstd) per row, and the qkv / fc1 GEMMs read the copy with W * gamma as
// weights and apply the row statistics in their epilogues.  Only the first LayerNorm of the stack needs its own (statistics + cast)
// launch, and the final norm stays a LayerNorm kernel.
// The residual stream x stays fp32 in HBM for the whole depth (the reference computes in fp32,
// src/stamp/preprocessing/__init__.py:324-325); only MFMA operands are rounded to the act dtype.
#include "common.h"

namespace amds {
int prefix_init(const float* prefix, float* x, int B, int T, int P, int dim, hipStream_t st);

struct VitPlan {
    int np, T, kp, dim, hidden;
    size_t off_x, off_h, off_qkv, off_mlp, off_h2, off_lo, off_rowpart, off_rowstat, off_xc, off_hc, off_qc, off_oc, off_uc, off_diag, off_q8, off_q8s, off_q8n, off_q8u, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int make_plan(const amds_vit_cfg* c, int batch, VitPlan* p) {
    AMDS_REQUIRE(c != nullptr, "vit: null cfg");
    AMDS_REQUIRE(c->img > 0 && c->patch > 0 && c->img % c->patch == 0, "vit: img=%d not divisible by patch=%d", c->img, c->patch);
    AMDS_REQUIRE(c->dim % 128 == 0 && c->heads > 0 && (c->heads * 64 == c->dim || c->heads * 80 == c->dim),
                 "vit: dim=%d must be a multiple of 128 and heads*64 or heads*80", c->dim);
    AMDS_REQUIRE(c->hidden % 64 == 0 && c->hidden > 0, "vit: hidden=%d must be a multiple of 64 (zero-pad)", c->hidden);
    AMDS_REQUIRE(c->mlp_kind >= 0 && c->mlp_kind <= 2, "vit: bad mlp_kind");
    AMDS_REQUIRE(c->mlp_kind == 1 || c->hidden % 128 == 0, "vit: GELU hidden=%d must be a multiple of 128", c->hidden);
    AMDS_REQUIRE(c->dtype == AMDS_F16 || c->dtype == AMDS_BF16, "vit: bad act dtype");
    AMDS_REQUIRE(c->depth > 0 && c->n_prefix >= 0 && batch > 0, "vit: bad depth/prefix/batch");
    const int g = c->img / c->patch;
    p->np = g * g;
    p->T = p->np + c->n_prefix;
    AMDS_REQUIRE(p->T <= 288, "vit: %d tokens > 288 unsupported", p->T);
    p->kp = ((3 * c->patch * c->patch + 63) / 64) * 64;
    p->dim = c->dim;
    p->hidden = c->hidden;
    const size_t rows = (size_t)batch * p->T;
    size_t o = 0;
    p->off_x = o;   o += align256(rows * c->dim * 4);
    p->off_h = o;   o += align256(rows * c->dim * 2);
    p->off_qkv = o; o += align256(rows * 3 * c->dim * 2);
    const size_t mlp_b = rows * c->hidden * 2, pm_b = (size_t)batch * p->np * p->kp * 2 * 2;    // patch matrix: room for the split form
    p->off_mlp = o; o += align256(mlp_b > pm_b ? mlp_b : pm_b);
    // LayerNorm-folded path: second 16-bit row buffer, partial row sums per 128-column slab, final row statistics
    p->off_h2 = o;      o += align256(rows * c->dim * 2);
    p->off_lo = o;      o += align256(rows * c->dim * 2);          // lo plane of the (hi | lo) residual stream (hi = off_h)
    p->off_rowpart = o; o += align256(rows * (size_t)(c->dim / 128) * 2 * 4);
    p->off_rowstat = o; o += align256(rows * 2 * 4);
    // exact class-token path (vit_exact.hip): fp32 class stream, its LayerNorm output, query rows, attention output, MLP hidden rows
    const size_t cls = (size_t)batch * c->dim * 4;
    p->off_xc = o; o += align256(cls);
    p->off_hc = o; o += align256(cls);
    p->off_qc = o; o += align256(cls);
    p->off_oc = o; o += align256(cls);
    p->off_uc = o; o += align256((size_t)batch * 2 * c->hidden * 4);
    p->off_diag = o; o += 256;          // int32[2] range diagnostics of the folded LayerNorms (amds_ln_rowstat_diag)
    // opt-in fp8 GEMMs: the quantised A operand (e4m3 bytes, widest = the MLP hidden) and its per-row scales
    p->off_q8 = o;  o += align256(rows * (size_t)(c->hidden > c->dim ? c->hidden : c->dim));
    p->off_q8s = o; o += align256(rows * 4);
    p->off_q8n = o; o += align256(rows * 4);          // L2 norms of the normalised rows entering fc1
    p->off_q8u = o; o += align256(rows * 4);          // the bounding row scales of fc1's e4m3 output
    p->total = o;
    return AMDS_OK;
}

// Side stream + fork / join events for the ragged-tail schedule below (nullptr side = single stream).
struct VitSide {
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static int vit_chunk(const amds_vit_cfg* c, const amds_vit_weights* w, const VitPlan& pl, const uint8_t* tiles,
                     void* feats_f16, float* tokens_f32, int Bc, char* ws, hipStream_t st, const VitSide& sd = VitSide()) {
    float* x = reinterpret_cast<float*>(ws + pl.off_x);
    char* h = ws + pl.off_h;
    char* qkv = ws + pl.off_qkv;
    char* mlp = ws + pl.off_mlp;
    char* h2 = ws + pl.off_h2;
    char* lo = ws + pl.off_lo;
    float* rowpart = reinterpret_cast<float*>(ws + pl.off_rowpart);
    float* rowstat = reinterpret_cast<float*>(ws + pl.off_rowstat);
    int* diag = reinterpret_cast<int*>(ws + pl.off_diag);
    const int D = c->dim, T = pl.T, M = Bc * T, dt = c->dtype, Hd = c->hidden, NP = D / 128;
    const int n_fc1 = c->mlp_kind == 1 ? 2 * Hd : Hd;
    int rc;
#define AMDS_TRY(call) do { rc = (call); if (rc != AMDS_OK) return rc; } while (0)
    // One kernel family per GEMM whatever the batch: a tile's features must not depend on which chunk it travels in (bit-exact,
    // tests/test_gpu_vit.py).  The library default switches from 128x128 tiles (v_mfma 32x32x16) to the 16x16x32 kernel once the
    // grid fills the chip, and those two differ in the last bits -- so the encoder names the kernel id itself.  AMDS_GEMM_CFG
    // (tuning) still overrides.
    static const int kcfg = (getenv("AMDS_GEMM_CFG") && *getenv("AMDS_GEMM_CFG")) ? -1 : 12;
    auto enc_gemm = [&](const void* A, long lda, const void* Wt, long ldw, int Mr, int N, int K, int epi, void* out, long ldo, const float* bias,
                        const float* scale, hipStream_t s) {
        return amds_gemm_ex(N % 256 == 0 ? kcfg : -1, A, lda, Wt, ldw, Mr, N, K, dt, epi, out, ldo, bias, scale, nullptr, 0, 0, 0, 1.0f, s);
    };
    // LayerNorm folded into the GEMMs: all blocks or none (a mixed stack would be a packing error)
    const bool fold = w->blocks_host[0].qkv_colsum != nullptr;
    for (int l = 0; l < c->depth; ++l) {
        const amds_vit_block& b = w->blocks_host[l];
        AMDS_REQUIRE((b.qkv_colsum != nullptr) == fold && (b.fc1_colsum != nullptr) == fold, "vit: block %d: qkv_colsum / fc1_colsum must be set in all blocks or in none", l);
    }
    if (fold)
        AMDS_REQUIRE(D % 256 == 0 && n_fc1 % 256 == 0, "vit: the LayerNorm-folded path needs dim %% 256 == 0 and fc1 rows %% 256 == 0 (dim=%d, fc1 rows=%d)", D, n_fc1);
    AMDS_REQUIRE(c->mlp_kind != 2 || (!fold && !w->exact_host && !w->fp8_host), "vit: the quick-GELU MLP (CLIP) runs on the plain packing only (no LayerNorm fold, exact rows or fp8)");
    // exact class-token rows (vit_exact.hip): an fp32 class stream beside the 16-bit-operand path
    const amds_vit_exact_block* ex = w->exact_host;
    const int xh_ = w->exact_hidden, xf1 = c->mlp_kind == 1 ? 2 * xh_ : xh_, hd = D / c->heads;
    if (ex) {
        AMDS_REQUIRE(xh_ > 0 && xh_ <= Hd && xh_ % 4 == 0, "vit: exact_hidden=%d must be a multiple of 4 and <= hidden=%d", xh_, Hd);
        for (int l = 0; l < c->depth; ++l)
            AMDS_REQUIRE(ex[l].q_w && ex[l].q_b && ex[l].proj_w && ex[l].proj_b && ex[l].fc1_w && ex[l].fc1_b && ex[l].fc2_w && ex[l].fc2_b &&
                         w->blocks_host[l].ln1_w && w->blocks_host[l].ln1_b && w->blocks_host[l].ln2_w && w->blocks_host[l].ln2_b,
                         "vit: exact block %d: incomplete weights", l);
    }
    // Class-row tail (amds_vit_weights.cls_tail): with only the class features requested, the last block computes keys / values for all
    // tokens and then the class row's own chain in fp32 -- nothing else of that block is ever read.  AMDS_VIT_CLS_TAIL=0 runs the full block (A/B).
    const bool tail_env = !(getenv("AMDS_VIT_CLS_TAIL") && atoi(getenv("AMDS_VIT_CLS_TAIL")) == 0);
    const amds_vit_exact_block* tl = (tail_env && tokens_f32 == nullptr && c->mlp_kind != 2) ? (ex ? &ex[c->depth - 1] : w->cls_tail) : nullptr;
    if (tl) {
        AMDS_REQUIRE(xh_ > 0 && xh_ <= Hd && xh_ % 4 == 0, "vit: exact_hidden=%d must be a multiple of 4 and <= hidden=%d", xh_, Hd);
        const amds_vit_block& lb = w->blocks_host[c->depth - 1];
        AMDS_REQUIRE(tl->q_w && tl->q_b && tl->proj_w && tl->proj_b && tl->fc1_w && tl->fc1_b && tl->fc2_w && tl->fc2_b && lb.ln1_w && lb.ln1_b && lb.ln2_w && lb.ln2_b,
                     "vit: cls_tail: incomplete weights");
    }
    // Residual stream as two fp16 planes (folded path, fp16 operands, no exact class stream): x = hi + lo with hi = fp16(x) -- the very rows the
    // qkv / fc1 GEMMs read as their A operand -- and lo = fp16(x - hi).  The proj / fc2 epilogues read 4 + write 4 bytes per element instead of
    // reading 4 and writing 4 + 2 (fp32 row + its 16-bit copy): 0.54 GB less per launch at M = 262 140, D = 1024.  x keeps ~22 mantissa bits
    // (fp32: 24; the GEMM operands: 11).  AMDS_VIT_PLANES=0: fp32 rows + copy (A/B; read at every call).
    const bool planes_env = !(getenv("AMDS_VIT_PLANES") && atoi(getenv("AMDS_VIT_PLANES")) == 0);
    const bool planes = planes_env && fold && !ex && dt == AMDS_F16;
    // (qkv + attention as ONE kernel -- q | k | v never in HBM -- was built in round 5, is parity-green and 0.8 % slower in situ: profiles/r05_qkv_attn_fused_ab.txt;
    //  the kernel, its check and its A/B scripts live in tools/ubench/attic/qkv_attn257/ since round 6.)
    // opt-in fp8 GEMMs (gemm_fp8.hip): plain (un-folded) weights, GELU MLP
    const amds_vit_fp8_block* f8 = w->fp8_host;
    if (f8) {
        AMDS_REQUIRE(!fold && !ex && dt == AMDS_F16, "vit: the fp8 path needs the plain packing (no LayerNorm fold, no exact rows) and fp16 activations");
        AMDS_REQUIRE(D % 256 == 0 && Hd % 256 == 0, "vit: the fp8 path needs dim %% 256 == 0 and hidden %% 256 == 0");
        for (int l = 0; l < c->depth; ++l)
            AMDS_REQUIRE(f8[l].qkv_w8 && f8[l].qkv_cs && f8[l].proj_w8 && f8[l].proj_cs && f8[l].proj_b && f8[l].fc1_w8 && f8[l].fc1_cs && f8[l].fc2_w8 && f8[l].fc2_cs && f8[l].fc2_b,
                         "vit: fp8 block %d: incomplete weights", l);
    }
    char* q8 = ws + pl.off_q8;
    float* q8s = reinterpret_cast<float*>(ws + pl.off_q8s);
    float* q8n = reinterpret_cast<float*>(ws + pl.off_q8n);
    float* q8u = reinterpret_cast<float*>(ws + pl.off_q8u);
    float* xc = reinterpret_cast<float*>(ws + pl.off_xc);
    float* hc = reinterpret_cast<float*>(ws + pl.off_hc);
    float* qc = reinterpret_cast<float*>(ws + pl.off_qc);
    float* oc = reinterpret_cast<float*>(ws + pl.off_oc);
    float* uc = reinterpret_cast<float*>(ws + pl.off_uc);

    // patch embedding: im2col (raw 0..255 values) -> GEMM with folded normalisation, + pos-embed
    // (patch_lo_shift > 0: weight and patch matrix in the split [hi | lo] form, K doubled -- include/amdstamp.h)
    const int kpe = w->patch_lo_shift > 0 ? 2 * pl.kp : pl.kp;
    AMDS_TRY(amds_tile_im2col_u8_ex(tiles, mlp, Bc, c->img, c->patch, pl.kp, dt, w->patch_lo_shift, st));
    if (c->n_prefix > 0) AMDS_TRY(prefix_init(w->prefix, x, Bc, T, c->n_prefix, D, st));
    AMDS_TRY(amds_gemm(mlp, kpe, w->patch_w, kpe, Bc * pl.np, D, kpe, dt, AMDS_EPI_PATCH, x, D, w->patch_b,
                       nullptr, w->pos_patch, pl.np, T, c->n_prefix, 1.0f / 255.0f, st));
    // CLIP-style trunks normalise the embeddings before the first block (HF `pre_layrnorm`, timm `norm_pre`): in place, each wave holds its row
    if (w->pre_norm_w) {
        AMDS_REQUIRE(w->pre_norm_b, "vit: pre_norm_b missing");
        AMDS_TRY(amds_layernorm(x, D, w->pre_norm_w, w->pre_norm_b, x, D, Bc * T, D, c->ln_eps, AMDS_F32, st));
    }

    // ---- ragged tail on a side stream.  The GEMMs work on 256-row tiles, one workgroup per CU, and every launch is a whole number of
    // waves of workgroups: M = 64 x 257 rows (the reference's DataLoader batch) is 64.25 row tiles, and the 65th row tile costs every GEMM
    // a whole extra wave on a mostly idle chip (proj / fc2: 260 workgroups on 256 CUs = two waves for the work of one; 19 tile-time units
    // per block instead of 12).  Tiles are independent through the WHOLE network (attention is per tile, everything else per row), so the
    // last tile(s) of the batch run as their own chain of launches on the side stream, from the first LayerNorm to the last fc2, and
    // meet the main tiles once, before the final norm: their few workgroups fill CUs the main chain leaves idle at its wave boundaries.
    // Same kernels, same arithmetic, same bits -- only the launch geometry changes.  A tail is split off when removing 1 or 2 tiles lowers the main chain's waves per block
    // (weights: qkv 1, proj 1, fc1 1, fc2 K/D).
    auto block_units = [&](int tiles) {
        const long rt = ((long)tiles * T + 255) / 256;                      // row tiles of the main chain
        auto waves = [&](long tn) { return (rt * tn + 255) / 256; };
        const long kf = Hd / D > 0 ? Hd / D : 1;
        return waves(3 * D / 256) + waves(D / 256) + waves(n_fc1 / 256) + kf * waves(D / 256);
    };
    int n_tail = 0;
    if (sd.side != nullptr && D % 256 == 0 && n_fc1 % 256 == 0 && Bc >= 8) {
        long best = block_units(Bc);
        for (int t = 1; t <= 2; ++t) {
            const long u = block_units(Bc - t);
            if (u < best) { best = u; n_tail = t; }
        }
        // Every main launch being a whole number of full waves, a tail workgroup always displaces a main one: a few more tail tiles leave
        // the main launches a few idle CUs per wave for the tail to land on (B = 64: 1 / 2 / 4 / 6 tail tiles -> 4 695 / 4 777 / 4 824 / 4 818 tiles/s)
        if (n_tail > 0) n_tail = Bc / 8 < 4 ? (Bc / 8 > n_tail ? Bc / 8 : n_tail) : 4;
    }
    struct Part { int t0, nt; hipStream_t s; };
    const Part parts[2] = {{0, Bc - n_tail, st}, {Bc - n_tail, n_tail, sd.side}};
    const int nparts = n_tail > 0 ? 2 : 1;
    auto rows16 = [&](char* base, long r0, long pitch_elems) { return base + (size_t)r0 * pitch_elems * 2; };   // 16-bit row buffers

    // the whole depth for tiles [t0, t0 + nt) on stream s
    auto run_tiles = [&](const Part& q) -> int {
        const long r0 = (long)q.t0 * T;
        const int n = q.nt * T;
        hipStream_t s = q.s;
        float* xq = x + (size_t)r0 * D;
        float* rp = rowpart + (size_t)r0 * NP * 2;
        float* rs = rowstat + 2 * (size_t)r0;
        char *hq = rows16(h, r0, D), *h2q = rows16(h2, r0, D), *loq = rows16(lo, r0, D), *qkvq = rows16(qkv, r0, 3 * D), *mlpq = rows16(mlp, r0, Hd);
        const int epi1 = c->mlp_kind == 0 ? AMDS_EPI_BIAS_GELU : (c->mlp_kind == 1 ? AMDS_EPI_SWIGLU : AMDS_EPI_BIAS);      // kind 2: + quick-GELU pass below
        float *xcq = xc + (size_t)q.t0 * D, *hcq = hc + (size_t)q.t0 * D, *qcq = qc + (size_t)q.t0 * D, *ocq = oc + (size_t)q.t0 * D;
        float* ucq = uc + (size_t)q.t0 * 2 * Hd;
        auto lin32 = [&](const float* A, int lda, const float* Wt, int K, float* Cm, int ldc, int N, const float* bias, int acc) {
            return bgemm_f32_exact(A, lda, 0, 0, Wt, K, 0, 0, 1, Cm, ldc, 0, 0, 1, 1, q.nt, N, K, 1.0f, 0.0f, bias, acc, s);
        };
        if (ex) AMDS_TRY(amds_vit_cls_gather(xq, xcq, q.nt, T, D, s));
        if (planes) AMDS_TRY(amds_ln_stats_split(xq, D, n, D, c->ln_eps, hq, loq, D, rs, s));
        else if (fold) AMDS_TRY(amds_ln_stats_cast(xq, D, n, D, c->ln_eps, hq, D, rs, dt, s));
        for (int l = 0; l < c->depth; ++l) {
            const amds_vit_block& b = w->blocks_host[l];
            const bool last = l + 1 == c->depth;
            const float* ls1 = c->layerscale ? b.ls1 : nullptr;
            const float* ls2 = c->layerscale ? b.ls2 : nullptr;
            if (last && tl) {
                // keys | values of all tokens: the k | v rows of the qkv weight (rows D .. 3D; bias, row sums and channel scales alike), written
                // into the k | v thirds of the packed tensor; the q third keeps the previous block's values and is not read
                const size_t wkv = (size_t)D * D;
                if (f8) {
                    char* a8 = q8 + (size_t)r0 * (Hd > D ? Hd : D);
                    float* as = q8s + r0;
                    AMDS_TRY(amds_layernorm_quant_e4m3(xq, D, b.ln1_w, b.ln1_b, c->ln_eps, a8, D, as, nullptr, n, D, s));
                    AMDS_TRY(amds_gemm_fp8(a8, D, (const char*)f8[l].qkv_w8 + wkv, D, n, 2 * D, D, AMDS_EPI_BIAS, qkvq + (size_t)D * 2, 3 * D, b.qkv_b + D,
                                           f8[l].qkv_cs + D, as, s));
                } else if (fold) {
                    AMDS_TRY(amds_gemm_lnfold(hq, D, (const char*)b.qkv_w + wkv * 2, D, n, 2 * D, D, dt, AMDS_EPI_BIAS, qkvq + (size_t)D * 2, 3 * D, b.qkv_b + D,
                                              nullptr, nullptr, nullptr, rs, b.qkv_colsum + D, s));
                } else {
                    AMDS_TRY(amds_layernorm(xq, D, b.ln1_w, b.ln1_b, hq, D, n, D, c->ln_eps, dt, s));
                    // the kernel family the full qkv product (N = 3 D) takes, so that k | v carry the same bits as in a full block
                    AMDS_TRY(amds_gemm_ex((3 * D) % 256 == 0 ? kcfg : -1, hq, D, (const char*)b.qkv_w + wkv * 2, D, n, 2 * D, D, dt, AMDS_EPI_BIAS, qkvq + (size_t)D * 2, 3 * D,
                                          b.qkv_b + D, nullptr, nullptr, 0, 0, 0, 1.0f, s));
                }
                // the class row: xc += proj(attention(Wq LN1(xc); K, V)); xc += fc2(act(fc1(LN2(xc)))) -- fp32, LayerScale inside the rows
                if (planes) AMDS_TRY(amds_planes_to_f32(hq, loq, D, T, xcq, D, q.nt, D, s));
                else if (!ex) AMDS_TRY(amds_vit_cls_gather(xq, xcq, q.nt, T, D, s));
                AMDS_TRY(amds_layernorm(xcq, D, b.ln1_w, b.ln1_b, hcq, D, q.nt, D, c->ln_eps, AMDS_F32, s));
                AMDS_TRY(lin32(hcq, D, tl->q_w, D, qcq, D, D, tl->q_b, 0));
                AMDS_TRY(amds_attention_cls_f32(qcq, D, qkvq, ocq, D, q.nt, T, c->heads, hd, dt, s));
                AMDS_TRY(lin32(ocq, D, tl->proj_w, D, xcq, D, D, tl->proj_b, 1));
                AMDS_TRY(amds_layernorm(xcq, D, b.ln2_w, b.ln2_b, hcq, D, q.nt, D, c->ln_eps, AMDS_F32, s));
                AMDS_TRY(lin32(hcq, D, tl->fc1_w, D, ucq, xf1, xf1, tl->fc1_b, 0));
                AMDS_TRY(amds_mlp_act_f32(ucq, xf1, q.nt, xh_, c->mlp_kind, s));
                AMDS_TRY(lin32(ucq, xf1, tl->fc2_w, xh_, xcq, D, D, tl->fc2_b, 1));
                continue;
            }
            if (f8) {      // every Linear: f16 rows -> per-row e4m3 -> fp8 MFMA GEMM with (row scale x channel scale) in the epilogue
                char* a8 = q8 + (size_t)r0 * (Hd > D ? Hd : D);
                float* as = q8s + r0;
                float *an = q8n + r0, *au = q8u + r0;
                char* u8 = mlp + (size_t)r0 * Hd;          // the MLP buffer holds the e4m3 hidden rows (1 byte per element: the first half of it)
                AMDS_TRY(amds_layernorm_quant_e4m3(xq, D, b.ln1_w, b.ln1_b, c->ln_eps, a8, D, as, nullptr, n, D, s));
                AMDS_TRY(amds_gemm_fp8(a8, D, f8[l].qkv_w8, D, n, 3 * D, D, AMDS_EPI_BIAS, qkvq, 3 * D, b.qkv_b, f8[l].qkv_cs, as, s));
                AMDS_TRY(amds_attention_vit_hd(qkvq, hq, q.nt, T, c->heads, D / c->heads, dt, s));
                AMDS_TRY(amds_quantize_rows_e4m3(hq, D, a8, D, as, n, D, AMDS_F16, s));
                AMDS_TRY(amds_gemm_fp8(a8, D, f8[l].proj_w8, D, n, D, D, AMDS_EPI_RESIDUAL, xq, D, f8[l].proj_b, f8[l].proj_cs, as, s));
                if (c->mlp_kind == 0) {
                    AMDS_TRY(amds_layernorm_quant_e4m3(xq, D, b.ln2_w, b.ln2_b, c->ln_eps, a8, D, as, an, n, D, s));
                    AMDS_TRY(amds_row_bound_scale(an, f8[l].fc1_wnorm_max, f8[l].fc1_babs_max, au, n, s));
                    AMDS_TRY(amds_gemm_fp8_out8(a8, D, f8[l].fc1_w8, D, n, Hd, D, AMDS_EPI_BIAS_GELU, u8, Hd, au, b.fc1_b, f8[l].fc1_cs, as, s));
                } else {      // SwiGLUPacked: silu(g) * v has no useful a-priori bound -> f16 out, then the row quantiser (u8 = the front of the q8 buffer)
                    AMDS_TRY(amds_layernorm_quant_e4m3(xq, D, b.ln2_w, b.ln2_b, c->ln_eps, a8, D, as, nullptr, n, D, s));
                    AMDS_TRY(amds_gemm_fp8(a8, D, f8[l].fc1_w8, D, n, 2 * Hd, D, AMDS_EPI_SWIGLU, mlpq, Hd, b.fc1_b, f8[l].fc1_cs, as, s));
                    u8 = a8;
                    AMDS_TRY(amds_quantize_rows_e4m3(mlpq, Hd, u8, Hd, au, n, Hd, AMDS_F16, s));
                }
                AMDS_TRY(amds_gemm_fp8(u8, Hd, f8[l].fc2_w8, Hd, n, D, Hd, AMDS_EPI_RESIDUAL, xq, D, f8[l].fc2_b, f8[l].fc2_cs, au, s));
                continue;
            }
            if (planes) {      // hi plane (hq) = the A operand of qkv / fc1; attention writes h2q; proj / fc2 update (hq, loq) in place
                AMDS_TRY(amds_gemm_lnfold(hq, D, b.qkv_w, D, n, 3 * D, D, dt, AMDS_EPI_BIAS, qkvq, 3 * D, b.qkv_b, nullptr, nullptr, nullptr, rs,
                                          b.qkv_colsum, s));
                AMDS_TRY(amds_attention_vit_hd(qkvq, h2q, q.nt, T, c->heads, D / c->heads, dt, s));
                AMDS_TRY(amds_gemm_lnfold_planes(h2q, D, b.proj_w, D, n, D, D, hq, loq, D, b.proj_b, ls1, rp, s));
                AMDS_TRY(amds_ln_rowstat_diag(rp, n, NP, D, c->ln_eps, rs, diag, dt, s));
                AMDS_TRY(amds_gemm_lnfold(hq, D, b.fc1_w, D, n, n_fc1, D, dt, epi1, mlpq, Hd, b.fc1_b, nullptr, nullptr, nullptr, rs, b.fc1_colsum, s));
                AMDS_TRY(amds_gemm_lnfold_planes(mlpq, Hd, b.fc2_w, Hd, n, D, Hd, hq, loq, D, b.fc2_b, ls2, rp, s));
                if (!last) AMDS_TRY(amds_ln_rowstat_diag(rp, n, NP, D, c->ln_eps, rs, diag, dt, s));
                continue;
            }
            if (fold) {
                AMDS_TRY(amds_gemm_lnfold(hq, D, b.qkv_w, D, n, 3 * D, D, dt, AMDS_EPI_BIAS, qkvq, 3 * D, b.qkv_b, nullptr, nullptr, nullptr, rs,
                                          b.qkv_colsum, s));
            } else {
                AMDS_TRY(amds_layernorm(xq, D, b.ln1_w, b.ln1_b, hq, D, n, D, c->ln_eps, dt, s));
                AMDS_TRY(enc_gemm(hq, D, b.qkv_w, D, n, 3 * D, D, AMDS_EPI_BIAS, qkvq, 3 * D, b.qkv_b, nullptr, s));
            }
            AMDS_TRY(amds_attention_vit_hd(qkvq, hq, q.nt, T, c->heads, D / c->heads, dt, s));
            if (ex) {      // class stream, attention branch: xc += proj(attention(Wq LN1(xc); K, V of all tokens as stored))
                AMDS_TRY(amds_layernorm(xcq, D, b.ln1_w, b.ln1_b, hcq, D, q.nt, D, c->ln_eps, AMDS_F32, s));
                AMDS_TRY(lin32(hcq, D, ex[l].q_w, D, qcq, D, D, ex[l].q_b, 0));
                AMDS_TRY(amds_attention_cls_f32(qcq, D, qkvq, ocq, D, q.nt, T, c->heads, hd, dt, s));
                AMDS_TRY(lin32(ocq, D, ex[l].proj_w, D, xcq, D, D, ex[l].proj_b, 1));
            }
            if (fold) {
                AMDS_TRY(amds_gemm_lnfold(hq, D, b.proj_w, D, n, D, D, dt, AMDS_EPI_RESIDUAL, xq, D, b.proj_b, ls1, h2q, rp, nullptr, nullptr, s));
                AMDS_TRY(amds_ln_rowstat_diag(rp, n, NP, D, c->ln_eps, rs, diag, dt, s));
                if (ex) AMDS_TRY(amds_vit_cls_scatter(xcq, xq, h2q, rs, q.nt, T, D, c->ln_eps, dt, s));
                AMDS_TRY(amds_gemm_lnfold(h2q, D, b.fc1_w, D, n, n_fc1, D, dt, epi1, mlpq, Hd, b.fc1_b, nullptr, nullptr, nullptr, rs, b.fc1_colsum, s));
                if (!last) {
                    AMDS_TRY(amds_gemm_lnfold(mlpq, Hd, b.fc2_w, Hd, n, D, Hd, dt, AMDS_EPI_RESIDUAL, xq, D, b.fc2_b, ls2, hq, rp, nullptr, nullptr, s));
                    AMDS_TRY(amds_ln_rowstat_diag(rp, n, NP, D, c->ln_eps, rs, diag, dt, s));
                } else {
                    AMDS_TRY(enc_gemm(mlpq, Hd, b.fc2_w, Hd, n, D, Hd, AMDS_EPI_RESIDUAL, xq, D, b.fc2_b, ls2, s));
                }
            } else {
                AMDS_TRY(enc_gemm(hq, D, b.proj_w, D, n, D, D, AMDS_EPI_RESIDUAL, xq, D, b.proj_b, ls1, s));
                if (ex) AMDS_TRY(amds_vit_cls_scatter(xcq, xq, nullptr, nullptr, q.nt, T, D, c->ln_eps, dt, s));
                AMDS_TRY(amds_layernorm(xq, D, b.ln2_w, b.ln2_b, hq, D, n, D, c->ln_eps, dt, s));
                AMDS_TRY(enc_gemm(hq, D, b.fc1_w, D, n, n_fc1, D, epi1, mlpq, Hd, b.fc1_b, nullptr, s));
                if (c->mlp_kind == 2) AMDS_TRY(amds_quick_gelu_inplace(mlpq, Hd, n, Hd, dt, s));        // CLIP: x * sigmoid(1.702 x)
                AMDS_TRY(enc_gemm(mlpq, Hd, b.fc2_w, Hd, n, D, Hd, AMDS_EPI_RESIDUAL, xq, D, b.fc2_b, ls2, s));
            }
            if (ex) {      // class stream, MLP branch: xc += fc2(act(fc1(LN2(xc)))), then over the main path's class rows
                AMDS_TRY(amds_layernorm(xcq, D, b.ln2_w, b.ln2_b, hcq, D, q.nt, D, c->ln_eps, AMDS_F32, s));
                AMDS_TRY(lin32(hcq, D, ex[l].fc1_w, D, ucq, xf1, xf1, ex[l].fc1_b, 0));
                AMDS_TRY(amds_mlp_act_f32(ucq, xf1, q.nt, xh_, c->mlp_kind, s));
                AMDS_TRY(lin32(ucq, xf1, ex[l].fc2_w, xh_, xcq, D, D, ex[l].fc2_b, 1));
                const bool copy16 = fold && !last;
                AMDS_TRY(amds_vit_cls_scatter(xcq, xq, copy16 ? hq : nullptr, copy16 ? rs : nullptr, q.nt, T, D, c->ln_eps, dt, s));
            }
        }
        return AMDS_OK;
    };
    if (nparts == 2) {     // fork: the tail chain starts once the patch embedding is done
        AMDS_HIP(hipEventRecord(sd.ev_fork, st));
        AMDS_HIP(hipStreamWaitEvent(sd.side, sd.ev_fork, 0));
    }
    rc = AMDS_OK;
    for (int p = nparts - 1; p >= 0 && rc == AMDS_OK; --p) rc = run_tiles(parts[p]);      // the tail's launches are queued first
    if (nparts == 2) {     // join on EVERY path: after an error nothing may still be running on the side stream unordered
        const hipError_t e1 = hipEventRecord(sd.ev_join, sd.side);
        const hipError_t e2 = hipStreamWaitEvent(st, sd.ev_join, 0);
        if (rc != AMDS_OK) return rc;
        if (e1 != hipSuccess) return hip_fail(e1, "hipEventRecord(join)");
        if (e2 != hipSuccess) return hip_fail(e2, "hipStreamWaitEvent(join)");
    }
    if (rc != AMDS_OK) return rc;
    // final LayerNorm: CLS rows -> fp16 features (".half()" of the reference); optionally all tokens in fp32
    if (planes && !tl) {      // the planes hold the stream: class rows (and, for the token tensor, every row) back to fp32
        AMDS_TRY(amds_planes_to_f32(h, lo, D, T, xc, D, Bc, D, st));
        if (tokens_f32) AMDS_TRY(amds_planes_to_f32(h, lo, D, 1, x, D, M, D, st));
    }
    if (tl || planes) AMDS_TRY(amds_layernorm(xc, D, w->norm_w, w->norm_b, feats_f16, D, Bc, D, c->ln_eps, AMDS_F16, st));      // the class stream holds the last block's output
    else AMDS_TRY(amds_layernorm(x, (long)T * D, w->norm_w, w->norm_b, feats_f16, D, Bc, D, c->ln_eps, AMDS_F16, st));
    if (tokens_f32) AMDS_TRY(amds_layernorm(x, D, w->norm_w, w->norm_b, tokens_f32, D, M, D, c->ln_eps, AMDS_F32, st));
#undef AMDS_TRY
    return AMDS_OK;
}

}  // namespace amds

using namespace amds;

extern "C" size_t amds_vit_workspace_diag_offset(const amds_vit_cfg* cfg_host, int batch) {
    VitPlan p;
    if (make_plan(cfg_host, batch, &p) != AMDS_OK) return 0;
    return p.off_diag;
}

extern "C" size_t amds_vit_workspace_bytes(const amds_vit_cfg* cfg_host, int batch) {
    VitPlan p;
    if (make_plan(cfg_host, batch, &p) != AMDS_OK) return 0;
    return p.total;
}

extern "C" int amds_vit_forward_tokens(const amds_vit_cfg* cfg_host, const amds_vit_weights* w_host, const uint8_t* tiles,
                                       void* feats_f16, float* tokens_f32, int B, int chunk, void* ws, size_t ws_bytes,
                                       void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && tiles && feats_f16 && ws, "amds_vit_forward: null pointer");
    AMDS_REQUIRE(B >= 0 && chunk > 0, "amds_vit_forward: bad B=%d chunk=%d", B, chunk);
    AMDS_REQUIRE(w_host->patch_w && w_host->patch_b && w_host->pos_patch && w_host->blocks_host && w_host->norm_w && w_host->norm_b,
                 "amds_vit_forward: incomplete weights");
    AMDS_REQUIRE(cfg_host->n_prefix == 0 || w_host->prefix, "amds_vit_forward: prefix tokens missing");
    VitPlan pl;
    int rc = make_plan(cfg_host, chunk, &pl);
    if (rc != AMDS_OK) return rc;
    if (ws_bytes < pl.total) {
        set_error("amds_vit_forward: workspace %zu < required %zu bytes", ws_bytes, pl.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_vit_forward: workspace must be 256-byte aligned");
    // Ragged-tail schedule (vit_chunk): needs a side stream, which belongs to the context of the current device -- used when the host
    // created one (amds_create).  The fork / join events are per call, so two host threads may run forwards on one device (their tail
    // chains then share the side stream, in order).  AMDS_VIT_TAIL=0 keeps everything on `stream` (A/B).
    VitSide sd;
    static const bool tail_on = !(getenv("AMDS_VIT_TAIL") && atoi(getenv("AMDS_VIT_TAIL")) == 0);
    bool any_tail = false;
    for (int b0 = 0; b0 < B; b0 += chunk) any_tail = any_tail || ((B - b0 < chunk ? B - b0 : chunk) >= 8);
    if (tail_on && any_tail) {
        if (amds_ctx* cx = ctx_of_current_device()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (ctx_side_stream(cx, &sd.side, &a, &b) != AMDS_OK) sd.side = nullptr;
            if (sd.side) {
                AMDS_HIP(hipEventCreateWithFlags(&sd.ev_fork, hipEventDisableTiming));
                AMDS_HIP(hipEventCreateWithFlags(&sd.ev_join, hipEventDisableTiming));
            }
        }
    }
    const size_t tile_bytes = (size_t)cfg_host->img * cfg_host->img * 3;
    rc = AMDS_OK;
    for (int b0 = 0; b0 < B && rc == AMDS_OK; b0 += chunk) {
        const int bc = (B - b0 < chunk) ? B - b0 : chunk;
        rc = vit_chunk(cfg_host, w_host, pl, tiles + (size_t)b0 * tile_bytes,
                       reinterpret_cast<char*>(feats_f16) + (size_t)b0 * cfg_host->dim * 2,
                       tokens_f32 ? tokens_f32 + (size_t)b0 * pl.T * cfg_host->dim : nullptr, bc,
                       reinterpret_cast<char*>(ws), (hipStream_t)stream, sd);
    }
    if (sd.ev_fork) (void)hipEventDestroy(sd.ev_fork);       // released once the recorded work has completed
    if (sd.ev_join) (void)hipEventDestroy(sd.ev_join);
    return rc;
}

// Two chunks in flight on two streams: the HBM-bound kernels of one chunk (LayerNorm, attention staging, epilogue
// tails) run in the shadow of the other chunk's MFMA-bound GEMMs.  `ws` must hold 2 x amds_vit_workspace_bytes(chunk).
// The side stream and its fork / join events belong to the context (created on first use on the context's device).
extern "C" int amds_vit_forward_overlapped(amds_ctx* ctx, const amds_vit_cfg* cfg_host, const amds_vit_weights* w_host, const uint8_t* tiles,
                                           void* feats_f16, int B, int chunk, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(ctx && cfg_host && w_host && tiles && feats_f16 && ws, "amds_vit_forward_overlapped: null pointer");
    AMDS_REQUIRE(B >= 0 && chunk > 0, "amds_vit_forward_overlapped: bad B=%d chunk=%d", B, chunk);
    VitPlan pl;
    int rc = make_plan(cfg_host, chunk, &pl);
    if (rc != AMDS_OK) return rc;
    if (ws_bytes < 2 * pl.total) {
        set_error("amds_vit_forward_overlapped: workspace %zu < required %zu bytes", ws_bytes, 2 * pl.total);
        return AMDS_ERR_WORKSPACE;
    }
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    rc = ctx_side_stream(ctx, &side, &ev_in, &ev_out);
    if (rc != AMDS_OK) return rc;
    hipStream_t mainst = (hipStream_t)stream;
    AMDS_HIP(hipEventRecord(ev_in, mainst));
    AMDS_HIP(hipStreamWaitEvent(side, ev_in, 0));
    const size_t tile_bytes = (size_t)cfg_host->img * cfg_host->img * 3;
    int idx = 0;
    for (int b0 = 0; b0 < B && rc == AMDS_OK; b0 += chunk, ++idx) {
        const int bc = (B - b0 < chunk) ? B - b0 : chunk;
        rc = vit_chunk(cfg_host, w_host, pl, tiles + (size_t)b0 * tile_bytes,
                       reinterpret_cast<char*>(feats_f16) + (size_t)b0 * cfg_host->dim * 2, nullptr, bc,
                       reinterpret_cast<char*>(ws) + (idx & 1) * pl.total, (idx & 1) ? side : mainst);
    }
    // join the side stream back into the caller's stream on EVERY path: after an error nothing may still be running on it unordered
    const hipError_t e1 = hipEventRecord(ev_out, side);
    const hipError_t e2 = hipStreamWaitEvent(mainst, ev_out, 0);
    if (rc != AMDS_OK) return rc;
    if (e1 != hipSuccess) return hip_fail(e1, "hipEventRecord(join)");
    if (e2 != hipSuccess) return hip_fail(e2, "hipStreamWaitEvent(join)");
    return AMDS_OK;
}

extern "C" int amds_vit_forward(const amds_vit_cfg* cfg_host, const amds_vit_weights* w_host, const uint8_t* tiles,
                                void* feats_f16, int B, int chunk, void* ws, size_t ws_bytes, void* stream) {
    return amds_vit_forward_tokens(cfg_host, w_host, tiles, feats_f16, nullptr, B, chunk, ws, ws_bytes, stream);
}
