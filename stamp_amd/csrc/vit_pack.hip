// vit_pack.hip -- amds_vit_pack: a timm VisionTransformer checkpoint (host fp32 tensors) -> the packed device image amds_vit_forward runs on.
// Host code only (no kernels): everything the Python loader used to do with torch ops -- tile transform folded into the patch
// embedding (optionally as a 16-bit hi | lo pair), class / register tokens + position embedding, LayerNorm folded into qkv / fc1
// (W * gamma re-rounded, b + W beta, row sums of the ROUNDED weights), SwiGLU gate / value padding and 32-row block interleave, K padding
// of fc2, LayerScale multiplied into the exact path's fp32 rows -- so that a non-Python host can use the tile encoder through the C ABI
// alone.  Reference call sites this stands behind: the extractor factories' `load_state_dict` + `.to(device)`
// (src/stamp/preprocessing/extractor/uni2.py:32-43, virchow2.py:34-45, h_optimus_0.py:15-30, reddino.py:40-57).
// Arithmetic in double, rounded once to fp32 and then (round-to-nearest-even) to the 16-bit act dtype -- the same double rounding the
// torch loader performed, so both loaders produce the same bits up to the summation order of the two fp64 reductions (b + W beta, the
// patch bias), which differ by ~1e-16 relative before the final fp32 rounding.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>

namespace amds {
namespace {

inline uint16_t f32_to_f16_bits(float v) {       // round to nearest even, subnormals kept, overflow -> inf
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint32_t sign = u & 0x80000000u;
    u ^= sign;
    uint16_t o;
    if (u >= ((127u + 16u) << 23)) {
        o = u > (255u << 23) ? 0x7e00 : 0x7c00;
    } else if (u < (113u << 23)) {
        const uint32_t magic_bits = ((127u - 15u) + (23u - 10u) + 1u) << 23;
        float f, magic;
        memcpy(&f, &u, 4);
        memcpy(&magic, &magic_bits, 4);
        f += magic;                                // the FPU aligns and rounds the mantissa
        uint32_t r;
        memcpy(&r, &f, 4);
        o = (uint16_t)(r - magic_bits);
    } else {
        const uint32_t odd = (u >> 13) & 1u;
        u += ((uint32_t)(15 - 127) << 23) + 0xfffu + odd;
        o = (uint16_t)(u >> 13);
    }
    return (uint16_t)(o | (sign >> 16));
}
inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) { u = sign; }
        else {                                      // subnormal: m * 2^-24
            float f = (float)m * 5.9604644775390625e-8f;
            memcpy(&u, &f, 4);
            u |= sign;
        }
    } else if (e == 31) { u = sign | 0x7f800000u | (m << 13); }
    else { u = sign | ((e + 112u) << 23) | (m << 13); }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint16_t f32_to_bf16_bits(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_bits_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
struct Act16 {
    int dtype;
    uint16_t enc(float v) const { return dtype == AMDS_F16 ? f32_to_f16_bits(v) : f32_to_bf16_bits(v); }
    float dec(uint16_t h) const { return dtype == AMDS_F16 ? f16_bits_to_f32(h) : bf16_bits_to_f32(h); }
};

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// Bump allocator over one image: the same walk sizes the buffer (base == nullptr) and fills it.
struct Arena {
    char* host;        // staging image (nullptr while sizing)
    char* target;      // address the image will live at (device buffer, or the host buffer itself)
    size_t off = 0;
    template <typename T> T* take(size_t n, const T** target_ptr) {
        off = up256(off);
        T* h = host ? reinterpret_cast<T*>(host + off) : nullptr;
        if (target_ptr) *target_ptr = reinterpret_cast<const T*>(target + off);
        off += n * sizeof(T);
        return h;
    }
};

struct Dims { int D, P, np, kp, Hp, Hr, fc1_rows_real, fc1_rows_pad, n_pos; bool swiglu, fold, split, exact, tail; int lo_shift; };

int make_dims(const amds_vit_cfg* c, const amds_vit_host_weights* s, int flags, Dims* d) {
    AMDS_REQUIRE(c && s, "amds_vit_pack: null cfg / weights");
    AMDS_REQUIRE(c->img > 0 && c->patch > 0 && c->img % c->patch == 0 && c->dim > 0 && c->depth > 0 && c->n_prefix >= 1, "amds_vit_pack: bad cfg");
    AMDS_REQUIRE(c->dtype == AMDS_F16 || c->dtype == AMDS_BF16, "amds_vit_pack: bad act dtype");
    AMDS_REQUIRE(s->patch_w && s->patch_b && s->cls_token && s->pos_embed && s->blocks && s->norm_w && s->norm_b, "amds_vit_pack: incomplete host weights");
    AMDS_REQUIRE(c->n_prefix == 1 || s->reg_token, "amds_vit_pack: %d prefix tokens but no reg_token", c->n_prefix);
    AMDS_REQUIRE(s->hidden > 0 && s->hidden <= c->hidden && c->hidden % 64 == 0, "amds_vit_pack: hidden=%d must be <= cfg.hidden=%d (the 64-padded width)", s->hidden, c->hidden);
    d->D = c->dim; d->P = c->n_prefix;
    const int g = c->img / c->patch;
    d->np = g * g;
    d->kp = ((3 * c->patch * c->patch + 63) / 64) * 64;
    d->Hp = c->hidden; d->Hr = s->hidden;
    d->swiglu = c->mlp_kind == 1;
    AMDS_REQUIRE(d->swiglu || d->Hr == d->Hp, "amds_vit_pack: a GELU MLP's hidden width (%d) must not need padding (cfg.hidden=%d)", d->Hr, d->Hp);
    d->fc1_rows_real = d->swiglu ? 2 * d->Hr : d->Hr;
    d->fc1_rows_pad = d->swiglu ? 2 * d->Hp : d->Hp;
    d->n_pos = d->np + (s->no_embed_class ? 0 : d->P);
    d->fold = (flags & AMDS_PACK_LNFOLD) != 0;
    d->split = (flags & AMDS_PACK_PATCH_SPLIT) != 0;
    d->exact = (flags & AMDS_PACK_EXACT) != 0;
    d->tail = (flags & AMDS_PACK_CLS_TAIL) != 0 && c->mlp_kind != 2;      // the class-row tail has no quick-GELU form
    d->lo_shift = d->split ? (c->dtype == AMDS_F16 ? 11 : 8) : 0;
    if (d->fold) AMDS_REQUIRE(d->D % 256 == 0 && d->fc1_rows_pad % 256 == 0, "amds_vit_pack: LayerNorm fold needs dim %% 256 == 0 and fc1 rows %% 256 == 0 (dim=%d, fc1 rows=%d)", d->D, d->fc1_rows_pad);
    AMDS_REQUIRE(!d->swiglu || d->Hp % 32 == 0, "amds_vit_pack: SwiGLU hidden_pad %% 32");
    for (int l = 0; l < c->depth; ++l) {
        const amds_vit_host_block& b = s->blocks[l];
        AMDS_REQUIRE(b.norm1_w && b.norm1_b && b.qkv_w && b.qkv_b && b.proj_w && b.proj_b && b.norm2_w && b.norm2_b && b.fc1_w && b.fc1_b && b.fc2_w && b.fc2_b,
                     "amds_vit_pack: block %d: incomplete host weights", l);
        AMDS_REQUIRE(!c->layerscale || (b.ls1 && b.ls2), "amds_vit_pack: block %d: LayerScale configured but ls1 / ls2 missing", l);
    }
    return AMDS_OK;
}

// dst [rows_out][ld] act dtype <- W[r][k] * (gamma ? gamma[k] : 1) for the listed source rows (row_src[r] < 0: zero row), zero-padded to ld;
// optionally bias' = b + W beta and the row sums of the ROUNDED weights.
void pack_linear(const Act16& a, const float* W, const float* bias, int K, const std::vector<int>& row_src, const float* gamma, const float* beta,
                 uint16_t* dst, int ld, float* bias_out, float* colsum_out) {
    const int R = (int)row_src.size();
    for (int r = 0; r < R; ++r) {
        uint16_t* drow = dst + (size_t)r * ld;
        const int sr = row_src[r];
        if (sr < 0) {
            for (int k = 0; k < ld; ++k) drow[k] = 0;
            if (bias_out) bias_out[r] = 0.f;
            if (colsum_out) colsum_out[r] = 0.f;
            continue;
        }
        const float* wrow = W + (size_t)sr * K;
        double acc_b = bias ? (double)bias[sr] : 0.0, acc_c = 0.0;
        for (int k = 0; k < K; ++k) {
            const double w = (double)wrow[k];
            const float wf = gamma ? (float)(w * (double)gamma[k]) : wrow[k];
            const uint16_t h = a.enc(wf);
            drow[k] = h;
            if (beta) acc_b += w * (double)beta[k];
            if (colsum_out) acc_c += (double)a.dec(h);
        }
        for (int k = K; k < ld; ++k) drow[k] = 0;
        if (bias_out) bias_out[r] = (float)acc_b;
        if (colsum_out) colsum_out[r] = (float)acc_c;
    }
}

void copy_f32(float* dst, const float* src, size_t n) { memcpy(dst, src, n * sizeof(float)); }

int walk(const amds_vit_cfg* c, const amds_vit_host_weights* s, const Dims& d, Arena& A, amds_vit_weights* ow, amds_vit_block* ob, amds_vit_exact_block* oe) {
    const Act16 a{c->dtype};
    const int D = d.D, p = c->patch, pp = p * p, kreal = 3 * pp;
    const bool fill = A.host != nullptr;
    // ---- patch embedding with the tile transform folded in: conv(W, (u8/255 - mean)/std) + b = (1/255) sum (W/std) u8 + (b - sum W mean/std)
    const int ldp = d.split ? 2 * d.kp : d.kp;
    const void* tp = nullptr;
    uint16_t* pw = A.take<uint16_t>((size_t)D * ldp, reinterpret_cast<const uint16_t**>(&tp));
    if (ow) ow->patch_w = tp;
    float* pb = A.take<float>(D, ow ? &ow->patch_b : nullptr);
    if (fill) {
        const double scale_lo = std::ldexp(1.0, d.lo_shift);
        for (int n = 0; n < D; ++n) {
            double bsum = (double)s->patch_b[n];
            uint16_t* row = pw + (size_t)n * ldp;
            for (int k = 0; k < ldp; ++k) row[k] = 0;
            for (int k = 0; k < kreal; ++k) {
                const int ch = k / pp;
                const double w = (double)s->patch_w[(size_t)n * kreal + k];
                const double wf = w / s->std[ch];
                bsum -= w * (s->mean[ch] / s->std[ch]);
                const uint16_t hi = a.enc((float)wf);
                row[k] = hi;
                if (d.split) row[d.kp + k] = a.enc((float)((wf - (double)a.dec(hi)) * scale_lo));
            }
            pb[n] = (float)bsum;
        }
    }
    // ---- prefix tokens (+ their position rows) and the patch position rows
    float* prefix = A.take<float>((size_t)d.P * D, ow ? &ow->prefix : nullptr);
    float* posp = A.take<float>((size_t)d.np * D, ow ? &ow->pos_patch : nullptr);
    if (fill) {
        for (int t = 0; t < d.P; ++t)
            for (int k = 0; k < D; ++k) {
                float v = t == 0 ? s->cls_token[k] : s->reg_token[(size_t)(t - 1) * D + k];
                if (!s->no_embed_class) v = v + s->pos_embed[(size_t)t * D + k];
                prefix[(size_t)t * D + k] = v;
            }
        copy_f32(posp, s->pos_embed + (s->no_embed_class ? 0 : (size_t)d.P * D), (size_t)d.np * D);
    }
    float* nw = A.take<float>(D, ow ? &ow->norm_w : nullptr);
    float* nb = A.take<float>(D, ow ? &ow->norm_b : nullptr);
    if (fill) { copy_f32(nw, s->norm_w, D); copy_f32(nb, s->norm_b, D); }
    if (ow) { ow->blocks_host = ob; ow->patch_lo_shift = d.lo_shift; ow->exact_host = d.exact ? oe : nullptr; ow->exact_hidden = d.Hr; ow->fp8_host = nullptr; ow->pre_norm_w = nullptr; ow->pre_norm_b = nullptr;
              ow->cls_tail = d.tail ? &oe[c->depth - 1] : nullptr; }

    // ---- row maps
    std::vector<int> id3(3 * D), idD(D), fc1_map(d.fc1_rows_pad);
    for (int i = 0; i < 3 * D; ++i) id3[i] = i;
    for (int i = 0; i < D; ++i) idD[i] = i;
    if (d.swiglu) {       // packed row r: block j = r / 64, gate rows first 32 of the block then the value rows (timm: fc1 -> chunk(2) -> silu(x1) * x2)
        for (int r = 0; r < d.fc1_rows_pad; ++r) {
            const int blk = r / 64, w = r % 64, unit = blk * 32 + (w % 32);
            fc1_map[r] = unit < d.Hr ? (w < 32 ? unit : d.Hr + unit) : -1;
        }
    } else {
        for (int r = 0; r < d.fc1_rows_pad; ++r) fc1_map[r] = r;
    }

    // ---- blocks: layout first (sequential), then the arithmetic in parallel over blocks
    struct Slot { uint16_t *qkv, *proj, *fc1, *fc2; float *qkv_b, *qkv_c, *proj_b, *fc1_b, *fc1_c, *fc2_b, *ln1w, *ln1b, *ln2w, *ln2b, *ls1, *ls2;
                  float *xq_w, *xq_b, *xp_w, *xp_b, *x1_w, *x1_b, *x2_w, *x2_b; };
    std::vector<Slot> slots(c->depth);
    for (int l = 0; l < c->depth; ++l) {
        Slot& t = slots[l];
        amds_vit_block* b = ob ? &ob[l] : nullptr;
        const void* q = nullptr;
#define TAKE16(field, n, dst) do { t.field = A.take<uint16_t>((n), reinterpret_cast<const uint16_t**>(&q)); if (b) b->dst = q; } while (0)
#define TAKE32(field, n, dst) t.field = A.take<float>((n), b ? &b->dst : nullptr)
        TAKE32(ln1w, D, ln1_w); TAKE32(ln1b, D, ln1_b);
        TAKE16(qkv, (size_t)3 * D * D, qkv_w); TAKE32(qkv_b, 3 * D, qkv_b);
        TAKE16(proj, (size_t)D * D, proj_w); TAKE32(proj_b, D, proj_b);
        TAKE32(ln2w, D, ln2_w); TAKE32(ln2b, D, ln2_b);
        TAKE16(fc1, (size_t)d.fc1_rows_pad * D, fc1_w); TAKE32(fc1_b, d.fc1_rows_pad, fc1_b);
        TAKE16(fc2, (size_t)D * d.Hp, fc2_w); TAKE32(fc2_b, D, fc2_b);
        if (c->layerscale) { TAKE32(ls1, D, ls1); TAKE32(ls2, D, ls2); }
        else { t.ls1 = t.ls2 = nullptr; if (b) b->ls1 = b->ls2 = nullptr; }
        if (d.fold) { TAKE32(qkv_c, 3 * D, qkv_colsum); TAKE32(fc1_c, d.fc1_rows_pad, fc1_colsum); }
        else { t.qkv_c = t.fc1_c = nullptr; if (b) b->qkv_colsum = b->fc1_colsum = nullptr; }
#undef TAKE16
#undef TAKE32
        t.xq_w = nullptr;
        if (d.exact || (d.tail && l + 1 == c->depth)) {
            amds_vit_exact_block* e = oe ? &oe[l] : nullptr;
            t.xq_w = A.take<float>((size_t)D * D, e ? &e->q_w : nullptr);        t.xq_b = A.take<float>(D, e ? &e->q_b : nullptr);
            t.xp_w = A.take<float>((size_t)D * D, e ? &e->proj_w : nullptr);     t.xp_b = A.take<float>(D, e ? &e->proj_b : nullptr);
            t.x1_w = A.take<float>((size_t)d.fc1_rows_real * D, e ? &e->fc1_w : nullptr); t.x1_b = A.take<float>(d.fc1_rows_real, e ? &e->fc1_b : nullptr);
            t.x2_w = A.take<float>((size_t)D * d.Hr, e ? &e->fc2_w : nullptr);   t.x2_b = A.take<float>(D, e ? &e->fc2_b : nullptr);
        }
    }
    if (!fill) return AMDS_OK;
    auto do_block = [&](int l) {
        const amds_vit_host_block& hb = s->blocks[l];
        const Slot& t = slots[l];
        copy_f32(t.ln1w, hb.norm1_w, D); copy_f32(t.ln1b, hb.norm1_b, D);
        copy_f32(t.ln2w, hb.norm2_w, D); copy_f32(t.ln2b, hb.norm2_b, D);
        if (d.fold) {
            pack_linear(a, hb.qkv_w, hb.qkv_b, D, id3, hb.norm1_w, hb.norm1_b, t.qkv, D, t.qkv_b, t.qkv_c);
            pack_linear(a, hb.fc1_w, hb.fc1_b, D, fc1_map, hb.norm2_w, hb.norm2_b, t.fc1, D, t.fc1_b, t.fc1_c);
        } else {
            pack_linear(a, hb.qkv_w, hb.qkv_b, D, id3, nullptr, nullptr, t.qkv, D, t.qkv_b, nullptr);
            pack_linear(a, hb.fc1_w, hb.fc1_b, D, fc1_map, nullptr, nullptr, t.fc1, D, t.fc1_b, nullptr);
        }
        pack_linear(a, hb.proj_w, hb.proj_b, D, idD, nullptr, nullptr, t.proj, D, t.proj_b, nullptr);
        pack_linear(a, hb.fc2_w, hb.fc2_b, d.Hr, idD, nullptr, nullptr, t.fc2, d.Hp, t.fc2_b, nullptr);
        if (t.ls1) { copy_f32(t.ls1, hb.ls1, D); copy_f32(t.ls2, hb.ls2, D); }
        if (t.xq_w) {
            copy_f32(t.xq_w, hb.qkv_w, (size_t)D * D); copy_f32(t.xq_b, hb.qkv_b, D);
            copy_f32(t.x1_w, hb.fc1_w, (size_t)d.fc1_rows_real * D); copy_f32(t.x1_b, hb.fc1_b, d.fc1_rows_real);
            for (int n = 0; n < D; ++n) {
                const float g1 = hb.ls1 && c->layerscale ? hb.ls1[n] : 1.0f, g2 = hb.ls2 && c->layerscale ? hb.ls2[n] : 1.0f;
                for (int k = 0; k < D; ++k) t.xp_w[(size_t)n * D + k] = hb.proj_w[(size_t)n * D + k] * g1;
                for (int k = 0; k < d.Hr; ++k) t.x2_w[(size_t)n * d.Hr + k] = hb.fc2_w[(size_t)n * d.Hr + k] * g2;
                t.xp_b[n] = hb.proj_b[n] * g1;
                t.x2_b[n] = hb.fc2_b[n] * g2;
            }
        }
    };
    unsigned nthreads = std::min<unsigned>(std::min<unsigned>((unsigned)c->depth, 16u), std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("AMDS_PACK_THREADS")) nthreads = std::max(1, atoi(e));
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (unsigned i = 0; i < nthreads; ++i)
        pool.emplace_back([&] { for (int l = next.fetch_add(1); l < c->depth; l = next.fetch_add(1)) do_block(l); });
    for (auto& th : pool) th.join();
    return AMDS_OK;
}

}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_vit_pack_bytes(const amds_vit_cfg* cfg_host, const amds_vit_host_weights* src_host, int flags) {
    Dims d;
    if (make_dims(cfg_host, src_host, flags, &d) != AMDS_OK) return 0;
    Arena A{nullptr, nullptr};
    if (walk(cfg_host, src_host, d, A, nullptr, nullptr, nullptr) != AMDS_OK) return 0;
    return up256(A.off);
}

extern "C" int amds_vit_pack_host(const amds_vit_cfg* cfg_host, const amds_vit_host_weights* src_host, int flags, void* image_host, size_t bytes,
                                  const void* target_base, amds_vit_weights* out_w_host, amds_vit_block* out_blocks_host,
                                  amds_vit_exact_block* out_exact_host) {
    Dims d;
    int rc = make_dims(cfg_host, src_host, flags, &d);
    if (rc != AMDS_OK) return rc;
    AMDS_REQUIRE(image_host && out_w_host && out_blocks_host, "amds_vit_pack_host: null output");
    AMDS_REQUIRE(!(d.exact || d.tail) || out_exact_host, "amds_vit_pack_host: AMDS_PACK_EXACT / AMDS_PACK_CLS_TAIL need out_exact [depth]");
    AMDS_REQUIRE(((uintptr_t)image_host & 15) == 0 && ((uintptr_t)target_base & 255) == 0, "amds_vit_pack_host: image must be 16-byte, target 256-byte aligned");
    const size_t need = amds_vit_pack_bytes(cfg_host, src_host, flags);
    if (bytes < need) { set_error("amds_vit_pack: buffer %zu < required %zu bytes", bytes, need); return AMDS_ERR_WORKSPACE; }
    Arena A{reinterpret_cast<char*>(image_host), const_cast<char*>(reinterpret_cast<const char*>(target_base ? target_base : image_host))};
    return walk(cfg_host, src_host, d, A, out_w_host, out_blocks_host, out_exact_host);
}

extern "C" int amds_vit_pack(const amds_vit_cfg* cfg_host, const amds_vit_host_weights* src_host, int flags, void* dev_image, size_t bytes,
                             amds_vit_weights* out_w_host, amds_vit_block* out_blocks_host, amds_vit_exact_block* out_exact_host, void* stream) {
    AMDS_REQUIRE(dev_image, "amds_vit_pack: null device buffer");
    const size_t need = amds_vit_pack_bytes(cfg_host, src_host, flags);
    if (need == 0) return AMDS_ERR_INVALID;
    if (bytes < need) { set_error("amds_vit_pack: buffer %zu < required %zu bytes", bytes, need); return AMDS_ERR_WORKSPACE; }
    void* staging = nullptr;
    AMDS_HIP(hipHostMalloc(&staging, need, hipHostMallocDefault));
    int rc = amds_vit_pack_host(cfg_host, src_host, flags, staging, need, dev_image, out_w_host, out_blocks_host, out_exact_host);
    if (rc == AMDS_OK) {
        hipError_t e = hipMemcpyAsync(dev_image, staging, need, hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);      // one-time call: the staging buffer is released below
        if (e != hipSuccess) rc = hip_fail(e, "amds_vit_pack: upload");
    }
    (void)hipHostFree(staging);
    return rc;
}
