// nystrom_train.hip -- one TransMIL layer's attention in TRAINING, forward and backward as one call each (SURVEY.md 8b:
// amds_nystrom_attn_fwd / _bwd).
//
// Forward: x_res += Dropout(to_out(NystromAttention(y)))  (reference src/stamp/modeling/models/trans_mil.py:81-163 with mask = None, the
// residual of :263, Dropout(0.1) of :66 live in train mode), keeping what the backward needs in ONE caller-owned arena.  Backward: what
// autograd derives from it (the reference calls loss.backward() through Lightning, models/__init__.py:239-279), hand-derived:
//   out_h = (a1 z)(a3 v) + conv33(v),  a_i = softmax(S_i),  S1 = scale q kl^T, S2 = ql kl^T, S3 = ql k^T,  z = pinv(a2)
//   d(a1 z) = do (a3 v)^T      d(a3 v) = (a1 z)^T do      da1 = d(a1 z) z^T      dz = a1^T d(a1 z)
//   da3 = d(a3 v) v^T          dv = a3^T d(a3 v) + conv33^T(do)
//   pinv iteration k (A = a2 z_k, T1 = 7I - A, T2 = 15I - A T1, T3 = 13I - A T2, z_{k+1} = z_k T3 / 4), given g = dz_{k+1}:
//       dz_k = g T3^T / 4 ; dT3 = z_k^T g / 4 ; dA = -dT3 T2^T ; dT2 = -A^T dT3 ; dA -= dT2 T1^T ; dA += A^T dT2
//       da2 += dA z_k^T ; dz_k += a2^T dA           and z_0 = a2^T / (max row-sum * max col-sum) through amds_pinv_init_bwd
//   dS_i = a_i o (da_i - rowsum(a_i o da_i))
//   dq = scale dS1 kl + (scale / l) broadcast(dS2 kl + dS3 k) ;  dk = dS3^T ql + (1 / l) broadcast(scale dS1^T q + dS2^T ql)
// Everything fp32 on the exact-fp32 MFMA (amds_bgemm_f32) and the kernels of transmil.hip / train.hip / dropout.hip: launch sequences,
// nothing allocated, no host synchronisation.
#include <algorithm>
#include "common.h"

namespace amds {
namespace {

inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
constexpr int HEADS = 8, ITERS = 6, CONV_K = 33;          // trans_mil.py:252-254, :52

struct NyDims {
    int Cd, d, m, n, pad, np, l, b;
    long Z;
};

int ny_dims(int dim, int b, int n, NyDims* p) {
    AMDS_REQUIRE(dim > 0 && dim % 8 == 0, "amds_nystrom_attn: dim=%d must be a positive multiple of 8", dim);
    AMDS_REQUIRE(b > 0 && n > 0, "amds_nystrom_attn: bad shape bags=%d tokens=%d", b, n);
    AMDS_REQUIRE((long)b * HEADS <= 65535, "amds_nystrom_attn: %d bags x 8 heads exceed one launch's batch dimension", b);
    p->Cd = dim; p->d = dim / HEADS; p->m = dim / 2; p->n = n; p->b = b;
    const int rem = n % p->m;
    p->pad = rem > 0 ? p->m - rem : 0;
    p->np = n + p->pad;
    p->l = (n + p->m - 1) / p->m;
    p->Z = (long)b * HEADS;
    return AMDS_OK;
}

struct NySaved {
    size_t yp, qkv, ql, kl, a1, a2, a3, zs, A, T1, T2, T3, av, a1z, merged, out, scratch, total, mm_bytes;
};

void ny_saved(const NyDims& p, NySaved* s) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t b = p.b, H = HEADS, np = p.np, m = p.m, d = p.d, Cd = p.Cd;
    s->mm_bytes = al(b * H * m * m * 4);
    s->yp = take(b * np * Cd * 4);
    s->qkv = take(b * np * 3 * Cd * 4);
    s->ql = take(b * H * m * d * 4);
    s->kl = take(b * H * m * d * 4);
    s->a1 = take(b * H * np * m * 4);
    s->a2 = take(b * H * m * m * 4);
    s->a3 = take(b * H * m * np * 4);
    s->zs = take(s->mm_bytes * (ITERS + 1));          // z_0 .. z_6
    s->A = take(s->mm_bytes * ITERS);
    s->T1 = take(s->mm_bytes * ITERS);
    s->T2 = take(s->mm_bytes * ITERS);
    s->T3 = take(s->mm_bytes * ITERS);
    s->av = take(b * H * m * d * 4);
    s->a1z = take(b * H * np * m * 4);
    s->merged = take(b * np * Cd * 4);
    s->out = take(b * p.n * Cd * 4);                  // to_out's output before the dropout (forward scratch)
    s->scratch = take(256);
    s->total = off;
}

struct NyWs {
    size_t dout, dmerged, dqkv, wflip, da1z, dav, da1, dzA, dzB, da3, da2, dT3, dA, dT2, pinv, dkl, dql, dyp, part, cs, conv, total;
    size_t pinv_bytes, cs_bytes, conv_bytes;
};

void ny_ws(const NyDims& p, NyWs* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t b = p.b, H = HEADS, np = p.np, m = p.m, d = p.d, Cd = p.Cd, mmb = b * H * m * m * 4;
    w->dout = take(b * p.n * Cd * 4);
    w->dmerged = take(b * np * Cd * 4);
    w->dqkv = take(b * np * 3 * Cd * 4);
    w->wflip = take(HEADS * CONV_K * 4);
    w->da1z = take(b * H * np * m * 4);
    w->dav = take(b * H * m * d * 4);
    w->da1 = take(b * H * np * m * 4);
    w->dzA = take(mmb); w->dzB = take(mmb);
    w->da3 = take(b * H * m * np * 4);
    w->da2 = take(mmb); w->dT3 = take(mmb); w->dA = take(mmb); w->dT2 = take(mmb);
    w->pinv_bytes = std::max<size_t>(amds_pinv_init_bwd_workspace_bytes((int)p.Z), 4);
    w->pinv = take(w->pinv_bytes);
    w->dkl = take(b * H * m * d * 4);
    w->dql = take(b * H * m * d * 4);
    w->dyp = take(b * np * Cd * 4);
    w->part = take(b * (size_t)3 * Cd * Cd * 4);
    w->cs_bytes = std::max<size_t>(std::max(amds_colsum_workspace_bytes(p.b, 3 * p.Cd * p.Cd), amds_colsum_workspace_bytes(p.b * p.n, p.Cd)), 4);
    w->cs = take(w->cs_bytes);
    w->conv_bytes = std::max<size_t>(amds_dwconv_seq_wgrad_workspace_bytes(p.b, HEADS, CONV_K), 4);
    w->conv = take(w->conv_bytes);
    w->total = off;
}

// y [b][n][Cd] -> yp [b][pad + n][Cd], `pad` zero rows in FRONT of every bag (:100); and its inverse (rows dropped)
__global__ void __launch_bounds__(128) ny_front_pad_kernel(const float* __restrict__ y, float* __restrict__ yp, int Cd, int n, int pad) {
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
__global__ void __launch_bounds__(128) ny_drop_pad_kernel(const float* __restrict__ yp, float* __restrict__ y, int Cd, int n, int pad) {
    const long row = blockIdx.x;
    const long b = row / n;
    const int s = (int)(row - b * n);
    const float* src = yp + (b * (n + pad) + pad + s) * Cd;
    float* dst = y + row * Cd;
    for (int c = threadIdx.x; c < Cd; c += 128) dst[c] = src[c];
}
__global__ void flip_taps_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * taps) {
        const int r = i / taps, k = i - r * taps;
        out[r * taps + (taps - 1 - k)] = w[i];
    }
}

// Backward of the 33-tap residual convolution when ONE position `row` of every sequence has a gradient g [outer][inner * d] (the class-row tail):
// dv[zo][row + k - taps/2][zi * d + c] = w[zi][k] g[zo][zi * d + c] for the taps that land inside the sequence (plain stores: dv is zero there), and
// dw[zi][k] = sum over zo, c of v[zo][row + k - taps/2][zi * d + c] g[zo][zi * d + c] (one workgroup per (tap, head), fixed summation order).
__global__ void ny_dwconv_row_bwd_kernel(const float* __restrict__ g, long sgo, const float* __restrict__ w, float* __restrict__ dv, long svo, long svi, int ldv,
                                         int inner, int n, int d, int taps, int row) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner, k = blockIdx.z;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int tt = row + k - taps / 2;
    if (c >= d || tt < 0 || tt >= n) return;
    dv[zo * svo + zi * svi + (long)tt * ldv + c] = w[(long)zi * taps + k] * g[zo * sgo + (long)zi * d + c];
}
__global__ void __launch_bounds__(256) ny_dwconv_row_wgrad_kernel(const float* __restrict__ g, long sgo, const float* __restrict__ v, long svo, long svi, int ldv,
                                                                  float* __restrict__ dw, int outer, int n, int d, int taps, int row) {
    __shared__ float red[4];
    const int k = blockIdx.x, zi = blockIdx.y, tid = threadIdx.x;
    const int tt = row + k - taps / 2;
    float s = 0.f;
    if (tt >= 0 && tt < n)
        for (int i = tid; i < outer * d; i += 256) {
            const int zo = i / d, c = i - zo * d;
            s = fmaf(v[zo * svo + zi * svi + (long)tt * ldv + c], g[zo * sgo + (long)zi * d + c], s);
        }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) dw[(long)zi * taps + k] = (red[0] + red[1]) + (red[2] + red[3]);
}

#define RC(call)                          \
    do {                                  \
        int rc__ = (call);                \
        if (rc__ != AMDS_OK) return rc__; \
    } while (0)

inline int bg(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int tflags, float* Cm, int ldc, long sCo, long sCi,
              int outer, int inner, int M, int N, int K, float alpha, float diag, const float* bias, int accumulate, void* st) {
    return amds_bgemm_f32(A, lda, sAo, sAi, B, ldb, sBo, sBi, tflags, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, diag, bias, accumulate, st);
}
// batched product of contiguous [Z][M][K] (or [Z][K][M], tflags & 2) with [Z][K][N] (or [Z][N][K], tflags & 1) -> [Z][M][N]
// square m x m products with two outputs (amds_bgemm_f32_dual)
int mm2(const float* A, const float* B, float* C1, float* C2, long Z, int m, float alpha, float diag, float alpha2, float diag2, void* st) {
    const long mmn = (long)m * m;
    for (long z0 = 0; z0 < Z; z0 += 32768) {
        const int nz = (int)std::min<long>(32768, Z - z0);
        RC(amds_bgemm_f32_dual(A + z0 * mmn, m, mmn, 0, B + z0 * mmn, m, mmn, 0, 0, C1 + z0 * mmn, C2 + z0 * mmn, m, mmn, 0, nz, 1, m, m, m, alpha, diag, alpha2, diag2, st));
    }
    return AMDS_OK;
}

int mm(const float* A, const float* B, int tflags, float* Cm, long Z, int M, int N, int K, float alpha, float diag, int accumulate, void* st) {
    const int lda = (tflags & 2) ? M : K, ldb = (tflags & 1) ? K : N;
    for (long z0 = 0; z0 < Z; z0 += 32768) {                       // gridDim.z limit of one launch
        const int nz = (int)std::min<long>(32768, Z - z0);
        RC(bg(A + z0 * M * K, lda, (long)M * K, 0, B + z0 * K * N, ldb, (long)K * N, 0, tflags, Cm + z0 * M * N, N, (long)M * N, 0, nz, 1, M, N, K, alpha, diag,
              nullptr, accumulate, st));
    }
    return AMDS_OK;
}

}  // namespace
}  // namespace amds

using namespace amds;

extern "C" size_t amds_nystrom_attn_saved_bytes(int dim, int n_bags, int n_tokens) {
    NyDims p;
    if (ny_dims(dim, n_bags, n_tokens, &p) != AMDS_OK) return 0;
    NySaved s;
    ny_saved(p, &s);
    return s.total;
}

extern "C" size_t amds_nystrom_attn_workspace_bytes(int dim, int n_bags, int n_tokens) {
    NyDims p;
    if (ny_dims(dim, n_bags, n_tokens, &p) != AMDS_OK) return 0;
    NyWs w;
    ny_ws(p, &w);
    return w.total;
}

extern "C" int amds_nystrom_attn_fwd(const amds_transmil_layer* w_host, int dim, const float* y, float* x_res, int n_bags, int n_tokens, float p_drop,
                                     uint64_t seed, uint32_t stream_id, void* saved, size_t saved_bytes, void* stream) {
    return amds::nystrom_attn_fwd_ex(w_host, dim, y, x_res, n_bags, n_tokens, p_drop, seed, stream_id, saved, saved_bytes, 0, stream);
}

// cls_only: the caller reads row 0 of every bag of x_res and nothing else (TransMIL's second layer: `self.norm(h)[:, 0]`, trans_mil.py:319-323) -- attn1 and its
// softmax, attn1 z, (attn1 z)(attn3 v), the residual convolution, to_out and its Dropout run for that one row per bag (row `pad` of the front-padded sequence; the
// dropout bits are those of the full tensor's rows b * n); landmarks, attn2 and its pseudo-inverse, attn3 and attn3 v need every token and stay as they are.
// The saved arena keeps its layout: a1 / a1z / merged hold [b][H][1][m] / [b][H][1][m] / [b][Cd] at the front of their regions.
int amds::nystrom_attn_fwd_ex(const amds_transmil_layer* w_host, int dim, const float* y, float* x_res, int n_bags, int n_tokens, float p_drop, uint64_t seed,
                              uint32_t stream_id, void* saved, size_t saved_bytes, int cls_only, void* stream) {
    AMDS_REQUIRE(w_host && y && x_res && saved, "amds_nystrom_attn_fwd: null pointer");
    const amds_transmil_layer& L = *w_host;
    AMDS_REQUIRE(L.qkv_w && L.out_w && L.out_b && L.conv_w, "amds_nystrom_attn_fwd: incomplete weights");
    NyDims p;
    RC(ny_dims(dim, n_bags, n_tokens, &p));
    NySaved s;
    ny_saved(p, &s);
    if (saved_bytes < s.total) {
        set_error("amds_nystrom_attn_fwd: saved-activation arena %zu < required %zu bytes", saved_bytes, s.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)saved & 255) == 0, "amds_nystrom_attn_fwd: arena must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* sv = reinterpret_cast<char*>(saved);
    const int Cd = p.Cd, H = HEADS, m = p.m, d = p.d, n = p.n, np = p.np, pad = p.pad, l = p.l, b = p.b;
    const long Z = p.Z;
    float* yp = reinterpret_cast<float*>(sv + s.yp);
    if (pad) {
        hipLaunchKernelGGL(ny_front_pad_kernel, dim3((unsigned)((long)b * np)), dim3(128), 0, st, y, yp, Cd, n, pad);
        AMDS_LAUNCH_CHECK("ny_front_pad_kernel");
    } else {
        AMDS_HIP(hipMemcpyAsync(yp, y, (size_t)b * n * Cd * 4, hipMemcpyDeviceToDevice, st));
    }
    float* qkv = reinterpret_cast<float*>(sv + s.qkv);
    RC(bg(yp, Cd, 0, 0, L.qkv_w, Cd, 0, 0, 1, qkv, 3 * Cd, 0, 0, 1, 1, b * np, 3 * Cd, Cd, 1.0f, 0.0f, nullptr, 0, stream));
    const float *qp = qkv, *kp = qkv + Cd, *vp = qkv + 2 * Cd;
    const long sb = (long)np * 3 * Cd, sh = d;
    const int ld = 3 * Cd;
    const double scale_d = 1.0 / sqrt((double)d);
    const float scale = (float)scale_d;
    float *ql = reinterpret_cast<float*>(sv + s.ql), *kl = reinterpret_cast<float*>(sv + s.kl);
    RC(amds_landmark_mean(qp, sb, sh, ld, ql, b, H, m, l, d, (float)(scale_d / l), stream));
    RC(amds_landmark_mean(kp, sb, sh, ld, kl, b, H, m, l, d, (float)(1.0 / l), stream));
    float *a1 = reinterpret_cast<float*>(sv + s.a1), *a2 = reinterpret_cast<float*>(sv + s.a2), *a3 = reinterpret_cast<float*>(sv + s.a3);
    const long md = (long)m * d, mmn = (long)m * m, nm = (long)np * m;
    const long hm = (long)H * m;
    if (cls_only) RC(bg(qp + (size_t)pad * ld, ld, sb, sh, kl, d, H * md, md, 1, a1, m, hm, m, b, H, 1, m, d, scale, 0.0f, nullptr, 0, stream));      // the class row of sim1
    else RC(bg(qp, ld, sb, sh, kl, d, H * md, md, 1, a1, m, H * nm, nm, b, H, np, m, d, scale, 0.0f, nullptr, 0, stream));
    RC(bg(ql, d, H * md, md, kl, d, H * md, md, 1, a2, m, H * mmn, mmn, b, H, m, m, d, 1.0f, 0.0f, nullptr, 0, stream));
    RC(bg(ql, d, H * md, md, kp, ld, sb, sh, 1, a3, np, H * nm, nm, b, H, m, np, d, 1.0f, 0.0f, nullptr, 0, stream));
    RC(amds_softmax_rows(a1, cls_only ? Z : Z * np, m, stream));
    RC(amds_softmax_rows(a2, Z * m, m, stream));
    RC(amds_softmax_rows(a3, Z * m, np, stream));
    auto at = [&](size_t base, int k) { return reinterpret_cast<float*>(sv + base + (size_t)k * s.mm_bytes); };
    AMDS_HIP(hipMemsetAsync(sv + s.scratch, 0, 8, st));
    RC(amds_pinv_init(a2, at(s.zs, 0), (int)Z, m, sv + s.scratch, stream));
    for (int k = 0; k < ITERS; ++k) {                                                    // (:29-35), every iterate kept for the backward
        const float* z = at(s.zs, k);
        float *A = at(s.A, k), *T1 = at(s.T1, k), *T2 = at(s.T2, k), *T3 = at(s.T3, k);
        RC(mm2(a2, z, A, T1, Z, m, 1.0f, 0.0f, -1.0f, 7.0f, stream));                    // A = a2 z and T1 = 7 I - A from one product
        RC(mm(A, T1, 0, T2, Z, m, m, m, -1.0f, 15.0f, 0, stream));
        RC(mm(A, T2, 0, T3, Z, m, m, m, -1.0f, 13.0f, 0, stream));
        RC(mm(z, T3, 0, at(s.zs, k + 1), Z, m, m, m, 0.25f, 0.0f, 0, stream));
    }
    const float* z = at(s.zs, ITERS);
    float *av = reinterpret_cast<float*>(sv + s.av), *a1z = reinterpret_cast<float*>(sv + s.a1z), *merged = reinterpret_cast<float*>(sv + s.merged);
    RC(bg(a3, np, H * nm, nm, vp, ld, sb, sh, 0, av, d, H * md, md, b, H, m, d, np, 1.0f, 0.0f, nullptr, 0, stream));
    if (cls_only) {
        RC(bg(a1, m, hm, m, z, m, H * mmn, mmn, 0, a1z, m, hm, m, b, H, 1, m, m, 1.0f, 0.0f, nullptr, 0, stream));                     // one row of attn1 pinv
        RC(bg(a1z, m, hm, m, av, d, H * md, md, 0, merged, Cd, Cd, d, b, H, 1, d, m, 1.0f, 0.0f, nullptr, 0, stream));                 // merged: [b][Cd], the class rows
        RC(amds_dwconv_seq_row(vp, sb, sh, ld, L.conv_w, merged, Cd, d, b, H, np, d, CONV_K, pad, stream));
        if (p_drop > 0.f) {
            float* out = reinterpret_cast<float*>(sv + s.out);
            RC(bg(merged, Cd, Cd, 0, L.out_w, Cd, 0, 0, 1, out, Cd, Cd, 0, b, 1, 1, Cd, Cd, 1.0f, 0.0f, L.out_b, 0, stream));
            return amds_dropout_add_rows(out, Cd, x_res, (long)n * Cd, x_res, (long)n * Cd, b, Cd, n, p_drop, seed, stream_id, stream);
        }
        return bg(merged, Cd, Cd, 0, L.out_w, Cd, 0, 0, 1, x_res, Cd, (long)n * Cd, 0, b, 1, 1, Cd, Cd, 1.0f, 0.0f, L.out_b, 1, stream);
    }
    RC(mm(a1, z, 0, a1z, Z, np, m, m, 1.0f, 0.0f, 0, stream));
    RC(bg(a1z, m, H * nm, nm, av, d, H * md, md, 0, merged, Cd, (long)np * Cd, d, b, H, np, d, m, 1.0f, 0.0f, nullptr, 0, stream));
    RC(amds_dwconv_seq(vp, sb, sh, ld, L.conv_w, merged, (long)np * Cd, d, Cd, b, H, np, d, CONV_K, stream));
    const float* tail = merged + (size_t)pad * Cd;                                        // the last n rows of every bag (:155)
    if (p_drop > 0.f) {
        float* out = reinterpret_cast<float*>(sv + s.out);
        RC(bg(tail, Cd, (long)np * Cd, 0, L.out_w, Cd, 0, 0, 1, out, Cd, (long)n * Cd, 0, b, 1, n, Cd, Cd, 1.0f, 0.0f, L.out_b, 0, stream));
        return amds_dropout_add(out, Cd, x_res, Cd, x_res, Cd, (long)b * n, Cd, p_drop, seed, stream_id, stream);
    }
    return bg(tail, Cd, (long)np * Cd, 0, L.out_w, Cd, 0, 0, 1, x_res, Cd, (long)n * Cd, 0, b, 1, n, Cd, Cd, 1.0f, 0.0f, L.out_b, 1, stream);
}

extern "C" int amds_nystrom_attn_bwd(const amds_transmil_layer* w_host, int dim, const float* dx, float* dy, const amds_nystrom_grads* grads_host, int n_bags,
                                     int n_tokens, float p_drop, uint64_t seed, uint32_t stream_id, const void* saved, size_t saved_bytes, void* ws,
                                     size_t ws_bytes, void* stream) {
    return amds::nystrom_attn_bwd_ex(w_host, dim, dx, dy, grads_host, n_bags, n_tokens, p_drop, seed, stream_id, saved, saved_bytes, ws, ws_bytes, 0, stream);
}

// cls_only: the backward of nystrom_attn_fwd_ex(cls_only = 1): dx has a gradient on row 0 of every bag only (its other rows are not read).
int amds::nystrom_attn_bwd_ex(const amds_transmil_layer* w_host, int dim, const float* dx, float* dy, const amds_nystrom_grads* grads_host, int n_bags, int n_tokens,
                              float p_drop, uint64_t seed, uint32_t stream_id, const void* saved, size_t saved_bytes, void* ws, size_t ws_bytes, int cls_only,
                              void* stream) {
    AMDS_REQUIRE(w_host && dx && dy && saved && ws, "amds_nystrom_attn_bwd: null pointer");
    const amds_transmil_layer& L = *w_host;
    AMDS_REQUIRE(L.qkv_w && L.out_w && L.conv_w, "amds_nystrom_attn_bwd: incomplete weights");
    NyDims p;
    RC(ny_dims(dim, n_bags, n_tokens, &p));
    NySaved s;
    ny_saved(p, &s);
    NyWs w;
    ny_ws(p, &w);
    if (saved_bytes < s.total || ws_bytes < w.total) {
        set_error("amds_nystrom_attn_bwd: arena %zu / workspace %zu < required %zu / %zu bytes", saved_bytes, ws_bytes, s.total, w.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE((((uintptr_t)saved | (uintptr_t)ws) & 255) == 0, "amds_nystrom_attn_bwd: arena and workspace must be 256-byte aligned");
    const amds_nystrom_grads* G = grads_host;
    AMDS_REQUIRE(!G || (G->qkv_w && G->out_w && G->out_b && G->conv_w), "amds_nystrom_attn_bwd: incomplete gradient buffers");
    hipStream_t st = (hipStream_t)stream;
    const char* sv = reinterpret_cast<const char*>(saved);
    char* wk = reinterpret_cast<char*>(ws);
    const int Cd = p.Cd, H = HEADS, m = p.m, d = p.d, n = p.n, np = p.np, pad = p.pad, l = p.l, b = p.b;
    const long Z = p.Z;
    const double scale_d = 1.0 / sqrt((double)d);
    const float scale = (float)scale_d;
    const float* qkv = reinterpret_cast<const float*>(sv + s.qkv);
    const float* merged = reinterpret_cast<const float*>(sv + s.merged);
    const float *qp = qkv, *kp = qkv + Cd, *vp = qkv + 2 * Cd;
    const long sb = (long)np * 3 * Cd, sh = d;
    const int ld = 3 * Cd;
    const long md = (long)m * d, mmn = (long)m * m, nm = (long)np * m;
    auto colsum = [&](const float* x, long ldx, float* out, long rows, int cols) { return amds_colsum(x, ldx, out, (int)rows, cols, AMDS_F32, 0, wk + w.cs, w.cs_bytes, stream); };
    // dW[M][N] = sum_b dy_b^T x_b: one product per bag into fp32 partials, then a fixed-order sum over the bags
    auto wgrad = [&](const float* dy3, int M, long sdy, const float* x3, int N, int ldx, long sx, int rows, float* out) -> int {
        float* part = reinterpret_cast<float*>(wk + w.part);
        RC(bg(dy3, M, sdy, 0, x3, ldx, sx, 0, 2, part, N, (long)M * N, 0, b, 1, M, N, rows, 1.0f, 0.0f, nullptr, 0, stream));
        return colsum(part, (long)M * N, out, b, M * N);
    };
    const long hm = (long)H * m;
    float* dqkv = reinterpret_cast<float*>(wk + w.dqkv);
    float *dqp = dqkv, *dkp = dqkv + Cd, *dvp = dqkv + 2 * Cd;
    const float *av = reinterpret_cast<const float*>(sv + s.av), *a1z = reinterpret_cast<const float*>(sv + s.a1z);
    const float *a1 = reinterpret_cast<const float*>(sv + s.a1), *a2 = reinterpret_cast<const float*>(sv + s.a2), *a3 = reinterpret_cast<const float*>(sv + s.a3);
    const float *ql = reinterpret_cast<const float*>(sv + s.ql), *kl = reinterpret_cast<const float*>(sv + s.kl);
    auto at = [&](size_t base, int k) { return reinterpret_cast<const float*>(sv + base + (size_t)k * s.mm_bytes); };
    float *da1z = reinterpret_cast<float*>(wk + w.da1z), *dav = reinterpret_cast<float*>(wk + w.dav), *da1 = reinterpret_cast<float*>(wk + w.da1);
    float *da3 = reinterpret_cast<float*>(wk + w.da3), *da2 = reinterpret_cast<float*>(wk + w.da2);
    const float* zf = at(s.zs, ITERS);
    float *dz = reinterpret_cast<float*>(wk + w.dzA), *dzk = reinterpret_cast<float*>(wk + w.dzB);
    if (cls_only) {
        // ---- the class rows alone: dout [b][Cd] = Dropout'(dx rows b * n), merged [b][Cd] as the forward left it
        float* dout = reinterpret_cast<float*>(wk + w.dout);
        RC(dropout_cast_bwd_rows_dt(dx, (long)n * Cd, dout, Cd, b, Cd, n, AMDS_F32, p_drop, seed, stream_id, stream));
        if (G) {
            RC(bg(dout, Cd, 0, 0, merged, Cd, 0, 0, 2, G->out_w, Cd, 0, 0, 1, 1, Cd, Cd, b, 1.0f, 0.0f, nullptr, 0, stream));                       // dWo = dout^T merged (K = bags)
            RC(colsum(dout, Cd, G->out_b, b, Cd));
        }
        float* dmc = reinterpret_cast<float*>(wk + w.dmerged);                                                                                    // d(merged) [b][Cd]
        RC(bg(dout, Cd, 0, 0, L.out_w, Cd, 0, 0, 0, dmc, Cd, 0, 0, 1, 1, b, Cd, Cd, 1.0f, 0.0f, nullptr, 0, stream));
        AMDS_HIP(hipMemsetAsync(dqkv, 0, (size_t)b * np * 3 * Cd * 4, st));
        hipLaunchKernelGGL(ny_dwconv_row_bwd_kernel, dim3(cdiv(d, 64), b * H, CONV_K), dim3(64), 0, st, dmc, (long)Cd, L.conv_w, dvp, sb, sh, ld, H, np, d, CONV_K, pad);
        AMDS_LAUNCH_CHECK("ny_dwconv_row_bwd_kernel");
        if (G) {
            hipLaunchKernelGGL(ny_dwconv_row_wgrad_kernel, dim3(CONV_K, H), dim3(256), 0, st, dmc, (long)Cd, vp, sb, sh, ld, G->conv_w, b, np, d, CONV_K, pad);
            AMDS_LAUNCH_CHECK("ny_dwconv_row_wgrad_kernel");
        }
        RC(bg(dmc, Cd, Cd, d, av, d, H * md, md, 1, da1z, m, hm, m, b, H, 1, m, d, 1.0f, 0.0f, nullptr, 0, stream));                               // do av^T        [1][m]
        RC(bg(a1z, m, hm, m, dmc, Cd, Cd, d, 2, dav, d, H * md, md, b, H, m, d, 1, 1.0f, 0.0f, nullptr, 0, stream));                                // a1z^T do       rank 1
        RC(bg(da1z, m, hm, m, zf, m, H * mmn, mmn, 1, da1, m, hm, m, b, H, 1, m, m, 1.0f, 0.0f, nullptr, 0, stream));                               // d(a1z) z^T     [1][m]
        RC(bg(a1, m, hm, m, da1z, m, hm, m, 2, dz, m, H * mmn, mmn, b, H, m, m, 1, 1.0f, 0.0f, nullptr, 0, stream));                                // a1^T d(a1z)    rank 1
    } else {
    // ---- to_out (+ Dropout)
    const float* dout = dx;
    if (p_drop > 0.f) {
        float* t = reinterpret_cast<float*>(wk + w.dout);
        RC(amds_dropout_cast_bwd(dx, Cd, t, Cd, (long)b * n, Cd, AMDS_F32, p_drop, seed, stream_id, stream));
        dout = t;
    }
    const float* tail = merged + (size_t)pad * Cd;
    if (G) {
        RC(wgrad(dout, Cd, (long)n * Cd, tail, Cd, Cd, (long)np * Cd, n, G->out_w));                                   // dWo = dout^T merged_tail
        RC(colsum(dout, Cd, G->out_b, (long)b * n, Cd));
    }
    float* dmerged = reinterpret_cast<float*>(wk + w.dmerged);
    AMDS_HIP(hipMemsetAsync(dmerged, 0, (size_t)b * np * Cd * 4, st));
    RC(bg(dout, Cd, (long)n * Cd, 0, L.out_w, Cd, 0, 0, 0, dmerged + (size_t)pad * Cd, Cd, (long)np * Cd, 0, b, 1, n, Cd, Cd, 1.0f, 0.0f, nullptr, 0, stream));
    AMDS_HIP(hipMemsetAsync(dqkv, 0, (size_t)b * np * 3 * Cd * 4, st));
    // 33-tap residual conv on v: data gradient = the same conv with reversed taps; weight gradient = a reduction
    float* wflip = reinterpret_cast<float*>(wk + w.wflip);
    hipLaunchKernelGGL(flip_taps_kernel, dim3(2), dim3(256), 0, st, L.conv_w, wflip, HEADS, CONV_K);
    AMDS_LAUNCH_CHECK("flip_taps_kernel");
    RC(amds_dwconv_seq(dmerged, (long)np * Cd, d, Cd, wflip, dvp, sb, sh, ld, b, H, np, d, CONV_K, stream));
    if (G) RC(amds_dwconv_seq_wgrad(dmerged, (long)np * Cd, d, Cd, vp, sb, sh, ld, G->conv_w, b, H, np, d, CONV_K, wk + w.conv, w.conv_bytes, stream));
    // out_h = a1z av  (do = head slice of dmerged, [np, d] at row pitch Cd)
    RC(bg(dmerged, Cd, (long)np * Cd, d, av, d, H * md, md, 1, da1z, m, H * nm, nm, b, H, np, m, d, 1.0f, 0.0f, nullptr, 0, stream));                  // do av^T
    RC(bg(a1z, m, H * nm, nm, dmerged, Cd, (long)np * Cd, d, 2, dav, d, H * md, md, b, H, m, d, np, 1.0f, 0.0f, nullptr, 0, stream));                   // a1z^T do
    RC(mm(da1z, zf, 1, da1, Z, np, m, m, 1.0f, 0.0f, 0, stream));                                                                                     // d(a1z) z^T
    RC(mm(a1, da1z, 2, dz, Z, m, m, np, 1.0f, 0.0f, 0, stream));                                                                                      // a1^T d(a1z)
    }
    RC(bg(dav, d, H * md, md, vp, ld, sb, sh, 1, da3, np, H * nm, nm, b, H, m, np, d, 1.0f, 0.0f, nullptr, 0, stream));                                // d(av) v^T
    RC(bg(a3, np, H * nm, nm, dav, d, H * md, md, 2, dvp, ld, sb, sh, b, H, np, d, m, 1.0f, 0.0f, nullptr, 1, stream));                                // dv += a3^T d(av)
    // pseudo-inverse iterations, last to first
    AMDS_HIP(hipMemsetAsync(da2, 0, (size_t)Z * mmn * 4, st));
    float *dT3 = reinterpret_cast<float*>(wk + w.dT3), *dA = reinterpret_cast<float*>(wk + w.dA), *dT2 = reinterpret_cast<float*>(wk + w.dT2);
    for (int k = ITERS - 1; k >= 0; --k) {
        const float *zk = at(s.zs, k), *A = at(s.A, k), *T1 = at(s.T1, k), *T2 = at(s.T2, k), *T3 = at(s.T3, k);
        RC(mm(dz, T3, 1, dzk, Z, m, m, m, 0.25f, 0.0f, 0, stream));               // g T3^T / 4
        RC(mm(zk, dz, 2, dT3, Z, m, m, m, 0.25f, 0.0f, 0, stream));               // z_k^T g / 4
        RC(mm(dT3, T2, 1, dA, Z, m, m, m, -1.0f, 0.0f, 0, stream));               // -dT3 T2^T
        RC(mm(A, dT3, 2, dT2, Z, m, m, m, -1.0f, 0.0f, 0, stream));               // -A^T dT3
        RC(mm(dT2, T1, 1, dA, Z, m, m, m, -1.0f, 0.0f, 1, stream));               // dA -= dT2 T1^T
        RC(mm(A, dT2, 2, dA, Z, m, m, m, 1.0f, 0.0f, 1, stream));                 // dA -= dT1, dT1 = -A^T dT2
        RC(mm(dA, zk, 1, da2, Z, m, m, m, 1.0f, 0.0f, 1, stream));                // da2 += dA z_k^T
        RC(mm(a2, dA, 2, dzk, Z, m, m, m, 1.0f, 0.0f, 1, stream));                // dz_k += a2^T dA
        std::swap(dz, dzk);
    }
    RC(amds_pinv_init_bwd(a2, dz, da2, (int)Z, m, wk + w.pinv, w.pinv_bytes, stream));
    // the three softmaxes (in place: da_i becomes dS_i)
    RC(amds_softmax_rows_bwd(a1, da1, cls_only ? Z : Z * np, m, stream));
    RC(amds_softmax_rows_bwd(a2, da2, Z * m, m, stream));
    RC(amds_softmax_rows_bwd(a3, da3, Z * m, np, stream));
    const float *dS1 = da1, *dS2 = da2, *dS3 = da3;
    float *dkl = reinterpret_cast<float*>(wk + w.dkl), *dql = reinterpret_cast<float*>(wk + w.dql);
    if (cls_only) {
        RC(bg(dS1, m, hm, m, kl, d, H * md, md, 0, dqp + (size_t)pad * ld, ld, sb, sh, b, H, 1, d, m, scale, 0.0f, nullptr, 0, stream));                // the class query's row
        RC(bg(dS1, m, hm, m, qp + (size_t)pad * ld, ld, sb, sh, 2, dkl, d, H * md, md, b, H, m, d, 1, scale, 0.0f, nullptr, 0, stream));                // dS1^T q, rank 1
    } else {
    RC(bg(dS1, m, H * nm, nm, kl, d, H * md, md, 0, dqp, ld, sb, sh, b, H, np, d, m, scale, 0.0f, nullptr, 0, stream));
    RC(bg(dS1, m, H * nm, nm, qp, ld, sb, sh, 2, dkl, d, H * md, md, b, H, m, d, np, scale, 0.0f, nullptr, 0, stream));                                // dS1^T q
    }
    RC(bg(dS2, m, H * mmn, mmn, ql, d, H * md, md, 2, dkl, d, H * md, md, b, H, m, d, m, 1.0f, 0.0f, nullptr, 1, stream));                             // + dS2^T q_l
    RC(bg(dS2, m, H * mmn, mmn, kl, d, H * md, md, 0, dql, d, H * md, md, b, H, m, d, m, 1.0f, 0.0f, nullptr, 0, stream));
    RC(bg(dS3, np, H * nm, nm, kp, ld, sb, sh, 0, dql, d, H * md, md, b, H, m, d, np, 1.0f, 0.0f, nullptr, 1, stream));
    RC(bg(dS3, np, H * nm, nm, ql, d, H * md, md, 2, dkp, ld, sb, sh, b, H, np, d, m, 1.0f, 0.0f, nullptr, 0, stream));                                // dS3^T q_l
    RC(amds_landmark_mean_bwd(dql, dqp, sb, sh, ld, b, H, m, l, d, (float)(scale_d / l), 1, stream));
    RC(amds_landmark_mean_bwd(dkl, dkp, sb, sh, ld, b, H, m, l, d, (float)(1.0 / l), 1, stream));
    // to_qkv (no bias)
    const float* yp = reinterpret_cast<const float*>(sv + s.yp);
    if (G) RC(wgrad(dqkv, 3 * Cd, (long)np * 3 * Cd, yp, Cd, Cd, (long)np * Cd, np, G->qkv_w));
    float* dyp = reinterpret_cast<float*>(wk + w.dyp);
    RC(bg(dqkv, 3 * Cd, 0, 0, L.qkv_w, Cd, 0, 0, 0, dyp, Cd, 0, 0, 1, 1, b * np, Cd, 3 * Cd, 1.0f, 0.0f, nullptr, 0, stream));
    hipLaunchKernelGGL(ny_drop_pad_kernel, dim3((unsigned)((long)b * n)), dim3(128), 0, st, dyp, dy, Cd, n, pad);
    AMDS_LAUNCH_CHECK("ny_drop_pad_kernel");
    return AMDS_OK;
}
