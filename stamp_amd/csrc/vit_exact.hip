// vit_exact.hip -- the opt-in "exact class-token rows" path of the tile encoder (amds_vit_weights.exact_host).
//
// What the reference stores per tile is ONE row of the network's output: model(tiles)[:, 0].half()
// (src/stamp/preprocessing/__init__.py:324-325, extractor/virchow2.py:29-30).  That row's own chain of products -- its query row, its
// attention output, its proj / fc1 / fc2 rows in every block -- carries most of its 16-bit rounding error (tools/rounding_budget.py:
// UNI2-h class row 7.6e-4 -> 2.3e-4 relative L2 when only those rows are kept exact; the other tokens' errors reach it through keys and
// values only).  It is 1 row in 257-265, so it can be afforded in exact fp32: a separate fp32 "class stream" xc [B][dim] runs beside
// the 16-bit-operand path, block by block:
//     xc += proj( attention( q = Wq LN1(xc), K / V of ALL tokens as the main path stored them ) )        fp32 MFMA + the kernel below
//     xc += fc2( act( fc1( LN2(xc) ) ) )                                                              fp32 MFMA (amds_bgemm_f32)
// with the ORIGINAL fp32 weights (no LayerNorm fold; LayerScale multiplied into proj / fc2 rows in fp32), and after every sub-layer the
// class rows of the main path's residual stream (and of its 16-bit copy + row statistics when LayerNorm is folded) are overwritten with
// xc, so the other tokens attend to the exact class token too.  The stored feature is LN(xc).
#include "common.h"

namespace amds {

// ---------------------------------------------------------------------------------------------
// Attention of ONE fp32 query row per (tile, head) against that tile's stored keys / values.
//   q    fp32 [B][ldq], head h at columns h*HD..          (class-token query, exact)
//   qkv  act dtype [B*T][3*H*HD], thirds q|k|v (the main path's packed tensor: keys / values of ALL tokens as stored)
//   out  fp32 [B][ldo] = softmax(q k^T * scale) v
// One wave per (tile, head).  Scores: 8 lanes share a key row (16-byte pieces, coalesced 128-byte rows), 8 keys per step, partial dot
// products reduced over the 8 lanes; softmax statistics in registers; probabilities through LDS; P.V with lane = (dim pair, key parity).
// Everything fp32, fixed summation order (bit-reproducible).
// ---------------------------------------------------------------------------------------------
constexpr int CLS_TMAX = 288;
template <typename T, int HD>
__global__ void __launch_bounds__(256) cls_attention_f32_kernel(const float* __restrict__ q, long ldq, const T* __restrict__ qkv,
                                                                float* __restrict__ out, long ldo, int B, int Tn, int H, float scale) {
    __shared__ float p_lds[4][CLS_TMAX];
    typedef T vec8 __attribute__((ext_vector_type(8)));
    typedef T vec2 __attribute__((ext_vector_type(2)));
    constexpr int PIECES = HD / 8;            // 16-byte pieces per key row: 8 (head_dim 64) or 10 (80)
    constexpr int NIT = CLS_TMAX / 8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= B * H) return;                // whole waves leave; no workgroup barrier below
    const int b = item / H, h = item - b * H;
    const int D = H * HD;
    const long rs = 3L * D;                   // row stride of the packed tensor
    const int j = lane & 7, g = lane >> 3;
    const float* qr = q + (long)b * ldq + h * HD;
    float qa[8], qb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        qa[e] = qr[8 * j + e];
        qb[e] = (PIECES > 8 && j + 8 < PIECES) ? qr[8 * (j + 8) + e] : 0.f;
    }
    const T* kbase = qkv + (long)b * Tn * rs + D + h * HD;
    float s[NIT];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        s[i] = -INFINITY;
        if (i * 8 < Tn) {                     // wave-uniform
            const int t = i * 8 + g;
            float dot = 0.f;
            if (t < Tn) {
                const vec8 ka = *reinterpret_cast<const vec8*>(kbase + t * rs + 8 * j);
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = fmaf(qa[e], (float)ka[e], dot);
                if (PIECES > 8 && j + 8 < PIECES) {
                    const vec8 kb = *reinterpret_cast<const vec8*>(kbase + t * rs + 8 * (j + 8));
#pragma unroll
                    for (int e = 0; e < 8; ++e) dot = fmaf(qb[e], (float)kb[e], dot);
                }
            }
            dot += __shfl_xor(dot, 1, 64);
            dot += __shfl_xor(dot, 2, 64);
            dot += __shfl_xor(dot, 4, 64);
            if (t < Tn) s[i] = dot * scale;
            m = fmaxf(m, s[i]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 8, 64));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        if (i * 8 < Tn) {
            s[i] = expf(s[i] - m);            // exp(-inf) = 0 for the slots past Tn
            sum += s[i];
        }
    }
    sum += __shfl_xor(sum, 8, 64);             // the 8 lanes of a key hold the same value: add across keys only
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
        if (i * 8 < Tn && j == 0 && i * 8 + g < Tn) p_lds[wave][i * 8 + g] = s[i] * inv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // P.V: lane = (dim pair dp, key group kg)
    constexpr int NDP = HD / 2, NG = 64 / NDP;          // 32 x 2 (head_dim 64), 40 x 1 (80; 24 lanes idle)
    const int dp = lane % NDP, kg = lane / NDP;
    float a0 = 0.f, a1 = 0.f;
    if (kg < NG) {
        const T* vbase = qkv + (long)b * Tn * rs + 2 * D + h * HD + 2 * dp;
#pragma unroll 8
        for (int t = kg; t < Tn; t += NG) {
            const vec2 v = *reinterpret_cast<const vec2*>(vbase + t * rs);
            const float p = p_lds[wave][t];
            a0 = fmaf(p, (float)v[0], a0);
            a1 = fmaf(p, (float)v[1], a1);
        }
    }
    if (NG == 2) {
        a0 += __shfl_xor(a0, 32, 64);
        a1 += __shfl_xor(a1, 32, 64);
    }
    if (kg == 0) *reinterpret_cast<f32x2*>(out + (long)b * ldo + h * HD + 2 * dp) = f32x2{a0, a1};
}

// ---------------------------------------------------------------------------------------------
// Class rows back into the main path: for tile b, x[b*T][:] = xc[b][:], and (LayerNorm folded) the 16-bit copy of that row and its
// (rstd, -mean*rstd).  One wave per tile, row in registers; statistics as amds_ln_stats_cast computes them.
// ---------------------------------------------------------------------------------------------
template <typename TO, int MAXV>
__global__ void __launch_bounds__(256) cls_scatter_kernel(const float* __restrict__ xc, float* __restrict__ x, TO* __restrict__ xh,
                                                          float* __restrict__ rowstat, int B, int Tn, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* src = xc + (long)b * D;
    const long row = (long)b * Tn;
    const int nv = D >> 2;
    typedef TO vec4 __attribute__((ext_vector_type(4)));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + c * 4);
            *reinterpret_cast<f32x4*>(x + row * D + c * 4) = v;
            if (xh != nullptr) {
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
                s2 += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
                vec4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (TO)v[e];
                *reinterpret_cast<vec4*>(xh + row * D + c * 4) = w;
            }
        }
    }
    if (xh != nullptr) {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float mean = s1 / (float)D, var = fmaxf(s2 / (float)D - mean * mean, 0.f), rstd = rsqrtf(var + eps);
        if (lane == 0) reinterpret_cast<f32x2*>(rowstat)[row] = f32x2{rstd, -mean * rstd};
    }
}

// class rows out of the residual stream: xc[b][:] = x[b*T][:]
__global__ void cls_gather_kernel(const float* __restrict__ x, float* __restrict__ xc, int B, int Tn, int D) {
    const long total = (long)B * (D >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / (D >> 2);
        const int c = (int)(i - b * (D >> 2));
        reinterpret_cast<f32x4*>(xc)[i] = *reinterpret_cast<const f32x4*>(x + b * Tn * D + c * 4);
    }
}

// MLP activation in fp32 on [rows][ld]: kind 0: u = gelu_erf(u) in place over `hidden` columns (nn.GELU, exact erf);
// kind 1: u[:, j] = silu(u[:, j]) * u[:, hidden + j] for j < hidden (timm SwiGLUPacked: fc1 -> chunk(2) -> silu(x1) * x2); kind 2: u = silu(u).
__global__ void mlp_act_f32_kernel(float* __restrict__ u, long ld, int rows, int hidden, int kind) {
    const long total = (long)rows * hidden;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / hidden;
        const int c = (int)(i - r * hidden);
        float* p = u + r * ld + c;
        if (kind == 0) {
            *p = gelu_erf(*p);
        } else if (kind == 2) {
            const float g = *p;
            *p = g / (1.0f + expf(-g));
        } else {
            const float g = *p, v = p[hidden];
            *p = (g / (1.0f + expf(-g))) * v;
        }
    }
}

}  // namespace amds

using namespace amds;

extern "C" int amds_attention_cls_f32(const float* q, long ldq, const void* qkv, float* out, long ldo, int B, int T, int H, int head_dim,
                                      int dtype, void* stream) {
    AMDS_REQUIRE(q && qkv && out, "amds_attention_cls_f32: null pointer");
    AMDS_REQUIRE(B >= 0 && H > 0 && T > 0 && T <= CLS_TMAX, "amds_attention_cls_f32: bad B=%d H=%d T=%d (T <= %d)", B, H, T, CLS_TMAX);
    AMDS_REQUIRE(head_dim == 64 || head_dim == 80, "amds_attention_cls_f32: head_dim=%d (64 or 80)", head_dim);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_attention_cls_f32: bad dtype %d", dtype);
    AMDS_REQUIRE(ldq >= (long)H * head_dim && ldo >= (long)H * head_dim && ldo % 2 == 0, "amds_attention_cls_f32: bad pitches");
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)head_dim);
    const dim3 grid(cdiv((long)B * H, 4));
#define AMDS_CLS(TT, HD_) hipLaunchKernelGGL((cls_attention_f32_kernel<TT, HD_>), grid, dim3(256), 0, st, q, ldq, (const TT*)qkv, out, ldo, B, T, H, scale)
    if (dtype == AMDS_F16) { if (head_dim == 64) AMDS_CLS(f16, 64); else AMDS_CLS(f16, 80); }
    else { if (head_dim == 64) AMDS_CLS(bf16, 64); else AMDS_CLS(bf16, 80); }
#undef AMDS_CLS
    AMDS_LAUNCH_CHECK("cls_attention_f32_kernel");
    return AMDS_OK;
}

extern "C" int amds_vit_cls_scatter(const float* xc, float* x, void* xh, float* rowstat, int B, int T, int D, float eps, int dtype, void* stream) {
    AMDS_REQUIRE(xc && x && B >= 0 && T > 0 && D > 0 && D % 4 == 0 && D <= 2048, "amds_vit_cls_scatter: bad arguments (D %% 4 == 0, D <= 2048)");
    AMDS_REQUIRE((xh == nullptr) == (rowstat == nullptr), "amds_vit_cls_scatter: xh and rowstat go together");
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_vit_cls_scatter: bad dtype %d", dtype);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv(B, 4));
    if (dtype == AMDS_F16) hipLaunchKernelGGL((cls_scatter_kernel<f16, 8>), grid, dim3(256), 0, st, xc, x, (f16*)xh, rowstat, B, T, D, eps);
    else hipLaunchKernelGGL((cls_scatter_kernel<bf16, 8>), grid, dim3(256), 0, st, xc, x, (bf16*)xh, rowstat, B, T, D, eps);
    AMDS_LAUNCH_CHECK("cls_scatter_kernel");
    return AMDS_OK;
}

extern "C" int amds_vit_cls_gather(const float* x, float* xc, int B, int T, int D, void* stream) {
    AMDS_REQUIRE(x && xc && B >= 0 && T > 0 && D > 0 && D % 4 == 0, "amds_vit_cls_gather: bad arguments");
    if (B == 0) return AMDS_OK;
    const long total = (long)B * (D >> 2);
    hipLaunchKernelGGL(cls_gather_kernel, dim3((unsigned)(total + 255) / 256 > 1024 ? 1024 : (unsigned)(total + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, x, xc, B, T, D);
    AMDS_LAUNCH_CHECK("cls_gather_kernel");
    return AMDS_OK;
}

extern "C" int amds_mlp_act_f32(float* u, long ld, int rows, int hidden, int kind, void* stream) {
    AMDS_REQUIRE(u && rows >= 0 && hidden > 0 && kind >= 0 && kind <= 2 && ld >= (long)hidden * (kind == 1 ? 2 : 1), "amds_mlp_act_f32: bad arguments");
    if (rows == 0) return AMDS_OK;
    const long total = (long)rows * hidden;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(mlp_act_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, u, ld, rows, hidden, kind);
    AMDS_LAUNCH_CHECK("mlp_act_f32_kernel");
    return AMDS_OK;
}
