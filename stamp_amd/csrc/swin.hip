// swin.hip -- the reference's in-tree tile encoder, CTransPath = ConvStem + Swin-T
// (src/stamp/preprocessing/extractor/ctranspath.py), u8 tiles -> 768-d features, as a fixed chain of launches.
//
// Token-major design: the residual stream x is fp32 [B][G*G][C] in NATURAL raster order for the whole depth.  The
// reference's roll / window_partition / window_reverse / roll-back (ctranspath.py:663-690) are pure row
// permutations and LayerNorm / Linear are row-wise, so only the window-attention kernel knows about windows: it
// gathers the 49 rows of a (shifted) window by index arithmetic and scatters its result to the same rows.
//   stem   : u8 -> normalise -> conv3x3 s2 + BN + ReLU -> conv3x3 s2 + BN + ReLU -> conv1x1 -> LayerNorm   (one kernel)
//   block  : h = LN1(x) | qkv = h Wqkv^T + b (MFMA GEMM) | window attention | x += a Wproj^T + b (GEMM, fp32 residual)
//            h = LN2(x) | u = gelu(h W1^T + b1) (GEMM) | x += u W2^T + b2 (GEMM)
//   merge  : gather 2x2 cells + LayerNorm(4C) (one kernel) | x' = . Wred^T (GEMM, fp32 out)
//   head   : LayerNorm + mean over tokens (one kernel) -> fp16 (".half()" of the reference loop) and/or fp32
#include "common.h"
#include "rowstream.h"
#include <stdlib.h>

namespace amds {

constexpr int SW_WS = 7, SW_N = 49, SW_HD = 32;
constexpr int SW_VS = 136;                       // V^T row stride in bytes (64 keys + 4 pad halves): conflict-free b64 reads
constexpr int SW_VT_BYTES = SW_HD * SW_VS;       // per wave

// ------------------------------------------------------------------------------------------------
// ConvStem (ctranspath.py:386-444) + patch LayerNorm (:905-911).  One workgroup = 7x7 output tokens of one tile.
// Packed fp32 parameter block (host: stamp_amd/swin.py::_pack_stem), BatchNorm (eval) folded into the conv weights:
//   [0..2] a_c = 1/(255 std_c)   [3..5] b_c = -mean_c/std_c   [6,7] unused
//   w1[(ci*9+ky*3+kx)*C1 + co]  b1[C1]  w2[(ci*9+ky*3+kx)*C2 + co]  b2[C2]  w3[ci*C0 + co]  b3[C0]  ln_w[C0]  ln_b[C0]
// Zero padding is applied in the NORMALISED domain (the reference normalises first, then the conv pads with 0) and
// conv-1 outputs outside the 112x112 map are forced to 0 (they are conv-2's padding), not computed.
// ------------------------------------------------------------------------------------------------
template <int C0>
__global__ void __launch_bounds__(256) swin_stem_kernel(const uint8_t* __restrict__ tiles, float* __restrict__ x,
                                                        const float* __restrict__ prm, int S, int G, float eps) {
    // Work split: a lane owns a pixel, a wave owns a channel group, so every weight is wave-uniform and is fetched by
    // scalar loads straight into SGPR operands of the FMAs (no LDS copy of the weights, one LDS read of the input per
    // 6-24 FMAs).  v1 (weights in LDS, one output element per thread, 2 LDS reads per FMA): 536 us per 256 tiles.
    constexpr int C1 = C0 / 8, C2 = C0 / 4, R = 7, R1 = 2 * R + 1, R0 = 4 * R + 3;
    constexpr int N_W1 = 27 * C1, N_W2 = 9 * C1 * C2, N_W3 = C2 * C0;
    constexpr int G1 = C1 / 4, G2 = C2 / 4, G3 = C0 / 4;       // channels per wave in conv1 / conv2 / conv1x1
    __shared__ float s_in[R0 * R0 * 3];
    __shared__ float s_c1[R1 * R1 * C1];
    __shared__ float s_c2[R * R * C2];
    __shared__ float s_red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, by = blockIdx.y, b = blockIdx.z;
    const float* __restrict__ p_w1 = prm + 8;
    const float* __restrict__ p_w2 = p_w1 + N_W1 + C1;
    const float* __restrict__ p_w3 = p_w2 + N_W2 + C2;
    const float* __restrict__ p_ln = p_w3 + N_W3 + C0;
    const float na0 = prm[0], na1 = prm[1], na2 = prm[2], nb0 = prm[3], nb1 = prm[4], nb2 = prm[5];
    const uint8_t* tile = tiles + (size_t)b * S * S * 3;
    const int y00 = 4 * R * by - 3, x00 = 4 * R * bx - 3;
    for (int i = tid; i < R0 * R0; i += 256) {
        const int iy = i / R0, ix = i - iy * R0, gy = y00 + iy, gx = x00 + ix;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (gy >= 0 && gy < S && gx >= 0 && gx < S) {
            const uint8_t* p = tile + ((size_t)gy * S + gx) * 3;
            v0 = fmaf((float)p[0], na0, nb0); v1 = fmaf((float)p[1], na1, nb1); v2 = fmaf((float)p[2], na2, nb2);
        }
        s_in[i * 3 + 0] = v0; s_in[i * 3 + 1] = v1; s_in[i * 3 + 2] = v2;
    }
    __syncthreads();
    // conv1 + BN + ReLU on the (2R+1)^2 halo region: wave = 3 output channels, lane = pixel (4 passes of 64)
    const int S1 = S / 2;
    for (int pix = lane; pix < R1 * R1; pix += 64) {
        const int ly = pix / R1, lx = pix - ly * R1;
        const int y1 = 2 * R * by - 1 + ly, x1 = 2 * R * bx - 1 + lx;
        float acc[G1];
#pragma unroll
        for (int c = 0; c < G1; ++c) acc[c] = p_w1[N_W1 + wave * G1 + c];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* in = &s_in[((2 * ly + ky) * R0 + 2 * lx + kx) * 3];
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float a = in[ci];
#pragma unroll
                    for (int c = 0; c < G1; ++c) acc[c] = fmaf(a, p_w1[(ci * 9 + ky * 3 + kx) * C1 + wave * G1 + c], acc[c]);
                }
            }
        const bool inside = y1 >= 0 && y1 < S1 && x1 >= 0 && x1 < S1;
#pragma unroll
        for (int c = 0; c < G1; ++c) s_c1[pix * C1 + wave * G1 + c] = inside ? fmaxf(acc[c], 0.f) : 0.f;
    }
    __syncthreads();
    // conv2 + BN + ReLU: wave = 6 output channels, lane = one of the 49 output pixels
    {
        const int pix = lane < R * R ? lane : R * R - 1;
        const int oy = pix / R, ox = pix - oy * R;
        float acc[G2];
#pragma unroll
        for (int c = 0; c < G2; ++c) acc[c] = p_w2[N_W2 + wave * G2 + c];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* in = &s_c1[((2 * oy + ky) * R1 + 2 * ox + kx) * C1];
#pragma unroll
                for (int ci = 0; ci < C1; ++ci) {
                    const float a = in[ci];
#pragma unroll
                    for (int c = 0; c < G2; ++c) acc[c] = fmaf(a, p_w2[(ci * 9 + ky * 3 + kx) * C2 + wave * G2 + c], acc[c]);
                }
            }
        if (lane < R * R) {
#pragma unroll
            for (int c = 0; c < G2; ++c) s_c2[pix * C2 + wave * G2 + c] = fmaxf(acc[c], 0.f);
        }
    }
    __syncthreads();
    // conv1x1 + bias: wave = 24 output channels, lane = token; LayerNorm statistics reduce across the 4 waves through LDS
    const int t = lane < R * R ? lane : R * R - 1;
    float v[G3];
#pragma unroll
    for (int k = 0; k < G3; ++k) v[k] = p_w3[N_W3 + wave * G3 + k];
#pragma unroll 4
    for (int ci = 0; ci < C2; ++ci) {
        const float a = s_c2[t * C2 + ci];
#pragma unroll
        for (int k = 0; k < G3; ++k) v[k] = fmaf(a, p_w3[ci * C0 + wave * G3 + k], v[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < G3; ++k) s += v[k];
    s_red[0][wave][lane] = s;
    __syncthreads();
    const float mean = (s_red[0][0][lane] + s_red[0][1][lane] + s_red[0][2][lane] + s_red[0][3][lane]) * (1.0f / C0);
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < G3; ++k) { const float d = v[k] - mean; ss = fmaf(d, d, ss); }
    s_red[1][wave][lane] = ss;
    __syncthreads();
    const float rstd = rsqrtf((s_red[1][0][lane] + s_red[1][1][lane] + s_red[1][2][lane] + s_red[1][3][lane]) * (1.0f / C0) + eps);
    if (lane < R * R) {
        const int oy = t / R, ox = t - oy * R;
        float* dst = x + ((size_t)b * G * G + (size_t)(R * by + oy) * G + R * bx + ox) * C0 + wave * G3;
#pragma unroll
        for (int k = 0; k < G3; k += 4) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf((v[k + e] - mean) * rstd, p_ln[wave * G3 + k + e], p_ln[C0 + wave * G3 + k + e]);
            *reinterpret_cast<f32x4*>(dst + k) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Window attention (ctranspath.py:510-547 inside :654-690).  A wave owns ONE HEAD for the whole launch and walks over
// (tile, window) pairs; per window: 49 tokens, head_dim 32.
//   S^T = K Q^T by MFMA 32x32x16 with the K and Q fragments loaded straight from the packed qkv rows (the 16-byte
//   fragment of a lane IS a contiguous piece of one token row) -> a lane owns one query column and 16 keys per tile;
//   s*scale*log2e + bias (dense per-lane table of the wave's head, rel-pos bias*log2e with -30000 on the 15 pad keys,
//   loaded ONCE into 64 registers) + shift mask (-100*log2e where the region labels differ: one 64-bit word per lane
//   and window type, bit = tile*16 + r) -> exp2 softmax (row reductions = 16 in-lane values + one cross-half shuffle)
//   -> P (already the MFMA B operand) times V^T staged through a wave-private LDS image.
//   Output: the two half-waves exchange 8-byte pieces so that every lane stores 16 contiguous bytes.
// qkv: [rows][ldq] act dtype, columns [q | k | v] x [head][32]; out: [rows][ldo], columns [head][32].
// bias_lane: [heads][kt 2][qt 2][lane 64][r 16] fp32;  mask_bits: [type 4][lane 64] u64 (type 0 = no mask).
// Measured steps on the one-(window,head)-per-wave first version (256 tiles, stage 1, 195 us): bias loads 59 us,
// 8-byte output stores 39 us, V^T staging 23 us, everything else 76 us.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) swin_wattn_kernel(const T* __restrict__ qkv, long ldq, T* __restrict__ out, long ldo,
                                                         const float* __restrict__ bias_lane, const unsigned long long* __restrict__ mask_bits,
                                                         int G, int C, int heads, int shift, float scale_l2, int nwin_total, int nslots) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    __shared__ __attribute__((aligned(16))) char smem[4 * SW_VT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wid = blockIdx.x * 4 + wave;
    const int h = wid % heads, slot = wid / heads;
    const int nwin_side = G / SW_WS, nW = nwin_side * nwin_side;
    char* vt = smem + wave * SW_VT_BYTES;
    // the head's bias table, in accumulator order
    f32x4 bl[4][4];
    {
        const float* bp = bias_lane + ((size_t)h * 4 * 64 + lane) * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) bl[t][g] = *reinterpret_cast<const f32x4*>(bp + (size_t)t * 64 * 16 + g * 4);
    }
    const int p0 = l31, p1 = min(32 + l31, SW_N - 1), pv = min(lane, SW_N - 1);
    const int i0 = p0 / SW_WS, j0 = p0 - i0 * SW_WS, i1 = p1 / SW_WS, j1 = p1 - i1 * SW_WS, iv = pv / SW_WS, jv = pv - iv * SW_WS;
    const bool kvalid = lane < SW_N;
    for (int step = 0, base = 0; base < nwin_total; ++step, base += nslots) {
        // every step covers the contiguous window range [base, base + nslots).  Shifted layers rotate the assignment inside
        // it by 37 per step so that the (slower) masked border windows do not always hit the same waves: measured at
        // 256 tiles, stage 1: 216 -> 202 us; un-shifted layers keep a fixed window position per wave (166 vs 185 us rotated)
        int idx = slot + (shift > 0 ? step * 37 : 0);
        idx -= (idx / nslots) * nslots;
        const int win = base + idx;
        if (win >= nwin_total) break;
        const int b = win / nW, wi = win - b * nW;
        const int wh = wi / nwin_side, ww = wi - wh * nwin_side;
        auto token_row = [&](int i, int j) -> long {
            int hh = wh * SW_WS + i + shift, wc = ww * SW_WS + j + shift;
            hh = hh >= G ? hh - G : hh;
            wc = wc >= G ? wc - G : wc;
            return (long)b * G * G + (long)hh * G + wc;
        };
        const long row0 = token_row(i0, j0), row1 = token_row(i1, j1);
        const T* r0p = qkv + row0 * ldq + h * SW_HD + hi * 8;
        const T* r1p = qkv + row1 * ldq + h * SW_HD + hi * 8;
        vec8 qf[2][2], kf[2][2], vv[4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[0][ks] = *reinterpret_cast<const vec8*>(r0p + ks * 16);
            qf[1][ks] = *reinterpret_cast<const vec8*>(r1p + ks * 16);
            kf[0][ks] = *reinterpret_cast<const vec8*>(r0p + C + ks * 16);
            kf[1][ks] = *reinterpret_cast<const vec8*>(r1p + C + ks * 16);
        }
        {
            const T* vp = qkv + token_row(iv, jv) * ldq + 2 * C + h * SW_HD;
#pragma unroll
            for (int c = 0; c < 4; ++c) vv[c] = *reinterpret_cast<const vec8*>(vp + c * 8);
        }
        const int wtype = shift > 0 ? ((wh == nwin_side - 1) ? 2 : 0) + ((ww == nwin_side - 1) ? 1 : 0) : 0;
        const unsigned long long mb = mask_bits[wtype * 64 + lane];      // unconditional (type 0 = zeros): a predicated load would
                                                                         // get a vmcnt(0) at its join and stall the q/k/v requests
        f32x16 s[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][qt][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) s[kt][qt] = Act<T>::mfma32(kf[kt][ks], qf[qt][ks], s[kt][qt]);
            }
        // V^T image (lane = key); the previous window's reads of it are complete (LDS is in order per wave)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<T*>(vt + (c * 8 + e) * SW_VS + lane * 2) = kvalid ? vv[c][e] : (T)0.f;
        float mx[2] = {-3.0e38f, -3.0e38f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int tile = kt * 2 + qt;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[kt][qt][g * 4 + e] = fmaf(s[kt][qt][g * 4 + e], scale_l2, bl[tile][g][e]);
                if (wtype) {
                    const unsigned bits = (unsigned)(mb >> (tile * 16)) & 0xffffu;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][qt][r] += ((bits >> r) & 1u) ? -144.26950408889634f : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx[qt] = fmaxf(mx[qt], s[kt][qt][r]);
            }
        float sum[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32, 64));
            float a = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kt][qt][r] - mx[qt]);
                    s[kt][qt][r] = p;
                    a += p;
                }
            sum[qt] = a + __shfl_xor(a, 32, 64);
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 o[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][r] = 0.f;
        const char* vr = vt + l31 * SW_VS + hi * 8;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const vec4 va = *reinterpret_cast<const vec4*>(vr + (32 * kt + 16 * s2) * 2);
                const vec4 vb = *reinterpret_cast<const vec4*>(vr + (32 * kt + 16 * s2 + 8) * 2);
                vec8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = va[e]; vf[4 + e] = vb[e]; }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    vec8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = Act<T>::from_f32(s[kt][qt][8 * s2 + e]);
                    o[qt] = Act<T>::mfma32(vf, pf, o[qt]);
                }
            }
        __builtin_amdgcn_wave_barrier();
        // lane (query l31, half hi) holds d = 8g + 4hi + (0..3).  Exchange so that hi = 0 owns d 0-7 and 16-23, hi = 1 owns
        // d 8-15 and 24-31: it sends the pieces the other half needs (g = 1,3 from hi = 0; g = 0,2 from hi = 1).
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float inv = 1.0f / sum[qt];
            u32x2 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = Act<T>::from_f32(o[qt][4 * g + e] * inv);
                pk[g] = *reinterpret_cast<u32x2*>(&ov);
            }
            u32x4 st[2];
#pragma unroll
            for (int pair = 0; pair < 2; ++pair) {
                const u32x2 keep = hi ? pk[2 * pair + 1] : pk[2 * pair];
                const u32x2 give = hi ? pk[2 * pair] : pk[2 * pair + 1];
                u32x2 got;
                got[0] = __shfl_xor(give[0], 32, 64);
                got[1] = __shfl_xor(give[1], 32, 64);
                st[pair] = hi ? u32x4{got[0], got[1], keep[0], keep[1]} : u32x4{keep[0], keep[1], got[0], got[1]};
            }
            if (qt * 32 + l31 < SW_N) {
                T* op = out + (qt ? row1 : row0) * ldo + h * SW_HD + 8 * hi;
                *reinterpret_cast<u32x4*>(op) = st[0];
                *reinterpret_cast<u32x4*>(op + 16) = st[1];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole attention branch of a 96-channel Swin block in ONE pass over the residual stream (ctranspath.py:654-692):
//     x[window rows] += proj( window_attention( qkv( LayerNorm(x[window rows]) ) ) )
// One wave owns one (tile, window) at a time (4 waves per workgroup, one workgroup per CU, 512 registers per lane);
// Wqkv, Wproj, the three heads' relative-position-bias tables, LayerNorm parameters and biases live in LDS (122 KB) for the
// whole launch.  Nothing but x is read or written: 2 x 384 B per token instead of the 2.9 KB of the unfused chain.
// Register-level dataflow (all operands of every MFMA come from registers or the stationary LDS images):
//   xf        = LayerNorm(x rows) as MFMA fragments (lane = token)                     [rowstream.h]
//   K, Q      = W(A) x xf(B)        -> (lane = token, registers = the head's 32 dims)  = A / B operands of S^T = K Q^T
//   V^T       = xf(A) x Wv(B)       -> (lane = dim, registers = tokens)                = A operand of P V, no LDS transpose
//   S^T       -> softmax (+ bias table, + shift mask bits) -> P in registers            = B operand of P V
//   O^T       = V^T(A) x P(B)       -> (lane = token, registers = dims)                = A operand of the projection
//   out      += O(A) x Wproj(B)     -> (lane = output channel, registers = token rows)  coalesced read-modify-write of x
// The contraction index of every chained MFMA pair is permuted consistently (accumulator register order = operand slot
// order), and Wproj is staged in LDS with that permutation baked in.
// ------------------------------------------------------------------------------------------------
constexpr int SA_WQKV = 9 * 6 * 1024, SA_WPROJ = 3 * 6 * 1024, SA_BIAS = 3 * 4 * 64 * 16 * 4;
constexpr int SA_LDS = SA_WQKV + SA_WPROJ + SA_BIAS + (2 * 96 + 288 + 96) * 4;

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
swin_attn96_kernel(float* x, const T* __restrict__ Wqkv, const float* __restrict__ bqkv, const T* __restrict__ Wproj,
                   const float* __restrict__ bproj, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                   const float* __restrict__ bias_lane, const unsigned long long* __restrict__ mask_bits, int G, int shift, float eps,
                   float scale_l2, int nwin_total) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int C = 96, KS = 6, NH = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_wqkv = smem;
    char* s_wproj = smem + SA_WQKV;
    float* s_bias = reinterpret_cast<float*>(s_wproj + SA_WPROJ);          // [head][tile][g][lane][4]
    float* s_ln = s_bias + SA_BIAS / 4;                                    // gamma[96] beta[96]
    float* s_bq = s_ln + 2 * C;                                            // qkv bias [288]
    float* s_bp = s_bq + 3 * C;                                            // proj bias [96]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    for (int blk = wave; blk < 9 * KS; blk += 4) {
        const int j = blk / KS, ks = blk - j * KS;
        glds16(Wqkv + (long)(32 * j + l31) * C + 16 * ks + 8 * hi, s_wqkv + blk * 1024);
    }
    for (int blk = wave; blk < 3 * 6; blk += 4) {                          // (cf, hs): B operand of the projection, k-permuted
        const int cf = blk / 6, hs = blk - cf * 6;
        const T* src = Wproj + (long)(32 * cf + l31) * C + 16 * hs + 4 * hi;
        const vec4 lo = *reinterpret_cast<const vec4*>(src), hi4 = *reinterpret_cast<const vec4*>(src + 8);
        vec8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi4[e]; }
        *reinterpret_cast<vec8*>(s_wproj + blk * 1024 + lane * 16) = v;
    }
    for (int i = tid; i < NH * 4 * 64 * 4; i += 256) {                     // bias table: [h][tile][lane][g*4..] -> [h][tile][g][lane][4]
        const int g4 = i & 3, ln = (i >> 2) & 63, ht = i >> 8;
        *reinterpret_cast<f32x4*>(s_bias + ((ht * 4 + g4) * 64 + ln) * 4) = *reinterpret_cast<const f32x4*>(bias_lane + ((size_t)ht * 64 + ln) * 16 + g4 * 4);
    }
    for (int i = tid; i < C; i += 256) { s_ln[i] = ln_g[i]; s_ln[C + i] = ln_b[i]; s_bp[i] = bproj[i]; }
    for (int i = tid; i < 3 * C; i += 256) s_bq[i] = bqkv[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nwin_side = G / SW_WS, nW = nwin_side * nwin_side;
    const int p0 = l31, p1 = min(32 + l31, SW_N - 1);
    const int i0 = p0 / SW_WS, j0 = p0 - i0 * SW_WS, i1 = p1 / SW_WS, j1 = p1 - i1 * SW_WS;
    const int wstep = gridDim.x * 4;
    int win = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    auto token_row = [&](int w, int i, int j) -> int {
        const int b = w / nW, wi = w - b * nW;
        const int wh = wi / nwin_side, ww = wi - wh * nwin_side;
        int hh = wh * SW_WS + i + shift, wc = ww * SW_WS + j + shift;
        hh = hh >= G ? hh - G : hh;
        wc = wc >= G ? wc - G : wc;
        return (b * G + hh) * G + wc;
    };
    f32x4 raw[2][KS][2];
    vec8 xf[2][KS];
    if (win < nwin_total) {
        rs_load_raw<KS>(raw[0], x, C, token_row(win, i0, j0), hi);
        rs_load_raw<KS>(raw[1], x, C, token_row(win, i1, j1), hi);
        rs_normalise<T, KS>(xf[0], raw[0], s_ln, hi, eps);
        rs_normalise<T, KS>(xf[1], raw[1], s_ln, hi, eps);
    }
    for (; win < nwin_total; win += wstep) {
        int lds_lane = lane * 16;                         // opaque per iteration: keeps the stationary-operand ds_reads inside the loop
        asm volatile("" : "+v"(lds_lane));
        const int wi = win % nW;
        const int wh = wi / nwin_side, ww = wi - wh * nwin_side;
        const int wtype = shift > 0 ? ((wh == nwin_side - 1) ? 2 : 0) + ((ww == nwin_side - 1) ? 1 : 0) : 0;
        const unsigned long long mb = mask_bits[wtype * 64 + lane];
        f32x16 acc_out[2][3];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int cf = 0; cf < 3; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_out[f][cf][r] = 0.f;
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
            // ---- K (A operand of S^T), Q (B operand), V^T (A operand of P V): 3 x 2 x 6 MFMAs ----
            vec8 kf[2][2], qf[2][2], vf[2][2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                f32x16 aq, ak, av;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 bq4 = *reinterpret_cast<const f32x4*>(s_bq + 32 * h + 8 * g4 + 4 * hi + (lds_lane & 1));
                    const f32x4 bk4 = *reinterpret_cast<const f32x4*>(s_bq + C + 32 * h + 8 * g4 + 4 * hi + (lds_lane & 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { aq[4 * g4 + e] = bq4[e]; ak[4 * g4 + e] = bk4[e]; }
                }
                const float bvl = s_bq[2 * C + 32 * h + l31 + (lds_lane & 1)];
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = bvl;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 wq = *reinterpret_cast<const vec8*>(s_wqkv + ((h)*KS + ks) * 1024 + lds_lane);
                    const vec8 wk = *reinterpret_cast<const vec8*>(s_wqkv + ((3 + h) * KS + ks) * 1024 + lds_lane);
                    const vec8 wv = *reinterpret_cast<const vec8*>(s_wqkv + ((6 + h) * KS + ks) * 1024 + lds_lane);
                    aq = Act<T>::mfma32(wq, xf[f][ks], aq);
                    ak = Act<T>::mfma32(wk, xf[f][ks], ak);
                    av = Act<T>::mfma32(xf[f][ks], wv, av);
                    if (ks & 1) __builtin_amdgcn_sched_barrier(0);          // at most 6 weight fragments in flight
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    qf[f][r >> 3][r & 7] = Act<T>::from_f32(aq[r]);
                    kf[f][r >> 3][r & 7] = Act<T>::from_f32(ak[r]);
                    vf[f][r >> 3][r & 7] = Act<T>::from_f32(av[r]);
                }
            }
            // ---- attention of this head, one 32-query half at a time ----
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x16 sT[2];
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sT[kt][r] = 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) sT[kt] = Act<T>::mfma32(kf[kt][s2], qf[qt][s2], sT[kt]);
                    const int tile = kt * 2 + qt;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_bias) + (((h * 4 + tile) * 4 + g4) * 64) * 16 + lds_lane);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sT[kt][4 * g4 + e] = fmaf(sT[kt][4 * g4 + e], scale_l2, bv[e]);
                    }
                    if (wtype) {
                        const unsigned bits = (unsigned)(mb >> (tile * 16)) & 0xffffu;
#pragma unroll
                        for (int r = 0; r < 16; ++r) sT[kt][r] += ((bits >> r) & 1u) ? -144.26950408889634f : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sT[kt][r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pexp = __builtin_amdgcn_exp2f(sT[kt][r] - mx);
                        sT[kt][r] = pexp;
                        sum += pexp;
                    }
                sum += __shfl_xor(sum, 32, 64);
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        vec8 pf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[e] = Act<T>::from_f32(sT[kt][8 * s2 + e]);
                        o = Act<T>::mfma32(vf[kt][s2], pf, o);
                    }
                const float inv = 1.0f / sum;
                vec8 of[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) of[r >> 3][r & 7] = Act<T>::from_f32(o[r] * inv);
                // ---- projection: this head's 32 input channels = two k-steps ----
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int cf = 0; cf < 3; ++cf)
                        acc_out[qt][cf] = Act<T>::mfma32(of[s2], *reinterpret_cast<const vec8*>(s_wproj + (cf * 6 + 2 * h + s2) * 1024 + lds_lane), acc_out[qt][cf]);
            }
        }
        const bool more = win + wstep < nwin_total;
        // ---- x rows += result + bias: register r of (half f) is token p = 32 f + (r&3) + 8 (r>>2) + 4 hi; lanes = channels.
        // (Starting the accumulators from the residual rows instead, as the MLP kernel does, was measured: same time, 29 spills.)
        const int b = win / nW;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * f + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int pc = min(p, SW_N - 1);
                const int i = pc / SW_WS, j = pc - i * SW_WS;
                int hh = wh * SW_WS + i + shift, wc = ww * SW_WS + j + shift;
                hh = hh >= G ? hh - G : hh;
                wc = wc >= G ? wc - G : wc;
                float* row = x + ((long)(b * G + hh) * G + wc) * C + l31;
                if (f == 0 || (r & 3) + 8 * (r >> 2) < 17) {                 // compile-time: registers that can hold a token < 49
                    float v0 = row[0], v1 = row[32], v2 = row[64];             // unconditional loads (clamped row), predicated stores
                    v0 += acc_out[f][0][r] + s_bp[l31];
                    v1 += acc_out[f][1][r] + s_bp[32 + l31];
                    v2 += acc_out[f][2][r] + s_bp[64 + l31];
                    if (p < SW_N) { row[0] = v0; row[32] = v1; row[64] = v2; }
                }
            }
        if (more) {
            rs_load_raw<KS>(raw[0], x, C, token_row(win + wstep, i0, j0), hi);
            rs_load_raw<KS>(raw[1], x, C, token_row(win + wstep, i1, j1), hi);
            rs_normalise<T, KS>(xf[0], raw[0], s_ln + (lds_lane & 1), hi, eps);
            rs_normalise<T, KS>(xf[1], raw[1], s_ln + (lds_lane & 1), hi, eps);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PatchMerging gather + LayerNorm(4C) (ctranspath.py:717-736).  One wave per output row; the four members of a 2x2
// cell are concatenated in the reference's order (0,0),(1,0),(0,1),(1,1) [dh = k&1, dw = k>>1].
// ------------------------------------------------------------------------------------------------
template <typename T, int NV>
__global__ void __launch_bounds__(256) swin_merge_ln_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ gw, const float* __restrict__ gb,
                                                            int G, int C, int rows_out, float eps) {
    typedef typename Act<T>::vec4 vec4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows_out) return;
    const int G2 = G / 2, cv = C / 4;            // float4s per member
    const int b = r / (G2 * G2), cell = r - b * G2 * G2, h2 = cell / G2, w2 = cell - h2 * G2;
    f32x4 buf[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        buf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (v < C) {
            const int k = v / cv, off = v - k * cv;
            const long src = (long)b * G * G + (long)(2 * h2 + (k & 1)) * G + 2 * w2 + (k >> 1);
            buf[i] = *reinterpret_cast<const f32x4*>(x + src * C + off * 4);
            s += buf[i][0] + buf[i][1] + buf[i][2] + buf[i][3];
        }
    }
    const float inv_n = 1.0f / (4 * C);
    const float mean = wave_sum(s) * inv_n;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < C)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = buf[i][e] - mean; ss = fmaf(d, d, ss); }
    const float rstd = rsqrtf(wave_sum(ss) * inv_n + eps);
    T* yr = y + (long)r * 4 * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < C) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gw + v * 4), bb = *reinterpret_cast<const f32x4*>(gb + v * 4);
            vec4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Act<T>::from_f32(fmaf((buf[i][e] - mean) * rstd, g[e], bb[e]));
            *reinterpret_cast<vec4*>(yr + v * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Final LayerNorm + mean over tokens (ctranspath.py:981-984): mean_t(g*n_t + b) = g*mean_t(n_t) + b.
// One workgroup per tile; lane owns columns lane + 64*i.
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ void __launch_bounds__(256) swin_norm_pool_kernel(const float* __restrict__ x, f16* __restrict__ out16,
                                                             float* __restrict__ out32, const float* __restrict__ gw,
                                                             const float* __restrict__ gb, int L, int C, float eps) {
    __shared__ float red[4][NC * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    float acc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = 0.f;
    const float inv_c = 1.0f / C;
    for (int t = wave; t < L; t += 4) {
        const float* xr = x + ((long)b * L + t) * C;
        float v[NC], s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) { v[i] = (lane + 64 * i < C) ? xr[lane + 64 * i] : 0.f; s += v[i]; }
        const float mean = wave_sum(s) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) { const float d = (lane + 64 * i < C) ? v[i] - mean : 0.f; ss = fmaf(d, d, ss); }
        const float rstd = rsqrtf(wave_sum(ss) * inv_c + eps);
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = fmaf(v[i] - mean, rstd, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) red[wave][i * 64 + lane] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = (red[0][c] + red[1][c] + red[2][c] + red[3][c]) * (1.0f / L);
        const float o = fmaf(m, gw[c], gb[c]);
        if (out16) out16[(long)b * C + c] = (f16)o;
        if (out32) out32[(long)b * C + c] = o;
    }
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct SwinPlan {
    int G, C0, nstage, ldh0;
    size_t off_xa, off_xb, off_h, off_big, total;
};

static int swin_plan(const amds_swin_cfg* c, int batch, SwinPlan* p) {
    AMDS_REQUIRE(c != nullptr, "swin: null cfg");
    AMDS_REQUIRE(c->embed == 96, "swin: embed=%d unsupported (the stem and attention kernels are built for 96 = 3 heads x 32)", c->embed);
    AMDS_REQUIRE(c->n_stages >= 1 && c->n_stages <= 4, "swin: n_stages=%d must be 1..4", c->n_stages);
    AMDS_REQUIRE(c->dtype == AMDS_F16 || c->dtype == AMDS_BF16, "swin: bad act dtype");
    AMDS_REQUIRE(batch > 0, "swin: bad batch");
    const int div = 4 * SW_WS << (c->n_stages - 1);
    AMDS_REQUIRE(c->img > 0 && c->img % div == 0, "swin: img=%d must be a multiple of %d", c->img, div);
    for (int s = 0; s < c->n_stages; ++s) {
        AMDS_REQUIRE(c->depths[s] > 0, "swin: depths[%d]=%d", s, c->depths[s]);
        AMDS_REQUIRE(c->heads[s] * SW_HD == (c->embed << s), "swin: heads[%d]=%d must be dim/32", s, c->heads[s]);
    }
    p->G = c->img / 4; p->C0 = c->embed; p->nstage = c->n_stages;
    p->ldh0 = c->embed;
    const size_t rows = (size_t)batch * p->G * p->G;
    size_t o = 0;
    p->off_xa = o;  o += align256(rows * c->embed * 4);
    p->off_xb = o;  o += align256(rows / 4 * 2 * c->embed * 4);
    p->off_h = o;   o += align256(rows * p->ldh0 * 2);
    p->off_big = o; o += align256(rows * 4 * c->embed * 2);
    p->total = o;
    return AMDS_OK;
}

template <typename T>
static int launch_wattn(const void* qkv, long ldq, void* out, long ldo, const float* bias_lane, const unsigned long long* mask_bits,
                        int B, int G, int C, int heads, int shift, hipStream_t st) {
    const int nW = (G / SW_WS) * (G / SW_WS);
    const long nwin = (long)B * nW;
    AMDS_REQUIRE(nwin * heads < (1L << 31), "window attention: too many windows");
    const float scale_l2 = 0.17677669529663687f * 1.4426950408889634f;    // 32^-0.5 * log2(e)
    // persistent waves: every wave keeps one head's bias table in registers; ~12 waves per CU
    int nslots = (256 * 12) / heads;
    if (nslots > nwin) nslots = (int)nwin;
    nslots = (nslots + 3) & ~3;                                            // whole workgroups of 4 waves
    ProfScope prof(PROF_ATTN, 4.0 * nwin * heads * SW_N * SW_N * SW_HD, st);
    hipLaunchKernelGGL((swin_wattn_kernel<T>), dim3(nslots * heads / 4), dim3(256), 0, st, (const T*)qkv, ldq, (T*)out, ldo,
                       bias_lane, mask_bits, G, C, heads, shift, scale_l2, (int)nwin, nslots);
    AMDS_LAUNCH_CHECK("swin_wattn_kernel");
    return AMDS_OK;
}

}  // namespace amds

using namespace amds;

extern "C" int amds_swin_stem(const uint8_t* tiles, float* x, const float* params, int B, int img, int embed, float eps,
                              void* stream) {
    AMDS_REQUIRE(tiles && x && params, "amds_swin_stem: null pointer");
    AMDS_REQUIRE(embed == 96, "amds_swin_stem: embed=%d unsupported (96 only)", embed);
    AMDS_REQUIRE(B >= 0 && img > 0 && img % 28 == 0, "amds_swin_stem: img=%d must be a multiple of 28", img);
    if (B == 0) return AMDS_OK;
    const int G = img / 4;
    hipStream_t st = (hipStream_t)stream;
    const double fl = 2.0 * B * ((double)(img / 2) * (img / 2) * 27 * 12 + (double)G * G * (108 * 24 + 24 * 96));
    ProfScope prof(PROF_OTHER, fl, st);
    hipLaunchKernelGGL((swin_stem_kernel<96>), dim3(G / 7, G / 7, B), dim3(256), 0, st, tiles, x, params, img, G, eps);
    AMDS_LAUNCH_CHECK("swin_stem_kernel");
    return AMDS_OK;
}

extern "C" int amds_window_attention(const void* qkv, long ldq, void* out, long ldo, const float* bias_lane,
                                     const uint64_t* mask_bits, int B, int grid, int dim, int heads, int shift, int dtype,
                                     void* stream) {
    AMDS_REQUIRE(qkv && out && bias_lane && mask_bits, "amds_window_attention: null pointer");
    AMDS_REQUIRE(grid > 0 && grid % SW_WS == 0, "amds_window_attention: grid=%d must be a multiple of 7", grid);
    AMDS_REQUIRE(heads * SW_HD == dim, "amds_window_attention: dim=%d must be heads*32", dim);
    AMDS_REQUIRE(shift >= 0 && shift < SW_WS && (shift == 0 || grid > SW_WS), "amds_window_attention: bad shift=%d for grid=%d", shift, grid);
    AMDS_REQUIRE(ldq >= 3L * dim && ldq % 8 == 0 && ldo >= dim && ldo % 8 == 0, "amds_window_attention: bad strides");
    AMDS_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "amds_window_attention: misaligned pointers");
    if (B <= 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AMDS_F16) return launch_wattn<f16>(qkv, ldq, out, ldo, bias_lane, reinterpret_cast<const unsigned long long*>(mask_bits), B, grid, dim, heads, shift, st);
    if (dtype == AMDS_BF16) return launch_wattn<bf16>(qkv, ldq, out, ldo, bias_lane, reinterpret_cast<const unsigned long long*>(mask_bits), B, grid, dim, heads, shift, st);
    set_error("amds_window_attention: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

extern "C" int amds_swin_attn96(float* x, const void* qkv_w, const float* qkv_b, const void* proj_w, const float* proj_b,
                               const float* ln_gamma, const float* ln_beta, const float* bias_lane, const uint64_t* mask_bits, int B,
                               int grid, int shift, float ln_eps, int dtype, void* stream) {
    AMDS_REQUIRE(x && qkv_w && qkv_b && proj_w && proj_b && ln_gamma && ln_beta && bias_lane && mask_bits, "amds_swin_attn96: null pointer");
    AMDS_REQUIRE(grid > 0 && grid % SW_WS == 0, "amds_swin_attn96: grid=%d must be a multiple of 7", grid);
    AMDS_REQUIRE(shift >= 0 && shift < SW_WS && (shift == 0 || grid > SW_WS), "amds_swin_attn96: bad shift=%d for grid=%d", shift, grid);
    AMDS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)qkv_w & 15) == 0 && ((uintptr_t)proj_w & 7) == 0, "amds_swin_attn96: misaligned pointers");
    if (B <= 0) return AMDS_OK;
    const long nwin = (long)B * (grid / SW_WS) * (grid / SW_WS);
    AMDS_REQUIRE(nwin < (1L << 30) && (long)B * grid * grid * 96 < (1L << 31), "amds_swin_attn96: batch too large for 32-bit token indexing");
    hipStream_t st = (hipStream_t)stream;
    const float scale_l2 = 0.17677669529663687f * 1.4426950408889634f;
    int gx = (int)((nwin + 3) / 4);
    if (gx > 256) gx = 256;
    ProfScope prof(PROF_ATTN, nwin * (2.0 * 49 * 96 * 384 + 4.0 * 3 * 49 * 49 * 32), st);
#define ATTN96_LAUNCH(T)                                                                                                              \
    do {                                                                                                                              \
        static bool attr_set = false;                                                                                                 \
        if (!attr_set) {                                                                                                              \
            AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(swin_attn96_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, SA_LDS)); \
            attr_set = true;                                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((swin_attn96_kernel<T>), dim3(gx), dim3(256), SA_LDS, st, x, reinterpret_cast<const T*>(qkv_w), qkv_b,     \
                           reinterpret_cast<const T*>(proj_w), proj_b, ln_gamma, ln_beta, bias_lane,                                  \
                           reinterpret_cast<const unsigned long long*>(mask_bits), grid, shift, ln_eps, scale_l2, (int)nwin);         \
    } while (0)
    if (dtype == AMDS_F16) ATTN96_LAUNCH(f16);
    else if (dtype == AMDS_BF16) ATTN96_LAUNCH(bf16);
    else { set_error("amds_swin_attn96: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
#undef ATTN96_LAUNCH
    AMDS_LAUNCH_CHECK("swin_attn96_kernel");
    return AMDS_OK;
}

extern "C" int amds_patch_merge_ln(const float* x, void* y, const float* gamma, const float* beta, int B, int grid, int dim,
                                   float eps, int dtype, void* stream) {
    AMDS_REQUIRE(x && y && gamma && beta, "amds_patch_merge_ln: null pointer");
    AMDS_REQUIRE(grid > 0 && grid % 2 == 0 && dim % 4 == 0 && dim <= 384, "amds_patch_merge_ln: grid=%d dim=%d unsupported", grid, dim);
    if (B <= 0) return AMDS_OK;
    const int rows = B * (grid / 2) * (grid / 2);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)rows * 4 * dim * 6, st);
#define MERGE_LAUNCH(T)                                                                                                   \
    do {                                                                                                                  \
        if (dim <= 128) hipLaunchKernelGGL((swin_merge_ln_kernel<T, 2>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, (T*)y, gamma, beta, grid, dim, rows, eps); \
        else if (dim <= 192) hipLaunchKernelGGL((swin_merge_ln_kernel<T, 3>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, (T*)y, gamma, beta, grid, dim, rows, eps); \
        else hipLaunchKernelGGL((swin_merge_ln_kernel<T, 6>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, (T*)y, gamma, beta, grid, dim, rows, eps); \
    } while (0)
    if (dtype == AMDS_F16) MERGE_LAUNCH(f16);
    else if (dtype == AMDS_BF16) MERGE_LAUNCH(bf16);
    else { set_error("amds_patch_merge_ln: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
#undef MERGE_LAUNCH
    AMDS_LAUNCH_CHECK("swin_merge_ln_kernel");
    return AMDS_OK;
}

extern "C" int amds_layernorm_meanpool(const float* x, void* out_f16, float* out_f32, const float* gamma, const float* beta,
                                       int B, int L, int dim, float eps, void* stream) {
    AMDS_REQUIRE(x && gamma && beta && (out_f16 || out_f32), "amds_layernorm_meanpool: null pointer");
    AMDS_REQUIRE(L > 0 && dim > 0 && dim <= 768, "amds_layernorm_meanpool: L=%d dim=%d unsupported (dim <= 768)", L, dim);
    if (B <= 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)B * L * dim * 4, st);
    if (dim <= 192)
        hipLaunchKernelGGL((swin_norm_pool_kernel<3>), dim3(B), dim3(256), 0, st, x, (f16*)out_f16, out_f32, gamma, beta, L, dim, eps);
    else
        hipLaunchKernelGGL((swin_norm_pool_kernel<12>), dim3(B), dim3(256), 0, st, x, (f16*)out_f16, out_f32, gamma, beta, L, dim, eps);
    AMDS_LAUNCH_CHECK("swin_norm_pool_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_swin_workspace_bytes(const amds_swin_cfg* cfg_host, int batch) {
    SwinPlan p;
    if (swin_plan(cfg_host, batch, &p) != AMDS_OK) return 0;
    return p.total;
}

static int swin_chunk(const amds_swin_cfg* c, const amds_swin_weights* w, const SwinPlan& pl, const uint8_t* tiles,
                      void* feats_f16, float* feats_f32, int Bc, char* ws, hipStream_t st) {
    float* xa = reinterpret_cast<float*>(ws + pl.off_xa);
    float* xb = reinterpret_cast<float*>(ws + pl.off_xb);
    void* h = ws + pl.off_h;
    void* big = ws + pl.off_big;
    const int dt = c->dtype;
    int rc;
#define AMDS_TRY(call) do { rc = (call); if (rc != AMDS_OK) return rc; } while (0)
    AMDS_TRY(amds_swin_stem(tiles, xa, w->stem, Bc, c->img, c->embed, c->ln_eps, st));
    float* x = xa;
    float* xo = xb;
    int G = pl.G, blk = 0;
    for (int s = 0; s < c->n_stages; ++s) {
        const int C = c->embed << s, M = Bc * G * G;
        // C = 96 / 192: weights-stationary streaming GEMMs with LayerNorm fused into the operand load (gemm_rowstream.hip);
        // wider stages: tiled MFMA GEMMs + stand-alone LayerNorm
        const bool narrow = C <= 192;
        for (int d = 0; d < c->depths[s]; ++d, ++blk) {
            const amds_swin_block& b = w->blocks_host[blk];
            const int shift = (d % 2 == 1 && G > SW_WS) ? SW_WS / 2 : 0;
            static const bool fuse_attn = getenv("AMDS_SWIN_FUSE_ATTN") ? atoi(getenv("AMDS_SWIN_FUSE_ATTN")) != 0 : true;
            if (narrow) {
                if (C == 96 && fuse_attn) {     // LN1 + qkv + window attention + proj + residual in one pass over x
                    AMDS_TRY(amds_swin_attn96(x, b.qkv_w, b.qkv_b, b.proj_w, b.proj_b, b.ln1_w, b.ln1_b, b.bias_lane, w->mask_bits, Bc, G, shift, c->ln_eps, dt, st));
                } else {
                    AMDS_TRY(amds_gemm_rowstream(x, C, b.ln1_w, b.ln1_b, c->ln_eps, b.qkv_w, C, M, 3 * C, C, dt, AMDS_EPI_BIAS, big, 3 * C, b.qkv_b, st));
                    AMDS_TRY(amds_window_attention(big, 3 * C, h, C, b.bias_lane, w->mask_bits, Bc, G, C, c->heads[s], shift, dt, st));
                    AMDS_TRY(amds_gemm_rowstream(h, C, nullptr, nullptr, 0.f, b.proj_w, C, M, C, C, dt, AMDS_EPI_RESIDUAL, x, C, b.proj_b, st));
                }
                if (C == 96) {        // LN2 + fc1 + GELU + fc2 + residual in one pass, hidden activation in registers only
                    AMDS_TRY(amds_swin_mlp96(x, M, b.fc1_w, b.fc1_b, b.fc2_w, b.fc2_b, b.ln2_w, b.ln2_b, c->ln_eps, dt, st));
                } else if (C == 192 && b.mlp_pack) {     // same fusion, weights streamed through LDS
                    AMDS_TRY(amds_swin_mlp192(x, M, b.mlp_pack, b.fc1_b, b.fc2_b, b.ln2_w, b.ln2_b, c->ln_eps, dt, st));
                } else {
                    AMDS_TRY(amds_gemm_rowstream(x, C, b.ln2_w, b.ln2_b, c->ln_eps, b.fc1_w, C, M, 4 * C, C, dt, AMDS_EPI_BIAS_GELU, big, 4 * C, b.fc1_b, st));
                    AMDS_TRY(amds_gemm(big, 4 * C, b.fc2_w, 4 * C, M, C, 4 * C, dt, AMDS_EPI_RESIDUAL, x, C, b.fc2_b, nullptr, nullptr, 0, 0, 0, 1.0f, st));
                }
            } else {
                AMDS_TRY(amds_layernorm(x, C, b.ln1_w, b.ln1_b, h, C, M, C, c->ln_eps, dt, st));
                AMDS_TRY(amds_gemm(h, C, b.qkv_w, C, M, 3 * C, C, dt, AMDS_EPI_BIAS, big, 3 * C, b.qkv_b, nullptr, nullptr, 0, 0, 0, 1.0f, st));
                AMDS_TRY(amds_window_attention(big, 3 * C, h, C, b.bias_lane, w->mask_bits, Bc, G, C, c->heads[s], shift, dt, st));
                AMDS_TRY(amds_gemm(h, C, b.proj_w, C, M, C, C, dt, AMDS_EPI_RESIDUAL, x, C, b.proj_b, nullptr, nullptr, 0, 0, 0, 1.0f, st));
                AMDS_TRY(amds_layernorm(x, C, b.ln2_w, b.ln2_b, h, C, M, C, c->ln_eps, dt, st));
                AMDS_TRY(amds_gemm(h, C, b.fc1_w, C, M, 4 * C, C, dt, AMDS_EPI_BIAS_GELU, big, 4 * C, b.fc1_b, nullptr, nullptr, 0, 0, 0, 1.0f, st));
                AMDS_TRY(amds_gemm(big, 4 * C, b.fc2_w, 4 * C, M, C, 4 * C, dt, AMDS_EPI_RESIDUAL, x, C, b.fc2_b, nullptr, nullptr, 0, 0, 0, 1.0f, st));
            }
        }
        if (s + 1 < c->n_stages) {
            const amds_swin_merge& m = w->merges[s];
            AMDS_TRY(amds_patch_merge_ln(x, h, m.ln_w, m.ln_b, Bc, G, C, c->ln_eps, dt, st));
            if (4 * C == 384)
                AMDS_TRY(amds_gemm_rowstream(h, 4 * C, nullptr, nullptr, 0.f, m.red_w, 4 * C, M / 4, 2 * C, 4 * C, dt, AMDS_EPI_BIAS_F32, xo, 2 * C, nullptr, st));
            else
                AMDS_TRY(amds_gemm(h, 4 * C, m.red_w, 4 * C, M / 4, 2 * C, 4 * C, dt, AMDS_EPI_BIAS_F32, xo, 2 * C, nullptr, nullptr, nullptr, 0, 0, 0, 1.0f, st));
            float* t = x; x = xo; xo = t;
            G /= 2;
        }
    }
    AMDS_TRY(amds_layernorm_meanpool(x, feats_f16, feats_f32, w->norm_w, w->norm_b, Bc, G * G, c->embed << (c->n_stages - 1), c->ln_eps, st));
#undef AMDS_TRY
    return AMDS_OK;
}

extern "C" int amds_swin_forward(const amds_swin_cfg* cfg_host, const amds_swin_weights* w_host, const uint8_t* tiles,
                                 void* feats_f16, float* feats_f32, int B, int chunk, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(cfg_host && w_host && tiles && (feats_f16 || feats_f32) && ws, "amds_swin_forward: null pointer");
    AMDS_REQUIRE(B >= 0 && chunk > 0, "amds_swin_forward: bad B=%d chunk=%d", B, chunk);
    AMDS_REQUIRE(w_host->stem && w_host->blocks_host && w_host->norm_w && w_host->norm_b && w_host->mask_bits, "amds_swin_forward: incomplete weights");
    SwinPlan pl;
    int rc = swin_plan(cfg_host, chunk, &pl);
    if (rc != AMDS_OK) return rc;
    if (ws_bytes < pl.total) {
        set_error("amds_swin_forward: workspace %zu < required %zu bytes", ws_bytes, pl.total);
        return AMDS_ERR_WORKSPACE;
    }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_swin_forward: workspace must be 256-byte aligned");
    int nblk = 0;
    for (int s = 0; s < cfg_host->n_stages; ++s) nblk += cfg_host->depths[s];
    AMDS_REQUIRE(w_host->n_blocks == nblk, "amds_swin_forward: %d blocks given, cfg needs %d", w_host->n_blocks, nblk);
    const size_t tile_bytes = (size_t)cfg_host->img * cfg_host->img * 3;
    const int Cl = cfg_host->embed << (cfg_host->n_stages - 1);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = (B - b0 < chunk) ? B - b0 : chunk;
        rc = swin_chunk(cfg_host, w_host, pl, tiles + (size_t)b0 * tile_bytes,
                        feats_f16 ? reinterpret_cast<char*>(feats_f16) + (size_t)b0 * Cl * 2 : nullptr,
                        feats_f32 ? feats_f32 + (size_t)b0 * Cl : nullptr, bc, reinterpret_cast<char*>(ws), (hipStream_t)stream);
        if (rc != AMDS_OK) return rc;
    }
    return AMDS_OK;
}
