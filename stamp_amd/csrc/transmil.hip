// transmil.hip -- building blocks of the TransMIL head, all in fp32 like the reference:
// reference src/stamp/modeling/models/trans_mil.py -- NystromAttention.forward :81-163, moore_penrose_iter_pinv :23-37,
// PPEG.forward :274-283.  The Nystrom pseudo-inverse iteration is numerically touchy (6 cubic iterations on softmax
// matrices), so by default every matmul here runs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain);
// under torch.set_float32_matmul_precision("high") -- which the reference sets before training, train.py:519 -- the tiled
// products take fp32 operands as hi + lo bf16 and three bf16 MFMAs each (amds_set_matmul_precision, bgemm_x3_kernel below).
//   amds_bgemm_f32        C[z] = diag*I + alpha * A[z] op(B[z])   batched over (outer, inner) with separate strides,
//                         op(B) = B^T (B stored [N][K], "x y^T" products) or B (stored [K][N])
//   amds_softmax_rows     in-place row softmax, any row length
//   amds_landmark_mean    segment sums of l consecutive tokens divided by l  (:114-124)
//   amds_pinv_init        z0 = x^T / (max_j sum_i |x_ij| * max_i sum_j |x_ij|), maxima over ALL batches and heads (:26-28)
//   amds_dwconv_seq       depth-wise (per head) 33-tap convolution along the sequence, added in place (:150-151)
//   amds_ppeg             x + dw7x7(x) + dw5x5(x) + dw3x3(x) on the sqrt(T) x sqrt(T) token grid (:274-283)
#include "common.h"
#include <atomic>

namespace amds {

template <int TRANSB>
__global__ void __launch_bounds__(256) bgemm_f32_kernel(const float* __restrict__ A, int lda, long sAo, long sAi,
                                                        const float* __restrict__ B, int ldb, long sBo, long sBi,
                                                        float* __restrict__ Cm, int ldc, long sCo, long sCi, int inner,
                                                        int M, int N, int K, float alpha, float diag,
                                                        const float* __restrict__ bias, int accumulate) {
    constexpr int BK = 32, LDT = BK + 1;
    __shared__ float sA[64 * LDT];
    __shared__ float sB[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int zo = blockIdx.z / inner, zi = blockIdx.z - zo * inner;
    A += zo * sAo + zi * sAi;
    B += zo * sBo + zi * sBi;
    Cm += zo * sCo + zi * sCi;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = it * 256 + tid, row = c >> 3, c4 = (c & 7) * 4;
            const int gm = m0 + row;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + c4 + e;
                sA[row * LDT + c4 + e] = (gm < M && k < K) ? A[(long)gm * lda + k] : 0.f;
            }
        }
        if (TRANSB) {     // B stored [N][K]
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = it * 256 + tid, row = c >> 3, c4 = (c & 7) * 4;
                const int gn = n0 + row;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + c4 + e;
                    sB[row * LDT + c4 + e] = (gn < N && k < K) ? B[(long)gn * ldb + k] : 0.f;
                }
            }
        } else {          // B stored [K][N]: coalesced along n, transposed into the [n][k] LDS image
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = it * 256 + tid, k = c >> 6, n = c & 63;
                const int gk = k0 + k, gn = n0 + n;
                sB[n * LDT + k] = (gk < K && gn < N) ? B[(long)gk * ldb + gn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = sA[(wm * 32 + l31) * LDT + kk + hi];
            const float b = sB[(wn * 32 + l31) * LDT + kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < M) {
                float v = alpha * acc[r] + (m == n ? diag : 0.f) + (bias ? bias[n] : 0.f);
                if (accumulate) v += Cm[(long)m * ldc + n];
                Cm[(long)m * ldc + n] = v;
            }
        }
    }
}

// second output of the same product (amds_bgemm_f32_dual): C2 = alpha2 * A op(B) + diag2 * I, laid out like C
struct BgDual { float* c2; float alpha2, diag2; };

// 128 x 128 tile variant for the large batched products (Nystrom pinv iterations 256^3, q k_l^T, projections): each wave owns
// 64 x 64 = 2 x 2 MFMA fragments, so one LDS operand read feeds two MFMAs; the LDS image is [row][k parity][k / 2] so that the
// 8 values a lane needs per 16-deep k-tile (k = hi, hi + 2, ...) are two ds_read_b128; global loads are float4 and the next
// k-tile is fetched into registers while the current one is multiplied.  Requires K, lda, ldb multiples of 4 and 16-byte
// aligned operands; the 64 x 64 kernel above remains the fallback.
// WM x WN = 4 waves: 2 x 2 -> 128 x 128 tile; 4 x 1 -> 256 x 64 (products with N <= 64: the per-head d = 64 outputs would leave half of
// a 128-wide tile empty); 1 x 4 -> 64 x 256 (M <= 64).
// X3 = 1 (amds_set_matmul_precision(AMDS_MATMUL_HIGH)): every fp32 operand value as the sum of two bf16 numbers, the product as three bf16 MFMAs
// (hi hi + hi lo + lo hi, fp32 accumulate: ~16 mantissa bits) -- what `torch.set_float32_matmul_precision("high")` names ("treat each float32 number as the
// sum of two bfloat16 numbers"), which the reference sets before every training run (src/stamp/modeling/train.py:519; deploy.py:398 asks for "medium").
// Same loaders, same fp32 LDS images; a fragment (8 k values of a row) is split in registers on its way to the matrix pipe: 12 MFMAs of 32 cycles per 16-deep
// K step and wave instead of 32 of 64 cycles (v_mfma_f32_32x32x2_f32).
__device__ __forceinline__ void split_bf16x2(const f32x4 (&f)[2], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = f[e >> 2][e & 3];
        const bf16 h = (bf16)x;
        hi[e] = h;
        lo[e] = (bf16)(x - (float)h);
    }
}
template <int TRANSB, int TRANSA = 0, int WM = 2, int WN = 2, int X3 = 0, bool DUAL = false>
__global__ void __launch_bounds__(256) bgemm_f32_big_kernel(const float* __restrict__ A, int lda, long sAo, long sAi,
                                                            const float* __restrict__ B, int ldb, long sBo, long sBi,
                                                            float* __restrict__ Cm, int ldc, long sCo, long sCi, int inner,
                                                            int M, int N, int K, float alpha, float diag,
                                                            const float* __restrict__ bias, int accumulate, int vec, int xcd, BgDual dual) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 16, LDT = 20;
    constexpr int ITA = BM * 4 / 256, ITB = BN * 4 / 256;          // float4 pieces per thread and K step
    // LDS image of an operand tile.  Row-major in k in memory: [row][k parity][k / 2] (pitch LDT), two ds_read_b128 per fragment row.
    // k-major in memory ([K][rows]: A^T products, B of A B): [k][row] (pitch rows + 4) -- the float4 along the rows is stored as it came
    // (ds_write_b128, conflict-free) and a fragment row is eight ds_read_b32 of 32 consecutive rows.  (Scattering that float4 into the
    // [row][k] image put 32 lanes on 2 banks: 80 % of the LDS cycles of the kernel were bank conflicts.)
    constexpr int TRA = TRANSA, TRB = TRANSB ? 0 : 1;
    constexpr int LKA = BM + 4, LKB = BN + 4;
    __shared__ __attribute__((aligned(16))) float sA[TRA ? BK * LKA : BM * LDT];
    __shared__ __attribute__((aligned(16))) float sB[TRB ? BK * LKB : BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // Workgroup b runs on XCD b % 8, each with its own 4 MB L2: in launch order the tiles that share an A row panel or a B column panel
    // (neighbours in x / y) land on eight different L2s and every one of them fetches its own copy.  xcd: XCD x takes the x-th contiguous
    // eighth of the tile list instead (n fastest, then m, then batch), so the ~96 workgroups resident on an XCD walk the same panels together.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd) {
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
        const int L = bx + gx * (by + gy * bz), q = total >> 3, r = total & 7, x = L & 7;
        const int Lp = x * q + min(x, r) + (L >> 3);
        bx = Lp % gx;
        const int t = Lp / gx;
        by = t % gy;
        bz = t / gy;
    }
    const int zo = bz / inner, zi = bz - zo * inner;
    A += zo * sAo + zi * sAi;
    B += zo * sBo + zi * sBi;
    Cm += zo * sCo + zi * sCi;
    if constexpr (DUAL) dual.c2 += zo * sCo + zi * sCi;
    const int m0 = by * BM, n0 = bx * BN;
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[ITA], rb[ITB];
    // one operand tile (R rows of the output dimension x 16 k) -> registers.  Row-major in k (TR = 0: float4 along k, 4 pieces per
    // row) or k-major (TR = 1: the operand is stored [K][R], float4 along the row dimension, R / 4 pieces per k).
    auto gload_op = [&](auto fast_c, auto tr_c, auto it_c, f32x4* r, const float* P, int ld, int R, int r0, int k0, int RT) {
        constexpr int TR = decltype(tr_c)::value, IT = decltype(it_c)::value;
        constexpr bool FAST = decltype(fast_c)::value;             // the tile is inside the operand, K % 16 == 0, float4 loads allowed: no checks
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * 256 + tid;
            if constexpr (FAST) {
                if (TR) {
                    const int k = c / (RT / 4), q4 = (c % (RT / 4)) * 4;
                    r[it] = *reinterpret_cast<const f32x4*>(P + (long)(k0 + k) * ld + r0 + q4);
                } else {
                    const int row = c >> 2, c4 = (c & 3) * 4;
                    r[it] = *reinterpret_cast<const f32x4*>(P + (long)(r0 + row) * ld + k0 + c4);
                }
                continue;
            }
            r[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TR) {
                const int k = c / (RT / 4), q4 = (c % (RT / 4)) * 4;
                const int gk = k0 + k, gr = r0 + q4;
                if (vec && gk < K && gr + 3 < R) r[it] = *reinterpret_cast<const f32x4*>(P + (long)gk * ld + gr);
                else if (gk < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gr + e < R) r[it][e] = P[(long)gk * ld + gr + e];
                }
            } else {
                const int row = c >> 2, c4 = (c & 3) * 4;
                const int gr = r0 + row, gk = k0 + c4;
                if (vec && gr < R && gk < K) r[it] = *reinterpret_cast<const f32x4*>(P + (long)gr * ld + gk);       // vec: K % 4 == 0
                else if (gr < R) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gk + e < K) r[it][e] = P[(long)gr * ld + gk + e];
                }
            }
        }
    };
    auto lstore_op = [&](auto tr_c, auto it_c, const f32x4* r, float* S, int RT) {
        constexpr int TR = decltype(tr_c)::value, IT = decltype(it_c)::value;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = it * 256 + tid;
            if (TR) {
                const int k = c / (RT / 4), q4 = (c % (RT / 4)) * 4;
                *reinterpret_cast<f32x4*>(S + k * (RT + 4) + q4) = r[it];
            } else {
                const int row = c >> 2, c4 = (c & 3) * 4;                 // k = c4 .. c4 + 3: the even pair, then the odd pair
                *reinterpret_cast<f32x2*>(S + row * LDT + (c4 >> 1)) = f32x2{r[it][0], r[it][2]};
                *reinterpret_cast<f32x2*>(S + row * LDT + 8 + (c4 >> 1)) = f32x2{r[it][1], r[it][3]};
            }
        }
    };
    typedef std::integral_constant<int, TRANSA> TA;
    typedef std::integral_constant<int, TRANSB ? 0 : 1> TBK;        // B stored [K][N] (no transb) is the k-major case
    typedef std::integral_constant<int, ITA> IA;
    typedef std::integral_constant<int, ITB> IB;
    auto k_loop = [&](auto fast_c) {
    auto gload = [&](int k0) {
        gload_op(fast_c, TA{}, IA{}, ra, A, lda, M, m0, k0, BM);
        gload_op(fast_c, TBK{}, IB{}, rb, B, ldb, N, n0, k0, BN);
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore_op(TA{}, IA{}, ra, sA, BM);
        lstore_op(TBK{}, IB{}, rb, sB, BN);
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
        f32x4 fa[2][2], fb[2][2];
        auto frag = [&](auto tr_c, const float* S, int row, int LK, f32x4 (&f)[2]) {      // k = hi, hi + 2, ..., hi + 14 of one row
            if constexpr (decltype(tr_c)::value) {
#pragma unroll
                for (int kp = 0; kp < 8; ++kp) f[kp >> 2][kp & 3] = S[(2 * kp + hi) * LK + row];
            } else {
                f[0] = *reinterpret_cast<const f32x4*>(S + row * LDT + hi * 8);
                f[1] = *reinterpret_cast<const f32x4*>(S + row * LDT + hi * 8 + 4);
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            frag(TA{}, sA, wm * 64 + i * 32 + l31, LKA, fa[i]);
            frag(TBK{}, sB, wn * 64 + i * 32 + l31, LKB, fb[i]);
        }
        if constexpr (X3) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                split_bf16x2(fa[i], ah[i], al[i]);
                split_bf16x2(fb[i], bh[i], bl[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        } else {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kp >> 2][kp & 3], fb[j][kp >> 2][kp & 3], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    };
    // interior tiles (all of them for the Nystrom / pinv / projection shapes) take the loop without bounds checks: each check is a branch
    const bool interior = m0 + BM <= M && n0 + BN <= N;
    if (vec && interior && K % BK == 0) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    auto store = [&](auto inside_c, auto diag_c) {                   // inside: no per-element bounds checks; diag: the + diag * I term is there
        constexpr bool IN = decltype(inside_c)::value, DG = decltype(diag_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (!IN && n >= N) continue;
            const float bn = bias ? bias[n] : 0.f;
            float* cn = Cm + n;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (IN || m < M) {
                        float v = alpha * acc[i][j][r] + bn;
                        if (DG) v += m == n ? diag : 0.f;
                        if (accumulate) v += cn[(long)m * ldc];
                        cn[(long)m * ldc] = v;
                    }
                }
        }
    };
    if (diag != 0.f) {
        if (interior) store(std::true_type{}, std::true_type{});
        else store(std::false_type{}, std::true_type{});
    } else {
        if (interior) store(std::true_type{}, std::false_type{});
        else store(std::false_type{}, std::false_type{});
    }
    if constexpr (DUAL) {                                            // the same product's second output, alpha2 * P + diag2 * I, as a pass of its own
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= N) continue;
            float* cn = dual.c2 + n;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < M) {
                        float v = dual.alpha2 * acc[i][j][r] + 0.f;
                        v += m == n ? dual.diag2 : 0.f;
                        cn[(long)m * ldc] = v;
                    }
                }
        }
    }
}

// ---- amds_set_matmul_precision(AMDS_MATMUL_HIGH), aligned products: the split happens ONCE per element, on its way into LDS ---------------------------------
// bgemm_f32_big_kernel<.., X3 = 1> splits every fragment in every wave that reads it, keeps fp32 LDS images, 16-deep K steps and two barriers per step: it is bound by
// that, not by its 12 MFMAs per step (1.6x the fp32 kernel).  Here: K steps of 32, global -> registers -> (hi | lo) bf16 images, double-buffered (one barrier per
// step), a fragment = 8 consecutive k of a row as ONE ds_read_b128 (row-major operands: [row][32 k], 64 B rows, chunk-swizzled) or four ds_read_b32 of (k, k + 1) pairs
// (k-major operands: [16 k pairs][rows + 8][2], written as ds_write_b128 of four rows' pairs: a thread fetches the float4 of k and of k + 1).  Only products whose
// tiles are all interior and aligned (M % BM == N % BN == K % 32 == 0, float4 loads) come here -- every Nystrom / pinv / projection shape; the rest stays on the
// kernel above.  Same epilogue (alpha, + diag I, bias, accumulate), same tile shapes and XCD order.
template <int TRANSB, int TRANSA, int WM, int WN, bool DUAL = false, bool ACC = false>
__global__ void __launch_bounds__(256, 2) bgemm_x3_kernel(const float* __restrict__ A, int lda, long sAo, long sAi, const float* __restrict__ B, int ldb, long sBo,
                                                       long sBi, float* __restrict__ Cm, int ldc, long sCo, long sCi, int inner, int M, int N, int K, float alpha,
                                                       float diag, const float* __restrict__ bias, int accumulate, int xcd, BgDual dual) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = 32;
    constexpr int TRA = TRANSA, TRB = TRANSB ? 0 : 1;
    constexpr int PR = 64;                                           // row-major image: bytes per row, no padding: the 16-byte chunk index is XORed with (row >> 2) & 3, so the
                                                                     // ds_read_b128 of 16 consecutive rows hit 16 different bank groups (pitch 80 B did too, but cost the second workgroup per CU)
    constexpr int IMG_A = TRA ? 16 * (BM + 8) * 4 : BM * PR, IMG_B = TRB ? 16 * (BN + 8) * 4 : BN * PR;      // bytes of ONE (hi or lo) image
    constexpr int STAGE = 2 * IMG_A + 2 * IMG_B;
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;    // float4 per thread and K step
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd) {                                                       // (as bgemm_f32_big_kernel)
        const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
        const int L = bx + gx * (by + gy * bz), q = total >> 3, r = total & 7, x = L & 7;
        const int Lp = x * q + min(x, r) + (L >> 3);
        bx = Lp % gx;
        const int t = Lp / gx;
        by = t % gy;
        bz = t / gy;
    }
    const int zo = bz / inner, zi = bz - zo * inner;
    A += zo * sAo + zi * sAi;
    B += zo * sBo + zi * sBi;
    Cm += zo * sCo + zi * sCi;
    if constexpr (DUAL) dual.c2 += zo * sCo + zi * sCi;
    const int m0 = by * BM, n0 = bx * BN;
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NA], rb[NB];
    // row-major source: piece c = (row, 4 k); k-major source: pieces come in pairs (k, k + 1) of the same four rows
    auto gload_op = [&](auto tr_c, auto n_c, f32x4* r, const float* P, int ld, int r0, int k0, auto rt_c) {
        constexpr int TR = decltype(tr_c)::value, NP = decltype(n_c)::value, RT = decltype(rt_c)::value;
        if constexpr (TR) {
#pragma unroll
            for (int it = 0; it < NP / 2; ++it) {
                const int c = it * 256 + tid, kp = c / (RT / 4), q4 = (c % (RT / 4)) * 4;
                const float* src = P + (long)(k0 + 2 * kp) * ld + r0 + q4;
                r[2 * it] = *reinterpret_cast<const f32x4*>(src);
                r[2 * it + 1] = *reinterpret_cast<const f32x4*>(src + ld);
            }
        } else {
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                const int c = it * 256 + tid, row = c >> 3, c4 = (c & 7) * 4;
                r[it] = *reinterpret_cast<const f32x4*>(P + (long)(r0 + row) * ld + k0 + c4);
            }
        }
    };
    auto split4 = [&](const f32x4& v, bf16x4& h, bf16x4& l) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16 t = (bf16)v[e];
            h[e] = t;
            l[e] = (bf16)(v[e] - (float)t);
        }
    };
    auto lstore_op = [&](auto tr_c, auto n_c, const f32x4* r, char* Sh, char* Sl, auto rt_c) {
        constexpr int TR = decltype(tr_c)::value, NP = decltype(n_c)::value, RT = decltype(rt_c)::value;
        if constexpr (TR) {
#pragma unroll
            for (int it = 0; it < NP / 2; ++it) {
                const int c = it * 256 + tid, kp = c / (RT / 4), q4 = (c % (RT / 4)) * 4;
                bf16x4 h0, l0, h1, l1;
                split4(r[2 * it], h0, l0);
                split4(r[2 * it + 1], h1, l1);
                bf16x8 wh, wl;                                       // four rows' (k, k + 1) pairs
#pragma unroll
                for (int e = 0; e < 4; ++e) { wh[2 * e] = h0[e]; wh[2 * e + 1] = h1[e]; wl[2 * e] = l0[e]; wl[2 * e + 1] = l1[e]; }
                *reinterpret_cast<bf16x8*>(Sh + (kp * (RT + 8) + q4) * 4) = wh;
                *reinterpret_cast<bf16x8*>(Sl + (kp * (RT + 8) + q4) * 4) = wl;
            }
        } else {
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                const int c = it * 256 + tid, row = c >> 3, c4 = (c & 7) * 4;
                bf16x4 h, l;
                split4(r[it], h, l);
                const int off = row * PR + (((c4 >> 3) ^ ((row >> 2) & 3)) << 4) + (c4 & 7) * 2;
                *reinterpret_cast<bf16x4*>(Sh + off) = h;
                *reinterpret_cast<bf16x4*>(Sl + off) = l;
            }
        }
    };
    auto frag = [&](auto tr_c, const char* S, int row, int s16, auto rt_c) -> bf16x8 {       // k = s16 * 16 + hi * 8 + (0 .. 7) of one row
        constexpr int TR = decltype(tr_c)::value, RT = decltype(rt_c)::value;
        if constexpr (TR) {
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const unsigned*>(S + ((s16 * 8 + hi * 4 + j) * (RT + 8) + row) * 4);
            return __builtin_bit_cast(bf16x8, w);
        } else {
            return *reinterpret_cast<const bf16x8*>(S + row * PR + (((s16 * 2 + hi) ^ ((row >> 2) & 3)) << 4));
        }
    };
    typedef std::integral_constant<int, TRA> TA;
    typedef std::integral_constant<int, TRB> TBK;
    typedef std::integral_constant<int, NA> CA;
    typedef std::integral_constant<int, NB> CB;
    typedef std::integral_constant<int, BM> RA;
    typedef std::integral_constant<int, BN> RB;
    auto stage_ptr = [&](int st, int which) { return smem + st * STAGE + (which == 0 ? 0 : which == 1 ? IMG_A : which == 2 ? 2 * IMG_A : 2 * IMG_A + IMG_B); };
    gload_op(TA{}, CA{}, ra, A, lda, m0, 0, RA{});
    gload_op(TBK{}, CB{}, rb, B, ldb, n0, 0, RB{});
    lstore_op(TA{}, CA{}, ra, stage_ptr(0, 0), stage_ptr(0, 1), RA{});
    lstore_op(TBK{}, CB{}, rb, stage_ptr(0, 2), stage_ptr(0, 3), RB{});
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) {
            gload_op(TA{}, CA{}, ra, A, lda, m0, k0 + BK, RA{});
            gload_op(TBK{}, CB{}, rb, B, ldb, n0, k0 + BK, RB{});
        }
        const char *sAh = stage_ptr(cur, 0), *sAl = stage_ptr(cur, 1), *sBh = stage_ptr(cur, 2), *sBl = stage_ptr(cur, 3);
#pragma unroll
        for (int s16 = 0; s16 < 2; ++s16) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = frag(TA{}, sAh, wm * 64 + i * 32 + l31, s16, RA{});
                al[i] = frag(TA{}, sAl, wm * 64 + i * 32 + l31, s16, RA{});
                bh[i] = frag(TBK{}, sBh, wn * 64 + i * 32 + l31, s16, RB{});
                bl[i] = frag(TBK{}, sBl, wn * 64 + i * 32 + l31, s16, RB{});
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            lstore_op(TA{}, CA{}, ra, stage_ptr(cur ^ 1, 0), stage_ptr(cur ^ 1, 1), RA{});
            lstore_op(TBK{}, CB{}, rb, stage_ptr(cur ^ 1, 2), stage_ptr(cur ^ 1, 3), RB{});
        }
        __syncthreads();
        cur ^= 1;
    }
    auto store = [&](auto diag_c) {
        constexpr bool DG = decltype(diag_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            const float bn = bias ? bias[n] : 0.f;
            float* cn = Cm + n;
            // ACC (C += ...): the 32 old values of this column block first.  Read-add-store per element made every load wait for the store in front of it (same pointer:
            // the compiler cannot reorder them) -- 64 dependent round trips per lane, 205 us for a 512 x 256^3 product against 90 us without the flag (round 6 trace).
            // A template parameter, not the runtime flag: the 32 registers must not exist in the plain instance (242 VGPRs and one workgroup per CU when they did).
            float old[2][16];
            if constexpr (ACC) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[i][r] = cn[(long)(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * ldc];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = alpha * acc[i][j][r] + bn;
                    if (DG) v += m == n ? diag : 0.f;
                    if constexpr (ACC) v += old[i][r];
                    cn[(long)m * ldc] = v;
                }
        }
    };
    if (diag != 0.f) store(std::true_type{});
    else store(std::false_type{});
    if constexpr (DUAL) {                                            // second output of the same product
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* cn = dual.c2 + n0 + wn * 64 + j * 32 + l31;
            const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = dual.alpha2 * acc[i][j][r] + 0.f;
                    v += m == n ? dual.diag2 : 0.f;
                    cn[(long)m * ldc] = v;
                }
        }
    }
}

__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ x, long rows, int cols) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    if (row >= rows) return;
    float* p = x + row * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int c = tid; c < cols; c += 256) m = fmaxf(m, p[c]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid; c < cols; c += 256) { const float e = expf(p[c] - m); p[c] = e; s += e; }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = tid; c < cols; c += 256) p[c] *= inv;
}

// One WAVE per row, the row in registers (cols <= 256 NV4, cols % 4 == 0): one read, one write, no LDS, no barrier.  (A 256-thread workgroup
// per row of 256 ... 1280 floats made three passes with two block reductions each: 1.5 TB/s on the Nystrom attention matrices.)
template <int NV4>
__global__ void __launch_bounds__(256) softmax_rows_wave_kernel(float* __restrict__ x, long rows, int cols) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, nv = cols >> 2;
    f32x4* p = reinterpret_cast<f32x4*>(x + row * cols);
    f32x4 v[NV4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int idx = i * 64 + lane;
        v[i] = idx < nv ? p[idx] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        m = fmaxf(fmaxf(m, fmaxf(v[i][0], v[i][1])), fmaxf(v[i][2], v[i][3]));
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = expf(v[i][e] - m); s += v[i][e]; }
    }
    const float inv = 1.0f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) p[idx] = v[i] * inv;
    }
}

// x: [outer][n][ld] view (head slice of width d at a column offset baked into the pointer); out [outer][inner][m][d]
__global__ void landmark_mean_kernel(const float* __restrict__ x, long sxo, long sxi, int ld, float* __restrict__ out, int inner,
                                     int m, int l, int d, float scale) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * d) return;
    const int j = idx / d, c = idx - j * d;
    const float* p = x + zo * sxo + zi * sxi + (long)j * l * ld + c;
    float s = 0.f;
    for (int t = 0; t < l; ++t) s += p[(long)t * ld];
    out[((long)z * m + j) * d + c] = s * scale;
}

// the same, four channels per thread (d, ld multiples of 4, 16-byte aligned rows): the head slices are 256-byte pieces of 6 KB rows -- wide loads, l of them in flight
__global__ void landmark_mean4_kernel(const float* __restrict__ x, long sxo, long sxi, int ld, float* __restrict__ out, int inner, int m, int l, int d, float scale) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int d4 = d >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * d4) return;
    const int j = idx / d4, c = (idx - j * d4) * 4;
    const float* p = x + zo * sxo + zi * sxi + (long)j * l * ld + c;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < l; ++t) s += *reinterpret_cast<const f32x4*>(p + (long)t * ld);
    *reinterpret_cast<f32x4*>(out + ((long)z * m + j) * d + c) = s * scale;
}

// per matrix: max over rows of sum_j |x_ij| and max over cols of sum_i |x_ij|; combined across matrices with integer
// atomicMax on the (non-negative) float bit patterns -> order independent, deterministic.
// Lanes along the row (coalesced 16-byte loads); a lane keeps the column sums of ITS columns over its rows, the eight half-waves' partials
// meet in LDS.  (Round 1's form -- a thread per row walking it element by element, every load of a wave touching 64 cache lines -- took 243 us per call at
// 512 matrices of 256 x 256: 0.55 TB/s.)  A HALF-wave per row (its sum by DPP, no LDS round trips), NV = float4 per lane and row: n <= 128 NV.
template <int NV>
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int n, unsigned* __restrict__ out2) {
    __shared__ float sc[8][128 * NV];
    const float* p = x + (long)blockIdx.x * n * n;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = threadIdx.x >> 5;           // eight half-waves, a row each, two rows of a half-wave in flight
    f32x4 cs[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) cs[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rmax = 0.f;
    for (int i = half; i < n; i += 32) {                          // rows i, i + 8, i + 16, i + 24: 4 NV loads of a lane in flight
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 v[4][NV];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int c = (k * 32 + l31) * 4;
                v[q][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c < n && i + 8 * q < n) v[q][k] = *reinterpret_cast<const f32x4*>(p + (long)(i + 8 * q) * n + c);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = fabsf(v[q][k][e]); rs[q] += a; cs[k][e] += a; }
#pragma unroll
        for (int q = 0; q < 4; ++q) rmax = fmaxf(rmax, half_wave_sum(rs[q]));
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(&sc[half][(k * 32 + l31) * 4]) = cs[k];
    __syncthreads();
    float cmax = 0.f;
    for (int c = threadIdx.x; c < n; c += 256)
        cmax = fmaxf(cmax, ((sc[0][c] + sc[1][c]) + (sc[2][c] + sc[3][c])) + ((sc[4][c] + sc[5][c]) + (sc[6][c] + sc[7][c])));
    rmax = wave_max(rmax); cmax = wave_max(cmax);
    __syncthreads();                                            // (sc is read: reuse its first words for the four waves' maxima -- ONE pair of atomics per workgroup)
    if (lane == 0) { sc[0][threadIdx.x >> 6] = rmax; sc[0][4 + (threadIdx.x >> 6)] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out2, __float_as_uint(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]))));
        atomicMax(out2 + 1, __float_as_uint(fmaxf(fmaxf(sc[0][4], sc[0][5]), fmaxf(sc[0][6], sc[0][7]))));
    }
}
// any n (unaligned rows, n > 1024): a thread per row / column
__global__ void __launch_bounds__(256) absmax_any_kernel(const float* __restrict__ x, int n, unsigned* __restrict__ out2) {
    const float* p = x + (long)blockIdx.x * n * n;
    float rmax = 0.f, cmax = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        float rs = 0.f, cs = 0.f;
        for (int j = 0; j < n; ++j) { rs += fabsf(p[(long)i * n + j]); cs += fabsf(p[(long)j * n + i]); }
        rmax = fmaxf(rmax, rs); cmax = fmaxf(cmax, cs);
    }
    rmax = wave_max(rmax); cmax = wave_max(cmax);
    if ((threadIdx.x & 63) == 0) { atomicMax(out2, __float_as_uint(rmax)); atomicMax(out2 + 1, __float_as_uint(cmax)); }
}
// z = x^T / (mx[0] * mx[1]) per matrix, 32 x 32 tiles through LDS (both sides coalesced; the 16 x 16 direct form read 64-byte pieces: 114 us per call)
__global__ void __launch_bounds__(256) transpose_scale_kernel(const float* __restrict__ x, float* __restrict__ z, int n, const unsigned* __restrict__ mx) {
    __shared__ float t[32][33];
    const float inv = 1.0f / (__uint_as_float(mx[0]) * __uint_as_float(mx[1]));
    const long base = (long)blockIdx.z * n * n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                      // 32 x 8
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;                        // z[i0 ..][j0 ..] = x[j0 ..][i0 ..]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + ty + 8 * r, i = i0 + tx;
        if (j < n && i < n) t[ty + 8 * r][tx] = x[base + (long)j * n + i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;
        if (i < n && j < n) z[base + (long)i * n + j] = t[tx][ty + 8 * r] * inv;
    }
}

// out[z][t][c] += sum_k w[head][k] * v[z][t + k - pad][c]   (z = b*H + head; v strided view with row stride ldv)
__global__ void dwconv_seq_kernel(const float* __restrict__ v, long svo, long svi, int ldv, const float* __restrict__ w,
                                  float* __restrict__ out, long soo, long soi, int ldo, int inner, int n, int d, int taps) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const int t = idx / d, c = idx - t * d;
    const float* p = v + zo * svo + zi * svi + c;
    const float* wk = w + (long)zi * taps;
    const int pad = taps / 2;
    float s = 0.f;
    for (int k = 0; k < taps; ++k) {
        const int tt = t + k - pad;
        if (tt >= 0 && tt < n) s += wk[k] * p[(long)tt * ldv];
    }
    out[zo * soo + zi * soi + (long)t * ldo + c] += s;
}

// the same sum for ONE position `row` of every sequence, into a compact [z][d] output (the class token's row of TransMIL's second layer, of which nothing else is read)
__global__ void dwconv_seq_row_kernel(const float* __restrict__ v, long svo, long svi, int ldv, const float* __restrict__ w, float* __restrict__ out, long soo, long soi,
                                      int inner, int n, int d, int taps, int row) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float* p = v + zo * svo + zi * svi + c;
    const float* wk = w + (long)zi * taps;
    const int pad = taps / 2;
    float s = 0.f;
    for (int k = 0; k < taps; ++k) {
        const int tt = row + k - pad;
        if (tt >= 0 && tt < n) s = fmaf(wk[k], p[(long)tt * ldv], s);
    }
    out[zo * soo + zi * soi + c] += s;
}

// The same sum with a register window: a lane owns one channel and TT consecutive positions, loads its TT + TAPS - 1 inputs once (a wave
// reads 64 consecutive channels per position: 256-byte segments) and runs the TAPS x TT multiply-adds out of registers with the taps in
// scalar registers.  (The one-thread-per-output form above issues TAPS dependent loads per output: 1.05 ms for the 64 x 8 x 1280 x 64 values of
// a TransMIL layer, ~0.3 TB/s.)  Four waves of a workgroup take four consecutive runs of TT positions.
template <int TAPS, int TT>
__global__ void __launch_bounds__(256) dwconv_seq_win_kernel(const float* __restrict__ v, long svo, long svi, int ldv, const float* __restrict__ w,
                                                             float* __restrict__ out, long soo, long soi, int ldo, int inner, int n, int d) {
    constexpr int PAD = TAPS / 2, NIN = TT + TAPS - 1;
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int cgroups = (d + 63) >> 6;
    const int cg = blockIdx.x % cgroups, tb = blockIdx.x / cgroups;
    const int c = cg * 64 + (threadIdx.x & 63);
    const int t0 = (tb * 4 + (threadIdx.x >> 6)) * TT;
    if (c >= d || t0 >= n) return;
    const float* p = v + zo * svo + zi * svi + c;
    const float* wk = w + (long)zi * TAPS;
    float in[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int tt = t0 + i - PAD;
        in[i] = (tt >= 0 && tt < n) ? p[(long)tt * ldv] : 0.f;
    }
    float wr[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) wr[k] = wk[k];
    float* o = out + zo * soo + zi * soi + c;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
        if (t0 + j < n) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) s = fmaf(wr[k], in[j + k], s);
            o[(long)(t0 + j) * ldo] += s;
        }
    }
}

// x: [B][1 + H*W][C] tokens (row 0 = class token, untouched); y same shape
// PPEG (reference trans_mil.py:265-283): y = x + dwconv7(x) + dwconv5(x) + dwconv3(x) on the HxW token grid (class token passed
// through).  The three depthwise kernels and the identity collapse into ONE 7x7 kernel per channel; a workgroup owns 256 channels
// of one grid row: the combined taps are built once in LDS (coalesced reads of the [C][k*k] weight tensors, tap-major image),
// held in 49 registers per lane, and the row is swept with a 7x7 register window that loads one new column (7 values, coalesced
// across channels) per output.  (v1: one thread per output with 83 strided weight loads each -- 6.6 ms for 64 bags.)
__global__ void __launch_bounds__(256) ppeg_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w7,
                                                   const float* __restrict__ b7, const float* __restrict__ w5, const float* __restrict__ b5,
                                                   const float* __restrict__ w3, const float* __restrict__ b3, int Hh, int Ww, int C) {
    constexpr int SWP = 257;                                     // tap pitch: the builders below write a channel's taps with consecutive lanes (pitch 256 = one bank)
    __shared__ float sw[49 * SWP];
    const int b = blockIdx.z, i = blockIdx.y, c0 = blockIdx.x * 256, tid = threadIdx.x, c = c0 + tid;
    const int nc = min(256, C - c0);
    for (int idx = tid; idx < 49 * SWP; idx += 256) sw[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < nc * 49; idx += 256) { const int cl = idx / 49, t = idx - cl * 49; sw[t * SWP + cl] = w7[(long)c0 * 49 + idx]; }
    __syncthreads();
    for (int idx = tid; idx < nc * 25; idx += 256) { const int cl = idx / 25, t = idx - cl * 25; sw[((t / 5 + 1) * 7 + t % 5 + 1) * SWP + cl] += w5[(long)c0 * 25 + idx]; }
    __syncthreads();
    for (int idx = tid; idx < nc * 9; idx += 256) { const int cl = idx / 9, t = idx - cl * 9; sw[((t / 3 + 2) * 7 + t % 3 + 2) * SWP + cl] += w3[(long)c0 * 9 + idx]; }
    __syncthreads();
    const long base = ((long)b * (1 + Hh * Ww)) * C;
    if (c >= C) return;
    if (i == 0) y[base + c] = x[base + c];                       // class token (trans_mil.py:275, 282)
    float w[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) w[t] = sw[t * SWP + tid];
    w[24] += 1.0f;                                               // the identity term
    const float bsum = b7[c] + b5[c] + b3[c];
    float win[7][7];                                             // win[r][q] = x[i + r - 3][j + q - 3]
    const float* xr[7];
    bool rok[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int ii = i + r - 3;
        rok[r] = ii >= 0 && ii < Hh;
        xr[r] = x + base + (long)(1 + (rok[r] ? ii : 0) * Ww) * C + c;
    }
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int jj = q - 3;
            win[r][q] = (rok[r] && jj >= 0 && jj < Ww) ? xr[r][(long)jj * C] : 0.f;
        }
    for (int j = 0; j < Ww; ++j) {
        float s = bsum;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int q = 0; q < 7; ++q) s = fmaf(w[r * 7 + q], win[r][q], s);
        y[base + (long)(1 + i * Ww + j) * C + c] = s;
        const int jn = j + 4;                                    // column entering the window
#pragma unroll
        for (int r = 0; r < 7; ++r) {
#pragma unroll
            for (int q = 0; q < 6; ++q) win[r][q] = win[r][q + 1];
            win[r][6] = (rok[r] && jn < Ww) ? xr[r][(long)jn * C] : 0.f;
        }
    }
}


// ---- backward pieces (TransMIL training; reference trans_mil.py differentiated by autograd in LitTileClassifier._step) ---------------
// ds = p o (dp - rowsum(p o dp)), in place on dp
__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows, int cols) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    if (row >= rows) return;
    const float* pr = p + row * cols;
    float* dr = dp + row * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int c = tid; c < cols; c += 256) s += pr[c] * dr[c];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    for (int c = tid; c < cols; c += 256) dr[c] = pr[c] * (dr[c] - s);
}

template <int NV4>
__global__ void __launch_bounds__(256) softmax_rows_bwd_wave_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows, int cols) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, nv = cols >> 2;
    const f32x4* pr = reinterpret_cast<const f32x4*>(p + row * cols);
    f32x4* dr = reinterpret_cast<f32x4*>(dp + row * cols);
    f32x4 pv[NV4], dv[NV4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int idx = i * 64 + lane;
        const bool ok = idx < nv;
        pv[i] = ok ? pr[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        dv[i] = ok ? dr[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) s += pv[i][e] * dv[i][e];
    }
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int idx = i * 64 + lane;
        if (idx < nv) dr[idx] = pv[i] * (dv[i] - s);
    }
}

// dx[z][j*l + t][c] (+)= scale * dout[z][j][c]   (dx addressed like the forward's x: head slice of a packed qkv-shaped tensor)
__global__ void landmark_mean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, long sxo, long sxi, int ld, int inner,
                                         int m, int l, int d, float scale, int accumulate) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * l * d) return;
    const int t = idx / d, c = idx - t * d;
    const float g = scale * dout[((long)z * m + t / l) * d + c];
    float* q = dx + zo * sxo + zi * sxi + (long)t * ld + c;
    *q = accumulate ? *q + g : g;
}

// the same, four channels per thread (d, ld multiples of 4, 16-byte aligned rows)
__global__ void landmark_mean_bwd4_kernel(const float* __restrict__ dout, float* __restrict__ dx, long sxo, long sxi, int ld, int inner, int m, int l, int d, float scale,
                                          int accumulate) {
    const int z = blockIdx.y, zo = z / inner, zi = z - zo * inner;
    const int d4 = d >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * l * d4) return;
    const int t = idx / d4, c = (idx - t * d4) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + ((long)z * m + t / l) * d + c) * scale;
    f32x4* q = reinterpret_cast<f32x4*>(dx + zo * sxo + zi * sxi + (long)t * ld + c);
    *q = accumulate ? *q + g : g;
}

// part[zo][zi][k] = sum over (t, c) of dout[zo,zi][t][c] * v[zo,zi][t + k - taps/2][c]: one workgroup per (tap k, inner zi, outer zo), lane = column
// (coalesced rows), fixed-order tree; amds_colsum then adds the `outer` partials in index order.  (One workgroup per (k, zi) looping over all
// bags took 10 ms per call at 64 bags x 1280 tokens: 264 workgroups with 20 k serial iterations each.)
__global__ void __launch_bounds__(256) dwconv_seq_wgrad_kernel(const float* __restrict__ dout, long soo, long soi, int ldo, const float* __restrict__ v,
                                                               long svo, long svi, int ldv, float* __restrict__ part, int inner, int n, int d, int taps) {
    __shared__ float red[256];
    const int k = blockIdx.x, zi = blockIdx.y, zo = blockIdx.z;
    const int pad = taps / 2;
    const float* po = dout + zo * soo + zi * soi;
    const float* pv = v + zo * svo + zi * svi;
    float s = 0.f;
    if (d <= 256 && 256 % d == 0) {
        const int c = threadIdx.x % d, tstep = 256 / d;
        const int lo = max(0, pad - k), hi = min(n, n + pad - k);          // rows t with 0 <= t + k - pad < n
        for (int t = lo + (int)threadIdx.x / d; t < hi; t += tstep) s += po[(long)t * ldo + c] * pv[(long)(t + k - pad) * ldv + c];
    } else {
        const long per = (long)n * d;
        for (long idx = threadIdx.x; idx < per; idx += 256) {
            const int t = (int)(idx / d), c = (int)(idx - (long)t * d);
            const int tt = t + k - pad;
            if (tt >= 0 && tt < n) s += po[(long)t * ldo + c] * pv[(long)tt * ldv + c];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[((long)zo * inner + zi) * taps + k] = red[0];
}

// The same sums with the 33 taps as 33 accumulators per lane: a lane owns a channel, a wave every fourth run of TT positions; per run it
// loads TT values of dout and TT + 32 of v once and does the 33 x TT multiply-adds out of registers (the kernel above reads both tensors once
// PER TAP: 11 GB through L2 per call, 1.15 ms).  Fixed-order reduction: lanes (butterfly), then the four waves in index order.
template <int TAPS, int TT>
__global__ void __launch_bounds__(256) dwconv_seq_wgrad_win_kernel(const float* __restrict__ dout, long soo, long soi, int ldo, const float* __restrict__ v,
                                                                   long svo, long svi, int ldv, float* __restrict__ part, int inner, int n, int d) {
    constexpr int PAD = TAPS / 2, NIN = TT + TAPS - 1;
    __shared__ float red[4][TAPS];
    const int zi = blockIdx.x, zo = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* po = dout + zo * soo + zi * soi;
    const float* pv = v + zo * svo + zi * svi;
    float s[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) s[k] = 0.f;
    for (int c0 = 0; c0 < d; c0 += 64) {
        const int c = c0 + lane;
        const bool cok = c < d;
        for (int t0 = wave * TT; t0 < n; t0 += 4 * TT) {
            float g[TT], in[NIN];
#pragma unroll
            for (int j = 0; j < TT; ++j) g[j] = (cok && t0 + j < n) ? po[(long)(t0 + j) * ldo + c] : 0.f;
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int tt = t0 + i - PAD;
                in[i] = (cok && tt >= 0 && tt < n) ? pv[(long)tt * ldv + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < TAPS; ++k)
#pragma unroll
                for (int j = 0; j < TT; ++j) s[k] = fmaf(g[j], in[j + k], s[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        const float w = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < TAPS) {
        const int k = threadIdx.x;
        part[((long)zo * inner + zi) * TAPS + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// PPEG weight gradients: part[chunk][tap][c] = sum over this chunk's bags and the whole grid of dy[b,i,j,c] * x[b,i+r-3,j+q-3,c] for
// tap = r*7+q < 49; tap 49 = sum dy (the three biases share it).  The 5x5 / 3x3 kernels' gradients are the central taps.
__global__ void __launch_bounds__(256) ppeg_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                         int B, int Hh, int Ww, int C, int bags_per_chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x, tap = blockIdx.y, chunk = blockIdx.z;
    if (c >= C) return;
    const int r = tap / 7 - 3, q = tap % 7 - 3;
    const int b0 = chunk * bags_per_chunk, b1 = min(B, b0 + bags_per_chunk);
    float s = 0.f;
    for (int b = b0; b < b1; ++b) {
        const long base = ((long)b * (1 + Hh * Ww)) * C + C + c;        // first grid token
        for (int i = 0; i < Hh; ++i) {
            const int ii = i + r;
            if (tap < 49 && (ii < 0 || ii >= Hh)) continue;
            for (int j = 0; j < Ww; ++j) {
                const float g = dy[base + (long)(i * Ww + j) * C];
                if (tap == 49) { s += g; continue; }
                const int jj = j + q;
                if (jj >= 0 && jj < Ww) s += g * x[base + (long)(ii * Ww + jj) * C];
            }
        }
    }
    part[((long)chunk * 50 + tap) * C + c] = s;
}

// The same 50 sums as 50 accumulators per lane: a lane owns a channel, a wave every fourth grid row of its chunk's bags; per row and run of
// JT columns it loads the JT values of dy once and, for each of the 7 kernel rows, JT + 6 values of x, and does the 7 x 7 x JT multiply-adds
// out of registers (the kernel above reads both tensors once PER TAP: 13 GB through L2, 1.7 ms per call).  Waves are added in index order.
template <int JT>
__global__ void __launch_bounds__(256) ppeg_wgrad_win_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                             int B, int Hh, int Ww, int C, int bags_per_chunk) {
    __shared__ float red[4][50][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, chunk = blockIdx.y;
    const bool cok = c < C;
    const int b0 = chunk * bags_per_chunk, b1 = min(B, b0 + bags_per_chunk);
    float s[50];
#pragma unroll
    for (int t = 0; t < 50; ++t) s[t] = 0.f;
    const int rows = (b1 - b0) * Hh;
    for (int ri = wave; ri < rows; ri += 4) {
        const int b = b0 + ri / Hh, i = ri % Hh;
        const long base = ((long)b * (1 + Hh * Ww)) * C + C + c;            // first grid token of the bag, this lane's channel
        for (int j0 = 0; j0 < Ww; j0 += JT) {
            float g[JT];
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                g[jj] = (cok && j0 + jj < Ww) ? dy[base + (long)(i * Ww + j0 + jj) * C] : 0.f;
                s[49] += g[jj];
            }
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int ii = i + r - 3;
                if (ii < 0 || ii >= Hh) continue;                       // wave-uniform
                float xin[JT + 6];
#pragma unroll
                for (int m = 0; m < JT + 6; ++m) {
                    const int jx = j0 + m - 3;
                    xin[m] = (cok && jx >= 0 && jx < Ww) ? x[base + (long)(ii * Ww + jx) * C] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 7; ++q)
#pragma unroll
                    for (int jj = 0; jj < JT; ++jj) s[r * 7 + q] = fmaf(g[jj], xin[jj + q], s[r * 7 + q]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 50; ++t) red[wave][t][lane] = s[t];
    __syncthreads();
    for (int t = wave; t < 50; t += 4)
        if (cok) part[((long)chunk * 50 + t) * C + c] = (red[0][t][lane] + red[1][t][lane]) + (red[2][t][lane] + red[3][t][lane]);
}

__global__ void relu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dh, float* __restrict__ dz, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dz[i] = h[i] > 0.f ? dh[i] : 0.f;
}

// pinv initialisation z0 = x^T / s, s = rmax * cmax (maxima over ALL matrices of the row sums / column sums of |x|): backward.
// stage 1: per matrix, dot[z] = sum_ij dz0[z][j][i] * x[z][i][j]; and the arg-max of the row sums / column sums, packed as
// (value bits << 32 | ~index) so that one 64-bit atomicMax gives the first index of the global maximum (deterministic).
// Aligned matrices (n a multiple of 32, n <= 1024): one wave per row for the row / column sums (as absmax_kernel), the dot product over 32 x 32 tiles of x against
// the transposed tiles of dz0 through LDS -- every load coalesced.  (A thread per row walking x[i][:], x[:][i] and dz0[:][i]: 244 us per call at 512 x 256 x 256.)
template <int NV>
__global__ void __launch_bounds__(256) pinv_init_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dz0, int n,
                                                                  float* __restrict__ dot, unsigned long long* __restrict__ arg2) {
    __shared__ float sc[8][128 * NV];
    __shared__ float tg[4][32][33];
    __shared__ float red[4];
    const long z = blockIdx.x;
    const float* p = x + z * n * n;
    const float* g = dz0 + z * n * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = threadIdx.x >> 5;
    f32x4 cs[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) cs[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long best_r = 0, best_c = 0;
    for (int i = half; i < n; i += 8) {                          // a half-wave per row (absmax_kernel)
        float rs = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * 32 + l31) * 4;
            if (c < n) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + (long)i * n + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = fabsf(v[e]); rs += a; cs[k][e] += a; }
            }
        }
        rs = half_wave_sum(rs);
        const unsigned long long kr = ((unsigned long long)__float_as_uint(rs) << 32) | (unsigned)(~(unsigned)(z * n + i));
        best_r = kr > best_r ? kr : best_r;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(&sc[half][(k * 32 + l31) * 4]) = cs[k];
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 256) {
        const float csum = ((sc[0][c] + sc[1][c]) + (sc[2][c] + sc[3][c])) + ((sc[4][c] + sc[5][c]) + (sc[6][c] + sc[7][c]));
        const unsigned long long kc = ((unsigned long long)__float_as_uint(csum) << 32) | (unsigned)(~(unsigned)(z * n + c));
        best_c = kc > best_c ? kc : best_c;
    }
    // dot = sum_ij dz0[j][i] x[i][j]: tile (I, J) of x against tile (J, I) of dz0, transposed in LDS; four tiles per barrier pair (32 loads of a thread in flight)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int nt = n / 32;
    float s = 0.f;
    for (int t0 = 0; t0 < nt * nt; t0 += 4) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = t0 + q, I = tile / nt, J = tile - I * nt;
            if (tile < nt * nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tg[q][ty + 8 * r][tx] = g[(long)(J * 32 + ty + 8 * r) * n + I * 32 + tx];       // tg[j][i] = dz0[J*32 + j][I*32 + i]
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = t0 + q, I = tile / nt, J = tile - I * nt;
            if (tile < nt * nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s += tg[q][tx][ty + 8 * r] * p[(long)(I * 32 + ty + 8 * r) * n + J * 32 + tx];   // x[i][j] * dz0[j][i], i = ty + 8 r, j = tx
            }
        }
    }
    s = wave_sum(s);
    // the arg-max keys: wave, then workgroup, then ONE pair of 64-bit atomics (every thread issuing its own pair was 262 k atomics on two addresses per call)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long r2 = __shfl_xor(best_r, o, 64), c2 = __shfl_xor(best_c, o, 64);
        best_r = r2 > best_r ? r2 : best_r;
        best_c = c2 > best_c ? c2 : best_c;
    }
    __shared__ unsigned long long keys[8];
    if (lane == 0) { red[wave] = s; keys[wave] = best_r; keys[4 + wave] = best_c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dot[z] = (red[0] + red[1]) + (red[2] + red[3]);
        unsigned long long br = keys[0], bc = keys[4];
#pragma unroll
        for (int w = 1; w < 4; ++w) { br = keys[w] > br ? keys[w] : br; bc = keys[4 + w] > bc ? keys[4 + w] : bc; }
        atomicMax(arg2, br);
        atomicMax(arg2 + 1, bc);
    }
}
// any n: a thread per row / column
__global__ void __launch_bounds__(256) pinv_init_bwd_stats_any_kernel(const float* __restrict__ x, const float* __restrict__ dz0, int n,
                                                                      float* __restrict__ dot, unsigned long long* __restrict__ arg2) {
    __shared__ float red[256];
    const long z = blockIdx.x;
    const float* p = x + z * n * n;
    const float* g = dz0 + z * n * n;
    float s = 0.f;
    unsigned long long best_r = 0, best_c = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        float rs = 0.f, cs = 0.f;
        for (int j = 0; j < n; ++j) {
            const float a = p[(long)i * n + j];
            rs += fabsf(a); cs += fabsf(p[(long)j * n + i]);
            s += g[(long)j * n + i] * a;
        }
        const unsigned idx = (unsigned)(z * n + i);
        const unsigned long long kr = ((unsigned long long)__float_as_uint(rs) << 32) | (unsigned)(~idx);
        const unsigned long long kc = ((unsigned long long)__float_as_uint(cs) << 32) | (unsigned)(~idx);
        best_r = kr > best_r ? kr : best_r;
        best_c = kc > best_c ? kc : best_c;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dot[z] = red[0];
    atomicMax(arg2, best_r);
    atomicMax(arg2 + 1, best_c);
}
// stage 2: dx[z][i][j] += dz0[z][j][i] / s  +  ds * (cmax on row i* of matrix z_r*  +  rmax on column j* of matrix z_c*),  ds = -sum(dot) / s^2
// (32 x 32 tiles of dz0 transposed through LDS: both sides coalesced)
__global__ void __launch_bounds__(256) pinv_init_bwd_apply_kernel(const float* __restrict__ dz0, float* __restrict__ dx, int n, int nmat, const float* __restrict__ dot_sum,
                                                                  const unsigned long long* __restrict__ arg2) {
    __shared__ float t[32][33];
    const float rmax = __uint_as_float((unsigned)(arg2[0] >> 32)), cmax = __uint_as_float((unsigned)(arg2[1] >> 32));
    const unsigned ridx = ~(unsigned)(arg2[0] & 0xFFFFFFFFull), cidx = ~(unsigned)(arg2[1] & 0xFFFFFFFFull);
    const float s = rmax * cmax;
    const float ds = -dot_sum[0] / (s * s);
    const long z = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;                        // dx[i0 ..][j0 ..] gets dz0[j0 ..][i0 ..]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + ty + 8 * r, i = i0 + tx;
        if (j < n && i < n) t[ty + 8 * r][tx] = dz0[z * n * n + (long)j * n + i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;
        if (i >= n || j >= n) continue;
        float g = t[tx][ty + 8 * r] / s;
        if ((unsigned)(z * n + i) == ridx) g += ds * cmax;         // d s / d (row sum i*) = cmax, spread over the row (x > 0: softmax output)
        if ((unsigned)(z * n + j) == cidx) g += ds * rmax;
        dx[z * n * n + (long)i * n + j] += g;
    }
}

}  // namespace amds

using namespace amds;

extern "C" int amds_softmax_rows_bwd(const float* p, float* dp, long rows, int cols, void* stream) {
    AMDS_REQUIRE(p && dp && rows >= 0 && cols > 0 && rows < (1L << 31), "amds_softmax_rows_bwd: bad arguments");
    if (rows == 0) return AMDS_OK;
    if (cols % 4 == 0 && cols <= 2048 && (((uintptr_t)p | (uintptr_t)dp) & 15) == 0 && (rows + 3) / 4 <= 0x7fffffffL) {
        const dim3 g((unsigned)((rows + 3) / 4));
        hipStream_t st = (hipStream_t)stream;
        if (cols <= 256) hipLaunchKernelGGL((softmax_rows_bwd_wave_kernel<1>), g, dim3(256), 0, st, p, dp, rows, cols);
        else if (cols <= 512) hipLaunchKernelGGL((softmax_rows_bwd_wave_kernel<2>), g, dim3(256), 0, st, p, dp, rows, cols);
        else if (cols <= 1024) hipLaunchKernelGGL((softmax_rows_bwd_wave_kernel<4>), g, dim3(256), 0, st, p, dp, rows, cols);
        else if (cols <= 1280) hipLaunchKernelGGL((softmax_rows_bwd_wave_kernel<5>), g, dim3(256), 0, st, p, dp, rows, cols);
        else hipLaunchKernelGGL((softmax_rows_bwd_wave_kernel<8>), g, dim3(256), 0, st, p, dp, rows, cols);
        AMDS_LAUNCH_CHECK("softmax_rows_bwd_wave_kernel");
        return AMDS_OK;
    }
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, rows, cols);
    AMDS_LAUNCH_CHECK("softmax_rows_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_landmark_mean_bwd(const float* dout, float* dx, long sxo, long sxi, int ld, int outer, int inner, int m, int l, int d,
                                      float scale, int accumulate, void* stream) {
    AMDS_REQUIRE(dout && dx && outer > 0 && inner > 0 && m > 0 && l > 0 && d > 0, "amds_landmark_mean_bwd: bad arguments");
    if (d % 4 == 0 && ld % 4 == 0 && sxo % 4 == 0 && sxi % 4 == 0 && (((uintptr_t)dx | (uintptr_t)dout) & 15) == 0)
        hipLaunchKernelGGL(landmark_mean_bwd4_kernel, dim3(cdiv((long)m * l * (d / 4), 256), outer * inner), dim3(256), 0, (hipStream_t)stream, dout, dx, sxo, sxi, ld, inner,
                           m, l, d, scale, accumulate);
    else
    hipLaunchKernelGGL(landmark_mean_bwd_kernel, dim3(cdiv((long)m * l * d, 256), outer * inner), dim3(256), 0, (hipStream_t)stream, dout, dx, sxo,
                       sxi, ld, inner, m, l, d, scale, accumulate);
    AMDS_LAUNCH_CHECK("landmark_mean_bwd_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_dwconv_seq_wgrad_workspace_bytes(int outer, int inner, int taps) {
    return (size_t)outer * inner * taps * 4 + amds_colsum_workspace_bytes(outer, inner * taps);
}
extern "C" int amds_dwconv_seq_wgrad(const float* dout, long soo, long soi, int ldo, const float* v, long svo, long svi, int ldv, float* dw,
                                     int outer, int inner, int n, int d, int taps, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(dout && v && dw && ws && outer > 0 && outer <= 65535 && inner > 0 && n > 0 && d > 0 && taps > 0 && (taps & 1), "amds_dwconv_seq_wgrad: bad arguments");
    if (ws_bytes < amds_dwconv_seq_wgrad_workspace_bytes(outer, inner, taps)) { set_error("amds_dwconv_seq_wgrad: workspace too small"); return AMDS_ERR_WORKSPACE; }
    float* part = (float*)ws;
    static const int win = [] { const char* e = getenv("AMDS_DWCONV_WIN"); return e ? atoi(e) : 1; }();       // 0: one workgroup per tap (A/B)
    if (taps == 33 && win) {
        hipLaunchKernelGGL((dwconv_seq_wgrad_win_kernel<33, 16>), dim3(inner, outer), dim3(256), 0, (hipStream_t)stream, dout, soo, soi, ldo, v, svo, svi,
                           ldv, part, inner, n, d);
        AMDS_LAUNCH_CHECK("dwconv_seq_wgrad_win_kernel");
    } else {
        hipLaunchKernelGGL(dwconv_seq_wgrad_kernel, dim3(taps, inner, outer), dim3(256), 0, (hipStream_t)stream, dout, soo, soi, ldo, v, svo, svi, ldv, part,
                           inner, n, d, taps);
        AMDS_LAUNCH_CHECK("dwconv_seq_wgrad_kernel");
    }
    char* cws = (char*)(part + (size_t)outer * inner * taps);
    return amds_colsum(part, (long)inner * taps, dw, outer, inner * taps, AMDS_F32, 0, cws, amds_colsum_workspace_bytes(outer, inner * taps), stream);
}

constexpr int PPEG_WGRAD_BAGS = 2;            // bags per partial (64 bags x 8 channel groups of the window kernel = 256 workgroups)
extern "C" size_t amds_ppeg_wgrad_workspace_bytes(int B, int C) {
    return (size_t)cdiv(B, PPEG_WGRAD_BAGS) * 50 * C * 4 + amds_colsum_workspace_bytes(cdiv(B, PPEG_WGRAD_BAGS), 50 * C);
}
extern "C" int amds_ppeg_wgrad(const float* x, const float* dy, float* dcorr, int B, int Hh, int Ww, int C, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(x && dy && dcorr && ws && B > 0 && Hh > 0 && Ww > 0 && C > 0, "amds_ppeg_wgrad: bad arguments");
    if (ws_bytes < amds_ppeg_wgrad_workspace_bytes(B, C)) { set_error("amds_ppeg_wgrad: workspace too small"); return AMDS_ERR_WORKSPACE; }
    const int nchunk = cdiv(B, PPEG_WGRAD_BAGS);
    float* part = (float*)ws;
    static const int win = [] { const char* e = getenv("AMDS_DWCONV_WIN"); return e ? atoi(e) : 1; }();       // 0: one thread per (tap, channel) (A/B)
    if (win) {
        hipLaunchKernelGGL((ppeg_wgrad_win_kernel<16>), dim3(cdiv(C, 64), nchunk), dim3(256), 0, (hipStream_t)stream, x, dy, part, B, Hh, Ww, C, PPEG_WGRAD_BAGS);
        AMDS_LAUNCH_CHECK("ppeg_wgrad_win_kernel");
    } else {
        hipLaunchKernelGGL(ppeg_wgrad_kernel, dim3(cdiv(C, 256), 50, nchunk), dim3(256), 0, (hipStream_t)stream, x, dy, part, B, Hh, Ww, C, PPEG_WGRAD_BAGS);
        AMDS_LAUNCH_CHECK("ppeg_wgrad_kernel");
    }
    char* cws = (char*)(part + (size_t)nchunk * 50 * C);
    return amds_colsum(part, 50L * C, dcorr, nchunk, 50 * C, AMDS_F32, 0, cws, amds_colsum_workspace_bytes(nchunk, 50 * C), stream);
}

extern "C" int amds_relu_bwd(const float* h, const float* dh, float* dz, long n, void* stream) {
    AMDS_REQUIRE(h && dh && dz && n >= 0, "amds_relu_bwd: bad arguments");
    if (n == 0) return AMDS_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)min((long)8192, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, dh, dz, n);
    AMDS_LAUNCH_CHECK("relu_bwd_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_pinv_init_bwd_workspace_bytes(int nmat) { return 16 + (size_t)nmat * 4 + 16 + amds_colsum_workspace_bytes(nmat, 1) + 64; }
extern "C" int amds_pinv_init_bwd(const float* x, const float* dz0, float* dx, int nmat, int n, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(x && dz0 && dx && ws && nmat > 0 && nmat <= 65535 && n > 0, "amds_pinv_init_bwd: bad arguments");
    if (ws_bytes < amds_pinv_init_bwd_workspace_bytes(nmat)) { set_error("amds_pinv_init_bwd: workspace too small"); return AMDS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* arg2 = (unsigned long long*)ws;
    float* dot = (float*)((char*)ws + 16);
    float* dot_sum = dot + nmat;
    char* cws = (char*)(dot_sum + 4);
    AMDS_HIP(hipMemsetAsync(arg2, 0, 16, st));
    const bool al = n % 32 == 0 && (((uintptr_t)x | (uintptr_t)dz0) & 15) == 0;
    if (al && n <= 256) hipLaunchKernelGGL((pinv_init_bwd_stats_kernel<2>), dim3(nmat), dim3(256), 0, st, x, dz0, n, dot, arg2);
    else if (al && n <= 1024) hipLaunchKernelGGL((pinv_init_bwd_stats_kernel<8>), dim3(nmat), dim3(256), 0, st, x, dz0, n, dot, arg2);
    else hipLaunchKernelGGL(pinv_init_bwd_stats_any_kernel, dim3(nmat), dim3(256), 0, st, x, dz0, n, dot, arg2);
    AMDS_LAUNCH_CHECK("pinv_init_bwd_stats_kernel");
    int rc = amds_colsum(dot, 1, dot_sum, nmat, 1, AMDS_F32, 0, cws, amds_colsum_workspace_bytes(nmat, 1), stream);
    if (rc != AMDS_OK) return rc;
    hipLaunchKernelGGL(pinv_init_bwd_apply_kernel, dim3(cdiv(n, 32), cdiv(n, 32), nmat), dim3(256), 0, st, dz0, dx, n, nmat, dot_sum, arg2);
    AMDS_LAUNCH_CHECK("pinv_init_bwd_apply_kernel");
    return AMDS_OK;
}

// the precision of the fp32 batched products (amds_bgemm_f32's 128 x 128 / 256 x 64 / 64 x 256 tile kernels) is a setting of the device's CONTEXT
// (amds_set_matmul_precision(ctx, level), api.hip); the Python side forwards torch's process-wide flag before every call
static int bgemm_f32_at(int precision, const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb,
                        float* Cm, int ldc, long sCo, long sCi, int outer, int inner, int M, int N, int K, float alpha,
                        float diag, const float* bias, int accumulate, void* stream, amds::BgDual dual = amds::BgDual{nullptr, 0.f, 0.f});
extern "C" int amds_bgemm_f32(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb,
                              float* Cm, int ldc, long sCo, long sCi, int outer, int inner, int M, int N, int K, float alpha,
                              float diag, const float* bias, int accumulate, void* stream) {
    return bgemm_f32_at(amds::ctx_matmul_precision(), A, lda, sAo, sAi, B, ldb, sBo, sBi, transb, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, diag,
                        bias, accumulate, stream);
}
// the feature-extraction paths that promise exact fp32 (the ViT's exact class-token stream, TICON, barspoon's class side) do not follow the process-wide level
namespace amds {
int bgemm_f32_exact(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb, float* Cm, int ldc, long sCo, long sCi,
                    int outer, int inner, int M, int N, int K, float alpha, float diag, const float* bias, int accumulate, void* stream) {
    return bgemm_f32_at(AMDS_MATMUL_HIGHEST, A, lda, sAo, sAi, B, ldb, sBo, sBi, transb, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, diag, bias, accumulate, stream);
}
}  // namespace amds
// C = alpha A op(B) + diag I and, from the SAME product, C2 = alpha2 A op(B) + diag2 I (same layout as C): the pinv iteration's A = a2 z and T1 = 7 I - A
// (reference trans_mil.py:31-33) as one launch instead of two -- the second product re-read both operands (0.27 GB at the bench shape) for the same
// accumulators.  Bits of the two separate calls.
extern "C" int amds_bgemm_f32_dual(const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb, float* Cm, float* C2, int ldc,
                                   long sCo, long sCi, int outer, int inner, int M, int N, int K, float alpha, float diag, float alpha2, float diag2, void* stream) {
    AMDS_REQUIRE(C2 && C2 != Cm, "amds_bgemm_f32_dual: the second output must be a different buffer");
    return bgemm_f32_at(amds::ctx_matmul_precision(), A, lda, sAo, sAi, B, ldb, sBo, sBi, transb, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, diag,
                        nullptr, 0, stream, amds::BgDual{C2, alpha2, diag2});
}
static int bgemm_f32_at(int precision, const float* A, int lda, long sAo, long sAi, const float* B, int ldb, long sBo, long sBi, int transb,
                        float* Cm, int ldc, long sCo, long sCi, int outer, int inner, int M, int N, int K, float alpha,
                        float diag, const float* bias, int accumulate, void* stream, amds::BgDual dual) {
    using namespace amds;
    AMDS_REQUIRE(A && B && Cm, "amds_bgemm_f32: null pointer");
    AMDS_REQUIRE(outer > 0 && inner > 0 && (long)outer * inner <= 65535 && M > 0 && N > 0 && K > 0, "amds_bgemm_f32: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM_F32, 2.0 * outer * inner * (double)M * N * K, st);
    const bool vec_ok = K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && sAo % 4 == 0 && sAi % 4 == 0 && sBo % 4 == 0 && sBi % 4 == 0 &&
                        ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0;
    const bool transa = (transb & 2) != 0;            // bit 1: A is stored [K][M] (pitch lda), the product is A^T B
    transb &= 1;
    if (transa || (vec_ok && (M >= 96 || N >= 96))) {
        // tile shape by the output's shape: 256 x 64 when N <= 64 (per-head d = 64 outputs), 64 x 256 when M <= 64, else 128 x 128
        const int shape = (N <= 64 && M > 64) ? 1 : (M <= 64 && N > 64) ? 2 : 0;
        const dim3 grid3(cdiv(N, shape == 1 ? 64 : shape == 2 ? 256 : 128), cdiv(M, shape == 1 ? 256 : shape == 2 ? 64 : 128), outer * inner);
        const int vec = vec_ok ? 1 : 0;
        static const int xcd = [] { const char* e = getenv("AMDS_BGEMM_XCD"); return e ? atoi(e) : 1; }();      // 0: launch order (A/B)
        const bool x3 = precision != AMDS_MATMUL_HIGHEST;
        if (dual.c2 && (transb || transa || shape != 0)) {               // the one-launch form exists for the plain 128 x 128 product (the pinv iteration's)
            int rc = bgemm_f32_at(precision, A, lda, sAo, sAi, B, ldb, sBo, sBi, transb | (transa ? 2 : 0), Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha, diag, bias, accumulate, stream);
            if (rc != AMDS_OK) return rc;
            return bgemm_f32_at(precision, A, lda, sAo, sAi, B, ldb, sBo, sBi, transb | (transa ? 2 : 0), dual.c2, ldc, sCo, sCi, outer, inner, M, N, K, dual.alpha2, dual.diag2, nullptr, 0, stream);
        }
        if (dual.c2) {
            if (x3 && vec_ok && M % 128 == 0 && N % 128 == 0 && K % 32 == 0)
                hipLaunchKernelGGL((bgemm_x3_kernel<0, 0, 2, 2, true>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, N, K, alpha, diag, bias,
                                   accumulate, xcd, dual);
            else if (x3)
                hipLaunchKernelGGL((bgemm_f32_big_kernel<0, 0, 2, 2, 1, true>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, N, K, alpha, diag,
                                   bias, accumulate, vec, xcd, dual);
            else
                hipLaunchKernelGGL((bgemm_f32_big_kernel<0, 0, 2, 2, 0, true>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, N, K, alpha, diag,
                                   bias, accumulate, vec, xcd, dual);
            AMDS_LAUNCH_CHECK("bgemm dual");
            return AMDS_OK;
        }
#define AMDS_BG(TB, TA, WM_, WN_)                                                                                                              \
    do {                                                                                                                                       \
        if (x3) hipLaunchKernelGGL((bgemm_f32_big_kernel<TB, TA, WM_, WN_, 1>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, \
                                   inner, M, N, K, alpha, diag, bias, accumulate, vec, xcd, dual);                                            \
        else hipLaunchKernelGGL((bgemm_f32_big_kernel<TB, TA, WM_, WN_, 0>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, \
                                inner, M, N, K, alpha, diag, bias, accumulate, vec, xcd, dual);                                               \
    } while (0)
#define AMDS_BG_SHAPE(TB, TA)                                     \
    do {                                                          \
        if (shape == 1) AMDS_BG(TB, TA, 4, 1);                    \
        else if (shape == 2) AMDS_BG(TB, TA, 1, 4);               \
        else AMDS_BG(TB, TA, 2, 2);                               \
    } while (0)
        // "high": products made of whole aligned tiles go to the kernel that splits on the way into LDS (AMDS_BGEMM_X3_LDS=0: the in-register split, A/B)
        static const bool x3_lds = !(getenv("AMDS_BGEMM_X3_LDS") && atoi(getenv("AMDS_BGEMM_X3_LDS")) == 0);
        const int bm = shape == 1 ? 256 : shape == 2 ? 64 : 128, bn = shape == 1 ? 64 : shape == 2 ? 256 : 128;
        if (x3 && x3_lds && vec_ok && M % bm == 0 && N % bn == 0 && K % 32 == 0) {
#define AMDS_BX(TB, TA, WM_, WN_)                                                                                                                                  \
    do {                                                                                                                                                          \
        if (accumulate)                                                                                                                                           \
            hipLaunchKernelGGL((bgemm_x3_kernel<TB, TA, WM_, WN_, false, true>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, \
                               N, K, alpha, diag, bias, accumulate, xcd, dual);                                                                                   \
        else                                                                                                                                                      \
            hipLaunchKernelGGL((bgemm_x3_kernel<TB, TA, WM_, WN_, false, false>), grid3, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, \
                               N, K, alpha, diag, bias, accumulate, xcd, dual);                                                                                   \
    } while (0)
#define AMDS_BX_SHAPE(TB, TA)                                     \
    do {                                                          \
        if (shape == 1) AMDS_BX(TB, TA, 4, 1);                    \
        else if (shape == 2) AMDS_BX(TB, TA, 1, 4);               \
        else AMDS_BX(TB, TA, 2, 2);                               \
    } while (0)
            if (transb && transa) AMDS_BX_SHAPE(1, 1);
            else if (transb) AMDS_BX_SHAPE(1, 0);
            else if (transa) AMDS_BX_SHAPE(0, 1);
            else AMDS_BX_SHAPE(0, 0);
#undef AMDS_BX_SHAPE
#undef AMDS_BX
            AMDS_LAUNCH_CHECK("bgemm_x3_kernel");
            return AMDS_OK;
        }
        if (transb && transa) AMDS_BG_SHAPE(1, 1);
        else if (transb) AMDS_BG_SHAPE(1, 0);
        else if (transa) AMDS_BG_SHAPE(0, 1);
        else AMDS_BG_SHAPE(0, 0);
#undef AMDS_BG_SHAPE
#undef AMDS_BG
        AMDS_LAUNCH_CHECK("bgemm_f32_big_kernel");
        return AMDS_OK;
    }
    const dim3 grid(cdiv(N, 64), cdiv(M, 64), outer * inner);
    if (transb) hipLaunchKernelGGL((bgemm_f32_kernel<1>), grid, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, N, K, alpha, diag, bias, accumulate);
    else hipLaunchKernelGGL((bgemm_f32_kernel<0>), grid, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, Cm, ldc, sCo, sCi, inner, M, N, K, alpha, diag, bias, accumulate);
    AMDS_LAUNCH_CHECK("bgemm_f32_kernel");
    if (dual.c2) {            // (small / unaligned products: the second output as a second launch)
        if (transb) hipLaunchKernelGGL((bgemm_f32_kernel<1>), grid, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, dual.c2, ldc, sCo, sCi, inner, M, N, K, dual.alpha2, dual.diag2, nullptr, 0);
        else hipLaunchKernelGGL((bgemm_f32_kernel<0>), grid, dim3(256), 0, st, A, lda, sAo, sAi, B, ldb, sBo, sBi, dual.c2, ldc, sCo, sCi, inner, M, N, K, dual.alpha2, dual.diag2, nullptr, 0);
        AMDS_LAUNCH_CHECK("bgemm_f32_kernel");
    }
    return AMDS_OK;
}

extern "C" int amds_softmax_rows(float* x, long rows, int cols, void* stream) {
    AMDS_REQUIRE(x && rows >= 0 && cols > 0 && rows < (1L << 31), "amds_softmax_rows: bad arguments");
    if (rows == 0) return AMDS_OK;
    if (cols % 4 == 0 && cols <= 2048 && ((uintptr_t)x & 15) == 0 && (rows + 3) / 4 <= 0x7fffffffL) {
        const dim3 g((unsigned)((rows + 3) / 4));
        hipStream_t st = (hipStream_t)stream;
        if (cols <= 256) hipLaunchKernelGGL((softmax_rows_wave_kernel<1>), g, dim3(256), 0, st, x, rows, cols);
        else if (cols <= 512) hipLaunchKernelGGL((softmax_rows_wave_kernel<2>), g, dim3(256), 0, st, x, rows, cols);
        else if (cols <= 1024) hipLaunchKernelGGL((softmax_rows_wave_kernel<4>), g, dim3(256), 0, st, x, rows, cols);
        else if (cols <= 1280) hipLaunchKernelGGL((softmax_rows_wave_kernel<5>), g, dim3(256), 0, st, x, rows, cols);
        else hipLaunchKernelGGL((softmax_rows_wave_kernel<8>), g, dim3(256), 0, st, x, rows, cols);
        AMDS_LAUNCH_CHECK("softmax_rows_wave_kernel");
        return AMDS_OK;
    }
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, rows, cols);
    AMDS_LAUNCH_CHECK("softmax_rows_kernel");
    return AMDS_OK;
}

extern "C" int amds_landmark_mean(const float* x, long sxo, long sxi, int ld, float* out, int outer, int inner, int m, int l, int d,
                                  float scale, void* stream) {
    AMDS_REQUIRE(x && out && outer > 0 && inner > 0 && m > 0 && l > 0 && d > 0, "amds_landmark_mean: bad arguments");
    if (d % 4 == 0 && ld % 4 == 0 && sxo % 4 == 0 && sxi % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(landmark_mean4_kernel, dim3(cdiv((long)m * (d / 4), 256), outer * inner), dim3(256), 0, (hipStream_t)stream, x, sxo, sxi, ld, out, inner, m, l, d,
                           scale);
    else
    hipLaunchKernelGGL(landmark_mean_kernel, dim3(cdiv((long)m * d, 256), outer * inner), dim3(256), 0, (hipStream_t)stream, x, sxo, sxi, ld,
                       out, inner, m, l, d, scale);
    AMDS_LAUNCH_CHECK("landmark_mean_kernel");
    return AMDS_OK;
}

extern "C" int amds_pinv_init(const float* x, float* z, int nmat, int n, void* scratch8, void* stream) {
    AMDS_REQUIRE(x && z && scratch8 && nmat > 0 && nmat <= 65535 && n > 0, "amds_pinv_init: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    AMDS_HIP(hipMemsetAsync(scratch8, 0, 8, st));
    if (n % 4 == 0 && n <= 256 && ((uintptr_t)x & 15) == 0) hipLaunchKernelGGL((absmax_kernel<2>), dim3(nmat), dim3(256), 0, st, x, n, (unsigned*)scratch8);
    else if (n % 4 == 0 && n <= 1024 && ((uintptr_t)x & 15) == 0) hipLaunchKernelGGL((absmax_kernel<8>), dim3(nmat), dim3(256), 0, st, x, n, (unsigned*)scratch8);
    else hipLaunchKernelGGL(absmax_any_kernel, dim3(nmat), dim3(256), 0, st, x, n, (unsigned*)scratch8);
    AMDS_LAUNCH_CHECK("absmax_kernel");
    hipLaunchKernelGGL(transpose_scale_kernel, dim3(cdiv(n, 32), cdiv(n, 32), nmat), dim3(256), 0, st, x, z, n, (const unsigned*)scratch8);
    AMDS_LAUNCH_CHECK("transpose_scale_kernel");
    return AMDS_OK;
}

extern "C" int amds_dwconv_seq(const float* v, long svo, long svi, int ldv, const float* w, float* out, long soo, long soi, int ldo,
                               int outer, int inner, int n, int d, int taps, void* stream) {
    AMDS_REQUIRE(v && w && out && outer > 0 && inner > 0 && n > 0 && d > 0 && taps > 0 && (taps & 1), "amds_dwconv_seq: bad arguments");
    static const int win = [] { const char* e = getenv("AMDS_DWCONV_WIN"); return e ? atoi(e) : 1; }();       // 0: one thread per output (A/B)
    if (taps == 33 && win) {           // TransMIL's residual convolution (trans_mil.py:108-110: kernel 33)
        constexpr int TT = 32;
        hipLaunchKernelGGL((dwconv_seq_win_kernel<33, TT>), dim3(cdiv(n, 4 * TT) * cdiv(d, 64), outer * inner), dim3(256), 0, (hipStream_t)stream, v,
                           svo, svi, ldv, w, out, soo, soi, ldo, inner, n, d);
        AMDS_LAUNCH_CHECK("dwconv_seq_win_kernel");
        return AMDS_OK;
    }
    hipLaunchKernelGGL(dwconv_seq_kernel, dim3(cdiv((long)n * d, 256), outer * inner), dim3(256), 0, (hipStream_t)stream, v, svo, svi, ldv, w,
                       out, soo, soi, ldo, inner, n, d, taps);
    AMDS_LAUNCH_CHECK("dwconv_seq_kernel");
    return AMDS_OK;
}

extern "C" int amds_dwconv_seq_row(const float* v, long svo, long svi, int ldv, const float* w, float* out, long soo, long soi, int outer, int inner, int n, int d,
                                   int taps, int row, void* stream) {
    AMDS_REQUIRE(v && w && out, "amds_dwconv_seq_row: null pointer");
    AMDS_REQUIRE(outer > 0 && inner > 0 && (long)outer * inner <= 65535 && n > 0 && d > 0 && taps > 0 && (taps & 1) && row >= 0 && row < n, "amds_dwconv_seq_row: bad arguments");
    hipLaunchKernelGGL(dwconv_seq_row_kernel, dim3(cdiv(d, 64), outer * inner), dim3(64), 0, (hipStream_t)stream, v, svo, svi, ldv, w, out, soo, soi, inner, n, d, taps, row);
    AMDS_LAUNCH_CHECK("dwconv_seq_row_kernel");
    return AMDS_OK;
}

extern "C" int amds_ppeg(const float* x, float* y, const float* w7, const float* b7, const float* w5, const float* b5, const float* w3,
                         const float* b3, int B, int Hh, int Ww, int C, void* stream) {
    AMDS_REQUIRE(x && y && x != y && w7 && b7 && w5 && b5 && w3 && b3, "amds_ppeg: null/aliased pointer");
    AMDS_REQUIRE(B > 0 && B <= 65535 && Hh > 0 && Ww > 0 && (long)Hh * Ww + 1 <= 65535 && C > 0, "amds_ppeg: bad shape");
    hipLaunchKernelGGL(ppeg_kernel, dim3(cdiv(C, 256), Hh, B), dim3(256), 0, (hipStream_t)stream, x, y, w7, b7, w5, b5, w3, b3, Hh, Ww, C);
    AMDS_LAUNCH_CHECK("ppeg_kernel");
    return AMDS_OK;
}
