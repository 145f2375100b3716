// gemm_rowstream.hip -- weights-stationary GEMM for the narrow layers of the Swin stages 1-2 (K = 96 / 192 / 384,
// N <= 768): out[M,N] = epilogue( [LayerNorm](A)[M,K] * W[N,K]^T ).
//
// Why a second GEMM design: with K <= 192 a block tile's K loop is 2-3 iterations, so the tiled kernels
// (gemm_kernel.h) spend their time in the prologue/epilogue latency of each 128-row tile (measured: 250-350 us for
// the 802816 x {288,384} x 96 layers of CTransPath stage 1 = 2.2-3 TB/s of compulsory traffic, unchanged by a
// coalesced epilogue).  Here the roles are inverted:
//   * a workgroup (8 waves) stages its slice of W ONCE into LDS, already in MFMA fragment order (1 KB per
//     (n-fragment, k-step): lane l reads its 16 bytes at lane*16 -> linear, conflict-free ds_read_b128), and then
//     streams row groups through it (persistent grid, one workgroup per CU); LDS holds nothing else;
//   * the activation never touches LDS: a lane's 16-byte MFMA fragment IS a contiguous piece of one row, so each wave
//     loads its 32*FM rows straight from global memory into registers (whole K);
//   * optional fused LayerNorm: A is the fp32 residual stream; the two lanes (l31, hi=0/1) that share a row hold all
//     K values of it between them -> statistics are an in-lane sum plus one cross-half shuffle; the normalised row
//     is rounded to the operand type in registers.  The separate LayerNorm kernel and its round trip disappear;
//   * MFMA orientation: x is the A operand and W the B operand, so an accumulator register holds ONE ROW and the 32
//     lanes of a half-wave hold 32 CONSECUTIVE COLUMNS: every global access of the epilogue is a contiguous 128-byte
//     row segment per half-wave with no LDS transpose.  For 16-bit outputs two n-fragments are staged with their W rows
//     interleaved (fragment 2p holds the even, 2p+1 the odd columns of a 64-column block), so a lane owns two adjacent
//     columns and stores them as one 32-bit word;
//   * latency: the residual rows (fp32 read-modify-write epilogues) are requested right after the operand rows, before
//     any MFMA, and become the accumulators' initial value; the next row group's operand rows are requested before
//     the current group's epilogue.  Each wave keeps 24-48 KB in flight.
#include "gemm_kernel.h"
#include "rowstream.h"
#include <stdlib.h>

namespace amds {

constexpr int RS_WAVES = 8;

// ---------------------------------------------------------------------------------------------------------------
// 16-bit outputs: out = act(LN?(A) W^T + bias), optional GELU.  nf n-fragments per slice, processed in pairs.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int KS, int FM, int EPI, bool LNF>
__global__ void __launch_bounds__(64 * RS_WAVES) rowstream_f16out_kernel(const void* __restrict__ Aptr, long lda, const T* __restrict__ W,
                                                                         long ldw, int M, int nf, T* __restrict__ out, long ldo,
                                                                         const float* __restrict__ bias, const float* __restrict__ ln_g,
                                                                         const float* __restrict__ ln_b, float eps, int groups) {
    typedef typename Act<T>::vec8 vec8;
    typedef T vec2 __attribute__((ext_vector_type(2)));
    constexpr int K = KS * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int n_base = blockIdx.y * nf * 32;
    const int npair = nf >> 1;
    char* s_w = smem;
    float* s_ln = reinterpret_cast<float*>(smem + (size_t)nf * KS * 1024);
    // W slice -> LDS in fragment order; paired fragments take the even / odd rows of their 64-row block
    for (int blk = wave; blk < nf * KS; blk += RS_WAVES) {
        const int j = blk / KS, ks = blk - j * KS;
        const int wrow = (j < 2 * npair) ? 64 * (j >> 1) + 2 * l31 + (j & 1) : 32 * j + l31;
        glds16(W + (long)(n_base + wrow) * ldw + 16 * ks + 8 * hi, s_w + blk * 1024);
    }
    if constexpr (LNF) {
        for (int i = tid; i < K; i += 64 * RS_WAVES) { s_ln[i] = ln_g[i]; s_ln[K + i] = ln_b[i]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int gstep = gridDim.x * RS_WAVES;
    const int lane_off2 = 4 * hi * (int)ldo + 2 * l31, lane_off1 = 4 * hi * (int)ldo + l31;     // lane-dependent part (elements)
    int g = __builtin_amdgcn_readfirstlane(blockIdx.x * RS_WAVES + wave);
    constexpr bool RAW_PF = LNF && KS * FM <= 6;       // prefetch the next group's fp32 rows only while 2 x raw + xf fit the register file
    f32x4 raw[LNF ? FM : 1][KS][2];
    vec8 xf[FM][KS];
    if (g < groups && (RAW_PF || !LNF)) {
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const int row = min(g * 32 * FM + f * 32 + l31, M - 1);
            if constexpr (LNF) rs_load_raw<KS>(raw[f], reinterpret_cast<const float*>(Aptr), lda, row, hi);
            else rs_load_f16<T, KS>(xf[f], reinterpret_cast<const T*>(Aptr), lda, row, hi);
        }
    }
    for (; g < groups; g += gstep) {
        const int row0 = g * 32 * FM;
        // the W fragments are loop invariant: without an opaque per-iteration offset the compiler hoists all nf*KS
        // ds_reads out of the group loop and spills them
        int lds_lane = lane * 16;
        asm volatile("" : "+v"(lds_lane));
        if constexpr (LNF) {
            if constexpr (!RAW_PF) {
#pragma unroll
                for (int f = 0; f < FM; ++f)
                    rs_load_raw<KS>(raw[f], reinterpret_cast<const float*>(Aptr), lda, min(row0 + f * 32 + l31, M - 1), hi);
            }
#pragma unroll
            for (int f = 0; f < FM; ++f) rs_normalise<T, KS>(xf[f], raw[f], s_ln, hi, eps);
            // request the next group's rows now: they arrive under this group's MFMAs / GELU / stores
            if (RAW_PF && g + gstep < groups) {
#pragma unroll
                for (int f = 0; f < FM; ++f)
                    rs_load_raw<KS>(raw[f], reinterpret_cast<const float*>(Aptr), lda, min((g + gstep) * 32 * FM + f * 32 + l31, M - 1), hi);
            }
        }
        for (int p = 0; p < npair; ++p) {
            f32x16 acc[FM][2];
            const float b0 = bias ? bias[n_base + 64 * p + 2 * l31] : 0.f, b1 = bias ? bias[n_base + 64 * p + 2 * l31 + 1] : 0.f;
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[f][0][r] = b0; acc[f][1][r] = b1; }
            const char* wp = s_w + (size_t)(2 * p) * KS * 1024 + lds_lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 w0 = *reinterpret_cast<const vec8*>(wp + ks * 1024);
                const vec8 w1 = *reinterpret_cast<const vec8*>(wp + (KS + ks) * 1024);
#pragma unroll
                for (int f = 0; f < FM; ++f) {
                    acc[f][0] = Act<T>::mfma32(xf[f][ks], w0, acc[f][0]);
                    acc[f][1] = Act<T>::mfma32(xf[f][ks], w1, acc[f][1]);
                }
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // bound how many W fragments are in flight
            }
#pragma unroll
            for (int f = 0; f < FM; ++f) {
                T* ub = out + (long)(row0 + f * 32) * ldo + n_base + 64 * p;          // wave-uniform part of the address
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    f32x2 v = {acc[f][0][r], acc[f][1][r]};
                    if constexpr (EPI == AMDS_EPI_BIAS_GELU) v = gelu_erf_poly2(v);
                    const vec2 o = {Act<T>::from_f32(v[0]), Act<T>::from_f32(v[1])};
                    const int rr = (r & 3) + 8 * (r >> 2);
                    if (row0 + f * 32 + 4 * hi + rr < M) *reinterpret_cast<vec2*>(ub + (long)rr * ldo + lane_off2) = o;
                }
            }
        }
        if (nf & 1) {                      // trailing unpaired fragment: plain column order, 16-bit stores
            const int j = nf - 1;
            f32x16 acc[FM];
            const float b0 = bias ? bias[n_base + 32 * j + l31] : 0.f;
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] = b0;
            const char* wp = s_w + (size_t)j * KS * 1024 + lds_lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 w0 = *reinterpret_cast<const vec8*>(wp + ks * 1024);
#pragma unroll
                for (int f = 0; f < FM; ++f) acc[f] = Act<T>::mfma32(xf[f][ks], w0, acc[f]);
            }
#pragma unroll
            for (int f = 0; f < FM; ++f) {
                T* ub = out + (long)(row0 + f * 32) * ldo + n_base + 32 * j;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = {acc[f][r], acc[f][r + 1]};
                    if constexpr (EPI == AMDS_EPI_BIAS_GELU) v = gelu_erf_poly2(v);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int rr = ((r + e) & 3) + 8 * ((r + e) >> 2);
                        if (row0 + f * 32 + 4 * hi + rr < M) ub[(long)rr * ldo + lane_off1] = Act<T>::from_f32(v[e]);
                    }
                }
            }
        }
        if constexpr (!LNF) {
            if (g + gstep < groups) {
#pragma unroll
                for (int f = 0; f < FM; ++f)
                    rs_load_f16<T, KS>(xf[f], reinterpret_cast<const T*>(Aptr), lda, min((g + gstep) * 32 * FM + f * 32 + l31, M - 1), hi);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 outputs with all NF n-fragments of the slice live: out (+)= A W^T + bias.  RESIDUAL: the accumulators start
// from the residual rows, requested before the MFMAs.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int KS, int FM, int NF, int EPI, bool PREFETCH>
__global__ void __launch_bounds__(64 * RS_WAVES) rowstream_f32out_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw,
                                                                         int M, float* out, long ldo, const float* __restrict__ bias,
                                                                         int groups) {
    typedef typename Act<T>::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int n_base = blockIdx.y * NF * 32;
    for (int blk = wave; blk < NF * KS; blk += RS_WAVES) {
        const int j = blk / KS, ks = blk - j * KS;
        glds16(W + (long)(n_base + 32 * j + l31) * ldw + 16 * ks + 8 * hi, smem + blk * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float bv[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) bv[j] = bias ? bias[n_base + 32 * j + l31] : 0.f;

    const int gstep = gridDim.x * RS_WAVES;
    int g = __builtin_amdgcn_readfirstlane(blockIdx.x * RS_WAVES + wave);
    const int lane_off = 4 * hi * (int)ldo + l31;
    vec8 xf[FM][KS], xn[PREFETCH ? FM : 1][KS];
    if (g < groups) {
#pragma unroll
        for (int f = 0; f < FM; ++f) rs_load_f16<T, KS>(xf[f], A, lda, min(g * 32 * FM + f * 32 + l31, M - 1), hi);
    }
    for (; g < groups; g += gstep) {
        const int row0 = g * 32 * FM;
        const bool full = row0 + 32 * FM <= M;
        int lds_lane = lane * 16;                         // opaque per iteration: keeps the W-fragment ds_reads inside the loop
        asm volatile("" : "+v"(lds_lane));
        f32x16 acc[FM][NF];
        if constexpr (EPI == AMDS_EPI_RESIDUAL) {
            // unconditional loads on the (usual) full group: a predicated load sits in its own basic block and gets a
            // vmcnt(0) at the join, which serialises the 48-96 requests this kernel lives on
            if (full) {
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float* ub = out + (long)(row0 + f * 32 + (r & 3) + 8 * (r >> 2)) * ldo + n_base + 32 * j;   // wave-uniform
                            acc[f][j][r] = ub[lane_off];
                        }
            } else {
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = min(row0 + f * 32 + 4 * hi + (r & 3) + 8 * (r >> 2), M - 1);
                            acc[f][j][r] = out[(long)row * ldo + n_base + 32 * j + l31];
                        }
            }
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[f][j][r] += bv[j];
        } else {
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[f][j][r] = bv[j];
        }
        if constexpr (PREFETCH) {
            if (g + gstep < groups) {
#pragma unroll
                for (int f = 0; f < FM; ++f) rs_load_f16<T, KS>(xn[f], A, lda, min((g + gstep) * 32 * FM + f * 32 + l31, M - 1), hi);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const vec8 wf = *reinterpret_cast<const vec8*>(smem + (size_t)(j * KS + ks) * 1024 + lds_lane);
#pragma unroll
                for (int f = 0; f < FM; ++f) acc[f][j] = Act<T>::mfma32(xf[f][ks], wf, acc[f][j]);
            }
        }
#pragma unroll
        for (int f = 0; f < FM; ++f) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    float* ub = out + (long)(row0 + f * 32 + rr) * ldo + n_base + 32 * j;
                    if (full || row0 + f * 32 + 4 * hi + rr < M) ub[lane_off] = acc[f][j][r];
                }
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xf[f][ks] = xn[f][ks];
        } else {
            if (g + gstep < groups) {
#pragma unroll
                for (int f = 0; f < FM; ++f) rs_load_f16<T, KS>(xf[f], A, lda, min((g + gstep) * 32 * FM + f * 32 + l31, M - 1), hi);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Whole MLP branch of a 96-channel Swin block in one pass over the residual stream (ctranspath.py:693-695, _Mlp :355-383):
//     x += W2 gelu(W1 LN(x) + b1) + b2          x fp32 [M][96], hidden 384
// Both weight matrices live in LDS in MFMA fragment order (72 KB each).  The 384-wide hidden activation never exists in
// memory: GEMM 1 runs with W1 as the A operand, so its accumulator holds (lane = row, registers = 32 hidden units);
// after bias + GELU the 16 registers, rounded to the operand type, ARE the A operand of GEMM 2 for two k-steps (the
// contraction index of an MFMA may be permuted freely as long as both operands agree, and W2's fragments are staged
// with that permutation) -- A and B operand layouts are symmetric, so GEMM 2 (h as A, W2 as B) ends with
// (lane = output channel, register = row): coalesced 128-byte row segments for the residual read-modify-write.
// HBM traffic: one read and one write of x (616 MB per 256 tiles) instead of 2.2 GB for LN / fc1 / fc2 kernels.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64 * RS_WAVES) swin_mlp96_kernel(float* x, int M, const T* __restrict__ W1, const float* __restrict__ b1,
                                                                   const T* __restrict__ W2, const float* __restrict__ b2,
                                                                   const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                                   float eps, int groups) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int C = 96, H = 384, KS = 6, NJ = H / 32, NC = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w1 = smem;                                   // [NJ][KS][1 KB]
    char* s_w2 = smem + NJ * KS * 1024;                  // [NC][2*NJ][1 KB]
    float* s_ln = reinterpret_cast<float*>(s_w2 + NC * 2 * NJ * 1024);   // gamma[96] beta[96]
    float* s_b1 = s_ln + 2 * C;                          // [384]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    for (int blk = wave; blk < NJ * KS; blk += RS_WAVES) {
        const int j = blk / KS, ks = blk - j * KS;
        glds16(W1 + (long)(32 * j + l31) * C + 16 * ks + 8 * hi, s_w1 + blk * 1024);
    }
    for (int blk = wave; blk < NC * 2 * NJ; blk += RS_WAVES) {
        const int cf = blk / (2 * NJ), hs = blk - cf * 2 * NJ;
        const T* src = W2 + (long)(32 * cf + l31) * H + 16 * hs + 4 * hi;      // slots 0-3: hidden 16hs+4hi.., slots 4-7: +8
        const vec4 lo = *reinterpret_cast<const vec4*>(src), hi4 = *reinterpret_cast<const vec4*>(src + 8);
        vec8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi4[e]; }
        *reinterpret_cast<vec8*>(s_w2 + blk * 1024 + lane * 16) = v;
    }
    for (int i = tid; i < C; i += 64 * RS_WAVES) { s_ln[i] = ln_g[i]; s_ln[C + i] = ln_b[i]; }
    for (int i = tid; i < H; i += 64 * RS_WAVES) s_b1[i] = b1[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float bv[NC];
#pragma unroll
    for (int cf = 0; cf < NC; ++cf) bv[cf] = b2[32 * cf + l31];

    const int gstep = gridDim.x * RS_WAVES;
    const int lane_off = 4 * hi * C + l31;
    int g = __builtin_amdgcn_readfirstlane(blockIdx.x * RS_WAVES + wave);
    f32x4 raw[KS][2];
    vec8 xf[KS];
    for (; g < groups; g += gstep) {
        const int row0 = g * 32;
        const bool full = row0 + 32 <= M;
        int lds_lane = lane * 16;                         // opaque per iteration: keeps the weight-fragment ds_reads inside the loop
        asm volatile("" : "+v"(lds_lane));
        // (prefetching the next group's rows here was measured: the 48 extra live registers spill, 29.7 -> 30.8 ms per 1024 tiles)
        rs_load_raw<KS>(raw, x, C, min(row0 + l31, M - 1), hi);
        rs_normalise<T, KS>(xf, raw, s_ln + (lds_lane & 1), hi, eps);
        // the accumulators of GEMM 2 start from the residual rows (+ b2); requested before any MFMA
        f32x16 acc2[NC];
        if (full) {
#pragma unroll
            for (int cf = 0; cf < NC; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[cf][r] = (x + (long)(row0 + (r & 3) + 8 * (r >> 2)) * C + 32 * cf)[lane_off];
        } else {
#pragma unroll
            for (int cf = 0; cf < NC; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[cf][r] = x[(long)min(row0 + 4 * hi + (r & 3) + 8 * (r >> 2), M - 1) * C + 32 * cf + l31];
        }
#pragma unroll
        for (int cf = 0; cf < NC; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[cf][r] += bv[cf];
#pragma unroll 1
        for (int j = 0; j < NJ; ++j) {
            f32x16 acc1;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + 32 * j + 8 * g4 + 4 * hi + (lds_lane & 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[4 * g4 + e] = bb[e];
            }
            const char* w1p = s_w1 + (size_t)j * KS * 1024 + lds_lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc1 = Act<T>::mfma32(*reinterpret_cast<const vec8*>(w1p + ks * 1024), xf[ks], acc1);
            vec8 hf[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v = gelu_erf_poly2(f32x2{acc1[r], acc1[r + 1]});
                hf[r >> 3][r & 7] = Act<T>::from_f32(v[0]);
                hf[r >> 3][(r & 7) + 1] = Act<T>::from_f32(v[1]);
            }
            const char* w2p = s_w2 + (size_t)(2 * j) * 1024 + lds_lane;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int cf = 0; cf < NC; ++cf)
                    acc2[cf] = Act<T>::mfma32(hf[s2], *reinterpret_cast<const vec8*>(w2p + (size_t)(cf * 2 * NJ + s2) * 1024), acc2[cf]);
        }
#pragma unroll
        for (int cf = 0; cf < NC; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float* ub = x + (long)(row0 + rr) * C + 32 * cf;
                if (full || row0 + 4 * hi + rr < M) ub[lane_off] = acc2[cf][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MLP branch of a 192-channel Swin block (stage 2) in one pass over the residual stream, weights STREAMED through LDS:
//     x += W2 gelu(W1 LN(x) + b1) + b2          x fp32 [M][192], hidden 768
// Same register-level chaining as swin_mlp96_kernel (the GELU'd fc1 accumulator is the A operand of fc2), but the two
// weight matrices (590 KB) do not fit LDS: a workgroup (4 waves x 32 rows = 128 rows) walks the hidden dimension in 12
// chunks of 64 units; chunk c of both matrices is one contiguous 48 KB block of the pre-packed weight image
// (amds_swin_mlp_pack: MFMA fragment order, fc2 with the k-permutation baked in), copied by LDS-DMA into a double buffer
// two chunks ahead of its use (three buffers) -- one barrier per chunk.
// ---------------------------------------------------------------------------------------------------------------
constexpr int MS_CHUNK_BYTES = 49152;      // 2 fc1 fragments x 12 k-steps + 6 channel fragments x 4 k-steps, 1 KB each

constexpr int MS_WAVES = 4;                // one wave per SIMD: the 96 + 48 + 16 accumulator / operand registers of a row group need > 256

template <typename T>
__global__ void __launch_bounds__(64 * MS_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1)))
swin_mlp192_kernel(float* x, int M, const char* __restrict__ wpack, const float* __restrict__ b1,
                                                                    const float* __restrict__ b2, const float* __restrict__ ln_g,
                                                                    const float* __restrict__ ln_b, float eps, int nblocks) {
    typedef typename Act<T>::vec8 vec8;
    constexpr int C = 192, H = 768, KS = 12, NC = 6, NCH = H / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                    // [3][48 KB]
    float* s_ln = reinterpret_cast<float*>(smem + 3 * MS_CHUNK_BYTES);   // gamma[192] beta[192]
    float* s_b1 = s_ln + 2 * C;                                          // [768]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < C; i += 64 * MS_WAVES) { s_ln[i] = ln_g[i]; s_ln[C + i] = ln_b[i]; }
    for (int i = tid; i < H; i += 64 * MS_WAVES) s_b1[i] = b1[i];
    float bv[NC];
#pragma unroll
    for (int cf = 0; cf < NC; ++cf) bv[cf] = b2[32 * cf + l31];
    // every wave copies 12 of the 48 1-KB pieces of a chunk
    auto load_chunk = [&](int hc, int buf) {
        const char* src = wpack + (size_t)hc * MS_CHUNK_BYTES;
        char* dst = s_w + buf * MS_CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < 48 / MS_WAVES; ++i) {
            const int piece = wave * (48 / MS_WAVES) + i;
            glds16(src + piece * 1024 + lane * 16, dst + piece * 1024);
        }
    };
    const int lane_off = 4 * hi * C + l31;
    // three LDS buffers: chunk c+2 is requested while chunk c is consumed (two chunks = ~4 us of latency cover)
    int buf = 0;
    load_chunk(0, 0);
    load_chunk(1, 1);
    for (int rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
        const int row0 = rb * (32 * MS_WAVES) + wave * 32;
        const bool full = row0 + 32 <= M;
        int lds_lane = lane * 16;
        asm volatile("" : "+v"(lds_lane));
        vec8 xf[KS];
        {
            f32x4 raw[KS][2];
            rs_load_raw<KS>(raw, x, C, min(row0 + l31, M - 1), hi);
            rs_normalise<T, KS>(xf, raw, s_ln + (lds_lane & 1), hi, eps);      // opaque 0: gamma / beta reads are loop invariant and
                                                                               // would otherwise be hoisted (192 registers) and spilled
        }
        __builtin_amdgcn_sched_barrier(0);      // the 96 raw registers are dead before the 96 residual loads are requested
        f32x16 acc2[NC];
        if (full) {
#pragma unroll
            for (int cf = 0; cf < NC; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[cf][r] = (x + (long)(row0 + (r & 3) + 8 * (r >> 2)) * C + 32 * cf)[lane_off];
        } else {
#pragma unroll
            for (int cf = 0; cf < NC; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[cf][r] = x[(long)min(row0 + 4 * hi + (r & 3) + 8 * (r >> 2), M - 1) * C + 32 * cf + l31];
        }
#pragma unroll
        for (int cf = 0; cf < NC; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[cf][r] += bv[cf];
#pragma unroll 1
        for (int hc = 0; hc < NCH; ++hc) {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // this wave's 12 pieces of chunk hc have landed (chunk hc+1 may be in flight) ...
            __syncthreads();                                     // ... and everybody's; every wave is done with the buffer of chunk hc-1
            const int nxt = hc + 2 < NCH ? hc + 2 : hc + 2 - NCH;
            const int b2 = buf == 0 ? 2 : buf - 1;               // (buf + 2) % 3 = the buffer chunk hc-1 occupied
            if (hc + 2 < NCH || rb + gridDim.x < nblocks) load_chunk(nxt, b2);   // wraps into the next row block's first chunks
            else asm volatile("s_nop 0");
            const char* wb = s_w + buf * MS_CHUNK_BYTES + lds_lane;
            // both hidden fragments of the chunk at once: two independent MFMA chains (one wave per SIMD: a single chain
            // would expose the full MFMA latency at every k-step)
            f32x16 acc1[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(s_b1 + 64 * hc + 32 * jj + 8 * g4 + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc1[jj][4 * g4 + e] = bb[e];
                }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc1[0] = Act<T>::mfma32(*reinterpret_cast<const vec8*>(wb + ks * 1024), xf[ks], acc1[0]);
                acc1[1] = Act<T>::mfma32(*reinterpret_cast<const vec8*>(wb + (KS + ks) * 1024), xf[ks], acc1[1]);
                if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            vec8 hf[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = gelu_erf_poly2(f32x2{acc1[jj][r], acc1[jj][r + 1]});
                    hf[jj][r >> 3][r & 7] = Act<T>::from_f32(v[0]);
                    hf[jj][r >> 3][(r & 7) + 1] = Act<T>::from_f32(v[1]);
                }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int cf = 0; cf < NC; ++cf)
                        acc2[cf] = Act<T>::mfma32(hf[jj][s2], *reinterpret_cast<const vec8*>(wb + 24576 + (cf * 4 + 2 * jj + s2) * 1024), acc2[cf]);
            buf = buf == 2 ? 0 : buf + 1;
        }
#pragma unroll
        for (int cf = 0; cf < NC; ++cf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float* ub = x + (long)(row0 + rr) * C + 32 * cf;
                if (full || row0 + 4 * hi + rr < M) ub[lane_off] = acc2[cf][r];
            }
    }
}

// weight image for swin_mlp192_kernel: 12 chunks of 48 KB = [fc1: 2 fragments x 12 k-steps][fc2: 6 channel fragments x 4 k-steps],
// every fragment 64 lanes x 8 elements in MFMA register order.
template <typename T>
__global__ void swin_mlp192_pack_kernel(const T* __restrict__ w1, const T* __restrict__ w2, T* __restrict__ out) {
    constexpr int C = 192, H = 768;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // one 8-element lane slot
    if (idx >= 12 * 48 * 64) return;
    const int lane = idx & 63, piece = (idx >> 6) % 48, hc = idx / (48 * 64);
    const int l31 = lane & 31, hi = lane >> 5;
    T* dst = out + (size_t)idx * 8;
    if (piece < 24) {                       // fc1: rows = hidden units, k = input channels
        const int jj = piece / 12, ks = piece - jj * 12;
        const T* src = w1 + (size_t)(64 * hc + 32 * jj + l31) * C + 16 * ks + 8 * hi;
        for (int e = 0; e < 8; ++e) dst[e] = src[e];
    } else {                                // fc2 as B operand: lane = output channel, slots = hidden units (permuted)
        const int q = piece - 24, cf = q >> 2, s4 = q & 3;
        const T* src = w2 + (size_t)(32 * cf + l31) * H + 64 * hc + 16 * s4 + 4 * hi;
        for (int e = 0; e < 4; ++e) { dst[e] = src[e]; dst[4 + e] = src[8 + e]; }
    }
}

static int rs_slices(int nfrag, int KS) {
    int slices = 1;
    while ((nfrag % slices) != 0 || (size_t)(nfrag / slices) * KS * 1024 > 73728) ++slices;
    return slices;
}

template <typename T, int KS, int FM, int EPI, bool LNF>
static int launch_rs_f16(const void* A, long lda, const void* W, long ldw, int M, int N, void* out, long ldo, const float* bias,
                         const float* ln_g, const float* ln_b, float eps, hipStream_t st) {
    constexpr int K = KS * 16;
    const int nfrag = N / 32, slices = rs_slices(nfrag, KS), nf = nfrag / slices;
    const size_t lds = (size_t)nf * KS * 1024 + (LNF ? 2 * K * 4 : 0);
    auto kern = rowstream_f16out_kernel<T, KS, FM, EPI, LNF>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int groups = cdiv(M, 32 * FM);
    int gx = cdiv(groups, RS_WAVES);
    const int cap = 256 / slices > 0 ? 256 / slices : 1;            // one resident workgroup per CU over all slices
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(64 * RS_WAVES), lds, st, A, lda, reinterpret_cast<const T*>(W), ldw, M, nf,
                       reinterpret_cast<T*>(out), ldo, bias, ln_g, ln_b, eps, groups);
    AMDS_LAUNCH_CHECK("rowstream_f16out_kernel");
    return AMDS_OK;
}

template <typename T, int KS, int FM, int NF, int EPI, bool PREFETCH>
static int launch_rs_f32(const void* A, long lda, const void* W, long ldw, int M, int N, void* out, long ldo, const float* bias,
                         hipStream_t st) {
    const int slices = N / (32 * NF);
    const size_t lds = (size_t)NF * KS * 1024;
    auto kern = rowstream_f32out_kernel<T, KS, FM, NF, EPI, PREFETCH>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int groups = cdiv(M, 32 * FM);
    int gx = cdiv(groups, RS_WAVES);
    const int cap = 256 / slices > 0 ? 256 / slices : 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(64 * RS_WAVES), lds, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, reinterpret_cast<float*>(out), ldo, bias, groups);
    AMDS_LAUNCH_CHECK("rowstream_f32out_kernel");
    return AMDS_OK;
}

template <typename T>
static int rowstream_dispatch(const void* A, long lda, bool lnf, const void* W, long ldw, int M, int N, int K, int epi, void* out, long ldo,
                              const float* bias, const float* ln_g, const float* ln_b, float eps, hipStream_t st) {
    const bool f16out = epi == AMDS_EPI_BIAS || epi == AMDS_EPI_BIAS_GELU;
    if (f16out && lnf) {
        if (K == 96 && epi == AMDS_EPI_BIAS) return launch_rs_f16<T, 6, 1, AMDS_EPI_BIAS, true>(A, lda, W, ldw, M, N, out, ldo, bias, ln_g, ln_b, eps, st);
        if (K == 96) return launch_rs_f16<T, 6, 1, AMDS_EPI_BIAS_GELU, true>(A, lda, W, ldw, M, N, out, ldo, bias, ln_g, ln_b, eps, st);
        if (K == 192 && epi == AMDS_EPI_BIAS) return launch_rs_f16<T, 12, 1, AMDS_EPI_BIAS, true>(A, lda, W, ldw, M, N, out, ldo, bias, ln_g, ln_b, eps, st);
        if (K == 192) return launch_rs_f16<T, 12, 1, AMDS_EPI_BIAS_GELU, true>(A, lda, W, ldw, M, N, out, ldo, bias, ln_g, ln_b, eps, st);
    }
    if (!lnf && (epi == AMDS_EPI_RESIDUAL || epi == AMDS_EPI_BIAS_F32)) {
#define RS_F32(KV, KSV, FMV, NFV, PF)                                                                                               \
    if (K == KV && N % (32 * NFV) == 0) {                                                                                           \
        if (epi == AMDS_EPI_RESIDUAL) return launch_rs_f32<T, KSV, FMV, NFV, AMDS_EPI_RESIDUAL, PF>(A, lda, W, ldw, M, N, out, ldo, bias, st); \
        return launch_rs_f32<T, KSV, FMV, NFV, AMDS_EPI_BIAS_F32, PF>(A, lda, W, ldw, M, N, out, ldo, bias, st);                    \
    }
        RS_F32(96, 6, 2, 3, true)          // stage-1 proj: N = 96
        RS_F32(192, 12, 1, 3, true)        // stage-2 proj: N = 192, two slices
        RS_F32(384, 24, 1, 3, false)       // stage-1 fc2 (N = 96) and the first patch-merging reduction (N = 192, two slices)
#undef RS_F32
    }
    set_error("amds_gemm_rowstream: no kernel for K=%d N=%d epi=%d fused_ln=%d", K, N, epi, (int)lnf);
    return AMDS_ERR_INVALID;
}

}  // namespace amds

using namespace amds;

extern "C" int amds_gemm_rowstream(const void* A, long lda, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* W,
                                   long ldw, int M, int N, int K, int dtype, int epi, void* out, long ldo, const float* bias,
                                   void* stream) {
    AMDS_REQUIRE(A && W && out, "amds_gemm_rowstream: null pointer");
    AMDS_REQUIRE(M >= 0 && N > 0 && N % 32 == 0, "amds_gemm_rowstream: bad shape M=%d N=%d", M, N);
    AMDS_REQUIRE(K == 96 || K == 192 || K == 384, "amds_gemm_rowstream: K=%d must be 96, 192 or 384", K);
    AMDS_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "amds_gemm_rowstream: gamma and beta go together");
    const bool lnf = ln_gamma != nullptr;
    AMDS_REQUIRE(lda >= K && lda % (lnf ? 4 : 8) == 0 && ldw >= K && ldw % 8 == 0 && ldo % 2 == 0, "amds_gemm_rowstream: bad strides");
    AMDS_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 3) == 0, "amds_gemm_rowstream: misaligned pointers");
    if (M == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * M * (double)N * K, st);
    if (dtype == AMDS_F16) return rowstream_dispatch<f16>(A, lda, lnf, W, ldw, M, N, K, epi, out, ldo, bias, ln_gamma, ln_beta, ln_eps, st);
    if (dtype == AMDS_BF16) return rowstream_dispatch<bf16>(A, lda, lnf, W, ldw, M, N, K, epi, out, ldo, bias, ln_gamma, ln_beta, ln_eps, st);
    set_error("amds_gemm_rowstream: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

extern "C" int amds_swin_mlp96(float* x, int M, const void* fc1_w, const float* fc1_b, const void* fc2_w, const float* fc2_b,
                               const float* ln_gamma, const float* ln_beta, float ln_eps, int dtype, void* stream) {
    AMDS_REQUIRE(x && fc1_w && fc1_b && fc2_w && fc2_b && ln_gamma && ln_beta, "amds_swin_mlp96: null pointer");
    AMDS_REQUIRE(M >= 0, "amds_swin_mlp96: bad M=%d", M);
    AMDS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)fc1_w & 15) == 0 && ((uintptr_t)fc2_w & 15) == 0, "amds_swin_mlp96: misaligned pointers");
    if (M == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(12 * 6 + 3 * 24) * 1024 + (2 * 96 + 384) * 4;
    const int groups = cdiv(M, 32);
    int gx = cdiv(groups, RS_WAVES);
    if (gx > 256) gx = 256;
    ProfScope prof(PROF_GEMM, 4.0 * M * 96.0 * 384.0, st);
#define MLP96_LAUNCH(T)                                                                                                              \
    do {                                                                                                                             \
        static bool attr_set = false;                                                                                                \
        if (!attr_set) {                                                                                                             \
            AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp96_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                                                         \
        }                                                                                                                            \
        hipLaunchKernelGGL((swin_mlp96_kernel<T>), dim3(gx), dim3(64 * RS_WAVES), lds, st, x, M, reinterpret_cast<const T*>(fc1_w), fc1_b, \
                           reinterpret_cast<const T*>(fc2_w), fc2_b, ln_gamma, ln_beta, ln_eps, groups);                             \
    } while (0)
    if (dtype == AMDS_F16) MLP96_LAUNCH(f16);
    else if (dtype == AMDS_BF16) MLP96_LAUNCH(bf16);
    else { set_error("amds_swin_mlp96: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
#undef MLP96_LAUNCH
    AMDS_LAUNCH_CHECK("swin_mlp96_kernel");
    return AMDS_OK;
}

extern "C" int amds_swin_mlp192_pack(const void* fc1_w, const void* fc2_w, void* packed, int dtype, void* stream) {
    AMDS_REQUIRE(fc1_w && fc2_w && packed, "amds_swin_mlp192_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int n = 12 * 48 * 64;
    if (dtype == AMDS_F16) hipLaunchKernelGGL((swin_mlp192_pack_kernel<f16>), dim3(cdiv(n, 256)), dim3(256), 0, st, (const f16*)fc1_w, (const f16*)fc2_w, (f16*)packed);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((swin_mlp192_pack_kernel<bf16>), dim3(cdiv(n, 256)), dim3(256), 0, st, (const bf16*)fc1_w, (const bf16*)fc2_w, (bf16*)packed);
    else { set_error("amds_swin_mlp192_pack: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("swin_mlp192_pack_kernel");
    return AMDS_OK;
}

extern "C" int amds_swin_mlp192(float* x, int M, const void* packed_w, const float* fc1_b, const float* fc2_b, const float* ln_gamma,
                                const float* ln_beta, float ln_eps, int dtype, void* stream) {
    AMDS_REQUIRE(x && packed_w && fc1_b && fc2_b && ln_gamma && ln_beta, "amds_swin_mlp192: null pointer");
    AMDS_REQUIRE(M >= 0, "amds_swin_mlp192: bad M=%d", M);
    AMDS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "amds_swin_mlp192: misaligned pointers");
    if (M == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 3 * MS_CHUNK_BYTES + (2 * 192 + 768) * 4;
    const int nblocks = cdiv(M, 32 * MS_WAVES);
    const int gx = nblocks < 256 ? nblocks : 256;
    ProfScope prof(PROF_GEMM, 4.0 * M * 192.0 * 768.0, st);
#define MLP192_LAUNCH(T)                                                                                                              \
    do {                                                                                                                              \
        static bool attr_set = false;                                                                                                 \
        if (!attr_set) {                                                                                                              \
            AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(swin_mlp192_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((swin_mlp192_kernel<T>), dim3(gx), dim3(64 * MS_WAVES), lds, st, x, M, reinterpret_cast<const char*>(packed_w), fc1_b, fc2_b, \
                           ln_gamma, ln_beta, ln_eps, nblocks);                                                                       \
    } while (0)
    if (dtype == AMDS_F16) MLP192_LAUNCH(f16);
    else if (dtype == AMDS_BF16) MLP192_LAUNCH(bf16);
    else { set_error("amds_swin_mlp192: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
#undef MLP192_LAUNCH
    AMDS_LAUNCH_CHECK("swin_mlp192_kernel");
    return AMDS_OK;
}
