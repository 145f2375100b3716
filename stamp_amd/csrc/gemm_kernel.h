// gemm_kernel.h -- C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, MFMA 32x32x16 (f16/bf16 in,
// fp32 accumulate) for gfx950.
//
// Structure (v1, "one barrier per K-step"):
//   * block tile BM x BN x 64, WM x WN waves of 64 lanes, each wave owns a (BM/WM) x (BN/WN) sub-tile
//     made of 32x32 MFMA fragments;
//   * both operands are K-contiguous (torch Linear weight is [out][in]), so A and W tiles are staged
//     identically: async global->LDS copies (global_load_lds_dwordx4, no VGPR round trip) into a
//     double-buffered LDS image of 128-byte rows;
//   * the LDS image is XOR-swizzled at 16-byte granularity (chunk ^= (row>>1)&7).  global_load_lds
//     writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address (a permutation
//     inside one 128-B line: still one fully-coalesced line per 8 lanes) and again on the ds_read_b128;
//     a wave's fragment read (32 rows x 16 B) then touches 16 distinct 16-B slots per lane group:
//     conflict free;
//   * the MFMA is issued with W as the "A" operand and the activation as the "B" operand, i.e. it
//     computes C^T fragments: lane (j = lane&31, hi = lane>>5) holds output row m = j and the 4
//     consecutive columns n = 8*(r>>2) + 4*hi + (r&3), so the epilogue does 16-byte fp32 / 8-byte
//     fp16 accesses per lane and per-column vectors (bias, LayerScale) are float4 loads;
//   * blockIdx -> tile map is XCD-aware: each of the 8 XCDs (private 4 MiB L2) gets a contiguous
//     range of tiles, walked in groups of GROUP_M row-tiles x all column-tiles so that the ~64
//     co-resident tiles of an XCD share A and W panels through its L2.
#pragma once
#include <type_traits>
#include "common.h"

namespace amds {

struct EpiArgs {
    void* out;
    long ldo;
    const float* bias;
    const float* scale;
    const float* pos;
    int np, T, P;
    float acc_scale;
    // batched / split-K launches (gridDim.y = batch count): element strides added per batch index
    long bsA = 0, bsW = 0, bsOut = 0;
    int nbatch = 1;
    // LayerNorm folded into the GEMMs around it (gemm_4w16.h only; amds_gemm_lnfold):
    //   producer (RESIDUAL): xh = 16-bit copy of the updated rows (pitch ldo elements), rowpart = [M][N/128][2] partial (sum, sum of squares)
    //   consumer (BIAS / BIAS_GELU / SWIGLU): rowstat = [M][2] (rstd, -mean*rstd), colsum = [N] sum_k W'[n][k]:
    //       out = act( acc * rstd[m] + (colsum[n] * (-mean*rstd)[m] + bias[n]) )
    void* xh = nullptr;
    // residual stream as a 16-bit (hi | lo) pair of planes (amds_gemm_lnfold_planes): x = xh + xl is read, updated and written back as
    // xh = round16(x), xl = round16(x - xh); no fp32 rows are touched (out is not read or written)
    void* xl = nullptr;
    float* rowpart = nullptr;
    const float* rowstat = nullptr;
    const float* colsum = nullptr;
    // token-major operands (gemm_4w16.h TN, kernel id 15): number of token rows that exist in A / W counted from batch 0's first row; batch b contracts
    // rows b K .. b K + K - 1 and those >= ktot read as zeros (0 = all K rows of every batch exist)
    long ktot = 0;
};

template <int EPI>
constexpr bool epi_is_staged() {
    return EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU || EPI == AMDS_EPI_RESIDUAL ||
           EPI == AMDS_EPI_BIAS_F32 || EPI == AMDS_EPI_BIAS_GELU_F32 || EPI == AMDS_EPI_BIAS_RELU_F32;
}

// Per-column epilogue vectors (bias, LayerScale) of G groups of 4 consecutive columns, loaded ONCE and up front.  Loaded
// inside the value loop, every one of these 16-byte loads is followed by its own s_waitcnt vmcnt(0): a serial L2 round
// trip per 4 outputs, 32 per wave and 256 x 256 tile = 7-9 us of every tile's ~9 us epilogue.
template <int G>
struct EpiCols {
    f32x4 bias[G], scale[G];
};
template <int EPI, int G, typename F>
__device__ __forceinline__ void epi_cols_load(const EpiArgs& ep, EpiCols<G>& c, F col_of) {
    if (ep.bias) {
#pragma unroll
        for (int q = 0; q < G; ++q) c.bias[q] = *reinterpret_cast<const f32x4*>(ep.bias + col_of(q));
    }
    if constexpr (EPI == AMDS_EPI_RESIDUAL) {
        if (ep.scale) {
#pragma unroll
            for (int q = 0; q < G; ++q) c.scale[q] = *reinterpret_cast<const f32x4*>(ep.scale + col_of(q));
        }
    }
}

// value transform shared by all staged epilogues: bias, activation, LayerScale (residual)
template <int EPI>
__device__ __forceinline__ f32x4 epi_value(const EpiArgs& ep, const f32x4& bias, const f32x4& scale, f32x4 v) {
    if (ep.acc_scale != 1.0f) v *= ep.acc_scale;
    if (ep.bias) v += bias;
    if constexpr (EPI == AMDS_EPI_BIAS_GELU) {
        const f32x2 a = gelu_erf_poly2(f32x2{v[0], v[1]}), b = gelu_erf_poly2(f32x2{v[2], v[3]});
        v = f32x4{a[0], a[1], b[0], b[1]};
    }
    if constexpr (EPI == AMDS_EPI_BIAS_GELU_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    }
    if constexpr (EPI == AMDS_EPI_BIAS_RELU || EPI == AMDS_EPI_BIAS_RELU_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if constexpr (EPI == AMDS_EPI_RESIDUAL) {
        if (ep.scale) v *= scale;
    }
    return v;
}

// Two column groups at once for the 16-bit-output epilogues: the GELU polynomial runs as four interleaved chains.
// FAST = (bias present, acc_scale == 1): decided once per kernel by a uniform branch around the value loop; left inside it the
// two tests become v_cndmask selects (12 of ~70 VALU instructions per 4 outputs).
// NOBIAS (with FAST): the bias is already in the accumulators (gemm_4w16.h initialises them with it).
template <int EPI, bool FAST, bool NOBIAS = false>
__device__ __forceinline__ void epi_value_pair(const EpiArgs& ep, const f32x4& b0, const f32x4& s0, const f32x4& b1, const f32x4& s1,
                                               f32x4& v0, f32x4& v1) {
    if constexpr (FAST && EPI == AMDS_EPI_BIAS_GELU) {
        if constexpr (!NOBIAS) {
            v0 += b0;
            v1 += b1;
        }
        f32x2 q[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
        gelu_erf_poly2_n<4>(q);
        v0 = f32x4{q[0][0], q[0][1], q[1][0], q[1][1]};
        v1 = f32x4{q[2][0], q[2][1], q[3][0], q[3][1]};
    } else if constexpr (FAST && (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_RELU)) {
        if constexpr (!NOBIAS) {
            v0 += b0;
            v1 += b1;
        }
        if constexpr (EPI == AMDS_EPI_BIAS_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
        }
    } else {
        v0 = epi_value<EPI>(ep, b0, s0, v0);
        v1 = epi_value<EPI>(ep, b1, s1, v1);
    }
}

template <int EPI, typename T>
__device__ __forceinline__ void epilogue4(const EpiArgs& ep, int m, int n, float v0, float v1, float v2,
                                          float v3) {
    typedef typename Act<T>::vec4 vec4;
    if (ep.acc_scale != 1.0f) { v0 *= ep.acc_scale; v1 *= ep.acc_scale; v2 *= ep.acc_scale; v3 *= ep.acc_scale; }
    if constexpr (EPI != AMDS_EPI_SWIGLU) {
        if (ep.bias) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(ep.bias + n);
            v0 += b[0]; v1 += b[1]; v2 += b[2]; v3 += b[3];
        }
    }
    if constexpr (EPI == AMDS_EPI_BIAS_GELU) {
        v0 = gelu_erf_fast(v0); v1 = gelu_erf_fast(v1); v2 = gelu_erf_fast(v2); v3 = gelu_erf_fast(v3);
    }
    if constexpr (EPI == AMDS_EPI_BIAS_GELU_F32) {
        v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
    }
    if constexpr (EPI == AMDS_EPI_BIAS_RELU || EPI == AMDS_EPI_BIAS_RELU_F32) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
    }
    if constexpr (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU) {
        vec4 o;
        o[0] = Act<T>::from_f32(v0); o[1] = Act<T>::from_f32(v1);
        o[2] = Act<T>::from_f32(v2); o[3] = Act<T>::from_f32(v3);
        *reinterpret_cast<vec4*>(reinterpret_cast<T*>(ep.out) + (long)m * ep.ldo + n) = o;
    } else if constexpr (EPI == AMDS_EPI_BIAS_F32 || EPI == AMDS_EPI_BIAS_GELU_F32 ||
                         EPI == AMDS_EPI_BIAS_RELU_F32) {
        f32x4 o = {v0, v1, v2, v3};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)m * ep.ldo + n) = o;
    } else if constexpr (EPI == AMDS_EPI_RESIDUAL) {
        if (ep.scale) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(ep.scale + n);
            v0 *= s[0]; v1 *= s[1]; v2 *= s[2]; v3 *= s[3];
        }
        f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)m * ep.ldo + n);
        f32x4 x = *p;
        x[0] += v0; x[1] += v1; x[2] += v2; x[3] += v3;
        *p = x;
    } else if constexpr (EPI == AMDS_EPI_PATCH) {
        const int b = m / ep.np, pi = m - b * ep.np;
        const f32x4 pe = *reinterpret_cast<const f32x4*>(ep.pos + (long)pi * ep.ldo + n);
        f32x4 o = {v0 + pe[0], v1 + pe[1], v2 + pe[2], v3 + pe[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) +
                                  ((long)b * ep.T + ep.P + pi) * ep.ldo + n) = o;
    }
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_tn_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K,
               EpiArgs ep, int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int NT = 64 * WM * WN;
    constexpr int BK = 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 32, FN = WTN / 32;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    constexpr int A_ITERS = BM * 8 / NT, W_ITERS = BN * 8 / NT;
    constexpr int GROUP_M = 8;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");
    static_assert(EPI != AMDS_EPI_SWIGLU || (FN % 2 == 0), "swiglu needs gate/value fragment pairs");
    constexpr int PAIRS = FM * FN < FM + FN ? FM * FN : FM + FN;     // MFMA/ds_read pairs interleaved per k-step

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- XCD-aware tile map -----------------------------------------------------------------
    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int group = GROUP_M * tiles_n;
        const int g = t / group, first_m = g * GROUP_M;
        const int gm = min(tiles_m - first_m, GROUP_M);
        const int rr = t - g * group;
        tm = first_m + rr % gm;
        tn = rr / gm;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: buffer-form LDS-DMA (buffer_load_dwordx4 ... offen lds): one constant 32-bit byte offset per
    // piece, the K advance in the scalar offset (no 64-bit vector address arithmetic in the K loop: worth 9 % of the loop
    // rate on the 256x256 kernel); rows past M are out of range and read as 0 ---------------------------------------
    const int rows_a = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(A + (long)m0 * lda), 0, (int)((((long)rows_a - 1) * lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(W + (long)n0 * ldw), 0, (int)(((long)(BN - 1) * ldw + K) * 2), 0x00020000);
    int aoff[A_ITERS], woff[W_ITERS];
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
        const int c = it * NT + tid, row = c >> 3, cp = c & 7, sc = cp ^ ((row >> 1) & 7);
        aoff[it] = (int)(((long)row * lda + sc * 8) * 2);
    }
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
        const int c = it * NT + tid, row = c >> 3, cp = c & 7, sc = cp ^ ((row >> 1) & 7);
        woff[it] = (int)(((long)row * ldw + sc * 8) * 2);
    }
    auto stage_load = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE;
        char* sw = sa + A_BYTES;
        const int koff = kt * BK * 2;
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it)
            bufl16(rsrc_a, sa + (it * NT + wave * 64) * 16, aoff[it], koff);
#pragma unroll
        for (int it = 0; it < W_ITERS; ++it)
            bufl16(rsrc_w, sw + (it * NT + wave * 64) * 16, woff[it], koff);
    };

    // ---- fragment read offsets ---------------------------------------------------------------
    const int wm = wave / WN, wn = wave % WN;
    const int swz = (l31 >> 1) & 7;
    const int a_row_off = (wm * WTM + l31) * 128;
    const int w_row_off = A_BYTES + (wn * WTN + l31) * 128;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    stage_load(0, 0);
    vec8 af[2][FM], wf[2][FN];
    auto load_frags = [&](const char* sb, int ks, int set) {
        const int coff = ((ks * 2 + hi) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < FM; ++i)
            af[set][i] = *reinterpret_cast<const vec8*>(sb + a_row_off + i * 32 * 128 + coff);
#pragma unroll
        for (int j = 0; j < FN; ++j)
            wf[set][j] = *reinterpret_cast<const vec8*>(sb + w_row_off + j * 32 * 128 + coff);
    };
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_load((kt + 1) & 1, kt + 1);
        const char* sb = smem + (kt & 1) * STAGE;
        // register double-buffered fragments: the ds_reads of k-step ks+1 are in flight under the MFMAs of ks
        load_frags(sb, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frags(sb, ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = Act<T>::mfma32(wf[ks & 1][j], af[ks & 1][i], acc[i][j]);
            if (ks + 1 < 4) {
                // interleave: one ds_read per MFMA for the first FM+FN MFMAs of this k-step
#pragma unroll
                for (int r = 0; r < PAIRS; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
                if constexpr (FM * FN > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - PAIRS, 0);
                if constexpr (FM + FN > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, FM + FN - PAIRS, 0);
            }
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    if constexpr (epi_is_staged<EPI>()) {
        // LDS-staged: the accumulator layout gives every lane a different row (8/16-byte pieces, 32 partial lines
        // per store instruction: ~2 TB/s); bounce the tile through LDS (odd chunk pitch -> conflict-free column
        // writes) and store whole rows, 16 bytes per lane.
        constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
        constexpr int CPR = BN * (F16OUT ? 2 : 4) / 16, PITCH = (CPR | 1) * 16;     // 16-byte chunks per row; row pitch in bytes
        EpiCols<FN * 4> cols;
        epi_cols_load<EPI>(ep, cols, [&](int q) { return n0 + wn * WTN + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi; });
        __syncthreads();                                   // every wave is done with the K-loop stages
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = wm * WTM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * WTN + j * 32 + 8 * g + 4 * hi;
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    v = epi_value<EPI>(ep, cols.bias[j * 4 + g], cols.scale[j * 4 + g], v);
                    if constexpr (F16OUT) {
                        vec4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = Act<T>::from_f32(v[e]);
                        *reinterpret_cast<vec4*>(smem + row * PITCH + nl * 2) = o;
                    } else {
                        *reinterpret_cast<f32x4*>(smem + row * PITCH + nl * 4) = v;
                    }
                }
        }
        __syncthreads();
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR, c = idx - row * CPR;
            if (m0 + row < M) {
                if constexpr (F16OUT) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * PITCH + c * 16);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + c * 8) = v;
                } else {
                    f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * PITCH + c * 16);
                    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + c * 4);
                    if constexpr (EPI == AMDS_EPI_RESIDUAL) v += *p;
                    *p = v;
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WTM + i * 32 + l31;
        if (m < M) {
            if constexpr (EPI == AMDS_EPI_SWIGLU) {
#pragma unroll
                for (int j = 0; j < FN; j += 2) {
                    // fragment j = gate block, j+1 = value block of hidden units
                    const int hbase = (n0 + wn * WTN + j * 32) / 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int hcol = hbase + 8 * g + 4 * hi;
                        float gte[4], val[4];
                        const f32x4 bg = *reinterpret_cast<const f32x4*>(
                            ep.bias + n0 + wn * WTN + j * 32 + 8 * g + 4 * hi);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(
                            ep.bias + n0 + wn * WTN + (j + 1) * 32 + 8 * g + 4 * hi);
                        vec4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gte[e] = acc[i][j][4 * g + e] * ep.acc_scale + bg[e];
                            val[e] = acc[i][j + 1][4 * g + e] * ep.acc_scale + bv[e];
                            o[e] = Act<T>::from_f32(silu(gte[e]) * val[e]);
                        }
                        *reinterpret_cast<vec4*>(reinterpret_cast<T*>(ep.out) + (long)m * ep.ldo + hcol) = o;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn * WTN + j * 32 + 8 * g + 4 * hi;
                        epilogue4<EPI, T>(ep, m, n, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1],
                                          acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    }
            }
        }
    }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
static int launch_gemm_cfg(const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                           const EpiArgs& ep, hipStream_t st) {
    constexpr int STAGE = (BM + BN) * 64 * 2;
    constexpr int EPI_LDS = BM * ((BN * 4 / 16) | 1) * 16;            // fp32 tile at its odd chunk pitch
    constexpr int LDS = 2 * STAGE > EPI_LDS ? 2 * STAGE : EPI_LDS;
    auto kern = gemm_tn_kernel<T, BM, BN, WM, WN, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = cdiv(M, BM), tiles_n = N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(64 * WM * WN), LDS, st,
                       reinterpret_cast<const T*>(A), lda, reinterpret_cast<const T*>(W), ldw, M, N, K, ep,
                       tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_tn_kernel");
    return AMDS_OK;
}

template <typename T, int EPI>
static int launch_gemm_8p64(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                            hipStream_t st);   // gemm_8p64.h
template <typename T, int EPI>
static int launch_gemm_4w64(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                            hipStream_t st);   // gemm_4w64.h
template <typename T, int EPI, bool SPREAD, int P3, int P0, int PRL, int PRS, bool TN>
static int launch_gemm_4w16(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                            hipStream_t st);   // gemm_4w16.h

// kernel ids (amds_gemm_ex): 0 = 128x128 tile, one barrier per K step (small problems, any N % 128 == 0)
//   1 = 128x96 tile, four waves stacked along M (N % 96 == 0: the Swin widths 96/192/288/576 that 128 does not divide)
//  12 = 256x256x64 four waves, 128x128 wave tiles on v_mfma 16x16x32, AGPR accumulators (gemm_4w16.h): PRODUCTION for every
//       epilogue but PATCH whenever N % 256 == 0 and the grid fills the chip; 13 = the same kernel with tile kt+2 requested
//       from mid-tile kt on (two barriers per K tile, gemm_4w16.h SCHED 2; A/B)
//   8 = 256x256x64 eight-wave staggered two-group pipeline (gemm_8p64.h): the PATCH epilogue, batched fallback
//  10 = the four-wave structure on v_mfma 32x32x16 (gemm_4w64.h): the one A/B sibling kept
// (the BK = 32 predecessors 3 / 7 and the ping-pong experiment 9 of round 1 were removed; their measurements stay in profiles/r01_*)
template <typename T, int EPI>
static int launch_gemm(int cfg, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                       const EpiArgs& ep, hipStream_t st) {
    if (cfg == 10 && N % 256 == 0 && ep.nbatch == 1) return launch_gemm_4w64<T, EPI>(A, lda, W, ldw, M, N, K, ep, st);
    if (cfg == 10) cfg = 8;
    if (cfg == 12 && N % 256 == 0) return launch_gemm_4w16<T, EPI, true, 6, 6, 0, 0, false>(A, lda, W, ldw, M, N, K, ep, st);
    if (cfg == 13 && N % 256 == 0) return launch_gemm_4w16<T, EPI, true, -2, 0, 0, 0, false>(A, lda, W, ldw, M, N, K, ep, st);
#ifdef AMDS_GEMM_PROBE      // overlap probe of round 4 (gemm_4w16.h PRL / PRS; make PROBE=1): K-loop traffic of (0, 2) / (1, 1) / (4, 4) loads, stores per lane and K tile
    if constexpr (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_RESIDUAL) {
        if (cfg == 21 && N % 256 == 0) return launch_gemm_4w16<T, EPI, true, 6, 6, 0, 2, false>(A, lda, W, ldw, M, N, K, ep, st);
        if (cfg == 22 && N % 256 == 0) return launch_gemm_4w16<T, EPI, true, 6, 6, 1, 1, false>(A, lda, W, ldw, M, N, K, ep, st);
        if (cfg == 23 && N % 256 == 0) return launch_gemm_4w16<T, EPI, true, 6, 6, 4, 4, false>(A, lda, W, ldw, M, N, K, ep, st);
    }
#endif
    if constexpr (EPI == AMDS_EPI_BIAS_F32) {      // 15 = the token-major (TN) form of id 12: weight gradients straight from dY / X (gemm_4w16.h)
        if (cfg == 15 && N % 256 == 0 && M % 256 == 0) return launch_gemm_4w16<T, EPI, true, 6, 6, 0, 0, true>(A, lda, W, ldw, M, N, K, ep, st);
    }
    if (cfg == 15) { set_error("amds_gemm: kernel 15 (token-major operands) takes the BIAS_F32 epilogue and M, N multiples of 256"); return AMDS_ERR_INVALID; }
    if (cfg == 12 || cfg == 13) cfg = 8;

    if (cfg == 8 && N % 256 == 0) return launch_gemm_8p64<T, EPI>(A, lda, W, ldw, M, N, K, ep, st);
    if (cfg == 8) cfg = 0;
    if (cfg == 0 && N % 128 != 0) cfg = 1;
    switch (cfg) {
        case 0: return launch_gemm_cfg<T, 128, 128, 2, 2, EPI>(A, lda, W, ldw, M, N, K, ep, st);
        case 1:
            if constexpr (EPI == AMDS_EPI_SWIGLU || EPI == AMDS_EPI_PATCH) {
                set_error("amds_gemm: the 128x96 tile has no SWIGLU / PATCH epilogue");
                return AMDS_ERR_INVALID;
            } else {
                if (N % 96 != 0) { set_error("amds_gemm: kernel 1 needs N %% 96 == 0 (N=%d)", N); return AMDS_ERR_INVALID; }
                return launch_gemm_cfg<T, 128, 96, 4, 1, EPI>(A, lda, W, ldw, M, N, K, ep, st);
            }
    }
    set_error("amds_gemm: unknown tile config %d", cfg);
    return AMDS_ERR_INVALID;
}

template <typename T>
int gemm_dispatch(int cfg, int epi, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                  const EpiArgs& ep, hipStream_t st);

#define AMDS_GEMM_DISPATCH_IMPL(T)                                                                        \
    template <>                                                                                           \
    int gemm_dispatch<T>(int cfg, int epi, const void* A, long lda, const void* W, long ldw, int M, int N, \
                         int K, const EpiArgs& ep, hipStream_t st) {                                      \
        switch (epi) {                                                                                    \
            case AMDS_EPI_BIAS: return launch_gemm<T, AMDS_EPI_BIAS>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_BIAS_GELU: return launch_gemm<T, AMDS_EPI_BIAS_GELU>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_BIAS_RELU: return launch_gemm<T, AMDS_EPI_BIAS_RELU>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_RESIDUAL: return launch_gemm<T, AMDS_EPI_RESIDUAL>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_BIAS_F32: return launch_gemm<T, AMDS_EPI_BIAS_F32>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_SWIGLU: return launch_gemm<T, AMDS_EPI_SWIGLU>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_PATCH: return launch_gemm<T, AMDS_EPI_PATCH>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_BIAS_GELU_F32: return launch_gemm<T, AMDS_EPI_BIAS_GELU_F32>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
            case AMDS_EPI_BIAS_RELU_F32: return launch_gemm<T, AMDS_EPI_BIAS_RELU_F32>(cfg, A, lda, W, ldw, M, N, K, ep, st); \
        }                                                                                                 \
        set_error("amds_gemm: unknown epilogue %d", epi);                                                 \
        return AMDS_ERR_INVALID;                                                                          \
    }

}  // namespace amds
