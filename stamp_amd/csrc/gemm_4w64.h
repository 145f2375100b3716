// gemm_4w64.h -- 256 x 256 block tile, FOUR waves (one per SIMD, 128 x 128 wave tiles, accumulators in AGPRs)
// with the copy path of gemm_8p64.h: K tiles of 64 as 128-byte LDS rows (8 full lines per LDS-DMA instruction instead
// of 16 half lines) in TWO 64 KB stages, one barrier per K tile.
//
// This is the structure the vendor library picks for the tile-encoder shapes (profiles/r01_gemm_vendor_and_power.txt:
// MT256x256x64, 4 waves, direct-to-LDS, 1.56 PFLOP/s loop rate).  Schedule of K tile kt (k-steps of 16, 16 MFMAs each):
//     ks0  MFMA(frag set A) | read set B <- (kt, ks1) | LDS-DMA pieces  6..11 of tile kt+1
//     ks1  MFMA(B)          | read A <- (kt, ks2)     | pieces 12..15 of tile kt+1
//     ks2  MFMA(A)          | read B <- (kt, ks3)
//     lgkmcnt(0), vmcnt(0), s_barrier                   <- tile kt+1 landed and visible; every wave is done reading tile kt
//     ks3  MFMA(B)          | read A <- (kt+1, ks0)   | pieces 0..5 of tile kt+2 (into the stage tile kt just released)
//   RAW  tile kt+1 is first read in ks3 of tile kt, after every wave waited for its own pieces (the youngest were issued in
//        ks1, one k-step = 512 MFMA cycles earlier) and passed the barrier.
//   WAR  stage kt&1 is re-filled (tile kt+2) only after the barrier that follows the retirement (lgkmcnt(0)) of the last
//        fragment reads of tile kt.
#pragma once
#include <type_traits>
#include "gemm_kernel.h"

namespace amds {

// ABL: ablation bits (results WRONG when non-zero; tuning instantiations only; the library instantiates ABL = 0): 1 = no LDS-DMA after the first
// two K tiles, 4 = no fragment ds_reads after the first, 8 = no barriers
// P3 / P0: LDS-DMA pieces of the next K tile requested in k-step 3 of the previous tile / k-step 0 (the rest in k-step 1)
// AUXA / AUXW: cache-policy bits of the LDS-DMA loads of the A / W tiles (0 = default, 1 = sc0, 2 = nt, 3 = sc0 nt); measured at the
// power cap: sc0 no change, nt on W -6...-12 %, nt on A 0...-5 %
template <typename T, int EPI, int ABL = 0, int P3 = 6, int P0 = 6, int AUXA = 0, int AUXW = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_4w64_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K,
                 EpiArgs ep, int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int BM = 256, BN = 256, BK = 64, NT = 256;
    constexpr int ROWB = BK * 2;                                   // 128 bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;   // 32 KB, 64 KB
    constexpr int FM = 4, FN = 4;
    constexpr int GROUP_M = 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

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

    // ---- copy addressing: 4096 16-byte chunks per K tile, 16 per thread (8 of A, 8 of W); chunk ^= (row>>1)&7 on the source.
    // Buffer form (buffer_load_dwordx4 ... offen lds): one 32-bit byte offset per piece, constant over the K loop, the K
    // advance in the scalar offset, no vector address arithmetic in the loop; rows past M are out of range and read as 0.
    const int rows_a = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(A + (long)m0 * lda), 0, (int)((((long)rows_a - 1) * lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(W + (long)n0 * ldw), 0, (int)(((long)(BN - 1) * ldw + K) * 2), 0x00020000);
    int voff[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int c = (it & 7) * NT + tid, row = c >> 3, cp = c & 7, sc = cp ^ ((row >> 1) & 7);
        voff[it] = (int)(((long)row * (it < 8 ? lda : ldw) + sc * 8) * 2);
    }
    auto issue_pieces = [&](int kt, int lo, int hi_) {
        if ((ABL & 1) && kt >= 2) return;
        char* st = smem + (kt & 1) * STAGE;
        const int koff = kt * BK * 2;
#pragma unroll
        for (int it = 0; it < 16; ++it)
            if (it >= lo && it < hi_) {
                if (it < 8) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(st + ((it & 7) * NT + wave * 64) * 16), 16, voff[it], koff, 0, AUXA);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(st + A_BYTES + ((it & 7) * NT + wave * 64) * 16), 16, voff[it], koff,
                                                             0, AUXW);
                }
            }
    };

    const int swz = (l31 >> 1) & 7;
    const int a_off = (wm * 128 + l31) * ROWB;
    const int w_off = A_BYTES + (wn * 128 + l31) * ROWB;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    vec8 afA[FM], wfA[FN], afB[FM], wfB[FN];
    auto load_frags = [&](int kt, int ks, vec8 (&af)[FM], vec8 (&wf)[FN]) {
        if ((ABL & 4) && (kt > 0 || ks > 1)) {
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(wf[j]));
            return;
        }
        const char* sb = smem + (kt & 1) * STAGE;
        const int co = ((ks * 2 + hi) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const vec8*>(sb + a_off + i * 32 * ROWB + co);
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const vec8*>(sb + w_off + j * 32 * ROWB + co);
    };
    auto mfmas = [&](vec8 (&af)[FM], vec8 (&wf)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = Act<T>::mfma32(wf[j], af[i], acc[i][j]);
    };
    // scheduling recipe of one k-step: 16 MFMAs with `reads` ds_reads and `copies` LDS-DMA requests in the gaps
    auto interleave = [&](auto reads_c, auto copies_c) {
        if constexpr ((ABL & 15) != 0) { __builtin_amdgcn_sched_barrier(0); return; }
        constexpr int READS = decltype(reads_c)::value, COPIES = decltype(copies_c)::value;
#pragma unroll
        for (int r = 0; r < READS; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
#pragma unroll
        for (int r = 0; r < COPIES; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA); alternating with the reads measured 3 % slower
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - READS - COPIES, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 8> I8;
    constexpr int P1 = 16 - P3 - P0;
    static_assert(P3 <= 8 && P0 <= 8 && P1 >= 0 && P1 <= 8, "at most 8 LDS-DMA pieces per k-step (16 MFMAs, 8 fragment reads)");
    typedef std::integral_constant<int, P3> IP3;
    typedef std::integral_constant<int, P0> IP0;
    typedef std::integral_constant<int, P1> IP1;

#define AMDS_BARRIER()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    // ABL & 16: thread 0 of every workgroup logs {s_memrealtime at start, s_memtime at 5 points, HW_ID, XCC_ID} into ep.pos (debug)
    unsigned long long* tlog = nullptr;
    if constexpr ((ABL & 16) != 0) {
        if (tid == 0) {
            tlog = reinterpret_cast<unsigned long long*>(const_cast<float*>(ep.pos)) + (long)blockIdx.x * 8;
            tlog[0] = __builtin_amdgcn_s_memrealtime();
            tlog[1] = __builtin_amdgcn_s_memtime();
            tlog[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            tlog[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
    }
#define AMDS_STAMP(k)                                                              \
    do {                                                                           \
        if constexpr ((ABL & 16) != 0) {                                           \
            if (tlog) tlog[k] = __builtin_amdgcn_s_memtime();                      \
        }                                                                          \
    } while (0)
    const int nk = K / BK;   // >= 1
    issue_pieces(0, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMDS_BARRIER();
    AMDS_STAMP(2);                  // first K tile landed
    load_frags(0, 0, afA, wfA);
    if (nk > 1) issue_pieces(1, 0, P3);
    __builtin_amdgcn_sched_barrier(0);

    auto k_tile = [&](int kt, auto next_c, auto next2_c) {
        constexpr bool NEXT = decltype(next_c)::value, NEXT2 = decltype(next2_c)::value;
        load_frags(kt, 1, afB, wfB);
        if constexpr (NEXT) issue_pieces(kt + 1, P3, P3 + P0);
        mfmas(afA, wfA);
        if constexpr (NEXT) interleave(I8{}, IP0{}); else interleave(I8{}, I0{});
        load_frags(kt, 2, afA, wfA);
        if constexpr (NEXT) issue_pieces(kt + 1, P3 + P0, 16);
        mfmas(afB, wfB);
        if constexpr (NEXT) interleave(I8{}, IP1{}); else interleave(I8{}, I0{});
        load_frags(kt, 3, afB, wfB);
        mfmas(afA, wfA);
        interleave(I8{}, I0{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        AMDS_BARRIER();
        if constexpr (NEXT) load_frags(kt + 1, 0, afA, wfA);
        if constexpr (NEXT2) issue_pieces(kt + 2, 0, P3);
        mfmas(afB, wfB);
        if constexpr (NEXT2) interleave(I8{}, IP3{});
        else if constexpr (NEXT) interleave(I8{}, I0{});
        else __builtin_amdgcn_sched_barrier(0);
    };
    int kt = 0;
    for (; kt < nk - 2; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
    if (nk >= 2) k_tile(kt++, std::true_type{}, std::false_type{});
    k_tile(kt, std::false_type{}, std::false_type{});
    AMDS_BARRIER();                 // every wave is done with the LDS stages
    AMDS_STAMP(3);                  // K loop done
#undef AMDS_BARRIER

    // ---- epilogue: LDS-staged, coalesced (wave tile 128 x 128) -----------------------------------------------
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
    constexpr bool STAGED = F16OUT || EPI == AMDS_EPI_RESIDUAL || EPI == AMDS_EPI_BIAS_F32 ||
                            EPI == AMDS_EPI_BIAS_GELU_F32 || EPI == AMDS_EPI_BIAS_RELU_F32;
    if constexpr (STAGED) {
        constexpr int NPASS = F16OUT ? 1 : 2;
        // (staging and storing the fp16 tile in two halves, so that the first half's stores drain under the second half's VALU
        //  work, measured 4 % slower per launch: one more barrier, shorter store batches)
        constexpr int NH = 1, FMH = FM / NH;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            if (pass) __syncthreads();
            // fragment column j outermost: its 4 bias / LayerScale vectors are loaded once (16 registers), then used by 4 row blocks
            auto values = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
                // bias / LayerScale vectors of fragment column jj + 1 are requested before column jj is processed: one exposed L2
                // round trip per pass instead of one per column (one wave per SIMD: nothing else would hide them)
                constexpr int NJ = F16OUT ? 4 : 2;
                auto load_cols = [&](EpiCols<4>& c, int j) {
                    epi_cols_load<EPI>(ep, c, [&](int g) { return n0 + wn * 128 + j * 32 + 8 * g + 4 * hi; });
                };
                EpiCols<4> cols, cols_next;
                load_cols(cols, F16OUT ? 0 : pass * 2);
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int j = F16OUT ? jj : pass * 2 + jj;
                    if (jj + 1 < NJ) load_cols(cols_next, j + 1);
#pragma unroll
                    for (int ii = 0; ii < FMH; ++ii) {
                        const int i = h * FMH + ii;
                        const int row = wm * 128 + i * 32 + l31;
#pragma unroll
                        for (int g = 0; g < 4; g += 2) {
                            f32x4 v0 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            f32x4 v1 = {acc[i][j][4 * g + 4], acc[i][j][4 * g + 5], acc[i][j][4 * g + 6], acc[i][j][4 * g + 7]};
                            epi_value_pair<EPI, FAST>(ep, cols.bias[g], cols.scale[g], cols.bias[g + 1], cols.scale[g + 1], v0, v1);
                            if constexpr (F16OUT) {
                                const vec4 o0 = Act<T>::from_f32x4(v0), o1 = Act<T>::from_f32x4(v1);
                                const int chunk = wn * 16 + j * 4 + g;
                                const int half = (hi ^ ((l31 >> 3) & 1)) * 8;      // see gemm_epilogue.h: conflict-free 8-byte stores
                                *reinterpret_cast<vec4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4) + half) = o0;
                                *reinterpret_cast<vec4*>(smem + row * 512 + (((chunk + 1) ^ (row & 31)) << 4) + half) = o1;
                            } else {
                                const int chunk = wn * 16 + jj * 8 + 2 * g + hi;
                                *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = v0;
                                *reinterpret_cast<f32x4*>(smem + row * 512 + (((chunk + 2) ^ (row & 31)) << 4)) = v1;
                            }
                        }
                    }
                    cols = cols_next;
                }
            };
            if (F16OUT && ep.bias != nullptr && ep.acc_scale == 1.0f) values(std::true_type{}); else values(std::false_type{});
            __syncthreads();
            if (h == NH - 1) AMDS_STAMP(4);              // values staged
            // one wave per SIMD: batch 8 rows (reads first, then the stores) so the LDS / L2 latencies overlap
#pragma unroll 1
            for (int b8 = 0; b8 < 4 / NH; ++b8) {
                if constexpr (F16OUT) {
                    u32x4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int rr = wave * (64 / NH) + (b8 * 8 + u) * 2 + hi, row = NH == 2 ? (rr >> 6) * 128 + h * 64 + (rr & 63) : rr;
                        v[u] = *reinterpret_cast<const u32x4*>(smem + row * 512 + l31 * 16);
                        if ((u >> 2) & 1) v[u] = u32x4{v[u][2], v[u][3], v[u][0], v[u][1]};      // row bit 3 set: halves stored swapped
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int rr = wave * (64 / NH) + (b8 * 8 + u) * 2 + hi, row = NH == 2 ? (rr >> 6) * 128 + h * 64 + (rr & 63) : rr;
                        const int chunk = l31 ^ (row & 31);
                        if (m0 + row < M)
                            *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + chunk * 8) = v[u];
                    }
                } else {
                    f32x4 v[8], o[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                        const int chunk = l31 ^ (row & 31);
                        const int n = n0 + (chunk >> 4) * 128 + (pass * 2 + ((chunk >> 3) & 1)) * 32 + (chunk & 7) * 4;
                        v[u] = *reinterpret_cast<const f32x4*>(smem + row * 512 + l31 * 16);
                        if constexpr (EPI == AMDS_EPI_RESIDUAL)
                            o[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ep.out) +
                                                                   (long)min(m0 + row, M - 1) * ep.ldo + n);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                        const int chunk = l31 ^ (row & 31);
                        const int n = n0 + (chunk >> 4) * 128 + (pass * 2 + ((chunk >> 3) & 1)) * 32 + (chunk & 7) * 4;
                        if constexpr (EPI == AMDS_EPI_RESIDUAL) v[u] += o[u];
                        if (m0 + row < M)
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)(m0 + row) * ep.ldo + n) = v[u];
                    }
                }
            }
          }
        }
    } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * 128 + i * 32 + l31;
            if (m < M) {
                if constexpr (EPI == AMDS_EPI_SWIGLU) {
#pragma unroll
                    for (int j = 0; j < FN; j += 2) {
                        const int hbase = (n0 + wn * 128 + j * 32) / 2;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int hcol = hbase + 8 * g + 4 * hi;
                            const f32x4 bg = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wn * 128 + j * 32 + 8 * g + 4 * hi);
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wn * 128 + (j + 1) * 32 + 8 * g + 4 * hi);
                            vec4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float gte = acc[i][j][4 * g + e] * ep.acc_scale + bg[e];
                                const float val = acc[i][j + 1][4 * g + e] * ep.acc_scale + bv[e];
                                o[e] = Act<T>::from_f32(silu(gte) * val);
                            }
                            *reinterpret_cast<vec4*>(reinterpret_cast<T*>(ep.out) + (long)m * ep.ldo + hcol) = o;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = n0 + wn * 128 + j * 32 + 8 * g + 4 * hi;
                            epilogue4<EPI, T>(ep, m, n, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                                              acc[i][j][4 * g + 3]);
                        }
                }
            }
        }
    }
    AMDS_STAMP(5);                      // all global stores issued
#undef AMDS_STAMP
}

template <typename T, int EPI, int ABL, int P3 = 6, int P0 = 6, int AUXA = 0, int AUXW = 0>
static int launch_gemm_4w64_abl(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                                hipStream_t st) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_4w64_kernel<T, EPI, ABL, P3, P0, AUXA, AUXW>;
    AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_4w64_kernel(abl)");
    return AMDS_OK;
}

template <typename T, int EPI>
static int launch_gemm_4w64(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                            hipStream_t st) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_4w64_kernel<T, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_4w64_kernel");
    return AMDS_OK;
}

}  // namespace amds
