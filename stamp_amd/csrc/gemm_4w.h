// gemm_4w.h -- 256 x 256 block tile, FOUR waves (one per SIMD), wave tile 128 x 128 = 4 x 4 MFMA 32x32x16
// fragments with the 256 fp32 accumulator registers in the AGPR half of the register file.
//
// Rationale (measured on gemm_8p.h, profiles/r01_gemm_ablation.txt + r01_gemm_timeline.txt): with two waves per
// SIMD in opposite roles, the loading wave needs ~2x as long to issue its 12 ds_reads + 4 LDS-DMA pieces as its
// partner needs for 16 MFMAs, so every barrier-to-barrier slot is bound by the loader.  Here each SIMD runs ONE
// wave that owns the whole 512-entry register file: the same instruction stream carries the MFMAs and, in the
// gaps between them (an MFMA occupies the pipe for 32 cycles, ~5 other instructions can issue meanwhile), the
// fragment reads of the NEXT k-step and the LDS-DMA requests of the tile three phases ahead:
//   per k-step (16 MFMAs, 512 cycles): 8 ds_read_b128 + 4 global_load_lds pieces      (0.75 fillers per gap)
//   LDS read traffic per MFMA is 2/3 of the 128x64 wave tile's (8 fragment reads feed 16 MFMAs instead of 6 -> 8)
// Synchronisation: a 4-deep ring of 32 KB stages (BK = 32), one s_barrier per phase (32 MFMAs), counted vmcnt(12).
//   RAW  tile p+1 is read (k-step 0 fragments) in the second half of phase p, after every wave waited for its own
//        pieces of tile p+1 (vmcnt(12): tile p+2 and the 4 pieces of tile p+3 issued so far may be in flight) and
//        passed the mid-phase barrier.
//   WAR  stage (p+3)&3 held tile p-1; its last fragment reads (k-step 1) are retired by lgkmcnt(0) before the
//        mid-phase barrier of phase p-1; the copies into it are issued in phase p.
#pragma once
#include "gemm_kernel.h"

namespace amds {

template <typename T, int EPI, int ABL = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_4w_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K,
               EpiArgs ep, int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int BM = 256, BN = 256, BK = 32, NT = 256;
    constexpr int ROWB = BK * 2;
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
    constexpr int NSTAGE = 4;
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

    // ---- copy addressing: 2048 16-byte chunks per tile, 8 per thread (4 of A, 4 of W); buffer-form LDS-DMA (common.h) ----
    const int rows_a = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(A + (long)m0 * lda), 0, (int)((((long)rows_a - 1) * lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(W + (long)n0 * ldw), 0, (int)(((long)(BN - 1) * ldw + K) * 2), 0x00020000);
    int voff[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = (it & 3) * NT + tid, row = c >> 2, cp = c & 3, sc = cp ^ ((row >> 2) & 3);
        voff[it] = (int)(((long)row * (it < 4 ? lda : ldw) + sc * 8) * 2);
    }
    // pieces [lo, hi) of tile kt
    auto issue_pieces = [&](int kt, int lo, int hi_) {
        if ((ABL & 1) && kt >= 3) return;
        char* st = smem + (kt & (NSTAGE - 1)) * STAGE;
        const int koff = kt * BK * 2;
#pragma unroll
        for (int it = 0; it < 8; ++it)
            if (it >= lo && it < hi_)
                bufl16(it < 4 ? rsrc_a : rsrc_w, st + (it >> 2) * A_BYTES + ((it & 3) * NT + wave * 64) * 16, voff[it], koff);
    };

    const int swz = (l31 >> 2) & 3;
    const int a_off = (wm * 128 + l31) * ROWB;
    const int w_off = A_BYTES + (wn * 128 + l31) * ROWB;
    const int c0 = ((0 + hi) ^ swz) << 4, c1 = ((2 + hi) ^ swz) << 4;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    vec8 afA[FM], wfA[FN], afB[FM], wfB[FN];
    auto load_frags = [&](int p, int ks, vec8 (&af)[FM], vec8 (&wf)[FN]) {
        if ((ABL & 4) && (p > 0 || ks > 0)) {
#pragma unroll
            for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(wf[j]));
            return;
        }
        const char* sb = smem + (p & (NSTAGE - 1)) * STAGE;
        const int co = ks ? c1 : c0;
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
    // scheduling recipe of a half phase: 16 MFMAs with 8 ds_reads and (optionally) 4 LDS-DMA requests in the gaps
    auto interleave = [&](bool with_copies) {
        if (ABL) return;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
        if (with_copies) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (LDS-DMA)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
    };

#define AMDS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AMDS_WAIT_LGKM0()                                   \
    do {                                                    \
        __builtin_amdgcn_sched_barrier(0);                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    } while (0)
#define AMDS_BARRIER()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    const int P = K / BK;   // >= 4

    // ---- prologue ---------------------------------------------------------------------------------------
    issue_pieces(0, 0, 8);
    issue_pieces(1, 0, 8);
    issue_pieces(2, 0, 8);
    AMDS_WAIT_VM(16);
    AMDS_BARRIER();                 // tile 0 visible
    load_frags(0, 0, afA, wfA);
    AMDS_WAIT_LGKM0();

    int p = 0;
    for (; p < P - 3; ++p) {
        // first half: k-step 0 of tile p
        load_frags(p, 1, afB, wfB);
        issue_pieces(p + 3, 0, 4);
        mfmas(afA, wfA);
        interleave(true);
        AMDS_WAIT_LGKM0();
        AMDS_WAIT_VM(12);           // own pieces of tile p+1 landed
        AMDS_BARRIER();             // tile p+1 visible to all; stage of tile p-1 ... p+3 handed over
        // second half: k-step 1 of tile p, prefetching k-step 0 of tile p+1
        load_frags(p + 1, 0, afA, wfA);
        issue_pieces(p + 3, 4, 8);
        mfmas(afB, wfB);
        interleave(true);
        AMDS_WAIT_LGKM0();
    }
    // tail: phases P-3, P-2 (no more copies), then P-1
    for (int t = 0; t < 2; ++t, ++p) {
        load_frags(p, 1, afB, wfB);
        mfmas(afA, wfA);
        interleave(false);
        AMDS_WAIT_LGKM0();
        if (t == 0) AMDS_WAIT_VM(8); else AMDS_WAIT_VM(0);
        AMDS_BARRIER();
        load_frags(p + 1, 0, afA, wfA);
        mfmas(afB, wfB);
        interleave(false);
        AMDS_WAIT_LGKM0();
    }
    load_frags(p, 1, afB, wfB);
    mfmas(afA, wfA);
    AMDS_WAIT_LGKM0();
    mfmas(afB, wfB);
    AMDS_BARRIER();                 // every wave is done with the LDS stages
#undef AMDS_WAIT_VM
#undef AMDS_WAIT_LGKM0
#undef AMDS_BARRIER

    // ---- epilogue: LDS-staged, coalesced (wave tile 128 x 128) -----------------------------------------------
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
    constexpr bool STAGED = F16OUT || EPI == AMDS_EPI_RESIDUAL || EPI == AMDS_EPI_BIAS_F32 ||
                            EPI == AMDS_EPI_BIAS_GELU_F32 || EPI == AMDS_EPI_BIAS_RELU_F32;
    if constexpr (STAGED) {
        constexpr int NPASS = F16OUT ? 1 : 2;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            if (pass) __syncthreads();
            // fragment column j outermost: its 4 bias / LayerScale vectors are loaded once (16 registers), then used by 4 row blocks
#pragma unroll
            for (int jj = 0; jj < (F16OUT ? 4 : 2); ++jj) {
                const int j = F16OUT ? jj : pass * 2 + jj;
                EpiCols<4> cols;
                epi_cols_load<EPI>(ep, cols, [&](int g) { return n0 + wn * 128 + j * 32 + 8 * g + 4 * hi; });
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = wm * 128 + i * 32 + l31;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        v = epi_value<EPI>(ep, cols.bias[g], cols.scale[g], v);
                        if constexpr (F16OUT) {
                            vec4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = Act<T>::from_f32(v[e]);
                            const int chunk = wn * 16 + j * 4 + g;
                            *reinterpret_cast<vec4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4) + hi * 8) = o;
                        } else {
                            const int chunk = wn * 16 + jj * 8 + 2 * g + hi;
                            *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = v;
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int row = wave * 64 + it * 2 + hi;
                const int chunk = l31 ^ (row & 31);
                if constexpr (F16OUT) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 512 + l31 * 16);
                    if (m0 + row < M)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + chunk * 8) = v;
                } else {
                    f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 512 + l31 * 16);
                    const int n = n0 + (chunk >> 4) * 128 + (pass * 2 + ((chunk >> 3) & 1)) * 32 + (chunk & 7) * 4;
                    if (m0 + row < M) {
                        f32x4* q = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)(m0 + row) * ep.ldo + n);
                        if constexpr (EPI == AMDS_EPI_RESIDUAL) v += *q;
                        *q = v;
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
}

template <typename T, int EPI, int ABL>
static int launch_gemm_4w_abl(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                              hipStream_t st) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    auto kern = gemm_4w_kernel<T, EPI, ABL>;
    AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_4w_kernel(abl)");
    return AMDS_OK;
}

template <typename T, int EPI>
static int launch_gemm_4w(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                          hipStream_t st) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    auto kern = gemm_4w_kernel<T, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_4w_kernel");
    return AMDS_OK;
}

}  // namespace amds
