// gemm_8p.h -- the production GEMM of the tile-encoder path: C[M,N] = A[M,K] W[N,K]^T (+ fused epilogue),
// 256 x 256 block tile, 8 waves, MFMA 32x32x16 (f16/bf16 in, fp32 accumulate), gfx950.
//
// Why this structure.  With one barrier per K-step (gemm_kernel.h) every wave of the block issues its
// global->LDS copies, then its ds_reads, then its MFMAs at the same time: the matrix pipe idles while the
// waves issue memory instructions (an LDS-DMA piece costs 60-180 issue cycles) and the measured rate is
// ~0.75 PF/s.  Here the two waves that share a SIMD are kept in OPPOSITE roles:
//
//   * 8 waves = 2 groups (rows of the 2x4 wave grid).  Waves w and w+4 sit on the same SIMD, one per group.
//   * the K loop is cut into phases of BK = 32; a phase of a wave is  LOAD | barrier | COMPUTE | barrier:
//       LOAD    : 12 ds_read_b128 (the wave's 4 A + 2 W fragments x 2 k-steps of this phase)
//                 + 4 global_load_lds pieces of the tile three phases ahead (+ counted vmcnt, never 0)
//       COMPUTE : 16 MFMAs (8 accumulators x 2 k-steps), s_setprio 1
//   * group 1 executes one extra barrier up front, so it always runs half a phase behind group 0: in every
//     barrier-to-barrier slot one wave of each SIMD is computing and the other is loading.
//   * LDS is a 4-deep ring of 32 KB stages (256 rows x 64 B per operand); tile p+3 is requested while tile p
//     is consumed, i.e. copies have 4-5 slots (~1 us) to land and the loop never drains the VM counter.
//   * LDS image rows are 64 B; the 16-byte chunk index is XOR-ed with (row>>2)&3 on the SOURCE address of
//     the copy (LDS-DMA writes lane-linearly) and again on the read: a fragment read touches 16 distinct
//     16-byte slots per 16-lane group (conflict free).
//
// Hazards (every cross-wave LDS dependency is ordered by "issuing wave's waitcnt, then a barrier"):
//   RAW  tile p+1 is first read in the slot after the one in which every wave executed vmcnt(8) (which covers
//        its own pieces of tile p+1: only tiles p+2, p+3 may still be in flight), then the barrier.
//   WAR  stage (p+3)&3 last held tile p-1, whose reads were retired by lgkmcnt(0) BEFORE the barrier that
//        ends the reading slot; the copies into it are issued at least one barrier later.
#pragma once
#include "gemm_kernel.h"
#include "gemm_epilogue.h"

namespace amds {

// ABL: ablation bits for performance archaeology (results are WRONG when non-zero; never dispatched by amds_gemm):
//   1 = no global->LDS copies in the loop, 2 = no MFMAs, 4 = no fragment ds_reads, 8 = no barriers
template <typename T, int EPI, int ABL = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_8p_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K,
               EpiArgs ep, int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int BM = 256, BN = 256, BK = 32, NT = 512;
    constexpr int ROWB = BK * 2;                       // 64 bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;   // 16 KB, 32 KB
    constexpr int NSTAGE = 4;
    constexpr int FM = 4, FN = 2;
    constexpr int GROUP_M = 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
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

    // ---- copy (global -> LDS) addressing: 2048 16-byte chunks per tile, 4 per thread -------------------
    const T* src[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (it & 1) * NT + tid, row = c >> 2, cp = c & 3, sc = cp ^ ((row >> 2) & 3);
        if (it < 2) src[it] = A + (long)min(m0 + row, M - 1) * lda + sc * 8;
        else src[it] = W + (long)(n0 + row) * ldw + sc * 8;
    }
    auto issue_tile = [&](int kt) {
        if ((ABL & 1) && kt >= 3) return;
        char* st = smem + (kt & (NSTAGE - 1)) * STAGE;
        const int koff = kt * BK;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            glds16(src[it] + koff, st + (it >> 1) * A_BYTES + ((it & 1) * NT + wave * 64) * 16);
    };

    // ---- fragment addressing ------------------------------------------------------------------------
    const int swz = (l31 >> 2) & 3;
    const int a_off = (grp * 128 + l31) * ROWB;
    const int w_off = A_BYTES + (wc * 64 + l31) * ROWB;
    const int c0 = ((0 + hi) ^ swz) << 4, c1 = ((2 + hi) ^ swz) << 4;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    vec8 af[2][FM], wf[2][FN];
    auto load_frags = [&](int p) {
        if (ABL & 64) __builtin_amdgcn_s_setprio(2);    // experiment: the LOADING wave outranks the computing one
        if ((ABL & 4) && p > 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[ks][i]));
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(wf[ks][j]));
            }
            return;
        }
        const char* sb = smem + (p & (NSTAGE - 1)) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = ks ? c1 : c0;
#pragma unroll
            for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const vec8*>(sb + a_off + i * 32 * ROWB + co);
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[ks][j] = *reinterpret_cast<const vec8*>(sb + w_off + j * 32 * ROWB + co);
        }
    };
    auto compute = [&]() {
        if (ABL & 64) __builtin_amdgcn_s_setprio(0);
        if (!(ABL & 32) && !(ABL & 64)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if (ABL & 2) { asm volatile("" :: "v"(wf[ks][j]), "v"(af[ks][i])); }
                    else if (ABL & 128) Act<T>::mfma32_agpr(wf[ks][j], af[ks][i], acc[i][j]);
                    else acc[i][j] = Act<T>::mfma32(wf[ks][j], af[ks][i], acc[i][j]);
                }
        if (!(ABL & 32) && !(ABL & 64)) __builtin_amdgcn_s_setprio(0);
    };

#define AMDS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AMDS_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define AMDS_BARRIER()                                  \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();   \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)

    const int P = K / BK;   // >= 4 (host checks K >= 128)
    // ABL & 16: lane 0 of each wave of block 0 logs s_memtime at every segment boundary into ep.pos (debug only)
    unsigned long long* tlog = nullptr;
    int tcnt = 0;
    if constexpr ((ABL & 16) != 0) {
        if (blockIdx.x == 0 && lane == 0) tlog = reinterpret_cast<unsigned long long*>(const_cast<float*>(ep.pos)) + wave * 4096;
    }
#define AMDS_TS()                                                                              \
    do {                                                                                       \
        if constexpr ((ABL & 16) != 0) {                                                       \
            if (tlog && tcnt < 4096) tlog[tcnt++] = __builtin_amdgcn_s_memtime();              \
        }                                                                                      \
    } while (0)

    // ---- prologue: tiles 0..2 in flight, tile 0 landed and visible, groups staggered ------------------
    issue_tile(0);
    issue_tile(1);
    issue_tile(2);
    AMDS_WAIT_VM(8);
    AMDS_BARRIER();
    if (grp == 1) AMDS_BARRIER();

    // phase with copies of tile p+3 and the steady-state wait
    int p = 0;
    for (; p < P - 3; ++p) {
        AMDS_TS();
        load_frags(p);
        AMDS_TS();
        issue_tile(p + 3);
        AMDS_TS();
        AMDS_WAIT_LGKM0();
        AMDS_TS();
        if (grp == 1) AMDS_WAIT_VM(8);       // tile p+1 landed (tiles p+2, p+3 may be in flight)
        AMDS_TS();
        AMDS_BARRIER();
        AMDS_TS();
        compute();
        AMDS_TS();
        if (grp == 0) AMDS_WAIT_VM(8);
        AMDS_TS();
        AMDS_BARRIER();
    }
    // tail: nothing left to request; drain 4 -> 0
    load_frags(p);
    AMDS_WAIT_LGKM0();
    if (grp == 1) AMDS_WAIT_VM(4);
    AMDS_BARRIER();
    compute();
    if (grp == 0) AMDS_WAIT_VM(4);
    AMDS_BARRIER();
    ++p;
    load_frags(p);
    AMDS_WAIT_LGKM0();
    if (grp == 1) AMDS_WAIT_VM(0);
    AMDS_BARRIER();
    compute();
    if (grp == 0) AMDS_WAIT_VM(0);
    AMDS_BARRIER();
    ++p;
    load_frags(p);
    AMDS_WAIT_LGKM0();
    AMDS_BARRIER();
    compute();
    AMDS_BARRIER();
    if (grp == 0) AMDS_BARRIER();   // matches group 1's extra barrier of the prologue
#undef AMDS_WAIT_VM
#undef AMDS_WAIT_LGKM0
#undef AMDS_BARRIER

    if (ABL & 128) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA D -> reader hazard (asm MFMAs are not padded)
    // ---- epilogue ---------------------------------------------------------------------------------------
    if constexpr (epi_is_staged<EPI>() && !(ABL & 256)) {
        epilogue_staged_256<EPI, T>(acc, ep, smem, m0, n0, M, grp, wc, wave, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + grp * 128 + i * 32 + l31;
        if (m < M) {
            if constexpr (EPI == AMDS_EPI_SWIGLU) {
                const int hbase = (n0 + wc * 64) / 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int hcol = hbase + 8 * g + 4 * hi;
                    const f32x4 bg = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wc * 64 + 8 * g + 4 * hi);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wc * 64 + 32 + 8 * g + 4 * hi);
                    vec4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gte = acc[i][0][4 * g + e] * ep.acc_scale + bg[e];
                        const float val = acc[i][1][4 * g + e] * ep.acc_scale + bv[e];
                        o[e] = Act<T>::from_f32(silu(gte) * val);
                    }
                    *reinterpret_cast<vec4*>(reinterpret_cast<T*>(ep.out) + (long)m * ep.ldo + hcol) = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wc * 64 + j * 32 + 8 * g + 4 * hi;
                        epilogue4<EPI, T>(ep, m, n, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                                          acc[i][j][4 * g + 3]);
                    }
            }
        }
    }
}

template <typename T, int EPI, int ABL>
static int launch_gemm_8p_abl(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                              hipStream_t st) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    auto kern = gemm_8p_kernel<T, EPI, ABL>;
    AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_8p_kernel(abl)");
    return AMDS_OK;
}

template <typename T, int EPI, int VARIANT>
static int launch_gemm_8p(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                          hipStream_t st) {
    constexpr int LDS = 4 * (256 + 256) * 64;
    auto kern = gemm_8p_kernel<T, EPI, VARIANT>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_8p_kernel");
    return AMDS_OK;
}

}  // namespace amds
