// gemm_pp.h -- "ping-pong" GEMM: C[M,N] = A[M,K] W[N,K]^T (+ fused epilogue) with TWO independent 4-wave workgroups
// per CU that are kept half a tile out of phase, so that one workgroup's epilogue (VALU: bias / GELU / conversions,
// LDS staging, global stores) runs beside the other workgroup's K loop (matrix pipe).
//
// Why: profiles/r01_gemm_fixed_cost.txt -- the 256x256 kernel (gemm_8p64.h) runs its K loop at ~1.2 PFLOP/s and then
// pays 200-480 us of epilogue per launch during which the CU's matrix pipes idle (one workgroup per CU: nothing else
// is resident).  The accumulators being drained and the accumulators being filled have to coexist for the two to
// overlap; with 512 registers per SIMD that means two waves per SIMD that belong to DIFFERENT tiles.
//
//   * workgroup = 4 waves (2 x 2), one per SIMD, block tile 256 x 128, wave tile 128 x 64 (4 x 2 MFMA 32x32x16
//     fragments, 128 accumulator registers); two workgroups fit a CU (<= 256 registers per wave, 72 KB of LDS each).
//   * K tiles of 32 in a 3-deep ring of 24 KB stages (64-byte LDS rows, chunk index XOR (row>>2)&3 as in gemm_8p.h).
//     Tile kt+2 is requested at the top of iteration kt; fragments are register double-buffered one k-step ahead,
//     ACROSS the per-iteration barrier, so a workgroup that has the matrix pipe to itself (its partner is in its
//     epilogue) still issues MFMAs back to back.
//   * one barrier per K tile.  RAW: tile kt+1 is read after every wave executed vmcnt(<=6) (its own pieces of kt+1
//     have landed; only kt+2 may be in flight) and then the barrier.  WAR: stage (kt+2)%3 held tile kt-1, whose last
//     fragment reads were retired (lgkmcnt(0)) before the barrier of iteration kt-1.
//   * the phase offset: workgroups are not persistent; the first two workgroups that arrive on a CU take a ticket
//     (atomic counter keyed by XCC_ID / HW_ID.{se,sh,cu}); ticket 1 sleeps half a tile period before it starts.  Every
//     later workgroup inherits the phase of the one whose slot it takes.
#pragma once
#include <type_traits>
#include "gemm_kernel.h"

namespace amds {

template <int EPI, typename T>
__device__ __forceinline__ void epilogue_staged_pp(f32x16 (&acc)[4][2], const EpiArgs& ep, char* smem, int m0, int n0, int M,
                                                   int wm, int wn, int wave, int lane) {
    typedef typename Act<T>::vec4 vec4;
    const int l31 = lane & 31, hi = lane >> 5;
    const int l15 = lane & 15, rsub = lane >> 4;
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
    EpiCols<8> cols;      // [j * 4 + g]: columns n0 + wn * 64 + j * 32 + 8 g + 4 hi
    epi_cols_load<EPI>(ep, cols, [&](int q) { return n0 + wn * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi; });
    // staging image: 256 rows x 256 B (16 chunks of 16 B, chunk index XOR row&15)
    if constexpr (F16OUT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 128 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    v = epi_value<EPI>(ep, cols.bias[j * 4 + g], cols.scale[j * 4 + g], v);
                    vec4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = Act<T>::from_f32(v[e]);
                    const int chunk = wn * 8 + j * 4 + g;
                    *reinterpret_cast<vec4*>(smem + row * 256 + ((chunk ^ (row & 15)) << 4) + hi * 8) = o;
                }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = wave * 64 + it * 4 + rsub;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 256 + l15 * 16);
            const int chunk = l15 ^ (row & 15);
            if (m0 + row < M)
                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + chunk * 8) = v;
        }
    } else {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 128 + i * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][pass][4 * g], acc[i][pass][4 * g + 1], acc[i][pass][4 * g + 2], acc[i][pass][4 * g + 3]};
                    v = epi_value<EPI>(ep, cols.bias[pass * 4 + g], cols.scale[pass * 4 + g], v);
                    const int chunk = wn * 8 + 2 * g + hi;
                    *reinterpret_cast<f32x4*>(smem + row * 256 + ((chunk ^ (row & 15)) << 4)) = v;
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = wave * 64 + it * 4 + rsub;
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 256 + l15 * 16);
                const int chunk = l15 ^ (row & 15);
                const int n = n0 + (chunk >> 3) * 64 + pass * 32 + (chunk & 7) * 4;
                if (m0 + row < M) {
                    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)(m0 + row) * ep.ldo + n);
                    if constexpr (EPI == AMDS_EPI_RESIDUAL) v += *p;
                    *p = v;
                }
            }
        }
    }
}

template <typename T, int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_pp_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K, EpiArgs ep,
               int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    constexpr int BM = 256, BN = 128, BK = 32, NT = 256;
    constexpr int ROWB = BK * 2;                                   // 64 bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;   // 16 KB, 24 KB
    constexpr int FM = 4, FN = 2;
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

    // ---- phase offset: the second workgroup to arrive on this CU waits pp_delay x 10 ns ----
    if (ep.pp_slots) {
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            const int key = ((xcc & 15) << 8) | ((hw >> 8) & 255);
            const int ticket = atomicAdd(ep.pp_slots + key, 1);
            if (ticket == 1) {
                const unsigned long long t0 = wall_clock64();
                while (wall_clock64() - t0 < (unsigned long long)ep.pp_delay) __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
    }

    // ---- copy addressing: A 1024 + W 512 16-byte chunks per K tile: 4 + 2 per thread ----
    const T* src[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int c = (it < 4 ? it : it - 4) * NT + tid, row = c >> 2, cp = c & 3, sc = cp ^ ((row >> 2) & 3);
        if (it < 4) src[it] = A + (long)min(m0 + row, M - 1) * lda + sc * 8;
        else src[it] = W + (long)(n0 + row) * ldw + sc * 8;
    }
    auto issue_tile = [&](int kt, int stage) {
        char* st = smem + stage * STAGE;
        const int koff = kt * BK;
#pragma unroll
        for (int it = 0; it < 6; ++it)
            glds16(src[it] + koff, st + (it < 4 ? (it * NT + wave * 64) * 16 : A_BYTES + ((it - 4) * NT + wave * 64) * 16));
    };

    const int swz = (l31 >> 2) & 3;
    const int a_off = (wm * 128 + l31) * ROWB;
    const int w_off = A_BYTES + (wn * 64 + l31) * ROWB;
    const int c0 = ((0 + hi) ^ swz) << 4, c1 = ((2 + hi) ^ swz) << 4;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    vec8 af[2][FM], wf[2][FN];
    auto load_frags = [&](int stage, int ks) {
        const char* sb = smem + stage * STAGE;
        const int co = ks ? c1 : c0;
#pragma unroll
        for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const vec8*>(sb + a_off + i * 32 * ROWB + co);
#pragma unroll
        for (int j = 0; j < FN; ++j) wf[ks][j] = *reinterpret_cast<const vec8*>(sb + w_off + j * 32 * ROWB + co);
    };
    auto compute = [&](int ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = Act<T>::mfma32(wf[ks][j], af[ks][i], acc[i][j]);
    };

#define AMDS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define AMDS_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define AMDS_BARRIER()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    const int nk = K / BK;            // >= 2 (host checks)
    issue_tile(0, 0);
    issue_tile(1, 1);
    AMDS_WAIT_VM(6);
    AMDS_BARRIER();
    load_frags(0, 0);

    // one K tile; ISSUE: request tile kt+2, NEXT: tile kt+1 exists.  Each half opens with an MFMA and slips one
    // ds_read (and one LDS-DMA piece) behind each of the next six, so the matrix pipe never waits on issue.
    int s0 = 0, s1 = 1, s2 = 2;       // stages of tiles kt, kt+1, kt+2
    auto k_tile = [&](int kt, auto issue_c, auto next_c) {
        constexpr bool ISSUE = decltype(issue_c)::value, NEXT = decltype(next_c)::value;
        if constexpr (ISSUE) issue_tile(kt + 2, s2);
        load_frags(s0, 1);
        compute(0);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if constexpr (ISSUE) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        AMDS_WAIT_LGKM(0);            // k-step 1 fragments landed: this wave is done reading stage s0
        if constexpr (ISSUE) AMDS_WAIT_VM(6); else AMDS_WAIT_VM(0);     // this wave's pieces of tile kt+1 landed
        AMDS_BARRIER();
        if constexpr (NEXT) load_frags(s1, 0);
        compute(1);
        if constexpr (NEXT) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int t = s0; s0 = s1; s1 = s2; s2 = t;
    };
    int kt = 0;
    for (; kt < nk - 2; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
    k_tile(kt++, std::false_type{}, std::true_type{});
    k_tile(kt, std::false_type{}, std::false_type{});
#undef AMDS_WAIT_VM
#undef AMDS_WAIT_LGKM
#undef AMDS_BARRIER

    if constexpr (epi_is_staged<EPI>()) {
        // every wave passed the last barrier after its final reads of the ring were retired
        epilogue_staged_pp<EPI, T>(acc, ep, smem, m0, n0, M, wm, wn, wave, lane);
    } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * 128 + i * 32 + l31;
            if (m < M) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * hi;
                        epilogue4<EPI, T>(ep, m, n, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                                          acc[i][j][4 * g + 3]);
                    }
            }
        }
    }
}

template <typename T, int EPI>
static int launch_gemm_pp(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                          hipStream_t st) {
    static int LDS = 0;
    if (!LDS) {
        const char* e = getenv("AMDS_GEMM_PP_LDS");       // experiment: > 80 KB forces one workgroup per CU
        LDS = e ? atoi(e) : 3 * (256 + 128) * 64;
    }
    auto kern = gemm_pp_kernel<T, EPI>;
    static bool attr_set = false;
    static int delay_pct = -1;          // start offset as a percentage of the estimated tile period (AMDS_GEMM_PP, default 50)
    static int* slots = nullptr;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    if (delay_pct < 0) {
        const char* e = getenv("AMDS_GEMM_PP");
        delay_pct = e ? atoi(e) : 50;
    }
    EpiArgs ep2 = ep;
    if (delay_pct > 0) {
        static int* s_slots = nullptr;
        if (!s_slots) AMDS_HIP(hipMalloc(&s_slots, 4096 * sizeof(int)));
        slots = s_slots;
        AMDS_HIP(hipMemsetAsync(slots, 0, 4096 * sizeof(int), st));
        // one workgroup's tile period when two share a CU at ~4 TFLOP/s per CU: 2 tiles x 2*256*128*K flop
        const double period_us = 2.0 * 2.0 * 256 * 128 * K / 4.0e6;
        ep2.pp_slots = slots;
        ep2.pp_delay = (int)(period_us * delay_pct);   // 10-ns ticks: period_us * 100 * pct / 100
    }
    const int tiles_m = cdiv(M, 256), tiles_n = N / 128;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep2, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_pp_kernel");
    return AMDS_OK;
}

}  // namespace amds
