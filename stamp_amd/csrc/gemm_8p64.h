// gemm_8p64.h -- 256 x 256 x 64 block tile, EIGHT waves (2 x 4; wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16 fragments, fp32
// accumulators) in two staggered groups: kernel id 8.  Dispatched for the PATCH epilogue (the patch-embedding GEMM) and as the
// A/B baseline of the four-wave kernels (gemm_4w16.h = production, gemm_4w64.h).
//
// Staggered two-group pipeline: waves w and w+4 share a SIMD.  The K loop is cut into phases `LOAD | s_barrier | COMPUTE |
// s_barrier` (LOAD = fragment ds_read_b128 + LDS-DMA requests, COMPUTE = 16 MFMAs); group 1 runs one barrier behind group 0, so
// every SIMD always has one wave computing while its partner loads.  W is the MFMA "A" operand and the activation the "B"
// operand: a lane then holds one output row and 4 consecutive columns per register group (C-transposed fragments), so bias /
// LayerScale are float4 loads and stores are 8-16 B per lane.  LDS rows are 128 bytes (K tile of 64) in a 2-deep ring of 64 KB
// stages; 16-B chunks are XOR-swizzled on the SOURCE address of the LDS-DMA (it writes lane-linearly) and again on the read
// (SQ_LDS_BANK_CONFLICT = 0 measured).  tools/ubench/copy_bench.hip: the global->LDS path moves 23.6 TB/s with 128-byte rows
// (8 full lines per wave instruction) against 15.5 TB/s with 64-byte rows, and the copy path, not the MFMA pipe, bounded the
// BK = 32 predecessor of this kernel.  A K tile spans two phases (k-steps {0,1} and {2,3}); all 8 copy pieces of tile kt+1 are
// issued in the first phase of tile kt and must have landed (vmcnt(0): nothing else is in flight) before the barrier that ends
// tile kt.
//   RAW  tile kt+1 is first read in the slot after every wave executed vmcnt(0) + the closing barrier of tile kt.
//   WAR  stage (kt+1)&1 held tile kt-1, whose last reads were retired (lgkmcnt(0)) before the barrier closing tile
//        kt-1; the copies are issued after that barrier.
#pragma once
#include "gemm_kernel.h"
#include "gemm_epilogue.h"

namespace amds {

template <typename T, int EPI>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_8p64_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K,
                 EpiArgs ep, int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
    constexpr int ROWB = BK * 2;                       // 128 bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;   // 32 KB, 64 KB
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
    if (ep.nbatch > 1) {       // batched / split-K: operands and output of batch blockIdx.y
        const long by = blockIdx.y;
        A += by * ep.bsA;
        W += by * ep.bsW;
        constexpr bool F32OUT = (EPI == AMDS_EPI_RESIDUAL || EPI == AMDS_EPI_BIAS_F32 || EPI == AMDS_EPI_PATCH ||
                                 EPI == AMDS_EPI_BIAS_GELU_F32 || EPI == AMDS_EPI_BIAS_RELU_F32);
        if (F32OUT) ep.out = reinterpret_cast<float*>(ep.out) + by * ep.bsOut;
        else ep.out = reinterpret_cast<T*>(ep.out) + by * ep.bsOut;
    }

    // ---- copy addressing: 4096 16-byte chunks per tile, 8 per thread; chunk index XOR (row>>1)&7 on the source.
    // Buffer form (buffer_load_dwordx4 ... offen lds): a constant 32-bit byte offset per piece, the K advance in the scalar
    // offset -> no 64-bit vector address arithmetic in the loop; rows past M are out of range and read as 0.
    const int rows_a = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(A + (long)m0 * lda), 0, (int)((((long)rows_a - 1) * lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(W + (long)n0 * ldw), 0, (int)(((long)(BN - 1) * ldw + K) * 2), 0x00020000);
    int voff[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = (it & 3) * NT + tid, row = c >> 3, cp = c & 7, sc = cp ^ ((row >> 1) & 7);
        voff[it] = (int)(((long)row * (it < 4 ? lda : ldw) + sc * 8) * 2);
    }
    auto issue_tile = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
        const int koff = kt * BK * 2;
#pragma unroll
        for (int it = 0; it < 8; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(it < 4 ? rsrc_a : rsrc_w,
                                                     (lptr_t)(st + (it >> 2) * A_BYTES + ((it & 3) * NT + wave * 64) * 16), 16, voff[it],
                                                     koff, 0, 0);
    };

    const int swz = (l31 >> 1) & 7;
    const int a_off = (grp * 128 + l31) * ROWB;
    const int w_off = A_BYTES + (wc * 64 + l31) * ROWB;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    vec8 af[2][FM], wf[2][FN];
    auto load_frags = [&](int kt, int h) {
        const char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = (((h * 2 + ks) * 2 + hi) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < FM; ++i) af[ks][i] = *reinterpret_cast<const vec8*>(sb + a_off + i * 32 * ROWB + co);
#pragma unroll
            for (int j = 0; j < FN; ++j) wf[ks][j] = *reinterpret_cast<const vec8*>(sb + w_off + j * 32 * ROWB + co);
        }
    };
    auto compute = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = Act<T>::mfma32(wf[ks][j], af[ks][i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

#define AMDS_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define AMDS_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define AMDS_BARRIER()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    const int nk = K / BK;
    issue_tile(0);
    AMDS_WAIT_VM0();
    AMDS_BARRIER();
    if (grp == 1) AMDS_BARRIER();

    for (int kt = 0; kt < nk; ++kt) {
        // phase 0 of tile kt: k-steps 0,1 ; request tile kt+1
        load_frags(kt, 0);
        if (kt + 1 < nk) issue_tile(kt + 1);
        AMDS_WAIT_LGKM0();
        AMDS_BARRIER();
        compute();
        AMDS_BARRIER();
        // phase 1: k-steps 2,3 ; tile kt+1 must have landed before the slot that follows
        load_frags(kt, 1);
        AMDS_WAIT_LGKM0();
        if (grp == 1) AMDS_WAIT_VM0();
        AMDS_BARRIER();
        compute();
        if (grp == 0) AMDS_WAIT_VM0();
        AMDS_BARRIER();
    }
    if (grp == 0) AMDS_BARRIER();
#undef AMDS_WAIT_VM0
#undef AMDS_WAIT_LGKM0
#undef AMDS_BARRIER

    if constexpr (epi_is_staged<EPI>()) {
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

template <typename T, int EPI>
static int launch_gemm_8p64(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                            hipStream_t st) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_8p64_kernel<T, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, ep.nbatch), dim3(512), LDS, st, reinterpret_cast<const T*>(A), lda,
                       reinterpret_cast<const T*>(W), ldw, M, N, K, ep, tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_8p64_kernel");
    return AMDS_OK;
}

}  // namespace amds
