// gemm_fp8.hip -- OPT-IN fp8 (OCP e4m3) variant of the tile encoder's GEMM on v_mfma_f32_16x16x128_f8f6f4 (gfx950's K = 128 form, twice the
// fp16 MFMA rate: 5 PFLOP/s dense peak), fp32 accumulate, per-row activation scales and per-output-channel weight scales applied in the
// epilogue:
//     out[m][n] = act( acc[m][n] * rowscale[m] * colscale[n] + bias[n] )           acc = sum_k A8[m][k] * W8[n][k]   (both e4m3)
// BASELINE.json configs[4] names "fp8 MFMA weights"; the reference itself computes in fp32 (src/stamp/preprocessing/__init__.py:324-325), so
// this is never the default: e4m3 carries 3 mantissa bits (2^-4 relative rounding), the accuracy delta is measured and stated
// (tests/test_gpu_fp8.py).
//
// Structure = gemm_4w16.h (256 x 256 tile, four waves with 128 x 128 wave tiles, 256 accumulator registers pinned to AGPRs, two 64 KB LDS
// stages of 128-byte rows filled by buffer-form LDS-DMA with the same XOR swizzle) with K tiles of 128 fp8 values: a lane's two 16-byte
// fragments of a row (chunks kb and 4 + kb: any 32 of the row's 128 bytes do, as long as A and W take the same ones -- tools/ubench/
// fp8_mfma_probe.hip) are ONE operand of ONE instruction, so a K tile is 64 MFMAs of 32 cycles instead of 128 of 16.  Because an MFMA needs
// both halves, the fragment pipeline differs from gemm_4w16.h: the 8 weight fragments of a K tile stay in registers for the whole tile (64
// VGPRs, double-buffered across tiles), activation fragments are fetched one 16-row block ahead, the next tile's LDS-DMA requests go out
// behind the first four row blocks, its weight fragments are read behind the last two (after the barrier that says it has landed).
#include <type_traits>
#include "gemm_kernel.h"

namespace amds {

typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_fp8_agpr(const v8i& a, const v8i& b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x4 agpr_read8(const f32x4& a) {
    f32x4 v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(a[0]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(a[1]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(a[2]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(a[3]));
    return v;
}

struct Fp8Epi {
    void* out;                 // f16 [M][ldo] (EPI BIAS / BIAS_GELU) or fp32 [M][ldo] read-modify-write (EPI RESIDUAL)
    long ldo;
    const float* bias;         // [N] or null
    const float* colscale;     // [N] or null (weight scale per output channel; for RESIDUAL the caller multiplies LayerScale in)
    const float* rowscale;     // [M] or null (activation scale per row)
    // BIAS / BIAS_GELU only: e4m3 output instead of f16 -- out8[m][n] = e4m3(value / out_rowscale[m]) (the A operand of the next fp8 GEMM)
    uint8_t* out8 = nullptr;
    long ldo8 = 0;
    const float* out_rowscale = nullptr;
};

template <int EPI>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_fp8_kernel(const uint8_t* __restrict__ A, long lda, const uint8_t* __restrict__ W, long ldw, int M, int N, int K, Fp8Epi ep, int tiles_m,
                int tiles_n) {
    constexpr int BM = 256, BN = 256, BK = 128, NT = 256;
    constexpr int ROWB = 128;
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
    constexpr int FI = 8, FJ = 8, GROUP_M = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kb = lane >> 4;
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
    const int rows_a = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A + (long)m0 * lda), 0, (int)(((long)rows_a - 1) * lda + K), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(W + (long)n0 * ldw), 0, (int)((long)(BN - 1) * ldw + K), 0x00020000);
    int voff[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int c = (it & 7) * NT + tid, row = c >> 3, cp = c & 7, sc = cp ^ ((row >> 1) & 7);
        voff[it] = (int)((long)row * (it < 8 ? lda : ldw) + sc * 16);
    }
    auto issue_pieces = [&](int kt, int lo, int hi_) {
        char* st = smem + (kt & 1) * STAGE;
        const int koff = kt * BK;
#pragma unroll
        for (int it = 0; it < 16; ++it)
            if (it >= lo && it < hi_) {
                if (it < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(st + ((it & 7) * NT + wave * 64) * 16), 16, voff[it], koff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(st + A_BYTES + ((it & 7) * NT + wave * 64) * 16), 16, voff[it], koff, 0, 0);
            }
    };
    const int swz = (l15 >> 1) & 7;
    const int a_off = (wm * 128 + l15) * ROWB;
    const int w_off = A_BYTES + (wn * 128 + l15) * ROWB;
    const int co0 = ((0 * 4 + kb) ^ swz) << 4, co1 = ((1 * 4 + kb) ^ swz) << 4;

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_pieces(0, 0, 16);
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        if (i == FI - 1)
            asm volatile("s_nop 7" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
        else
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
    }
    __builtin_amdgcn_sched_barrier(0);

    v8i wfr[FJ], afr[2];
    auto load_w = [&](int kt, int j) {
        const char* sb = smem + (kt & 1) * STAGE + w_off + j * 16 * ROWB;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + co0), hi4 = *reinterpret_cast<const u32x4*>(sb + co1);
        wfr[j] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
    };
    auto load_a = [&](int kt, int s, int i) {
        const char* sb = smem + (kt & 1) * STAGE + a_off + i * 16 * ROWB;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + co0), hi4 = *reinterpret_cast<const u32x4*>(sb + co1);
        afr[s] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
    };
#define AMDS_BARRIER()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
    const int nk = K / BK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMDS_BARRIER();
#pragma unroll
    for (int j = 0; j < FJ; ++j) load_w(0, j);
    load_a(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);

    // One K tile = 8 row blocks x 8 MFMAs.  Behind the MFMAs: the next row block's activation fragment (2 reads); rows 0..3: the 16 LDS-DMA
    // pieces of tile kt+1 (its stage was released by the barrier that ended tile kt-1); after row 5: vmcnt(0) + barrier = tile kt+1 has
    // landed; row 7: weight fragment j of tile kt+1 replaces fragment j right behind its last use (one register set: 64 VGPRs).
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const bool next = kt + 1 < nk;                 // wave-uniform
#pragma unroll
        for (int i = 0; i < FI; ++i) {
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                mfma_fp8_agpr(wfr[j], afr[i & 1], acc[i][j]);
                if (j == 0) {
                    if (i < FI - 1) load_a(kt, (i + 1) & 1, i + 1);
                    else if (next) load_a(kt + 1, 0, 0);
                }
                if (i < 4 && (j & 1) == 1 && next) issue_pieces(kt + 1, i * 4 + (j >> 1), i * 4 + (j >> 1) + 1);
                if (i == FI - 1 && next) load_w(kt + 1, j);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i == 5 && next) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                AMDS_BARRIER();
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        AMDS_BARRIER();                                // every wave is done reading this tile's stage: the tile after next may overwrite it
    }
#undef AMDS_BARRIER
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        if (i == 0)
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
        else
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: block (i, j): row = wm*128 + 16 i + l15, columns wn*128 + 16 j + 4 kb .. + 3; LDS-staged, coalesced row-wise stores ----
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU);
    typedef Act<f16>::vec4 h4;
    float rsc[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) rsc[i] = ep.rowscale ? ep.rowscale[min(m0 + wm * 128 + i * 16 + l15, M - 1)] : 1.0f;
    if constexpr (EPI == AMDS_EPI_SWIGLU) {
        // packed fc1 (timm SwiGLUPacked, rows interleaved in blocks of 32: [gate 32 | value 32]): column blocks j = 0, 1, 4, 5 hold gates, j + 2 the
        // values of the same outputs -- in the same lane.  out[m][c] = silu(g) * v, f16, N / 2 columns: 256 rows x 128 columns per workgroup,
        // staged through LDS in 256-byte rows (16-byte chunk index XOR row & 15) and stored row-wise, 16 bytes per lane.
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int row = wm * 128 + i * 16 + l15;
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {
                const int j = (jg >> 1) * 4 + (jg & 1);
                const int col = n0 + wn * 128 + j * 16 + 4 * kb;
                const f32x4 csg = ep.colscale ? *reinterpret_cast<const f32x4*>(ep.colscale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 csv = ep.colscale ? *reinterpret_cast<const f32x4*>(ep.colscale + col + 32) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 cbg = ep.bias ? *reinterpret_cast<const f32x4*>(ep.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 cbv = ep.bias ? *reinterpret_cast<const f32x4*>(ep.bias + col + 32) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 g = agpr_read8(acc[i][j]) * rsc[i] * csg + cbg;
                const f32x4 v = agpr_read8(acc[i][j + 2]) * rsc[i] * csv + cbv;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = g[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-g[e])) * v[e];
                const int c = wn * 64 + jg * 16 + 4 * kb;                       // output column inside the workgroup's 128
                *reinterpret_cast<h4*>(smem + row * 256 + (((c >> 3) ^ (row & 15)) << 4) + (c & 4) * 2) = Act<f16>::from_f32x4(o);
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = wave * 64 + it * 4 + kb;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 256 + l15 * 16);
            const int chunk = l15 ^ (row & 15);
            if (m0 + row < M) *reinterpret_cast<u32x4*>(reinterpret_cast<f16*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 / 2 + chunk * 8) = v;
        }
        return;
    }
    constexpr int NPASS = F16OUT ? 1 : 2, JP = FJ / NPASS;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass) __syncthreads();
        f32x4 cs[JP], cb[JP];
#pragma unroll
        for (int jj = 0; jj < JP; ++jj) {
            const int col = n0 + wn * 128 + (pass * JP + jj) * 16 + 4 * kb;
            cs[jj] = ep.colscale ? *reinterpret_cast<const f32x4*>(ep.colscale + col) : f32x4{1.f, 1.f, 1.f, 1.f};
            cb[jj] = ep.bias ? *reinterpret_cast<const f32x4*>(ep.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int row = wm * 128 + i * 16 + l15;
#pragma unroll
            for (int jj = 0; jj < JP; jj += 2) {
                const int j = pass * JP + jj;
                f32x4 v0 = agpr_read8(acc[i][j]) * rsc[i] * cs[jj] + cb[jj];
                f32x4 v1 = agpr_read8(acc[i][j + 1]) * rsc[i] * cs[jj + 1] + cb[jj + 1];
                if constexpr (EPI == AMDS_EPI_BIAS_GELU) {
                    f32x2 q[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
                    gelu_erf_poly2_n<4>(q);
                    v0 = f32x4{q[0][0], q[0][1], q[1][0], q[1][1]};
                    v1 = f32x4{q[2][0], q[2][1], q[3][0], q[3][1]};
                }
                if constexpr (F16OUT) {
                    const h4 o0 = Act<f16>::from_f32x4(v0), o1 = Act<f16>::from_f32x4(v1);
                    const int chunk = wn * 16 + j * 2 + (kb >> 1);
                    const int half = ((kb & 1) ^ ((l15 >> 3) & 1)) * 8;
                    *reinterpret_cast<h4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4) + half) = o0;
                    *reinterpret_cast<h4*>(smem + row * 512 + (((chunk + 2) ^ (row & 31)) << 4) + half) = o1;
                } else {
                    const int chunk = wn * 16 + jj * 4 + kb;
                    *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = v0;
                    *reinterpret_cast<f32x4*>(smem + row * 512 + (((chunk + 4) ^ (row & 31)) << 4)) = v1;
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int b8 = 0; b8 < 4; ++b8) {
            if constexpr (F16OUT) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    v[u] = *reinterpret_cast<const u32x4*>(smem + row * 512 + l31 * 16);
                    if ((u >> 2) & 1) v[u] = u32x4{v[u][2], v[u][3], v[u][0], v[u][1]};
                }
                if (ep.out8) {      // 8 f16 values of one row -> 8 e4m3 bytes, scaled by the row's output scale (the 8 scales requested together)
                    float inv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) inv[u] = ep.out_rowscale[min(m0 + wave * 64 + (b8 * 8 + u) * 2 + hi, M - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) inv[u] = __builtin_amdgcn_rcpf(inv[u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                        const int chunk = l31 ^ (row & 31);
                        const f16x8 h = __builtin_bit_cast(f16x8, v[u]);
                        int w0 = 0, w1 = 0;
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)h[0] * inv[u], (float)h[1] * inv[u], w0, false);
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)h[2] * inv[u], (float)h[3] * inv[u], w0, true);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)h[4] * inv[u], (float)h[5] * inv[u], w1, false);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)h[6] * inv[u], (float)h[7] * inv[u], w1, true);
                        if (m0 + row < M) *reinterpret_cast<u32x2*>(ep.out8 + (long)(m0 + row) * ep.ldo8 + n0 + chunk * 8) = u32x2{(unsigned)w0, (unsigned)w1};
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                        const int chunk = l31 ^ (row & 31);
                        if (m0 + row < M) *reinterpret_cast<u32x4*>(reinterpret_cast<f16*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + chunk * 8) = v[u];
                    }
                }
            } else {
                f32x4 v[8], o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    const int chunk = l31 ^ (row & 31);
                    const int n = n0 + (chunk >> 4) * 128 + pass * 64 + (chunk & 15) * 4;
                    v[u] = *reinterpret_cast<const f32x4*>(smem + row * 512 + l31 * 16);
                    o[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ep.out) + (long)min(m0 + row, M - 1) * ep.ldo + n);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    const int chunk = l31 ^ (row & 31);
                    const int n = n0 + (chunk >> 4) * 128 + pass * 64 + (chunk & 15) * 4;
                    v[u] += o[u];
                    if (m0 + row < M) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)(m0 + row) * ep.ldo + n) = v[u];
                }
            }
        }
    }
}

template <int EPI>
static int launch_fp8(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const Fp8Epi& ep, hipStream_t st) {
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_fp8_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
    ProfScope prof(PROF_GEMM_FP8, 2.0 * M * (double)N * K, st);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), LDS, st, reinterpret_cast<const uint8_t*>(A), lda, reinterpret_cast<const uint8_t*>(W), ldw, M, N, K, ep,
                       tiles_m, tiles_n);
    AMDS_LAUNCH_CHECK("gemm_fp8_kernel");
    return AMDS_OK;
}

// ---------------------------------------------------------------------------------------------
// Row-wise quantisation to e4m3: q[m][k] = e4m3(x[m][k] / s[m]), s[m] = max_k |x[m][k]| / 448 (1 if the row is all zero).  One wave per row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
template <typename TI, int MAXV>
__global__ void __launch_bounds__(256) quant_rows_e4m3_kernel(const TI* __restrict__ x, long ldx, uint8_t* __restrict__ q, long ldq, float* __restrict__ scale,
                                                              int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + (long)row * ldx;
    const int nv = cols >> 2;
    f32x4 v[MAXV];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            if constexpr (std::is_same<TI, float>::value) v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
            else {
                const Act<f16>::vec4 h = *reinterpret_cast<const Act<f16>::vec4*>(xr + c * 4);
                v[i] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            }
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3]))));
        }
    }
    mx = wave_max(mx);
    const float s = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f, inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) *reinterpret_cast<uint32_t*>(q + (long)row * ldq + c * 4) = pack4_e4m3(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
    }
    if (lane == 0) scale[row] = s;
}

// LayerNorm (nn.LayerNorm semantics, two-pass statistics in fp32 as layernorm_kernel) fused with the row quantisation: x fp32 row -> e4m3 row +
// scale (+ the normalised row's L2 norm, from which the caller bounds the next Linear's outputs).  One wave per row, row in registers.
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_quant_e4m3_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float eps, uint8_t* __restrict__ q, long ldq,
                                                                   float* __restrict__ scale, float* __restrict__ rownorm, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
    const int nv = cols >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)cols;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                qq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(qq) / (float)cols + eps);
    float mx = 0.f, n2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4), b = *reinterpret_cast<const f32x4*>(beta + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = (v[i][e] - mean) * rstd * g[e] + b[e];
                mx = fmaxf(mx, fabsf(v[i][e]));
                n2 = fmaf(v[i][e], v[i][e], n2);
            }
        }
    }
    mx = wave_max(mx);
    n2 = wave_sum(n2);
    const float sc = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f, inv = 1.0f / sc;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) *reinterpret_cast<uint32_t*>(q + (long)row * ldq + c * 4) = pack4_e4m3(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
    }
    if (lane == 0) {
        scale[row] = sc;
        if (rownorm) rownorm[row] = sqrtf(n2);
    }
}

// us[m] = (rownorm[m] * c0 + c1) / 448: a per-row scale that provably covers |Linear(h_m)| <= ||h_m|| max_n ||w_n|| + max |b| (Cauchy-Schwarz) and
// therefore |gelu(.)| too -- the output scale of a GEMM whose epilogue writes e4m3 directly (no second pass over a 16-bit copy to find the row maximum)
__global__ void row_bound_scale_kernel(const float* __restrict__ rownorm, float c0, float c1, float* __restrict__ us, int rows) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) us[r] = fmaxf(rownorm[r] * c0 + c1, 1e-30f) * (1.0f / 448.0f);
}

}  // namespace amds

using namespace amds;

extern "C" int amds_layernorm_quant_e4m3(const float* x, long ldx, const float* gamma, const float* beta, float eps, void* q, long ldq, float* scale,
                                         float* rownorm, int rows, int cols, void* stream) {
    AMDS_REQUIRE(x && gamma && beta && q && scale && rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 2048 && ldx >= cols && ldq >= cols && ldx % 4 == 0 && ldq % 4 == 0,
                 "amds_layernorm_quant_e4m3: bad arguments (cols %% 4 == 0, cols <= 2048)");
    if (rows == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)rows * cols * 5.0, st);
    if (cols <= 1024) hipLaunchKernelGGL((layernorm_quant_e4m3_kernel<4>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, gamma, beta, eps, (uint8_t*)q, ldq, scale, rownorm, rows, cols);
    else hipLaunchKernelGGL((layernorm_quant_e4m3_kernel<8>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, gamma, beta, eps, (uint8_t*)q, ldq, scale, rownorm, rows, cols);
    AMDS_LAUNCH_CHECK("layernorm_quant_e4m3_kernel");
    return AMDS_OK;
}

extern "C" int amds_row_bound_scale(const float* rownorm, float c0, float c1, float* us, int rows, void* stream) {
    AMDS_REQUIRE(rownorm && us && rows >= 0, "amds_row_bound_scale: bad arguments");
    if (rows == 0) return AMDS_OK;
    hipLaunchKernelGGL(row_bound_scale_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, rownorm, c0, c1, us, rows);
    AMDS_LAUNCH_CHECK("row_bound_scale_kernel");
    return AMDS_OK;
}

extern "C" int amds_gemm_fp8_out8(const void* A8, long lda, const void* W8, long ldw, int M, int N, int K, int epi, void* out8, long ldo8,
                                  const float* out_rowscale, const float* bias, const float* colscale, const float* rowscale, void* stream) {
    AMDS_REQUIRE(A8 && W8 && out8 && out_rowscale, "amds_gemm_fp8_out8: null pointer");
    AMDS_REQUIRE(epi == AMDS_EPI_BIAS || epi == AMDS_EPI_BIAS_GELU, "amds_gemm_fp8_out8: epilogue %d (BIAS or BIAS_GELU)", epi);
    AMDS_REQUIRE(M > 0 && N > 0 && N % 256 == 0 && K > 0 && K % 128 == 0, "amds_gemm_fp8_out8: N %% 256 == 0 and K %% 128 == 0 required (M=%d N=%d K=%d)", M, N, K);
    AMDS_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldo8 >= N && ldo8 % 8 == 0, "amds_gemm_fp8_out8: pitches");
    AMDS_REQUIRE((((uintptr_t)A8 | (uintptr_t)W8) & 15) == 0 && ((uintptr_t)out8 & 7) == 0, "amds_gemm_fp8_out8: alignment");
    Fp8Epi ep{nullptr, 0, bias, colscale, rowscale};
    ep.out8 = reinterpret_cast<uint8_t*>(out8);
    ep.ldo8 = ldo8;
    ep.out_rowscale = out_rowscale;
    hipStream_t st = (hipStream_t)stream;
    if (epi == AMDS_EPI_BIAS) return launch_fp8<AMDS_EPI_BIAS>(A8, lda, W8, ldw, M, N, K, ep, st);
    return launch_fp8<AMDS_EPI_BIAS_GELU>(A8, lda, W8, ldw, M, N, K, ep, st);
}

extern "C" int amds_gemm_fp8(const void* A8, long lda, const void* W8, long ldw, int M, int N, int K, int epi, void* out, long ldo, const float* bias,
                             const float* colscale, const float* rowscale, void* stream) {
    AMDS_REQUIRE(A8 && W8 && out, "amds_gemm_fp8: null pointer");
    AMDS_REQUIRE(M > 0 && N > 0 && N % 256 == 0 && K > 0 && K % 128 == 0, "amds_gemm_fp8: N %% 256 == 0 and K %% 128 == 0 required (M=%d N=%d K=%d)", M, N, K);
    AMDS_REQUIRE(lda % 16 == 0 && ldw % 16 == 0 && lda >= K && ldw >= K && ldo >= (epi == AMDS_EPI_SWIGLU ? N / 2 : N) && ldo % 8 == 0,
                 "amds_gemm_fp8: pitches must be multiples of 16 bytes");
    AMDS_REQUIRE((((uintptr_t)A8 | (uintptr_t)W8 | (uintptr_t)out) & 15) == 0, "amds_gemm_fp8: operands must be 16-byte aligned");
    const Fp8Epi ep{out, ldo, bias, colscale, rowscale};
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case AMDS_EPI_BIAS: return launch_fp8<AMDS_EPI_BIAS>(A8, lda, W8, ldw, M, N, K, ep, st);
        case AMDS_EPI_BIAS_GELU: return launch_fp8<AMDS_EPI_BIAS_GELU>(A8, lda, W8, ldw, M, N, K, ep, st);
        case AMDS_EPI_RESIDUAL: return launch_fp8<AMDS_EPI_RESIDUAL>(A8, lda, W8, ldw, M, N, K, ep, st);
        case AMDS_EPI_SWIGLU: return launch_fp8<AMDS_EPI_SWIGLU>(A8, lda, W8, ldw, M, N, K, ep, st);
        default: set_error("amds_gemm_fp8: epilogue %d not supported (BIAS, BIAS_GELU, SWIGLU, RESIDUAL)", epi); return AMDS_ERR_INVALID;
    }
}

extern "C" int amds_quantize_rows_e4m3(const void* x, long ldx, void* q, long ldq, float* scale, int rows, int cols, int in_dtype, void* stream) {
    AMDS_REQUIRE(x && q && scale && rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 8192 && ldx >= cols && ldq >= cols && ldx % 4 == 0 && ldq % 4 == 0,
                 "amds_quantize_rows_e4m3: bad arguments (cols %% 4 == 0, cols <= 8192)");
    if (rows == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv(rows, 4));
    if (in_dtype == AMDS_F16) {
        if (cols <= 2048) hipLaunchKernelGGL((quant_rows_e4m3_kernel<f16, 8>), grid, dim3(256), 0, st, (const f16*)x, ldx, (uint8_t*)q, ldq, scale, rows, cols);
        else hipLaunchKernelGGL((quant_rows_e4m3_kernel<f16, 32>), grid, dim3(256), 0, st, (const f16*)x, ldx, (uint8_t*)q, ldq, scale, rows, cols);
    } else if (in_dtype == AMDS_F32) {
        if (cols <= 2048) hipLaunchKernelGGL((quant_rows_e4m3_kernel<float, 8>), grid, dim3(256), 0, st, (const float*)x, ldx, (uint8_t*)q, ldq, scale, rows, cols);
        else hipLaunchKernelGGL((quant_rows_e4m3_kernel<float, 32>), grid, dim3(256), 0, st, (const float*)x, ldx, (uint8_t*)q, ldq, scale, rows, cols);
    } else { set_error("amds_quantize_rows_e4m3: input dtype must be f16 or f32"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("quant_rows_e4m3_kernel");
    return AMDS_OK;
}
