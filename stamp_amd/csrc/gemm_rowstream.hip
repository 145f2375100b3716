// gemm_rowstream.hip -- weights-stationary GEMM for the narrow layers of the Swin stages 1-2 (K = 96 / 192 / 384,
// N <= 768): out[M,N] = epilogue( [LayerNorm](A)[M,K] * W[N,K]^T ).
//
// Why a second GEMM design: with K <= 192 a block tile's K loop is 2-3 iterations, so the tiled kernels
// (gemm_kernel.h) spend their time in the prologue/epilogue latency of each 128-row tile (measured: 250-350 us for
// the 802816 x {288,384} x 96 layers of CTransPath stage 1 = 2.2-3 TB/s of compulsory traffic, unchanged by a
// coalesced epilogue).  Here the roles are inverted:
//   * a workgroup (8 waves) stages its slice of W ONCE into LDS, already in MFMA fragment order (1 KB per
//     (n-fragment, k-step): lane l reads its 16 bytes at lane*16 -> linear, conflict-free ds_read_b128), and then
//     streams row groups through it (persistent grid: ~one workgroup per CU);
//   * the activation never touches LDS: a lane's 16-byte MFMA B fragment IS a contiguous piece of one row, so each
//     wave loads its 32*FM rows straight from global memory into registers (whole K);
//   * optional fused LayerNorm: A is the fp32 residual stream; the two lanes (l31, hi=0/1) that share a row hold all
//     K values of it between them -> statistics are an in-lane sum plus one cross-half shuffle; the normalised row
//     is rounded to the operand type in registers.  The separate LayerNorm kernel and its round trip disappear;
//   * epilogue through a wave-private LDS transpose so that global stores / residual read-modify-writes are
//     128-byte row segments (16 bytes per lane) instead of 8-byte pieces on 32 different rows.
// MFMA orientation as everywhere else: W is the A operand, the activation the B operand (lane = row).
#include "gemm_kernel.h"

namespace amds {

constexpr int RS_WAVES = 8;
constexpr int RS_STG_PITCH = 144;                       // bytes per staged row: 128 data + 16 pad (2-way conflicts at most)
constexpr int RS_STG_BYTES = 32 * RS_STG_PITCH;         // per 32-row fragment

template <typename T, int KS, int FM, int EPI, bool LNF>
__global__ void __launch_bounds__(64 * RS_WAVES) rowstream_kernel(const void* __restrict__ Aptr, long lda, const T* __restrict__ W,
                                                                  long ldw, int M, int nf, EpiArgs ep, const float* __restrict__ ln_g,
                                                                  const float* __restrict__ ln_b, float eps, int groups) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int K = KS * 16;
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int n_base = blockIdx.y * nf * 32;            // first output column of this workgroup's slice

    // ---- stage the W slice in fragment order (async global -> LDS, per-lane gather addresses) ----
    char* s_w = smem;
    float* s_ln = reinterpret_cast<float*>(smem + (size_t)nf * KS * 1024);
    char* s_stg = reinterpret_cast<char*>(s_ln) + (LNF ? 2 * K * 4 : 0) + wave * (FM * RS_STG_BYTES);
    for (int blk = wave; blk < nf * KS; blk += RS_WAVES) {
        const int j = blk / KS, ks = blk - j * KS;
        glds16(W + (long)(n_base + 32 * j + l31) * ldw + 16 * ks + 8 * hi, s_w + blk * 1024);
    }
    if constexpr (LNF) {
        for (int i = tid; i < K; i += 64 * RS_WAVES) { s_ln[i] = ln_g[i]; s_ln[K + i] = ln_b[i]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int g = blockIdx.x * RS_WAVES + wave; g < groups; g += gridDim.x * RS_WAVES) {
        const int row0 = g * 32 * FM;
        // ---- activation fragments: whole K of 32*FM rows, registers only ----
        vec8 xf[FM][KS];
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const int row = min(row0 + f * 32 + l31, M - 1);
            if constexpr (LNF) {
                const float* xr = reinterpret_cast<const float*>(Aptr) + (long)row * lda + 8 * hi;
                f32x4 raw[KS][2];
                float s = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    raw[ks][0] = *reinterpret_cast<const f32x4*>(xr + 16 * ks);
                    raw[ks][1] = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 4);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    s += ((raw[ks][0][0] + raw[ks][0][1]) + (raw[ks][0][2] + raw[ks][0][3])) +
                         ((raw[ks][1][0] + raw[ks][1][1]) + (raw[ks][1][2] + raw[ks][1][3]));
                s += __shfl_xor(s, 32, 64);
                const float mean = s * (1.0f / K);
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float d = raw[ks][h2][e] - mean; q = fmaf(d, d, q); }
                q += __shfl_xor(q, 32, 64);
                const float rstd = rsqrtf(q * (1.0f / K) + eps);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const f32x4 gg = *reinterpret_cast<const f32x4*>(s_ln + 16 * ks + 8 * hi + 4 * h2);
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(s_ln + K + 16 * ks + 8 * hi + 4 * h2);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            xf[f][ks][4 * h2 + e] = Act<T>::from_f32(fmaf((raw[ks][h2][e] - mean) * rstd, gg[e], bb[e]));
                    }
            } else {
                const T* xr = reinterpret_cast<const T*>(Aptr) + (long)row * lda + 8 * hi;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xf[f][ks] = *reinterpret_cast<const vec8*>(xr + 16 * ks);
            }
        }
        // ---- n fragments ----
        for (int j = 0; j < nf; ++j) {
            f32x16 acc[FM];
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
            const char* wj = s_w + (size_t)j * KS * 1024 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 wf = *reinterpret_cast<const vec8*>(wj + ks * 1024);
#pragma unroll
                for (int f = 0; f < FM; ++f) acc[f] = Act<T>::mfma32(wf, xf[f][ks], acc[f]);
            }
            const int n_frag = n_base + 32 * j;
            if constexpr (F16OUT) {
                // two fragments (64 columns = 128 bytes per row) are collected before a flush
                const int half = j & 1;
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v = {acc[f][4 * g4], acc[f][4 * g4 + 1], acc[f][4 * g4 + 2], acc[f][4 * g4 + 3]};
                        v = epi_value<EPI>(ep, n_frag + 8 * g4 + 4 * hi, v);
                        vec4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = Act<T>::from_f32(v[e]);
                        *reinterpret_cast<vec4*>(s_stg + f * RS_STG_BYTES + l31 * RS_STG_PITCH + half * 64 + 16 * g4 + 8 * hi) = o;
                    }
                if (half == 1 || j == nf - 1) {
                    __builtin_amdgcn_wave_barrier();
                    const int ncol0 = n_frag - half * 32;              // first column of the staged segment
                    const int nchunk = (half + 1) * 4;                 // 16-byte chunks per row: 4 (one fragment) or 8
#pragma unroll
                    for (int f = 0; f < FM; ++f)
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = it * 8 + (lane >> 3), c = lane & 7;
                            const int row = row0 + f * 32 + r;
                            const u32x4 v = *reinterpret_cast<const u32x4*>(s_stg + f * RS_STG_BYTES + r * RS_STG_PITCH + c * 16);
                            if (row < M && c < nchunk)
                                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)row * ep.ldo + ncol0 + c * 8) = v;
                        }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v = {acc[f][4 * g4], acc[f][4 * g4 + 1], acc[f][4 * g4 + 2], acc[f][4 * g4 + 3]};
                        v = epi_value<EPI>(ep, n_frag + 8 * g4 + 4 * hi, v);
                        *reinterpret_cast<f32x4*>(s_stg + f * RS_STG_BYTES + l31 * RS_STG_PITCH + 32 * g4 + 16 * hi) = v;
                    }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = it * 8 + (lane >> 3), c = lane & 7;
                        const int row = row0 + f * 32 + r;
                        f32x4 v = *reinterpret_cast<const f32x4*>(s_stg + f * RS_STG_BYTES + r * RS_STG_PITCH + c * 16);
                        if (row < M) {
                            f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)row * ep.ldo + n_frag + c * 4);
                            if constexpr (EPI == AMDS_EPI_RESIDUAL) v += *p;
                            *p = v;
                        }
                    }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

template <typename T, int KS, int FM, int EPI, bool LNF>
static int launch_rowstream(const void* A, long lda, const void* W, long ldw, int M, int N, const EpiArgs& ep, const float* ln_g,
                            const float* ln_b, float eps, hipStream_t st) {
    constexpr int K = KS * 16;
    // slice N so that the resident W slice stays <= 72 KB (staging + LN parameters take the rest of the 160 KB)
    const int nfrag = N / 32;
    int slices = 1;
    while ((nfrag % slices) != 0 || (size_t)(nfrag / slices) * KS * 1024 > 73728) ++slices;
    const int nf = nfrag / slices;
    const size_t lds = (size_t)nf * KS * 1024 + (LNF ? 2 * K * 4 : 0) + (size_t)RS_WAVES * FM * RS_STG_BYTES;
    auto kern = rowstream_kernel<T, KS, FM, EPI, LNF>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int groups = cdiv(M, 32 * FM);
    int gx = cdiv(groups, RS_WAVES);
    const int cap = 256 / slices > 0 ? 256 / slices : 1;            // ~one resident workgroup per CU over all slices
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(64 * RS_WAVES), lds, st, A, lda, reinterpret_cast<const T*>(W), ldw, M, nf, ep,
                       ln_g, ln_b, eps, groups);
    AMDS_LAUNCH_CHECK("rowstream_kernel");
    return AMDS_OK;
}

template <typename T>
static int rowstream_dispatch(const void* A, long lda, bool lnf, const void* W, long ldw, int M, int N, int K, int epi, const EpiArgs& ep,
                              const float* ln_g, const float* ln_b, float eps, hipStream_t st) {
#define RS_CASE(KSV, FMV, EPIV, LNV) \
    if (K == KSV * 16 && epi == EPIV && lnf == LNV) return launch_rowstream<T, KSV, FMV, EPIV, LNV>(A, lda, W, ldw, M, N, ep, ln_g, ln_b, eps, st);
    RS_CASE(6, 2, AMDS_EPI_BIAS, true)
    RS_CASE(6, 2, AMDS_EPI_BIAS_GELU, true)
    RS_CASE(6, 2, AMDS_EPI_RESIDUAL, false)
    RS_CASE(12, 1, AMDS_EPI_BIAS, true)
    RS_CASE(12, 1, AMDS_EPI_BIAS_GELU, true)
    RS_CASE(12, 2, AMDS_EPI_RESIDUAL, false)
    RS_CASE(24, 1, AMDS_EPI_RESIDUAL, false)
    RS_CASE(24, 1, AMDS_EPI_BIAS_F32, false)
#undef RS_CASE
    set_error("amds_gemm_rowstream: no kernel for K=%d epi=%d fused_ln=%d", K, epi, (int)lnf);
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
    AMDS_REQUIRE(lda >= K && lda % (lnf ? 4 : 8) == 0 && ldw >= K && ldw % 8 == 0 && ldo % 4 == 0, "amds_gemm_rowstream: bad strides");
    AMDS_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0, "amds_gemm_rowstream: pointers must be 16-byte aligned");
    if (M == 0) return AMDS_OK;
    EpiArgs ep;
    ep.out = out; ep.ldo = ldo; ep.bias = bias; ep.scale = nullptr; ep.pos = nullptr; ep.np = ep.T = ep.P = 0; ep.acc_scale = 1.0f;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_GEMM, 2.0 * M * (double)N * K, st);
    if (dtype == AMDS_F16) return rowstream_dispatch<f16>(A, lda, lnf, W, ldw, M, N, K, epi, ep, ln_gamma, ln_beta, ln_eps, st);
    if (dtype == AMDS_BF16) return rowstream_dispatch<bf16>(A, lda, lnf, W, ldw, M, N, K, epi, ep, ln_gamma, ln_beta, ln_eps, st);
    set_error("amds_gemm_rowstream: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}
