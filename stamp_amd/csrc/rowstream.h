// rowstream.h -- operand loading shared by the row-streaming kernels (gemm_rowstream.hip, swin.hip): a lane's 16-byte MFMA
// fragment is a contiguous piece of one activation row, so rows go from global memory straight into operand registers; with
// the fp32 residual stream as input, LayerNorm is applied on the way (the two lanes (l31, hi = 0/1) sharing a row hold all of
// its K values between them).
#pragma once
#include "common.h"

namespace amds {

template <typename T, int KS>
__device__ __forceinline__ void rs_load_f16(typename Act<T>::vec8 (&xf)[KS], const T* A, long lda, int row, int hi) {
    const T* xr = A + (long)row * lda + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const typename Act<T>::vec8*>(xr + 16 * ks);
}

template <int KS>
__device__ __forceinline__ void rs_load_raw(f32x4 (&raw)[KS][2], const float* A, long lda, int row, int hi) {
    const float* xr = A + (long)row * lda + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        raw[ks][0] = *reinterpret_cast<const f32x4*>(xr + 16 * ks);
        raw[ks][1] = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 4);
    }
}

// LayerNorm of one row spread over the lane pair (l31, hi = 0/1): two-pass statistics in fp32, output in operand type
template <typename T, int KS>
__device__ __forceinline__ void rs_normalise(typename Act<T>::vec8 (&xf)[KS], const f32x4 (&raw)[KS][2], const float* s_ln, int hi,
                                             float eps) {
    constexpr int K = KS * 16;
    float s = 0.f;
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
            for (int e = 0; e < 4; ++e) xf[ks][4 * h2 + e] = Act<T>::from_f32(fmaf((raw[ks][h2][e] - mean) * rstd, gg[e], bb[e]));
            if (h2 == 1 && (ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // at most 8 gamma/beta fragments in flight
        }
}

}  // namespace amds
