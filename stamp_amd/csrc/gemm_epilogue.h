// gemm_epilogue.h -- LDS-staged, fully coalesced epilogue for the 256x256 block tile (8 waves as 2 x 4, wave
// tile 128 x 64, C^T fragments: lane (l31, hi) holds row m = l31 and 4 consecutive columns per register group).
//
// Storing straight from the accumulator layout costs one 8-byte (fp16) or 16-byte (fp32) piece per lane per
// instruction, each lane on a different row: 32 partial lines per store instruction.  Measured on the 8-phase
// kernel with an empty main loop, that tail alone is 256 us for the 65536 x 4096 fp16 output (2.1 TB/s).
// Here the tile is first written to LDS (free once the K loop is done: 128 KB = 256 rows x 512 B), rows XOR-
// swizzled at 16-byte granularity, then read back row-wise so that every global access is 16 B per lane and
// 512 contiguous bytes per half-wave:
//   fp16 outputs   : one pass  (256 x 256 x 2 B = 128 KB)
//   fp32 outputs / residual read-modify-write : two passes of 128 columns (fragment column j = pass)
#pragma once
#include "gemm_kernel.h"

namespace amds {

// Caller guarantees: every wave has finished reading the K-loop stages (a barrier has been passed).
template <int EPI, typename T>
__device__ __forceinline__ void epilogue_staged_256(f32x16 (&acc)[4][2], const EpiArgs& ep, char* smem, int m0, int n0, int M,
                                                    int grp, int wc, int wave, int lane) {
    typedef typename Act<T>::vec4 vec4;
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
    EpiCols<8> cols;      // [j * 4 + g]: columns n0 + wc * 64 + j * 32 + 8 g + 4 hi
    epi_cols_load<EPI>(ep, cols, [&](int q) { return n0 + wc * 64 + (q >> 2) * 32 + 8 * (q & 3) + 4 * hi; });
    if constexpr (F16OUT) {
        auto values = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = grp * 128 + i * 32 + l31;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        f32x4 v0 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        f32x4 v1 = {acc[i][j][4 * g + 4], acc[i][j][4 * g + 5], acc[i][j][4 * g + 6], acc[i][j][4 * g + 7]};
                        epi_value_pair<EPI, FAST>(ep, cols.bias[j * 4 + g], cols.scale[j * 4 + g], cols.bias[j * 4 + g + 1],
                                                  cols.scale[j * 4 + g + 1], v0, v1);
                        const vec4 o0 = Act<T>::from_f32x4(v0), o1 = Act<T>::from_f32x4(v1);
                        const int chunk = wc * 8 + j * 4 + g;
                        // the 8-byte half inside the 16-byte chunk is XOR-ed with row bit 3: the 16 lanes of one LDS cycle then
                        // hit 32 distinct banks (without it rows r and r+8 collide 2-way: SQ_LDS_BANK_CONFLICT = 4 cycles per store)
                        const int half = (hi ^ ((l31 >> 3) & 1)) * 8;
                        *reinterpret_cast<vec4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4) + half) = o0;
                        *reinterpret_cast<vec4*>(smem + row * 512 + (((chunk + 1) ^ (row & 31)) << 4) + half) = o1;
                    }
            }
        };
        if (ep.bias != nullptr && ep.acc_scale == 1.0f) values(std::true_type{}); else values(std::false_type{});
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = wave * 32 + it * 2 + hi;
            u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 512 + l31 * 16);
            if ((it >> 2) & 1) v = u32x4{v[2], v[3], v[0], v[1]};       // row bit 3 set: the halves were stored swapped
            const int chunk = l31 ^ (row & 31);
            if (m0 + row < M)
                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + chunk * 8) = v;
        }
    } else {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = grp * 128 + i * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][pass][4 * g], acc[i][pass][4 * g + 1], acc[i][pass][4 * g + 2], acc[i][pass][4 * g + 3]};
                    v = epi_value<EPI>(ep, cols.bias[pass * 4 + g], cols.scale[pass * 4 + g], v);
                    const int chunk = wc * 8 + 2 * g + hi;
                    *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = v;
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = wave * 32 + it * 2 + hi;
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 512 + l31 * 16);
                const int chunk = l31 ^ (row & 31);
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

}  // namespace amds
