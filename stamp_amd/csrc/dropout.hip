// dropout.hip -- the dropout sites of the MIL `vit` training step outside the attention kernels (reference
// src/stamp/modeling/models/vision_tranformer.py): `project_features` = Linear -> GELU -> Dropout(p) (:314-318) and
// `feed_forward` = LayerNorm -> Linear -> GELU -> Dropout(0.5) -> Linear -> Dropout(0.5) (:157-169; the factory is called
// WITHOUT a dropout argument at :268-271, so the feed-forward rate is the hard-coded 0.5 whatever the config says).
// Masks are regenerated from (seed, stream, element index) in the backward (common.h: drop_*), never stored.
//   amds_gelu_dropout_fwd / _bwd   u = drop(gelu(z)) ;  dz = gelu'(z) * drop'(du)
//   amds_dropout_add               x_out = x_in + drop(y)          (second feed-forward dropout + residual add)
//   amds_dropout_cast_bwd          dy16 = (16-bit) drop'(dx)       (gradient entering fc2's backward GEMMs)
//   amds_dropout_mask / amds_attention_dropout_mask   the masks themselves as u8, for the parity tests
#include <algorithm>
#include "common.h"

namespace amds {

template <typename TI, typename TO>
__global__ void gelu_dropout_fwd_kernel(const TI* __restrict__ z, TO* __restrict__ u, long n, uint64_t seed, uint32_t stream, uint32_t thr, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float g = gelu_erf((float)z[i]);
        u[i] = (TO)(drop_keep_flat(seed, stream, i, thr) ? g * scale : 0.f);
    }
}
template <typename TZ, typename TG, typename TO>
__global__ void gelu_dropout_bwd_kernel(const TZ* __restrict__ z, const TG* __restrict__ du, TO* __restrict__ dz, long n, uint64_t seed,
                                        uint32_t stream, uint32_t thr, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = (float)z[i];
        const float d = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
        dz[i] = (TO)(drop_keep_flat(seed, stream, i, thr) ? (float)du[i] * d * scale : 0.f);
    }
}
// 8 elements per lane: 16-byte loads / stores on the 16-bit tensors, ONE hash per element pair (the scalar kernels above compute every pair's hash twice and
// move 2 bytes per lane and instruction: 1.9-2.0 TB/s on tensors that HBM could stream at 5).  Same arithmetic per element -> same bits.
template <typename T> struct Vec8 { typedef T type __attribute__((ext_vector_type(8))); };
__device__ __forceinline__ void drop_keep8(uint64_t seed, uint32_t stream, long i0, uint32_t thr, bool (&keep)[8]) {
    const uint32_t key = drop_rowkey(seed, stream, (uint64_t)(i0 >> 16));
    const uint32_t p0 = (uint32_t)(i0 & 0xFFFF) >> 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t bits = drop_pair_bits(key, p0 + q);
        keep[2 * q] = drop_keep(bits, 0, thr);
        keep[2 * q + 1] = drop_keep(bits, 1, thr);
    }
}
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) gelu_dropout_fwd8_kernel(const TI* __restrict__ z, TO* __restrict__ u, long n8, uint64_t seed, uint32_t stream, uint32_t thr, float scale) {
    typedef typename Vec8<TI>::type vi;
    typedef typename Vec8<TO>::type vo;
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; v < n8; v += stride) {
        const vi a = reinterpret_cast<const vi*>(z)[v];
        bool keep[8];
        drop_keep8(seed, stream, v * 8, thr, keep);
        vo o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = gelu_erf((float)a[e]);
            o[e] = (TO)(keep[e] ? g * scale : 0.f);
        }
        reinterpret_cast<vo*>(u)[v] = o;
    }
}
template <typename TZ, typename TG, typename TO>
__global__ void __launch_bounds__(256) gelu_dropout_bwd8_kernel(const TZ* __restrict__ z, const TG* __restrict__ du, TO* __restrict__ dz, long n8, uint64_t seed,
                                                                uint32_t stream, uint32_t thr, float scale) {
    typedef typename Vec8<TZ>::type vz;
    typedef typename Vec8<TG>::type vg;
    typedef typename Vec8<TO>::type vo;
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; v < n8; v += stride) {
        const vz a = reinterpret_cast<const vz*>(z)[v];
        const vg g = reinterpret_cast<const vg*>(du)[v];
        bool keep[8];
        drop_keep8(seed, stream, v * 8, thr, keep);
        vo o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)a[e];
            const float d = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
            o[e] = (TO)(keep[e] ? (float)g[e] * d * scale : 0.f);
        }
        reinterpret_cast<vo*>(dz)[v] = o;
    }
}
template <typename TO>
__global__ void __launch_bounds__(256) dropout_cast_bwd8_kernel(const float* __restrict__ dx, long ldx, TO* __restrict__ dy, long ldy, long rows, int cols, uint64_t seed,
                                                                uint32_t stream, uint32_t thr, float scale) {
    typedef typename Vec8<TO>::type vo;
    const long n8 = rows * cols / 8;
    const int c8 = cols / 8;
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; v < n8; v += stride) {
        const long r = v / c8;
        const int c = (int)(v - r * c8) * 8;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(dx + r * ldx + c), a1 = *reinterpret_cast<const f32x4*>(dx + r * ldx + c + 4);
        bool keep[8];
        drop_keep8(seed, stream, v * 8, thr, keep);
        vo o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (TO)(keep[e] ? (e < 4 ? a0[e] : a1[e - 4]) * scale : 0.f);
        *reinterpret_cast<vo*>(dy + r * ldy + c) = o;
    }
}
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// the same, 8 elements per lane: 16-byte accesses, one hash per element pair (drop_keep8) -- the bits of the scalar form
__global__ void __launch_bounds__(256) dropout_add8_kernel(const float* __restrict__ y, long ldy, const float* __restrict__ xin, long ldx, float* __restrict__ xout, long ldo,
                                                           long rows, int cols, uint64_t seed, uint32_t stream, uint32_t thr, float scale) {
    const long n8 = rows * cols / 8;
    const int c8 = cols / 8;
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; v < n8; v += stride) {
        const long r = v / c8;
        const int c = (int)(v - r * c8) * 8;
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(y + r * ldy + c), y1 = *reinterpret_cast<const f32x4*>(y + r * ldy + c + 4);
        f32x4 x0 = *reinterpret_cast<const f32x4*>(xin + r * ldx + c), x1 = *reinterpret_cast<const f32x4*>(xin + r * ldx + c + 4);
        bool keep[8];
        drop_keep8(seed, stream, v * 8, thr, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x0[e] = x0[e] + (keep[e] ? y0[e] * scale : 0.f);
            x1[e] = x1[e] + (keep[4 + e] ? y1[e] * scale : 0.f);
        }
        *reinterpret_cast<f32x4*>(xout + r * ldo + c) = x0;
        *reinterpret_cast<f32x4*>(xout + r * ldo + c + 4) = x1;
    }
}
// rows x cols view with row pitches (the residual stream may be padded): element index = r * cols + c
__global__ void dropout_add_kernel(const float* __restrict__ y, long ldy, const float* __restrict__ xin, long ldx, float* __restrict__ xout, long ldo,
                                   long rows, int cols, uint64_t seed, uint32_t stream, uint32_t thr, float scale) {
    const long n = rows * cols;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = y[r * ldy + c];
        xout[r * ldo + c] = xin[r * ldx + c] + (drop_keep_flat(seed, stream, i, thr) ? v * scale : 0.f);
    }
}
template <typename TO>
__global__ void dropout_cast_bwd_kernel(const float* __restrict__ dx, long ldx, TO* __restrict__ dy, long ldy, long rows, int cols, uint64_t seed,
                                        uint32_t stream, uint32_t thr, float scale) {
    const long n = rows * cols;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        dy[r * ldy + c] = (TO)(drop_keep_flat(seed, stream, i, thr) ? dx[r * ldx + c] * scale : 0.f);
    }
}
__global__ void dropout_mask_kernel(uint8_t* __restrict__ m, long n, uint64_t seed, uint32_t stream, uint32_t thr) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) m[i] = drop_keep_flat(seed, stream, i, thr) ? 1 : 0;
}
// attention probabilities: row = (b*H + h)*T + q, pair = k >> 1 (attention_flash.hip / attention_train.hip use the same rule)
__global__ void attn_dropout_mask_kernel(uint8_t* __restrict__ m, int H, int Tn, long rows, uint64_t seed, uint32_t stream, uint32_t thr) {
    const long row = blockIdx.x;
    if (row >= rows) return;
    const uint32_t key = drop_rowkey(seed, stream, (uint64_t)row);
    for (int k = threadIdx.x; k < Tn; k += blockDim.x) m[row * Tn + k] = drop_keep(drop_pair_bits(key, (uint32_t)k >> 1), k & 1, thr) ? 1 : 0;
}

static inline int grid1d_(long n) { return (int)min((long)8192, (n + 255) / 256); }

// ---- the same four element-wise sites on a FEW rows of the tensor (pitched, logical row = r * row_mul): the class rows of the MIL `vit` head's last block, whose other
// rows nothing reads.  The mask of element (r, c) is that of element (r * row_mul, c) of the full [rows * row_mul][cols] tensor -- the bits the kernels above draw there.
template <typename TZ, typename TO>
__global__ void gelu_dropout_fwd_rows_kernel(const TZ* __restrict__ z, long ldz, TO* __restrict__ u, long ldu, long rows, int cols, long row_mul, uint64_t seed,
                                             uint32_t stream, uint32_t thr, float scale) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float g = gelu_erf((float)z[r * ldz + c]);
        u[r * ldu + c] = (TO)(drop_keep_flat(seed, stream, r * row_mul * cols + c, thr) ? g * scale : 0.f);
    }
}
template <typename TZ, typename TG, typename TO>
__global__ void gelu_dropout_bwd_rows_kernel(const TZ* __restrict__ z, long ldz, const TG* __restrict__ du, long ldu, TO* __restrict__ dz, long lddz, long rows, int cols,
                                             long row_mul, uint64_t seed, uint32_t stream, uint32_t thr, float scale) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float x = (float)z[r * ldz + c];
        const float d = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
        dz[r * lddz + c] = (TO)(drop_keep_flat(seed, stream, r * row_mul * cols + c, thr) ? (float)du[r * ldu + c] * d * scale : 0.f);
    }
}
__global__ void dropout_add_rows_kernel(const float* __restrict__ y, long ldy, const float* __restrict__ xin, long ldx, float* __restrict__ xout, long ldo, long rows,
                                        int cols, long row_mul, uint64_t seed, uint32_t stream, uint32_t thr, float scale) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = y[r * ldy + c];
        xout[r * ldo + c] = xin[r * ldx + c] + (drop_keep_flat(seed, stream, r * row_mul * cols + c, thr) ? v * scale : 0.f);
    }
}
template <typename TO>
__global__ void dropout_cast_bwd_rows_kernel(const float* __restrict__ dx, long ldx, TO* __restrict__ dy, long ldy, long rows, int cols, long row_mul, uint64_t seed,
                                             uint32_t stream, uint32_t thr, float scale) {
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        dy[r * ldy + c] = (TO)(drop_keep_flat(seed, stream, r * row_mul * cols + c, thr) ? dx[r * ldx + c] * scale : 0.f);
    }
}
}  // namespace amds

using namespace amds;

#define DROP_ARGS_OK(p) ((p) >= 0.f && (p) < 1.f)


// p = 0: plain GELU / its derivative / a plain add / a plain cast (threshold 0 keeps everything at scale 1).  The C entries take bf16 tensors (the MIL training
// step's operand type at float32_matmul_precision "medium"); the *_dt forms (csrc-internal, common.h) also take fp16 (the step's operand type at "high").
int amds::gelu_dropout_fwd_rows_dt(const void* z, long ldz, void* u, long ldu, long rows, int cols, long row_mul, int dtype, float p, uint64_t seed, uint32_t stream_id,
                                   void* stream) {
    AMDS_REQUIRE(z && u && rows >= 0 && cols > 0 && row_mul > 0 && p >= 0.f && p < 1.f && (dtype == AMDS_BF16 || dtype == AMDS_F16), "amds_gelu_dropout_fwd_rows: bad arguments");
    if (rows == 0) return AMDS_OK;
    const uint32_t thr = p > 0.f ? drop_thr16(p) : 0;
    const int grid = (int)std::min<long>(4096, (rows * cols + 255) / 256);
    if (dtype == AMDS_BF16)
        hipLaunchKernelGGL((gelu_dropout_fwd_rows_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)z, ldz, (bf16*)u, ldu, rows, cols, row_mul, seed,
                           stream_id, thr, thr ? drop_scale(thr) : 1.0f);
    else
        hipLaunchKernelGGL((gelu_dropout_fwd_rows_kernel<f16, f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)z, ldz, (f16*)u, ldu, rows, cols, row_mul, seed,
                           stream_id, thr, thr ? drop_scale(thr) : 1.0f);
    AMDS_LAUNCH_CHECK("gelu_dropout_fwd_rows_kernel");
    return AMDS_OK;
}
extern "C" int amds_gelu_dropout_fwd_rows(const void* z, long ldz, void* u, long ldu, long rows, int cols, long row_mul, float p, uint64_t seed, uint32_t stream_id,
                                          void* stream) {
    return gelu_dropout_fwd_rows_dt(z, ldz, u, ldu, rows, cols, row_mul, AMDS_BF16, p, seed, stream_id, stream);
}
int amds::gelu_dropout_bwd_rows_dt(const void* z, long ldz, const void* du, long ldu, void* dz, long lddz, long rows, int cols, long row_mul, int dtype, float p,
                                   uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(z && du && dz && rows >= 0 && cols > 0 && row_mul > 0 && p >= 0.f && p < 1.f && (dtype == AMDS_BF16 || dtype == AMDS_F16), "amds_gelu_dropout_bwd_rows: bad arguments");
    if (rows == 0) return AMDS_OK;
    const uint32_t thr = p > 0.f ? drop_thr16(p) : 0;
    const int grid = (int)std::min<long>(4096, (rows * cols + 255) / 256);
    if (dtype == AMDS_BF16)
        hipLaunchKernelGGL((gelu_dropout_bwd_rows_kernel<bf16, bf16, bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)z, ldz, (const bf16*)du, ldu, (bf16*)dz,
                           lddz, rows, cols, row_mul, seed, stream_id, thr, thr ? drop_scale(thr) : 1.0f);
    else
        hipLaunchKernelGGL((gelu_dropout_bwd_rows_kernel<f16, f16, f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f16*)z, ldz, (const f16*)du, ldu, (f16*)dz,
                           lddz, rows, cols, row_mul, seed, stream_id, thr, thr ? drop_scale(thr) : 1.0f);
    AMDS_LAUNCH_CHECK("gelu_dropout_bwd_rows_kernel");
    return AMDS_OK;
}
extern "C" int amds_gelu_dropout_bwd_rows(const void* z, long ldz, const void* du, long ldu, void* dz, long lddz, long rows, int cols, long row_mul, float p,
                                          uint64_t seed, uint32_t stream_id, void* stream) {
    return gelu_dropout_bwd_rows_dt(z, ldz, du, ldu, dz, lddz, rows, cols, row_mul, AMDS_BF16, p, seed, stream_id, stream);
}
extern "C" int amds_dropout_add_rows(const float* y, long ldy, const float* x_in, long ldx, float* x_out, long ldo, long rows, int cols, long row_mul, float p,
                                     uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(y && x_in && x_out && rows >= 0 && cols > 0 && row_mul > 0 && p >= 0.f && p < 1.f, "amds_dropout_add_rows: bad arguments");
    if (rows == 0) return AMDS_OK;
    const uint32_t thr = p > 0.f ? drop_thr16(p) : 0;
    const int grid = (int)std::min<long>(4096, (rows * cols + 255) / 256);
    hipLaunchKernelGGL(dropout_add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ldy, x_in, ldx, x_out, ldo, rows, cols, row_mul, seed, stream_id, thr,
                       thr ? drop_scale(thr) : 1.0f);
    AMDS_LAUNCH_CHECK("dropout_add_rows_kernel");
    return AMDS_OK;
}
int amds::dropout_cast_bwd_rows_dt(const float* dx, long ldx, void* dy, long ldy, long rows, int cols, long row_mul, int dtype, float p, uint64_t seed, uint32_t stream_id,
                                   void* stream) {
    AMDS_REQUIRE(dx && dy && rows >= 0 && cols > 0 && row_mul > 0 && p >= 0.f && p < 1.f && (dtype == AMDS_BF16 || dtype == AMDS_F16 || dtype == AMDS_F32),
                 "amds_dropout_cast_bwd_rows: bad arguments");
    if (rows == 0) return AMDS_OK;
    const uint32_t thr = p > 0.f ? drop_thr16(p) : 0;
    const int grid = (int)std::min<long>(4096, (rows * cols + 255) / 256);
    if (dtype == AMDS_BF16)
        hipLaunchKernelGGL((dropout_cast_bwd_rows_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dx, ldx, (bf16*)dy, ldy, rows, cols, row_mul, seed, stream_id, thr,
                           thr ? drop_scale(thr) : 1.0f);
    else if (dtype == AMDS_F32)
        hipLaunchKernelGGL((dropout_cast_bwd_rows_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dx, ldx, (float*)dy, ldy, rows, cols, row_mul, seed, stream_id, thr,
                           thr ? drop_scale(thr) : 1.0f);
    else
        hipLaunchKernelGGL((dropout_cast_bwd_rows_kernel<f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, dx, ldx, (f16*)dy, ldy, rows, cols, row_mul, seed, stream_id, thr,
                           thr ? drop_scale(thr) : 1.0f);
    AMDS_LAUNCH_CHECK("dropout_cast_bwd_rows_kernel");
    return AMDS_OK;
}
extern "C" int amds_dropout_cast_bwd_rows(const float* dx, long ldx, void* dy, long ldy, long rows, int cols, long row_mul, float p, uint64_t seed, uint32_t stream_id,
                                          void* stream) {
    return dropout_cast_bwd_rows_dt(dx, ldx, dy, ldy, rows, cols, row_mul, AMDS_BF16, p, seed, stream_id, stream);
}
extern "C" float amds_dropout_keep_scale(float p) { return drop_scale(drop_thr16(p)); }

extern "C" int amds_gelu_dropout_fwd(const void* z, void* u, long n, int in_dtype, int out_dtype, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(z && u && n >= 0 && DROP_ARGS_OK(p), "amds_gelu_dropout_fwd: bad arguments");
    if (n == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_thr16(p);
    const float sc = drop_scale(thr);
    const bool v8 = n % 8 == 0 && al16(z) && al16(u);
    if (in_dtype == AMDS_BF16 && out_dtype == AMDS_BF16 && v8) hipLaunchKernelGGL((gelu_dropout_fwd8_kernel<bf16, bf16>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const bf16*)z, (bf16*)u, n / 8, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_BF16 && out_dtype == AMDS_F32 && v8) hipLaunchKernelGGL((gelu_dropout_fwd8_kernel<bf16, float>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const bf16*)z, (float*)u, n / 8, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_BF16 && out_dtype == AMDS_BF16) hipLaunchKernelGGL((gelu_dropout_fwd_kernel<bf16, bf16>), dim3(grid1d_(n)), dim3(256), 0, st, (const bf16*)z, (bf16*)u, n, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_BF16 && out_dtype == AMDS_F32) hipLaunchKernelGGL((gelu_dropout_fwd_kernel<bf16, float>), dim3(grid1d_(n)), dim3(256), 0, st, (const bf16*)z, (float*)u, n, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F16 && v8) hipLaunchKernelGGL((gelu_dropout_fwd8_kernel<f16, f16>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const f16*)z, (f16*)u, n / 8, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F32 && v8) hipLaunchKernelGGL((gelu_dropout_fwd8_kernel<f16, float>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const f16*)z, (float*)u, n / 8, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F16) hipLaunchKernelGGL((gelu_dropout_fwd_kernel<f16, f16>), dim3(grid1d_(n)), dim3(256), 0, st, (const f16*)z, (f16*)u, n, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F32) hipLaunchKernelGGL((gelu_dropout_fwd_kernel<f16, float>), dim3(grid1d_(n)), dim3(256), 0, st, (const f16*)z, (float*)u, n, seed, stream_id, thr, sc);
    else if (in_dtype == AMDS_F32 && out_dtype == AMDS_F32) hipLaunchKernelGGL((gelu_dropout_fwd_kernel<float, float>), dim3(grid1d_(n)), dim3(256), 0, st, (const float*)z, (float*)u, n, seed, stream_id, thr, sc);
    else { set_error("amds_gelu_dropout_fwd: unsupported dtype pair"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("gelu_dropout_fwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_gelu_dropout_bwd(const void* z, const void* du, void* dz, long n, int z_dtype, int du_dtype, int dz_dtype, float p, uint64_t seed,
                                     uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(z && du && dz && n >= 0 && DROP_ARGS_OK(p), "amds_gelu_dropout_bwd: bad arguments");
    if (n == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_thr16(p);
    const float sc = drop_scale(thr);
    const bool v8 = n % 8 == 0 && al16(z) && al16(du) && al16(dz);
    if (z_dtype == AMDS_BF16 && du_dtype == AMDS_BF16 && dz_dtype == AMDS_BF16 && v8)
        hipLaunchKernelGGL((gelu_dropout_bwd8_kernel<bf16, bf16, bf16>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const bf16*)z, (const bf16*)du, (bf16*)dz, n / 8, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_BF16 && du_dtype == AMDS_F32 && dz_dtype == AMDS_BF16 && v8)
        hipLaunchKernelGGL((gelu_dropout_bwd8_kernel<bf16, float, bf16>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const bf16*)z, (const float*)du, (bf16*)dz, n / 8, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_BF16 && du_dtype == AMDS_BF16 && dz_dtype == AMDS_BF16)
        hipLaunchKernelGGL((gelu_dropout_bwd_kernel<bf16, bf16, bf16>), dim3(grid1d_(n)), dim3(256), 0, st, (const bf16*)z, (const bf16*)du, (bf16*)dz, n, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_BF16 && du_dtype == AMDS_F32 && dz_dtype == AMDS_BF16)
        hipLaunchKernelGGL((gelu_dropout_bwd_kernel<bf16, float, bf16>), dim3(grid1d_(n)), dim3(256), 0, st, (const bf16*)z, (const float*)du, (bf16*)dz, n, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_F16 && du_dtype == AMDS_F16 && dz_dtype == AMDS_F16 && v8)
        hipLaunchKernelGGL((gelu_dropout_bwd8_kernel<f16, f16, f16>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const f16*)z, (const f16*)du, (f16*)dz, n / 8, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_F16 && du_dtype == AMDS_F32 && dz_dtype == AMDS_F16 && v8)
        hipLaunchKernelGGL((gelu_dropout_bwd8_kernel<f16, float, f16>), dim3(grid1d_(n / 8)), dim3(256), 0, st, (const f16*)z, (const float*)du, (f16*)dz, n / 8, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_F16 && du_dtype == AMDS_F16 && dz_dtype == AMDS_F16)
        hipLaunchKernelGGL((gelu_dropout_bwd_kernel<f16, f16, f16>), dim3(grid1d_(n)), dim3(256), 0, st, (const f16*)z, (const f16*)du, (f16*)dz, n, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_F16 && du_dtype == AMDS_F32 && dz_dtype == AMDS_F16)
        hipLaunchKernelGGL((gelu_dropout_bwd_kernel<f16, float, f16>), dim3(grid1d_(n)), dim3(256), 0, st, (const f16*)z, (const float*)du, (f16*)dz, n, seed, stream_id, thr, sc);
    else if (z_dtype == AMDS_F32 && du_dtype == AMDS_F32 && dz_dtype == AMDS_F32)
        hipLaunchKernelGGL((gelu_dropout_bwd_kernel<float, float, float>), dim3(grid1d_(n)), dim3(256), 0, st, (const float*)z, (const float*)du, (float*)dz, n, seed, stream_id, thr, sc);
    else { set_error("amds_gelu_dropout_bwd: unsupported dtype combination"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("gelu_dropout_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_dropout_add(const float* y, long ldy, const float* x_in, long ldx, float* x_out, long ldo, long rows, int cols, float p,
                                uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(y && x_in && x_out && rows >= 0 && cols > 0 && ldy >= cols && ldx >= cols && ldo >= cols && DROP_ARGS_OK(p), "amds_dropout_add: bad arguments");
    if (rows == 0) return AMDS_OK;
    const uint32_t thr = drop_thr16(p);
    if (cols % 8 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && al16(y) && al16(x_in) && al16(x_out))
        hipLaunchKernelGGL(dropout_add8_kernel, dim3(grid1d_(rows * cols / 8)), dim3(256), 0, (hipStream_t)stream, y, ldy, x_in, ldx, x_out, ldo, rows, cols, seed,
                           stream_id, thr, drop_scale(thr));
    else
        hipLaunchKernelGGL(dropout_add_kernel, dim3(grid1d_(rows * cols)), dim3(256), 0, (hipStream_t)stream, y, ldy, x_in, ldx, x_out, ldo, rows, cols, seed,
                           stream_id, thr, drop_scale(thr));
    AMDS_LAUNCH_CHECK("dropout_add_kernel");
    return AMDS_OK;
}

extern "C" int amds_dropout_cast_bwd(const float* dx, long ldx, void* dy, long ldy, long rows, int cols, int out_dtype, float p, uint64_t seed,
                                     uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(dx && dy && rows >= 0 && cols > 0 && ldx >= cols && ldy >= cols && DROP_ARGS_OK(p), "amds_dropout_cast_bwd: bad arguments");
    if (rows == 0) return AMDS_OK;
    const uint32_t thr = drop_thr16(p);
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == AMDS_BF16 && cols % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0 && al16(dx) && al16(dy))
        hipLaunchKernelGGL((dropout_cast_bwd8_kernel<bf16>), dim3(grid1d_(rows * cols / 8)), dim3(256), 0, st, dx, ldx, (bf16*)dy, ldy, rows, cols, seed, stream_id, thr, drop_scale(thr));
    else if (out_dtype == AMDS_BF16) hipLaunchKernelGGL((dropout_cast_bwd_kernel<bf16>), dim3(grid1d_(rows * cols)), dim3(256), 0, st, dx, ldx, (bf16*)dy, ldy, rows, cols, seed, stream_id, thr, drop_scale(thr));
    else if (out_dtype == AMDS_F16 && cols % 8 == 0 && ldx % 4 == 0 && ldy % 8 == 0 && al16(dx) && al16(dy))
        hipLaunchKernelGGL((dropout_cast_bwd8_kernel<f16>), dim3(grid1d_(rows * cols / 8)), dim3(256), 0, st, dx, ldx, (f16*)dy, ldy, rows, cols, seed, stream_id, thr, drop_scale(thr));
    else if (out_dtype == AMDS_F16) hipLaunchKernelGGL((dropout_cast_bwd_kernel<f16>), dim3(grid1d_(rows * cols)), dim3(256), 0, st, dx, ldx, (f16*)dy, ldy, rows, cols, seed, stream_id, thr, drop_scale(thr));
    else if (out_dtype == AMDS_F32) hipLaunchKernelGGL((dropout_cast_bwd_kernel<float>), dim3(grid1d_(rows * cols)), dim3(256), 0, st, dx, ldx, (float*)dy, ldy, rows, cols, seed, stream_id, thr, drop_scale(thr));
    else { set_error("amds_dropout_cast_bwd: bad dtype"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("dropout_cast_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_dropout_mask(uint8_t* mask, long n, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(mask && n >= 0 && DROP_ARGS_OK(p), "amds_dropout_mask: bad arguments");
    if (n == 0) return AMDS_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid1d_(n)), dim3(256), 0, (hipStream_t)stream, mask, n, seed, stream_id, drop_thr16(p));
    AMDS_LAUNCH_CHECK("dropout_mask_kernel");
    return AMDS_OK;
}

extern "C" int amds_attention_dropout_mask(uint8_t* mask, int B, int H, int T, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(mask && B > 0 && H > 0 && T > 0 && DROP_ARGS_OK(p), "amds_attention_dropout_mask: bad arguments");
    const long rows = (long)B * H * T;
    AMDS_REQUIRE(rows < (1L << 31), "amds_attention_dropout_mask: too many rows");
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, mask, H, T, rows, seed, stream_id, drop_thr16(p));
    AMDS_LAUNCH_CHECK("attn_dropout_mask_kernel");
    return AMDS_OK;
}
