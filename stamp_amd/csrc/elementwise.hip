// elementwise.hip -- HBM-bound kernels of the tile-encoder path: LayerNorm, weight casts, the u8 tile
// transform and the u8 -> patch-matrix (im2col) staging.  One wave per row / 16-byte accesses per lane;
// none of these has inter-block reuse, so no XCD remap (guide T1: 0 % on LayerNorm).
#include "common.h"

namespace amds {

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave64 per row, row kept in registers (cols <= 64*4*MAXV), two-pass statistics in
// fp32 (mean, then centred sum of squares) -- matches torch.nn.LayerNorm's biased variance.
// ---------------------------------------------------------------------------------------------
template <typename TO, int MAXV>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, long xs,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TO* __restrict__ y,
                                                        long ys, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * xs;
    const int nv = cols >> 2;  // float4 count
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
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    TO* yr = y + (long)row * ys;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if constexpr (sizeof(TO) == 4) {
                f32x4 w = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(yr + c * 4) = w;
            } else {
                typedef TO vec4 __attribute__((ext_vector_type(4)));
                vec4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (TO)o[e];
                *reinterpret_cast<vec4*>(yr + c * 4) = w;
            }
        }
    }
}

// Narrow rows (cols <= 512): a full wave per row would idle most lanes (96 columns = 24 float4 of 64 lanes), so a row
// is owned by a group of G = 8/16/32 lanes (64/G rows per wave); statistics reduce inside the group.
template <typename TO, int G, int MAXV>
__global__ void __launch_bounds__(256) layernorm_narrow_kernel(const float* __restrict__ x, long xs,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, TO* __restrict__ y,
                                                               long ys, int rows, int cols, float eps) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int row = blockIdx.x * RPB + threadIdx.x / G;
    const bool live = row < rows;
    const float* xr = x + (long)(live ? row : rows - 1) * xs;
    const int nv = cols >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * G + sub;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i * G + sub < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)cols + eps);
    if (!live) return;
    TO* yr = y + (long)row * ys;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * G + sub;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if constexpr (sizeof(TO) == 4) {
                f32x4 w = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(yr + c * 4) = w;
            } else {
                typedef TO vec4 __attribute__((ext_vector_type(4)));
                vec4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (TO)o[e];
                *reinterpret_cast<vec4*>(yr + c * 4) = w;
            }
        }
    }
}

template <typename TO>
static int launch_ln(const float* x, long xs, const float* g, const float* b, void* y, long ys, int rows,
                     int cols, float eps, hipStream_t st) {
    const int nv = cols / 4;
    const int grid = cdiv(rows, 4);
    if (nv <= 32)
        hipLaunchKernelGGL((layernorm_narrow_kernel<TO, 8, 4>), dim3(cdiv(rows, 32)), dim3(256), 0, st, x, xs, g, b, (TO*)y, ys, rows, cols, eps);
    else if (nv <= 64)
        hipLaunchKernelGGL((layernorm_narrow_kernel<TO, 16, 4>), dim3(cdiv(rows, 16)), dim3(256), 0, st, x, xs, g, b, (TO*)y, ys, rows, cols, eps);
    else if (nv <= 128)
        hipLaunchKernelGGL((layernorm_narrow_kernel<TO, 32, 4>), dim3(cdiv(rows, 8)), dim3(256), 0, st, x, xs, g, b, (TO*)y, ys, rows, cols, eps);
    else if (nv <= 64 * 4)
        hipLaunchKernelGGL((layernorm_kernel<TO, 4>), dim3(grid), dim3(256), 0, st, x, xs, g, b, (TO*)y, ys, rows, cols, eps);
    else if (nv <= 64 * 8)
        hipLaunchKernelGGL((layernorm_kernel<TO, 8>), dim3(grid), dim3(256), 0, st, x, xs, g, b, (TO*)y, ys, rows, cols, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<TO, 32>), dim3(grid), dim3(256), 0, st, x, xs, g, b, (TO*)y, ys, rows, cols, eps);
    AMDS_LAUNCH_CHECK("layernorm_kernel");
    return AMDS_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs (amds_gemm_lnfold): the two small kernels around them
// ---------------------------------------------------------------------------------------------
// partial (sum, sum of squares) per 128-column slab [M][NP][2] -> (rstd, -mean * rstd) per row, slabs added in index order
// diag (optional, int[2]): [0] += rows whose sum of squares reaches sq_limit (the 16-bit copy of such a row MAY hold an element beyond the act
// dtype's range: |x| <= sqrt(sum x^2), so rows below the limit provably cannot), [1] += rows with |mean| > 8 sigma (the folded form rounds
// x, not x - mean: its error grows with |mean| / sigma).  Rare events: the atomics never run on healthy rows.
__global__ void __launch_bounds__(256) ln_rowstat_kernel(const float* __restrict__ rowpart, int M, int NP, float inv_d, float eps,
                                                         float* __restrict__ rowstat, int* __restrict__ diag, float sq_limit) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= M) return;
    const f32x2* p = reinterpret_cast<const f32x2*>(rowpart) + (long)r * NP;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < NP; ++i) {
        const f32x2 v = p[i];
        s1 += v[0];
        s2 += v[1];
    }
    const float mean = s1 * inv_d, var = fmaxf(s2 * inv_d - mean * mean, 0.f), rstd = rsqrtf(var + eps);
    reinterpret_cast<f32x2*>(rowstat)[r] = f32x2{rstd, -mean * rstd};
    if (diag) {
        if (!(s2 < sq_limit)) atomicAdd(diag, 1);              // also catches NaN / inf sums
        if (mean * mean > 64.f * var) atomicAdd(diag + 1, 1);
    }
}

// the first LayerNorm of a stack (its input does not come out of a RESIDUAL GEMM): x fp32 -> 16-bit copy + (rstd, -mean * rstd)
template <typename TO, int MAXV>
__global__ void __launch_bounds__(256) ln_stats_cast_kernel(const float* __restrict__ x, long xs, TO* __restrict__ y, long ys, int rows,
                                                            int cols, float eps, float* __restrict__ rowstat, TO* __restrict__ ylo = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * xs;
    TO* yr = y + (long)row * ys;
    const int nv = cols >> 2;
    typedef TO vec4 __attribute__((ext_vector_type(4)));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c * 4);
            s1 += (v[0] + v[1]) + (v[2] + v[3]);
            s2 += fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]);
            vec4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (TO)v[e];
            *reinterpret_cast<vec4*>(yr + c * 4) = w;
            if (ylo != nullptr) {           // second plane of the (hi | lo) residual stream: what the 16-bit rounding of x left over
                vec4 wl;
#pragma unroll
                for (int e = 0; e < 4; ++e) wl[e] = (TO)(v[e] - (float)w[e]);
                *reinterpret_cast<vec4*>(ylo + (long)row * ys + c * 4) = wl;
            }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float mean = s1 / (float)cols, var = fmaxf(s2 / (float)cols - mean * mean, 0.f), rstd = rsqrtf(var + eps);
    if (lane == 0) reinterpret_cast<f32x2*>(rowstat)[row] = f32x2{rstd, -mean * rstd};
}

// (hi | lo) planes -> fp32 rows: out[i][:] = hi[i * rstride][:] + lo[i * rstride][:]   (rstride = 1: every row; = T: the class rows)
__global__ void planes_to_f32_kernel(const f16* __restrict__ hi, const f16* __restrict__ lo, long ld, long rstride, float* __restrict__ out, long ldo,
                                     long rows, int cols) {
    const int nv = cols >> 2;
    const long total = rows * nv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / nv;
        const int c = (int)(i - r * nv) * 4;
        const long src = r * rstride * ld + c;
        const f16x4 h = *reinterpret_cast<const f16x4*>(hi + src), l = *reinterpret_cast<const f16x4*>(lo + src);
        *reinterpret_cast<f32x4*>(out + r * ldo + c) = __builtin_convertvector(h, f32x4) + __builtin_convertvector(l, f32x4);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 -> act dtype cast with zero padding of the trailing columns (weight packing, one time)
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ void cast_pad_kernel(const float* __restrict__ src, int ld_src, TO* __restrict__ dst, int ld_dst,
                                long total, int cols) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / ld_dst;
        const int c = (int)(i - r * ld_dst);
        dst[i] = c < cols ? (TO)src[r * ld_src + c] : (TO)0.f;
    }
}

// rows of a SwiGLUPacked fc1: [gate 0..H) | value 0..H)]  ->  blocks of 32: [gate 32j..][value 32j..]
__global__ void pack_swiglu_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int cols) {
    const long total = 2L * H * cols;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const int blk = (int)(r >> 6), within = (int)(r & 63);
        const int is_val = within >> 5, u = blk * 32 + (within & 31);
        dst[i] = src[((long)is_val * H + u) * cols + c];
    }
}

// ---------------------------------------------------------------------------------------------
// Tile transform: u8 HWC -> fp32 CHW, (x/255 - mean)/std  (ToTensor + Normalize)
// ---------------------------------------------------------------------------------------------
struct Norm3 { float mean[3], inv_std[3]; };
__global__ void __launch_bounds__(256) tile_normalize_kernel(const uint8_t* __restrict__ hwc, float* __restrict__ chw,
                                                             long npix_total, int hw, Norm3 nm) {
    // each thread: 4 consecutive pixels (12 bytes in) -> 3 float4 stores (one per channel plane)
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nq = npix_total >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; q < nq; q += stride) {
        const long p = q << 2;
        const long b = p / hw;
        const int off = (int)(p - b * hw);
        const uint32_t* s = reinterpret_cast<const uint32_t*>(hwc + p * 3);
        const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
        uint8_t px[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { px[k] = (w0 >> (8 * k)) & 255; px[4 + k] = (w1 >> (8 * k)) & 255; px[8 + k] = (w2 >> (8 * k)) & 255; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ((float)px[k * 3 + c] / 255.0f - nm.mean[c]) * nm.inv_std[c];
            *reinterpret_cast<f32x4*>(chw + (b * 3 + c) * hw + off) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// u8 HWC tile -> patch matrix rows (im2col for Conv2d(3, D, p, stride p)):
//   out[(b*g*g + py*g + px)][c*p*p + i*p + j] = (act) u8[b][py*p+i][px*p+j][c],  zero for k >= 3*p*p.
// One block per (tile, patch-row): the p image rows (p*img*3 contiguous bytes) are staged in LDS with
// 16-byte coalesced loads, then each thread assembles 8 consecutive k (one 16-byte store).
// ---------------------------------------------------------------------------------------------
// lo_scale != 0: rows are 2*kp wide and columns kp.. hold the same values times lo_scale (a power of two, exact) -- the A operand of a
// patch-embedding GEMM whose weight is split into [hi | lo / lo_scale] (amds_tile_im2col_u8_ex).
constexpr int IM2COL_KMAX = 4096;          // kp = roundup(3 p^2, 64): 640 (p = 14), 768 (p = 16), 3072 (p = 32: CLIP ViT-B/32); 16 KB of LDS
template <typename TO>
__global__ void __launch_bounds__(256) im2col_u8_kernel(const uint8_t* __restrict__ tiles, TO* __restrict__ out,
                                                        int img, int p, int kp, float lo_scale) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srow[];
    const int g = img / p;
    const int b = blockIdx.x / g, py = blockIdx.x - b * g;
    const int nbytes = p * img * 3;  // multiple of 16 for img=224
    const uint8_t* src = tiles + ((long)b * img + (long)py * p) * img * 3;
    for (int i = threadIdx.x * 16; i < nbytes; i += 256 * 16)
        *reinterpret_cast<u32x4*>(srow + i) = *reinterpret_cast<const u32x4*>(src + i);
    // column k = (c, i, j) of the patch matrix -> byte offset of that pixel inside the staged rows, relative to the patch's first pixel: computed
    // once per block (two runtime integer divisions per entry) instead of once per output element (they were most of this kernel's 412 us)
    __shared__ int koff[IM2COL_KMAX];
    const int pp = p * p, k_real = 3 * pp;
    for (int k = threadIdx.x; k < kp; k += 256) {
        const int c = k / pp, r = k - c * pp, i = r / p, j = r - i * p;
        koff[k] = k < k_real ? (i * img + j) * 3 + c : -1;
    }
    __syncthreads();
    const int ld = lo_scale != 0.f ? 2 * kp : kp;
    const int chunks = ld >> 3, half = kp >> 3;
    TO* orow = out + ((long)b * g * g + (long)py * g) * ld;
    typedef TO vec8 __attribute__((ext_vector_type(8)));
    for (int w = threadIdx.x; w < g * chunks; w += 256) {
        const int px = w / chunks, ch = w - px * chunks;
        const int ch0 = ch >= half ? ch - half : ch;
        const float sc = ch >= half ? lo_scale : 1.0f;
        const uint8_t* pbase = srow + px * p * 3;
        vec8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int off = koff[ch0 * 8 + e];
            o[e] = (TO)(off >= 0 ? (float)pbase[off] * sc : 0.f);
        }
        *reinterpret_cast<vec8*>(orow + (long)px * ld + ch * 8) = o;
    }
}

// x[b*T + t][:] = prefix[t][:] for t < P  (cls / register tokens, with their pos-embed already added)
__global__ void prefix_init_kernel(const float* __restrict__ prefix, float* __restrict__ x, int B, int T, int P, int dim) {
    const long total = (long)B * P * (dim >> 2);
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int dv = dim >> 2;
    for (; i < total; i += stride) {
        const long bt = i / dv;
        const int c = (int)(i - bt * dv);
        const long b = bt / P;
        const int t = (int)(bt - b * P);
        *reinterpret_cast<f32x4*>(x + ((b * T + t) * (long)dim) + c * 4) =
            *reinterpret_cast<const f32x4*>(prefix + (long)t * dim + c * 4);
    }
}

int prefix_init(const float* prefix, float* x, int B, int T, int P, int dim, hipStream_t st) {
    const long total = (long)B * P * (dim / 4);
    const int grid = (int)min((long)2048, (total + 255) / 256);
    hipLaunchKernelGGL(prefix_init_kernel, dim3(grid), dim3(256), 0, st, prefix, x, B, T, P, dim);
    AMDS_LAUNCH_CHECK("prefix_init_kernel");
    return AMDS_OK;
}

// CLIP's activation (HF `quick_gelu`: x * sigmoid(1.702 x)), in place on 16-bit rows [rows][ld], `cols` columns (the PLIP extractor's MLP:
// fc1 writes the pre-activation with the plain bias epilogue, this turns it into fc2's A operand)
template <typename T>
__global__ void __launch_bounds__(256) quick_gelu_inplace_kernel(T* __restrict__ u, long ld, long rows, int cols) {
    typedef T vec8 __attribute__((ext_vector_type(8)));
    const int c8 = cols >> 3;
    const long total = rows * c8;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long r = i / c8;
        const int c = (int)(i - r * c8);
        vec8* p = reinterpret_cast<vec8*>(u + r * ld) + c;
        vec8 v = *p;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)v[e];
            v[e] = (T)(x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)));      // v_rcp_f32 (1 ulp): the result is rounded to 16 bits
        }
        *p = v;
    }
}

}  // namespace amds

using namespace amds;

extern "C" int amds_layernorm(const float* x, long x_row_stride, const float* gamma, const float* beta, void* y,
                              long y_row_stride, int rows, int cols, float eps, int out_dtype, void* stream) {
    AMDS_REQUIRE(x && gamma && beta && y, "amds_layernorm: null pointer");
    AMDS_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 8192, "amds_layernorm: cols=%d must be a multiple of 4 and <= 8192", cols);
    AMDS_REQUIRE(x_row_stride % 4 == 0 && y_row_stride % 4 == 0, "amds_layernorm: row strides must be multiples of 4");
    if (rows == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)rows * cols * (4 + (out_dtype == AMDS_F32 ? 4 : 2)), st);
    switch (out_dtype) {
        case AMDS_F16: return launch_ln<f16>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, cols, eps, st);
        case AMDS_BF16: return launch_ln<bf16>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, cols, eps, st);
        case AMDS_F32: return launch_ln<float>(x, x_row_stride, gamma, beta, y, y_row_stride, rows, cols, eps, st);
    }
    set_error("amds_layernorm: bad out_dtype %d", out_dtype);
    return AMDS_ERR_INVALID;
}

extern "C" int amds_cast_pad(const float* src, int ld_src, void* dst, int ld_dst, int rows, int cols, int dtype,
                             void* stream) {
    AMDS_REQUIRE(src && dst, "amds_cast_pad: null pointer");
    AMDS_REQUIRE(rows >= 0 && cols >= 0 && ld_dst >= cols && ld_src >= cols, "amds_cast_pad: bad shape");
    const long total = (long)rows * ld_dst;
    if (total == 0) return AMDS_OK;
    const int grid = (int)min((long)4096, (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AMDS_F16)
        hipLaunchKernelGGL((cast_pad_kernel<f16>), dim3(grid), dim3(256), 0, st, src, ld_src, (f16*)dst, ld_dst, total, cols);
    else if (dtype == AMDS_BF16)
        hipLaunchKernelGGL((cast_pad_kernel<bf16>), dim3(grid), dim3(256), 0, st, src, ld_src, (bf16*)dst, ld_dst, total, cols);
    else if (dtype == AMDS_F32)
        hipLaunchKernelGGL((cast_pad_kernel<float>), dim3(grid), dim3(256), 0, st, src, ld_src, (float*)dst, ld_dst, total, cols);
    else { set_error("amds_cast_pad: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("cast_pad_kernel");
    return AMDS_OK;
}

extern "C" int amds_pack_swiglu_rows(const float* src, float* dst, int H, int cols, void* stream) {
    AMDS_REQUIRE(src && dst && src != dst, "amds_pack_swiglu_rows: null/aliased pointer");
    AMDS_REQUIRE(H > 0 && H % 32 == 0 && cols > 0, "amds_pack_swiglu_rows: H=%d must be a positive multiple of 32", H);
    const long total = 2L * H * cols;
    const int grid = (int)min((long)4096, (total + 255) / 256);
    hipLaunchKernelGGL(pack_swiglu_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, H, cols);
    AMDS_LAUNCH_CHECK("pack_swiglu_rows_kernel");
    return AMDS_OK;
}

extern "C" int amds_tile_normalize_u8(const uint8_t* hwc, float* chw, int B, int H, int W, const float mean_host[3],
                                      const float std_host[3], void* stream) {
    AMDS_REQUIRE(hwc && chw && mean_host && std_host, "amds_tile_normalize_u8: null pointer");
    AMDS_REQUIRE(B >= 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "amds_tile_normalize_u8: H*W must be a multiple of 4");
    if (B == 0) return AMDS_OK;
    Norm3 nm;
    for (int c = 0; c < 3; ++c) {
        AMDS_REQUIRE(std_host[c] != 0.f, "amds_tile_normalize_u8: std[%d] == 0", c);
        nm.mean[c] = mean_host[c];
        nm.inv_std[c] = 1.0f / std_host[c];
    }
    const long npix = (long)B * H * W;
    const int grid = (int)min((long)4096, (npix / 4 + 255) / 256);
    hipLaunchKernelGGL(tile_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, hwc, chw, npix, H * W, nm);
    AMDS_LAUNCH_CHECK("tile_normalize_kernel");
    return AMDS_OK;
}

extern "C" int amds_tile_im2col_u8(const uint8_t* tiles, void* out, int B, int img, int patch, int kp, int dtype,
                                   void* stream) {
    return amds_tile_im2col_u8_ex(tiles, out, B, img, patch, kp, dtype, 0, stream);
}

extern "C" int amds_tile_im2col_u8_ex(const uint8_t* tiles, void* out, int B, int img, int patch, int kp, int dtype, int lo_shift,
                                      void* stream) {
    AMDS_REQUIRE(tiles && out, "amds_tile_im2col_u8: null pointer");
    AMDS_REQUIRE(lo_shift >= 0 && lo_shift <= 14, "amds_tile_im2col_u8: lo_shift=%d out of range", lo_shift);
    const float lo_scale = lo_shift ? ldexpf(1.0f, -lo_shift) : 0.f;
    AMDS_REQUIRE(img > 0 && patch > 0 && img % patch == 0, "amds_tile_im2col_u8: img=%d not divisible by patch=%d", img, patch);
    AMDS_REQUIRE(kp % 8 == 0 && kp >= 3 * patch * patch && kp <= IM2COL_KMAX, "amds_tile_im2col_u8: kp=%d too small / not a multiple of 8 / above 4096", kp);
    AMDS_REQUIRE((patch * img * 3) % 16 == 0 && ((long)img * img * 3) % 16 == 0, "amds_tile_im2col_u8: row block not 16-byte aligned");
    if (B == 0) return AMDS_OK;
    const int g = img / patch;
    const size_t lds = (size_t)patch * img * 3;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_OTHER, (double)B * ((double)img * img * 3 + (double)g * g * kp * (lo_shift ? 4 : 2)), st);
    if (dtype == AMDS_F16)
        hipLaunchKernelGGL((im2col_u8_kernel<f16>), dim3(B * g), dim3(256), lds, st, tiles, (f16*)out, img, patch, kp, lo_scale);
    else if (dtype == AMDS_BF16)
        hipLaunchKernelGGL((im2col_u8_kernel<bf16>), dim3(B * g), dim3(256), lds, st, tiles, (bf16*)out, img, patch, kp, lo_scale);
    else { set_error("amds_tile_im2col_u8: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("im2col_u8_kernel");
    return AMDS_OK;
}

extern "C" int amds_ln_rowstat(const float* rowpart, int M, int NP, int D, float eps, float* rowstat, void* stream) {
    AMDS_REQUIRE(rowpart && rowstat && M >= 0 && NP > 0 && D > 0, "amds_ln_rowstat: bad arguments");
    if (M == 0) return AMDS_OK;
    ProfScope prof(PROF_LN, (double)M * (NP * 8.0 + 8.0), (hipStream_t)stream);
    hipLaunchKernelGGL(ln_rowstat_kernel, dim3(cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, rowpart, M, NP, 1.0f / (float)D, eps, rowstat,
                       (int*)nullptr, 0.f);
    AMDS_LAUNCH_CHECK("ln_rowstat_kernel");
    return AMDS_OK;
}

extern "C" int amds_ln_rowstat_diag(const float* rowpart, int M, int NP, int D, float eps, float* rowstat, int* diag, int dtype, void* stream) {
    AMDS_REQUIRE(rowpart && rowstat && M >= 0 && NP > 0 && D > 0, "amds_ln_rowstat_diag: bad arguments");
    if (M == 0) return AMDS_OK;
    ProfScope prof(PROF_LN, (double)M * (NP * 8.0 + 8.0), (hipStream_t)stream);
    const float lim = dtype == AMDS_F16 ? 65504.f * 65504.f : 3.0e38f;
    hipLaunchKernelGGL(ln_rowstat_kernel, dim3(cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, rowpart, M, NP, 1.0f / (float)D, eps, rowstat, diag, lim);
    AMDS_LAUNCH_CHECK("ln_rowstat_kernel");
    return AMDS_OK;
}

extern "C" int amds_ln_stats_cast(const float* x, long ldx, int M, int D, float eps, void* xh, long ldxh, float* rowstat, int dtype,
                                  void* stream) {
    AMDS_REQUIRE(x && xh && rowstat && M >= 0, "amds_ln_stats_cast: bad arguments");
    AMDS_REQUIRE(D > 0 && D % 4 == 0 && D <= 64 * 4 * 8 && ldx % 4 == 0 && ldxh % 4 == 0, "amds_ln_stats_cast: D=%d must be a multiple of 4, at most 2048", D);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_ln_stats_cast: bad dtype %d", dtype);
    if (M == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)M * D * 6.0, st);
    const int grid = cdiv(M, 4);
    if (dtype == AMDS_F16) {
        if (D <= 1024) hipLaunchKernelGGL((ln_stats_cast_kernel<f16, 4>), dim3(grid), dim3(256), 0, st, x, ldx, (f16*)xh, ldxh, M, D, eps, rowstat);
        else hipLaunchKernelGGL((ln_stats_cast_kernel<f16, 8>), dim3(grid), dim3(256), 0, st, x, ldx, (f16*)xh, ldxh, M, D, eps, rowstat);
    } else {
        if (D <= 1024) hipLaunchKernelGGL((ln_stats_cast_kernel<bf16, 4>), dim3(grid), dim3(256), 0, st, x, ldx, (bf16*)xh, ldxh, M, D, eps, rowstat);
        else hipLaunchKernelGGL((ln_stats_cast_kernel<bf16, 8>), dim3(grid), dim3(256), 0, st, x, ldx, (bf16*)xh, ldxh, M, D, eps, rowstat);
    }
    AMDS_LAUNCH_CHECK("ln_stats_cast_kernel");
    return AMDS_OK;
}

extern "C" int amds_ln_stats_split(const float* x, long ldx, int M, int D, float eps, void* hi, void* lo, long ld, float* rowstat, void* stream) {
    AMDS_REQUIRE(x && hi && lo && rowstat && M >= 0, "amds_ln_stats_split: bad arguments");
    AMDS_REQUIRE(D > 0 && D % 4 == 0 && D <= 64 * 4 * 8 && ldx % 4 == 0 && ld % 4 == 0, "amds_ln_stats_split: D=%d must be a multiple of 4, at most 2048", D);
    if (M == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_LN, (double)M * D * 8.0, st);
    const int grid = cdiv(M, 4);
    if (D <= 1024) hipLaunchKernelGGL((ln_stats_cast_kernel<f16, 4>), dim3(grid), dim3(256), 0, st, x, ldx, (f16*)hi, ld, M, D, eps, rowstat, (f16*)lo);
    else hipLaunchKernelGGL((ln_stats_cast_kernel<f16, 8>), dim3(grid), dim3(256), 0, st, x, ldx, (f16*)hi, ld, M, D, eps, rowstat, (f16*)lo);
    AMDS_LAUNCH_CHECK("ln_stats_cast_kernel");
    return AMDS_OK;
}

extern "C" int amds_planes_to_f32(const void* hi, const void* lo, long ld, long row_stride, float* out, long ldo, long rows, int D, void* stream) {
    AMDS_REQUIRE(hi && lo && out && rows >= 0 && D > 0 && D % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && row_stride > 0, "amds_planes_to_f32: bad arguments");
    if (rows == 0) return AMDS_OK;
    const long total = rows * (D >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)hi, (const f16*)lo, ld, row_stride, out, ldo, rows, D);
    AMDS_LAUNCH_CHECK("planes_to_f32_kernel");
    return AMDS_OK;
}

extern "C" int amds_quick_gelu_inplace(void* u, long ld, long rows, int cols, int dtype, void* stream) {
    AMDS_REQUIRE(u && rows >= 0 && cols > 0 && cols % 8 == 0 && ld >= cols && ld % 8 == 0, "amds_quick_gelu_inplace: bad arguments (cols, ld multiples of 8)");
    if (rows == 0) return AMDS_OK;
    const long total = rows * (cols >> 3);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((quick_gelu_inplace_kernel<f16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (f16*)u, ld, rows, cols);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((quick_gelu_inplace_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16*)u, ld, rows, cols);
    else { set_error("amds_quick_gelu_inplace: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("quick_gelu_inplace_kernel");
    return AMDS_OK;
}

// ---- amds_check_finite: the guard in front of the feature file (include/amdstamp.h) ----
namespace amds {
template <typename T>
__global__ void __launch_bounds__(256) count_nonfinite_kernel(const T* __restrict__ x, long n, int* __restrict__ count) {
    int bad = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = (float)x[i];
        bad += !(fabsf(v) <= 3.402823466e38f);          // NaN compares false, +-inf exceeds FLT_MAX
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}
}  // namespace amds

extern "C" int amds_check_finite(const void* x, long n, int dtype, int* count_dev, int* count_host, void* stream) {
    using namespace amds;
    AMDS_REQUIRE(count_dev && count_host && n >= 0 && (x || n == 0), "amds_check_finite: bad arguments");
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16 || dtype == AMDS_F32, "amds_check_finite: dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    AMDS_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), st));
    if (n > 0) {
        const int grid = (int)((n + 256 * 8 - 1) / (256 * 8) < 2048 ? (n + 256 * 8 - 1) / (256 * 8) : 2048);
        if (dtype == AMDS_F16) hipLaunchKernelGGL(count_nonfinite_kernel<_Float16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const _Float16*>(x), n, count_dev);
        else if (dtype == AMDS_BF16) hipLaunchKernelGGL(count_nonfinite_kernel<__bf16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const __bf16*>(x), n, count_dev);
        else hipLaunchKernelGGL(count_nonfinite_kernel<float>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float*>(x), n, count_dev);
        AMDS_LAUNCH_CHECK("count_nonfinite_kernel");
    }
    AMDS_HIP(hipMemcpyAsync(count_host, count_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    AMDS_HIP(hipStreamSynchronize(st));
    if (*count_host != 0) {
        set_error("amds_check_finite: %d of %ld values are not finite (an intermediate left the 16-bit activation range)", *count_host, n);
        return AMDS_ERR_RANGE;
    }
    return AMDS_OK;
}

// ---- amds_export_words: device words -> host-mapped (pinned) memory by a KERNEL's stores (include/amdstamp.h) ----
namespace amds {
__global__ void __launch_bounds__(256) export_words_kernel(unsigned* __restrict__ src, unsigned* __restrict__ dst, long n, int zero_src) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        dst[i] = src[i];
        if (zero_src) src[i] = 0u;
    }
}
}  // namespace amds

extern "C" int amds_export_words(void* src_dev, void* dst_host_mapped, long n_words, int zero_src, void* stream) {
    using namespace amds;
    AMDS_REQUIRE(src_dev && dst_host_mapped && n_words >= 0, "amds_export_words: bad arguments");
    AMDS_REQUIRE(((uintptr_t)src_dev & 3) == 0 && ((uintptr_t)dst_host_mapped & 3) == 0, "amds_export_words: pointers must be 4-byte aligned");
    if (n_words == 0) return AMDS_OK;
    const long g = (n_words + 255) / 256;
    hipLaunchKernelGGL(export_words_kernel, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(src_dev),
                       reinterpret_cast<unsigned*>(dst_host_mapped), n_words, zero_src);
    AMDS_LAUNCH_CHECK("export_words_kernel");
    return AMDS_OK;
}
