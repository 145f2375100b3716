// train.hip -- the HBM-bound kernels of the MIL training step: everything around
// the GEMMs and the attention of `stamp train` on the `vit` head -- reference
// src/stamp/modeling/models/__init__.py:239-279 (LitTileClassifier._step: forward, loss, backward) and :133-141
// (AdamW + OneCycleLR).  Mixed precision as usual for training: bf16 MFMA operands (activations AND gradients: bf16
// keeps the fp32 exponent range, so no loss scaling), fp32 accumulation, fp32 residual stream / LayerNorm / master
// weights / optimizer state.
//   amds_transpose16          [R][C] 16-bit -> [C][ld] (weight-gradient GEMMs contract over the token dimension)
//   amds_colsum               bias gradients: deterministic two-stage column sums
//   amds_layernorm_train      LayerNorm forward that also stores mean / rstd
//   amds_layernorm_bwd        dx (+ skip gradient), per-block partial d(gamma), d(beta)  -> amds_colsum finishes them
//   amds_gelu_fwd / _bwd      exact-erf GELU on a stored pre-activation
//   amds_adamw                fused decoupled-weight-decay Adam on one flat fp32 parameter buffer
#include "common.h"

namespace amds {

// ---- 16-bit transpose through LDS (64 x 64 tiles, +1 padding) -------------------------------------------------------
__global__ void __launch_bounds__(256) transpose16_kernel(const uint16_t* __restrict__ src, long ld_src, uint16_t* __restrict__ dst,
                                                          long ld_dst, int R, int Cc) {
    __shared__ uint16_t t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        t[r][c] = (r0 + r < R && c0 + c < Cc) ? src[(long)(r0 + r) * ld_src + c0 + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < Cc && r0 + r < R) dst[(long)(c0 + c) * ld_dst + r0 + r] = t[r][c];
    }
}

// Fast path (R, Cc multiples of 64; 16-byte aligned rows on both sides): 16-byte global loads and stores, 4-byte LDS writes,
// rows pitched 33 dwords so that the 8 rows a lane gathers for one output chunk sit in 8 different banks.  The 2-byte-per-
// lane kernel above moved 3.2 TB/s (13 % of a MIL training step went into it).
__global__ void __launch_bounds__(256) transpose16_vec_kernel(const uint16_t* __restrict__ src, long ld_src, uint16_t* __restrict__ dst,
                                                              long ld_dst) {
    __shared__ uint32_t t[64 * 33];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u, r = i >> 3, ch = i & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)(r0 + r) * ld_src + c0 + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) t[r * 33 + ch * 4 + e] = v[e];
    }
    __syncthreads();
    const uint16_t* th = reinterpret_cast<const uint16_t*>(t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u, c = i >> 3, rg = i & 7;     // output row c, its elements r = rg*8 .. rg*8+7
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t lo = th[(rg * 8 + 2 * e) * 66 + c], hi = th[(rg * 8 + 2 * e + 1) * 66 + c];
            o[e] = lo | (hi << 16);
        }
        *reinterpret_cast<u32x4*>(dst + (long)(c0 + c) * ld_dst + r0 + rg * 8) = o;
    }
}

// ---- column sums: stage 1 = per 256-row chunk partials, stage 2 = reduce partials ---------------------------------------
// Stage 1: a block owns 64 columns x CS_ROWS rows; 16 column groups of 4 (one 8/16-byte load per row) x 16 row lanes, four
// independent accumulator sets per thread so that 4 loads are in flight; the 16 row lanes reduce through LDS in a fixed
// order.  (v1 walked 256 rows with one dependent 2-byte load per thread and iteration: 46 us for a 65600 x 512 bf16 matrix,
// 1.4 TB/s, and its second stage summed 257 partials serially in two blocks.)
constexpr int CS_ROWS = 1024;
template <typename TI>
__device__ __forceinline__ void colsum_block(const TI* __restrict__ x, long ld, float* __restrict__ partial, int M, int N, int vec_ok, int cs_rows, int bx, int by) {
    typedef TI vec4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 red[16][16];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = bx * 64 + cg * 4;
    const int r0 = by * cs_rows, r1 = min(M, r0 + cs_rows);
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (vec_ok && n + 3 < N) {
        const TI* p = x + n;
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const vec4 v = *reinterpret_cast<const vec4*>(p + (long)(r + 16 * u) * ld);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u][e] += (float)v[e];
            }
        }
        for (; r < r1; r += 16) {
            const vec4 v = *reinterpret_cast<const vec4*>(p + (long)r * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][e] += (float)v[e];
        }
    } else {
        for (int r = r0 + rl; r < r1; r += 16)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < N) acc[0][e] += (float)x[(long)r * ld + n + e];
    }
    red[rl][cg] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;            // column inside the block
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][c >> 2][c & 3];
        if (bx * 64 + c < N) partial[(long)by * N + bx * 64 + c] = s;
    }
}
template <typename TI>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const TI* __restrict__ x, long ld, float* __restrict__ partial, int M, int N, int vec_ok, int cs_rows) {
    colsum_block<TI>(x, ld, partial, M, N, vec_ok, cs_rows, blockIdx.x, blockIdx.y);
}
__device__ __forceinline__ void colsum_final_cols(const float* __restrict__ partial, float* __restrict__ out, int nchunk, int N, int accumulate, int n) {
    if (n >= N) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 3 < nchunk; c += 4) {
        s0 += partial[(long)c * N + n]; s1 += partial[(long)(c + 1) * N + n];
        s2 += partial[(long)(c + 2) * N + n]; s3 += partial[(long)(c + 3) * N + n];
    }
    for (; c < nchunk; ++c) s0 += partial[(long)c * N + n];
    const float s = (s0 + s1) + (s2 + s3);
    out[n] = accumulate ? out[n] + s : s;
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nchunk, int N, int accumulate) {
    colsum_final_cols(partial, out, nchunk, N, accumulate, blockIdx.x * blockDim.x + threadIdx.x);
}
// Many small column sums as ONE launch (amds_colsum_multi): entry i owns the blocks [block0[i], block0[i + 1]); kind 0 = the one-chunk sum of
// amds_colsum (64 columns per block), kind 1 = the second stage over chunk partials (256 columns per block).  The arithmetic of the two single-entry
// kernels above, block for block: the same bits.
struct ColsumTable {
    const float* x[32];
    float* out[32];
    long ld[32];
    int rows[32], cols[32], kind[32], vec_ok[32];
    int block0[33];
    int n;
};
__global__ void __launch_bounds__(256) colsum_multi_kernel(ColsumTable tb) {
    int ei = 0;
    while (ei + 1 < tb.n && (int)blockIdx.x >= tb.block0[ei + 1]) ++ei;
    const int bx = (int)blockIdx.x - tb.block0[ei];
    if (tb.kind[ei] == 0) colsum_block<float>(tb.x[ei], tb.ld[ei], tb.out[ei], tb.rows[ei], tb.cols[ei], tb.vec_ok[ei], tb.rows[ei], bx, 0);
    else colsum_final_cols(tb.x[ei], tb.out[ei], tb.rows[ei], tb.cols[ei], 0, bx * 256 + (int)threadIdx.x);
}

// ---- LayerNorm forward with saved statistics ---------------------------------------------------------------------------
template <typename TO, int MAXV>
__global__ void __launch_bounds__(256) ln_train_kernel(const float* __restrict__ x, long xs, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, TO* __restrict__ y, long ys,
                                                       float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows, int cols, float eps,
                                                       float* __restrict__ xcopy, long xcs, int copy_cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * xs;
    const int nv = cols >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) { v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    if (xcopy) {      // the rows also go out unchanged (the residual GEMM that follows updates the copy in place): saves a device-to-device copy launch
        float* cr = xcopy + (long)row * xcs;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) *reinterpret_cast<f32x4*>(cr + c * 4) = v[i];
        }
        for (int c = nv + lane; c < (copy_cols >> 2); c += 64) *reinterpret_cast<f32x4*>(cr + c * 4) = *reinterpret_cast<const f32x4*>(xr + c * 4);
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    TO* yr = y + (long)row * ys;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
            typedef TO vec4 __attribute__((ext_vector_type(4)));
            vec4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = (TO)((v[i][e] - mean) * rstd * g[e] + b[e]);
            *reinterpret_cast<vec4*>(yr + c * 4) = w;
        }
    }
}

// ---- LayerNorm backward: one wave per row; 64 rows per block; per-block partial d(gamma), d(beta) ----------------------------
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  dx_io = (add_skip ? dx_io : 0) + dx
template <int MAXV, typename T16 = bf16>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ dy, long dys, const float* __restrict__ x, long xs,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dx, long dxs, int add_skip,
                                                     float* __restrict__ dgp, float* __restrict__ dbp, int rows, int cols,
                                                     T16* __restrict__ o16, long o16s, uint64_t dseed, uint32_t dsid, uint32_t dthr, float dscale) {
    extern __shared__ float red[];        // [2][4][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = cols >> 2;
    f32x4 ag[MAXV], ab[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int r0 = blockIdx.x * 64;
    for (int rr = wave; rr < 64; rr += 4) {
        const int row = r0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        f32x4 gv[MAXV], xh[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (long)row * dys + c * 4);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)row * xs + c * 4);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    gv[i][e] = d[e] * gm[e];
                    s1 += gv[i][e];
                    s2 += gv[i][e] * xh[i][e];
                    ag[i][e] += d[e] * xh[i][e];
                    ab[i][e] += d[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                f32x4* p = reinterpret_cast<f32x4*>(dx + (long)row * dxs + c * 4);
                f32x4 o = add_skip ? *p : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rs * (gv[i][e] - s1 - xh[i][e] * s2);
                *p = o;
                if (o16) {      // the bf16 operand of the GEMMs that consume dx next, with the NEXT dropout site's mask when there is one (amds_dropout_cast_bwd's bits:
                                // flat element index row * cols + column) -- the rows are in registers anyway, a separate cast pass re-reads them
                    typedef T16 bvec4 __attribute__((ext_vector_type(4)));
                    bvec4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool keep = dthr == 0 || drop_keep_flat(dseed, dsid, (long)row * cols + c * 4 + e, dthr);
                        w[e] = (T16)(keep ? (dthr ? o[e] * dscale : o[e]) : 0.f);
                    }
                    *reinterpret_cast<bvec4*>(o16 + (long)row * o16s + c * 4) = w;
                }
            }
        }
    }
    // block reduction of the 4 waves' d(gamma) / d(beta) partials
    float* rg = red + wave * cols;
    float* rb = red + (4 + wave) * cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { rg[c * 4 + e] = ag[i][e]; rb[c * 4 + e] = ab[i][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        dgp[(long)blockIdx.x * cols + c] = (red[c] + red[cols + c]) + (red[2 * cols + c] + red[3 * cols + c]);
        dbp[(long)blockIdx.x * cols + c] = (red[4 * cols + c] + red[5 * cols + c]) + (red[6 * cols + c] + red[7 * cols + c]);
    }
}

// ---- GELU on a stored pre-activation ---------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void gelu_fwd_kernel(const TI* __restrict__ z, TO* __restrict__ u, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) u[i] = (TO)gelu_erf((float)z[i]);
}
template <typename TZ, typename TG, typename TO>
__global__ void gelu_bwd_kernel(const TZ* __restrict__ z, const TG* __restrict__ du, TO* __restrict__ dz, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = (float)z[i];
        const float d = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
        dz[i] = (TO)((float)du[i] * d);
    }
}

// ---- AdamW (torch.optim.AdamW semantics, amsgrad=False) -----------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float pi = p[i] * (1.0f - lr * wd);
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
        p[i] = pi;
    }
}

__global__ void f16_to_bf16_kernel(const f16* __restrict__ src, bf16* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (bf16)(float)src[i];
}

static inline int grid1d(long n) { return (int)min((long)4096, (n + 255) / 256); }

}  // namespace amds

using namespace amds;

extern "C" int amds_transpose16(const void* src, long ld_src, void* dst, long ld_dst, int R, int Cc, void* stream) {
    AMDS_REQUIRE(src && dst && src != dst, "amds_transpose16: null/aliased pointer");
    AMDS_REQUIRE(R > 0 && Cc > 0 && ld_src >= Cc && ld_dst >= R, "amds_transpose16: bad shape");
    const bool vec = R % 64 == 0 && Cc % 64 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(transpose16_vec_kernel, dim3(Cc / 64, R / 64), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, ld_src,
                           (uint16_t*)dst, ld_dst);
    else
        hipLaunchKernelGGL(transpose16_kernel, dim3(cdiv(Cc, 64), cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, ld_src,
                           (uint16_t*)dst, ld_dst, R, Cc);
    AMDS_LAUNCH_CHECK("transpose16_kernel");
    return AMDS_OK;
}

// ---- amds_cast_transpose_multi: fp32 masters -> 16-bit operand copies + their transposes, every matrix of a model in ONE launch ----
namespace amds {
struct CastTable {
    amds_cast_entry e[32];
    int tile0[33];          // first 64 x 64 tile of entry i in the launch's tile sequence
    int n;
};
template <typename TO>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef TO v2 __attribute__((ext_vector_type(2)));
    v2 p;
    p[0] = (TO)a;
    p[1] = (TO)b;
    return __builtin_bit_cast(uint32_t, p);
}
__global__ void __launch_bounds__(256) cast_transpose_multi_kernel(CastTable tb) {
    __shared__ uint32_t t[64 * 33];
    int ei = 0;
    while (ei + 1 < tb.n && (int)blockIdx.x >= tb.tile0[ei + 1]) ++ei;
    const amds_cast_entry& e = tb.e[ei];
    const int tl = blockIdx.x - tb.tile0[ei], tc = e.cols / 64;
    const int r0 = (tl / tc) * 64, c0 = (tl % tc) * 64;
    uint16_t* dst = reinterpret_cast<uint16_t*>(e.dst);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u, r = i >> 3, ch = i & 7;
        const float* sp = e.src + (long)(r0 + r) * e.ld_src + c0 + ch * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
        u32x4 v;
        if (e.dtype == AMDS_F16) v = u32x4{pack2<f16>(a[0], a[1]), pack2<f16>(a[2], a[3]), pack2<f16>(b[0], b[1]), pack2<f16>(b[2], b[3])};
        else v = u32x4{pack2<bf16>(a[0], a[1]), pack2<bf16>(a[2], a[3]), pack2<bf16>(b[0], b[1]), pack2<bf16>(b[2], b[3])};
        *reinterpret_cast<u32x4*>(dst + (long)(r0 + r) * e.ld_dst + c0 + ch * 8) = v;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[r * 33 + ch * 4 + q] = v[q];
    }
    if (e.dst_t == nullptr) return;
    __syncthreads();
    const uint16_t* th = reinterpret_cast<const uint16_t*>(t);
    uint16_t* dt = reinterpret_cast<uint16_t*>(e.dst_t);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u, c = i >> 3, rg = i & 7;     // output row c, its elements r = rg*8 .. rg*8+7
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lo = th[(rg * 8 + 2 * q) * 66 + c], hi = th[(rg * 8 + 2 * q + 1) * 66 + c];
            o[q] = lo | (hi << 16);
        }
        *reinterpret_cast<u32x4*>(dt + (long)(c0 + c) * e.ld_dst_t + r0 + rg * 8) = o;
    }
}
}  // namespace amds

extern "C" int amds_cast_transpose_multi(const amds_cast_entry* entries_host, int n, void* stream) {
    AMDS_REQUIRE(entries_host && n > 0 && n <= 32, "amds_cast_transpose_multi: 1 .. 32 entries (n=%d)", n);
    CastTable tb;
    tb.n = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const amds_cast_entry& e = entries_host[i];
        AMDS_REQUIRE(e.src && e.dst && e.rows > 0 && e.cols > 0 && e.rows % 64 == 0 && e.cols % 64 == 0, "amds_cast_transpose_multi: entry %d: rows / cols must be multiples of 64", i);
        AMDS_REQUIRE(e.ld_src >= e.cols && e.ld_dst >= e.cols && e.ld_src % 4 == 0 && e.ld_dst % 8 == 0 && (e.dst_t == nullptr || (e.ld_dst_t >= e.rows && e.ld_dst_t % 8 == 0)),
                     "amds_cast_transpose_multi: entry %d: bad leading dimensions", i);
        AMDS_REQUIRE(((uintptr_t)e.src & 15) == 0 && ((uintptr_t)e.dst & 15) == 0 && ((uintptr_t)e.dst_t & 15) == 0, "amds_cast_transpose_multi: entry %d: pointers must be 16-byte aligned", i);
        AMDS_REQUIRE(e.dtype == AMDS_F16 || e.dtype == AMDS_BF16, "amds_cast_transpose_multi: entry %d: dtype %d", i, e.dtype);
        tb.e[i] = e;
        tb.tile0[i] = tiles;
        tiles += (e.rows / 64) * (e.cols / 64);
    }
    tb.tile0[n] = tiles;
    hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, tb);
    AMDS_LAUNCH_CHECK("cast_transpose_multi_kernel");
    return AMDS_OK;
}

// ---- amds_sum_partials_multi: out_i[e] = sum over s of part_i[s][e] for EVERY weight gradient of a backward pass in one launch ----
namespace amds {
struct SumTable {
    const float* part[32];
    float* out[32];
    long count[32];          // elements of entry i (a multiple of 4)
    long vec0[33];           // first float4 of entry i in the launch's sequence
    int n, rows;
};
// the association of amds_colsum's one-chunk path (16 row lanes, each adding its rows r, r + 16, r + 32, ... in order; the lanes added in order): the same bits
__global__ void __launch_bounds__(256) sum_partials_multi_kernel(SumTable tb) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= tb.vec0[tb.n]) return;
    int ei = 0;
    while (ei + 1 < tb.n && v >= tb.vec0[ei + 1]) ++ei;
    const long e = (v - tb.vec0[ei]) * 4;
    const float* p = tb.part[ei] + e;
    const long ld = tb.count[ei];
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 16 && k < tb.rows; ++k) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = k; r < tb.rows; r += 16) a += *reinterpret_cast<const f32x4*>(p + (long)r * ld);
        s += a;
    }
    *reinterpret_cast<f32x4*>(tb.out[ei] + e) = s;
}
}  // namespace amds

extern "C" int amds_sum_partials_multi(const float* const* parts_host, float* const* outs_host, const long* counts_host, int n, int rows, void* stream) {
    AMDS_REQUIRE(parts_host && outs_host && counts_host && n > 0 && n <= 32 && rows > 0 && rows <= 2048, "amds_sum_partials_multi: 1 .. 32 entries of 1 .. 2048 partial rows (n=%d rows=%d)", n, rows);
    SumTable tb;
    tb.n = n;
    tb.rows = rows;
    long vecs = 0;
    for (int i = 0; i < n; ++i) {
        AMDS_REQUIRE(parts_host[i] && outs_host[i] && counts_host[i] > 0 && counts_host[i] % 4 == 0, "amds_sum_partials_multi: entry %d: count must be a positive multiple of 4", i);
        AMDS_REQUIRE(((uintptr_t)parts_host[i] & 15) == 0 && ((uintptr_t)outs_host[i] & 15) == 0, "amds_sum_partials_multi: entry %d: pointers must be 16-byte aligned", i);
        tb.part[i] = parts_host[i];
        tb.out[i] = outs_host[i];
        tb.count[i] = counts_host[i];
        tb.vec0[i] = vecs;
        vecs += counts_host[i] / 4;
    }
    tb.vec0[n] = vecs;
    hipLaunchKernelGGL(sum_partials_multi_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tb);
    AMDS_LAUNCH_CHECK("sum_partials_multi_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_colsum_workspace_bytes(int M, int N) { return (size_t)cdiv(M, CS_ROWS) * N * 4; }
// stage 1 of amds_colsum alone: part[c][n] = sum of the rows of chunk c (chunks of CS_ROWS rows; up to 2 x CS_ROWS rows are ONE chunk)
static int colsum_stage1(const void* x, long ld, float* part, int M, int N, int in_dtype, int nchunk, hipStream_t st) {
    const int cs_rows = nchunk == 1 ? M : CS_ROWS;
    const dim3 grid(cdiv(N, 64), nchunk);
    const int esz = in_dtype == AMDS_F32 ? 4 : 2;
    const int vec_ok = (ld % 4 == 0) && (((uintptr_t)x % (4 * esz)) == 0);      // 4-element vector loads need aligned rows
    // (Round 5 tried the second stage inside the first launch -- the block that arrives last for its 64 columns adds the chunk partials: the
    //  agent-scope fence that makes the partials of the other XCDs' L2s visible costs ~100 us per launch on this 8-XCD part, 6x the launch it saves.)
    if (in_dtype == AMDS_F32) hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, st, (const float*)x, ld, part, M, N, vec_ok, cs_rows);
    else if (in_dtype == AMDS_BF16) hipLaunchKernelGGL((colsum_partial_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, ld, part, M, N, vec_ok, cs_rows);
    else if (in_dtype == AMDS_F16) hipLaunchKernelGGL((colsum_partial_kernel<f16>), grid, dim3(256), 0, st, (const f16*)x, ld, part, M, N, vec_ok, cs_rows);
    else { set_error("amds_colsum: bad dtype"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("colsum_partial_kernel");
    return AMDS_OK;
}
static inline int colsum_chunks(int M) { return M <= 2 * CS_ROWS ? 1 : cdiv(M, CS_ROWS); }

extern "C" int amds_colsum(const void* x, long ld, float* out, int M, int N, int in_dtype, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(x && out && ws, "amds_colsum: null pointer");
    AMDS_REQUIRE(M > 0 && N > 0, "amds_colsum: bad shape");
    // up to 2 x CS_ROWS rows are ONE chunk (split-K partials: 32 rows; LayerNorm parameter-gradient partials: rows / 64; bags): without
    // `accumulate` its "partial" IS the result and the second launch is skipped (30 -> 19 colsum_final launches per MIL training step)
    const int nchunk = colsum_chunks(M);
    if (ws_bytes < (size_t)nchunk * N * 4) { set_error("amds_colsum: workspace too small"); return AMDS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const bool direct = nchunk == 1 && !accumulate;
    float* part = direct ? out : (float*)ws;
    int rc = colsum_stage1(x, ld, part, M, N, in_dtype, nchunk, st);
    if (rc != AMDS_OK || direct) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, part, out, nchunk, N, accumulate);
    AMDS_LAUNCH_CHECK("colsum_final_kernel");
    return AMDS_OK;
}

extern "C" int amds_colsum_partials(const void* x, long ld, float* part, int M, int N, int in_dtype, int* nchunk_out, void* stream) {
    AMDS_REQUIRE(x && part && nchunk_out, "amds_colsum_partials: null pointer");
    AMDS_REQUIRE(M > 0 && N > 0, "amds_colsum_partials: bad shape");
    const int nchunk = colsum_chunks(M);
    *nchunk_out = nchunk;
    return colsum_stage1(x, ld, part, M, N, in_dtype, nchunk, (hipStream_t)stream);
}

extern "C" int amds_colsum_multi(const amds_colsum_entry* entries_host, int n, void* stream) {
    AMDS_REQUIRE(entries_host && n > 0 && n <= 32, "amds_colsum_multi: 1 .. 32 entries (n=%d)", n);
    ColsumTable tb;
    tb.n = n;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const amds_colsum_entry& e = entries_host[i];
        AMDS_REQUIRE(e.x && e.out && e.rows > 0 && e.cols > 0 && (e.kind == 0 || e.kind == 1), "amds_colsum_multi: entry %d: bad pointers / shape / kind", i);
        AMDS_REQUIRE(e.kind == 1 || (e.rows <= 2 * CS_ROWS && e.ld >= e.cols), "amds_colsum_multi: entry %d: a direct sum takes at most %d rows (got %d) and ld >= cols", i, 2 * CS_ROWS, e.rows);
        tb.x[i] = e.x; tb.out[i] = e.out; tb.ld[i] = e.kind == 0 ? e.ld : e.cols;
        tb.rows[i] = e.rows; tb.cols[i] = e.cols; tb.kind[i] = e.kind;
        tb.vec_ok[i] = (e.ld % 4 == 0) && (((uintptr_t)e.x % 16) == 0);
        tb.block0[i] = (int)blocks;
        blocks += e.kind == 0 ? cdiv(e.cols, 64) : cdiv(e.cols, 256);
        AMDS_REQUIRE(blocks < (1L << 30), "amds_colsum_multi: too many columns");
    }
    tb.block0[n] = (int)blocks;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tb);
    AMDS_LAUNCH_CHECK("colsum_multi_kernel");
    return AMDS_OK;
}

extern "C" int amds_layernorm_train(const float* x, long x_row_stride, const float* gamma, const float* beta, void* y, long y_row_stride,
                                    float* mean, float* rstd, int rows, int cols, float eps, int out_dtype, void* stream) {
    return amds_layernorm_train_copy(x, x_row_stride, gamma, beta, y, y_row_stride, mean, rstd, rows, cols, eps, out_dtype, nullptr, 0, 0, stream);
}

extern "C" int amds_layernorm_train_copy(const float* x, long x_row_stride, const float* gamma, const float* beta, void* y, long y_row_stride,
                                         float* mean, float* rstd, int rows, int cols, float eps, int out_dtype, float* x_copy, long copy_row_stride,
                                         int copy_cols, void* stream) {
    AMDS_REQUIRE(x && gamma && beta && y && mean && rstd, "amds_layernorm_train: null pointer");
    AMDS_REQUIRE(!x_copy || (copy_cols >= cols && copy_cols % 4 == 0 && copy_cols <= x_row_stride && copy_cols <= copy_row_stride && x_copy != x),
                 "amds_layernorm_train_copy: copy_cols=%d must be a multiple of 4 in [cols, row strides]", copy_cols);
    AMDS_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "amds_layernorm_train: cols=%d must be a multiple of 4 and <= 2048", cols);
    if (rows == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv(rows, 4)), block(256);
    // float4 slots per lane by row width (2 = 512 columns: the MIL heads): the same arithmetic in a quarter of the registers
#define AMDS_LN_TRAIN(TO_, MV)                                                                                                                \
    hipLaunchKernelGGL((ln_train_kernel<TO_, MV>), grid, block, 0, st, x, x_row_stride, gamma, beta, (TO_*)y, y_row_stride, mean, rstd, rows, cols, eps, x_copy, copy_row_stride, copy_cols)
#define AMDS_LN_TRAIN_W(TO_)                                   \
    do {                                                       \
        if (cols <= 512) AMDS_LN_TRAIN(TO_, 2);                \
        else if (cols <= 1024) AMDS_LN_TRAIN(TO_, 4);          \
        else AMDS_LN_TRAIN(TO_, 8);                            \
    } while (0)
    if (out_dtype == AMDS_BF16) AMDS_LN_TRAIN_W(bf16);
    else if (out_dtype == AMDS_F16) AMDS_LN_TRAIN_W(f16);
    else if (out_dtype == AMDS_F32) AMDS_LN_TRAIN_W(float);
    else { set_error("amds_layernorm_train: bad dtype"); return AMDS_ERR_INVALID; }
#undef AMDS_LN_TRAIN_W
#undef AMDS_LN_TRAIN
    AMDS_LAUNCH_CHECK("ln_train_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_layernorm_bwd_workspace_bytes(int rows, int cols) { return (size_t)cdiv(rows, 64) * cols * 4 * 2 + amds_colsum_workspace_bytes(cdiv(rows, 64), cols); }
extern "C" int amds_layernorm_bwd(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd,
                                  const float* gamma, float* dx, long dx_stride, int add_skip, float* dgamma, float* dbeta, int accumulate_params,
                                  int rows, int cols, void* ws, size_t ws_bytes, void* stream) {
    return amds_layernorm_bwd_cast(dy, dy_stride, x, x_stride, mean, rstd, gamma, dx, dx_stride, add_skip, dgamma, dbeta, accumulate_params, rows, cols, ws, ws_bytes,
                                   nullptr, 0, 0.f, 0, 0, stream);
}

extern "C" int amds_layernorm_bwd_partials(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd,
                                           const float* gamma, float* dx, long dx_stride, int add_skip, float* dgamma_part, float* dbeta_part,
                                           int rows, int cols, void* dx_bf16, long dx_bf16_stride, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    return layernorm_bwd_partials_dt(dy, dy_stride, x, x_stride, mean, rstd, gamma, dx, dx_stride, add_skip, dgamma_part, dbeta_part, rows, cols, dx_bf16, dx_bf16_stride,
                                     AMDS_BF16, p, seed, stream_id, stream);
}

int amds::layernorm_bwd_partials_dt(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd, const float* gamma, float* dx,
                                    long dx_stride, int add_skip, float* dgamma_part, float* dbeta_part, int rows, int cols, void* dx_bf16, long dx_bf16_stride,
                                    int dx16_dtype, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma_part && dbeta_part, "amds_layernorm_bwd: null pointer");
    AMDS_REQUIRE(dx16_dtype == AMDS_BF16 || dx16_dtype == AMDS_F16, "amds_layernorm_bwd: the 16-bit copy is bf16 or fp16");
    AMDS_REQUIRE(!dx_bf16 || (dx_bf16_stride >= cols && dx_bf16_stride % 4 == 0 && p >= 0.f && p < 1.f), "amds_layernorm_bwd_cast: bad 16-bit output / rate");
    const uint32_t dthr = (dx_bf16 && p > 0.f) ? drop_thr16(p) : 0;
    const float dscale = dthr ? drop_scale(dthr) : 1.0f;
    AMDS_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "amds_layernorm_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = cdiv(rows, 64);
#define AMDS_LN_BWD(MV)                                                                                                                                \
    do {                                                                                                                                               \
        if (dx16_dtype == AMDS_F16)                                                                                                                    \
            hipLaunchKernelGGL((ln_bwd_kernel<MV, f16>), dim3(nblk), dim3(256), (size_t)8 * cols * 4, st, dy, dy_stride, x, x_stride, mean, rstd, gamma, dx, \
                               dx_stride, add_skip, dgamma_part, dbeta_part, rows, cols, (f16*)dx_bf16, dx_bf16_stride, seed, stream_id, dthr, dscale);  \
        else                                                                                                                                           \
            hipLaunchKernelGGL((ln_bwd_kernel<MV, bf16>), dim3(nblk), dim3(256), (size_t)8 * cols * 4, st, dy, dy_stride, x, x_stride, mean, rstd, gamma, dx, \
                               dx_stride, add_skip, dgamma_part, dbeta_part, rows, cols, (bf16*)dx_bf16, dx_bf16_stride, seed, stream_id, dthr, dscale); \
    } while (0)
    if (cols <= 512) AMDS_LN_BWD(2);
    else if (cols <= 1024) AMDS_LN_BWD(4);
    else AMDS_LN_BWD(8);
#undef AMDS_LN_BWD
    AMDS_LAUNCH_CHECK("ln_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_layernorm_bwd_cast(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd,
                                       const float* gamma, float* dx, long dx_stride, int add_skip, float* dgamma, float* dbeta, int accumulate_params,
                                       int rows, int cols, void* ws, size_t ws_bytes, void* dx_bf16, long dx_bf16_stride, float p, uint64_t seed,
                                       uint32_t stream_id, void* stream) {
    return layernorm_bwd_cast_dt(dy, dy_stride, x, x_stride, mean, rstd, gamma, dx, dx_stride, add_skip, dgamma, dbeta, accumulate_params, rows, cols, ws, ws_bytes, dx_bf16,
                                 dx_bf16_stride, AMDS_BF16, p, seed, stream_id, stream);
}

int amds::layernorm_bwd_cast_dt(const float* dy, long dy_stride, const float* x, long x_stride, const float* mean, const float* rstd, const float* gamma, float* dx,
                                long dx_stride, int add_skip, float* dgamma, float* dbeta, int accumulate_params, int rows, int cols, void* ws, size_t ws_bytes, void* dx_bf16,
                                long dx_bf16_stride, int dx16_dtype, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(dgamma && dbeta && ws, "amds_layernorm_bwd: null pointer");
    AMDS_REQUIRE(rows > 0 && cols > 0, "amds_layernorm_bwd: bad shape");
    if (ws_bytes < amds_layernorm_bwd_workspace_bytes(rows, cols)) { set_error("amds_layernorm_bwd: workspace too small"); return AMDS_ERR_WORKSPACE; }
    const int nblk = cdiv(rows, 64);
    float* dgp = (float*)ws;
    float* dbp = dgp + (size_t)nblk * cols;
    char* cws = (char*)(dbp + (size_t)nblk * cols);
    const size_t cws_bytes = amds_colsum_workspace_bytes(nblk, cols);
    int rc = layernorm_bwd_partials_dt(dy, dy_stride, x, x_stride, mean, rstd, gamma, dx, dx_stride, add_skip, dgp, dbp, rows, cols, dx_bf16, dx_bf16_stride, dx16_dtype, p,
                                       seed, stream_id, stream);
    if (rc != AMDS_OK) return rc;
    rc = amds_colsum(dgp, cols, dgamma, nblk, cols, AMDS_F32, accumulate_params, cws, cws_bytes, stream);
    if (rc != AMDS_OK) return rc;
    return amds_colsum(dbp, cols, dbeta, nblk, cols, AMDS_F32, accumulate_params, cws, cws_bytes, stream);
}

extern "C" int amds_gelu_fwd(const void* z, void* u, long n, int in_dtype, int out_dtype, void* stream) {
    AMDS_REQUIRE(z && u && n >= 0, "amds_gelu_fwd: bad arguments");
    if (n == 0) return AMDS_OK;
    // 16-bit input, n a multiple of 8: the 8-elements-per-lane kernel of dropout.hip at rate 0 (every element kept, scale exactly 1: same bits)
    if ((in_dtype == AMDS_BF16 || in_dtype == AMDS_F16) && n % 8 == 0 && (((uintptr_t)z | (uintptr_t)u) & 15) == 0)
        return amds_gelu_dropout_fwd(z, u, n, in_dtype, out_dtype, 0.f, 0, 0, stream);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == AMDS_BF16 && out_dtype == AMDS_BF16) hipLaunchKernelGGL((gelu_fwd_kernel<bf16, bf16>), dim3(grid1d(n)), dim3(256), 0, st, (const bf16*)z, (bf16*)u, n);
    else if (in_dtype == AMDS_BF16 && out_dtype == AMDS_F32) hipLaunchKernelGGL((gelu_fwd_kernel<bf16, float>), dim3(grid1d(n)), dim3(256), 0, st, (const bf16*)z, (float*)u, n);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F16) hipLaunchKernelGGL((gelu_fwd_kernel<f16, f16>), dim3(grid1d(n)), dim3(256), 0, st, (const f16*)z, (f16*)u, n);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F32) hipLaunchKernelGGL((gelu_fwd_kernel<f16, float>), dim3(grid1d(n)), dim3(256), 0, st, (const f16*)z, (float*)u, n);
    else if (in_dtype == AMDS_F32 && out_dtype == AMDS_F32) hipLaunchKernelGGL((gelu_fwd_kernel<float, float>), dim3(grid1d(n)), dim3(256), 0, st, (const float*)z, (float*)u, n);
    else { set_error("amds_gelu_fwd: unsupported dtype pair"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("gelu_fwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_gelu_bwd(const void* z, const void* du, void* dz, long n, int z_dtype, int du_dtype, int dz_dtype, void* stream) {
    AMDS_REQUIRE(z && du && dz && n >= 0, "amds_gelu_bwd: bad arguments");
    if (n == 0) return AMDS_OK;
    if (((z_dtype == AMDS_BF16 && dz_dtype == AMDS_BF16) || (z_dtype == AMDS_F16 && dz_dtype == AMDS_F16)) && n % 8 == 0 &&
        (((uintptr_t)z | (uintptr_t)du | (uintptr_t)dz) & 15) == 0)
        return amds_gelu_dropout_bwd(z, du, dz, n, z_dtype, du_dtype, dz_dtype, 0.f, 0, 0, stream);
    hipStream_t st = (hipStream_t)stream;
    if (z_dtype == AMDS_BF16 && du_dtype == AMDS_BF16 && dz_dtype == AMDS_BF16)
        hipLaunchKernelGGL((gelu_bwd_kernel<bf16, bf16, bf16>), dim3(grid1d(n)), dim3(256), 0, st, (const bf16*)z, (const bf16*)du, (bf16*)dz, n);
    else if (z_dtype == AMDS_BF16 && du_dtype == AMDS_F32 && dz_dtype == AMDS_BF16)
        hipLaunchKernelGGL((gelu_bwd_kernel<bf16, float, bf16>), dim3(grid1d(n)), dim3(256), 0, st, (const bf16*)z, (const float*)du, (bf16*)dz, n);
    else if (z_dtype == AMDS_F16 && du_dtype == AMDS_F16 && dz_dtype == AMDS_F16)
        hipLaunchKernelGGL((gelu_bwd_kernel<f16, f16, f16>), dim3(grid1d(n)), dim3(256), 0, st, (const f16*)z, (const f16*)du, (f16*)dz, n);
    else if (z_dtype == AMDS_F16 && du_dtype == AMDS_F32 && dz_dtype == AMDS_F16)
        hipLaunchKernelGGL((gelu_bwd_kernel<f16, float, f16>), dim3(grid1d(n)), dim3(256), 0, st, (const f16*)z, (const float*)du, (f16*)dz, n);
    else if (z_dtype == AMDS_F32 && du_dtype == AMDS_F32 && dz_dtype == AMDS_F32)
        hipLaunchKernelGGL((gelu_bwd_kernel<float, float, float>), dim3(grid1d(n)), dim3(256), 0, st, (const float*)z, (const float*)du, (float*)dz, n);
    else { set_error("amds_gelu_bwd: unsupported dtype combination"); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("gelu_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, void* stream) {
    AMDS_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "amds_adamw: bad arguments");
    if (n == 0) return AMDS_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    AMDS_LAUNCH_CHECK("adamw_kernel");
    return AMDS_OK;
}

extern "C" int amds_convert_f16_bf16(const void* src, void* dst, long n, void* stream) {
    AMDS_REQUIRE(src && dst && n >= 0, "amds_convert_f16_bf16: bad arguments");
    if (n == 0) return AMDS_OK;
    hipLaunchKernelGGL(f16_to_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)src, (bf16*)dst, n);
    AMDS_LAUNCH_CHECK("f16_to_bf16_kernel");
    return AMDS_OK;
}
