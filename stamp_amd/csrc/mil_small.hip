// mil_small.hip -- the small, HBM/latency-bound rows of the MIL path:
//   * amds_gather_rows     fixed-size bag building: gather sampled tile rows (+ fp16 -> fp32 ".float()") and zero-pad
//                          (reference src/stamp/modeling/data.py:811-862 `_to_fixed_size_bag`, :584-655 BagDataset)
//   * amds_vary_precision  random mantissa truncation, pure integer bit-ops (reference src/stamp/modeling/transforms.py:5-29)
//   * amds_mean_pool       mean over the tiles of a bag (reference src/stamp/modeling/models/mlp.py:40-41, 58-59)
//   * amds_linear_f32      exact-fp32 Linear (+ReLU) on the fp32 MFMA, for the MLP / Linear heads (mlp.py:24-33, 50-51)
#include "common.h"

namespace amds {
int gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* Cout, int ldc, int M, int N,
             int K, int relu, hipStream_t st);   // gap.hip

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) gather_rows_kernel(const TI* __restrict__ src, long src_ld, const long* __restrict__ idx,
                                                          int n_idx, TO* __restrict__ dst, long dst_ld, int n_out, int cols) {
    const int row = blockIdx.x;
    if (row >= n_out) return;
    TO* d = dst + (long)row * dst_ld;
    if (row < n_idx) {
        const TI* s = src + idx[row] * src_ld;
        for (int c = threadIdx.x; c < cols; c += 256) d[c] = (TO)s[c];
    } else {
        for (int c = threadIdx.x; c < cols; c += 256) d[c] = (TO)0.f;
    }
}

template <typename W>
__global__ void vary_precision_kernel(const W* __restrict__ in, const uint8_t* __restrict__ shifts, W* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i] & (W)(~(W)0 << shifts[i]);
}

template <typename TI>
__global__ void __launch_bounds__(256) mean_pool_kernel(const TI* __restrict__ x, float* __restrict__ out, int T, int F) {
    const int b = blockIdx.y, f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const TI* p = x + (long)b * T * F + f;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += (float)p[(long)t * F];    // sequential over tiles: deterministic
    out[(long)b * F + f] = s / (float)T;
}

// gradient of the mean over tiles: dx[b][t][f] = dy[b][f] / T
__global__ void mean_pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long n, int T, int F) {
    const float inv = 1.0f / (float)T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long bt = i / F;
        dx[i] = dy[(bt / T) * F + (i - bt * F)] * inv;
    }
}


// EAGLE's tile selection (reference src/stamp/encoding/encoder/eagle.py:106-118): the k largest attention scores of a slide (k <= 32,
// ties -> the lower index), in descending order, and the mean of the matching rows of a second feature matrix.  One workgroup: k rounds
// of an argmax over the n scores (n ~ 10^4..10^5, k = 25: a few 10 us), then the column means.
template <typename T>
__global__ void __launch_bounds__(1024) topk_rows_mean_kernel(const float* __restrict__ score, int n, int k, const T* __restrict__ rows, long ld, int cols,
                                                               int* __restrict__ idx_out, float* __restrict__ mean_out) {
    __shared__ int sel[32];
    __shared__ float wv[16];
    __shared__ int wi[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) {
            bool taken = false;
            for (int q = 0; q < r; ++q) taken |= (sel[q] == i);
            if (taken) continue;
            const float v = score[i];
            if (bi == 0x7fffffff || v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = wv[0];
            int ix = wi[0];
            for (int w2 = 1; w2 < 16; ++w2)
                if (wi[w2] != 0x7fffffff && (ix == 0x7fffffff || wv[w2] > v || (wv[w2] == v && wi[w2] < ix))) { v = wv[w2]; ix = wi[w2]; }
            sel[r] = ix;
            idx_out[r] = ix;
        }
        __syncthreads();
    }
    const float inv = 1.0f / (float)k;
    for (int c = tid; c < cols; c += 1024) {
        float acc = 0.f;
        for (int r = 0; r < k; ++r) acc += (float)rows[(long)sel[r] * ld + c];
        mean_out[c] = acc * inv;
    }
}


// torch.nn.functional.normalize(x, dim=-1): x / max(||x||_2, eps), one wave per row (KEEP's encode_image, reference keep.py:45-47)
__global__ void __launch_bounds__(256) l2_normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s = fmaf(xr[c], xr[c], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    for (int c = lane; c < cols; c += 64) out[(long)row * cols + c] = xr[c] * inv;
}

__global__ void __launch_bounds__(256) f16_rows_to_f32_kernel(const f16* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}

}  // namespace amds

using namespace amds;

extern "C" int amds_mean_pool_bwd(const float* dy, float* dx, int B, int T, int F, void* stream) {
    AMDS_REQUIRE(dy && dx && B >= 0 && T > 0 && F > 0, "amds_mean_pool_bwd: bad arguments");
    if (B == 0) return AMDS_OK;
    const long n = (long)B * T * F;
    hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, n, T, F);
    AMDS_LAUNCH_CHECK("mean_pool_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_gather_rows(const void* src, long src_ld, const long* idx, int n_idx, void* dst, long dst_ld, int n_out,
                                int cols, int in_dtype, int out_dtype, void* stream) {
    AMDS_REQUIRE(src && dst && (idx || n_idx == 0), "amds_gather_rows: null pointer");
    AMDS_REQUIRE(n_idx >= 0 && n_out >= n_idx && cols > 0, "amds_gather_rows: bad sizes n_idx=%d n_out=%d cols=%d", n_idx, n_out, cols);
    if (n_out == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(n_out), block(256);
    if (in_dtype == AMDS_F16 && out_dtype == AMDS_F32)
        hipLaunchKernelGGL((gather_rows_kernel<f16, float>), grid, block, 0, st, (const f16*)src, src_ld, idx, n_idx, (float*)dst, dst_ld, n_out, cols);
    else if (in_dtype == AMDS_F16 && out_dtype == AMDS_F16)
        hipLaunchKernelGGL((gather_rows_kernel<f16, f16>), grid, block, 0, st, (const f16*)src, src_ld, idx, n_idx, (f16*)dst, dst_ld, n_out, cols);
    else if (in_dtype == AMDS_F32 && out_dtype == AMDS_F32)
        hipLaunchKernelGGL((gather_rows_kernel<float, float>), grid, block, 0, st, (const float*)src, src_ld, idx, n_idx, (float*)dst, dst_ld, n_out, cols);
    else { set_error("amds_gather_rows: unsupported dtype pair %d -> %d", in_dtype, out_dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("gather_rows_kernel");
    return AMDS_OK;
}

extern "C" int amds_vary_precision(const void* bits_in, const uint8_t* shifts, void* bits_out, long n, int elem_bytes, void* stream) {
    AMDS_REQUIRE(bits_in && shifts && bits_out, "amds_vary_precision: null pointer");
    AMDS_REQUIRE(n >= 0 && (elem_bytes == 2 || elem_bytes == 4), "amds_vary_precision: elem_bytes must be 2 or 4");
    if (n == 0) return AMDS_OK;
    const int grid = (int)min((long)4096, (n + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (elem_bytes == 2) hipLaunchKernelGGL((vary_precision_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, (const uint16_t*)bits_in, shifts, (uint16_t*)bits_out, n);
    else hipLaunchKernelGGL((vary_precision_kernel<uint32_t>), dim3(grid), dim3(256), 0, st, (const uint32_t*)bits_in, shifts, (uint32_t*)bits_out, n);
    AMDS_LAUNCH_CHECK("vary_precision_kernel");
    return AMDS_OK;
}

extern "C" int amds_mean_pool(const void* x, float* out, int B, int T, int F, int in_dtype, void* stream) {
    AMDS_REQUIRE(x && out, "amds_mean_pool: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && F > 0, "amds_mean_pool: bad shape B=%d T=%d F=%d (empty bags have no mean)", B, T, F);
    if (B == 0) return AMDS_OK;
    const dim3 grid(cdiv(F, 256), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == AMDS_F32) hipLaunchKernelGGL((mean_pool_kernel<float>), grid, block, 0, st, (const float*)x, out, T, F);
    else if (in_dtype == AMDS_F16) hipLaunchKernelGGL((mean_pool_kernel<f16>), grid, block, 0, st, (const f16*)x, out, T, F);
    else { set_error("amds_mean_pool: bad dtype %d", in_dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("mean_pool_kernel");
    return AMDS_OK;
}

extern "C" int amds_linear_f32(const float* x, const float* w, const float* bias, float* out, int M, int N, int K, int relu, void* stream) {
    AMDS_REQUIRE(x && w && out, "amds_linear_f32: null pointer");
    AMDS_REQUIRE(M >= 0 && N > 0 && K > 0, "amds_linear_f32: bad shape");
    if (M == 0) return AMDS_OK;
    return gemm_f32(x, K, w, K, bias, out, N, M, N, K, relu, (hipStream_t)stream);
}

extern "C" int amds_topk_rows_mean(const float* score, int n, int k, const void* rows, long ld, int cols, int rows_dtype, int* idx_out, float* mean_out,
                                   void* stream) {
    AMDS_REQUIRE(score && rows && idx_out && mean_out, "amds_topk_rows_mean: null pointer");
    AMDS_REQUIRE(n > 0 && k > 0 && k <= 32 && k <= n && cols > 0 && ld >= cols, "amds_topk_rows_mean: bad sizes n=%d k=%d cols=%d (1 <= k <= min(32, n))", n, k, cols);
    hipStream_t st = (hipStream_t)stream;
    if (rows_dtype == AMDS_F32)
        hipLaunchKernelGGL((topk_rows_mean_kernel<float>), dim3(1), dim3(1024), 0, st, score, n, k, (const float*)rows, ld, cols, idx_out, mean_out);
    else if (rows_dtype == AMDS_F16)
        hipLaunchKernelGGL((topk_rows_mean_kernel<f16>), dim3(1), dim3(1024), 0, st, score, n, k, (const f16*)rows, ld, cols, idx_out, mean_out);
    else { set_error("amds_topk_rows_mean: bad dtype %d", rows_dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("topk_rows_mean_kernel");
    return AMDS_OK;
}

extern "C" size_t amds_proj_head_l2norm_workspace_bytes(int rows, int in_dim, int proj_dim) {
    return (((size_t)rows * in_dim * 4 + 255) & ~(size_t)255) + 2 * (((size_t)rows * proj_dim * 4 + 255) & ~(size_t)255);
}

// KEEP's image head (reference src/stamp/preprocessing/extractor/keep.py:38-47): normalize(Linear(GELU(Linear(feats)))), exact fp32
extern "C" int amds_proj_head_l2norm(const void* feats, int feats_dtype, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int rows,
                                     int in_dim, int proj_dim, void* ws, size_t ws_bytes, void* stream) {
    if (rows == 0) return AMDS_OK;
    AMDS_REQUIRE(feats && w1 && b1 && w2 && b2 && out && ws, "amds_proj_head_l2norm: null pointer");
    AMDS_REQUIRE(rows > 0 && in_dim > 0 && proj_dim > 0 && (feats_dtype == AMDS_F32 || feats_dtype == AMDS_F16), "amds_proj_head_l2norm: bad arguments");
    if (ws_bytes < amds_proj_head_l2norm_workspace_bytes(rows, in_dim, proj_dim)) { set_error("amds_proj_head_l2norm: workspace too small"); return AMDS_ERR_WORKSPACE; }
    AMDS_REQUIRE(((uintptr_t)ws & 255) == 0, "amds_proj_head_l2norm: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* base = reinterpret_cast<char*>(ws);
    const size_t o1 = ((size_t)rows * in_dim * 4 + 255) & ~(size_t)255, o2 = o1 + (((size_t)rows * proj_dim * 4 + 255) & ~(size_t)255);
    const float* x = reinterpret_cast<const float*>(feats);
    if (feats_dtype == AMDS_F16) {
        const long n = (long)rows * in_dim;
        hipLaunchKernelGGL(f16_rows_to_f32_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, (const f16*)feats, (float*)base, n);
        AMDS_LAUNCH_CHECK("f16_rows_to_f32_kernel");
        x = reinterpret_cast<const float*>(base);
    }
    float *h = reinterpret_cast<float*>(base + o1), *y = reinterpret_cast<float*>(base + o2);
    int rc = gemm_f32(x, in_dim, w1, in_dim, b1, h, proj_dim, rows, proj_dim, in_dim, 0, st);
    if (rc != AMDS_OK) return rc;
    if ((rc = amds_mlp_act_f32(h, proj_dim, rows, proj_dim, 0, stream)) != AMDS_OK) return rc;                   // nn.GELU (exact erf)
    if ((rc = gemm_f32(h, proj_dim, w2, proj_dim, b2, y, proj_dim, rows, proj_dim, proj_dim, 0, st)) != AMDS_OK) return rc;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, y, out, rows, proj_dim, 1e-12f);
    AMDS_LAUNCH_CHECK("l2_normalize_rows_kernel");
    return AMDS_OK;
}
