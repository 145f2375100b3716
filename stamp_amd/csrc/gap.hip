// gap.hip -- gated-attention pooling of a bag of tile features (CHIEF slide encoder): the SIX-LAUNCH form (two exact-fp32 GEMMs through HBM, gate,
// softmax statistics, partial pooling, reduce).  Since round 6 the product path is the single fused launch of gap_fused.hip; this form serves the
// shapes the fused kernel does not take (L not 256 / 512, F or D not a multiple of 16) and stays callable as amds_gated_attn_pool_unfused for A/B.
// Reference: src/stamp/encoding/encoder/chief.py:74-89 (CHIEFModel.forward), :255-275 (Attn_Net_Gated).
//   h   = relu(x Wfc^T + bfc)                 [N, L]
//   A_n = Wc (tanh(Wa h_n + ba) * sigmoid(Wb h_n + bb)) + bc
//   out = softmax_N(A) @ x                    [F]   (pooled over the ORIGINAL features, chief.py:82)
// The reference runs this encoder in fp32 (chief.py:117), so the two GEMMs use the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TF peak) rather than f16/bf16 operands.
// Dropout layers are identity in eval mode.
#include "common.h"

namespace amds {

// C[M,N] = act(A[M,K] W[N,K]^T + bias[N]); fp32 in/out, 64x64 block tile, 4 waves (2x2) of one 32x32
// fragment each, K staged 32 at a time through LDS (row stride 33 floats -> conflict-free b32 reads).
template <int RELU>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       int ldw, const float* __restrict__ bias, float* __restrict__ Cout,
                                                       int ldc, int M, int N, int K) {
    constexpr int BK = 32, LDT = BK + 1;
    __shared__ float sA[64 * LDT];
    __shared__ float sW[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += BK) {
        // 64 rows x 32 cols per operand = 512 float4 -> 2 per thread
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = it * 256 + tid, row = c >> 3, c4 = (c & 7) * 4;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vw = {0.f, 0.f, 0.f, 0.f};
            const int gm = m0 + row, gn = n0 + row;
            if (gm < M) {
                if (k0 + c4 + 3 < K) va = *reinterpret_cast<const f32x4*>(A + (long)gm * lda + k0 + c4);
                else for (int e = 0; e < 4; ++e) if (k0 + c4 + e < K) va[e] = A[(long)gm * lda + k0 + c4 + e];
            }
            if (gn < N) {
                if (k0 + c4 + 3 < K) vw = *reinterpret_cast<const f32x4*>(W + (long)gn * ldw + k0 + c4);
                else for (int e = 0; e < 4; ++e) if (k0 + c4 + e < K) vw[e] = W[(long)gn * ldw + k0 + c4 + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { sA[row * LDT + c4 + e] = va[e]; sW[row * LDT + c4 + e] = vw[e]; }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = sA[(wm * 32 + l31) * LDT + kk + hi];
            const float b = sW[(wn * 32 + l31) * LDT + kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < M) {
                float v = acc[r] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                Cout[(long)m * ldc + n] = v;
            }
        }
    }
}

int gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* Cout, int ldc, int M, int N,
             int K, int relu, hipStream_t st) {
    const dim3 grid(cdiv(N, 64), cdiv(M, 64));
    ProfScope prof(PROF_GEMM_F32, 2.0 * M * (double)N * K, st);
    if (relu) hipLaunchKernelGGL((gemm_f32_kernel<1>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, Cout, ldc, M, N, K);
    else hipLaunchKernelGGL((gemm_f32_kernel<0>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, Cout, ldc, M, N, K);
    AMDS_LAUNCH_CHECK("gemm_f32_kernel");
    return AMDS_OK;
}

// ab: [N][2D] pre-activation (a | b halves, biases already added) -> A_raw[n]; one wave per row
__global__ void __launch_bounds__(256) gap_gate_kernel(const float* __restrict__ ab, const float* __restrict__ cw,
                                                       const float* __restrict__ cb, float* __restrict__ araw, int N, int D) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* row = ab + (long)n * 2 * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float a = tanhf(row[d]);
        const float b = 1.0f / (1.0f + expf(-row[D + d]));
        s += a * b * cw[d];
    }
    s = wave_sum(s);
    if (lane == 0) araw[n] = s + cb[0];
}

// stats[0] = max_n A, stats[1] = sum_n exp(A - max); single block
__global__ void __launch_bounds__(1024) gap_softmax_stats_kernel(const float* __restrict__ araw, float* __restrict__ stats, int N) {
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int i = tid; i < N; i += 1024) m = fmaxf(m, araw[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < N; i += 1024) s += expf(araw[i] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        stats[0] = m;
        stats[1] = t;
    }
}

// partial[c][f] = sum over rows of chunk c of softmax weight * x[n][f]; chunk = 128 rows, block = 256 threads x float4
__global__ void __launch_bounds__(256) gap_pool_partial_kernel(const float* __restrict__ x, const float* __restrict__ araw,
                                                               const float* __restrict__ stats, float* __restrict__ partial,
                                                               int N, int F) {
    const int c = blockIdx.y, f4 = blockIdx.x * 256 + threadIdx.x;
    if (f4 * 4 >= F) return;
    const float m = stats[0], inv = 1.0f / stats[1];
    const int n_beg = c * 128, n_end = min(N, n_beg + 128);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int n = n_beg; n < n_end; ++n) {
        const float w = expf(araw[n] - m) * inv;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (long)n * F + f4 * 4);
        acc[0] += w * v[0]; acc[1] += w * v[1]; acc[2] += w * v[2]; acc[3] += w * v[3];
    }
    *reinterpret_cast<f32x4*>(partial + (long)c * F + f4 * 4) = acc;
}

__global__ void gap_pool_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int nchunks, int F) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += partial[(long)c * F + f];
    out[f] = s;
}

}  // namespace amds

using namespace amds;

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

namespace amds {
bool gap_fused_supported(const float* x, const amds_gap_weights* w, int F, int L, int D);
size_t gap_fused_workspace_bytes(long total_rows, int bags, int F, int L, int D);
int gap_fused_launch(const float* x, const long long* off_dev, int bags, long total_rows, const amds_gap_weights* w, float* out, float* attn_raw, int F,
                     int L, int D, int mode, void* ws, size_t ws_bytes, hipStream_t st);
}  // namespace amds

extern "C" size_t amds_gated_attn_pool_unfused_workspace_bytes(int N, int F, int L, int D) {
    if (N <= 0 || F <= 0 || L <= 0 || D <= 0) return 0;
    const size_t nch = (size_t)(N + 127) / 128;
    return al256((size_t)N * L * 4) + al256((size_t)N * 2 * D * 4) + al256((size_t)2 * D * L * 4) + al256((size_t)2 * D * 4) +
           al256((size_t)N * 4) + al256(64) + al256(nch * F * 4);
}

// enough for either form: the caller sizes the workspace before the weights' alignment is known
extern "C" size_t amds_gated_attn_pool_workspace_bytes(int N, int F, int L, int D) {
    if (N <= 0 || F <= 0 || L <= 0 || D <= 0) return 0;
    if (amds_gated_attn_pool_batched_supported(F, L, D)) {
        const size_t a = gap_fused_workspace_bytes(N, 1, F, L, D);
        return a;       // misaligned pointers are rejected there, not silently rerouted (torch / hipMalloc allocations are 256-byte aligned)
    }
    return amds_gated_attn_pool_unfused_workspace_bytes(N, F, L, D);
}

extern "C" int amds_gated_attn_pool(const float* x, const amds_gap_weights* w, float* out, float* attn_raw, int N, int F,
                                    int L, int D, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(x && w && out && ws, "amds_gated_attn_pool: null pointer");
    AMDS_REQUIRE(N > 0 && F > 0 && L > 0 && D > 0, "amds_gated_attn_pool: empty bag or bad dims (N=%d F=%d L=%d D=%d)", N, F, L, D);
    AMDS_REQUIRE(w->fc_w && w->fc_b && w->a_w && w->a_b && w->b_w && w->b_b && w->c_w && w->c_b, "amds_gated_attn_pool: incomplete weights");
    if (amds_gated_attn_pool_batched_supported(F, L, D)) {
        AMDS_REQUIRE(gap_fused_supported(x, w, F, L, D), "amds_gated_attn_pool: x and the weight arrays must be 16-byte aligned");
        return gap_fused_launch(x, nullptr, 1, N, w, out, attn_raw, F, L, D, AMDS_GAP_AUTO, ws, ws_bytes, (hipStream_t)stream);
    }
    return amds_gated_attn_pool_unfused(x, w, out, attn_raw, N, F, L, D, ws, ws_bytes, stream);
}

extern "C" int amds_gated_attn_pool_unfused(const float* x, const amds_gap_weights* w, float* out, float* attn_raw, int N, int F,
                                            int L, int D, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(x && w && out && ws, "amds_gated_attn_pool: null pointer");
    AMDS_REQUIRE(N > 0 && F > 0 && L > 0 && D > 0, "amds_gated_attn_pool: empty bag or bad dims (N=%d F=%d L=%d D=%d)", N, F, L, D);
    AMDS_REQUIRE(F % 4 == 0 && L % 4 == 0, "amds_gated_attn_pool: F and L must be multiples of 4");
    AMDS_REQUIRE(w->fc_w && w->fc_b && w->a_w && w->a_b && w->b_w && w->b_b && w->c_w && w->c_b, "amds_gated_attn_pool: incomplete weights");
    const size_t need = amds_gated_attn_pool_unfused_workspace_bytes(N, F, L, D);
    if (ws_bytes < need) {
        set_error("amds_gated_attn_pool: workspace %zu < required %zu bytes", ws_bytes, need);
        return AMDS_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* p = reinterpret_cast<char*>(ws);
    float* h = reinterpret_cast<float*>(p);      p += al256((size_t)N * L * 4);
    float* ab = reinterpret_cast<float*>(p);     p += al256((size_t)N * 2 * D * 4);
    float* wab = reinterpret_cast<float*>(p);    p += al256((size_t)2 * D * L * 4);
    float* bab = reinterpret_cast<float*>(p);    p += al256((size_t)2 * D * 4);
    float* araw = reinterpret_cast<float*>(p);   p += al256((size_t)N * 4);
    float* stats = reinterpret_cast<float*>(p);  p += al256(64);
    float* partial = reinterpret_cast<float*>(p);
    // stack [Wa; Wb] so both gate branches are one GEMM
    AMDS_HIP(hipMemcpyAsync(wab, w->a_w, (size_t)D * L * 4, hipMemcpyDeviceToDevice, st));
    AMDS_HIP(hipMemcpyAsync(wab + (size_t)D * L, w->b_w, (size_t)D * L * 4, hipMemcpyDeviceToDevice, st));
    AMDS_HIP(hipMemcpyAsync(bab, w->a_b, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
    AMDS_HIP(hipMemcpyAsync(bab + D, w->b_b, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
    int rc;
    if ((rc = gemm_f32(x, F, w->fc_w, F, w->fc_b, h, L, N, L, F, 1, st)) != AMDS_OK) return rc;
    if ((rc = gemm_f32(h, L, wab, L, bab, ab, 2 * D, N, 2 * D, L, 0, st)) != AMDS_OK) return rc;
    hipLaunchKernelGGL(gap_gate_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, ab, w->c_w, w->c_b, araw, N, D);
    AMDS_LAUNCH_CHECK("gap_gate_kernel");
    hipLaunchKernelGGL(gap_softmax_stats_kernel, dim3(1), dim3(1024), 0, st, araw, stats, N);
    AMDS_LAUNCH_CHECK("gap_softmax_stats_kernel");
    const int nch = cdiv(N, 128);
    hipLaunchKernelGGL(gap_pool_partial_kernel, dim3(cdiv(F / 4, 256), nch), dim3(256), 0, st, x, araw, stats, partial, N, F);
    AMDS_LAUNCH_CHECK("gap_pool_partial_kernel");
    hipLaunchKernelGGL(gap_pool_reduce_kernel, dim3(cdiv(F, 256)), dim3(256), 0, st, partial, out, nch, F);
    AMDS_LAUNCH_CHECK("gap_pool_reduce_kernel");
    if (attn_raw) AMDS_HIP(hipMemcpyAsync(attn_raw, araw, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
    return AMDS_OK;
}
