// attention_vit257.hip -- the tile encoder's attention for the shape every "class token + 16 x 16 patches" model has (T = 257 tokens,
// head_dim 64): persistent workgroups, DOUBLE-BUFFERED K / V^T images in LDS, the next item staged while the current one is computed.
//
// attention_vit.hip's one-shot kernel (one workgroup per (tile, head), 2 per CU) spent ~26 k cycles per item and CU slot against 12.9 k for
// its HBM bytes (132 KB per item at the CU's share of the achievable bandwidth) and ~8 k of MFMA + softmax work: every item exposed one
// HBM round trip for the K / V staging loads and one per query block for the Q fragments, ~2 us each under load, and two workgroups per
// CU do not cover that.  Here ONE 512-thread workgroup per CU (two waves per SIMD: one wave's MFMAs beside the other's softmax VALU)
// walks items blockIdx.x, blockIdx.x + gridDim.x, ...; while item i is computed out of LDS buffer i & 1 the K / V rows of item i + 1 and
// this wave's Q fragments travel HBM -> registers (32 + 16 VGPRs in flight for a whole item's compute time) and are written into buffer
// (i + 1) & 1 at the end: one barrier per item, no exposed round trip.  8 query blocks of 32 on 8 waves: one block per wave (the one-shot
// kernel had 2 per wave on 4 waves), the odd key as a rank-1 VALU update, the odd query as a 257-key GEMV (lane = key, then lane = dim).
// Arithmetic, LDS images and the no-shuffle MFMA operand layout are those of attention_vit.hip; results are bit-identical to it.
#include "common.h"
#include <type_traits>

namespace amds {

constexpr int A7_NKT = 8, A7_KP = 256, A7_VS = 576;                    // vt_row_bytes(8): 9 x 64 B
constexpr int A7_K_BYTES = A7_KP * 128, A7_V_BYTES = 64 * A7_VS + 8 * 16, A7_T_BYTES = 3 * 64 * 4;
constexpr int A7_BUF = A7_K_BYTES + A7_V_BYTES + A7_T_BYTES;          // 70 528 B per item image
constexpr int A7_SCRATCH = (A7_KP + 16 + 8 * 64) * 4;
constexpr int A7_LDS = 2 * A7_BUF + A7_SCRATCH;                       // 144 192 B

// phase timeline for tools/ubench/attn257_trace.hip (compiled with -DA7_TRACE only): s_memtime of every wave of workgroup 0 at the phase
// boundaries of its first items
#ifdef A7_TRACE
__device__ unsigned long long a7_trace[32 * 8 * 8];
#define A7_MARK(k)                                                                                            \
    do {                                                                                                      \
        if (blockIdx.x == 0 && lane == 0 && trace_it < 32) a7_trace[(trace_it * 8 + (k)) * 8 + wave] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define A7_MARK(k) do { } while (0)
#endif

template <typename T>
__global__ void __launch_bounds__(512, 2) attn_vit257_kernel(const T* __restrict__ qkv, T* __restrict__ out, int H, int n_items) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int Tn = 257, KP = A7_KP, VS = A7_VS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sP = reinterpret_cast<float*>(smem + 2 * A7_BUF);         // softmax weights of the odd query [KP] | red [16] | part [8][64]
    float* sRed = sP + KP;
    float* sPart = sRed + 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const float sc = 0.125f * 1.44269504088896340736f;                // 1/sqrt(64) * log2(e)
    const int swz = (l31 >> 1) & 7;

    // ---- staging registers of ONE item: 4 K chunks, 2 V key pairs, the odd token, this wave's Q fragments ----
    u32x4 kv[4];
    vec8 v0[2], v1[2];
    T tq = (T)0.f, tk = (T)0.f, tv = (T)0.f;     // raw 16-bit values: converting here would put an s_waitcnt right behind the loads
    vec8 qn[4];
    auto stage_load = [&](int item) {                                 // issues every global load of an item, waits for nothing
        const int b = item / H, h = item - b * H;
        const T* base = qkv + (long)b * Tn * ld + h * 64;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int c = it * 512 + tid, key = c >> 3, ch = c & 7;
            kv[it] = *reinterpret_cast<const u32x4*>(base + (long)key * ld + Dm + ch * 8);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = it * 512 + tid, k0 = (c >> 3) * 2, ch = c & 7;
            v0[it] = *reinterpret_cast<const vec8*>(base + (long)k0 * ld + 2 * Dm + ch * 8);
            v1[it] = *reinterpret_cast<const vec8*>(base + (long)(k0 + 1) * ld + 2 * Dm + ch * 8);
        }
        if (tid < 64) {
            const T* tr = base + (long)KP * ld + tid;                 // token 256 = the odd one
            tq = tr[0];
            tk = tr[Dm];
            tv = tr[2 * Dm];
        }
        const int q = wave * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qn[ks] = *reinterpret_cast<const vec8*>(base + (long)q * ld + (ks * 2 + hi) * 8);
    };
    auto stage_store = [&](char* buf) {
        char* sK = buf;
        char* sVt = buf + A7_K_BYTES;
        float* sT = reinterpret_cast<float*>(buf + A7_K_BYTES + A7_V_BYTES);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int c = it * 512 + tid, key = c >> 3, ch = c & 7;
            *reinterpret_cast<u32x4*>(sK + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kv[it];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = it * 512 + tid, k0 = (c >> 3) * 2, ch = c & 7;
            const int pos = (k0 & ~12) | ((k0 & 4) << 1) | ((k0 & 8) >> 1);       // key order inside 16-groups: bits 2 <-> 3
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                typedef T vec2 __attribute__((ext_vector_type(2)));
                vec2 w;
                w[0] = v0[it][e]; w[1] = v1[it][e];
                *reinterpret_cast<vec2*>(sVt + (ch * 8 + e) * VS + ch * 16 + pos * 2) = w;
            }
        }
        if (tid < 64) { sT[tid] = Act<T>::to_f32(tk); sT[64 + tid] = Act<T>::to_f32(tv); sT[128 + tid] = Act<T>::to_f32(tq); }    // tail key | value | query
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    stage_load(item);
    stage_store(smem);
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    __syncthreads();

    int cur = 0;
#ifdef A7_TRACE
    int trace_it = 0;
#endif
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        const char* sK = smem + cur * A7_BUF;
        const char* sVt = sK + A7_K_BYTES;
        const float* sKt = reinterpret_cast<const float*>(sK + A7_K_BYTES + A7_V_BYTES);
        const float* sVl = sKt + 64;
        const float* sQt = sKt + 128;
        const bool has_next = item + (int)gridDim.x < n_items;
        A7_MARK(0);
        if (has_next) stage_load(item + gridDim.x);                   // in flight during everything below
        A7_MARK(1);

        // ---- this wave's 32 queries: online softmax over 4 chunks of 2 key tiles ----
        {
            f32x16 o[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            float mrun = -INFINITY, l = 0.f;
#pragma unroll 1
            for (int c = 0; c < A7_NKT / 2; ++c) {
                f32x16 s[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const vec8 kf = *reinterpret_cast<const vec8*>(sK + ((c * 2 + t) * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
                        s[t] = Act<T>::mfma32(kf, qf[ks], s[t]);
                    }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrun, mx * sc);
                const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                mrun = mnew;
                float ls = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], sc, -mnew));
                        s[t][r] = p;
                        ls += p;
                    }
                l = l * alpha + ls;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        vec8 pf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[e] = Act<T>::from_f32(s[t][ks * 8 + e]);
                        const int pos = (c * 2 + t) * 32 + ks * 16 + hi * 8;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const int d = dt * 32 + l31;
                            const vec8 vf = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
                            o[dt] = Act<T>::mfma32(vf, pf, o[dt]);
                        }
                    }
            }
            A7_MARK(2);
            {   // the odd key: rank-1 update of this block's 32 queries (dot product split over the lane pair)
                float dot = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f32x4 k0 = *reinterpret_cast<const f32x4*>(sKt + (ks * 2 + hi) * 8), k1 = *reinterpret_cast<const f32x4*>(sKt + (ks * 2 + hi) * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dot = fmaf(Act<T>::to_f32(qf[ks][e]), k0[e], fmaf(Act<T>::to_f32(qf[ks][4 + e]), k1[e], dot));
                }
                dot += __shfl_xor(dot, 32, 64);
                const float st = dot * sc, mnew = fmaxf(mrun, st);
                const float alpha = __builtin_amdgcn_exp2f(mrun - mnew), pt = __builtin_amdgcn_exp2f(st - mnew);
                mrun = mnew;
                l = l * alpha + (hi == 0 ? pt : 0.f);                 // l is a per-lane partial: count the key once per query
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 vv = *reinterpret_cast<const f32x4*>(sVl + dt * 32 + 8 * g + 4 * hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[dt][4 * g + e] = fmaf(o[dt][4 * g + e], alpha, pt * vv[e]);
                    }
            }
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.0f / l;
            T* orow = out + ((long)b * Tn + wave * 32 + l31) * Dm + h * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    vec4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = Act<T>::from_f32(o[dt][4 * g + e] * inv);
                    *reinterpret_cast<vec4*>(orow + dt * 32 + 8 * g + 4 * hi) = w;
                }
        }

        A7_MARK(3);
        // ---- the odd query against all 257 keys: scores with lane = key (threads 0..255), output with lane = dim (8 waves x 32 keys) ----
        {
            const int key = tid;
            float sv = -INFINITY;
            if (key < KP) {
                float dot = 0.f;
#pragma unroll 2
                for (int ch = 0; ch < 8; ++ch) {
                    const vec8 kk = *reinterpret_cast<const vec8*>(sK + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4));
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(sQt + ch * 8), q1 = *reinterpret_cast<const f32x4*>(sQt + ch * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dot = fmaf(Act<T>::to_f32(kk[e]), q0[e], fmaf(Act<T>::to_f32(kk[4 + e]), q1[e], dot));
                }
                sv = dot * sc;
            }
            float st = 0.f;
#pragma unroll 4
            for (int d4 = 0; d4 < 16; ++d4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sQt + d4 * 4), bq = *reinterpret_cast<const f32x4*>(sKt + d4 * 4);
                st += (a[0] * bq[0] + a[1] * bq[1]) + (a[2] * bq[2] + a[3] * bq[3]);
            }
            st *= sc;
            const float wm = wave_max(sv);
            if (lane == 0) sRed[wave] = wm;
            __syncthreads();
            const float m = fmaxf(fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3])), st);       // waves 4..7 hold -inf
            const float pk = key < KP ? __builtin_amdgcn_exp2f(sv - m) : 0.f;
            if (key < KP) sP[(key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1)] = pk;                 // the V^T image's key order
            const float ws = wave_sum(pk);
            if (lane == 0) sRed[8 + wave] = ws;
            __syncthreads();
            const float ptl = __builtin_amdgcn_exp2f(st - m);
            const float ltot = ((sRed[8] + sRed[9]) + (sRed[10] + sRed[11])) + ptl;
            const int d = lane;
            float acc = 0.f;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                const int pos0 = wave * 32 + c8 * 8;
                const vec8 vv = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos0 * 2);
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(sP + pos0), p1 = *reinterpret_cast<const f32x4*>(sP + pos0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fmaf(Act<T>::to_f32(vv[e]), p0[e], fmaf(Act<T>::to_f32(vv[4 + e]), p1[e], acc));
            }
            sPart[wave * 64 + d] = acc;
            __syncthreads();
            if (wave == 0) {
                float ov = ptl * sVl[d];
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) ov += sPart[w8 * 64 + d];
                out[((long)b * Tn + KP) * Dm + h * 64 + d] = Act<T>::from_f32(ov / ltot);
            }
        }

        A7_MARK(4);
        // ---- the next item: registers -> the other LDS buffer (its loads have had this whole item to arrive) ----
        if (has_next) {
            stage_store(smem + (cur ^ 1) * A7_BUF);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
        }
        A7_MARK(5);
        __syncthreads();
        A7_MARK(6);
        cur ^= 1;
#ifdef A7_TRACE
        ++trace_it;
#endif
    }
}

static int g_a7_cus = 0;

template <typename T>
static int launch_attn257(const void* qkv, void* out, int B, int H, hipStream_t st) {
    auto kern = attn_vit257_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, A7_LDS));
        attr_set = true;
    }
    if (!g_a7_cus) {
        int dev = 0;
        hipDeviceProp_t p;
        AMDS_HIP(hipGetDevice(&dev));
        AMDS_HIP(hipGetDeviceProperties(&p, dev));
        g_a7_cus = p.multiProcessorCount;
    }
    const int n_items = B * H;
    hipLaunchKernelGGL(kern, dim3(min(n_items, g_a7_cus)), dim3(512), A7_LDS, st, (const T*)qkv, (T*)out, H, n_items);
    AMDS_LAUNCH_CHECK("attn_vit257_kernel");
    return AMDS_OK;
}

// called by amds_attention_vit for T = 257 (attention_vit.hip); dtype already validated
int attention_vit257(const void* qkv, void* out, int B, int H, int dtype, hipStream_t st) {
    return dtype == AMDS_F16 ? launch_attn257<f16>(qkv, out, B, H, st) : launch_attn257<bf16>(qkv, out, B, H, st);
}

}  // namespace amds
