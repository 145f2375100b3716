// attention_vit.hip -- fused multi-head self-attention for the tile encoder (T <= 288 tokens, head_dim 64).
//
// One workgroup per (tile, head).  The head's whole K (<= 288 x 64) and V^T (64 x <= 288) live in LDS
// (73 KiB -> 2 workgroups / CU), so K/V are read from HBM exactly once; scores are produced 96 keys at a time and
// folded with an online softmax (running max / sum per query lane):
//   S^T tile = K_tile(32 keys x 64) . Q^T(64 x 32 queries)    -- MFMA 32x32x16, K rows as the "A" operand
//   lane (q = lane&31, hi = lane>>5) then holds, for ITS query, keys 8*(r>>2)+4*hi+(r&3) of every tile:
//   max / exp2 / sum are per-lane loops plus one cross-half exchange;
//   O^T(64 x 32) += V^T(64 x 16 keys) . P^T(16 keys x 32)     -- P is already the "B" operand in-lane.
// The only data movement the MFMA layouts force is the V transpose; it is done once per head while
// staging (packed key-pair ds_write_b32 into a skewed image, see vt_row_bytes), and the key
// order inside each 16-key group is permuted (bits 2<->3) in the LDS image so that a lane's 8 P values of
// a K=16 step are exactly its accumulator registers 8*ks .. 8*ks+7 -- no cross-lane shuffles at all.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace amds {

constexpr int ATT_D = 64;
// V^T image: row d starts at byte d*VS + (d>>3)*16 with VS/16 = 4 (mod 8) 16-byte slots.  With that
// stride + per-8-row shift the ds_read_b128 of a fragment (16 rows per lane group) hits 16 distinct slots
// and the transposing ds_write_b32 of staging is at most 2-way (free) -- found by exhaustive search.
__host__ __device__ constexpr int vt_row_bytes(int nkt) { return (nkt & 1) ? nkt * 64 : nkt * 64 + 64; }

// TAIL: T = 32 NKT + 1 -- the shape of every "class token + 16x16 patches" encoder (257).  Rounding 257 up to 9 query blocks and
// 9 key tiles costs 12 % + 12 % padded MFMA work and, worse, 9 query blocks over 4 waves leave one wave with 3 of them (the
// workgroup's critical path: 27 tile pairs where 16 would do).  With TAIL the MFMA path covers exactly 32 NKT queries x 32 NKT
// keys (2 query blocks per wave), the odd KEY is folded into every query's online softmax as a rank-1 VALU update, and the odd
// QUERY is a 257-key GEMV split over the four waves (lane = key for the scores, lane = dim for the output).
template <typename T, int NKT, bool TAIL = false>  // NKT = key tiles of 32 on the MFMA path
__global__ void __launch_bounds__(256, 2) attn_vit_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn, int H) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int KP = NKT * 32;
    constexpr int VS = vt_row_bytes(NKT);
    constexpr int MAIN = KP * 128 + ATT_D * VS + 8 * 16;
    constexpr int XTRA = TAIL ? (3 * 64 + KP + 8 + 4 * 64) * 4 : 0;
    __shared__ __attribute__((aligned(16))) char smem[MAIN + XTRA];
    char* sK = smem;
    char* sVt = smem + KP * 128;
    float* sKt = reinterpret_cast<float*>(smem + MAIN);     // tail key [64] | tail value [64] | tail query [64] | P [KP] | red [8] | part [4][64]
    float* sVl = sKt + 64;
    float* sQt = sVl + 64;
    float* sP = sQt + 64;
    float* sRed = sP + KP;
    float* sPart = sRed + 8;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int Dm = H * ATT_D;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * ATT_D;

    // ---- stage K and V^T: ALL global loads are issued first (one L2/HBM round trip for the whole head instead of
    //      one per loop iteration), then the LDS writes ----------------------------------------------------------
    constexpr int K_IT = NKT;                 // KP*8 chunks / 256 threads
    constexpr int V_IT = (NKT + 1) / 2;       // (KP/2)*8 key-pair items / 256 threads
    u32x4 kv[K_IT];
    vec8 v0[V_IT], v1[V_IT];
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
        const int c = it * 256 + tid, key = c >> 3, ch = c & 7;
        kv[it] = u32x4{0u, 0u, 0u, 0u};
        if (key < (TAIL ? KP : Tn)) kv[it] = *reinterpret_cast<const u32x4*>(base + (long)key * ld + Dm + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
        const int c = it * 256 + tid, k0 = (c >> 3) * 2, ch = c & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v0[it][e] = (T)0.f; v1[it][e] = (T)0.f; }
        if (k0 < (TAIL ? KP : Tn)) v0[it] = *reinterpret_cast<const vec8*>(base + (long)k0 * ld + 2 * Dm + ch * 8);
        if (k0 + 1 < (TAIL ? KP : Tn)) v1[it] = *reinterpret_cast<const vec8*>(base + (long)(k0 + 1) * ld + 2 * Dm + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
        const int c = it * 256 + tid, key = c >> 3, ch = c & 7;
        *reinterpret_cast<u32x4*>(sK + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kv[it];
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
        const int c = it * 256 + tid, k0 = (c >> 3) * 2, ch = c & 7;
        if (k0 < KP) {
            // position of key k0 inside the permuted image: swap bits 2 and 3 of the key index
            const int pos = (k0 & ~12) | ((k0 & 4) << 1) | ((k0 & 8) >> 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                typedef T vec2 __attribute__((ext_vector_type(2)));
                vec2 w;
                w[0] = v0[it][e]; w[1] = v1[it][e];
                *reinterpret_cast<vec2*>(sVt + (ch * 8 + e) * VS + ch * 16 + pos * 2) = w;
            }
        }
    }
    if constexpr (TAIL) {
        if (tid < 64) {
            const T* tr = base + (long)KP * ld + tid;                  // token KP = the odd one
            sQt[tid] = Act<T>::to_f32(tr[0]);
            sKt[tid] = Act<T>::to_f32(tr[Dm]);
            sVl[tid] = Act<T>::to_f32(tr[2 * Dm]);
        }
    }
    __syncthreads();

    const float sc = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)
    const int swz = (l31 >> 1) & 7;
    const int nqb = TAIL ? NKT : (Tn + 31) >> 5;

    // 9 query blocks over 4 waves leaves one wave with 3: rotate which wave (= which SIMD) that is from block to block, so
    // the two (or more) workgroups sharing a CU do not pile their heavy waves on the same SIMD
    for (int qb = (wave + blockIdx.x) & 3; qb < nqb; qb += 4) {
        const int q = qb * 32 + l31;
        const int qc = min(q, Tn - 1);
        vec8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const vec8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);

        // ---- online softmax over chunks of CH key tiles (keeps the live score registers at CH*16) ------
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float mrun = -INFINITY, l = 0.f;
        auto chunk = [&](auto nt_tag, const int t0) {
            constexpr int NTC = decltype(nt_tag)::value;
            f32x16 s[NTC];
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const vec8 kf = *reinterpret_cast<const vec8*>(sK + ((t0 + t) * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
                    s[t] = Act<T>::mfma32(kf, qf[ks], s[t]);
                }
            }
            // raw (unscaled) scores: only a ragged last tile needs masking -- one wave-uniform test per TILE, not per element
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
                if ((t0 + t + 1) * 32 > Tn) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = (t0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (key >= Tn) s[t][r] = -INFINITY;
                    }
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrun, mx * sc);            // running max of the SCALED (log2-domain) scores
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);   // first chunk: exp2(-inf) = 0
            mrun = mnew;
            float ls = 0.f;
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], sc, -mnew));   // v_exp_f32; exp2(-inf) = 0
                    s[t][r] = p;
                    ls += p;
                }
            l = l * alpha + ls;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    vec8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = Act<T>::from_f32(s[t][ks * 8 + e]);
                    const int pos = (t0 + t) * 32 + ks * 16 + hi * 8;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int d = dt * 32 + l31;
                        const vec8 vf = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
                        o[dt] = Act<T>::mfma32(vf, pf, o[dt]);
                    }
                }
        };
        constexpr int CH = 3;
#pragma unroll 1
        for (int c = 0; c < NKT / CH; ++c) chunk(std::integral_constant<int, CH>{}, c * CH);
        if constexpr (NKT % CH != 0) chunk(std::integral_constant<int, (NKT % CH == 0 ? 1 : NKT % CH)>{}, (NKT / CH) * CH);
        if constexpr (TAIL) {      // the odd key: rank-1 update of this block's 32 queries (dot product split over the lane pair)
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
            l = l * alpha + (hi == 0 ? pt : 0.f);                     // l is a per-lane partial: count the key once per query
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
        if (q < Tn) {
            const float inv = 1.0f / l;
            T* orow = out + ((long)b * Tn + q) * Dm + h * ATT_D;
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
    }
    if constexpr (TAIL) {
        // ---- the odd query against all KP + 1 keys: scores with lane = key (wave w owns keys 64 w ..), output with lane = dim ----
        const int key = tid;
        float sv = -INFINITY;
        if (key < KP) {
            float dot = 0.f;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const vec8 kk = *reinterpret_cast<const vec8*>(sK + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4));
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(sQt + ch * 8), q1 = *reinterpret_cast<const f32x4*>(sQt + ch * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot = fmaf(Act<T>::to_f32(kk[e]), q0[e], fmaf(Act<T>::to_f32(kk[4 + e]), q1[e], dot));
            }
            sv = dot * sc;
        }
        float st = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sQt + d4 * 4), bq = *reinterpret_cast<const f32x4*>(sKt + d4 * 4);
            st += (a[0] * bq[0] + a[1] * bq[1]) + (a[2] * bq[2] + a[3] * bq[3]);
        }
        st *= sc;
        const float wm = wave_max(sv);
        if (lane == 0) sRed[wave] = wm;
        __syncthreads();
        const float m = fmaxf(fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3])), st);
        const float pk = key < KP ? __builtin_amdgcn_exp2f(sv - m) : 0.f;
        if (key < KP) sP[(key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1)] = pk;       // the V^T image's key order
        const float ws = wave_sum(pk);
        if (lane == 0) sRed[4 + wave] = ws;
        __syncthreads();
        const float ptl = __builtin_amdgcn_exp2f(st - m);
        const float ltot = ((sRed[4] + sRed[5]) + (sRed[6] + sRed[7])) + ptl;
        const int d = lane;
        float acc = 0.f;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const int pos0 = wave * 64 + c8 * 8;
            if (pos0 < KP) {
                const vec8 vv = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos0 * 2);
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(sP + pos0), p1 = *reinterpret_cast<const f32x4*>(sP + pos0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fmaf(Act<T>::to_f32(vv[e]), p0[e], fmaf(Act<T>::to_f32(vv[4 + e]), p1[e], acc));
            }
        }
        sPart[wave * 64 + d] = acc;
        __syncthreads();
        if (wave == 0) {
            const float ov = ((sPart[d] + sPart[64 + d]) + (sPart[128 + d] + sPart[192 + d])) + ptl * sVl[d];
            out[((long)b * Tn + KP) * Dm + h * ATT_D + d] = Act<T>::from_f32(ov / ltot);
        }
    }
}

template <typename T>
static int launch_attn(const void* qkv, void* out, int B, int Tn, int H, hipStream_t st) {
    const dim3 grid(B * H), block(256);
    static const bool tail_on = getenv("AMDS_ATTN_TAIL") ? atoi(getenv("AMDS_ATTN_TAIL")) != 0 : true;
    if (tail_on && Tn % 32 == 1 && Tn > 32) {      // class token + a multiple of 32 patches: MFMA on the 32-multiple, VALU for the odd token
        switch (Tn / 32) {
#define AMDS_ATT_TAIL(N) \
    case N: hipLaunchKernelGGL((attn_vit_kernel<T, N, true>), grid, block, 0, st, (const T*)qkv, (T*)out, Tn, H); break;
            AMDS_ATT_TAIL(1) AMDS_ATT_TAIL(2) AMDS_ATT_TAIL(3) AMDS_ATT_TAIL(4) AMDS_ATT_TAIL(5) AMDS_ATT_TAIL(6) AMDS_ATT_TAIL(7) AMDS_ATT_TAIL(8)
#undef AMDS_ATT_TAIL
        }
        AMDS_LAUNCH_CHECK("attn_vit_kernel<tail>");
        return AMDS_OK;
    }
    const int nkt = (Tn + 31) / 32;
    switch (nkt) {
#define AMDS_ATT_CASE(N) \
    case N: hipLaunchKernelGGL((attn_vit_kernel<T, N>), grid, block, 0, st, (const T*)qkv, (T*)out, Tn, H); break;
        AMDS_ATT_CASE(1) AMDS_ATT_CASE(2) AMDS_ATT_CASE(3) AMDS_ATT_CASE(4) AMDS_ATT_CASE(5)
        AMDS_ATT_CASE(6) AMDS_ATT_CASE(7) AMDS_ATT_CASE(8) AMDS_ATT_CASE(9)
#undef AMDS_ATT_CASE
        default:
            set_error("amds_attention_vit: T=%d > 288 unsupported by the LDS-resident kernel", Tn);
            return AMDS_ERR_INVALID;
    }
    AMDS_LAUNCH_CHECK("attn_vit_kernel");
    return AMDS_OK;
}

int attention_vit257(const void* qkv, void* out, int B, int H, int dtype, hipStream_t st);       // attention_vit257.hip
int attention_vit26x(const void* qkv, void* out, int B, int T, int H, int dtype, hipStream_t st);      // attention_vit26x.hip

}  // namespace amds

using namespace amds;

extern "C" int amds_attention_vit(const void* qkv, void* out, int B, int T, int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out, "amds_attention_vit: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0, "amds_attention_vit: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, 4.0 * B * H * (double)T * T * 64, st);
    // T = 257 (class token + 16 x 16 patches): the persistent double-buffered kernel; AMDS_ATTN_257=0 selects the one-shot kernel (A/B)
    static const bool use257 = getenv("AMDS_ATTN_257") ? atoi(getenv("AMDS_ATTN_257")) != 0 : true;
    if (use257 && T == 257 && (dtype == AMDS_F16 || dtype == AMDS_BF16)) return attention_vit257(qkv, out, B, H, dtype, st);
    // T = 261 / 265 (class + 4 / 8 register tokens + 16 x 16 patches: Virchow2-like trunks at head_dim 64, UNI2-h, H-optimus): the same pipeline with
    // the tail tokens as a ninth key tile and an R-row query operand (attention_vit26x.hip); AMDS_ATTN_26X=0 selects the one-shot kernel (A/B)
    static const bool use26x = getenv("AMDS_ATTN_26X") ? atoi(getenv("AMDS_ATTN_26X")) != 0 : true;
    if (use26x && (T == 261 || T == 265) && (dtype == AMDS_F16 || dtype == AMDS_BF16)) return attention_vit26x(qkv, out, B, T, H, dtype, st);
    if (dtype == AMDS_F16) return launch_attn<f16>(qkv, out, B, T, H, st);
    if (dtype == AMDS_BF16) return launch_attn<bf16>(qkv, out, B, T, H, st);
    set_error("amds_attention_vit: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}
