// attention_vit80.hip -- the tile-encoder attention of attention_vit.hip for head_dim 80 (ViT-H/14: Virchow2, dim 1280,
// 16 heads; reference src/stamp/preprocessing/extractor/virchow2.py:34-39).  Same design (whole K and V^T of one
// (tile, head) in LDS, S^T = K Q^T, chunked online softmax, P already in MFMA-operand position); what changes:
//   * 5 k-steps of 16 for Q K^T; K rows are 160 B, stored at a 176-byte stride (11 sixteen-byte slots, odd -> a
//     fragment's 16 rows per lane group hit 16 distinct slots, no XOR needed);
//   * O^T has 80 rows = 2.5 MFMA tiles: V^T is zero-padded to 96 rows and three 32-row tiles are accumulated;
//   * LDS 104 KB -> one workgroup per CU.
#include "common.h"
#include <type_traits>

namespace amds {

constexpr int A80_HD = 80, A80_KRS = 176, A80_VROWS = 96;
__host__ __device__ constexpr int vt80_row_bytes(int nkt) { return (nkt & 1) ? nkt * 64 : nkt * 64 + 64; }

template <typename T, int NKT, int NTH>
__global__ void __launch_bounds__(NTH, 1) attn_vit80_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn, int H, int n_items) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int KP = NKT * 32;
    constexpr int VS = vt80_row_bytes(NKT);
    constexpr int K_BYTES = KP * A80_KRS;
    __shared__ __attribute__((aligned(16))) char smem[K_BYTES + A80_VROWS * VS + 12 * 16];
    char* sK = smem;
    char* sVt = smem + K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int Dm = H * A80_HD;
    const long ld = 3L * Dm;

    constexpr int K_ITEMS = KP * 10, V_ITEMS = (KP / 2) * 10;
    constexpr int K_IT = (K_ITEMS + NTH - 1) / NTH, V_IT = (V_ITEMS + NTH - 1) / NTH, NW = NTH / 64;
    // Persistent workgroups (one per CU: 104 KB of LDS): the K / V rows of item i+1 travel HBM -> registers while item i is computed out of
    // LDS, and are written into LDS behind a barrier -- the one-shot form exposed the whole round trip of every item.
    u32x4 kv[K_IT];
    vec8 v0[V_IT], v1[V_IT];
    auto stage_load = [&](int item) {
    const int b = item / H, h = item - b * H;
    const T* base = qkv + (long)b * Tn * ld + h * A80_HD;
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
        const int c = it * NTH + tid, key = c / 10, ch = c - key * 10;
        kv[it] = u32x4{0u, 0u, 0u, 0u};
        if (c < K_ITEMS && key < Tn) kv[it] = *reinterpret_cast<const u32x4*>(base + (long)key * ld + Dm + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
        const int c = it * NTH + tid, kp2 = c / 10, ch = c - kp2 * 10, k0 = kp2 * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v0[it][e] = (T)0.f; v1[it][e] = (T)0.f; }
        if (c < V_ITEMS && k0 < Tn) v0[it] = *reinterpret_cast<const vec8*>(base + (long)k0 * ld + 2 * Dm + ch * 8);
        if (c < V_ITEMS && k0 + 1 < Tn) v1[it] = *reinterpret_cast<const vec8*>(base + (long)(k0 + 1) * ld + 2 * Dm + ch * 8);
    }
    };
    auto stage_store = [&]() {
#pragma unroll
    for (int it = 0; it < K_IT; ++it) {
        const int c = it * NTH + tid, key = c / 10, ch = c - key * 10;
        if (c < K_ITEMS) *reinterpret_cast<u32x4*>(sK + key * A80_KRS + ch * 16) = kv[it];
    }
#pragma unroll
    for (int it = 0; it < V_IT; ++it) {
        const int c = it * NTH + tid, kp2 = c / 10, ch = c - kp2 * 10, k0 = kp2 * 2;
        if (c < V_ITEMS) {
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
    };
    // zero rows 80..95 of V^T (skew groups 10 and 11): once, the items only ever write rows 0..79
    for (int c = tid; c < 16 * (VS / 16); c += NTH) {
        const int r = 80 + c / (VS / 16), s16 = c % (VS / 16);
        *reinterpret_cast<u32x4*>(sVt + r * VS + (r >> 3) * 16 + s16 * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    int item = blockIdx.x;
    if (item >= n_items) return;
    stage_load(item);
    stage_store();
    __syncthreads();

    const float sc = 0.11180339887498948f * 1.44269504088896340736f;  // 80^-0.5 * log2(e)
    const int nqb = (Tn + 31) >> 5;
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
    const int b = item / H, h = item - b * H;
    const T* base = qkv + (long)b * Tn * ld + h * A80_HD;
    const bool has_next = item + (int)gridDim.x < n_items;
    if (has_next) stage_load(item + gridDim.x);               // in flight during this item's compute

    // 9 query blocks over NW waves (8: two per SIMD, so that one wave's MFMAs run under its SIMD partner's softmax; with 4 waves the
    // wave holding 3 blocks ran them back to back with nothing to overlap): rotate which wave gets the extra block from item to item
    for (int qb = (wave + item) & (NW - 1); qb < nqb; qb += NW) {
        const int q = qb * 32 + l31;
        const int qc = min(q, Tn - 1);
        vec8 qf[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) qf[ks] = *reinterpret_cast<const vec8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);

        f32x16 o[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)
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
                for (int ks = 0; ks < 5; ++ks) {
                    const vec8 kf = *reinterpret_cast<const vec8*>(sK + ((t0 + t) * 32 + l31) * A80_KRS + (ks * 2 + hi) * 16);
                    s[t] = Act<T>::mfma32(kf, qf[ks], s[t]);
                }
            }
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
            const float mnew = fmaxf(mrun, mx * sc);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            mrun = mnew;
            float ls = 0.f;
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], sc, -mnew));
                    s[t][r] = p;
                    ls += p;
                }
            l = l * alpha + ls;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
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
                    for (int dt = 0; dt < 3; ++dt) {
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
        l += __shfl_xor(l, 32, 64);
        if (q < Tn) {
            const float inv = 1.0f / l;
            T* orow = out + ((long)b * Tn + q) * Dm + h * A80_HD;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dcol = dt * 32 + 8 * g + 4 * hi;
                    if (dcol < A80_HD) {
                        vec4 w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = Act<T>::from_f32(o[dt][4 * g + e] * inv);
                        *reinterpret_cast<vec4*>(orow + dcol) = w;
                    }
                }
        }
    }
    __syncthreads();                                          // every wave is done with this item's K / V^T images
    if (has_next) {
        stage_store();
        __syncthreads();
    }
    }
}

template <typename T>
static int launch_attn80(const void* qkv, void* out, int B, int Tn, int H, hipStream_t st) {
    const int nkt = (Tn + 31) / 32;
    static int nth = 0;
    if (!nth) {
        const char* e = getenv("AMDS_ATTN80_THREADS");            // 256 = the four-wave form (A/B)
        nth = (e && atoi(e) == 256) ? 256 : 512;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        AMDS_HIP(hipGetDevice(&dev));
        AMDS_HIP(hipGetDeviceProperties(&p, dev));
        cus = p.multiProcessorCount;
    }
    const int n_items = B * H;
    const dim3 grid(n_items < cus ? n_items : cus);
    switch (nkt) {
#define AMDS_ATT_CASE(N)                                                                                                             \
    case N:                                                                                                                          \
        if (nth == 512) hipLaunchKernelGGL((attn_vit80_kernel<T, N, 512>), grid, dim3(512), 0, st, (const T*)qkv, (T*)out, Tn, H, n_items);   \
        else hipLaunchKernelGGL((attn_vit80_kernel<T, N, 256>), grid, dim3(256), 0, st, (const T*)qkv, (T*)out, Tn, H, n_items);              \
        break;
        AMDS_ATT_CASE(1) AMDS_ATT_CASE(2) AMDS_ATT_CASE(3) AMDS_ATT_CASE(4) AMDS_ATT_CASE(5)
        AMDS_ATT_CASE(6) AMDS_ATT_CASE(7) AMDS_ATT_CASE(8) AMDS_ATT_CASE(9)
#undef AMDS_ATT_CASE
        default:
            set_error("amds_attention_vit_hd: T=%d > 288 unsupported by the LDS-resident kernel", Tn);
            return AMDS_ERR_INVALID;
    }
    AMDS_LAUNCH_CHECK("attn_vit80_kernel");
    return AMDS_OK;
}

}  // namespace amds

using namespace amds;

extern "C" int amds_attention_vit_hd(const void* qkv, void* out, int B, int T, int H, int head_dim, int dtype, void* stream) {
    if (head_dim == 64) return amds_attention_vit(qkv, out, B, T, H, dtype, stream);
    AMDS_REQUIRE(head_dim == 80, "amds_attention_vit_hd: head_dim=%d unsupported (64 or 80)", head_dim);
    AMDS_REQUIRE(qkv && out, "amds_attention_vit_hd: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0, "amds_attention_vit_hd: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, 4.0 * B * H * (double)T * T * 80, st);
    if (dtype == AMDS_F16) return launch_attn80<f16>(qkv, out, B, T, H, st);
    if (dtype == AMDS_BF16) return launch_attn80<bf16>(qkv, out, B, T, H, st);
    set_error("amds_attention_vit_hd: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}
