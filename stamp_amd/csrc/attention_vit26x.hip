// attention_vit26x.hip -- the persistent double-buffered attention of attention_vit257.hip for the token counts of the register-token models:
// T = 256 + R with R in 4 .. 9 (Virchow2: class + 4 registers = 261; UNI2-h / H-optimus: class + 8 registers = 265), head_dim 64.
//
// Same structure for the 256 "main" tokens: ONE 512-thread workgroup per CU walks items (tile, head); K / V^T images of the next item are staged
// HBM -> registers -> the idle LDS image while the current one is computed; wave w owns queries 32 w .. 32 w + 31 and software-pipelines
// QK^T / softmax / P V over four chunks of 64 keys.  What differs is the TAIL (tokens 256 .. 255 + R), for which attention_vit257.hip has a
// rank-1 VALU update (one key) and a 1-row MFMA operand merged from 9 partials (one query):
//   * tail KEYS are a ninth key tile: rows 256 .. 271 of the K image (rows >= 256 + R stay zero) and the ninth 64-byte column of the V^T image's rows --
//     the column that is bank-conflict padding for eight tiles holds exactly one more tile.  A main query block handles them after its four chunks
//     as 4 + 2 MFMAs (lanes 16 .. 31 of the K operand read zeros; scores of keys >= R are masked to -inf) and one more online-softmax step.
//   * tail QUERIES: an R-row MFMA operand (lane = query row, rows >= R read as 0 through the buffer descriptor's range check) against key tiles
//     {w, w + 4} on waves w = 0 .. 3 -- one wave per SIMD; wave 0 also takes the tail tile -- scores C[query][key] with lane = key, so the row
//     maxima / sums are half-wave reductions over <= 5 registers; P goes through 1 KB of LDS into the operand of the P V product; a wave merges its
//     2 - 3 tiles in registers (one max over all of them, no rescaling) and leaves ONE partial (o[64], max, sum) per query.  Waves 4 .. 7 -- the SIMD
//     partners -- merge the 4 partials of a query after the item's barrier and store the row.  Partials are double-buffered like the images; the
//     P staging area is the wave's own (still unwritten) partial slot.
// LDS: 2 x (272 x 128 + 64 x 576 + 128) + 16 + 2 x 4 x R x 272 B = 163 216 B at R = 9.
// Results: same arithmetic as the one-shot kernel of attention_vit.hip up to the summation order of the softmax denominators (tests hold both to
// the fp64 reference at the same bar).
#include "common.h"
#include <type_traits>

namespace amds {

constexpr int AX_KP = 256, AX_VS = 576, AX_KROWS = 272;               // 8 key tiles of 32 + 16 tail rows; V^T rows: 9 x 64 B, the ninth = the tail tile
constexpr int AX_K_BYTES = AX_KROWS * 128, AX_V_BYTES = 64 * AX_VS + 8 * 16;
constexpr int AX_BUF = AX_K_BYTES + AX_V_BYTES;                       // 71 808 B per item image
constexpr int AX_PART = 68;                                           // one partial of a tail query's row: o[64] | max | sum | pad (272 B: 16-byte multiples)
constexpr int ax_lds(int R) { return 2 * AX_BUF + 16 + 2 * 4 * R * AX_PART * 4; }

// (half_wave_max: common.h)

template <typename T, int R>
__global__ void __launch_bounds__(512) attn_vit26x_kernel(const T* __restrict__ qkv, T* __restrict__ out, int H, int n_items) {
    static_assert(R >= 4 && R <= 9, "tail of 4 .. 9 tokens (P staging needs 1 KB of a partial slot; the partials must fit beside the two images)");
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int Tn = 256 + R, KP = AX_KP, VS = AX_VS;
    constexpr int NREG = R > 8 ? 5 : (R > 4 ? 4 : R);                  // score registers of C[query][key] that hold a query < R in at least one half
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sZero = smem + 2 * AX_BUF;                                 // 16 zero bytes: what the lanes outside a short MFMA operand read
    float* sPart = reinterpret_cast<float*>(sZero + 16);             // [2][4 waves][R][AX_PART]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const float sc = 0.125f * 1.44269504088896340736f;                // 1/sqrt(64) * log2(e)
    const int swz = (l31 >> 1) & 7;

    // ---- staging of the NEXT item (as attention_vit257.hip: four parts of 64 keys through two register sets, written into the idle image
    // two chunk iterations later; Q fragments of the main block for the whole item) + the tail rows: K as 16-byte pieces (R x 8 threads),
    // V as 8-byte pieces (R x 16 threads: 4 dims of one tail key each, scattered into the ninth column of the V^T rows) ----
    struct Part { u32x4 k; u32x2 v0, v1; };
    Part pa, pb;
    vec8 qn[4];
    u32x4 tkn = u32x4{0u, 0u, 0u, 0u};
    u32x2 tvn = u32x2{0u, 0u};
    const int ldb = (int)ld * 2;                                      // bytes per token row
    const int vu = tid >> 1, vhalf = tid & 1;                         // V: (key pair, 16-byte chunk) = vu, 8-byte half of the chunk
    const int voff_k = (tid >> 3) * ldb + Dm * 2 + (tid & 7) * 16;
    const int voff_v = (vu >> 3) * 2 * ldb + Dm * 4 + (vu & 7) * 16 + vhalf * 8;
    const int voff_q = (wave * 32 + l31) * ldb + hi * 16;
    const int voff_qt = (KP + l31) * ldb + hi * 16;                   // tail query rows: lanes l31 >= R lie past the item's last row and read 0
    const int oob = Tn * ldb;                                         // first byte past the item: loads from here return 0
    const int voff_tk = tid < R * 8 ? (KP + (tid >> 3)) * ldb + Dm * 2 + (tid & 7) * 16 : oob;
    const int voff_tv = tid < R * 16 ? (KP + (tid >> 4)) * ldb + Dm * 4 + (tid & 15) * 8 : oob;
    const int lk_off = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);                     // K image: row = key, chunk ^ ((key >> 1) & 7)
    const int vk0 = (vu >> 3) * 2;                                                                    // first key of the pair inside the part
    const int lv_off = AX_K_BYTES + ((vu & 7) * 8 + vhalf * 4) * VS + (vu & 7) * 16 + ((vk0 & ~12) | ((vk0 & 4) << 1) | ((vk0 & 8) >> 1)) * 2;
    const int tkr = tid >> 3;                                                                         // tail key row of this thread's K piece
    const int ltk_off = (KP + tkr) * 128 + (((tid & 7) ^ ((tkr >> 1) & 7)) << 4);
    const int tvr = tid >> 4, tvd = (tid & 15) * 4;                                                   // tail key and first dim of this thread's V piece
    const int ltv_off = AX_K_BYTES + tvd * VS + (tvd >> 3) * 16 + (KP + ((tvr & ~12) | ((tvr & 4) << 1) | ((tvr & 8) >> 1))) * 2;
    auto part_load = [&](Part& pt, __amdgpu_buffer_rsrc_t rs, int c) {          // waits for nothing
        pt.k = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_k, c * 64 * ldb, 0);
        pt.v0 = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_v, c * 64 * ldb, 0);
        pt.v1 = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_v, c * 64 * ldb + ldb, 0);
    };
    auto part_store = [&](const Part& pt, char* buf, int c) {
        *reinterpret_cast<u32x4*>(buf + lk_off + c * 64 * 128) = pt.k;
        typedef T vec4t __attribute__((ext_vector_type(4)));
        typedef T vec2 __attribute__((ext_vector_type(2)));
        const vec4t a = __builtin_bit_cast(vec4t, pt.v0), b2 = __builtin_bit_cast(vec4t, pt.v1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                 // V^T image: row = dim, key order inside 16-groups: bits 2 <-> 3
            vec2 w;
            w[0] = a[e]; w[1] = b2[e];
            *reinterpret_cast<vec2*>(buf + lv_off + e * VS + c * 128) = w;
        }
    };
    auto rest_load = [&](__amdgpu_buffer_rsrc_t rs) {                 // this wave's Q fragments and the tail rows' K / V pieces
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qn[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_q + ks * 32, 0, 0));
        tkn = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_tk, 0, 0);
        tvn = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_tv, 0, 0);
    };
    auto rest_store = [&](char* buf) {
        if (tid < R * 8) *reinterpret_cast<u32x4*>(buf + ltk_off) = tkn;
        if (tid < R * 16) {
            typedef T vec4t __attribute__((ext_vector_type(4)));
            const vec4t a = __builtin_bit_cast(vec4t, tvn);
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<T*>(buf + ltv_off + e * VS + (((tvd + e) >> 3) - (tvd >> 3)) * 16) = a[e];
        }
    };
    auto item_rsrc = [&](int item) {
        const int b = item / H, h = item - b * H;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv + (long)b * Tn * ld + h * 64), 0, Tn * ldb, 0x00020000);
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    if (tid < 4) reinterpret_cast<float*>(sZero)[tid] = 0.f;
    // the tail rows of both K images and the tail column of both V^T images start as zeros: rows / keys >= R are never written, their scores
    // are masked and their softmax weights are 0 -- but 0 x (whatever the LDS held) must not be NaN
    for (int i = tid; i < 2 * (16 * 8 + 64 * 4); i += 512) {
        const int bsel = i / (16 * 8 + 64 * 4), j = i - bsel * (16 * 8 + 64 * 4);
        char* buf = smem + bsel * AX_BUF;
        if (j < 16 * 8) *reinterpret_cast<u32x4*>(buf + KP * 128 + j * 16) = u32x4{0u, 0u, 0u, 0u};
        else {
            const int d = (j - 128) >> 2, q4 = (j - 128) & 3;
            *reinterpret_cast<u32x4*>(buf + AX_K_BYTES + d * VS + (d >> 3) * 16 + KP * 2 + q4 * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();
    // a tail query's row, merged out of the 4 partials of waves 0 .. 3 one barrier after they were written (lane = dim)
    auto merge_tail = [&](int pbuf, int it, int q) {
        const float* pp = sPart + (pbuf * 4 * R + q) * AX_PART;
        float m = pp[64];
#pragma unroll
        for (int j = 1; j < 4; ++j) m = fmaxf(m, pp[j * R * AX_PART + 64]);
        float lsum = 0.f, ov = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float w = __builtin_amdgcn_exp2f(pp[j * R * AX_PART + 64] - m);
            lsum = fmaf(w, pp[j * R * AX_PART + 65], lsum);
            ov = fmaf(w, pp[j * R * AX_PART + lane], ov);
        }
        const int b = it / H, h = it - b * H;
        out[((long)b * Tn + KP + q) * Dm + h * 64 + lane] = Act<T>::from_f32(ov / lsum);
    };
    auto merge_item = [&](int pbuf, int it) {                         // waves 4 .. 7: queries wave - 4, wave, wave + 4 (< R)
#pragma unroll
        for (int q0 = 0; q0 < 12; q0 += 4)
            if (q0 + wave - 4 < R) merge_tail(pbuf, it, q0 + wave - 4);
    };
    {
        const __amdgpu_buffer_rsrc_t rs = item_rsrc(item);
        rest_load(rs);
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            part_load(pa, rs, c);
            part_load(pb, rs, c + 1);
            part_store(pa, smem, c);
            part_store(pb, smem, c + 1);
        }
        rest_store(smem);
    }
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    __syncthreads();

    int cur = 0;
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        const char* sK = smem + cur * AX_BUF;
        const char* sVt = sK + AX_K_BYTES;
        const bool has_next = item + (int)gridDim.x < n_items;
        if (wave >= 4 && item != (int)blockIdx.x) merge_item(cur ^ 1, item - (int)gridDim.x);
        const __amdgpu_buffer_rsrc_t crs = item_rsrc(item);
        const __amdgpu_buffer_rsrc_t nrs = item_rsrc(has_next ? item + (int)gridDim.x : item);      // its loads go out inside the chunk loop
        vec8 qt[4];                                                   // the tail queries of THIS item (R-row operand), requested before the last chunk

        // ---- this wave's 32 queries: online softmax over 4 chunks of 2 key tiles + the tail tile ----
        {
            f32x16 o[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            float mrun = -INFINITY, l = 0.f;
            // Software pipeline INSIDE the wave (attention_vit257.hip): stage c carries QK^T of chunk c + 1 (8 MFMAs, slices 0-7) and P V of chunk
            // c - 1 (8 MFMAs, slices 8-15) beside the softmax of chunk c, as 16 slices the scheduler may not move across.
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto stage = [&](auto has_pv_c, auto has_qk_c, int c, float alpha_prev, f32x16 (&sc_)[2], vec8 (&pout)[2][2], const vec8 (&pprev)[2][2],
                             f32x16 (&sn)[2]) {
                constexpr bool HP = decltype(has_pv_c)::value, HQ = decltype(has_qk_c)::value;
                vec8 opnd[4];
                auto ld = [&](int i) {                                   // operand of MFMA i: 0-7 = QK^T (K rows), 8-15 = P V (V^T rows)
                    if (i < 0 || i >= 16) return;
                    if (i < 8) {
                        if (!HQ) return;
                        const int t = i & 1, ks = i >> 1;
                        opnd[i & 3] = *reinterpret_cast<const vec8*>(sK + (((c + 1) * 2 + t) * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
                    } else {
                        if (!HP) return;
                        const int q = i - 8, t = q >> 2, ks = (q >> 1) & 1, dt = q & 1;
                        const int pos = ((c - 1) * 2 + t) * 32 + ks * 16 + hi * 8, d = dt * 32 + l31;
                        opnd[i & 3] = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
                    }
                };
                auto mf = [&](int i) {
                    if (i < 8) {
                        if (!HQ) return;
                        const int t = i & 1, ks = i >> 1;
                        sn[t] = Act<T>::mfma32(opnd[i & 3], qf[ks], ks == 0 ? zero16 : sn[t]);
                    } else {
                        if (!HP) return;
                        const int q = i - 8, t = q >> 2, ks = (q >> 1) & 1, dt = q & 1;
                        o[dt] = Act<T>::mfma32(opnd[i & 3], pprev[t][ks], o[dt]);
                    }
                };
                float mx = -INFINITY, mnew = 0.f, alpha = 0.f, ls = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) ld(i);
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) {
                    ld(sl + 2);
                    mf(sl);
                    if (sl < 2) {                                         // running max of the 32 scores of this lane
#pragma unroll
                        for (int f = 16 * sl; f < 16 * sl + 16; ++f) mx = fmaxf(mx, sc_[f >> 4][f & 15]);
                    }
                    if (sl == 2) {
                        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                        mnew = fmaxf(mrun, mx * sc);
                        alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                        mrun = mnew;
                    }
                    if ((sl == 2 || sl == 3) && HP) {                     // (stage 0 has nothing to rescale; stage 1 multiplies zeros by 0)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[sl - 2][r] *= alpha_prev;
                    }
                    if (sl >= 3 && sl < 14) {                             // three weights per slice, rounded to the operand type at once
#pragma unroll
                        for (int f = 3 * (sl - 3); f < 3 * (sl - 3) + 3 && f < 32; ++f) {
                            const float pw = __builtin_amdgcn_exp2f(fmaf(sc_[f >> 4][f & 15], sc, -mnew));
                            ls += pw;
                            pout[f >> 4][(f >> 3) & 1][f & 7] = Act<T>::from_f32(pw);
                        }
                    }
                    if (sl == 14) l = l * alpha + ls;
                    __builtin_amdgcn_sched_barrier(0);
                }
                return alpha;
            };
            typedef std::true_type Y;
            typedef std::false_type N_;
            char* nbuf = smem + (cur ^ 1) * AX_BUF;
            f32x16 sa[2], sb[2];
            vec8 p0[2][2], p1[2][2];
            part_load(pa, nrs, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sa[t] = zero16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const vec8 kf = *reinterpret_cast<const vec8*>(sK + (t * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
                    sa[t] = Act<T>::mfma32(kf, qf[ks], sa[t]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            part_load(pb, nrs, 1);
            __builtin_amdgcn_sched_barrier(0);
            const float al0 = stage(N_{}, Y{}, 0, 0.f, sa, p0, p1, sb);
            part_store(pa, nbuf, 0);
            part_load(pa, nrs, 2);
            __builtin_amdgcn_sched_barrier(0);
            const float al1 = stage(Y{}, Y{}, 1, al0, sb, p1, p0, sa);
            part_store(pb, nbuf, 1);
            part_load(pb, nrs, 3);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qt[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(crs, voff_qt + ks * 32, 0, 0));
            __builtin_amdgcn_sched_barrier(0);
            const float al2 = stage(Y{}, Y{}, 2, al1, sa, p0, p1, sb);
            rest_load(nrs);
            __builtin_amdgcn_sched_barrier(0);
            const float al3 = stage(Y{}, N_{}, 3, al2, sb, p1, p0, sa);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= al3;
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                 // P V of the last chunk
                const int t = i >> 2, ks = (i >> 1) & 1, dt = i & 1;
                const int pos = (6 + t) * 32 + ks * 16 + hi * 8, d = dt * 32 + l31;
                const vec8 vf = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
                o[dt] = Act<T>::mfma32(vf, p1[t][ks], o[dt]);
            }
            {   // the tail tile: keys 256 .. 255 + R as rows 0 .. R - 1 of a 32-row operand (lanes 16 .. 31 read zeros; rows R .. 15 are zero rows)
                f32x16 st = zero16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const char* ksrc = l31 < 16 ? sK + (KP + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4) : sZero;
                    st = Act<T>::mfma32(*reinterpret_cast<const vec8*>(ksrc), qf[ks], st);
                }
                // register r of C[key][query]: key = 8 (r >> 2) + 4 hi + (r & 3); only r < 8 can hold a key < 16
                float sv[8], mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int key = 8 * (r >> 2) + 4 * hi + (r & 3);
                    sv[r] = key < R ? st[r] * sc : -INFINITY;
                    mx = fmaxf(mx, sv[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(mrun, mx), alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                mrun = mnew;
                float ls = 0.f;
                vec8 pt;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float pw = __builtin_amdgcn_exp2f(sv[r] - mnew);
                    ls += pw;
                    pt[r] = Act<T>::from_f32(pw);
                }
                l = l * alpha + ls;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                    const int d = dt * 32 + l31;
                    const vec8 vf = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + (KP + hi * 8) * 2);
                    o[dt] = Act<T>::mfma32(vf, pt, o[dt]);
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

        // ---- the tail queries on waves 0 .. 3 (one per SIMD): key tiles w and w + 4, wave 0 also the tail tile.  C[query][key]: lane = key (l31),
        // register r <-> query 8 (r >> 2) + 4 hi + (r & 3).  One max over the wave's tiles, then P -> 1 KB of LDS (this wave's partial slot) -> P V ----
        if (wave < 4) {
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float* slot = sPart + ((cur * 4 + wave) * R) * AX_PART;
            char* sP = reinterpret_cast<char*>(slot);                 // [16 query rows][32 keys] 16-bit, key order of the V^T image
            const int ntile = wave == 0 ? 3 : 2;
            float s[3][NREG];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < ntile) {
                    const bool tail = t == 2;
                    const int row0 = tail ? KP : (wave + 4 * t) * 32;
                    f32x16 acc = zero16;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const char* ksrc = (!tail || l31 < 16) ? sK + (row0 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4) : sZero;
                        acc = Act<T>::mfma32(qt[ks], *reinterpret_cast<const vec8*>(ksrc), acc);
                    }
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int rr = r < 4 ? r : 4;                     // (NREG = 5: registers 0 .. 3 and 4)
                        const int q = 8 * (rr >> 2) + 4 * hi + (rr & 3);
                        const bool ok = q < R && (!tail || l31 < R);
                        s[t][r] = ok ? acc[rr] * sc : -INFINITY;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < NREG; ++r) s[t][r] = -INFINITY;
                }
            }
            float m[NREG], lq[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int rr = r < 4 ? r : 4;
                const int q = 8 * (rr >> 2) + 4 * hi + (rr & 3);
                const float mw = half_wave_max(fmaxf(fmaxf(s[0][r], s[1][r]), s[2][r]));
                m[r] = q < R ? mw : 0.f;                                  // (rows >= R: every score is -inf; keep exp2(-inf - m) = 0, not NaN)
                lq[r] = 0.f;
            }
            f32x16 oq[2] = {zero16, zero16};
            const int kperm = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < ntile) {
                    const bool tail = t == 2;
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const int rr = r < 4 ? r : 4;
                        const int q = 8 * (rr >> 2) + 4 * hi + (rr & 3);
                        const float pk = __builtin_amdgcn_exp2f(s[t][r] - m[r]);
                        lq[r] += pk;
                        reinterpret_cast<T*>(sP + q * 64)[kperm] = Act<T>::from_f32(pk);      // (q <= 12 < 16 rows of the staging area)
                    }
                    asm volatile("" ::: "memory");                        // same wave, LDS in order: the reads below see the writes above
                    const int pos0 = tail ? KP : (wave + 4 * t) * 32;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        if (tail && ks == 1) continue;                    // tail keys 16 .. 31 do not exist
                        const char* psrc = l31 < 16 ? sP + l31 * 64 + hi * 16 + ks * 32 : sZero;
                        const vec8 pf = *reinterpret_cast<const vec8*>(psrc);
                        const int pos = pos0 + ks * 16 + hi * 8;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            const int d = dt * 32 + l31;
                            const vec8 vf = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
                            oq[dt] = Act<T>::mfma32(vf, pf, oq[dt]);
                        }
                    }
                    asm volatile("" ::: "memory");
                }
            }
            // the partial: o[64] from the lanes l31 = query (C[dim][query]), max / sum from lane l31 = 0 of the half that holds the query
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the staging rows are read before the slot is overwritten
            if (l31 < R) {
                float* pp = slot + l31 * AX_PART;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(pp + dt * 32 + 8 * g + 4 * hi) = f32x4{oq[dt][4 * g], oq[dt][4 * g + 1], oq[dt][4 * g + 2], oq[dt][4 * g + 3]};
            }
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int rr = r < 4 ? r : 4;
                const int q = 8 * (rr >> 2) + 4 * hi + (rr & 3);
                const float lw = half_wave_sum(lq[r]);
                if (l31 == 0 && q < R) { slot[q * AX_PART + 64] = m[r]; slot[q * AX_PART + 65] = lw; }
            }
        }

        // ---- the next item: registers -> the other LDS buffer (its loads have had this whole item to arrive) ----
        {   // (past the last item this restages the item itself into the idle image: no branch, no conditional definitions)
            char* nbuf = smem + (cur ^ 1) * AX_BUF;
            part_store(pa, nbuf, 2);
            part_store(pb, nbuf, 3);
            rest_store(nbuf);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
        }
        __syncthreads();
        cur ^= 1;
    }
    if (wave >= 4) merge_item(cur ^ 1, item - (int)gridDim.x);       // the last item's tail queries (its barrier is the loop's last one)
}


template <typename T, int R>
static int launch_attn26x(const void* qkv, void* out, int B, int H, hipStream_t st) {
    auto kern = attn_vit26x_kernel<T, R>;
    constexpr int LDS = ax_lds(R);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int n_cus = device_cu_count();
    AMDS_REQUIRE(n_cus > 0, "attention: cannot read the device's multiprocessor count");
    const int n_items = B * H;
    hipLaunchKernelGGL(kern, dim3(min(n_items, n_cus)), dim3(512), LDS, st, (const T*)qkv, (T*)out, H, n_items);
    AMDS_LAUNCH_CHECK("attn_vit26x_kernel");
    return AMDS_OK;
}

// called by amds_attention_vit for T = 261 / 265, head_dim 64 (attention_vit.hip); dtype already validated.  -1 = not this kernel's shape.
int attention_vit26x(const void* qkv, void* out, int B, int T, int H, int dtype, hipStream_t st) {
    if (T == 261) return dtype == AMDS_F16 ? launch_attn26x<f16, 5>(qkv, out, B, H, st) : launch_attn26x<bf16, 5>(qkv, out, B, H, st);
    if (T == 265) return dtype == AMDS_F16 ? launch_attn26x<f16, 9>(qkv, out, B, H, st) : launch_attn26x<bf16, 9>(qkv, out, B, H, st);
    return -1;
}

}  // namespace amds
