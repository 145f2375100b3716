// attention_vit257.hip -- the tile encoder's attention for the shape every "class token + 16 x 16 patches" model has (T = 257 tokens,
// head_dim 64): persistent workgroups, DOUBLE-BUFFERED K / V^T images in LDS, the next item staged while the current one is computed.
//
// attention_vit.hip's one-shot kernel (one workgroup per (tile, head), 2 per CU) spent ~26 k cycles per item and CU slot against 12.9 k for
// its HBM bytes (132 KB per item at the CU's share of the achievable bandwidth) and ~8 k of MFMA + softmax work: every item exposed one
// HBM round trip for the K / V staging loads and one per query block for the Q fragments, ~2 us each under load, and two workgroups per
// CU do not cover that.  Here ONE 512-thread workgroup per CU (two waves per SIMD: one wave's MFMAs beside the other's softmax VALU)
// walks items blockIdx.x, blockIdx.x + gridDim.x, ...; while item i is computed out of LDS image i & 1, the K / V rows of item i + 1 travel
// HBM -> registers -> image (i + 1) & 1 in four parts of 64 keys pipelined through the chunk loop (two 8-register sets), its Q fragments and
// odd token HBM -> registers for the whole item: one barrier per item, no exposed round trip.  8 query blocks of 32 on 8 waves: one block per
// wave (the one-shot kernel had 2 per wave on 4 waves); the odd key as a rank-1 VALU update; the odd query on the MFMA pipe, 32 keys per wave
// (a one-row operand), its row merged from 9 flash-style partials by wave 0 after the item's barrier (as a 257-key GEMV with lane = key, then
// lane = dim, behind three barriers it was 18 % of an item).  The four chunks of a wave's online softmax are software-pipelined INSIDE the wave:
// stage c issues QK^T of chunk c + 1 and P V of chunk c - 1 (16 MFMAs) in the gaps of chunk c's softmax (16 slices the scheduler may not
// move across).  Round 2: 775 -> 742 (odd query) -> 695 us (pipeline) per 1020-tile launch.
// Arithmetic of the 256 even queries, LDS images and the no-shuffle MFMA operand layout are those of attention_vit.hip.
#include "common.h"
#include <type_traits>

namespace amds {

constexpr int A7_KP = 256, A7_VS = 576;                                // 8 key tiles of 32; V^T rows of vt_row_bytes(8) = 9 x 64 B
constexpr int A7_K_BYTES = A7_KP * 128, A7_V_BYTES = 64 * A7_VS + 8 * 16, A7_T_BYTES = 3 * 64 * 4 + 128;       // + the odd query in 16 bit
constexpr int A7_BUF = A7_K_BYTES + A7_V_BYTES + A7_T_BYTES;          // 70 656 B per item image
constexpr int A7_PART = 68;                                           // one partial of the odd query's row: o[64] | max | sum | pad
constexpr int A7_SCRATCH = 16 + 8 * 64 + 2 * 9 * A7_PART * 4;         // 16 zero bytes | P of the odd query, 32 keys per wave | partials x 2
constexpr int A7_OUT = 8 * 2048;                                      // per wave: 16 output rows x 128 B, staged so that a row leaves as one 128-byte line
constexpr int A7_LDS = 2 * A7_BUF + A7_SCRATCH + A7_OUT;              // 163 120 B

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
__global__ void __launch_bounds__(512) attn_vit257_kernel(const T* __restrict__ qkv, T* __restrict__ out, int H, int n_items) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int Tn = 257, KP = A7_KP, VS = A7_VS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sZero = smem + 2 * A7_BUF;                                 // 16 zero bytes: what the lanes outside a 1-row MFMA operand read
    char* sPw = sZero + 16;                                          // [8 waves][32] 16-bit softmax weights of the odd query
    float* sPart = reinterpret_cast<float*>(sPw + 8 * 64);           // [2][9][A7_PART]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const float sc = 0.125f * 1.44269504088896340736f;                // 1/sqrt(64) * log2(e)
    const int swz = (l31 >> 1) & 7;

    // ---- staging of the NEXT item.  Its K / V rows come in four parts of 64 keys, part c requested at the top of chunk iteration c and
    // written into the other LDS image two chunk iterations later (that image is idle for the whole item), through two 8-register sets
    // (even / odd parts); its Q fragments and the odd token are requested before the chunk loop and kept until the item's end.  All 14 loads
    // of a thread at once, from 8 waves, queued up behind the CU's 64 B / clock vector-memory path and held the slowest wave for > 4 k cycles
    // before it started computing.  Buffer addressing: one 32-bit byte offset per stream and thread, the item in the descriptor, the part in
    // a scalar offset. ----
    struct Part { u32x4 k; u32x2 v0, v1; };
    Part pa, pb;
    T tq = (T)0.f, tk = (T)0.f, tv = (T)0.f;     // raw 16-bit values: converting here would put an s_waitcnt right behind the loads
    vec8 qn[4];
    const int ldb = (int)ld * 2;                                      // bytes per token row
    const int vu = tid >> 1, vhalf = tid & 1;                         // V: (key pair, 16-byte chunk) = vu, 8-byte half of the chunk
    const int voff_k = (tid >> 3) * ldb + Dm * 2 + (tid & 7) * 16;
    const int voff_v = (vu >> 3) * 2 * ldb + Dm * 4 + (vu & 7) * 16 + vhalf * 8;
    const int voff_q = (wave * 32 + l31) * ldb + hi * 16;
    const int lk_off = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);                     // K image: row = key, chunk ^ ((key >> 1) & 7)
    const int vk0 = (vu >> 3) * 2;                                                                    // first key of the pair inside the part
    const int lv_off = A7_K_BYTES + ((vu & 7) * 8 + vhalf * 4) * VS + (vu & 7) * 16 + ((vk0 & ~12) | ((vk0 & 4) << 1) | ((vk0 & 8) >> 1)) * 2;
    auto part_load = [&](Part& pt, __amdgpu_buffer_rsrc_t rs, int c) {          // waits for nothing
#if defined(A7_ABL) && (A7_ABL & 2)      // ablation: the next item's K / V rows are not fetched (registers keep their content)
        if (c >= 0) return;
#endif
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
    auto rest_load = [&](__amdgpu_buffer_rsrc_t rs) {                 // this wave's Q fragments and the odd token
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qn[ks] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_q + ks * 32, 0, 0));
        // token 256 = the odd one; every wave loads it (only wave 0 stores it): a branch here would cut the pipelined section into two
        // basic blocks, and the compiler sinks a stage's vector work into the later one
        tq = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, KP * ldb, 0));
        tk = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, KP * ldb + Dm * 2, 0));
        tv = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2, KP * ldb + Dm * 4, 0));
    };
    auto rest_store = [&](char* buf) {
        float* sT = reinterpret_cast<float*>(buf + A7_K_BYTES + A7_V_BYTES);
        if (tid < 64) {                                               // tail key | value | query in fp32, the query again as it came
            sT[tid] = Act<T>::to_f32(tk); sT[64 + tid] = Act<T>::to_f32(tv); sT[128 + tid] = Act<T>::to_f32(tq);
            reinterpret_cast<T*>(sT + 192)[tid] = tq;
        }
    };
    auto item_rsrc = [&](int item) {
        const int b = item / H, h = item - b * H;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv + (long)b * Tn * ld + h * 64), 0, Tn * ldb, 0x00020000);
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    if (tid < 4) reinterpret_cast<float*>(sZero)[tid] = 0.f;
    // the odd query's row, merged out of 9 partials (8 waves x 32 keys + the odd key) one barrier after they were written
    auto merge_odd = [&](int pbuf, int it) {
        const float* pp = sPart + pbuf * 9 * A7_PART;
        float m = pp[64];
#pragma unroll
        for (int j = 1; j < 9; ++j) m = fmaxf(m, pp[j * A7_PART + 64]);
        float lsum = 0.f, ov = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float w = __builtin_amdgcn_exp2f(pp[j * A7_PART + 64] - m);
            lsum = fmaf(w, pp[j * A7_PART + 65], lsum);
            ov = fmaf(w, pp[j * A7_PART + lane], ov);
        }
        const int b = it / H, h = it - b * H;
        out[((long)b * Tn + KP) * Dm + h * 64 + lane] = Act<T>::from_f32(ov / lsum);
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
        if (wave == 0 && item != (int)blockIdx.x) merge_odd(cur ^ 1, item - (int)gridDim.x);
        const __amdgpu_buffer_rsrc_t nrs = item_rsrc(has_next ? item + (int)gridDim.x : item);      // its loads go out inside the chunk loop
        A7_MARK(1);

        // ---- this wave's 32 queries: online softmax over 4 chunks of 2 key tiles ----
        {
            f32x16 o[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            float mrun = -INFINITY, l = 0.f;
            // Software pipeline INSIDE the wave.  The matrix pipe takes 32 cycles per MFMA and a wave can issue ~7 independent vector
            // instructions in each gap -- if independent work stands there in program order.  Stage c therefore carries
            //   QK^T of chunk c + 1 (8 MFMAs, slices 0-7) and P V of chunk c - 1 (8 MFMAs, slices 8-15)   beside   the softmax of chunk c,
            // written out as 16 slices (one MFMA, the LDS read of the MFMA two slices on, ~10 vector instructions) that the scheduler may
            // not move across.  The accumulators are rescaled by the PREVIOUS stage's factor in slices 2-3: after P V (c - 2), which ended
            // a stage ago, and before P V (c - 1).
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto stage = [&](auto has_pv_c, auto has_qk_c, int c, float alpha_prev, f32x16 (&sc_)[2], vec8 (&pout)[2][2], const vec8 (&pprev)[2][2],
                             f32x16 (&sn)[2]) {
                constexpr bool HP = decltype(has_pv_c)::value, HQ = decltype(has_qk_c)::value;
                vec8 opnd[4];
                auto ld = [&](int i) {                                   // operand of MFMA i: 0-7 = QK^T (K rows), 8-15 = P V (V^T rows)
                    if (i < 0 || i >= 16) return;
#if defined(A7_ABL) && (A7_ABL & 4)      // ablation: no operand reads in the pipelined stages (stale registers)
                    if (i >= 0) { asm volatile("" : "+v"(opnd[i & 3])); return; }
#endif
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
#if defined(A7_ABL) && (A7_ABL & 8)      // ablation: no MFMAs in the pipelined stages
                    if (i >= 0) { asm volatile("" : "+v"(sn[i & 1]), "+v"(o[i & 1]) : "v"(opnd[i & 3])); return; }
#endif
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
#ifndef A7_LD_DIST
#define A7_LD_DIST 2
#endif
#pragma unroll
                for (int i = 0; i < A7_LD_DIST; ++i) ld(i);
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) {
                    ld(sl + A7_LD_DIST);
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
#if defined(A7_ABL) && (A7_ABL & 1)      // ablation (tools/ubench/attn257_abl.hip): no v_exp_f32 in the chunk softmax
                            const float pw = fmaf(sc_[f >> 4][f & 15], sc, -mnew);
#else
                            const float pw = __builtin_amdgcn_exp2f(fmaf(sc_[f >> 4][f & 15], sc, -mnew));
#endif
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
            char* nbuf = smem + (cur ^ 1) * A7_BUF;
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
            // Output rows through 2 KB of LDS per wave, 16 queries at a time: the accumulators hold a query's 64 dims as 8-byte groups spread over the lane pair
            // (l31, hi), and stored from there every store instruction of a wave hit 32 rows with 16 contiguous bytes each (8 such instructions per item; ablation
            // tools/ubench/attn257_abl.hip bit 16: the stores cost 104 of the kernel's 595 us).  Staged [query][dim] (16-byte chunk index XOR query & 7) and read
            // back row-wise, 8 consecutive lanes store one row's 128 bytes: 4 instructions of full lines per item; stores + staging now cost 80 us (595 -> 570).
            char* so = smem + 2 * A7_BUF + A7_SCRATCH + wave * 2048;
            T* obase = out + ((long)b * Tn + wave * 32) * Dm + h * 64;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if ((l31 >> 4) == r) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            vec4 w;
#pragma unroll
                            for (int e = 0; e < 4; ++e) w[e] = Act<T>::from_f32(o[dt][4 * g + e] * inv);
                            *reinterpret_cast<vec4*>(so + (l31 & 15) * 128 + (((dt * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = w;
                        }
                }
                asm volatile("" ::: "memory");                            // same wave, LDS in order: the reads below see the writes above
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int row = k * 8 + (lane >> 3), c = lane & 7;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(so + row * 128 + ((c ^ (row & 7)) << 4));
#if defined(A7_ABL) && (A7_ABL & 16)     // ablation: the 256 main rows are not stored
                    if (n_items < 0)
#endif
                    *reinterpret_cast<u32x4*>(obase + (long)(r * 16 + row) * Dm + c * 8) = v;
                }
                asm volatile("" ::: "memory");
            }
        }

        A7_MARK(3);
        // ---- the odd query: wave w takes keys 32 w .. 32 w + 31 on the MFMA pipe.  Scores = (a 32-row operand whose row 0 is the query, the
        // other lanes read zeros) x K^T: row 0 of the result = one key per lane (hi = 0); its softmax weights go through 64 B of LDS into the
        // column-0 operand of the P x V product; (max, sum, o[64]) of the 32 keys is a partial that wave 0 merges after the item's barrier ----
        {
            const char* sQh = reinterpret_cast<const char*>(sKt + 192);
            const char* qsrc = l31 == 0 ? sQh + hi * 16 : sZero;
            const int qstep = l31 == 0 ? 32 : 0;
            f32x16 s1;
#pragma unroll
            for (int r = 0; r < 16; ++r) s1[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 qa = *reinterpret_cast<const vec8*>(qsrc + ks * qstep);
                const vec8 kf = *reinterpret_cast<const vec8*>(sK + (wave * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
                s1 = Act<T>::mfma32(qa, kf, s1);
            }
            // (the 32 scores sit in the hi == 0 half: 32-lane DPP butterflies instead of two 6-step ds_bpermute chains; the V^T operands of the products below are
            //  requested before the reductions, not behind them)
            vec8 vfo[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int pos = wave * 32 + ks * 16 + hi * 8, d = dt * 32 + l31;
                    vfo[ks][dt] = *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
                }
            const float sv = hi == 0 ? s1[0] * sc : -INFINITY;
            const float mw = half_wave_max(sv);
            const float pk = hi == 0 ? __builtin_amdgcn_exp2f(sv - mw) : 0.f;
            const float lw = half_wave_sum(pk);
            char* pw = sPw + wave * 64;
            if (hi == 0) reinterpret_cast<T*>(pw)[(l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)] = Act<T>::from_f32(pk);      // the V^T image's key order
            asm volatile("" ::: "memory");                            // same wave, LDS in order: the reads below see the writes above
            const char* psrc = l31 == 0 ? pw + hi * 16 : sZero;
            f32x16 oq[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oq[dt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const vec8 pf = *reinterpret_cast<const vec8*>(psrc + ks * qstep);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) oq[dt] = Act<T>::mfma32(vfo[ks][dt], pf, oq[dt]);
            }
            float* pp = sPart + (cur * 9 + wave) * A7_PART;
            if (l31 == 0) {                                           // column 0 of the product: 32 dims in lane 0, 32 in lane 32
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(pp + dt * 32 + 8 * g + 4 * hi) = f32x4{oq[dt][4 * g], oq[dt][4 * g + 1], oq[dt][4 * g + 2], oq[dt][4 * g + 3]};
                if (hi == 0) { pp[64] = mw; pp[65] = lw; }
            }
            if (wave == 1) {                                          // the odd key's term of that row as the ninth partial
                float* p8 = sPart + (cur * 9 + 8) * A7_PART;
                float st = half_wave_sum(sQt[lane] * sKt[lane]);
                st = (st + __shfl_xor(st, 32, 64)) * sc;
                p8[lane] = sVl[lane];
                if (lane == 0) { p8[64] = st; p8[65] = 1.0f; }
            }
        }

        A7_MARK(4);
        // ---- the next item: registers -> the other LDS buffer (its loads have had this whole item to arrive) ----
        {   // (past the last item this restages the item itself into the idle image: no branch, no conditional definitions)
            char* nbuf = smem + (cur ^ 1) * A7_BUF;
            part_store(pa, nbuf, 2);
            part_store(pb, nbuf, 3);
            rest_store(nbuf);
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
    if (wave == 0) merge_odd(cur ^ 1, item - (int)gridDim.x);        // the last item's odd query (its barrier is the loop's last one)
}


template <typename T>
static int launch_attn257(const void* qkv, void* out, int B, int H, hipStream_t st) {
    auto kern = attn_vit257_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, A7_LDS));
        attr_set = true;
    }
    const int n_cus = device_cu_count();
    AMDS_REQUIRE(n_cus > 0, "attention: cannot read the device's multiprocessor count");
    const int n_items = B * H;
    hipLaunchKernelGGL(kern, dim3(min(n_items, n_cus)), dim3(512), A7_LDS, st, (const T*)qkv, (T*)out, H, n_items);
    AMDS_LAUNCH_CHECK("attn_vit257_kernel");
    return AMDS_OK;
}

// called by amds_attention_vit for T = 257 (attention_vit.hip); dtype already validated
int attention_vit257(const void* qkv, void* out, int B, int H, int dtype, hipStream_t st) {
    return dtype == AMDS_F16 ? launch_attn257<f16>(qkv, out, B, H, st) : launch_attn257<bf16>(qkv, out, B, H, st);
}

}  // namespace amds
