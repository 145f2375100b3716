// attention_flash.hip -- multi-head self-attention for ANY sequence length (head_dim 64): the MIL heads attend over
// a whole bag (T+1 = 1025 tokens in training, up to tens of thousands at deploy time, where the reference simply
// materialises the T x T matrix -- src/stamp/modeling/models/__init__.py:286-313 disables the mask "to reduce
// memory").  Reference op: nn.MultiheadAttention / F.scaled_dot_product_attention at
// src/stamp/modeling/models/vision_tranformer.py:191, 217-227 (mask = None path).
//
// One workgroup = 128 queries (4 waves x 32) of one (bag, head); K / V stream through LDS in tiles of 64 keys,
// double buffered by LDS-DMA (row-major images; V^T fragments by transpose reads; see attention_vit.hip for the layout
// rules: S^T = K Q^T keeps a query's scores in one lane, the key order inside 16-key groups is such that P is
// already the MFMA B operand).  Scores never leave registers: N^2 is never materialised, K/V are re-read once per
// 128-query block (L2-resident: 2 x T x 128 B per head).
#include "common.h"

namespace amds {

constexpr int FA_KT = 64;                       // keys per tile
constexpr int FA_K_BYTES = FA_KT * 128;         // a row-major image of 64 keys (K, and V alike): 8 KB
constexpr int FA_C_BYTES = FA_KT * 8;            // key coordinates (float2) of a tile, ALiBi variant only

// ALIBI: the reference's MultiHeadALiBi (src/stamp/modeling/models/vision_tranformer.py:42-74): the distance bias is
// SUBTRACTED FROM THE PROBABILITIES (after the softmax):  out = softmax(q k^T/8) v  -  s_h * cdist(c_q, c_k) v,
// s_h = bias_scale_h / running_mean_h.  The second product is accumulated by a third MFMA chain on the same V^T
// fragments; distances are evaluated in fp32 from the token coordinates in registers (no T x T tensor).
// TO: output element type.  The ALiBi variant always writes bf16: its second term sums |V| over ALL keys with weights
// of order 1-10 (no softmax normalisation), which overflows fp16 (65504) on long bags; bf16 has the fp32 range.
//
// MASK: the reference's `mask != None` path (vision_tranformer.py:355-381), restated literally.  pad[b][t] (u8, class token
// included at t = 0 and never padded) marks padded tiles; blocked(q, k) = (pad[q] & pad[k]) | (q > 0 & k == 0): padded queries
// do not see padded keys, tiles never see the class token (:363-367) -- an UNpadded query still attends to every key, padded or
// not, exactly as the reference's outer-product mask does.  nn.MultiheadAttention branch: blocked scores are -inf before the
// softmax, and because the reference hands MultiheadAttention `attn_mask.repeat(heads, 1, 1)` (:224), (bag b, head h) is masked
// with the pad row of bag (b*H + h) % B.  ALiBi branch (:62-70): the softmax runs over ALL keys, blocked products are zeroed
// afterwards, and the distance term is dropped on the class-token row and column (alibi_mask, :370-372); mask row = bag b.
// DROP (plain attention, training): nn.MultiheadAttention's dropout on the attention probabilities: out = drop(P) v with the
// normaliser and the saved log-sum-exp those of the undropped P; mask bits from (seed, stream, (b,h,q), k) -- common.h drop_*.
// Data path (round 6; the first form -- global -> registers -> LDS with a transposing store pass for V, 42 KB and 146-161 registers -- is kept as text in
// tools/ubench/attic/attention_first_forms/, A/B in profiles/r06_attn_bwd_ab.txt: same bits, -7.5 %): K and V tiles by buffer-form LDS-DMA (no staging registers),
// the V^T fragments of the P V product by ds_read_b64_tr_b16 transpose reads of the row-major V image, two 16 KB stages, a raw s_barrier per tile; 121-128 registers:
// four workgroups per CU (ALiBi: 211, two).  16-byte chunk c of key row k sits at chunk c ^ attn_swz(k): row reads and transpose reads are both conflict-free.
#define FA_TR8(out, a00, a01, a10, a11, imm)                                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %10 offset:%12\n\t"                                  \
                 "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %11 offset:%12\n\t"                                   \
                 "ds_read_b64_tr_b16 %4, %8 offset:%13\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"                                   \
                 "ds_read_b64_tr_b16 %6, %9 offset:%13\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\ts_waitcnt lgkmcnt(0)"               \
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])  \
                 : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "n"(imm), "n"((imm) + 2048)                                                  \
                 : "memory")
// the LDS-DMA addresses one bag's q | k | v rows through a buffer descriptor: 32-bit byte offsets
#define FA_SPAN_OK(T, H) ((long)(T) * 3 * (H) * 128 < (1L << 31))
__device__ __forceinline__ int attn_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

template <typename T, bool ALIBI, typename TO = T, bool MASK = false, bool DROP = false>
__global__ void __launch_bounds__(256, 2) attn_flash_kernel(const T* __restrict__ qkv, TO* __restrict__ out, int Tn, int H,
                                                            const float* __restrict__ coords, const float* __restrict__ head_scale,
                                                            float* __restrict__ lse_out, const float* __restrict__ out_scale = nullptr,
                                                            TO* __restrict__ u_out = nullptr, TO* __restrict__ osm_out = nullptr,
                                                            const uint8_t* __restrict__ pad = nullptr, uint64_t seed = 0, uint32_t drop_stream = 0,
                                                            uint32_t thr16 = 0, float keep_scale = 1.f, int mask_heads = 0) {
    typedef typename Act<T>::vec8 vec8;
    constexpr int F2_STAGE = 2 * FA_K_BYTES + FA_C_BYTES;                // K rows | V rows | key coordinates (ALiBi)
    __shared__ __attribute__((aligned(16))) char smem[2 * F2_STAGE + (MASK ? 2 * FA_KT : 0)];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, qblk = blockIdx.x;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;
    const int ntile = (Tn + FA_KT - 1) / FA_KT;

    // LDS-DMA (attention_train.hip, attn_bwd_dkdv2_kernel): wave w requests 8-row pieces w and w + 4 of the K and V images, rows past the sequence read as 0;
    // ALiBi: the tile's 128 coordinate dwords by waves 0 and 1.  Only the padding flags (bytes at an odd stride) still travel through a register.
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + Dm), 0, (int)((((long)Tn - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + 2 * Dm), 0, (int)((((long)Tn - 1) * ld + 64) * 2), 0x00020000);
    const float* cbase = ALIBI ? coords + (long)b * Tn * 2 : nullptr;
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ALIBI ? cbase : head_scale), 0, ALIBI ? Tn * 8 : 0, 0x00020000);
    uint8_t mreg = 0;
    const uint8_t* prow = nullptr;
    if constexpr (MASK) prow = pad + (long)(ALIBI ? b : (int)(((long)b * mask_heads + min(h, mask_heads - 1)) % gridDim.z)) * Tn;
    const int drow_ = wave * 8 + (lane >> 3);
    const int vo = drow_ * (int)ld * 2 + (((lane & 7) ^ attn_swz(drow_)) << 4);
    auto stage_load = [&](int j, int buf) {
        char* sK = smem + buf * F2_STAGE;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            bufl16(rsrc_k, sK + (wave + 4 * pc) * 1024, vo, (j * FA_KT + 32 * pc) * (int)ld * 2);
            bufl16(rsrc_v, sK + FA_K_BYTES + (wave + 4 * pc) * 1024, vo, (j * FA_KT + 32 * pc) * (int)ld * 2);
        }
        if constexpr (ALIBI) {
            if (wave < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lptr_t)(sK + 2 * FA_K_BYTES + wave * 256), 4, lane * 4, j * FA_KT * 8 + wave * 256, 0, 0);
        }
        if constexpr (MASK) {
            mreg = 1;
            if (tid < FA_KT && j * FA_KT + tid < Tn) mreg = prow[j * FA_KT + tid];
        }
    };
    auto stage_store = [&](int buf) {
        if constexpr (MASK) {
            if (tid < FA_KT) (smem + 2 * F2_STAGE + buf * FA_KT)[tid] = (char)mreg;
        }
    };

    // this wave's 32 queries
    const int q = qblk * 128 + wave * 32 + l31;
    const int qc = min(q, Tn - 1);
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const vec8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);

    f32x16 o[2], o2[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[dt][r] = 0.f; o2[dt][r] = 0.f; }
    float mrun = -INFINITY, l = 0.f;
    float xq = 0.f, yq = 0.f, sh = 0.f;
    if constexpr (ALIBI) { xq = cbase[(long)qc * 2]; yq = cbase[(long)qc * 2 + 1]; sh = head_scale[h]; }
    const float sc = 0.125f * 1.44269504088896340736f;
    const int swz = attn_swz(l31);
    // transpose reads of the V image (attention_train.hip): lane i of 16-lane group g hands in (key row (i >> 2) of the 4, 8-byte piece i & 3 of its 32 bytes)
    unsigned tra[2][2];
    {
        const int i16 = lane & 15, gd = (lane >> 4) & 1;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int row = hi * 4 + w * 8 + (i16 >> 2);
                const int lc = dt * 4 + gd * 2 + ((i16 & 3) >> 1);
                tra[w][dt] = (unsigned)(size_t)smem + FA_K_BYTES + row * 128 + ((lc ^ attn_swz(row)) << 4) + (i16 & 1) * 8;
            }
    }
    int flip = F2_STAGE;
    bool qpad = false;
    if constexpr (MASK) qpad = prow[qc] != 0;
    uint32_t rowkey = 0;
    if constexpr (DROP) rowkey = drop_rowkey(seed, drop_stream, (uint64_t)(((long)b * H + h) * Tn + qc));

    const bool wave_live = qblk * 128 + wave * 32 < Tn;
    stage_load(0, 0);
    stage_store(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < ntile; ++j) {
        const int buf = j & 1;
        if (j + 1 < ntile) stage_load(j + 1, buf ^ 1);
        const char* sK = smem + buf * F2_STAGE;
        const char* sV = sK + FA_K_BYTES;
        const char* sM = smem + 2 * F2_STAGE + buf * FA_KT;
        // A wave whose 32 queries all lie past the sequence (T = 1025 = 8 blocks of 128 + ONE query: three of the ninth block's four waves) only helps staging:
        // its issue slots go to the workgroup that shares the CU (the ninth block was 11 % of the launch for 0.1 % of the queries)
        if (wave_live) {
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 kf = *reinterpret_cast<const vec8*>(sK + (t * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4));
                s[t] = Act<T>::mfma32(kf, qf[ks], s[t]);
            }
        }
        const int key0 = j * FA_KT;
        const bool ragged = key0 + FA_KT > Tn;
        if (ragged) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= Tn) s[t][r] = -INFINITY;
                }
        }
        uint32_t blocked = 0;            // bit t*16 + r: this (query, key) product is masked out
        uint32_t nodist = 0;             // ALiBi: bit t*16 + r: no distance term either (alibi_mask: blocked, or the class token's row / column)
        if constexpr (MASK) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool bl = (qpad && sM[kl] != 0) || (q > 0 && key0 + kl == 0);
                    blocked |= (bl ? 1u : 0u) << (t * 16 + r);
                    if constexpr (ALIBI) nodist |= ((bl || q == 0 || key0 + kl == 0) ? 1u : 0u) << (t * 16 + r);
                    if (!ALIBI && bl) s[t][r] = -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float mnew = fmaxf(mrun, mx * sc);
        float msafe = mnew;
        if constexpr (MASK) msafe = (mnew == -INFINITY) ? 0.f : mnew;       // a query whose keys so far are all blocked
        const float alpha = __builtin_amdgcn_exp2f(mrun - msafe);
        mrun = mnew;
        mnew = msafe;
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
        if constexpr (MASK && ALIBI) {       // softmax over all keys, blocked products zeroed afterwards (:66-68)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((blocked >> (t * 16 + r)) & 1u) s[t][r] = 0.f;
        }
        if constexpr (DROP) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int k = key0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;      // even key, its partner k + 1 sits in register r + 1
                    const uint32_t bits = drop_pair_bits(rowkey, (uint32_t)k >> 1);
                    s[t][r] = drop_keep(bits, 0, thr16) ? s[t][r] * keep_scale : 0.f;
                    s[t][r + 1] = drop_keep(bits, 1, thr16) ? s[t][r + 1] * keep_scale : 0.f;
                }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x2 tv[8];                    // V^T fragment [ks*4 + dt*2 + w]: keys 16 ks + 8 w + 4 hi + 0..3 of the tile's half t, feature dt*32 + l31
            if (t == 0) FA_TR8(tv, tra[0][0], tra[0][1], tra[1][0], tra[1][1], 0); else FA_TR8(tv, tra[0][0], tra[0][1], tra[1][0], tra[1][1], 4096);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = Act<T>::from_f32(s[t][ks * 8 + e]);
                vec8 bf;
                if constexpr (ALIBI) {
                    const float* sC = reinterpret_cast<const float*>(sV + FA_K_BYTES);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = ks * 8 + e;
                        const int kl = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const float dx = xq - sC[kl * 2], dy = yq - sC[kl * 2 + 1];
                        float dist = dist_sqrt(dx * dx + dy * dy) * sh;
                        // (no bounds test: a key past the sequence has zero coordinates -- a finite distance -- and a zero V row)
                        if constexpr (MASK)     // AND with a sign-extended bit: as selects, the 32 conditions were held as wave masks and spilled the scalar file
                            dist = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, dist) & ~(uint32_t)(-(int32_t)((nodist >> (t * 16 + r)) & 1u)));
                        bf[e] = Act<T>::from_f32(dist);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const vec8 vf = __builtin_bit_cast(vec8, u32x4{tv[ks * 4 + dt * 2][0], tv[ks * 4 + dt * 2][1], tv[ks * 4 + dt * 2 + 1][0], tv[ks * 4 + dt * 2 + 1][1]});
                    o[dt] = Act<T>::mfma32(vf, pf, o[dt]);
                    if constexpr (ALIBI) o2[dt] = Act<T>::mfma32(vf, bf, o2[dt]);
                }
            }
        }
        }
        if (j + 1 < ntile) stage_store(buf ^ 1);
        tra[0][0] += flip; tra[0][1] += flip; tra[1][0] += flip; tra[1][1] += flip;
        flip = -flip;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the next tile has landed, this wave's reads of the current one are done
        __builtin_amdgcn_s_barrier();
    }
    l += __shfl_xor(l, 32, 64);
    if (lse_out && q < Tn && hi == 0) lse_out[((long)b * H + h) * Tn + q] = mrun + log2f(l);   // log2-domain log-sum-exp
    if (q < Tn) {
        const float inv = 1.0f / l;
        const float osc = (ALIBI && out_scale) ? out_scale[h] : 1.0f;
        TO* orow = out + ((long)b * Tn + q) * Dm + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename Act<TO>::vec4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = Act<TO>::from_f32(o[dt][4 * g + e] * inv - (ALIBI ? osc * o2[dt][4 * g + e] : 0.f));
                *reinterpret_cast<typename Act<TO>::vec4*>(orow + dt * 32 + 8 * g + 4 * hi) = w;
                if constexpr (ALIBI) {
                    if (u_out) {       // training: U = sum_k (dist / running_mean) v and the softmax part alone (out = Osm - bias_scale * U;
                                       // Osm cannot be rebuilt from the rounded out and U: |U| >> |Osm| cancels catastrophically)
                        const long off = ((long)b * Tn + q) * Dm + h * 64 + dt * 32 + 8 * g + 4 * hi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = Act<TO>::from_f32(o2[dt][4 * g + e]);
                        *reinterpret_cast<typename Act<TO>::vec4*>(u_out + off) = w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = Act<TO>::from_f32(o[dt][4 * g + e] * inv);
                        *reinterpret_cast<typename Act<TO>::vec4*>(osm_out + off) = w;
                    }
                }
            }
    }
}



// sum over the 8 lanes that share lane >> 3 (every lane gets it): xor 1, xor 2 (quad_perm), xor 7 (row_half_mirror)
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

// ---- ONE query row per (bag, head) against all keys / values of the bag: the class token's attention in the LAST layer of the MIL `vit` head -----------------
// The head reads only the class token's final row (reference vision_tranformer.py: `self.mlp_head(x[:, 0])` behind the last block), so the last block needs keys and
// values of every token but queries, output projection and MLP of the class rows alone (amds_mil_vit_forward).  One 256-thread workgroup per (bag, head): scores
// (thread = key, the 128-byte key row against the query in registers) into LDS, block maximum and sum, then the weighted value sum with thread = (4 dims, one of 16 key
// phases) and a 16-way LDS reduction.  fp32 throughout; q [B][ldq] and out [B][ldo] are 16-bit rows of the heads' 64-channel slices.
// DROP / lse: the training forward's form (amds_attention_fwd_train for query row `qrow` of every bag): dropout on the attention probabilities with the bits of
// the blocked kernel (row key of (bag, head, qrow), one hash per key pair), the log2-domain log-sum-exp of the UNdropped softmax into lse[(b H + h) T + qrow].
template <typename T, bool DROP = false>
__global__ void __launch_bounds__(256) attn_row_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ qkv, T* __restrict__ out, long ldo, int Tn, int H,
                                                       float* __restrict__ lse = nullptr, int qrow = 0, uint64_t seed = 0, uint32_t drop_stream = 0,
                                                       uint32_t thr16 = 0, float keep_scale = 1.f) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    extern __shared__ __attribute__((aligned(16))) float sS[];          // [Tn] scores -> weights | [16][64] partial outputs | [8] reductions
    float* sRed = sS + ((Tn + 3) & ~3);
    float* sW = sRed + 16 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;
    // scores: eight lanes per key, one 16-byte chunk of the 128-byte key row each (a wave reads eight whole rows per instruction), the dot product summed over
    // the eight lanes by DPP.  (One key per thread -- 64 lanes on 64 different rows per instruction -- ran at 2.9 TB/s.)
    const int sub = tid & 7, kk = tid >> 3;
    float qc[8];
    {
        const vec8 v = *reinterpret_cast<const vec8*>(q + (long)b * ldq + h * 64 + sub * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qc[e] = Act<T>::to_f32(v[e]);
    }
    const float sc = 0.125f * 1.44269504088896340736f;
    float mx = -INFINITY;
    for (int key = kk; key < Tn; key += 32) {
        const vec8 v = *reinterpret_cast<const vec8*>(base + (long)key * ld + Dm + sub * 8);
        float a0 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a0 = fmaf(qc[e], Act<T>::to_f32(v[e]), a0);
        const float sv = sum8_dpp(a0) * sc;
        if (sub == 0) sS[key] = sv;
        mx = fmaxf(mx, sv);
    }
    mx = wave_max(mx);
    if (lane == 0) sW[wave] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(sW[0], sW[1]), fmaxf(sW[2], sW[3]));
    float ls = 0.f;
    for (int key = tid; key < Tn; key += 256) {
        const float pw = __builtin_amdgcn_exp2f(sS[key] - m);
        sS[key] = pw;
        ls += pw;
    }
    ls = wave_sum(ls);
    if (lane == 0) sW[4 + wave] = ls;
    __syncthreads();
    const float l = (sW[4] + sW[5]) + (sW[6] + sW[7]);
    if (lse && tid == 0) lse[((long)b * H + h) * Tn + qrow] = m + log2f(l);
    uint32_t rowkey = 0;
    if constexpr (DROP) rowkey = drop_rowkey(seed, drop_stream, (uint64_t)(((long)b * H + h) * Tn + qrow));
    const int d4 = (tid & 15) * 4, ph = tid >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int key = ph; key < Tn; key += 16) {
        const vec4 v = *reinterpret_cast<const vec4*>(base + (long)key * ld + 2 * Dm + d4);
        float pw = sS[key];
        if constexpr (DROP) pw = drop_keep(drop_pair_bits(rowkey, (uint32_t)key >> 1), key & 1, thr16) ? pw * keep_scale : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(pw, Act<T>::to_f32(v[e]), acc[e]);
    }
    *reinterpret_cast<f32x4*>(sRed + ph * 64 + d4) = acc;
    __syncthreads();
    if (tid < 64) {
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) o += sRed[k * 64 + tid];
        out[(long)b * ldo + h * 64 + tid] = Act<T>::from_f32(o / l);
    }
}


// Backward of the one-query attention above for query row `qrow` of every bag, when NO other query row of the block has a gradient (the last block of the MIL
// `vit` head): the whole dqkv tensor of the block in one pass -- dK_k = dS_k q / 8 and dV_k = (M o P)_k dO are rank-1 in the one query, dQ is zero but for that
// row.  dS_k = P_k (M_k dP_k - D), dP_k = dO . v_k, D = dO . O (O the dropped output, as stored), P from the saved log-sum-exp; M regenerated from the counters.
template <typename T, bool DROP>
__global__ void __launch_bounds__(256) attn_row_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ o, long ldo, const T* __restrict__ dout, long ldd,
                                                           const float* __restrict__ lse, T* __restrict__ dqkv, int Tn, int H, int qrow, uint64_t seed,
                                                           uint32_t drop_stream, uint32_t thr16, float keep_scale) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    extern __shared__ __attribute__((aligned(16))) float sS[];          // [Tn] dS | [16][64] partial dq
    float* sRed = sS + ((Tn + 3) & ~3);
    const int tid = threadIdx.x;
    const int h = blockIdx.x, b = blockIdx.y;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;
    T* dbase = dqkv + (long)b * Tn * ld + h * 64;
    // eight lanes per key, one 16-byte chunk of each 128-byte row (q, k, v in; dQ, dK, dV out): whole rows per instruction, the two dot products by DPP
    const int sub = tid & 7, kk = tid >> 3;
    float qc[8], gc[8];
    float Dq = 0.f;
    {
        const vec8 qv = *reinterpret_cast<const vec8*>(base + (long)qrow * ld + sub * 8);
        const vec8 gv = *reinterpret_cast<const vec8*>(dout + (long)b * ldd + h * 64 + sub * 8);
        const vec8 ov = *reinterpret_cast<const vec8*>(o + (long)b * ldo + h * 64 + sub * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qc[e] = Act<T>::to_f32(qv[e]);
            gc[e] = Act<T>::to_f32(gv[e]);
            Dq = fmaf(gc[e], Act<T>::to_f32(ov[e]), Dq);
        }
        Dq = sum8_dpp(Dq);
    }
    const float sc = 0.125f * 1.44269504088896340736f;
    const float L = lse[((long)b * H + h) * Tn + qrow];
    uint32_t rowkey = 0;
    if constexpr (DROP) rowkey = drop_rowkey(seed, drop_stream, (uint64_t)(((long)b * H + h) * Tn + qrow));
    for (int key = kk; key < Tn; key += 32) {
        const T* krow = base + (long)key * ld + Dm + sub * 8;
        const vec8 kv = *reinterpret_cast<const vec8*>(krow), vv = *reinterpret_cast<const vec8*>(krow + Dm);
        float s0 = 0.f, p0 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s0 = fmaf(qc[e], Act<T>::to_f32(kv[e]), s0);
            p0 = fmaf(gc[e], Act<T>::to_f32(vv[e]), p0);
        }
        s0 = sum8_dpp(s0);
        p0 = sum8_dpp(p0);
        const float pk = __builtin_amdgcn_exp2f(s0 * sc - L);
        float mk = 1.f;
        if constexpr (DROP) mk = drop_keep(drop_pair_bits(rowkey, (uint32_t)key >> 1), key & 1, thr16) ? keep_scale : 0.f;
        const float dS = pk * (mk * p0 - Dq), pw = pk * mk, dk = dS * 0.125f;
        if (sub == 0) sS[key] = dS;
        T* drow = dbase + (long)key * ld + sub * 8;
        vec8 wq, wk, wv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wq[e] = (T)0.f;
            wk[e] = Act<T>::from_f32(dk * qc[e]);
            wv[e] = Act<T>::from_f32(pw * gc[e]);
        }
        if (key != qrow) *reinterpret_cast<vec8*>(drow) = wq;                      // (the query row's own dQ is written below)
        *reinterpret_cast<vec8*>(drow + Dm) = wk;
        *reinterpret_cast<vec8*>(drow + 2 * Dm) = wv;
    }
    __syncthreads();
    const int d4 = (tid & 15) * 4, ph = tid >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int key = ph; key < Tn; key += 16) {
        const vec4 kv = *reinterpret_cast<const vec4*>(base + (long)key * ld + Dm + d4);
        const float dS = sS[key];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(dS, Act<T>::to_f32(kv[e]), acc[e]);
    }
    *reinterpret_cast<f32x4*>(sRed + ph * 64 + d4) = acc;
    __syncthreads();
    if (tid < 64) {
        float dq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dq += sRed[k * 64 + tid];
        dbase[(long)qrow * ld + tid] = Act<T>::from_f32(dq * 0.125f);
    }
}

// ---- the one-query attention WITH the ALiBi distance term (training; reference vision_tranformer.py:42-74 with mask = None, i.e. alibi_mask = None: the class
// row carries the term too, :349-351 puts the class token at (0, 0)):  out = sum_k (p_k - bias_scale_h d_k) v_k,  d_k = |c_q - c_k| / running_mean_h.
// Saves, like the blocked kernel, the softmax part Osm = sum_k p_k v_k and U = sum_k d_k v_k (rows of the (bag, qrow) token) and the log-sum-exp.  fp32 distances
// (the blocked kernel rounds them to 16 bits for its MFMA).  No dropout: nn.MultiheadAttention's rate does not exist on the ALiBi path (:124-154).
template <typename T>
__global__ void __launch_bounds__(256) attn_row_alibi_kernel(const T* __restrict__ qkv, const float* __restrict__ coords, const float* __restrict__ inv_rm,
                                                             const float* __restrict__ bias_scale, T* __restrict__ out, T* __restrict__ u_out, T* __restrict__ osm_out,
                                                             float* __restrict__ lse, int Tn, int H, int qrow) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    extern __shared__ __attribute__((aligned(16))) float sS[];          // [Tn] scores -> weights | [Tn] scaled distances | 2 x [16][64] partial sums | [8] reductions
    const int Tp = (Tn + 3) & ~3;
    float* sD = sS + Tp;
    float* sRed = sD + Tp;
    float* sW = sRed + 2 * 16 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;
    const float* cb = coords + (long)b * Tn * 2;
    const float xq = cb[2 * qrow], yq = cb[2 * qrow + 1], irm = inv_rm[h];
    const int sub = tid & 7, kk = tid >> 3;
    float qc[8];
    {
        const vec8 v = *reinterpret_cast<const vec8*>(base + (long)qrow * ld + sub * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qc[e] = Act<T>::to_f32(v[e]);
    }
    const float sc = 0.125f * 1.44269504088896340736f;
    float mx = -INFINITY;
    for (int key = kk; key < Tn; key += 32) {
        const vec8 v = *reinterpret_cast<const vec8*>(base + (long)key * ld + Dm + sub * 8);
        float a0 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a0 = fmaf(qc[e], Act<T>::to_f32(v[e]), a0);
        const float sv = sum8_dpp(a0) * sc;
        if (sub == 0) {
            sS[key] = sv;
            const float dx = xq - cb[2 * key], dy = yq - cb[2 * key + 1];
            sD[key] = dist_sqrt(dx * dx + dy * dy) * irm;
        }
        mx = fmaxf(mx, sv);
    }
    mx = wave_max(mx);
    if (lane == 0) sW[wave] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(sW[0], sW[1]), fmaxf(sW[2], sW[3]));
    float ls = 0.f;
    for (int key = tid; key < Tn; key += 256) {
        const float pw = __builtin_amdgcn_exp2f(sS[key] - m);
        sS[key] = pw;
        ls += pw;
    }
    ls = wave_sum(ls);
    if (lane == 0) sW[4 + wave] = ls;
    __syncthreads();
    const float l = (sW[4] + sW[5]) + (sW[6] + sW[7]);
    if (tid == 0) lse[((long)b * H + h) * Tn + qrow] = m + log2f(l);
    const int d4 = (tid & 15) * 4, ph = tid >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acu = {0.f, 0.f, 0.f, 0.f};
    for (int key = ph; key < Tn; key += 16) {
        const vec4 v = *reinterpret_cast<const vec4*>(base + (long)key * ld + 2 * Dm + d4);
        const float pw = sS[key], dk = sD[key];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float vf = Act<T>::to_f32(v[e]);
            acc[e] = fmaf(pw, vf, acc[e]);
            acu[e] = fmaf(dk, vf, acu[e]);
        }
    }
    *reinterpret_cast<f32x4*>(sRed + ph * 64 + d4) = acc;
    *reinterpret_cast<f32x4*>(sRed + 1024 + ph * 64 + d4) = acu;
    __syncthreads();
    if (tid < 64) {
        float o = 0.f, uu = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { o += sRed[k * 64 + tid]; uu += sRed[1024 + k * 64 + tid]; }
        o /= l;
        const long at = ((long)b * Tn + qrow) * Dm + h * 64 + tid;
        out[at] = Act<T>::from_f32(o - bias_scale[h] * uu);
        u_out[at] = Act<T>::from_f32(uu);
        osm_out[at] = Act<T>::from_f32(o);
    }
}

// Its backward when no other query row of the block has a gradient: dV_k = (p_k - dist_scale_h d_k) dO with dist_scale_h = bias_scale_h / running_mean_h folded
// into d_k's factor below, dS_k = p_k (dO . v_k - dO . Osm), dK_k = dS_k q / 8, dQ = sum_k dS_k k_k / 8 on the query row alone, and
// dbs[b][h] = - dO . U: summed over the bags it is the gradient of bias_scale_h.
template <typename T>
__global__ void __launch_bounds__(256) attn_row_alibi_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ osm, const T* __restrict__ u, const T* __restrict__ dout,
                                                                 const float* __restrict__ lse, const float* __restrict__ coords, const float* __restrict__ bias_scale,
                                                                 const float* __restrict__ inv_rm, T* __restrict__ dqkv, float* __restrict__ dbs, int Tn, int H, int qrow) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    extern __shared__ __attribute__((aligned(16))) float sS[];          // [Tn] dS | [16][64] partial dq
    float* sRed = sS + ((Tn + 3) & ~3);
    const int tid = threadIdx.x;
    const int h = blockIdx.x, b = blockIdx.y;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;
    T* dbase = dqkv + (long)b * Tn * ld + h * 64;
    const float* cb = coords + (long)b * Tn * 2;
    const float xq = cb[2 * qrow], yq = cb[2 * qrow + 1], dsc = bias_scale[h] * inv_rm[h];
    const int sub = tid & 7, kk = tid >> 3;
    float qc[8], gc[8];
    float Dq = 0.f, gu = 0.f;
    {
        const long at = ((long)b * Tn + qrow) * Dm + h * 64 + sub * 8;
        const vec8 qv = *reinterpret_cast<const vec8*>(base + (long)qrow * ld + sub * 8);
        const vec8 gv = *reinterpret_cast<const vec8*>(dout + at);
        const vec8 ov = *reinterpret_cast<const vec8*>(osm + at);
        const vec8 uv = *reinterpret_cast<const vec8*>(u + at);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qc[e] = Act<T>::to_f32(qv[e]);
            gc[e] = Act<T>::to_f32(gv[e]);
            Dq = fmaf(gc[e], Act<T>::to_f32(ov[e]), Dq);
            gu = fmaf(gc[e], Act<T>::to_f32(uv[e]), gu);
        }
        Dq = sum8_dpp(Dq);
        gu = sum8_dpp(gu);
    }
    if (tid == 0) dbs[(long)b * H + h] = -gu;
    const float sc = 0.125f * 1.44269504088896340736f;
    const float L = lse[((long)b * H + h) * Tn + qrow];
    for (int key = kk; key < Tn; key += 32) {
        const T* krow = base + (long)key * ld + Dm + sub * 8;
        const vec8 kv = *reinterpret_cast<const vec8*>(krow), vv = *reinterpret_cast<const vec8*>(krow + Dm);
        float s0 = 0.f, p0 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s0 = fmaf(qc[e], Act<T>::to_f32(kv[e]), s0);
            p0 = fmaf(gc[e], Act<T>::to_f32(vv[e]), p0);
        }
        s0 = sum8_dpp(s0);
        p0 = sum8_dpp(p0);
        const float pk = __builtin_amdgcn_exp2f(s0 * sc - L);
        const float dx = xq - cb[2 * key], dy = yq - cb[2 * key + 1];
        const float dS = pk * (p0 - Dq), pw = pk - dsc * dist_sqrt(dx * dx + dy * dy), dk = dS * 0.125f;
        if (sub == 0) sS[key] = dS;
        T* drow = dbase + (long)key * ld + sub * 8;
        vec8 wq, wk, wv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            wq[e] = (T)0.f;
            wk[e] = Act<T>::from_f32(dk * qc[e]);
            wv[e] = Act<T>::from_f32(pw * gc[e]);
        }
        if (key != qrow) *reinterpret_cast<vec8*>(drow) = wq;
        *reinterpret_cast<vec8*>(drow + Dm) = wk;
        *reinterpret_cast<vec8*>(drow + 2 * Dm) = wv;
    }
    __syncthreads();
    const int d4 = (tid & 15) * 4, ph = tid >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int key = ph; key < Tn; key += 16) {
        const vec4 kv = *reinterpret_cast<const vec4*>(base + (long)key * ld + Dm + d4);
        const float dS = sS[key];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(dS, Act<T>::to_f32(kv[e]), acc[e]);
    }
    *reinterpret_cast<f32x4*>(sRed + ph * 64 + d4) = acc;
    __syncthreads();
    if (tid < 64) {
        float dq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dq += sRed[k * 64 + tid];
        dbase[(long)qrow * ld + tid] = Act<T>::from_f32(dq * 0.125f);
    }
}

}  // namespace amds

using namespace amds;

extern "C" int amds_attention(const void* qkv, void* out, int B, int T, int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out, "amds_attention: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H), "amds_attention: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 4.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, false>), grid, block, 0, st, (const f16*)qkv, (f16*)out, T, H, nullptr, nullptr, nullptr);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, false>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out, T, H, nullptr, nullptr, nullptr);
    else { set_error("amds_attention: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel");
    return AMDS_OK;
}

extern "C" int amds_attention_row(const void* q, long ldq, const void* qkv, void* out, long ldo, int B, int T, int H, int dtype, void* stream) {
    AMDS_REQUIRE(q && qkv && out, "amds_attention_row: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && T <= 32768 && H > 0 && H <= 65535 && B <= 65535 && ldq >= H * 64 && ldo >= H * 64 && ldq % 8 == 0, "amds_attention_row: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)((T + 3) & ~3) + 16 * 64 + 8) * 4;
    static bool attr_set[2] = {false, false};
    const int ti = dtype == AMDS_F16 ? 0 : 1;
    if (dtype != AMDS_F16 && dtype != AMDS_BF16) { set_error("amds_attention_row: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    if (!attr_set[ti]) {
        if (ti == 0) AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        else AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set[ti] = true;
    }
    if (ti == 0) hipLaunchKernelGGL((attn_row_kernel<f16>), dim3(H, B), dim3(256), lds, st, (const f16*)q, ldq, (const f16*)qkv, (f16*)out, ldo, T, H);
    else hipLaunchKernelGGL((attn_row_kernel<bf16>), dim3(H, B), dim3(256), lds, st, (const bf16*)q, ldq, (const bf16*)qkv, (bf16*)out, ldo, T, H);
    AMDS_LAUNCH_CHECK("attn_row_kernel");
    return AMDS_OK;
}

extern "C" int amds_attention_row_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int qrow, int dtype, float p, uint64_t seed,
                                            uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(qkv && out && lse, "amds_attention_row_fwd_train: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && T <= 32768 && H > 0 && H <= 65535 && B <= 65535 && qrow >= 0 && qrow < T && p >= 0.f && p < 1.f,
                 "amds_attention_row_fwd_train: bad arguments B=%d T=%d H=%d row=%d p=%f", B, T, H, qrow, p);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_attention_row_fwd_train: bad dtype %d", dtype);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)((T + 3) & ~3) + 16 * 64 + 8) * 4;
    const long Dm = (long)H * 64, ldq = (long)T * 3 * Dm, ldo = (long)T * Dm;
    const uint32_t thr = p > 0.f ? drop_thr16(p) : 0;
    const float ks = p > 0.f ? drop_scale(thr) : 1.f;
    static bool attr[4] = {false, false, false, false};
#define AMDS_ROW_FWD(TT, DR, IDX)                                                                                                                         \
    do {                                                                                                                                                  \
        if (!attr[IDX]) { AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_kernel<TT, DR>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); attr[IDX] = true; } \
        hipLaunchKernelGGL((attn_row_kernel<TT, DR>), dim3(H, B), dim3(256), lds, st, (const TT*)qkv + (long)qrow * 3 * Dm, ldq, (const TT*)qkv,          \
                           (TT*)out + (long)qrow * Dm, ldo, T, H, lse, qrow, seed, stream_id, thr, ks);                                                   \
    } while (0)
    if (dtype == AMDS_F16) { if (p > 0.f) AMDS_ROW_FWD(f16, true, 0); else AMDS_ROW_FWD(f16, false, 1); }
    else { if (p > 0.f) AMDS_ROW_FWD(bf16, true, 2); else AMDS_ROW_FWD(bf16, false, 3); }
#undef AMDS_ROW_FWD
    AMDS_LAUNCH_CHECK("attn_row_kernel<train>");
    return AMDS_OK;
}

extern "C" int amds_attention_row_bwd_train(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T, int H, int qrow,
                                            int dtype, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(qkv && out && dout && lse && dqkv, "amds_attention_row_bwd_train: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && T <= 32768 && H > 0 && H <= 65535 && B <= 65535 && qrow >= 0 && qrow < T && p >= 0.f && p < 1.f,
                 "amds_attention_row_bwd_train: bad arguments B=%d T=%d H=%d row=%d p=%f", B, T, H, qrow, p);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_attention_row_bwd_train: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)((T + 3) & ~3) + 16 * 64) * 4;
    const long Dm = (long)H * 64, ldo = (long)T * Dm;
    const uint32_t thr = p > 0.f ? drop_thr16(p) : 0;
    const float ks = p > 0.f ? drop_scale(thr) : 1.f;
    static bool attr[4] = {false, false, false, false};
#define AMDS_ROW_BWD(TT, DR, IDX)                                                                                                                         \
    do {                                                                                                                                                  \
        if (!attr[IDX]) { AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_bwd_kernel<TT, DR>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); attr[IDX] = true; } \
        hipLaunchKernelGGL((attn_row_bwd_kernel<TT, DR>), dim3(H, B), dim3(256), lds, st, (const TT*)qkv, (const TT*)out + (long)qrow * Dm, ldo,          \
                           (const TT*)dout + (long)qrow * Dm, ldo, lse, (TT*)dqkv, T, H, qrow, seed, stream_id, thr, ks);                                 \
    } while (0)
    if (dtype == AMDS_F16) { if (p > 0.f) AMDS_ROW_BWD(f16, true, 0); else AMDS_ROW_BWD(f16, false, 1); }
    else { if (p > 0.f) AMDS_ROW_BWD(bf16, true, 2); else AMDS_ROW_BWD(bf16, false, 3); }
#undef AMDS_ROW_BWD
    AMDS_LAUNCH_CHECK("attn_row_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_attention_row_alibi_fwd_train(const void* qkv, const float* coords, const float* inv_running_mean, const float* bias_scale, void* out, void* u,
                                                  void* osm, float* lse, int B, int T, int H, int qrow, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && coords && inv_running_mean && bias_scale && out && u && osm && lse, "amds_attention_row_alibi_fwd_train: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && T <= 16384 && H > 0 && H <= 65535 && B <= 65535 && qrow >= 0 && qrow < T, "amds_attention_row_alibi_fwd_train: bad arguments B=%d T=%d H=%d row=%d",
                 B, T, H, qrow);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_attention_row_alibi_fwd_train: bad dtype %d", dtype);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)2 * ((T + 3) & ~3) + 2 * 16 * 64 + 8) * 4;
    static bool attr[2] = {false, false};
    if (dtype == AMDS_F16) {
        if (!attr[0]) { AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_alibi_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); attr[0] = true; }
        hipLaunchKernelGGL((attn_row_alibi_kernel<f16>), dim3(H, B), dim3(256), lds, st, (const f16*)qkv, coords, inv_running_mean, bias_scale, (f16*)out, (f16*)u, (f16*)osm, lse, T, H, qrow);
    } else {
        if (!attr[1]) { AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_alibi_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); attr[1] = true; }
        hipLaunchKernelGGL((attn_row_alibi_kernel<bf16>), dim3(H, B), dim3(256), lds, st, (const bf16*)qkv, coords, inv_running_mean, bias_scale, (bf16*)out, (bf16*)u, (bf16*)osm, lse, T, H, qrow);
    }
    AMDS_LAUNCH_CHECK("attn_row_alibi_kernel");
    return AMDS_OK;
}

extern "C" int amds_attention_row_alibi_bwd_train(const void* qkv, const void* osm, const void* u, const void* dout, const float* lse, const float* coords,
                                                  const float* bias_scale, const float* inv_running_mean, void* dqkv, float* dbs, int B, int T, int H, int qrow, int dtype,
                                                  void* stream) {
    AMDS_REQUIRE(qkv && osm && u && dout && lse && coords && bias_scale && inv_running_mean && dqkv && dbs, "amds_attention_row_alibi_bwd_train: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && T <= 32768 && H > 0 && H <= 65535 && B <= 65535 && qrow >= 0 && qrow < T, "amds_attention_row_alibi_bwd_train: bad arguments B=%d T=%d H=%d row=%d",
                 B, T, H, qrow);
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_attention_row_alibi_bwd_train: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)((T + 3) & ~3) + 16 * 64) * 4;
    static bool attr[2] = {false, false};
    if (dtype == AMDS_F16) {
        if (!attr[0]) { AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_alibi_bwd_kernel<f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); attr[0] = true; }
        hipLaunchKernelGGL((attn_row_alibi_bwd_kernel<f16>), dim3(H, B), dim3(256), lds, st, (const f16*)qkv, (const f16*)osm, (const f16*)u, (const f16*)dout, lse, coords, bias_scale,
                           inv_running_mean, (f16*)dqkv, dbs, T, H, qrow);
    } else {
        if (!attr[1]) { AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_row_alibi_bwd_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); attr[1] = true; }
        hipLaunchKernelGGL((attn_row_alibi_bwd_kernel<bf16>), dim3(H, B), dim3(256), lds, st, (const bf16*)qkv, (const bf16*)osm, (const bf16*)u, (const bf16*)dout, lse, coords,
                           bias_scale, inv_running_mean, (bf16*)dqkv, dbs, T, H, qrow);
    }
    AMDS_LAUNCH_CHECK("attn_row_alibi_bwd_kernel");
    return AMDS_OK;
}

extern "C" int amds_attention_alibi(const void* qkv, const float* coords, const float* head_scale, void* out, int B, int T,
                                    int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out && coords && head_scale, "amds_attention_alibi: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H), "amds_attention_alibi: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 6.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, true, bf16>), grid, block, 0, st, (const f16*)qkv, (bf16*)out, T, H, coords, head_scale, nullptr);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, true, bf16>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out, T, H, coords, head_scale, nullptr);
    else { set_error("amds_attention_alibi: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel<alibi>");
    return AMDS_OK;
}

// forward that also stores L[b][h][q] = log2(sum_k exp2(s_qk * log2(e)/8)) for amds_attention_bwd
extern "C" int amds_attention_fwd_lse(const void* qkv, void* out, float* lse, int B, int T, int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out && lse, "amds_attention_fwd_lse: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H), "amds_attention_fwd_lse: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 4.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, false>), grid, block, 0, st, (const f16*)qkv, (f16*)out, T, H, nullptr, nullptr, lse);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, false>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out, T, H, nullptr, nullptr, lse);
    else { set_error("amds_attention_fwd_lse: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel<lse>");
    return AMDS_OK;
}

// Training forward of the ALiBi attention (reference vision_tranformer.py:42-74 in train mode): head_scale = 1 / running_mean
// (already updated by the caller, :24-29), out = softmax(q k^T / 8) v - bias_scale_h * U with U = sum_k (dist / running_mean) v.
// Saves what the backward needs: L = log2-sum-exp per query, U and the softmax part Osm (both bf16).
extern "C" int amds_attention_alibi_fwd_train(const void* qkv, const float* coords, const float* inv_running_mean, const float* bias_scale,
                                              void* out_bf16, void* u_bf16, void* osm_bf16, float* lse, int B, int T, int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && coords && inv_running_mean && bias_scale && out_bf16 && u_bf16 && osm_bf16 && lse, "amds_attention_alibi_fwd_train: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H), "amds_attention_alibi_fwd_train: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 6.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, true, bf16>), grid, block, 0, st, (const f16*)qkv, (bf16*)out_bf16, T, H, coords, inv_running_mean, lse, bias_scale, (bf16*)u_bf16, (bf16*)osm_bf16);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, true, bf16>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out_bf16, T, H, coords, inv_running_mean, lse, bias_scale, (bf16*)u_bf16, (bf16*)osm_bf16);
    else { set_error("amds_attention_alibi_fwd_train: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel<alibi,train>");
    return AMDS_OK;
}

// the same with out / U / Osm in the operand type itself (dtype = AMDS_F16: the training step at float32_matmul_precision "high"; AMDS_BF16 = the entry above)
int amds::attention_alibi_fwd_train_dt(const void* qkv, const float* coords, const float* inv_running_mean, const float* bias_scale, void* out, void* u, void* osm,
                                       float* lse, int B, int T, int H, int dtype, void* stream) {
    if (dtype != AMDS_F16) return amds_attention_alibi_fwd_train(qkv, coords, inv_running_mean, bias_scale, out, u, osm, lse, B, T, H, dtype, stream);
    AMDS_REQUIRE(qkv && coords && inv_running_mean && bias_scale && out && u && osm && lse, "amds_attention_alibi_fwd_train: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H), "amds_attention_alibi_fwd_train: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 6.0 * B * H * (double)T * T * 64, st);
    hipLaunchKernelGGL((attn_flash_kernel<f16, true, f16>), grid, block, 0, st, (const f16*)qkv, (f16*)out, T, H, coords, inv_running_mean, lse, bias_scale, (f16*)u, (f16*)osm);
    AMDS_LAUNCH_CHECK("attn_flash_kernel<alibi,train,f16>");
    return AMDS_OK;
}

// `mask != None` forward of the reference (vision_tranformer.py:355-381; pinned by the reference's tests/test_model.py:28-32): pad u8 [B][T]
// with the class token included at t = 0 (never padded).  See the kernel comment for the literal blocking rule.
extern "C" int amds_attention_masked(const void* qkv, const uint8_t* pad, void* out, int B, int T, int H, int mask_heads, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out && pad, "amds_attention_masked: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H) && mask_heads > 0 && mask_heads <= H, "amds_attention_masked: bad shape B=%d T=%d H=%d mask_heads=%d", B, T, H, mask_heads);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 4.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, false, f16, true>), grid, block, 0, st, (const f16*)qkv, (f16*)out, T, H, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pad, 0, 0, 0, 1.f, mask_heads);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, false, bf16, true>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out, T, H, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pad, 0, 0, 0, 1.f, mask_heads);
    else { set_error("amds_attention_masked: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel<mask>");
    return AMDS_OK;
}

extern "C" int amds_attention_alibi_masked(const void* qkv, const float* coords, const float* head_scale, const uint8_t* pad, void* out, int B, int T,
                                           int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out && coords && head_scale && pad, "amds_attention_alibi_masked: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H), "amds_attention_alibi_masked: bad shape B=%d T=%d H=%d", B, T, H);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    ProfScope prof(PROF_ATTN, 6.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, true, bf16, true>), grid, block, 0, st, (const f16*)qkv, (bf16*)out, T, H, coords, head_scale, nullptr, nullptr, nullptr, nullptr, pad, 0, 0, 0, 1.f);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, true, bf16, true>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out, T, H, coords, head_scale, nullptr, nullptr, nullptr, nullptr, pad, 0, 0, 0, 1.f);
    else { set_error("amds_attention_alibi_masked: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel<alibi,mask>");
    return AMDS_OK;
}

// Training forward of nn.MultiheadAttention with dropout p on the attention probabilities (vision_tranformer.py:191: the `dropout`
// constructor argument reaches MultiheadAttention; train mode): out = drop(P) v; lse as amds_attention_fwd_lse.  p = 0 is that call.
extern "C" int amds_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int dtype, float p, uint64_t seed,
                                        uint32_t stream_id, void* stream) {
    if (p == 0.f) return amds_attention_fwd_lse(qkv, out, lse, B, T, H, dtype, stream);
    AMDS_REQUIRE(qkv && out && lse, "amds_attention_fwd_train: null pointer");
    AMDS_REQUIRE(B >= 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && FA_SPAN_OK(T, H) && p > 0.f && p < 1.f, "amds_attention_fwd_train: bad arguments B=%d T=%d H=%d p=%f", B, T, H, p);
    if (B == 0) return AMDS_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((T + 127) / 128, H, B), block(256);
    const uint32_t thr = drop_thr16(p);
    const float ks = drop_scale(thr);
    ProfScope prof(PROF_ATTN, 4.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_F16) hipLaunchKernelGGL((attn_flash_kernel<f16, false, f16, false, true>), grid, block, 0, st, (const f16*)qkv, (f16*)out, T, H, nullptr, nullptr, lse, nullptr, nullptr, nullptr, nullptr, seed, stream_id, thr, ks);
    else if (dtype == AMDS_BF16) hipLaunchKernelGGL((attn_flash_kernel<bf16, false, bf16, false, true>), grid, block, 0, st, (const bf16*)qkv, (bf16*)out, T, H, nullptr, nullptr, lse, nullptr, nullptr, nullptr, nullptr, seed, stream_id, thr, ks);
    else { set_error("amds_attention_fwd_train: bad dtype %d", dtype); return AMDS_ERR_INVALID; }
    AMDS_LAUNCH_CHECK("attn_flash_kernel<drop>");
    return AMDS_OK;
}
