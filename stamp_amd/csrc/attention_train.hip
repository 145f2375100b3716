// attention_train.hip -- backward of softmax(q k^T / 8) v for the MIL training step (head_dim 64, any T), flash style:
// nothing T x T is stored; the forward saves only L = log2-sum-exp per query (amds_attention_fwd_lse), the backward
// recomputes S and P tile by tile.  Reference op: nn.MultiheadAttention inside SelfAttention
// (src/stamp/modeling/models/vision_tranformer.py:191, 217-227) differentiated by autograd in LitTileClassifier._step
// (src/stamp/modeling/models/__init__.py:239-279).
//   P   = exp2(s * c - L[q]),  s = q . k,  c = log2(e) / 8
//   dV  = P^T dO ;  dP = dO V^T ;  dS = P o (dP - Dq),  Dq = rowsum(dO o O) ;  dQ = dS K / 8 ;  dK = dS^T Q / 8
// Two kernels, both built from the forward's MFMA layout rules (a lane's accumulator registers ARE the next MFMA's B fragment -- no cross-lane data movement
// anywhere) and the forward's data path (round 6): tiles by buffer-form LDS-DMA into two row-major stages, "transposed" operands by ds_read_b64_tr_b16 transpose
// reads of those rows, one raw s_barrier per tile:
//   attn_bwd_dkdv2_kernel : one workgroup = 128 keys (lane = key), loops over query tiles  -> dK, dV      (150-158 registers, 34 KB: three workgroups per CU)
//   attn_bwd_dq2_kernel   : one workgroup = 128 queries (lane = query), loops over key tiles -> dQ         (120-124 registers, 32 KB: four)
// The kernels are VALU-issue-bound (per 64-token tile and wave ~400 VALU + 32 exp2 + 32 integer multiplies against 24-32 MFMAs): what round 6 removed is VALU work
// (one dropout hash per key PAIR, no bounds tests in full tiles, no transposing store pass) and registers (more waves per SIMD); profiles/r06_attn_bwd_ab.txt.
// The first forms (register staging, transposed LDS images) are kept as text in tools/ubench/attic/attention_first_forms/.
#include "common.h"

namespace amds {

constexpr int BT_TILE = 64;                        // tokens per streamed tile
constexpr int BT_ROW_BYTES = BT_TILE * 128;        // row-major image: 64 tokens x 128 B

// Dq[b][h][q] = sum_d dO[q][h*64+d] * Osm[q][h*64+d], Osm = the softmax part of the output (`o`; for plain attention the output).
// ALiBi (u != NULL): out = Osm - bias_scale_h U; dbs_part[b][h][q] = -sum_d dO U, whose sum over (b, q) is the gradient of
// bias_scale_h.
template <typename T>
__global__ void __launch_bounds__(256) attn_bwd_prep_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ dq_sum,
                                                            int Tn, int H, long total, const T* __restrict__ u = nullptr,
                                                            const float* __restrict__ bias_scale = nullptr, float* __restrict__ dbs_part = nullptr) {
    // eight (b, q, h) rows per wave: 8 lanes x 16 bytes per 64-element row, the row sum over its 8 lanes by three DPP butterflies (one wave per row with
    // 2-byte loads moved 1.3 TB/s: 104 us per call at 65 600 x 8 rows)
    typedef typename Act<T>::vec8 vec8;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + ((threadIdx.x & 63) >> 3);
    const int sub = threadIdx.x & 7;
    const bool live = row < total;
    const long r = live ? row : total - 1;
    const long bq = r / H;
    const int h = (int)(r - bq * H);
    const long off = bq * (long)H * 64 + h * 64 + sub * 8;
    const vec8 gv = *reinterpret_cast<const vec8*>(dout + off), ov = *reinterpret_cast<const vec8*>(o + off);
    float s = 0.f, gu = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(Act<T>::to_f32(ov[e]), Act<T>::to_f32(gv[e]), s);
    if (u) {
        const vec8 uv = *reinterpret_cast<const vec8*>(u + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) gu = fmaf(-Act<T>::to_f32(gv[e]), Act<T>::to_f32(uv[e]), gu);
    }
    auto sum8 = [](float v) {      // over the 8 lanes that share lane >> 3: xor 1, xor 2 (quad_perm), xor 7 (row_half_mirror)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
        return v;
    };
    s = sum8(s);
    if (u) gu = sum8(gu);
    if (sub == 0 && live) {
        const long b = bq / Tn;
        const int q = (int)(bq - b * Tn);
        dq_sum[(b * H + h) * (long)Tn + q] = s;
        if (u) dbs_part[(b * H + h) * (long)Tn + q] = gu;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV: lane = key.  Per query tile (64 queries, two 32-query halves):
//   S[i=query][j=key]  = Q_rows . K^T(regs)        dP[i=query][j=key] = dO_rows . V^T(regs)
//   dV^T[d][key] += dO^T[d][q] P[q][key]           dK^T[d][key] += Q^T[d][q] dS[q][key]
// ALIBI: dV^T += dO^T (P - c_h D) with D = cdist(coords) and c_h = bias_scale_h / running_mean_h (the value path of the post-softmax distance bias,
// vision_tranformer.py:60-72); dS, dK, dQ are those of the softmax part alone.
// DROP: dropout on the attention probabilities (amds_attention_fwd_train): with M = keep-mask * 1/(1-p) regenerated from the same counters, dV = (M o P)^T dO,
// dP = M o (dO V^T), dS = P o (dP - Dq) where Dq = rowsum(dO o O) still holds for O = (M o P) V.
// Data path:
//   * the Q and dO tiles go global -> LDS by buffer-form LDS-DMA (no staging registers; rows past the sequence are out of the descriptor's range and read as 0),
//     two stages of 17 KB, ONE barrier per tile;
//   * there are no transposed images: the dO^T / Q^T fragments of the two second products are ds_read_b64_tr_b16 transpose reads of the row-major images (a
//     16-lane group hands in 4 query rows x 32 B and each lane receives its feature column: 4 consecutive queries, exactly the order the accumulator registers
//     of S / dP hold them in);
//   * 16-byte chunk c of row q sits at chunk c ^ x(q), x(q) = bit 1 of q as bit 2 | bits 3..2 of q as bits 1..0: the row reads (ds_read_b128) AND the transpose
//     reads (a 32-lane group = 4 consecutive rows x 64 B: rows q, q + 2 in opposite halves of their 128 bytes) are conflict-free;
//   * 150-158 registers and 34 KB: THREE workgroups per CU (the first form: 208-245 registers, 42 KB, two).
// The transpose reads are inline asm (the builtin makes the compiler wait for the LDS-DMA in flight before every read, gemm_4w16.h) and carry their own
// s_waitcnt lgkmcnt(0): what the statement returns is there.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bt_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// eight transpose reads of one image (k-half ks in {0, 1} x feature half dt x query group w) + the wait; a[w][dt] = the lane's addresses for ks = 0
#define BT_TR8(out, a00, a01, a10, a11, imm)                                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %10 offset:%12\n\t"                                  \
                 "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %11 offset:%12\n\t"                                   \
                 "ds_read_b64_tr_b16 %4, %8 offset:%13\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"                                   \
                 "ds_read_b64_tr_b16 %6, %9 offset:%13\n\tds_read_b64_tr_b16 %7, %11 offset:%13\n\ts_waitcnt lgkmcnt(0)"               \
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])  \
                 : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "n"(imm), "n"((imm) + 2048)                                                  \
                 : "memory")

template <typename T, bool ALIBI = false, bool DROP = false>
__global__ void __launch_bounds__(256, 3) attn_bwd_dkdv2_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ dq_sum,
                                                                T* __restrict__ dqkv, int Tn, int H, const float* __restrict__ coords = nullptr,
                                                                const float* __restrict__ dist_scale = nullptr, uint64_t seed = 0,
                                                                uint32_t drop_stream = 0, uint32_t thr16 = 0, float keep_scale = 1.f) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int SC_OFF = 2 * BT_ROW_BYTES;                        // per-query scalars: L[64] | D[64] | row key[64] or (x, y)[64]
    constexpr int STAGE = SC_OFF + 4 * BT_TILE * 4;                 // 17 408 bytes
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, kblk = blockIdx.x;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;           // q at +0, k at +Dm, v at +2Dm
    const T* dobase = dout + (long)b * Tn * Dm + h * 64;
    const int ntile = (Tn + BT_TILE - 1) / BT_TILE;

    // this lane's key: K and V fragments stay in registers for the whole kernel
    const int key = kblk * 128 + wave * 32 + l31;
    const int keyc = min(key, Tn - 1);
    vec8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *reinterpret_cast<const vec8*>(base + (long)keyc * ld + Dm + (ks * 2 + hi) * 8);
        vf[ks] = *reinterpret_cast<const vec8*>(base + (long)keyc * ld + 2 * Dm + (ks * 2 + hi) * 8);
    }
    float xk = 0.f, yk = 0.f, ch_ = 0.f;
    if constexpr (ALIBI) { xk = coords[((long)b * Tn + keyc) * 2]; yk = coords[((long)b * Tn + keyc) * 2 + 1]; ch_ = dist_scale[h]; }

    // LDS-DMA: an image is 8 pieces of 8 rows (1 KB, lane-linear); wave w requests pieces w and w + 4 of both images (rows 32 apart share the swizzle: one lane
    // offset per image, the second piece in the scalar offset).  The per-query scalars travel the same way, 4 bytes per lane: L by wave 0, D by wave 1, the ALiBi
    // coordinates (x, y interleaved) by waves 2 and 3.  Everything past the sequence is out of its descriptor's range and reads as 0.
    const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)((((long)Tn - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dobase), 0, (int)((((long)Tn - 1) * Dm + 64) * 2), 0x00020000);
    const float* scal = wave == 0 ? lse + ((long)b * H + h) * Tn : wave == 1 ? dq_sum + ((long)b * H + h) * Tn : ALIBI ? coords + (long)b * Tn * 2 : lse;
    const int scal_bytes = wave < 2 ? Tn * 4 : ALIBI ? Tn * 8 : 0;
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scal), 0, scal_bytes, 0x00020000);
    const int drow_ = wave * 8 + (lane >> 3);
    const int dlc = (lane & 7) ^ bt_swz(drow_);
    const int voq = drow_ * (int)ld * 2 + dlc * 16, vog = drow_ * Dm * 2 + dlc * 16;
    auto tile_request = [&](int j, int buf) {
        char* sQ = smem + buf * STAGE;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            bufl16(rsrc_q, sQ + (wave + 4 * pc) * 1024, voq, (j * BT_TILE + 32 * pc) * (int)ld * 2);
            bufl16(rsrc_g, sQ + BT_ROW_BYTES + (wave + 4 * pc) * 1024, vog, (j * BT_TILE + 32 * pc) * Dm * 2);
        }
        // wave 0: L -> dwords 0..63, wave 1: D -> 64..127, waves 2, 3 (ALiBi): the tile's 128 coordinate dwords -> 128..255
        if (wave < 2 || ALIBI)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lptr_t)(sQ + SC_OFF + wave * 256), 4, lane * 4, (wave < 2 ? j * BT_TILE * 4 : j * BT_TILE * 8 + (wave - 2) * 256), 0, 0);
    };
    auto tile_rowkeys = [&](int j, int buf) {           // dropout: the tile's 64 row keys (hashes of the row index: nothing to wait for)
        if constexpr (DROP) {
            if (tid < BT_TILE)
                reinterpret_cast<uint32_t*>(smem + buf * STAGE + SC_OFF)[2 * BT_TILE + tid] =
                    drop_rowkey(seed, drop_stream, (uint64_t)(((long)b * H + h) * Tn + min(j * BT_TILE + tid, Tn - 1)));
        }
    };

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const float sc = 0.125f * 1.44269504088896340736f;
    // LDS addresses, one register each, moved between the two stages in place (+/- STAGE per tile):
    //   ra[ks]: this lane's 16-byte chunk ks*2 + hi of row l31 (the S / dP operand rows; + 4096 = the second half's rows, + 8192 = dO)
    //   tra[w][dt]: transpose reads -- lane i of 16-lane group g hands in (query row (i >> 2) of the 4, 8-byte piece i & 3 of its 32 bytes); g & 1 = which 16 features
    //   sla: the per-query scalars of queries 4 hi ..
    const unsigned lds0 = (unsigned)(size_t)smem;
    unsigned ra[4], tra[2][2], sla = lds0 + SC_OFF + 16 * hi;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ra[ks] = lds0 + l31 * 128 + (((ks * 2 + hi) ^ bt_swz(l31)) << 4);
    {
        const int i16 = lane & 15, gd = (lane >> 4) & 1;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int row = hi * 4 + w * 8 + (i16 >> 2);
                const int lc = dt * 4 + gd * 2 + ((i16 & 3) >> 1);
                tra[w][dt] = lds0 + row * 128 + ((lc ^ bt_swz(row)) << 4) + (i16 & 1) * 8;
            }
    }
    int flip = STAGE;

    const bool wave_live = kblk * 128 + wave * 32 < Tn;
    const int par = lane & 1;
    const uint32_t pair_g = DROP ? ((uint32_t)key >> 1) * 0x9E3779B1u : 0u;
    constexpr uint64_t EVEN = 0x5555555555555555ull;
    typedef const __attribute__((address_space(3))) char* lds_cp;
    tile_request(0, 0);
    tile_rowkeys(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) tile_request(j + 1, (j & 1) ^ 1);
        if (wave_live)                                      // (a wave whose 32 keys all lie past the sequence only requests: T = 1025's ninth block)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (j * BT_TILE + half * 32 >= Tn) continue;    // 32 queries past the sequence (T = 1025: the second half of the 17th tile)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 qa = *reinterpret_cast<const __attribute__((address_space(3))) vec8*>((lds_cp)(size_t)(ra[ks] + half * 4096));
                const vec8 ga = *reinterpret_cast<const __attribute__((address_space(3))) vec8*>((lds_cp)(size_t)(ra[ks] + half * 4096 + BT_ROW_BYTES));
                s = Act<T>::mfma32(qa, kf[ks], s);
                dp = Act<T>::mfma32(ga, vf[ks], dp);
            }
            const __attribute__((address_space(3))) float* sL = reinterpret_cast<const __attribute__((address_space(3))) float*>((lds_cp)(size_t)sla) + half * 32;
            // lane = key, register r = query (r&3) + 8(r>>2) + 4hi of this 32-query half; four registers at a time: two dropout hashes, four keep masks.  No bounds
            // tests: a query past the sequence reads as zeros (Q, dO, L, D: p = 1, dS = 0, and its dO / Q columns of the two products below are zero), a key past
            // it is a lane of its own whose dK / dV are never stored.
            vec8 pf[2], df[2];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                uint64_t keepm[4] = {0, 0, 0, 0};
                // Dropout bits are one hash per (query, key PAIR) and a lane is a key: lanes 2m and 2m + 1 would compute the same 16 hashes per half.  Instead the even
                // lane hashes the even registers' queries and the odd lane the odd ones; the two 16-bit compares of a hash become wave masks (SGPR pairs) that the scalar
                // unit re-deals to the two lanes of the pair (the lane with the even key reads the low half-word, its neighbour the high one).
                if constexpr (DROP) {
                    const __attribute__((address_space(3))) uint32_t* sRK = reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(sL) + 2 * BT_TILE + par;
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {             // this lane's register 4 r4 + 2 e2 + par
                        const uint32_t hsh = drop_mix(sRK[8 * r4 + 2 * e2] ^ pair_g);
                        const uint64_t k0 = __builtin_amdgcn_ballot_w64((hsh & 0xFFFFu) >= thr16);
                        const uint64_t k1 = __builtin_amdgcn_ballot_w64((hsh >> 16) >= thr16);
                        keepm[2 * e2] = (k0 & EVEN) | ((k1 & EVEN) << 1);              // hashed by the even lanes
                        keepm[2 * e2 + 1] = (k1 & ~EVEN) | ((k0 & ~EVEN) >> 1);        // hashed by the odd lanes
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * r4 + e;
                    const int ql = e + 8 * r4;                   // (+ 4 hi: in sla)
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -sL[ql]));
                    float dpr = dp[r];
                    float pw = p;                                   // weight on v: P (dropped: M o P), minus the scaled distance for ALiBi
                    if constexpr (DROP) {
                        const bool keep = __builtin_amdgcn_inverse_ballot_w64(keepm[e]);
                        dpr = keep ? dpr * keep_scale : 0.f;
                        pw = keep ? pw * keep_scale : 0.f;
                    }
                    const float dsv = p * (dpr - sL[BT_TILE + ql]);
                    if constexpr (ALIBI) {
                        const float ddx = sL[2 * BT_TILE + 2 * ql + 4 * hi + half * 32] - xk, ddy = sL[2 * BT_TILE + 2 * ql + 4 * hi + half * 32 + 1] - yk;   // (x, y) pairs, 8 bytes per query: sL already points 4 hi + 32 half floats in
                        pw -= ch_ * dist_sqrt(ddx * ddx + ddy * ddy);
                    }
                    pf[r >> 3][r & 7] = Act<T>::from_f32(pw);
                    df[r >> 3][r & 7] = Act<T>::from_f32(dsv);
                }
            }
            // fragment [ks*4 + dt*2 + w]: queries 16 ks + 8 w + 4 hi + 0..3 of the half, feature dt*32 + l31
            u32x2 tg[8], tq[8];
            if (half == 0) BT_TR8(tg, tra[0][0], tra[0][1], tra[1][0], tra[1][1], BT_ROW_BYTES); else BT_TR8(tg, tra[0][0], tra[0][1], tra[1][0], tra[1][1], BT_ROW_BYTES + 4096);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const vec8 gt = __builtin_bit_cast(vec8, u32x4{tg[ks * 4 + dt * 2][0], tg[ks * 4 + dt * 2][1], tg[ks * 4 + dt * 2 + 1][0], tg[ks * 4 + dt * 2 + 1][1]});
                    dv[dt] = Act<T>::mfma32(gt, pf[ks], dv[dt]);
                }
            if (half == 0) BT_TR8(tq, tra[0][0], tra[0][1], tra[1][0], tra[1][1], 0); else BT_TR8(tq, tra[0][0], tra[0][1], tra[1][0], tra[1][1], 4096);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const vec8 qt = __builtin_bit_cast(vec8, u32x4{tq[ks * 4 + dt * 2][0], tq[ks * 4 + dt * 2][1], tq[ks * 4 + dt * 2 + 1][0], tq[ks * 4 + dt * 2 + 1][1]});
                    dk[dt] = Act<T>::mfma32(qt, df[ks], dk[dt]);
                }
            __builtin_amdgcn_sched_barrier(0);               // (the halves are not interleaved: 168 registers hold one of them)
        }
        if (j + 1 < ntile) tile_rowkeys(j + 1, (j & 1) ^ 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ra[ks] += flip;
        tra[0][0] += flip; tra[0][1] += flip; tra[1][0] += flip; tra[1][1] += flip;
        sla += flip;
        flip = -flip;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the next tile has landed, this wave's reads of the current one are done
        __builtin_amdgcn_s_barrier();
    }
    if (key < Tn) {
        T* krow = dqkv + ((long)b * Tn + key) * ld + Dm + h * 64;
        T* vrow = krow + Dm;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 wk, wv;
#pragma unroll
                for (int e = 0; e < 4; ++e) { wk[e] = Act<T>::from_f32(dk[dt][4 * g + e] * 0.125f); wv[e] = Act<T>::from_f32(dv[dt][4 * g + e]); }
                *reinterpret_cast<vec4*>(krow + dt * 32 + 8 * g + 4 * hi) = wk;
                *reinterpret_cast<vec4*>(vrow + dt * 32 + 8 * g + 4 * hi) = wv;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ: lane = query.  Per key tile (64 keys, two 32-key halves):
//   S^T[i=key][j=query] = K_rows . Q^T(regs)      dP^T[i=key][j=query] = V_rows . dO^T(regs)
//   dQ^T[d][query] += K^T[d][key] dS^T[key][query]
// The data path of attn_bwd_dkdv2_kernel: K and V tiles by LDS-DMA into two 16 KB stages, the K^T fragments of the dQ product by transpose reads of the row-major
// K image, one barrier per tile, no staging registers.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool DROP = false>
__global__ void __launch_bounds__(256, 4) attn_bwd_dq2_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ dq_sum,
                                                                    T* __restrict__ dqkv, int Tn, int H, uint64_t seed = 0, uint32_t drop_stream = 0,
                                                                    uint32_t thr16 = 0, float keep_scale = 1.f) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int STAGE = 2 * BT_ROW_BYTES;                         // K rows | V rows
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, qblk = blockIdx.x;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;
    const T* dobase = dout + (long)b * Tn * Dm + h * 64;
    const int ntile = (Tn + BT_TILE - 1) / BT_TILE;

    const int q = qblk * 128 + wave * 32 + l31;
    const int qc = min(q, Tn - 1);
    vec8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const vec8*>(base + (long)qc * ld + (ks * 2 + hi) * 8);
        gf[ks] = *reinterpret_cast<const vec8*>(dobase + (long)qc * Dm + (ks * 2 + hi) * 8);
    }
    const float Lq = lse[((long)b * H + h) * Tn + qc], Dq = dq_sum[((long)b * H + h) * Tn + qc];
    uint32_t rowkey = 0;
    if constexpr (DROP) rowkey = drop_rowkey(seed, drop_stream, (uint64_t)(((long)b * H + h) * Tn + qc));

    // LDS-DMA (see attn_bwd_dkdv2_kernel): wave w requests 8-row pieces w and w + 4 of both images; key rows past the sequence read as 0
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + Dm), 0, (int)((((long)Tn - 1) * ld + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + 2 * Dm), 0, (int)((((long)Tn - 1) * ld + 64) * 2), 0x00020000);
    const int drow_ = wave * 8 + (lane >> 3);
    const int vo = drow_ * (int)ld * 2 + (((lane & 7) ^ bt_swz(drow_)) << 4);
    auto tile_request = [&](int j, int buf) {
        char* sK = smem + buf * STAGE;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            bufl16(rsrc_k, sK + (wave + 4 * pc) * 1024, vo, (j * BT_TILE + 32 * pc) * (int)ld * 2);
            bufl16(rsrc_v, sK + BT_ROW_BYTES + (wave + 4 * pc) * 1024, vo, (j * BT_TILE + 32 * pc) * (int)ld * 2);
        }
    };

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const float sc = 0.125f * 1.44269504088896340736f;
    const unsigned lds0 = (unsigned)(size_t)smem;
    unsigned ra[4], tra[2][2];                              // (attn_bwd_dkdv2_kernel: operand rows, transpose reads; moved between the stages in place)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ra[ks] = lds0 + l31 * 128 + (((ks * 2 + hi) ^ bt_swz(l31)) << 4);
    {
        const int i16 = lane & 15, gd = (lane >> 4) & 1;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int row = hi * 4 + w * 8 + (i16 >> 2);
                const int lc = dt * 4 + gd * 2 + ((i16 & 3) >> 1);
                tra[w][dt] = lds0 + row * 128 + ((lc ^ bt_swz(row)) << 4) + (i16 & 1) * 8;
            }
    }
    int flip = STAGE;
    typedef const __attribute__((address_space(3))) char* lds_cp;

    const bool wave_live = qblk * 128 + wave * 32 < Tn;
    uint32_t pair_g = DROP ? (uint32_t)(2 * hi) * 0x9E3779B1u : 0u;            // (key pair index of the lane's first key in the tile) x the hash's multiplier
    tile_request(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) tile_request(j + 1, (j & 1) ^ 1);
        const bool ragged = (j + 1) * BT_TILE > Tn;             // the last tile of a T that is no multiple of 64
        if (wave_live)                                      // (a wave whose 32 queries all lie past the sequence only requests: T = 1025's ninth block)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (j * BT_TILE + half * 32 >= Tn) continue;    // 32 keys past the sequence (T = 1025: the second half of the 17th tile)
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 ka = *reinterpret_cast<const __attribute__((address_space(3))) vec8*>((lds_cp)(size_t)(ra[ks] + half * 4096));
                const vec8 va = *reinterpret_cast<const __attribute__((address_space(3))) vec8*>((lds_cp)(size_t)(ra[ks] + half * 4096 + BT_ROW_BYTES));
                s = Act<T>::mfma32(ka, qf[ks], s);
                dp = Act<T>::mfma32(va, gf[ks], dp);
            }
            if (ragged) {       // keys past the sequence: p = exp2(-inf) = 0.  A branch taken once per workgroup: the empty asm statements cannot be speculated, so
                                // the block is not if-converted into 16 compares + 16 selects on every tile of a loop that is short of VALU issue slots
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (j * BT_TILE + half * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= Tn) s[r] = -INFINITY;
                    asm volatile("" : "+v"(s[r]));
                }
            }
            vec8 df[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {                       // registers r, r + 1 = an even key and its odd partner: one hash
                bool keep0 = true, keep1 = true;
                if constexpr (DROP) {
                    const uint32_t bits = drop_mix(rowkey ^ (pair_g + (uint32_t)(half * 16 + ((r & 3) >> 1) + 4 * (r >> 2)) * 0x9E3779B1u));
                    keep0 = drop_keep(bits, 0, thr16);
                    keep1 = drop_keep(bits, 1, thr16);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r + e], sc, -Lq));
                    float dpr = dp[r + e];
                    if constexpr (DROP) dpr = (e ? keep1 : keep0) ? dpr * keep_scale : 0.f;
                    df[r >> 3][(r & 7) + e] = Act<T>::from_f32(p * (dpr - Dq));
                }
            }
            u32x2 tk[8];                                    // fragment [ks*4 + dt*2 + w]: keys 16 ks + 8 w + 4 hi + 0..3 of the half, feature dt*32 + l31
            if (half == 0) BT_TR8(tk, tra[0][0], tra[0][1], tra[1][0], tra[1][1], 0); else BT_TR8(tk, tra[0][0], tra[0][1], tra[1][0], tra[1][1], 4096);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const vec8 kt = __builtin_bit_cast(vec8, u32x4{tk[ks * 4 + dt * 2][0], tk[ks * 4 + dt * 2][1], tk[ks * 4 + dt * 2 + 1][0], tk[ks * 4 + dt * 2 + 1][1]});
                    dq[dt] = Act<T>::mfma32(kt, df[ks], dq[dt]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DROP) pair_g += 32u * 0x9E3779B1u;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ra[ks] += flip;
        tra[0][0] += flip; tra[0][1] += flip; tra[1][0] += flip; tra[1][1] += flip;
        flip = -flip;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the next tile has landed, this wave's reads of the current one are done
        __builtin_amdgcn_s_barrier();
    }
    if (q < Tn) {
        T* qrow = dqkv + ((long)b * Tn + q) * ld + h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = Act<T>::from_f32(dq[dt][4 * g + e] * 0.125f);
                *reinterpret_cast<vec4*>(qrow + dt * 32 + 8 * g + 4 * hi) = w;
            }
    }
}

template <typename T>
static int launch_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* dq_sum, void* dqkv, int B, int T_, int H,
                           hipStream_t st, const void* u = nullptr, const float* coords = nullptr, const float* bias_scale = nullptr,
                           const float* dist_scale = nullptr, float* dbs_part = nullptr, float p_drop = 0.f, uint64_t seed = 0, uint32_t drop_stream = 0) {
    const long total = (long)B * T_ * H;
    AMDS_REQUIRE((long)T_ * 3 * H * 64 * 2 < (1L << 31), "attention backward: one bag's qkv rows (T=%d x %d bytes) exceed the 2 GB a buffer descriptor addresses", T_, 3 * H * 128);
    hipLaunchKernelGGL((attn_bwd_prep_kernel<T>), dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const T*)o, (const T*)dout, dq_sum, T_, H, total,
                       (const T*)u, bias_scale, dbs_part);
    AMDS_LAUNCH_CHECK("attn_bwd_prep_kernel");
    const dim3 grid((T_ + 127) / 128, H, B), block(256);
    if (!u && p_drop > 0.f) {
        const uint32_t thr = drop_thr16(p_drop);
        const float ks = drop_scale(thr);
        hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<T, false, true>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, nullptr, nullptr, seed, drop_stream, thr, ks);
        AMDS_LAUNCH_CHECK("attn_bwd_dkdv2_kernel<drop>");
        hipLaunchKernelGGL((attn_bwd_dq2_kernel<T, true>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, seed, drop_stream, thr, ks);
        AMDS_LAUNCH_CHECK("attn_bwd_dq2_kernel<drop>");
        return AMDS_OK;
    }
    if (u) hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<T, true>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, coords, dist_scale);
    else hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<T, false>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, nullptr, nullptr);
    AMDS_LAUNCH_CHECK("attn_bwd_dkdv2_kernel");
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<T>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H);
    AMDS_LAUNCH_CHECK("attn_bwd_dq2_kernel");
    return AMDS_OK;
}

// mean over b, q, k of |c[b,q] - c[b,k]| -- the statistic `_RunningMeanScaler` folds into its running mean in train mode
// (vision_tranformer.py:24-29 applied to torch.cdist(coords, coords), :59-60).  One wave per (b, q); deterministic two-stage sum.
__global__ void __launch_bounds__(256) cdist_rowsum_kernel(const float* __restrict__ coords, float* __restrict__ rowsum, int Tn, long rows) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const long b = r / Tn;
    const float* cb = coords + b * (long)Tn * 2;
    const float xq = coords[r * 2], yq = coords[r * 2 + 1];
    float s = 0.f;
    for (int k = lane; k < Tn; k += 64) {
        const float dx = xq - cb[k * 2], dy = yq - cb[k * 2 + 1];
        s += dist_sqrt(dx * dx + dy * dy);
    }
    s = wave_sum(s);
    if (lane == 0) rowsum[r] = s;
}

}  // namespace amds

using namespace amds;

extern "C" int amds_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dq_sum_ws, void* dqkv,
                                  int B, int T, int H, int dtype, void* stream) {
    AMDS_REQUIRE(qkv && out && dout && lse && dq_sum_ws && dqkv, "amds_attention_bwd: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535, "amds_attention_bwd: bad shape B=%d T=%d H=%d", B, T, H);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, 10.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_BF16) return launch_attn_bwd<bf16>(qkv, out, dout, lse, dq_sum_ws, dqkv, B, T, H, st);
    if (dtype == AMDS_F16) return launch_attn_bwd<f16>(qkv, out, dout, lse, dq_sum_ws, dqkv, B, T, H, st);
    set_error("amds_attention_bwd: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

// backward of amds_attention_fwd_train (same p, seed, stream_id as the forward)
extern "C" int amds_attention_bwd_train(const void* qkv, const void* out, const void* dout, const float* lse, float* dq_sum_ws, void* dqkv,
                                        int B, int T, int H, int dtype, float p, uint64_t seed, uint32_t stream_id, void* stream) {
    AMDS_REQUIRE(qkv && out && dout && lse && dq_sum_ws && dqkv, "amds_attention_bwd_train: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535 && p >= 0.f && p < 1.f, "amds_attention_bwd_train: bad arguments B=%d T=%d H=%d p=%f", B, T, H, p);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, 10.0 * B * H * (double)T * T * 64, st);
    if (dtype == AMDS_BF16) return launch_attn_bwd<bf16>(qkv, out, dout, lse, dq_sum_ws, dqkv, B, T, H, st, nullptr, nullptr, nullptr, nullptr, nullptr, p, seed, stream_id);
    if (dtype == AMDS_F16) return launch_attn_bwd<f16>(qkv, out, dout, lse, dq_sum_ws, dqkv, B, T, H, st, nullptr, nullptr, nullptr, nullptr, nullptr, p, seed, stream_id);
    set_error("amds_attention_bwd_train: bad dtype %d", dtype);
    return AMDS_ERR_INVALID;
}

extern "C" int amds_attention_alibi_bwd(const void* qkv, const void* osm, const void* u, const void* dout, const float* lse, const float* coords,
                                        const float* bias_scale, const float* dist_scale, float* dq_sum_ws, float* dbs_part, void* dqkv,
                                        int B, int T, int H, void* stream) {
    AMDS_REQUIRE(qkv && osm && u && dout && lse && coords && bias_scale && dist_scale && dq_sum_ws && dbs_part && dqkv, "amds_attention_alibi_bwd: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535, "amds_attention_alibi_bwd: bad shape B=%d T=%d H=%d", B, T, H);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, 10.0 * B * H * (double)T * T * 64, st);
    return launch_attn_bwd<bf16>(qkv, osm, dout, lse, dq_sum_ws, dqkv, B, T, H, st, u, coords, bias_scale, dist_scale, dbs_part);
}

// the same on fp16 tensors (dtype = AMDS_F16; AMDS_BF16 = the entry above)
int amds::attention_alibi_bwd_dt(const void* qkv, const void* osm, const void* u, const void* dout, const float* lse, const float* coords, const float* bias_scale,
                                 const float* dist_scale, float* dq_sum_ws, float* dbs_part, void* dqkv, int B, int T, int H, int dtype, void* stream) {
    if (dtype != AMDS_F16) return amds_attention_alibi_bwd(qkv, osm, u, dout, lse, coords, bias_scale, dist_scale, dq_sum_ws, dbs_part, dqkv, B, T, H, stream);
    AMDS_REQUIRE(qkv && osm && u && dout && lse && coords && bias_scale && dist_scale && dq_sum_ws && dbs_part && dqkv, "amds_attention_alibi_bwd: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && H > 0 && H <= 65535 && B <= 65535, "amds_attention_alibi_bwd: bad shape B=%d T=%d H=%d", B, T, H);
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(PROF_ATTN, 10.0 * B * H * (double)T * T * 64, st);
    return launch_attn_bwd<f16>(qkv, osm, dout, lse, dq_sum_ws, dqkv, B, T, H, st, u, coords, bias_scale, dist_scale, dbs_part);
}

extern "C" int amds_cdist_rowsum(const float* coords, float* rowsum, int B, int T, void* stream) {
    AMDS_REQUIRE(coords && rowsum, "amds_cdist_rowsum: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0, "amds_cdist_rowsum: bad shape");
    const long rows = (long)B * T;
    hipLaunchKernelGGL(cdist_rowsum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, coords, rowsum, T, rows);
    AMDS_LAUNCH_CHECK("cdist_rowsum_kernel");
    return AMDS_OK;
}
