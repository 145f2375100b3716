// attention_train.hip -- backward of softmax(q k^T / 8) v for the MIL training step (head_dim 64, any T), flash style:
// nothing T x T is stored; the forward saves only L = log2-sum-exp per query (amds_attention_fwd_lse), the backward
// recomputes S and P tile by tile.  Reference op: nn.MultiheadAttention inside SelfAttention
// (src/stamp/modeling/models/vision_tranformer.py:191, 217-227) differentiated by autograd in LitTileClassifier._step
// (src/stamp/modeling/models/__init__.py:239-279).
//   P   = exp2(s * c - L[q]),  s = q . k,  c = log2(e) / 8
//   dV  = P^T dO ;  dP = dO V^T ;  dS = P o (dP - Dq),  Dq = rowsum(dO o O) ;  dQ = dS K / 8 ;  dK = dS^T Q / 8
// Two kernels, both built from the forward's MFMA layout rules (W/K-type operand rows in LDS row-major XOR-swizzled,
// "transposed" operands as a skewed [d][token] LDS image whose token order inside 16-groups has bits 2<->3 swapped so
// that a lane's accumulator registers ARE the next MFMA's B fragment -- no cross-lane data movement anywhere):
//   attn_bwd_dkdv_kernel : one workgroup = 128 keys (lane = key), loops over query tiles  -> dK, dV
//   attn_bwd_dq_kernel   : one workgroup = 128 queries (lane = query), loops over key tiles -> dQ
#include "common.h"

namespace amds {

constexpr int BT_TILE = 64;                        // tokens per streamed tile
constexpr int BT_RS = 192;                         // row stride (bytes) of the transposed images: 12 slots + 16 B skew / 8 rows
constexpr int BT_ROW_BYTES = BT_TILE * 128;        // row-major image: 64 tokens x 128 B
constexpr int BT_TR_BYTES = 64 * BT_RS + 8 * 16;   // transposed image: 64 feature rows x 64 tokens (+ skew)

__device__ __forceinline__ int perm16(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

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
// ---------------------------------------------------------------------------------------------------------------------
// ALIBI: dV^T += dO^T (P - c_h D) with D = cdist(coords) and c_h = bias_scale_h / running_mean_h (the value path of the
// post-softmax distance bias, vision_tranformer.py:60-72); dS, dK, dQ are those of the softmax part alone.
// DROP: dropout on the attention probabilities (amds_attention_fwd_train): with M = keep-mask * 1/(1-p) regenerated from the same
// counters, dV = (M o P)^T dO, dP = M o (dO V^T), dS = P o (dP - Dq) where Dq = rowsum(dO o O) still holds for O = (M o P) V.
template <typename T, bool ALIBI = false, bool DROP = false>
__global__ void __launch_bounds__(256, 2) attn_bwd_dkdv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ dq_sum,
                                                               T* __restrict__ dqkv, int Tn, int H, const float* __restrict__ coords = nullptr,
                                                               const float* __restrict__ dist_scale = nullptr, uint64_t seed = 0,
                                                               uint32_t drop_stream = 0, uint32_t thr16 = 0, float keep_scale = 1.f) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    // ONE LDS stage (42 KB) and <= 256 registers: two workgroups per CU.  With two stages (84 KB, 288-304 registers) a CU held a single
    // workgroup -- one wave per SIMD, every MFMA -> exp2 -> MFMA chain exposed -- and the kernel ran at 385 TFLOP/s against 620-630 for the
    // forward and the dQ kernel, which always had two.  The next tile still travels global -> registers under the current tile's compute; it
    // is written to the stage between two barriers, and the partner workgroup's MFMAs fill those.
    constexpr int STAGE = 2 * BT_ROW_BYTES + 2 * BT_TR_BYTES + 2 * BT_TILE * 4 + ((ALIBI || DROP) ? 2 * BT_TILE * 4 : 0);
    __shared__ __attribute__((aligned(16))) char smem[STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, kblk = blockIdx.x;
    const float* cbase = ALIBI ? coords + (long)b * Tn * 2 : nullptr;
    const int Dm = H * 64;
    const long ld = 3L * Dm;
    const T* base = qkv + (long)b * Tn * ld + h * 64;           // q at +0, k at +Dm, v at +2Dm
    const T* dobase = dout + (long)b * Tn * Dm + h * 64;
    const float* lrow = lse + ((long)b * H + h) * Tn;
    const float* drow = dq_sum + ((long)b * H + h) * Tn;
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

    // staging: per tile 64 queries x (Q row 128 B + dO row 128 B); thread -> (token pair, 8-wide d chunk)
    const int pr = tid >> 3, ch = tid & 7;          // pair 0..31 -> tokens 2pr, 2pr+1
    vec8 q0, q1, g0, g1;
    float lreg = 0.f, dreg = 0.f, cxreg = 0.f, cyreg = 0.f;
    uint32_t rkreg = 0;
    float xk = 0.f, yk = 0.f, ch_ = 0.f;
    if constexpr (ALIBI) { xk = cbase[(long)keyc * 2]; yk = cbase[(long)keyc * 2 + 1]; ch_ = dist_scale[h]; }
    auto stage_load = [&](int j) {
        const int t0 = j * BT_TILE + pr * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { q0[e] = (T)0.f; q1[e] = (T)0.f; g0[e] = (T)0.f; g1[e] = (T)0.f; }
        if (t0 < Tn) { q0 = *reinterpret_cast<const vec8*>(base + (long)t0 * ld + ch * 8); g0 = *reinterpret_cast<const vec8*>(dobase + (long)t0 * Dm + ch * 8); }
        if (t0 + 1 < Tn) { q1 = *reinterpret_cast<const vec8*>(base + (long)(t0 + 1) * ld + ch * 8); g1 = *reinterpret_cast<const vec8*>(dobase + (long)(t0 + 1) * Dm + ch * 8); }
        lreg = dreg = 0.f;
        if (tid < BT_TILE && j * BT_TILE + tid < Tn) { lreg = lrow[j * BT_TILE + tid]; dreg = drow[j * BT_TILE + tid]; }
        if constexpr (ALIBI) {
            cxreg = cyreg = 0.f;
            if (tid < BT_TILE && j * BT_TILE + tid < Tn) { cxreg = cbase[(long)(j * BT_TILE + tid) * 2]; cyreg = cbase[(long)(j * BT_TILE + tid) * 2 + 1]; }
        }
        if constexpr (DROP) {      // per-query row keys of this tile (ALiBi has no attention dropout: the slot is free)
            if (tid < BT_TILE) rkreg = drop_rowkey(seed, drop_stream, (uint64_t)(((long)b * H + h) * Tn + min(j * BT_TILE + tid, Tn - 1)));
        }
    };
    auto stage_store = [&](int buf) {
        char* sQ = smem + buf * STAGE;
        char* sG = sQ + BT_ROW_BYTES;
        char* sQt = sG + BT_ROW_BYTES;
        char* sGt = sQt + BT_TR_BYTES;
        float* sL = reinterpret_cast<float*>(sGt + BT_TR_BYTES);
        if constexpr (DROP) {
            if (tid < BT_TILE) reinterpret_cast<uint32_t*>(sL)[2 * BT_TILE + tid] = rkreg;
        }
        const int t0 = pr * 2;
        *reinterpret_cast<vec8*>(sQ + t0 * 128 + ((ch ^ ((t0 >> 1) & 7)) << 4)) = q0;
        *reinterpret_cast<vec8*>(sQ + (t0 + 1) * 128 + ((ch ^ (((t0 + 1) >> 1) & 7)) << 4)) = q1;
        *reinterpret_cast<vec8*>(sG + t0 * 128 + ((ch ^ ((t0 >> 1) & 7)) << 4)) = g0;
        *reinterpret_cast<vec8*>(sG + (t0 + 1) * 128 + ((ch ^ (((t0 + 1) >> 1) & 7)) << 4)) = g1;
        const int pos = perm16(t0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            typedef T vec2 __attribute__((ext_vector_type(2)));
            vec2 w;
            w[0] = q0[e]; w[1] = q1[e];
            *reinterpret_cast<vec2*>(sQt + (ch * 8 + e) * BT_RS + ch * 16 + pos * 2) = w;
            w[0] = g0[e]; w[1] = g1[e];
            *reinterpret_cast<vec2*>(sGt + (ch * 8 + e) * BT_RS + ch * 16 + pos * 2) = w;
        }
        if (tid < BT_TILE) { sL[tid] = lreg; sL[BT_TILE + tid] = dreg; }
        if constexpr (ALIBI) {
            if (tid < BT_TILE) { sL[2 * BT_TILE + tid] = cxreg; sL[3 * BT_TILE + tid] = cyreg; }
        }
    };

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    const float sc = 0.125f * 1.44269504088896340736f;
    const int swz = (l31 >> 1) & 7;

    const bool wave_live = kblk * 128 + wave * 32 < Tn;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int j = 0; j < ntile; ++j) {
        const int buf = 0;
        if (j + 1 < ntile) stage_load(j + 1);
        const char* sQ = smem + buf * STAGE;
        const char* sG = sQ + BT_ROW_BYTES;
        const char* sQt = sG + BT_ROW_BYTES;
        const char* sGt = sQt + BT_TR_BYTES;
        const float* sL = reinterpret_cast<const float*>(sGt + BT_TR_BYTES);
        if (wave_live)                                      // (a wave whose 32 keys all lie past the sequence only helps staging: T = 1025's ninth block)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = (half * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4);
                const vec8 qa = *reinterpret_cast<const vec8*>(sQ + off);
                const vec8 ga = *reinterpret_cast<const vec8*>(sG + off);
                s = Act<T>::mfma32(qa, kf[ks], s);
                dp = Act<T>::mfma32(ga, vf[ks], dp);
            }
            // lane = key, register r = query (r&3) + 8(r>>2) + 4hi of this 32-query half
            vec8 pf[2], df[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int qg = j * BT_TILE + ql;
                float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -sL[ql]));
                if (qg >= Tn || key >= Tn) p = 0.f;
                float dpr = dp[r];
                float pw = p;                                   // weight on v: P (dropped: M o P), minus the scaled distance for ALiBi
                if constexpr (DROP) {
                    const uint32_t bits = drop_pair_bits(reinterpret_cast<const uint32_t*>(sL)[2 * BT_TILE + ql], (uint32_t)key >> 1);
                    const float mk = drop_keep(bits, key & 1, thr16) ? keep_scale : 0.f;
                    dpr *= mk;
                    pw *= mk;
                }
                const float dsv = p * (dpr - sL[BT_TILE + ql]);
                if constexpr (ALIBI) {
                    const float ddx = sL[2 * BT_TILE + ql] - xk, ddy = sL[3 * BT_TILE + ql] - yk;
                    if (qg < Tn && key < Tn) pw -= ch_ * sqrtf(ddx * ddx + ddy * ddy);
                }
                pf[r >> 3][r & 7] = Act<T>::from_f32(pw);
                df[r >> 3][r & 7] = Act<T>::from_f32(dsv);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pos = half * 32 + ks * 16 + hi * 8;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + l31;
                    const int off = d * BT_RS + (d >> 3) * 16 + pos * 2;
                    const vec8 gt = *reinterpret_cast<const vec8*>(sGt + off);
                    const vec8 qt = *reinterpret_cast<const vec8*>(sQt + off);
                    dv[dt] = Act<T>::mfma32(gt, pf[ks], dv[dt]);
                    dk[dt] = Act<T>::mfma32(qt, df[ks], dk[dt]);
                }
            }
        }
        __syncthreads();                                  // every wave is done reading the stage
        if (j + 1 < ntile) stage_store(0);
        __syncthreads();
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
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool DROP = false>
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ dq_sum,
                                                             T* __restrict__ dqkv, int Tn, int H, uint64_t seed = 0, uint32_t drop_stream = 0,
                                                             uint32_t thr16 = 0, float keep_scale = 1.f) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int STAGE = 2 * BT_ROW_BYTES + BT_TR_BYTES;
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

    const int pr = tid >> 3, ch = tid & 7;
    vec8 k0, k1, v0, v1;
    auto stage_load = [&](int j) {
        const int t0 = j * BT_TILE + pr * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) { k0[e] = (T)0.f; k1[e] = (T)0.f; v0[e] = (T)0.f; v1[e] = (T)0.f; }
        if (t0 < Tn) { k0 = *reinterpret_cast<const vec8*>(base + (long)t0 * ld + Dm + ch * 8); v0 = *reinterpret_cast<const vec8*>(base + (long)t0 * ld + 2 * Dm + ch * 8); }
        if (t0 + 1 < Tn) { k1 = *reinterpret_cast<const vec8*>(base + (long)(t0 + 1) * ld + Dm + ch * 8); v1 = *reinterpret_cast<const vec8*>(base + (long)(t0 + 1) * ld + 2 * Dm + ch * 8); }
    };
    auto stage_store = [&](int buf) {
        char* sK = smem + buf * STAGE;
        char* sV = sK + BT_ROW_BYTES;
        char* sKt = sV + BT_ROW_BYTES;
        const int t0 = pr * 2;
        *reinterpret_cast<vec8*>(sK + t0 * 128 + ((ch ^ ((t0 >> 1) & 7)) << 4)) = k0;
        *reinterpret_cast<vec8*>(sK + (t0 + 1) * 128 + ((ch ^ (((t0 + 1) >> 1) & 7)) << 4)) = k1;
        *reinterpret_cast<vec8*>(sV + t0 * 128 + ((ch ^ ((t0 >> 1) & 7)) << 4)) = v0;
        *reinterpret_cast<vec8*>(sV + (t0 + 1) * 128 + ((ch ^ (((t0 + 1) >> 1) & 7)) << 4)) = v1;
        const int pos = perm16(t0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            typedef T vec2 __attribute__((ext_vector_type(2)));
            vec2 w;
            w[0] = k0[e]; w[1] = k1[e];
            *reinterpret_cast<vec2*>(sKt + (ch * 8 + e) * BT_RS + ch * 16 + pos * 2) = w;
        }
    };

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const float sc = 0.125f * 1.44269504088896340736f;
    const int swz = (l31 >> 1) & 7;

    const bool wave_live = qblk * 128 + wave * 32 < Tn;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int j = 0; j < ntile; ++j) {
        const int buf = j & 1;
        if (j + 1 < ntile) stage_load(j + 1);
        const char* sK = smem + buf * STAGE;
        const char* sV = sK + BT_ROW_BYTES;
        const char* sKt = sV + BT_ROW_BYTES;
        if (wave_live)                                      // (a wave whose 32 queries all lie past the sequence only helps staging: T = 1025's ninth block)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = (half * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz) << 4);
                const vec8 ka = *reinterpret_cast<const vec8*>(sK + off);
                const vec8 va = *reinterpret_cast<const vec8*>(sV + off);
                s = Act<T>::mfma32(ka, qf[ks], s);
                dp = Act<T>::mfma32(va, gf[ks], dp);
            }
            vec8 df[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kg = j * BT_TILE + half * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -Lq));
                if (kg >= Tn) p = 0.f;
                float dpr = dp[r];
                if constexpr (DROP) dpr *= drop_keep(drop_pair_bits(rowkey, (uint32_t)kg >> 1), kg & 1, thr16) ? keep_scale : 0.f;
                df[r >> 3][r & 7] = Act<T>::from_f32(p * (dpr - Dq));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pos = half * 32 + ks * 16 + hi * 8;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int d = dt * 32 + l31;
                    const vec8 kt = *reinterpret_cast<const vec8*>(sKt + d * BT_RS + (d >> 3) * 16 + pos * 2);
                    dq[dt] = Act<T>::mfma32(kt, df[ks], dq[dt]);
                }
            }
        }
        if (j + 1 < ntile) stage_store(buf ^ 1);
        __syncthreads();
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
    hipLaunchKernelGGL((attn_bwd_prep_kernel<T>), dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const T*)o, (const T*)dout, dq_sum, T_, H, total,
                       (const T*)u, bias_scale, dbs_part);
    AMDS_LAUNCH_CHECK("attn_bwd_prep_kernel");
    const dim3 grid((T_ + 127) / 128, H, B), block(256);
    if (!u && p_drop > 0.f) {
        const uint32_t thr = drop_thr16(p_drop);
        const float ks = drop_scale(thr);
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, false, true>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, nullptr, nullptr, seed, drop_stream, thr, ks);
        AMDS_LAUNCH_CHECK("attn_bwd_dkdv_kernel<drop>");
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, true>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, seed, drop_stream, thr, ks);
        AMDS_LAUNCH_CHECK("attn_bwd_dq_kernel<drop>");
        return AMDS_OK;
    }
    if (u) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, true>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, coords, dist_scale);
    else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, false>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H, nullptr, nullptr);
    AMDS_LAUNCH_CHECK("attn_bwd_dkdv_kernel");
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), grid, block, 0, st, (const T*)qkv, (const T*)dout, lse, dq_sum, (T*)dqkv, T_, H);
    AMDS_LAUNCH_CHECK("attn_bwd_dq_kernel");
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
        s += sqrtf(dx * dx + dy * dy);
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
