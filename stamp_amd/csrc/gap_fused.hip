// gap_fused.hip -- gated-attention pooling of one or MANY bags of tile features in ONE launch (CHIEF slide encoder).
// Reference: src/stamp/encoding/encoder/chief.py:74-89 (CHIEFModel.forward), :255-275 (Attn_Net_Gated).
//   h   = relu(x Wfc^T + bfc)                              [N, L]
//   A_n = Wc (tanh(Wa h_n + ba) * sigmoid(Wb h_n + bb)) + bc
//   out = softmax_N(A) @ x                                 [F]   (pooled over the ORIGINAL features, chief.py:82)
// fp32 throughout (chief.py:117): every product runs on the exact-fp32 MFMA v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain, 157 TF peak).
//
// Decomposition: a workgroup (4 waves) owns a SLAB of 64 consecutive rows of one bag; a wave owns 16 of them and never exchanges activations with
// its neighbours -- the three stages chain through registers:
//   1. h^T = Wfc x^T        A operand = Wfc rows (LDS tile, shared by the 4 waves), B operand = the wave's 16 rows of x (straight from global:
//                           one 16-byte load per lane per 16 k).  All L/16 output blocks accumulate at once, so x is read once.  The C layout of
//                           16x16x4 (lane = x row, lane >> 4 and register = 4 consecutive L indices) IS a valid B-operand layout for stage 2 with
//                           the k order permuted inside each group of 16 -- h never leaves the register file.
//   2. [a; b]^T = [Wa; Wb] h^T, 16 gate channels d at a time (runtime loop), the matching rows of Wa and Wb staged together so that
//                           tanh(a_d) * sigmoid(b_d) * Wc_d is formed in registers and summed per lane; two shuffles finish A_n.
//   3. slab softmax statistics (max, sum of exp) and the exp-weighted sum of the slab's rows of x (second read: L2-hot), written as one partial
//      per slab; the LAST slab of a bag to arrive (agent-scope release / acquire around a ticket counter) merges the bag's partials in slab
//      order -- a fixed association, so the result does not depend on which workgroup arrives last.
// Small inputs (gap_split_kernel, total rows <= GF_SPLIT_AUTO_ROWS): a slab kernel wave is a serial chain of (F L + 2 L D) / 64 MFMAs = 137 us at 2.4 GHz
// however few rows there are, and 1024 rows occupy 64 of the chip's 1024 SIMDs.  There a workgroup owns 16 rows and its 4 waves split L instead:
// wave w forms h^T for L/4 of the hidden units (weights straight from L2 as MFMA fragments -- nothing is shared between the waves, so no LDS tile),
// multiplies them into K-partial sums of all 2D gate pre-activations, and the four partials meet in LDS 64 channels at a time (fixed order).
// Weights are read from a PACKED copy (amds_gated_attn_pack, once per set of weights; the entries pack per call when the caller holds none): every LDS
// tile image -- [16 hidden units][16 k] sub-tiles of 1 KB, 16-byte chunks XOR-swizzled so that the fragment reads (ds_read_b128, lane = (row i, chunk
// q)) are conflict-free in all four lane groups of the instruction -- is one contiguous run in HBM.  Reading [rows][16 k] pieces out of the row-major
// matrices instead (64 bytes per row, rows 2-3 KB apart) left the slab kernel at 194 us and the split kernel at 92 us for a 137 / 34 us MFMA chain.
// The slab kernel's tiles arrive by LDS-DMA (global_load_lds, 16 bytes per lane, lane-linear = verbatim), two 32 KB buffers, the copy of tile t + 1
// in flight under the MFMAs of tile t; the split kernel's fragments are 1 KB coalesced loads.  <= 256 registers -> 2 workgroups per CU (one wave's LDS
// / barrier waits hide behind the other's MFMAs).
#include "common.h"

namespace amds {

constexpr int GF_ROWS = 64;          // rows per slab (4 waves x 16)
constexpr int GF_TILE = 8192;        // floats per LDS buffer: [L rows][16 k] (L = 512) or a stage-2 piece [32 rows][256 k]

struct GfArgs {
    const float* x;                  // [total_rows][F]
    const long long* off;            // [bags + 1] row offsets (device), or nullptr: one bag of total_rows rows
    int bags;
    long total_rows;
    const float *fc_b, *a_b, *b_b, *c_w, *c_b;
    const float* packed;             // amds_gated_attn_pack's image of fc_w, a_w, b_w
    float* out;                      // [bags][F]
    float* araw;                     // [total_rows] or nullptr
    int F, D;
    unsigned* counters;              // [bags], zeroed before the launch
    float* stats;                    // [grid][2]  (slab max, slab sum of exp)
    float* part;                     // [grid][F]  exp-weighted slab sums
};

// chunk swizzle of a 16-row x 16-float sub-tile: chunk c of row i sits at chunk position c ^ swz(i), swz = (0, 3, 2, 1)[(i >> 2) & 3].
// ds_read_b128 serves lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} / {32-35, 44-47, 52-59} / {36-43, 48-51, 60-63} together; with lane =
// 16 q + i reading (row i, chunk q) each group then touches 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int gf_swz(int i) { return (4 - ((i >> 2) & 3)) & 3; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct GfUnit { int bag, first, units, nv; long srow; };

// Which bag and which unit (slab / row tile of ROWS rows) block id g is.  Bag b owns ids [off[b] / ROWS + b, ...): strictly increasing in b and never
// fewer than ceil(N_b / ROWS) apart, so no table has to be built: a 256-ary search over the offsets (one round up to 256 bags, two up to 65 536).
// Returns false for the one spare id a bag may own and for every id of an empty bag (block-uniform).
template <int ROWS>
__device__ __forceinline__ bool gf_locate(const GfArgs& p, int g, int* s_cnt, GfUnit& u) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bag = 0, N = (int)p.total_rows, first = 0;
    long row0 = 0;
    if (p.off) {
        int lo = 0, hi = p.bags;
        while (hi - lo > 1) {
            const int step = (hi - lo + 255) / 256;
            const int b = lo + tid * step;
            const bool pred = b < hi && (long)(p.off[b] / ROWS) + b <= (long)g;
            const unsigned long long m = __ballot(pred);
            if (lane == 0) s_cnt[wave] = __popcll(m);
            __syncthreads();
            const int cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            __syncthreads();
            lo += (cnt - 1) * step;
            hi = min(hi, lo + step);
        }
        bag = lo;
        row0 = p.off[bag];
        N = (int)(p.off[bag + 1] - row0);
        first = (int)(row0 / ROWS) + bag;
    }
    const int unit = g - first;
    u.bag = bag; u.first = first; u.units = (N + ROWS - 1) / ROWS;
    if (unit >= u.units) return false;
    u.nv = min(ROWS, N - unit * ROWS);
    u.srow = row0 + (long)unit * ROWS;
    return true;
}

// sum_n wgt[n] * row_n[c4] over n < cnt in ascending n, rows `stride` floats apart; DEPTH 16-byte loads are issued before the first is consumed (a
// plain loop leaves one load in flight per lane: the merge of a 16 384-row bag's 256 partials then took 124 us, 0.5 us per dependent step)
template <int DEPTH>
__device__ __forceinline__ f32x4 gf_weighted_rows(const float* base, long stride, const float* wgt, int cnt) {
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    for (; n + DEPTH <= cnt; n += DEPTH) {
        f32x4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = *reinterpret_cast<const f32x4*>(base + (long)(n + k) * stride);
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const float e = wgt[n + k];
            a4[0] += e * v[k][0]; a4[1] += e * v[k][1]; a4[2] += e * v[k][2]; a4[3] += e * v[k][3];
        }
    }
    for (; n < cnt; ++n) {
        const float e = wgt[n];
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long)n * stride);
        a4[0] += e * v[0]; a4[1] += e * v[1]; a4[2] += e * v[2]; a4[3] += e * v[3];
    }
    return a4;
}

constexpr int GF_MERGE_CHUNK = 4096;     // partial weights held in LDS at a time by the merging workgroup

// Stage 3 of both kernels.  sA[ROWS] (LDS, visible to the block) = attention_raw of the unit's rows; `scratch` = 2 GF_MERGE_CHUNK + 1024 + F floats of LDS
// nobody else uses any more (F <= 4096: a condition of the fused form).  Unit softmax statistics (max, sum of exp) and the exp-weighted sum of the unit's rows of x (second read: L1 / L2-hot)
// -> one partial per unit; a bag of one unit is finished here.  Otherwise the partial is published (agent-scope release), a ticket drawn, and the LAST unit
// of the bag to arrive merges the bag's partials (cdna_hip_programming.md section 6 guideline 16, counter form: plain stores -> every wave vmcnt(0) ->
// barrier -> lane 0 release fence -> relaxed ticket; last arriver: acquire fence -> barrier -> plain loads).  The merge has a FIXED association whichever
// workgroup performs it: wave w sums quarter w of the bag's units in ascending order, the four quarter sums add up in wave order.
template <int ROWS>
__device__ __forceinline__ void gf_finish(const GfArgs& p, float* sA, float* scratch, int* s_last, int g, int bag, int first, int units, int nv, long srow) {
    const int tid = threadIdx.x, F = p.F;
    float m = -INFINITY;
    for (int n = 0; n < nv; ++n) m = fmaxf(m, sA[n]);
    __syncthreads();
    if (tid < ROWS) sA[tid] = tid < nv ? expf(sA[tid] - m) : 0.f;
    __syncthreads();
    float ssum = 0.f;
    for (int n = 0; n < nv; ++n) ssum += sA[n];
    const bool single = units == 1;
    const float inv1 = 1.0f / ssum;
    for (int c4 = tid; c4 * 4 < F; c4 += 256) {
        const f32x4 a4 = gf_weighted_rows<16>(p.x + srow * F + c4 * 4, F, sA, nv);
        if (single) *reinterpret_cast<f32x4*>(p.out + (long)bag * F + c4 * 4) = f32x4{a4[0] * inv1, a4[1] * inv1, a4[2] * inv1, a4[3] * inv1};
        else *reinterpret_cast<f32x4*>(p.part + (long)g * F + c4 * 4) = a4;
    }
    if (single) return;
    if (tid == 0) { p.stats[2 * (long)g] = m; p.stats[2 * (long)g + 1] = ssum; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(p.counters + bag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)(units - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    // ---- merge: global max, then chunks of GF_MERGE_CHUNK units: weights exp(m_s - M) into LDS, wave w sums its quarter of the chunk -------------
    const int lane = tid & 63, wave = tid >> 6;
    const float* stb = p.stats + 2 * (long)first;
    float M = -INFINITY;
    for (int s = tid; s < units; s += 256) M = fmaxf(M, stb[2 * s]);
    M = wave_max(M);
    float* sW = scratch;                         // [GF_MERGE_CHUNK] weights exp(m_s - M) of the chunk's units
    float* sS = sW + GF_MERGE_CHUNK;             // [GF_MERGE_CHUNK] the units' sums of exp
    float* sR = sS + GF_MERGE_CHUNK;             // [4][64] x 16 bytes: the waves' quarter sums of one column pass (first: the 4 per-wave maxima)
    float* sO = sR + 1024;                       // [F] the bag's weighted sum, accumulated across passes and chunks by wave 0
    if (lane == 0) sR[wave] = M;
    for (int c = tid; c < F; c += 256) sO[c] = 0.f;
    __syncthreads();
    M = fmaxf(fmaxf(sR[0], sR[1]), fmaxf(sR[2], sR[3]));
    __syncthreads();
    float den = 0.f;                             // every thread carries the same value
    const int NC4 = F / 4;
    for (int c0 = 0; c0 < units; c0 += GF_MERGE_CHUNK) {
        const int cn = min(GF_MERGE_CHUNK, units - c0);
        for (int s = tid; s < cn; s += 256) {
            sW[s] = expf(stb[2 * (c0 + s)] - M);
            sS[s] = stb[2 * (c0 + s) + 1];
        }
        __syncthreads();
        for (int s = 0; s < cn; ++s) den += sW[s] * sS[s];
        const int qn = (cn + 3) / 4, s0 = min(wave * qn, cn), s1 = min(s0 + qn, cn);        // this wave's quarter of the chunk
        for (int pass = 0; pass * 64 < NC4; ++pass) {
            const int c4 = pass * 64 + lane;
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
            if (c4 < NC4) a4 = gf_weighted_rows<16>(p.part + (long)(first + c0 + s0) * F + c4 * 4, F, sW + s0, s1 - s0);
            *reinterpret_cast<f32x4*>(sR + (wave * 64 + lane) * 4) = a4;
            __syncthreads();
            if (wave == 0 && c4 < NC4) {
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(sR + lane * 4), q1 = *reinterpret_cast<const f32x4*>(sR + (64 + lane) * 4),
                            q2 = *reinterpret_cast<const f32x4*>(sR + (128 + lane) * 4), q3 = *reinterpret_cast<const f32x4*>(sR + (192 + lane) * 4);
                f32x4 o = *reinterpret_cast<const f32x4*>(sO + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += ((q0[e] + q1[e]) + q2[e]) + q3[e];
                *reinterpret_cast<f32x4*>(sO + c4 * 4) = o;
            }
            __syncthreads();
        }
    }
    const float inv = 1.0f / den;
    for (int c = tid; c < F; c += 256) p.out[(long)bag * F + c] = sO[c] * inv;
}

// packed image: P1 = [F / 16 tiles][L rows][16 k] (tile kg = Wfc[:, 16 kg : 16 kg + 16]), then P2 = [D / 16][L / 256 pieces][16 sub-tiles lbp][a | b][16 rows i][16 k]
// (piece = rows 16 db .. 16 db + 15 of Wa and of Wb, k = 256 kp + 16 lbp ..); inside every 16 x 16 sub-tile chunk c of row i sits at position c ^ swz(i)
__device__ __forceinline__ long gf_p1(int L, int kg) { return (long)kg * L * 16; }
__device__ __forceinline__ long gf_p2(int L, int F, int piece) { return (long)L * F + (long)piece * GF_TILE; }

template <int NLB>          // L = 16 * NLB
__global__ void __launch_bounds__(256, 2) gap_fused_kernel(GfArgs p) {
    constexpr int L = NLB * 16;
    constexpr int ND1 = NLB / 4;                 // 1 KB LDS-DMA pieces per wave of a stage-1 tile [L][16]
    constexpr int PP = L / 256;                  // stage-2 pieces (256 k each) per block of 16 gate channels
    __shared__ __attribute__((aligned(16))) float lds[2 * GF_TILE + 16];      // ONE LDS object (a second one makes hipcc drain the DMA queue before every ds_read)
    int* s_cnt = reinterpret_cast<int*>(lds + 2 * GF_TILE);
    int* s_last = s_cnt + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int F = p.F, g = blockIdx.x;

    GfUnit u;
    if (!gf_locate<GF_ROWS>(p, g, s_cnt, u)) return;          // the one spare id a bag may own; also every id of an empty bag
    const int bag = u.bag, first = u.first, nslabs = u.units, nv = u.nv;
    const long srow = u.srow;

    // ---- stage 1: h^T = Wfc x^T ------------------------------------------------------------------------------------------------------
    f32x4 acc[NLB];
#pragma unroll
    for (int lb = 0; lb < NLB; ++lb) acc[lb] = *reinterpret_cast<const f32x4*>(p.fc_b + lb * 16 + 4 * q);     // the bias is the chain's first addend (C layout: row 4 q + r)
    const float* xrow = p.x + (srow + min(wave * 16 + j, nv - 1)) * F + 4 * q;       // rows past the bag's end re-read its last row (weight 0 later)
    const float* rb = lds + j * 16 + 4 * (q ^ gf_swz(j));                           // this lane's fragment address inside a 16 x 16 sub-tile
    // verbatim copy of a packed tile into an LDS buffer: wave w moves the 1 KB runs w, 4 + w, ... (destination = wave-uniform base + 16 lane)
    auto dma = [&](const float* src, float* buf, int n_per_wave) {
        for (int it = 0; it < n_per_wave; ++it) {
            const int chunk = (it * 4 + wave) * 256;
            glds16(src + chunk + lane * 4, buf + chunk);
        }
    };
    const int KG = F / 16;
    dma(p.packed + gf_p1(L, 0), lds, ND1);
    f32x4 xv = *reinterpret_cast<const f32x4*>(xrow), xn = xv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int kg = 0; kg < KG; ++kg) {
        const int cur = (kg & 1) * GF_TILE;
        if (kg + 1 < KG) {                       // the other buffer was last read in iteration kg - 1: every wave has passed that iteration's barrier
            dma(p.packed + gf_p1(L, kg + 1), lds + (GF_TILE - cur), ND1);
            xn = *reinterpret_cast<const f32x4*>(xrow + (kg + 1) * 16);
        }
        // fragment reads run one pair of output blocks ahead of the MFMAs that use them (one wave per SIMD slot sees the ds_read latency otherwise)
        f32x4 w0 = *reinterpret_cast<const f32x4*>(rb + cur), w1 = *reinterpret_cast<const f32x4*>(rb + cur + 256);
#pragma unroll
        for (int lb = 0; lb < NLB; lb += 2) {
            f32x4 n0 = w0, n1 = w1;
            if (lb + 2 < NLB) {
                n0 = *reinterpret_cast<const f32x4*>(rb + cur + (lb + 2) * 256);
                n1 = *reinterpret_cast<const f32x4*>(rb + cur + (lb + 3) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks the two reads back to their first use (lowest pressure) and every pair of blocks waits out the LDS latency
#pragma unroll
            for (int t = 0; t < 4; ++t) {        // two accumulators alternate: the 40-cycle dependent latency of 16x16x4 hides behind the other's issue
                acc[lb] = mfma4(w0[t], xv[t], acc[lb]);
                acc[lb + 1] = mfma4(w1[t], xv[t], acc[lb + 1]);
            }
            w0 = n0; w1 = n1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next tile has landed (this wave's share; the barrier covers the others')
        __syncthreads();
        xv = xn;
    }
#pragma unroll
    for (int lb = 0; lb < NLB; ++lb)             // lane (j, q), register r = h[row j][16 lb + 4 q + r]
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[lb][r] = fmaxf(acc[lb][r], 0.f);

    // ---- stage 2: gate channels 16 at a time, k = L in pieces of 256 -------------------------------------------------------------------
    const int NDB = p.D / 16, NP = NDB * PP;
    dma(p.packed + gf_p2(L, F, 0), lds, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s_lane = 0.f;
#pragma unroll 1
    for (int db = 0; db < NDB; ++db) {
        f32x4 ga = *reinterpret_cast<const f32x4*>(p.a_b + db * 16 + 4 * q), gb = *reinterpret_cast<const f32x4*>(p.b_b + db * 16 + 4 * q);
        const f32x4 cw4 = *reinterpret_cast<const f32x4*>(p.c_w + db * 16 + 4 * q);
#pragma unroll
        for (int kp = 0; kp < PP; ++kp) {
            const int piece = db * PP + kp;
            const int cur = (piece & 1) * GF_TILE;
            if (piece + 1 < NP) dma(p.packed + gf_p2(L, F, piece + 1), lds + (GF_TILE - cur), 8);
            f32x4 wa = *reinterpret_cast<const f32x4*>(rb + cur), wb = *reinterpret_cast<const f32x4*>(rb + cur + 256);
#pragma unroll
            for (int lbp = 0; lbp < 16; ++lbp) {
                f32x4 na = wa, nb = wb;
                if (lbp + 1 < 16) {
                    na = *reinterpret_cast<const f32x4*>(rb + cur + (lbp * 2 + 2) * 256);
                    nb = *reinterpret_cast<const f32x4*>(rb + cur + (lbp * 2 + 3) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {    // k = 16 lb + 4 q + r: the B operand is the stage-1 accumulator register itself
                    ga = mfma4(wa[r], acc[kp * 16 + lbp][r], ga);
                    gb = mfma4(wb[r], acc[kp * 16 + lbp][r], gb);
                }
                wa = na; wb = nb;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {            // lane (j, q), register r = channel d = 16 db + 4 q + r of row j
            const float a = tanhf(ga[r]);
            const float b = 1.0f / (1.0f + expf(-gb[r]));
            s_lane += a * b * cw4[r];
        }
    }
    float araw = s_lane + __shfl_xor(s_lane, 16, 64);
    araw += __shfl_xor(araw, 32, 64);
    araw += p.c_b[0];

    // ---- stage 3: slab softmax statistics + exp-weighted row sum, publish, merge ----------------------------------------------------------
    float* sA = lds;                 // [64] attention_raw of the slab's rows
    const int n_loc = wave * 16 + j;
    if (q == 0) {
        sA[n_loc] = araw;
        if (p.araw && n_loc < nv) p.araw[srow + n_loc] = araw;
    }
    __syncthreads();
    gf_finish<GF_ROWS>(p, sA, lds + 64, s_last, g, bag, first, nslabs, nv, srow);
}

// ---- the small-input form: 16 rows per workgroup, the 4 waves split L (stage 1: output blocks; stage 2: the contraction) ---------------------------
constexpr int GS_ROWS = 16;

template <int NLB>          // L = 16 * NLB
__global__ void __launch_bounds__(256, 2) gap_split_kernel(GfArgs p) {
    constexpr int L = NLB * 16;
    constexpr int NB = NLB / 4;                  // hidden blocks (16 units each) per wave
    constexpr int PP = L / 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * GF_TILE + 16];     // two round buffers [wave][pair s][a | b][lane] x 16 bytes
    int* s_cnt = reinterpret_cast<int*>(lds + 2 * GF_TILE);
    int* s_last = s_cnt + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int F = p.F, g = blockIdx.x;
    GfUnit u;
    if (!gf_locate<GS_ROWS>(p, g, s_cnt, u)) return;
    const int bag = u.bag, first = u.first, ntiles = u.units, nv = u.nv;
    const long srow = u.srow;

    // ---- stage 1: this wave's NB hidden blocks, fragments straight from the packed image (a 16 x 16 sub-tile = 1 KB = one coalesced wave load) ------
    const int lb0 = wave * NB;
    f32x4 h[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) h[b] = *reinterpret_cast<const f32x4*>(p.fc_b + (lb0 + b) * 16 + 4 * q);
    const float* xrow = p.x + (srow + min(j, nv - 1)) * F + 4 * q;
    const int fo = j * 16 + 4 * (q ^ gf_swz(j));                          // this lane's 16 bytes inside a sub-tile
    const float* wrow = p.packed + lb0 * 256 + fo;
    // two fragment sets: the loads of step kg + 1 are issued in front of the MFMAs of step kg (a ring of three measured no faster: 63.8 vs 61.8 us)
    f32x4 wA[NB], wB[NB], xA, xB;
    auto load1 = [&](f32x4 (&wv)[NB], f32x4& xv, int kg) {
        xv = *reinterpret_cast<const f32x4*>(xrow + kg * 16);
#pragma unroll
        for (int b = 0; b < NB; ++b) wv[b] = *reinterpret_cast<const f32x4*>(wrow + gf_p1(L, kg) + b * 256);
    };
    auto mma1 = [&](const f32x4 (&wv)[NB], const f32x4& xv) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) h[b] = mfma4(wv[b][t], xv[t], h[b]);      // consecutive MFMAs on different accumulators
    };
    const int KG = F / 16;                       // even (F % 32 == 0 is a condition of this form)
    load1(wA, xA, 0);
#pragma unroll 1
    for (int kg = 0; kg < KG; kg += 2) {
        load1(wB, xB, kg + 1);
        mma1(wA, xA);
        if (kg + 2 < KG) load1(wA, xA, kg + 2);
        mma1(wB, xB);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[b][r] = fmaxf(h[b][r], 0.f);

    // ---- stage 2: K-partials of the gate pre-activations over this wave's 16 NB hidden units, 4 pairs (a-block, b-block) per round; the four waves'
    //      partials meet in LDS and wave w finishes pair w of the round: bias + p0 + p1 + p2 + p3, gate, x Wc, per-lane sum -----------------------------
    const int NR = p.D / 64;
    const int kp0 = lb0 / 16, lbp0 = lb0 % 16;   // this wave's hidden blocks inside a piece: sub-tiles lbp0 .. lbp0 + NB - 1 of piece kp0
    const float* prow = p.packed + gf_p2(L, F, kp0) + lbp0 * 512 + fo;
    f32x4 fa[NB], fb[NB], na[NB], nb[NB];
    auto load2 = [&](f32x4 (&va)[NB], f32x4 (&vb)[NB], int dblk) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            va[b] = *reinterpret_cast<const f32x4*>(prow + (long)dblk * PP * GF_TILE + b * 512);
            vb[b] = *reinterpret_cast<const f32x4*>(prow + (long)dblk * PP * GF_TILE + b * 512 + 256);
        }
    };
    auto mma2 = [&](const f32x4 (&va)[NB], const f32x4 (&vb)[NB], float* dst) {
        f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ga = mfma4(va[b][r], h[b][r], ga);
                gb = mfma4(vb[b][r], h[b][r], gb);
            }
        *reinterpret_cast<f32x4*>(dst) = ga;
        *reinterpret_cast<f32x4*>(dst + 256) = gb;
    };
    float s_lane = 0.f;
    load2(fa, fb, 0);
#pragma unroll 1
    for (int rd = 0; rd < NR; ++rd) {
        float* buf = lds + (rd & 1) * GF_TILE;
        float* mine = buf + (wave * 4) * 512 + lane * 4;                 // [wave][s][a | b][lane][4]
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
            load2(na, nb, rd * 4 + s + 1);
            mma2(fa, fb, mine + s * 512);
            if (rd * 4 + s + 2 < NR * 4) load2(fa, fb, rd * 4 + s + 2);
            mma2(na, nb, mine + (s + 1) * 512);
        }
        __syncthreads();
        const int dblk = rd * 4 + wave;
        f32x4 va = *reinterpret_cast<const f32x4*>(p.a_b + dblk * 16 + 4 * q), vb = *reinterpret_cast<const f32x4*>(p.b_b + dblk * 16 + 4 * q);
        const f32x4 cw4 = *reinterpret_cast<const f32x4*>(p.c_w + dblk * 16 + 4 * q);
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const float* src = buf + (w2 * 4 + wave) * 512 + lane * 4;
            const f32x4 pa = *reinterpret_cast<const f32x4*>(src), pb = *reinterpret_cast<const f32x4*>(src + 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) { va[r] += pa[r]; vb[r] += pb[r]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_lane += tanhf(va[r]) * (1.0f / (1.0f + expf(-vb[r]))) * cw4[r];
    }
    float part = s_lane + __shfl_xor(s_lane, 16, 64);
    part += __shfl_xor(part, 32, 64);
    __syncthreads();                             // the last round's partials have been read
    float* sP = lds;                             // [4][16] per-wave channel sums, then [16] attention_raw
    if (q == 0) sP[wave * 16 + j] = part;
    __syncthreads();
    float araw = 0.f;
    if (tid < GS_ROWS) araw = ((sP[tid] + sP[16 + tid]) + sP[32 + tid]) + sP[48 + tid] + p.c_b[0];
    __syncthreads();
    float* sA = lds;
    if (tid < GS_ROWS) {
        sA[tid] = araw;
        if (p.araw && tid < nv) p.araw[srow + tid] = araw;
    }
    __syncthreads();
    gf_finish<GS_ROWS>(p, sA, lds + 64, s_last, g, bag, first, ntiles, nv, srow);
}

// ---- packing: row-major Wfc [L][F], Wa / Wb [D][L]  ->  the tile images above.  One thread per 16-byte chunk -----------------------------------
__global__ void __launch_bounds__(256) gap_pack_kernel(const float* __restrict__ fc_w, const float* __restrict__ a_w, const float* __restrict__ b_w,
                                                       float* __restrict__ packed, int F, int L, int D) {
    const long n1 = (long)L * F / 4, n2 = 2L * D * L / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n1) {                              // P1 chunk: (kg, row, position)
        const int pos = (int)(idx & 3), row = (int)((idx >> 2) % L), kg = (int)((idx >> 2) / L);
        const int c = pos ^ gf_swz(row & 15);
        *reinterpret_cast<f32x4*>(packed + idx * 4) = *reinterpret_cast<const f32x4*>(fc_w + (long)row * F + kg * 16 + 4 * c);
    } else if (idx < n1 + n2) {                  // P2 chunk: (piece, lbp, a | b, i, position)
        const long t = idx - n1;
        const int pos = (int)(t & 3), i = (int)((t >> 2) & 15), ab = (int)((t >> 6) & 1), lbp = (int)((t >> 7) & 15);
        const int PP = L / 256;
        const long piece = t >> 11;
        const int db = (int)(piece / PP), kp = (int)(piece % PP);
        const int c = pos ^ gf_swz(i);
        *reinterpret_cast<f32x4*>(packed + (long)L * F + t * 4) =
            *reinterpret_cast<const f32x4*>((ab ? b_w : a_w) + (long)(db * 16 + i) * L + kp * 256 + lbp * 16 + 4 * c);
    }
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

bool gap_fused_shape_ok(int F, int L, int D) { return (L == 256 || L == 512) && F > 0 && F % 16 == 0 && F <= 4096 && D > 0 && D % 16 == 0; }

bool gap_fused_supported(const float* x, const amds_gap_weights* w, int F, int L, int D) {
    return gap_fused_shape_ok(F, L, D) && al16(x) && al16(w->fc_w) && al16(w->fc_b) && al16(w->a_w) && al16(w->a_b) && al16(w->b_w) && al16(w->b_b) &&
           al16(w->c_w) && al16(w->packed);
}

size_t gap_packed_floats(int F, int L, int D) { return (size_t)L * F + (size_t)2 * D * L; }

int gap_pack(const amds_gap_weights* w, float* packed, int F, int L, int D, hipStream_t st) {
    const long chunks = (long)gap_packed_floats(F, L, D) / 4;
    hipLaunchKernelGGL(gap_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, w->fc_w, w->a_w, w->b_w, packed, F, L, D);
    AMDS_LAUNCH_CHECK("gap_pack_kernel");
    return AMDS_OK;
}

// total rows up to which the split form is chosen: 256 row tiles (4096 rows) fill the chip once (one workgroup per CU, every SIMD busy; 83 us), 512
// twice (two workgroups per CU); the last arriver's merge of a bag's partials (3 KB each) stays a few microseconds
constexpr long GF_SPLIT_MAX_ROWS = 12288;      // what AMDS_GAP_SPLIT accepts
constexpr long GF_SPLIT_AUTO_ROWS = 8192;       // what AMDS_GAP_AUTO gives to the split form (8192 rows: 189 us split, 205 us slab; 12 288: 254 / 204)

static bool split_ok(long total_rows, int F, int D) { return total_rows <= GF_SPLIT_MAX_ROWS && F % 32 == 0 && D % 64 == 0; }
static size_t units_of(long total_rows, int bags, int rows) { return (size_t)(total_rows / rows) + (size_t)bags; }

// counters | unit statistics | unit partials | room for the packed weights (used when the caller passes none)
size_t gap_fused_workspace_bytes(long total_rows, int bags, int F, int L, int D) {
    const size_t G = units_of(total_rows, bags, split_ok(total_rows, F, D) ? GS_ROWS : GF_ROWS);      // enough for either form where both are allowed
    return al256((size_t)bags * 4) + al256(G * 2 * 4) + al256(G * (size_t)F * 4) + al256(gap_packed_floats(F, L, D) * 4);
}

// mode: AMDS_GAP_AUTO (split form up to GF_SPLIT_MAX_ROWS total rows, slab form beyond), AMDS_GAP_SLAB, AMDS_GAP_SPLIT
int gap_fused_launch(const float* x, const long long* off_dev, int bags, long total_rows, const amds_gap_weights* w, float* out, float* attn_raw, int F,
                     int L, int D, int mode, void* ws, size_t ws_bytes, hipStream_t st) {
    AMDS_REQUIRE(mode == AMDS_GAP_AUTO || mode == AMDS_GAP_SLAB || mode == AMDS_GAP_SPLIT, "gated_attn_pool: unknown mode %d", mode);
    AMDS_REQUIRE(mode != AMDS_GAP_SPLIT || split_ok(total_rows, F, D),
                 "gated_attn_pool: the split form takes at most %ld rows in total, F a multiple of 32 and D a multiple of 64 (rows=%ld F=%d D=%d)",
                 GF_SPLIT_MAX_ROWS, total_rows, F, D);
    const bool split = mode == AMDS_GAP_SPLIT || (mode == AMDS_GAP_AUTO && total_rows <= GF_SPLIT_AUTO_ROWS && split_ok(total_rows, F, D));
    const size_t need = gap_fused_workspace_bytes(total_rows, bags, F, L, D);
    if (ws_bytes < need) {
        set_error("gated_attn_pool: workspace %zu < required %zu bytes", ws_bytes, need);
        return AMDS_ERR_WORKSPACE;
    }
    const size_t Gmax = units_of(total_rows, bags, split_ok(total_rows, F, D) ? GS_ROWS : GF_ROWS);
    const size_t G = units_of(total_rows, bags, split ? GS_ROWS : GF_ROWS);
    AMDS_REQUIRE(G < ((size_t)1 << 31), "gated_attn_pool: %zu units exceed the grid limit", G);
    char* q = reinterpret_cast<char*>(ws);
    GfArgs a;
    a.x = x; a.off = off_dev; a.bags = bags; a.total_rows = total_rows;
    a.fc_b = w->fc_b; a.a_b = w->a_b; a.b_b = w->b_b; a.c_w = w->c_w; a.c_b = w->c_b;
    a.out = out; a.araw = attn_raw; a.F = F; a.D = D;
    a.counters = reinterpret_cast<unsigned*>(q);  q += al256((size_t)bags * 4);
    a.stats = reinterpret_cast<float*>(q);        q += al256(Gmax * 2 * 4);
    a.part = reinterpret_cast<float*>(q);         q += al256(Gmax * (size_t)F * 4);
    a.packed = w->packed;
    if (!a.packed) {                             // the caller holds no packed copy: make one (one more launch; ops.py / HipGatedAttentionEncoder keep theirs)
        float* pk = reinterpret_cast<float*>(q);
        int rc = gap_pack(w, pk, F, L, D, st);
        if (rc != AMDS_OK) return rc;
        a.packed = pk;
    }
    AMDS_HIP(hipMemsetAsync(a.counters, 0, (size_t)bags * 4, st));
    ProfScope prof(PROF_GEMM_F32, 2.0 * (double)total_rows * ((double)F * L + 2.0 * (double)L * D), st);
    if (split) {
        if (L == 512) hipLaunchKernelGGL((gap_split_kernel<32>), dim3((unsigned)G), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gap_split_kernel<16>), dim3((unsigned)G), dim3(256), 0, st, a);
        AMDS_LAUNCH_CHECK("gap_split_kernel");
    } else {
        if (L == 512) hipLaunchKernelGGL((gap_fused_kernel<32>), dim3((unsigned)G), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gap_fused_kernel<16>), dim3((unsigned)G), dim3(256), 0, st, a);
        AMDS_LAUNCH_CHECK("gap_fused_kernel");
    }
    return AMDS_OK;
}

}  // namespace amds

using namespace amds;

extern "C" int amds_gated_attn_pool_batched_supported(int F, int L, int D) { return gap_fused_shape_ok(F, L, D); }

extern "C" size_t amds_gated_attn_packed_floats(int F, int L, int D) { return gap_fused_shape_ok(F, L, D) ? gap_packed_floats(F, L, D) : 0; }

extern "C" int amds_gated_attn_pack(const amds_gap_weights* w, float* packed, int F, int L, int D, void* stream) {
    AMDS_REQUIRE(w && packed && w->fc_w && w->a_w && w->b_w, "amds_gated_attn_pack: null pointer");
    AMDS_REQUIRE(gap_fused_shape_ok(F, L, D), "amds_gated_attn_pack: needs L in {256, 512}, F <= 4096 and D multiples of 16 (F=%d L=%d D=%d)", F, L, D);
    AMDS_REQUIRE(al16(w->fc_w) && al16(w->a_w) && al16(w->b_w) && al16(packed), "amds_gated_attn_pack: 16-byte aligned arrays needed");
    return gap_pack(w, packed, F, L, D, (hipStream_t)stream);
}

extern "C" size_t amds_gated_attn_pool_batched_workspace_bytes(long total_rows, int bags, int F, int L, int D) {
    if (total_rows <= 0 || bags <= 0 || !gap_fused_shape_ok(F, L, D)) return 0;
    return gap_fused_workspace_bytes(total_rows, bags, F, L, D);
}

extern "C" int amds_gated_attn_pool_batched(const float* x, const long long* row_offsets, int bags, long total_rows, const amds_gap_weights* w, float* out,
                                            float* attn_raw, int F, int L, int D, int mode, void* ws, size_t ws_bytes, void* stream) {
    AMDS_REQUIRE(x && w && out && ws, "amds_gated_attn_pool_batched: null pointer");
    AMDS_REQUIRE(row_offsets || bags == 1, "amds_gated_attn_pool_batched: row_offsets may be NULL for a single bag only");
    AMDS_REQUIRE(bags > 0 && total_rows > 0 && F > 0 && L > 0 && D > 0, "amds_gated_attn_pool_batched: empty input or bad dims (bags=%d rows=%ld F=%d L=%d D=%d)",
                 bags, total_rows, F, L, D);
    AMDS_REQUIRE(w->fc_w && w->fc_b && w->a_w && w->a_b && w->b_w && w->b_b && w->c_w && w->c_b, "amds_gated_attn_pool_batched: incomplete weights");
    AMDS_REQUIRE(gap_fused_supported(x, w, F, L, D),
                 "amds_gated_attn_pool_batched: needs L in {256, 512}, F <= 4096 and D multiples of 16 and 16-byte aligned x / weights (F=%d L=%d D=%d); pool other "
                 "shapes one bag at a time with amds_gated_attn_pool", F, L, D);
    return gap_fused_launch(x, row_offsets, bags, total_rows, w, out, attn_raw, F, L, D, mode, ws, ws_bytes, (hipStream_t)stream);
}
