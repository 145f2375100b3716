// qkv_attn257.hip -- the qkv Linear and the attention of a ViT block as ONE kernel for T = 257 tokens, head_dim 64: q / k / v never leave the chip.
//
// Why (DESIGN section 9, item 00 a): per block and chunk of 1020 tiles the qkv GEMM wrote 1.6 GB of q | k | v to HBM and the attention kernel
// read them back -- 3.2 of the block's ~14 GB, in a regime where bytes cost time wherever they are put (profiles/r04_gemm_overlap_probe.txt).
//
// One persistent 256-thread workgroup per CU (four waves, one per SIMD, 512 registers each) walks items (tile b, head h):
//   G phase  [q | k | v](256 x 192) = X_b[256 x D] . W_h[192 x D]^T on the production GEMM's K loop (gemm_4w16.h: two LDS stages of 128-byte rows
//            filled by buffer-form LDS-DMA, inline-asm v_mfma_f32_16x16x32 with the accumulators pinned to AGPRs, one barrier per K tile of 64);
//            wave (wm, wn) owns rows 128 wm .. + 127 and, of each of q, k, v, the 16-column blocks 2 wn, 2 wn + 1 (8 x 6 accumulator blocks).
//            For the q and k blocks W is the MFMA "A" operand (a lane ends up with one token and 4 consecutive dims: a row-major image);
//            for the v blocks the operands are SWAPPED (a lane ends up with one dim and 4 consecutive tokens): V^T comes out of the matrix
//            pipe already transposed.
//   hand-off the folded-LayerNorm epilogue of the qkv GEMM (acc * rstd[m] + colsum[n] * (-mean rstd)[m] + bias[n]; k and v rounded to the operand type as the
//            unfused path stores them, q multiplied by log2(e) / sqrt(64) first) writes the K image, the V^T image and a Q image over the (now idle) stages,
//            in the layouts attention_vit257.hip reads.  Stage 0 sits on the Q image: once every wave holds its query fragments the NEXT item's first
//            K tile is requested into it, under the S phase.
//   S phase  each wave takes its two blocks of 32 queries in ONE instruction stream; softmax in two passes over the 8 key tiles (pass 1: Q K^T for the row
//            maxima; pass 2: Q K^T again with the shift subtracted inside the product, p = exp2(score), P V, row sums against a ones operand); the odd
//            key as a rank-1 term at the end, the odd query on the MFMA pipe as 32-key partials merged after the item's barrier.
// Token 256 of a tile (the odd one: 257 = 16 x 16 + 1) is not part of the 256-row G phase: its q | k | v row is computed beforehand by an ordinary GEMM
// on the 1020 gathered rows (vit.hip) and read from HBM here (384 bytes per item).
// STATUS (round 5): parity-tested (tests/test_gpu_qkv_attn.py), 1.7 % faster than the two launches alone, 0.8 % slower inside the encoder -- opt-in
// (AMDS_VIT_QKVATTN=1); the measurements and the reasons are in profiles/r05_qkv_attn_fused_ab.txt and DESIGN.md section 4.14.
// Output: attention rows [B * 257][D] in the operand type, as attention_vit257.hip writes them.
#include "common.h"
#include <type_traits>

namespace amds {

constexpr int QA_XB = 256 * 128;                       // X part of a stage: 256 token rows x 128 B (K tile of 64)
constexpr int QA_STAGE = (256 + 192) * 128;            // + 192 weight rows: 57 344 B
constexpr int QA_VS = 576;                             // V^T rows: 8 key tiles x 64 B + 64 B skew room (attention_vit257.hip)
// The images alias the stages: stage 1 (odd K tiles, the LAST tile of an item) = the K | V^T region, stage 0 (even K tiles, the FIRST tile of an item) starts
// at the Q image.  The Q image is dead once every wave holds its query fragments (one barrier into the S phase), so the next item's first K tile is
// requested into stage 0 while the S phase still reads K and V^T: the item-to-item pipeline never drains.
constexpr int QA_KIMG = 0, QA_VIMG = 32768, QA_VBYTES = 64 * QA_VS + 8 * 16, QA_QIMG = QA_VIMG + QA_VBYTES;
constexpr int QA_S1 = 0, QA_S0 = QA_QIMG;
static_assert(QA_S1 + QA_STAGE <= QA_S0, "the stages must not overlap");
constexpr int QA_PARTF = 68;                           // one partial of the odd query's row: o[64] | max | sum | pad
constexpr int QA_ZERO = QA_S0 + QA_STAGE;              // 16 zero bytes
constexpr int QA_PW = QA_ZERO + 16;                    // [8 query blocks][32] 16-bit softmax weights of the odd query
constexpr int QA_PART = QA_PW + 8 * 64;                // [9][QA_PARTF] f32
constexpr int QA_TAIL = QA_PART + 9 * QA_PARTF * 4;    // odd token: key | value | query in fp32 (3 x 64), the query again in 16 bit (128 B)
constexpr int QA_RS = QA_TAIL + 3 * 64 * 4 + 128;      // (rstd, -mean rstd) of the item's 256 rows
constexpr int QA_OUT = QA_RS + 256 * 8;                // per wave: 32 output rows x 128 B
constexpr int QA_LDS = QA_OUT + 4 * 4096;
static_assert(QA_RS % 16 == 0 && QA_OUT % 16 == 0 && QA_TAIL % 16 == 0 && QA_LDS <= 160 * 1024, "LDS map");

// phase timeline for tools/ubench/qa257_trace.hip (compiled with -DQA_TRACE only): s_memtime of the four waves of workgroup 0 at the phase boundaries
#ifdef QA_TRACE
__device__ unsigned long long qa_trace[32 * 12 * 4];
#define QA_MARK(k)                                                                                                          \
    do {                                                                                                                    \
        if (blockIdx.x == 0 && lane == 0 && round < 32) qa_trace[(round * 12 + (k)) * 4 + wave] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define QA_MARK(k) do { } while (0)
#endif

template <typename T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
qkv_attn257_kernel(const T* __restrict__ X, const T* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ colsum,
                   const float* __restrict__ rowstat, const T* __restrict__ qkv_tail, T* __restrict__ out, int B, int H, int D, int n_slots) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int Tn = 257, VS = QA_VS, ROWB = 128, FI = 8, FJ = 6, NM = 24;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kb = lane >> 4, l31 = lane & 31, hi = lane >> 5;
    const int nk = D / 64;
    const float sc = 0.125f * 1.44269504088896340736f;                // 1/sqrt(64) * log2(e)

    char* sZero = smem + QA_ZERO;
    char* sPw = smem + QA_PW;
    float* sPart = reinterpret_cast<float*>(smem + QA_PART);
    float* sT = reinterpret_cast<float*>(smem + QA_TAIL);
    if (tid < 4) reinterpret_cast<float*>(sZero)[tid] = 0.f;

    // ---- LDS-DMA addressing (gemm_4w16.h): 14 pieces of 16 bytes per thread and K tile, 8 of X and 6 of W; 16-byte chunk ^= (row >> 1) & 7 on the SOURCE
    int voffx[8], voffw[6];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = it * 256 + tid, row = c >> 3, cp = c & 7, scn = cp ^ ((row >> 1) & 7);
        voffx[it] = (row * D + scn * 8) * 2;
    }
#pragma unroll
    for (int it = 0; it < 6; ++it) {        // stage row r = 64 part + d (part 0 / 1 / 2 = q / k / v, d = dim): weight row part * D + h * 64 + d (h in the scalar offset)
        const int c = it * 256 + tid, row = c >> 3, cp = c & 7, scn = cp ^ ((row >> 1) & 7);
        voffw[it] = (((row >> 6) * D + (row & 63)) * D + scn * 8) * 2;
    }
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W), 0, 3 * D * D * 2, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_x = rsrc_w;
    int wsoff = 0;
    auto issue_pieces = [&](int kt, int lo, int hi_) {
        char* st = smem + ((kt & 1) ? QA_S1 : QA_S0);
        const int koff = kt * 128;
#pragma unroll
        for (int it = 0; it < 14; ++it)
            if (it >= lo && it < hi_) {
                if (it < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lptr_t)(st + (it * 256 + wave * 64) * 16), 16, voffx[it], koff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(st + QA_XB + ((it - 8) * 256 + wave * 64) * 16), 16, voffw[it - 8], koff + wsoff, 0, 0);
            }
    };
    const int swz = (l15 >> 1) & 7;
    const int a_off = (wm * 128 + l15) * ROWB;
    int w_off[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) w_off[j] = QA_XB + ((j >> 1) * 64 + (2 * wn + (j & 1)) * 16 + l15) * ROWB;

    vec8 af[2][FI], wf[2][FJ];
    auto load_frags = [&](int kt, int ks, int s, int lo, int hi_) {
        const char* sb = smem + ((kt & 1) ? QA_S1 : QA_S0);
        const int co = ((ks * 4 + kb) ^ swz) << 4;
#pragma unroll
        for (int q = 0; q < 14; ++q)
            if (q >= lo && q < hi_) {
                if (q < 8) af[s][q] = *reinterpret_cast<const vec8*>(sb + a_off + q * 16 * ROWB + co);
                else wf[s][q - 8] = *reinterpret_cast<const vec8*>(sb + w_off[q - 8] + co);
            }
    };
    f32x4 acc[FI][FJ];
    // A unit = 24 MFMAs (set s, row blocks 4 ih .. 4 ih + 3, all 6 column blocks); behind them fragment reads and LDS-DMA pieces as in gemm_4w16.h
    auto unit = [&](int s, int ih, auto nr_c, int rkt, int rks, int rs_, int rlo, auto nc_c, int ckt, int clo) {
        constexpr int NR = decltype(nr_c)::value, NC = decltype(nc_c)::value;
        constexpr int R_SPAN = NR >= 14 ? 14 : NM, C_LO = NR >= 14 ? 14 : 0, C_SPAN = NM - C_LO;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int i = ih * 4 + m / FJ, j = m % FJ;
            if (j < 4) Act<T>::mfma16_agpr(wf[s][j], af[s][i], acc[i][j]);          // lane: token l15, dims 4 kb .. 4 kb + 3
            else Act<T>::mfma16_agpr(af[s][i], wf[s][j], acc[i][j]);                // lane: dim l15, tokens 4 kb .. 4 kb + 3
            if (NR > 0 && m < R_SPAN) {
                const int r0 = m * NR / R_SPAN, r1 = (m + 1) * NR / R_SPAN;
                if (r1 > r0) load_frags(rkt, rks, rs_, rlo + r0, rlo + r1);
            }
            if (NC > 0 && m >= C_LO && m < C_LO + C_SPAN) {
                const int c0 = (m - C_LO) * NC / C_SPAN, c1 = (m - C_LO + 1) * NC / C_SPAN;
                if (c1 > c0) issue_pieces(ckt, clo + c0, clo + c1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 7> I7;
    typedef std::integral_constant<int, 14> I14;
#define QA_BARRIER()                          \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)
    // pieces of tile kt + 1: 0 .. P3 - 1 in the last unit of tile kt - 1, then P0 and P1 in the first two units of tile kt; waited for (vmcnt(0)) at 3/4 of tile kt
#ifndef QA_P3
#define QA_P3 4
#endif
#ifndef QA_P0
#define QA_P0 5
#endif
    constexpr int P3 = QA_P3, P0 = QA_P0, P1 = 14 - P3 - P0;
    static_assert(P3 >= 0 && P3 <= 10 && P0 >= 0 && P1 >= 0, "LDS-DMA piece split");
    typedef std::integral_constant<int, P3> IP3;
    typedef std::integral_constant<int, P0> IP0;
    typedef std::integral_constant<int, P1> IP1;
    auto k_tile = [&](int kt, auto next_c, auto next2_c) {
        constexpr bool NEXT = decltype(next_c)::value, NEXT2 = decltype(next2_c)::value;
        if constexpr (NEXT) unit(0, 0, I7{}, kt, 1, 1, 0, IP0{}, kt + 1, P3); else unit(0, 0, I7{}, kt, 1, 1, 0, I0{}, 0, 0);
        if constexpr (NEXT) unit(0, 1, I7{}, kt, 1, 1, 7, IP1{}, kt + 1, P3 + P0); else unit(0, 1, I7{}, kt, 1, 1, 7, I0{}, 0, 0);
        unit(1, 0, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QA_BARRIER();
        if constexpr (NEXT2) unit(1, 1, I14{}, kt + 1, 0, 0, 0, IP3{}, kt + 2, 0);
        else if constexpr (NEXT) unit(1, 1, I14{}, kt + 1, 0, 0, 0, I0{}, 0, 0);
        else unit(1, 1, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
    };

    // ---- item order: an XCD's workgroups (blockIdx & 7 == xcd) take CONSECUTIVE slots, and 32 consecutive slots are 4 tiles x 8 heads: per round an
    // XCD fetches 4 x 526 KB of X and 8 x 393 KB of W once and serves the other reads from its L2.  (With H % 8 == 0 a workgroup keeps its head for the
    // whole launch and walks tiles: slot -> tile is monotonic, so the first slot past the batch ends the walk.)
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3, per = gridDim.x >> 3;
    const bool oct = (H & 7) == 0;
    auto slot_item = [&](int round, int& b, int& h) {
        const int u = (round * 8 + xcd) * per + wq;
        if (u >= n_slots) return false;
        if (oct) {
            const int per_quad = 4 * H, quad = u / per_quad, rem = u - quad * per_quad;
            b = quad * 4 + ((rem & 31) >> 3);
            h = (rem >> 5) * 8 + (rem & 7);
        } else {
            b = u / H;
            h = u - b * H;
        }
        return b < B;
    };
    auto point_at = [&](int b, int h) {          // the LDS-DMA descriptors of item (b, h)
        rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X + (long)b * Tn * D), 0, 256 * D * 2, 0x00020000);
        wsoff = h * 64 * D * 2;
    };
    // hand-off addresses: lane-only, four of them; everything else is an immediate offset (block row 16 i -> 2048 B of a row-major image, 32 B of a V^T row)
    const int ho_sw = (l15 >> 1) & 7;
    int ho_qk[2], ho_v[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int c0 = (2 * wn + jj) * 16 + 4 * kb;                                      // first of the lane's 4 dims in a q / k block
        ho_qk[jj] = (wm * 128 + l15) * 128 + (((c0 >> 3) ^ ho_sw) << 4) + ((c0 >> 2) & 1) * 8;
        const int d = (2 * wn + jj) * 16 + l15, t0 = wm * 128 + 4 * kb;                  // a v block: dim d, tokens t0 + 16 i .. + 3
        ho_v[jj] = QA_VIMG + d * VS + (d >> 3) * 16 + ((t0 & ~12) | ((t0 & 4) << 1) | ((t0 & 8) >> 1)) * 2;      // key order inside 16-groups: bits 2 <-> 3
    }
    const int ho_rs1 = QA_RS + (wm * 128 + l15) * 8, ho_rs4 = QA_RS + (wm * 128 + 4 * kb) * 8;

    int b = 0, h = 0;
    bool have = slot_item(0, b, h);
    if (have) {
        point_at(b, h);
        issue_pieces(0, 0, 14);
    }
#pragma unroll 1
    for (int round = 0; have; ++round) {
        const long row0 = (long)b * Tn;
        int nb = 0, nh = 0;
        const bool have_next = slot_item(round + 1, nb, nh);

        QA_MARK(0);
        // ================= G phase (K tile 0 was requested during the previous item's S phase) =================
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the initial values must BE in their AGPRs well before the first inline-asm MFMA reads them: gemm_4w16.h)
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            if (i == FI - 1) asm volatile("s_nop 7" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
            else asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K tile 0
        QA_BARRIER();
        QA_MARK(1);
        // small per-item operands (row statistics, the odd token's q | k | v): requested now, written to LDS after the K loop
        f32x2 rsv = f32x2{1.f, 0.f};
        if (rowstat) rsv = *reinterpret_cast<const f32x2*>(rowstat + 2 * (row0 + tid));
        const T* tp = qkv_tail + (row0 + 256) * 3 * D + h * 64 + lane;
        const T tq = tp[0], tk = tp[D], tv = tp[2 * D];
        load_frags(0, 0, 0, 0, 14);
        issue_pieces(1, 0, P3);
        __builtin_amdgcn_sched_barrier(0);
        int kt = 0;
        for (; kt < nk - 2; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
        k_tile(kt++, std::true_type{}, std::false_type{});
        // per-column (bias, colsum) of this wave's blocks: requested before the last K tile, which issues no LDS-DMA and waits for none
        f32x4 cb[4], cs[4];
        float cbv[2], csv[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = (j >> 1) * D + h * 64 + (2 * wn + (j & 1)) * 16 + 4 * kb;
            cb[j] = *reinterpret_cast<const f32x4*>(bias + n);
            cs[j] = colsum ? *reinterpret_cast<const f32x4*>(colsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 2 * D + h * 64 + (2 * wn + j) * 16 + l15;
            cbv[j] = bias[n];
            csv[j] = colsum ? colsum[n] : 0.f;
        }
        k_tile(kt, std::false_type{}, std::false_type{});
        *reinterpret_cast<f32x2*>(smem + QA_RS + tid * 8) = rsv;
        if (tid < 64) {
            sT[tid] = Act<T>::to_f32(tk); sT[64 + tid] = Act<T>::to_f32(tv); sT[128 + tid] = Act<T>::to_f32(tq);
            reinterpret_cast<T*>(sT + 192)[tid] = tq;
        }
        QA_BARRIER();                  // every wave is done with the LDS stages; the row statistics are in place
        QA_MARK(2);
        // MFMA result -> accumulator read hazard (gemm_4w16.h): keep every accumulator in its AGPR until the nops have run
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            if (i == 0) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
            else asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
        }
        __builtin_amdgcn_sched_barrier(0);

        // ================= hand-off: accumulators -> Q / K / V^T images =================
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const f32x2 rs1 = *reinterpret_cast<const f32x2*>(smem + ho_rs1 + i * 128);                 // q / k blocks: this lane's token 16 i + l15
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(acc[i][j][0]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(acc[i][j][1]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(acc[i][j][2]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(acc[i][j][3]));
                v = v * rs1[0] + (cs[j] * rs1[1] + cb[j]);
                if (j < 2) v *= sc;                 // q leaves pre-scaled by log2(e) / sqrt(64): the score products ARE the exp2 arguments (up to the row maximum)
                *reinterpret_cast<vec4*>(smem + (j < 2 ? QA_QIMG : QA_KIMG) + ho_qk[j & 1] + i * 2048) = Act<T>::from_f32x4(v);
            }
            const f32x4 ra = *reinterpret_cast<const f32x4*>(smem + ho_rs4 + i * 128), rb = *reinterpret_cast<const f32x4*>(smem + ho_rs4 + i * 128 + 16);
#pragma unroll
            for (int j = 4; j < 6; ++j) {                                                            // v blocks: this lane's tokens 16 i + 4 kb .. + 3
                f32x4 v;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(acc[i][j][0]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(acc[i][j][1]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(acc[i][j][2]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(acc[i][j][3]));
                const float c_b = cbv[j - 4], c_s = csv[j - 4];
                v = f32x4{v[0] * ra[0] + (c_s * ra[1] + c_b), v[1] * ra[2] + (c_s * ra[3] + c_b), v[2] * rb[0] + (c_s * rb[1] + c_b),
                          v[3] * rb[2] + (c_s * rb[3] + c_b)};
                *reinterpret_cast<vec4*>(smem + ho_v[j - 4] + i * 32) = Act<T>::from_f32x4(v);
            }
        }
        QA_MARK(3);
        __syncthreads();
        QA_MARK(4);

        // ================= S phase (each wave: query blocks 2 wave, 2 wave + 1) =================
        const char* sK = smem + QA_KIMG;
        const char* sVt = smem + QA_VIMG;
        const char* sQ = smem + QA_QIMG;
        const float* sKt = sT;
        const float* sVl = sT + 64;
        const float* sQt = sT + 128;
        int ols = lane;
        asm volatile("" : "+v"(ols));
        const int l31 = ols & 31, hi = ols >> 5;
        const int swz32 = (l31 >> 1) & 7;
        // Both query blocks of the wave go through ONE instruction stream: a wave alone on its SIMD has nobody to hide its dependent MFMA / VALU latencies
        // behind; two independent blocks in every slice do that for each other, and every K / V^T operand fragment is read from LDS once for both.
        // Softmax in TWO passes over the 8 key tiles instead of an online one: pass 1 = Q K^T on the (otherwise idle) matrix pipe for the row maxima only;
        // pass 2 = Q K^T again, p = exp2(s * scale - max), P V, and the row sums as one more product of P against a ones operand.  No running maximum, hence
        // no rescaling of the output accumulators, which never leave the AGPRs; one key tile (16 scores per lane) in flight instead of two, which is what
        // lets two blocks fit the 256 architectural VGPRs; the score products land in VGPRs (inline asm) so the softmax reads them without v_accvgpr_read.
        constexpr int NB = 2;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        vec8 qf[NB][4];
#pragma unroll
        for (int r = 0; r < NB; ++r)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[r][ks] = *reinterpret_cast<const vec8*>(sQ + ((2 * wave + r) * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz32) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        QA_BARRIER();                  // every wave holds its queries: the Q image (= the head of stage 0) is free
        if (have_next) point_at(nb, nh);
        auto k_frag = [&](int t, int ks) { return *reinterpret_cast<const vec8*>(sK + (t * 32 + l31) * 128 + (((ks * 2 + hi) ^ swz32) << 4)); };
        auto v_frag = [&](int t, int ks, int dt) {
            const int pos = t * 32 + ks * 16 + hi * 8, d = dt * 32 + l31;
            return *reinterpret_cast<const vec8*>(sVt + d * VS + (d >> 3) * 16 + pos * 2);
        };
        float mrow[NB], pt[NB];
        {   // ---- pass 1: row maxima.  Iteration t issues tile t + 1's eight products, each followed by a quarter of tile t's maxima; the next item's first
            // K tile is requested on the way, two LDS-DMA pieces per iteration
            f32x16 sA[NB], sB[NB];
            float mx[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) mx[r] = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 kf = k_frag(0, ks);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    if (ks == 0) Act<T>::mfma32_vgpr_zero(sA[r], kf, qf[r][ks]); else Act<T>::mfma32_vgpr_acc(sA[r], kf, qf[r][ks]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            auto iter = [&](int t, f32x16 (&cur)[NB], f32x16 (&nxt)[NB]) {
                vec8 kf[4];
                if (t + 1 < 8) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) kf[ks] = k_frag(t + 1, ks);
                }
                if (have_next && t < 7) issue_pieces(0, 2 * t, 2 * t + 2);
                if (t == 0) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // tile 0's chain has nothing in front of its first reader
                __builtin_amdgcn_sched_barrier(0);
                // (a quarter of the maxima behind each pair of products, one pair late: the first VALU read of `cur` comes four MFMA issues after the asm that wrote it)
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    if (t + 1 < 8 && ks < 4) {
#pragma unroll
                        for (int r = 0; r < NB; ++r) {
                            if (ks == 0) Act<T>::mfma32_vgpr_zero(nxt[r], kf[ks], qf[r][ks]); else Act<T>::mfma32_vgpr_acc(nxt[r], kf[ks], qf[r][ks]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (the maxima stay BEHIND this slice's products)
                    if (ks >= 1) {
#pragma unroll
                        for (int r = 0; r < NB; ++r)
#pragma unroll
                            for (int e = 4 * (ks - 1); e < 4 * ks; ++e) mx[r] = fmaxf(mx[r], cur[r][e]);
                        // opaque to the optimiser: max is associative, and left alone the whole 256-leaf reduction is re-associated and sunk behind the last tile --
                        // every tile's scores then stay live (spills, and copies of the asm MFMAs' outputs taken right behind them: stale values)
                        asm volatile("" : "+v"(mx[0]), "+v"(mx[1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (t == 6) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // ... nor has tile 7's
            };
            iter(0, sA, sB); iter(1, sB, sA); iter(2, sA, sB); iter(3, sB, sA); iter(4, sA, sB); iter(5, sB, sA); iter(6, sA, sB); iter(7, sB, sA);
            // the odd key's score (dot product split over the lane pair) joins the maximum
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                float dot = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f32x4 k0 = *reinterpret_cast<const f32x4*>(sKt + (ks * 2 + hi) * 8), k1 = *reinterpret_cast<const f32x4*>(sKt + (ks * 2 + hi) * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dot = fmaf(Act<T>::to_f32(qf[r][ks][e]), k0[e], fmaf(Act<T>::to_f32(qf[r][ks][4 + e]), k1[e], dot));
                }
                dot += __shfl_xor(dot, 32, 64);
                mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], 32, 64));
                // (softmax is invariant under the shift: ANY common value near the maximum does; this one is representable in the operand type because it
                //  enters pass 2 through the matrix pipe, as a fifth k-step (-shift) x 1 of every score product)
                mrow[r] = Act<T>::to_f32(Act<T>::from_f32(fmaxf(mx[r], dot)));
                pt[r] = __builtin_amdgcn_exp2f(dot - mrow[r]);
            }
        }
        QA_MARK(7);
        {   // ---- pass 2: stage t = Q K^T of tile t + 1 (MFMAs 0-7), P V of tile t - 1 (8-15) and its row sums (16-19) beside the exponentials of tile t, in 20 slices
            f32x16 o[NB][2], lsum[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                lsum[r] = zero16;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[r][dt] = zero16;
            }
            vec8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = Act<T>::from_f32(1.0f);
            // the shift as a fifth k-step of every score product: K side = 1 in k-slot 0, Q side = -shift of the lane's query in k-slot 0, zeros elsewhere
            vec8 kaug, qaug[NB];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kaug[e] = Act<T>::from_f32(e == 0 && hi == 0 ? 1.0f : 0.0f);
#pragma unroll
                for (int r = 0; r < NB; ++r) qaug[r][e] = Act<T>::from_f32(e == 0 && hi == 0 ? -mrow[r] : 0.0f);
            }
            auto stage = [&](auto has_pv_c, auto has_qk_c, int t, f32x16 (&scur)[NB], f32x16 (&snext)[NB], vec8 (&pout)[NB][2], const vec8 (&pprev)[NB][2]) {
                constexpr bool HP = decltype(has_pv_c)::value, HQ = decltype(has_qk_c)::value;
                vec8 opnd[4];
                auto ld = [&](int q) {                                   // LDS operand q of this stage: 0-3 = K rows of tile t + 1 (ks = q), 4-7 = V^T rows of tile t - 1
                    if (q < 0 || q >= 8) return;
                    if (q < 4) { if (HQ) opnd[q & 3] = k_frag(t + 1, q); }
                    else if (HP) opnd[q & 3] = v_frag(t - 1, (q - 4) >> 1, (q - 4) & 1);
                };
                auto mf = [&](int i) {                                   // MFMAs 0-9: scores (k-steps 0-3 + the shift), 10-17: P V, 18-21: row sums of P; block i & 1
                    const int r = i & 1;
                    if (i < 10) {
                        const int j = i >> 1;
                        if (!HQ) return;
                        if (j == 0) Act<T>::mfma32_vgpr_zero(snext[r], opnd[0], qf[r][0]);
                        else if (j < 4) Act<T>::mfma32_vgpr_acc(snext[r], opnd[j & 3], qf[r][j]);
                        else Act<T>::mfma32_vgpr_acc(snext[r], kaug, qaug[r]);
                    } else if (i < 18) {
                        const int q = (i - 10) >> 1;
                        if (HP) o[r][q & 1] = Act<T>::mfma32(opnd[q & 3], pprev[r][q >> 1], o[r][q & 1]);
                    } else if (HP) lsum[r] = Act<T>::mfma32(ones, pprev[r][(i - 18) >> 1], lsum[r]);
                };
                ld(0);
                ld(1);
#pragma unroll
                for (int sl = 0; sl < 22; ++sl) {
                    if ((sl & 1) == 0 && sl <= 10) ld((sl >> 1) + 2);
                    mf(sl);
                    if (sl >= 2 && sl < 18) {                             // two weights per slice, rounded to the operand type at once: the score IS the exp2 argument
#pragma unroll
                        for (int f = 2 * (sl - 2); f < 2 * (sl - 2) + 2; ++f) {
                            const int r = f >> 4, e = f & 15;
                            pout[r][e >> 3][e & 7] = Act<T>::from_f32(__builtin_amdgcn_exp2f(scur[r][e]));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            typedef std::true_type Y;
            typedef std::false_type N_;
            f32x16 sA[NB], sB[NB];
            vec8 pA[NB][2], pB[NB][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const vec8 kf = k_frag(0, ks);
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    if (ks == 0) Act<T>::mfma32_vgpr_zero(sA[r], kf, qf[r][ks]); else Act<T>::mfma32_vgpr_acc(sA[r], kf, qf[r][ks]);
                }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) Act<T>::mfma32_vgpr_acc(sA[r], kaug, qaug[r]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");      // stage 0 reads sA with nothing in between: let the chain finish (72 wait states)
            stage(N_{}, Y{}, 0, sA, sB, pA, pB);
            stage(Y{}, Y{}, 1, sB, sA, pB, pA);
            stage(Y{}, Y{}, 2, sA, sB, pA, pB);
            stage(Y{}, Y{}, 3, sB, sA, pB, pA);
            stage(Y{}, Y{}, 4, sA, sB, pA, pB);
            stage(Y{}, Y{}, 5, sB, sA, pB, pA);
            stage(Y{}, Y{}, 6, sA, sB, pA, pB);
            stage(Y{}, N_{}, 7, sB, sA, pB, pA);
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                 // P V and the row sums of the last tile
                const vec8 vf = v_frag(7, j >> 1, j & 1);
#pragma unroll
                for (int r = 0; r < NB; ++r) o[r][j & 1] = Act<T>::mfma32(vf, pB[r][j >> 1], o[r][j & 1]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < NB; ++r) lsum[r] = Act<T>::mfma32(ones, pB[r][ks], lsum[r]);
            QA_MARK(8);
            {   // normalise (+ the odd key's rank-1 term: its value row, 8 x 4 dims per lane, is read once for both blocks) and store: a wave's 32 output rows of
                // a block go through 4 KB of LDS so that a row leaves as one 128-byte line (attention_vit257.hip), all 64 lanes in every instruction
                f32x4 vv[2][4];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) vv[dt][g] = *reinterpret_cast<const f32x4*>(sVl + dt * 32 + 8 * g + 4 * hi);
                char* so = smem + QA_OUT + wave * 4096;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const int vw = 2 * wave + r;
                    const float inv = __builtin_amdgcn_rcpf(lsum[r][0] + pt[r]), ptn = pt[r] * inv;      // every row of the ones product holds the query's sum over all 256 keys
                    T* obase = out + (row0 + vw * 32) * D + h * 64;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            vec4 w;
#pragma unroll
                            for (int e = 0; e < 4; ++e) w[e] = Act<T>::from_f32(fmaf(o[r][dt][4 * g + e], inv, ptn * vv[dt][g][e]));
                            *reinterpret_cast<vec4*>(so + l31 * 128 + (((dt * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = w;
                        }
                    asm volatile("" ::: "memory");                            // same wave, LDS in order: the reads below see the writes above
                    u32x4 ov[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = k * 8 + (lane >> 3), cc = lane & 7;
                        ov[k] = *reinterpret_cast<const u32x4*>(so + row * 128 + ((cc ^ (row & 7)) << 4));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = k * 8 + (lane >> 3), cc = lane & 7;
                        *reinterpret_cast<u32x4*>(obase + (long)row * D + cc * 8) = ov[k];
                    }
                    asm volatile("" ::: "memory");
                }
            }
        }
        QA_MARK(9);
        // ---- the odd query against keys 32 vw .. 32 vw + 31 on the MFMA pipe: a partial (max, sum, o[64]) per query block, merged after the barrier.
        // Both blocks step by step together; the 32-lane reductions are DPP butterflies (six ds_bpermute round trips each cost a lone wave 4 k cycles per item)
        {
            const char* sQh = reinterpret_cast<const char*>(sT + 192);
            const char* qsrc = l31 == 0 ? sQh + hi * 16 : sZero;
            const int qstep = l31 == 0 ? 32 : 0;
            // every LDS operand of this section is requested up front (a lone wave pays each round trip in full): the query's four fragments, the K rows of both
            // blocks, and the V^T rows the P V products will need once the weights exist
            vec8 qa[4], kfo[NB][4], vfo[NB][2][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qa[ks] = *reinterpret_cast<const vec8*>(qsrc + ks * qstep);
#pragma unroll
                for (int r = 0; r < NB; ++r) kfo[r][ks] = k_frag(2 * wave + r, ks);
            }
#pragma unroll
            for (int r = 0; r < NB; ++r)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) vfo[r][ks][dt] = v_frag(2 * wave + r, ks, dt);
            f32x16 s1[NB];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < NB; ++r) s1[r] = Act<T>::mfma32(qa[ks], kfo[r][ks], ks == 0 ? zero16 : s1[r]);
            float mw[NB], lw[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const float sv = hi == 0 ? s1[r][0] * sc : -INFINITY;
                mw[r] = half_wave_max(sv);
                const float pk = hi == 0 ? __builtin_amdgcn_exp2f(sv - mw[r]) : 0.f;
                lw[r] = half_wave_sum(pk);
                if (hi == 0) reinterpret_cast<T*>(sPw + (2 * wave + r) * 64)[(l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)] = Act<T>::from_f32(pk);      // the V^T image's key order
            }
            asm volatile("" ::: "memory");                                // same wave, LDS in order: the reads below see the writes above
            f32x16 oq[NB][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const char* psrc = l31 == 0 ? sPw + (2 * wave + r) * 64 + hi * 16 : sZero;
                    const vec8 pf = *reinterpret_cast<const vec8*>(psrc + ks * qstep);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) oq[r][dt] = Act<T>::mfma32(vfo[r][ks][dt], pf, ks == 0 ? zero16 : oq[r][dt]);
                }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                float* pp = sPart + (2 * wave + r) * QA_PARTF;
                if (l31 == 0) {                                           // column 0 of the product: 32 dims in lane 0, 32 in lane 32
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<f32x4*>(pp + dt * 32 + 8 * g + 4 * hi) = f32x4{oq[r][dt][4 * g], oq[r][dt][4 * g + 1], oq[r][dt][4 * g + 2], oq[r][dt][4 * g + 3]};
                    if (hi == 0) { pp[64] = mw[r]; pp[65] = lw[r]; }
                }
            }
            if (wave == 0) {                                              // the odd key's term of that row as the ninth partial
                float* p8 = sPart + 8 * QA_PARTF;
                float st = half_wave_sum(sQt[lane] * sKt[lane]);
                st = (st + __shfl_xor(st, 32, 64)) * sc;
                p8[lane] = sVl[lane];
                if (lane == 0) { p8[64] = st; p8[65] = 1.0f; }
            }
        }
        QA_MARK(5);
        __syncthreads();               // the K | V^T images are free (the next item's second K tile overwrites them); the partials are complete
        if (wave == 0) {               // the odd query's row out of its 9 partials
            const float* pp = sPart;
            float m = pp[64];
#pragma unroll
            for (int j = 1; j < 9; ++j) m = fmaxf(m, pp[j * QA_PARTF + 64]);
            float lsum = 0.f, ov = 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float w = __builtin_amdgcn_exp2f(pp[j * QA_PARTF + 64] - m);
                lsum = fmaf(w, pp[j * QA_PARTF + 65], lsum);
                ov = fmaf(w, pp[j * QA_PARTF + lane], ov);
            }
            out[(row0 + 256) * D + h * 64 + lane] = Act<T>::from_f32(ov / lsum);
        }
        QA_MARK(6);
        have = have_next;
        b = nb;
        h = nh;
    }
#undef QA_BARRIER
}

// row `row` of every tile: 16-bit [B*T][D] -> [B][D], and its (rstd, -mean rstd) pair; with `lo` (the second plane of a two-plane residual stream) and
// `normalize`, the gathered row leaves as the LayerNorm's normalised value (hi + lo) * rstd - mean * rstd, rounded once -- gamma and beta live in the
// folded weights -- so that a PLAIN GEMM (any kernel family) can follow
template <typename T>
__global__ void gather_token_rows16_kernel(const T* __restrict__ src, const T* __restrict__ lo, const float* __restrict__ stat, T* __restrict__ dst,
                                           float* __restrict__ dstat, int Tt, int D, int row, int normalize) {
    const long r = (long)blockIdx.x * Tt + row;
    if (!normalize) {
        const u32x4* s = reinterpret_cast<const u32x4*>(src + r * D);
        u32x4* d = reinterpret_cast<u32x4*>(dst + (long)blockIdx.x * D);
        for (int i = threadIdx.x; i < D / 8; i += blockDim.x) d[i] = s[i];
        if (stat && threadIdx.x < 2) dstat[2 * blockIdx.x + threadIdx.x] = stat[2 * r + threadIdx.x];
        return;
    }
    const float a = stat[2 * r], b = stat[2 * r + 1];
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float x = Act<T>::to_f32(src[r * D + i]) + (lo ? Act<T>::to_f32(lo[r * D + i]) : 0.f);
        dst[(long)blockIdx.x * D + i] = Act<T>::from_f32(fmaf(x, a, b));
    }
}

static int g_qa_cus = 0;

template <typename T>
static int launch_qkv_attn257(const void* X, const void* W, const float* bias, const float* colsum, const float* rowstat, const void* qkv_tail, void* out,
                              int B, int H, int D, hipStream_t st) {
    auto kern = qkv_attn257_kernel<T>;
    static bool attr_set = false;
    if (!attr_set) {
        AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS));
        attr_set = true;
    }
    if (!g_qa_cus) {
        int dev = 0;
        hipDeviceProp_t p;
        AMDS_HIP(hipGetDevice(&dev));
        AMDS_HIP(hipGetDeviceProperties(&p, dev));
        g_qa_cus = p.multiProcessorCount;
    }
    const int n_slots = (H & 7) == 0 ? ((B + 3) / 4) * 4 * H : B * H;
    int grid = g_qa_cus & ~7;                                   // a multiple of the 8 XCDs
    if (grid < 8) grid = 8;
    while (grid > 8 && (grid - 8) >= n_slots) grid -= 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), QA_LDS, st, (const T*)X, (const T*)W, bias, colsum, rowstat, (const T*)qkv_tail, (T*)out, B, H, D, n_slots);
    AMDS_LAUNCH_CHECK("qkv_attn257_kernel");
    return AMDS_OK;
}

}  // namespace amds

using namespace amds;

// see include/amdstamp.h
extern "C" int amds_qkv_attention_vit257(const void* x, const void* w_qkv, const float* bias, const float* colsum, const float* rowstat, const void* qkv_tail,
                                         void* out, int B, int heads, int dim, int dtype, void* stream) {
    AMDS_REQUIRE(x && w_qkv && bias && qkv_tail && out, "amds_qkv_attention_vit257: null pointer");
    AMDS_REQUIRE(B > 0 && heads > 0 && dim == heads * 64, "amds_qkv_attention_vit257: head_dim must be 64 (dim=%d, heads=%d)", dim, heads);
    AMDS_REQUIRE(dim % 64 == 0 && dim >= 128, "amds_qkv_attention_vit257: dim=%d must be a multiple of 64, >= 128", dim);
    AMDS_REQUIRE((long)3 * dim * dim * 2 < (1L << 31) && (long)256 * dim * 2 < (1L << 31), "amds_qkv_attention_vit257: dim=%d too large", dim);
    AMDS_REQUIRE((colsum == nullptr) == (rowstat == nullptr), "amds_qkv_attention_vit257: rowstat and colsum go together (folded LayerNorm) or are both null");
    AMDS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
                 (colsum == nullptr || ((uintptr_t)colsum & 15) == 0) && (rowstat == nullptr || ((uintptr_t)rowstat & 7) == 0),
                 "amds_qkv_attention_vit257: pointers must be 16-byte aligned");
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_qkv_attention_vit257: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const double flops = 2.0 * B * 256 * 3.0 * dim * dim + 4.0 * B * heads * 257.0 * 257.0 * 64;
    ProfScope prof(PROF_ATTN, flops, st);
    return dtype == AMDS_F16 ? launch_qkv_attn257<f16>(x, w_qkv, bias, colsum, rowstat, qkv_tail, out, B, heads, dim, st)
                             : launch_qkv_attn257<bf16>(x, w_qkv, bias, colsum, rowstat, qkv_tail, out, B, heads, dim, st);
}

extern "C" int amds_gather_token_rows16(const void* src16, const float* stat, void* dst16, float* dst_stat, int B, int T, int D, int row, void* stream) {
    return amds_gather_token_rows16_ex(src16, nullptr, stat, dst16, dst_stat, B, T, D, row, AMDS_F16, 0, stream);
}

extern "C" int amds_gather_token_rows16_ex(const void* src16, const void* lo16, const float* stat, void* dst16, float* dst_stat, int B, int T, int D, int row, int dtype,
                                           int normalize, void* stream) {
    AMDS_REQUIRE(src16 && dst16 && (normalize ? stat != nullptr : (stat == nullptr || dst_stat)), "amds_gather_token_rows16: null pointer");
    AMDS_REQUIRE(B > 0 && T > 0 && row >= 0 && row < T && D > 0 && D % 8 == 0, "amds_gather_token_rows16: bad shape B=%d T=%d D=%d row=%d", B, T, D, row);
    AMDS_REQUIRE(((uintptr_t)src16 & 15) == 0 && ((uintptr_t)dst16 & 15) == 0, "amds_gather_token_rows16: pointers must be 16-byte aligned");
    AMDS_REQUIRE(dtype == AMDS_F16 || dtype == AMDS_BF16, "amds_gather_token_rows16: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AMDS_F16) hipLaunchKernelGGL(gather_token_rows16_kernel<f16>, dim3(B), dim3(128), 0, st, (const f16*)src16, (const f16*)lo16, stat, (f16*)dst16, dst_stat, T, D, row, normalize);
    else hipLaunchKernelGGL(gather_token_rows16_kernel<bf16>, dim3(B), dim3(128), 0, st, (const bf16*)src16, (const bf16*)lo16, stat, (bf16*)dst16, dst_stat, T, D, row, normalize);
    AMDS_LAUNCH_CHECK("gather_token_rows16_kernel");
    return AMDS_OK;
}
