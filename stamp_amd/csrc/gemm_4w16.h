// gemm_4w16.h -- gemm_4w64.h (256 x 256 x 64 tiles, four waves, 128 x 128 wave tiles, two 64 KB LDS stages, buffer-form LDS-DMA)
// on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16.
//
// Why: under the socket power cap the matrix pipe's own draw sets the clock, and the 16x16x32 instruction is the cheaper one
// per flop -- tools/ubench/mfma_rate.hip, bare MFMA streams on pseudo-random operands, 2.5 s sustained: 32x32x16 1.74 PFLOP/s at
// 1.82 GHz / 1310 W, 16x16x32 2.05 PFLOP/s at 2.13 GHz / 1355 W (profiles/r01_gemm_vendor_and_power.txt).  The vendor kernel
// for these shapes uses it too (MI16x16x1).
//
// Same LDS image as gemm_4w64.h (128-byte rows, 16-byte chunk index XOR (row>>1)&7): a 16x16x32 operand fragment is 16 rows x
// 4 chunks (lane = row l15, chunk kb = lane>>4), and with that swizzle the 16 lanes of every ds_read_b128 cycle still fall into
// 16 distinct 16-byte bank slots.  Accumulators: 8 x 8 blocks of 16 x 16 (f32x4 per lane: row m = l15, columns 4 kb .. 4 kb + 3
// -- W is again the MFMA "A" operand).  A K tile is four units of 32 MFMAs (512 cycles each, like a k-step of gemm_4w64.h):
//     u0  MFMA(set 0, row blocks 0-3) | 8 reads of set 1 <- (kt, k 32..63) | LDS-DMA pieces P3 .. P3+P0 of tile kt+1
//     u1  MFMA(set 0, row blocks 4-7) | 8 reads of set 1                    | the remaining pieces of tile kt+1
//     u2  MFMA(set 1, row blocks 0-3)
//     lgkmcnt(0), vmcnt(0), s_barrier
//     u3  MFMA(set 1, row blocks 4-7) | 16 reads of set 0 <- (kt+1, k 0..31) | pieces 0 .. P3 of tile kt+2
// Results differ from the 32x32x16 kernels in the last bits (32 products are summed per instruction instead of 16).
#pragma once
#include <type_traits>
#include "gemm_kernel.h"

namespace amds {

// Accumulator block AGPR -> VGPR, explicitly and where it is used: left to the compiler, the copies of ALL 256 accumulators are
// placed right behind the asm block that last defines them (another basic block than the epilogue's uses), which leaves the
// epilogue no registers.
__device__ __forceinline__ f32x4 agpr_read(const f32x4& a) {
    f32x4 v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(a[0]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(a[1]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(a[2]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(a[3]));
    return v;
}

// TN (kernel id 15, weight gradients): both operands are stored TOKEN-major -- A = dY [tokens][lda], W = X [tokens][ldw] -- and the contraction runs over the
// tokens: C[M = dY columns][N = X columns] = sum_t dY[t][m] X[t][n], K = tokens per batch (split-K over blockIdx.y, fp32 partials).  A K tile is then 64 token
// rows x 512 B per operand: the LDS-DMA image is [64 rows][32 chunks of 16 B] (lane-linear as ever; the 32-byte pair index of a row is XOR-ed with
// g(row) = (row & 3) | ((row >> 1) & 4) on the source address), and a 16 x 32 MFMA fragment -- 8 consecutive TOKENS of one column per lane -- is two
// ds_read_b64_tr_b16 (gfx950's transpose read: 16 lanes hand in 4 rows x 32 B and each gets one column of the 4 x 16 block; tools/ubench/tr16_probe.hip),
// conflict-free by construction (a half-wave's 8 rows land in 8 different 32-byte bank slots).  No transposed copies of dY / X exist, and token rows past
// ep.ktot are out of the buffer descriptors' range and read as 0 -- no padded operand, no memset.
// PRL / PRS (AMDS_GEMM_PROBE builds only, kernel ids 21-23): the overlap probe of round 4.  Every K tile additionally issues PRL 16-byte
// loads and PRS 16-byte stores per lane against a scratch region (ep.pos), spread behind the MFMAs of the third unit, and the K loop waits for
// its operands with a COUNTED vmcnt so that this traffic stays in flight -- i.e. an epilogue's worth of HBM traffic perfectly overlapped with
// the matrix pipe, on top of the unchanged real epilogue.  T(probe) - T(id 12) = what overlapped epilogue bytes would still cost.
template <typename T, int EPI, int P3 = 6, int P0 = 6, bool SPREAD = true, int PRL = 0, int PRS = 0, bool TN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gemm_4w16_kernel(const T* __restrict__ A, long lda, const T* __restrict__ W, long ldw, int M, int N, int K,
                 EpiArgs ep, int tiles_m, int tiles_n) {
    typedef typename Act<T>::vec8 vec8;
    typedef typename Act<T>::vec4 vec4;
    constexpr int BM = 256, BN = 256, BK = 64, NT = 256;
    constexpr int ROWB = BK * 2;                                   // 128 bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;   // 32 KB, 64 KB
    constexpr int FI = 8, FJ = 8;                                  // 16 x 16 blocks per wave tile
    constexpr int GROUP_M = 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kb = lane >> 4;
    const int l31 = lane & 31, hi = lane >> 5;      // read-back of the staged tile (row-wise, as in gemm_4w64.h)

    int tm, tn;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int group = GROUP_M * tiles_n;
        const int g = t / group, first_m = g * GROUP_M;
        const int gm = min(tiles_m - first_m, GROUP_M);
        const int rr = t - g * group;
        tm = first_m + rr % gm;
        tn = rr / gm;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if (ep.nbatch > 1) {       // batched / split-K: operands and output of batch blockIdx.y (element strides)
        const long by = blockIdx.y;
        A += by * ep.bsA;
        W += by * ep.bsW;
        constexpr bool F32O = (EPI == AMDS_EPI_RESIDUAL || EPI == AMDS_EPI_BIAS_F32 || EPI == AMDS_EPI_BIAS_GELU_F32 || EPI == AMDS_EPI_BIAS_RELU_F32);
        if (F32O) ep.out = reinterpret_cast<float*>(ep.out) + by * ep.bsOut;
        else ep.out = reinterpret_cast<T*>(ep.out) + by * ep.bsOut;
    }

    // ---- copy addressing: 4096 16-byte chunks per K tile, 16 per thread (8 of A, 8 of W); chunk ^= (row>>1)&7 on the source.
    // Buffer form (buffer_load_dwordx4 ... offen lds): one 32-bit byte offset per piece, constant over the K loop, the K
    // advance in the scalar offset, no vector address arithmetic in the loop; rows past M are out of range and read as 0.
    const int rows_a = min(BM, M - m0);
    // TN: the tokens of this batch that exist (the rest of the K range lies past num_records and reads as 0)
    const long tn_valid = TN ? (ep.ktot > 0 ? min((long)K, ep.ktot - (long)blockIdx.y * K) : (long)K) : 0;
    const __amdgpu_buffer_rsrc_t rsrc_a = TN ? __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A + m0), 0, tn_valid > 0 ? (int)(((tn_valid - 1) * lda + BM) * 2) : 0, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A + (long)m0 * lda), 0, (int)((((long)rows_a - 1) * lda + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = TN ? __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W + n0), 0, tn_valid > 0 ? (int)(((tn_valid - 1) * ldw + BN) * 2) : 0, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W + (long)n0 * ldw), 0, (int)(((long)(BN - 1) * ldw + K) * 2), 0x00020000);
    int voff[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        if constexpr (TN) {       // LDS position c = token row c >> 5, 16-byte chunk c & 31 of the tile's 512-byte row; the source chunk is the swizzled one
            const int c = (it & 7) * NT + tid, row = c >> 5, pos = c & 31, g = (row & 3) | ((row >> 1) & 4);
            voff[it] = (int)(((long)row * (it < 8 ? lda : ldw) + (pos ^ (g << 1)) * 8) * 2);
        } else {
            const int c = (it & 7) * NT + tid, row = c >> 3, cp = c & 7, sc = cp ^ ((row >> 1) & 7);
            voff[it] = (int)(((long)row * (it < 8 ? lda : ldw) + sc * 8) * 2);
        }
    }
    auto issue_pieces = [&](int kt, int lo, int hi_) {
        char* st = smem + (kt & 1) * STAGE;
        const int koff = TN ? kt * BK * (int)lda * 2 : kt * BK * 2;
        const int koff_w = TN ? kt * BK * (int)ldw * 2 : koff;
#pragma unroll
        for (int it = 0; it < 16; ++it)
            if (it >= lo && it < hi_) {
                if (it < 8) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t)(st + ((it & 7) * NT + wave * 64) * 16), 16, voff[it], koff, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t)(st + A_BYTES + ((it & 7) * NT + wave * 64) * 16), 16, voff[it], koff_w,
                                                             0, 0);
                }
            }
    };

    const int swz = (l15 >> 1) & 7;
    const int a_off = (wm * 128 + l15) * ROWB;
    const int w_off = A_BYTES + (wn * 128 + l15) * ROWB;

    // The accumulators start from the bias (when there is one and acc_scale == 1): the epilogue then has no bias add (128 packed
    // adds per tile) and no bias loads to wait for; the 8 vectors are requested here and land under the first K tile's DMA.
    const bool lnf = ep.rowstat != nullptr;        // LayerNorm folded in: out = act(acc * rstd[m] + (colsum[n] * nmr[m] + bias[n]))
    const bool bias_in_acc = ep.bias != nullptr && ep.acc_scale == 1.0f && !lnf;
    f32x4 acc[FI][FJ];
    {
        f32x4 b0[FJ];
#pragma unroll
        for (int j = 0; j < FJ; ++j) b0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias_in_acc) {
#pragma unroll
            for (int j = 0; j < FJ; ++j) b0[j] = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wn * 128 + j * 16 + 4 * kb);
        }
        // the first K tile's LDS-DMA requests go out BEHIND the bias requests and BEFORE anything waits for the bias: the accumulator
        // initialisation then only waits for the bias (counted), and its round trip is hidden under the operand tile's instead of preceding it
        __builtin_amdgcn_sched_barrier(0);
        issue_pieces(0, 0, 16);
        if constexpr (P3 < 0) {
            if (K / BK > 1) issue_pieces(1, 0, 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = b0[j];
    }
    // The initial values must BE in their AGPRs well before the first inline-asm MFMA reads them: left alone, the compiler sinks
    // the v_accvgpr_write of an accumulator right in front of the asm that first uses it, and -- not knowing that asm is an MFMA --
    // without the wait states the hardware needs (wrong sums in single 16 x 16 blocks).
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        if (i == FI - 1)
            asm volatile("s_nop 7" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]),
                         "+a"(acc[i][6]), "+a"(acc[i][7]));
        else
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]),
                         "+a"(acc[i][6]), "+a"(acc[i][7]));
    }
    __builtin_amdgcn_sched_barrier(0);

    vec8 af[2][FI], wf[2][FJ];
    // fragments [lo, hi) of the 16 (8 A then 8 W) of k-half ks of tile kt into set s
    // TN: lane (l15, kb) of a fragment hands in the address of 4 contiguous columns of token row 8 kb + (l15 >> 2) [+ 4 for the second read] and receives
    // tokens 8 kb .. 8 kb + 3 [+ 4 ..] of column l15 of the 16-column block; the block's 32-byte pair sits at pair index (q ^ g), g as on the source side
    const int tn_g = (l15 >> 2) | ((kb & 1) << 2);
    const int tn_row = (kb * 8 + (l15 >> 2)) * 512 + (l15 & 1) * 8 + ((l15 >> 1) & 1) * 16;
    auto load_frags = [&](int kt, int ks, int s, int lo, int hi_) {
        const char* sb = smem + (kt & 1) * STAGE;
        if constexpr (TN) {
            // Inline asm, not __builtin_amdgcn_ds_read_tr16_b64: the compiler treats the builtin as possibly aliasing the LDS-DMA writes in flight and puts
            // an s_waitcnt vmcnt(0) in front of every pair of reads (2.2x the kernel's time).  The asm reads are invisible to its counters, so the
            // K loop waits for them itself (lgkmcnt(0) where a fragment set is first used: k_tile below); the two halves of a fragment are written straight
            // into the halves of its register quadruple (no moves: a move would read the registers before the data is there).  That property is the
            // compiler's to break: tools/check_tn_isa.py checks it on the compiled kernel (tests/test_cpu_isa.py), the GPU tests hold the binary bit-identical
            // to the transposed form; a first attempt to add compiler-visible MFMAs on these fragments (bias-gradient sums) made it move them early.
            const unsigned base = (unsigned)(size_t)(sb + tn_row + ks * 32 * 512);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q >= lo && q < hi_) {
                    const int qq = q & 7;
                    const unsigned pq = base + (q < 8 ? 0 : A_BYTES) + ((q < 8 ? wm : wn) * 16 + 2 * (qq ^ tn_g)) * 16;
                    u32x2 a0, a1;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0) : "v"(pq) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(a1) : "v"(pq) : "memory");
                    const vec8 f = __builtin_bit_cast(vec8, u32x4{a0[0], a0[1], a1[0], a1[1]});
                    if (q < 8) af[s][qq] = f; else wf[s][qq] = f;
                }
        } else {
            const int co = ((ks * 4 + kb) ^ swz) << 4;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q >= lo && q < hi_) {
                    if (q < 8) af[s][q] = *reinterpret_cast<const vec8*>(sb + a_off + q * 16 * ROWB + co);
                    else wf[s][q - 8] = *reinterpret_cast<const vec8*>(sb + w_off + (q - 8) * 16 * ROWB + co);
                }
        }
    };
    // A unit = 32 MFMAs (set s, row blocks 4 ih .. 4 ih + 3, all 8 column blocks) with, behind the first NR of them, one fragment
    // read each (fragments rlo .. rlo + NR - 1 of k-half rks of tile rkt into set rs) and behind the next NC one LDS-DMA piece each
    // (pieces clo .. of tile ckt).  The MFMAs are inline asm with the accumulators pinned to AGPRs: with the builtin, the register
    // allocator keeps part of the 64 f32x4 accumulators in VGPRs and shuffles them through AGPRs (380 v_accvgpr moves per K tile).
    // Inline asm is invisible to the sched_group_barrier masks, so the order is pinned with a scheduling barrier after every slot.
    auto unit = [&](int s, int ih, auto nr_c, int rkt, int rks, int rs, int rlo, auto nc_c, int ckt, int clo) {
        constexpr int NR = decltype(nr_c)::value, NC = decltype(nc_c)::value;
        // slot plan: 16 reads (the next tile's first k-half, needed by the very next unit) go behind the first 16 MFMAs and the
        // LDS-DMA pieces are spread over the rest; otherwise reads and pieces are both spread evenly over the 32 slots.  An
        // LDS-DMA request costs ~60 issue cycles on a quiet memory pipeline and 100-185 in a burst (MI355X guide), an MFMA 16.
        // (SPREAD = false: reads behind the first NR MFMAs, pieces behind the next NC -- the A/B baseline)
        constexpr int R_SPAN = SPREAD ? (NR >= 16 ? 16 : 32) : (NR > 0 ? NR : 1), C_LO = SPREAD ? (NR >= 16 ? 16 : 0) : NR,
                      C_SPAN = SPREAD ? 32 - C_LO : (NC > 0 ? NC : 1);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            Act<T>::mfma16_agpr(wf[s][m & 7], af[s][ih * 4 + (m >> 3)], acc[ih * 4 + (m >> 3)][m & 7]);
            if (NR > 0 && m < R_SPAN) {
                const int r0 = m * NR / R_SPAN, r1 = (m + 1) * NR / R_SPAN;
                if (r1 > r0) load_frags(rkt, rks, rs, rlo + r0, rlo + r1);
            }
            if (NC > 0 && m >= C_LO && m < C_LO + C_SPAN) {
                const int c0 = (m - C_LO) * NC / C_SPAN, c1 = (m - C_LO + 1) * NC / C_SPAN;
                if (c1 > c0) issue_pieces(ckt, clo + c0, clo + c1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 8> I8;
    typedef std::integral_constant<int, 16> I16;
    // SCHED 0 (P3 >= 0): tile kt+1 is requested while tile kt is consumed (P3 pieces in the last quarter of tile kt-1, P0 / P1 in the
    //   first two quarters of kt) and waited for with vmcnt(0) at 3/4 of tile kt: the late pieces have a quarter to half a K tile
    //   (~250-500 ns) to arrive -- enough for an L2 hit, not for an HBM miss, and a miss of one wave stalls all four at the barrier.
    // SCHED 2 (P3 = -2, kernel id 13, A/B): a K tile's fragments (both k-halves) are in registers by mid-tile, so its stage is free
    //   from there on: tile kt+2 is requested in the SECOND half of tile kt (behind a barrier at mid-tile) and tile kt+1's data
    //   (requested in the second half of kt-1) is waited for at 3/4 with the first 8 pieces of kt+2 still in flight (vmcnt(8)):
    //   0.75-1.25 tiles of flight time, as the vendor kernel does.  Measured +1 % on the K = 4096 / residual shapes, -0.5 % on qkv,
    //   waiting at mid-tile with one barrier +-0.5 %: the K loop does not wait for its operands (profiles/r02_gemm_schedules.txt).
    constexpr int SCHED = P3 < 0 ? -P3 : 0;
    constexpr int Q3 = SCHED ? 0 : P3, Q0 = SCHED ? 0 : P0, P1 = SCHED ? 0 : 16 - Q3 - Q0;
    static_assert((SCHED == 0 || SCHED == 2) && Q3 <= 8 && Q0 <= 8 && P1 >= 0 && P1 <= 8, "LDS-DMA piece split");
    typedef std::integral_constant<int, Q3> IP3;
    typedef std::integral_constant<int, Q0> IP0;
    typedef std::integral_constant<int, P1> IP1;

#define AMDS_BARRIER()                        \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    const int nk = K / BK;   // >= 1   (tile 0 -- and tile 1 with SCHED -- were requested above, behind the bias)
    if constexpr (SCHED) {
        if (nk > 1) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    AMDS_BARRIER();
    load_frags(0, 0, 0, 0, 16);
    if constexpr (!SCHED) {
        if (nk > 1) issue_pieces(1, 0, Q3);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- residual epilogue state (see the epilogue)
    constexpr bool RMW = (EPI == AMDS_EPI_RESIDUAL);
    const bool rmw_fast = RMW && (bias_in_acc || ep.bias == nullptr) && ep.acc_scale == 1.0f;
    const int ldo4 = (int)ep.ldo * 4;
    __amdgpu_buffer_rsrc_t rsrc_o = rsrc_a;              // (re-pointed at the output rows by the residual epilogue)
    int offu[RMW ? 2 : 1][RMW ? 8 : 1];
    f32x4 old[RMW ? 4 : 1][RMW ? 8 : 1];
    auto load_old = [&](int pass, int b8) {
        if constexpr (RMW) {
            int rowoff = b8 * 16 * ldo4 + pass * 256;
            asm volatile("" : "+s"(rowoff));          // keep the 16 lane offsets, not 64 sums, in registers
#pragma unroll
            for (int u = 0; u < 8; ++u)
                old[b8][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_o, offu[b8 & 1][u] + rowoff, 0, 0));
        }
    };

    // ---- overlap probe (PRL + PRS > 0): scratch traffic issued inside the K loop
    constexpr int PRN = PRL + PRS, PRM = PRL > PRS ? PRL : PRS;
    static_assert(PRN <= 8 && (PRN == 0 || SCHED == 0), "probe: at most 8 requests per K tile, SCHED 0 only");
    i32x4 pdesc = i32x4{0, 0, 0, 0};
    u32x4 pd[PRL > 0 ? PRL : 1];
    if constexpr (PRN > 0) {
        const long tile_bytes = (long)(K / BK) * 4096 * PRM;
        const unsigned long base = reinterpret_cast<unsigned long>(ep.pos) + (unsigned long)blockIdx.x * tile_bytes;
        pdesc = i32x4{(int)(unsigned)(base & 0xffffffffu), (int)(unsigned)((base >> 32) & 0xffffu), (int)tile_bytes, 0x00020000};
        pdesc[0] = __builtin_amdgcn_readfirstlane(pdesc[0]);
        pdesc[1] = __builtin_amdgcn_readfirstlane(pdesc[1]);
        pdesc[2] = __builtin_amdgcn_readfirstlane(pdesc[2]);
#pragma unroll
        for (int q = 0; q < (PRL > 0 ? PRL : 1); ++q) pd[q] = u32x4{0u, 0u, 0u, 0u};
    }
    const int pvoff = tid * 16;
    auto unit_probe = [&](int s, int ih, int kt) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            Act<T>::mfma16_agpr(wf[s][m & 7], af[s][ih * 4 + (m >> 3)], acc[ih * 4 + (m >> 3)][m & 7]);
            if constexpr (PRN > 0) {
                if ((m & 3) == 0 && (m >> 2) < PRN) {
                    const int q = m >> 2;
                    int soff = (kt * PRM + (q < PRL ? q : q - PRL)) * 4096;
                    if (q < PRL) {
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(pd[q < PRL ? q : 0]) : "v"(pvoff), "s"(pdesc), "s"(soff) : "memory");
                    } else {
                        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(wf[s][q & 7]), "v"(pvoff), "s"(pdesc), "s"(soff) : "memory");
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto k_tile = [&](int kt, auto next_c, auto next2_c) {
        constexpr bool NEXT = decltype(next_c)::value, NEXT2 = decltype(next2_c)::value;
        if constexpr (TN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // set 0 of this tile (read under the previous tile's last unit)
        if constexpr (SCHED == 0 && PRN > 0) {
            if constexpr (NEXT) unit(0, 0, I8{}, kt, 1, 1, 0, IP0{}, kt + 1, Q3); else unit(0, 0, I8{}, kt, 1, 1, 0, I0{}, 0, 0);
            if constexpr (NEXT) unit(0, 1, I8{}, kt, 1, 1, 8, IP1{}, kt + 1, Q3 + Q0); else unit(0, 1, I8{}, kt, 1, 1, 8, I0{}, 0, 0);
            unit_probe(1, 0, kt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PRN) : "memory");   // the probe's requests of THIS tile may stay in flight
            AMDS_BARRIER();
            if constexpr (NEXT2) unit(1, 1, I16{}, kt + 1, 0, 0, 0, IP3{}, kt + 2, 0);
            else if constexpr (NEXT) unit(1, 1, I16{}, kt + 1, 0, 0, 0, I0{}, 0, 0);
            else unit(1, 1, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
        } else if constexpr (SCHED == 0) {
            if constexpr (NEXT) unit(0, 0, I8{}, kt, 1, 1, 0, IP0{}, kt + 1, Q3); else unit(0, 0, I8{}, kt, 1, 1, 0, I0{}, 0, 0);
            if constexpr (NEXT) unit(0, 1, I8{}, kt, 1, 1, 8, IP1{}, kt + 1, Q3 + Q0); else unit(0, 1, I8{}, kt, 1, 1, 8, I0{}, 0, 0);
            if constexpr (TN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // set 1 (read under the two units above)
            unit(1, 0, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (last tile: no LDS-DMA request is outstanding)
            AMDS_BARRIER();
            if constexpr (NEXT2) unit(1, 1, I16{}, kt + 1, 0, 0, 0, IP3{}, kt + 2, 0);
            else if constexpr (NEXT) unit(1, 1, I16{}, kt + 1, 0, 0, 0, I0{}, 0, 0);
            else unit(1, 1, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
        } else {
            unit(0, 0, I8{}, kt, 1, 1, 0, I0{}, 0, 0);
            unit(0, 1, I8{}, kt, 1, 1, 8, I0{}, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // this tile's fragments are all in registers
            if constexpr (NEXT2) {
                AMDS_BARRIER();                                                   // the stage of tile kt is free
                unit(1, 0, I0{}, 0, 0, 0, 0, I8{}, kt + 2, 0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  // tile kt+1 has landed, 8 pieces of kt+2 may fly
                AMDS_BARRIER();
                unit(1, 1, I16{}, kt + 1, 0, 0, 0, I8{}, kt + 2, 8);
            } else if constexpr (NEXT) {
                unit(1, 0, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                AMDS_BARRIER();
                unit(1, 1, I16{}, kt + 1, 0, 0, 0, I0{}, 0, 0);
            } else {
                unit(1, 0, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
                unit(1, 1, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
            }
        }
    };
    int kt = 0;
    for (; kt < nk - 2; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
    if (nk >= 2) k_tile(kt++, std::true_type{}, std::false_type{});
    // LayerNorm-folded consumer: per-row (rstd, -mean*rstd) of this lane's 8 rows and the per-column (bias, colsum) vectors are
    // requested before the last K tile (which issues no LDS-DMA and waits for none), so the epilogue finds them in registers.
    constexpr bool LNC = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_SWIGLU);
    f32x2 rs[LNC ? FI : 1];
    f32x4 cb[LNC ? FJ : 1], cs[LNC ? FJ : 1];
    if constexpr (LNC) {
        if (lnf) {
#pragma unroll
            for (int i = 0; i < FI; ++i)
                rs[i] = *reinterpret_cast<const f32x2*>(ep.rowstat + 2 * (long)min(m0 + wm * 128 + i * 16 + l15, M - 1));
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                cb[j] = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wn * 128 + j * 16 + 4 * kb);
                cs[j] = *reinterpret_cast<const f32x4*>(ep.colsum + n0 + wn * 128 + j * 16 + 4 * kb);
            }
        }
    }
    k_tile(kt, std::false_type{}, std::false_type{});
    if constexpr (PRN > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < (PRL > 0 ? PRL : 1); ++q) asm volatile("" ::"v"(pd[q]));
    }
    AMDS_BARRIER();                 // every wave is done with the LDS stages
#undef AMDS_BARRIER
    // MFMA result -> accumulator read hazard: the compiler does not know the inline asm is an MFMA, so it neither pads the read of
    // a result nor keeps it away -- left alone it copies an accumulator to VGPRs right behind the asm that last wrote it (one
    // s_nop later: the copy returns the value from BEFORE that MFMA).  Passing every accumulator through these asm statements as
    // an AGPR operand keeps all of them in AGPRs until the nops have been executed.
#pragma unroll
    for (int i = 0; i < FI; ++i) {
        if (i == 0)
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]),
                         "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
        else
            asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]),
                         "+a"(acc[i][6]), "+a"(acc[i][7]));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: LDS-staged, coalesced.  Block (i, j): row = wm*128 + 16 i + l15, columns wn*128 + 16 j + 4 kb .. + 3 ----
    constexpr bool F16OUT = (EPI == AMDS_EPI_BIAS || EPI == AMDS_EPI_BIAS_GELU || EPI == AMDS_EPI_BIAS_RELU);
    static_assert(epi_is_staged<EPI>() || EPI == AMDS_EPI_SWIGLU, "gemm_4w16: staged epilogues and SWIGLU");
    if constexpr (EPI == AMDS_EPI_SWIGLU) {
        // Packed fc1 of a SwiGLU MLP (ops.pack_swiglu_rows): columns [64 q, 64 q + 32) are the gates and [64 q + 32, 64 q + 64) the values
        // of hidden units [32 q, 32 q + 32).  In 16 x 16 blocks: gate block 4 q + t pairs with value block 4 q + 2 + t (t = 0, 1) in the
        // SAME lane, hidden unit 32 q + 16 t + 4 kb + r.  out [M, N/2] (16-bit): 128 hidden units per tile and row = 256 B, staged as
        // 256 rows x 256 B (16-byte chunk index XOR row & 15) and stored row-wise.  The bias is in the accumulators (host requires one).
        const float as = ep.acc_scale;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            const int row = wm * 128 + i * 16 + l15;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 gt = agpr_read(acc[i][4 * q + t]), vl = agpr_read(acc[i][4 * q + 2 + t]);
                    if (lnf) {
                        gt = gt * rs[i][0] + (cs[4 * q + t] * rs[i][1] + cb[4 * q + t]);
                        vl = vl * rs[i][0] + (cs[4 * q + 2 + t] * rs[i][1] + cb[4 * q + 2 + t]);
                    } else if (!bias_in_acc) {
                        const f32x4 bg = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wn * 128 + (4 * q + t) * 16 + 4 * kb);
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(ep.bias + n0 + wn * 128 + (4 * q + 2 + t) * 16 + 4 * kb);
                        gt = gt * as + bg;
                        vl = vl * as + bv;
                    }
                    f32x4 o;
                    {
                        const f32x2 o01 = silu_mul2(f32x2{gt[0], gt[1]}, f32x2{vl[0], vl[1]}), o23 = silu_mul2(f32x2{gt[2], gt[3]}, f32x2{vl[2], vl[3]});
                        o = f32x4{o01[0], o01[1], o23[0], o23[1]};
                    }
                    const int hcol = wn * 64 + q * 32 + t * 16 + 4 * kb;          // hidden column inside the tile (0..127)
                    const int chunk = hcol >> 3, half = ((hcol >> 2) & 1) * 8;
                    *reinterpret_cast<vec4*>(smem + row * 256 + ((chunk ^ (row & 15)) << 4) + half) = Act<T>::from_f32x4(o);
                }
        }
        __syncthreads();
        const int l15r = lane & 15, rsub = lane >> 4;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int row = wave * 64 + it * 4 + rsub;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * 256 + l15r * 16);
            const int chunk = l15r ^ (row & 15);
            if (m0 + row < M)
                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 / 2 + chunk * 8) = v;
        }
        return;
    }
    if constexpr (EPI == AMDS_EPI_RESIDUAL) {
        // Fast read-modify-write (bias already in the accumulators or absent, acc_scale == 1 -- every call of the tile encoder).
        // The OLD values of a whole pass (this wave's 64 rows x 128 columns = 32 x 16 bytes per lane) are requested BEFORE the
        // accumulators are staged, and pass 1's as soon as pass 0 has consumed a batch of registers.  Requested batch by batch
        // behind the LDS read-back (8 serial L2 / HBM round trips per tile) this epilogue cost ~20 us per tile against 31 us for
        // proj's whole K loop.  Buffer addressing (32-bit offsets, no clamps): rows past M lie beyond num_records -- their loads
        // return 0, their stores are dropped.  Row of (b8, u): wave*64 + 16 b8 + 2 u + hi; its 16-byte chunk l31 ^ (row & 31) splits
        // bitwise into the 128-column half (l31 >> 4) ^ (b8 & 1) and the 4-column group (l31 & 15) ^ (2 u + hi).
        // Raw s_barrier instead of __syncthreads(): the fence of the latter waits for the loads in flight (vmcnt(0)).
        if (rmw_fast) {
            const int rows_o = min(BM, M - m0);
            rsrc_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(ep.out) + (long)m0 * ep.ldo + n0, 0,
                                                       (int)((((long)rows_o - 1) * ep.ldo + BN) * 4), 0x00020000);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    offu[h][u] = (wave * 64 + 2 * u + hi) * ldo4 + (((l31 >> 4) ^ h) << 9) + (((l31 & 15) ^ (2 * u + hi)) << 4);
            const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(
                ep.xh ? reinterpret_cast<T*>(ep.xh) + (long)m0 * ep.ldo + n0 : reinterpret_cast<T*>(ep.out), 0,
                ep.xh ? (int)((((long)rows_o - 1) * ep.ldo + BN) * 2) : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsrc_l = __builtin_amdgcn_make_buffer_rsrc(
                ep.xl ? reinterpret_cast<T*>(ep.xl) + (long)m0 * ep.ldo + n0 : reinterpret_cast<T*>(ep.out), 0,
                ep.xl ? (int)((((long)rows_o - 1) * ep.ldo + BN) * 2) : 0, 0x00020000);
            const int np_part = N / 128;
            auto run = [&](auto has_scale_c, auto lnp_c, auto pl_c) {
                constexpr bool HS = decltype(has_scale_c)::value, LNP = decltype(lnp_c)::value, PL = decltype(pl_c)::value;
                // PL: the residual rows live in two 16-bit planes (hi = the next GEMM's A operand, lo = x - hi): the old value of a lane's 4
                // columns is 8 + 8 bytes instead of 16, and nothing but the planes is written back (10 -> 8 bytes per element moved)
                auto load_prev = [&](int pass, int b8) {
                    if constexpr (PL) {
                        int rowoff = b8 * 16 * ldo4 + pass * 256;
                        asm volatile("" : "+s"(rowoff));
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int o = (offu[b8 & 1][u] + rowoff) >> 1;
                            const u32x2 ph = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_h, o, 0, 0));
                            const u32x2 pl = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_l, o, 0, 0));
                            old[b8][u] = __builtin_bit_cast(f32x4, u32x4{ph[0], ph[1], pl[0], pl[1]});
                        }
                    } else {
                        load_old(pass, b8);
                    }
                };
                f32x4 sc[4];
                auto load_scale = [&](int pass) {
                    if constexpr (HS) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            sc[jj] = *reinterpret_cast<const f32x4*>(ep.scale + n0 + wn * 128 + (pass * 4 + jj) * 16 + 4 * kb);
                    }
                };
                load_scale(0);
#pragma unroll
                for (int b8 = 0; b8 < 4; ++b8) load_prev(0, b8);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                    for (int i = 0; i < FI; ++i) {
                        const int row = wm * 128 + i * 16 + l15;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            f32x4 v = agpr_read(acc[i][pass * 4 + jj]);
                            if constexpr (HS) v *= sc[jj];
                            const int chunk = wn * 16 + jj * 4 + kb;       // 16-byte chunk = 4 fp32 columns of this pass's 128
                            *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = v;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (pass == 0) load_scale(1);
#pragma unroll
                    for (int b8 = 0; b8 < 4; ++b8) {
                        f32x4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                            v[u] = *reinterpret_cast<const f32x4*>(smem + row * 512 + l31 * 16);
                        }
                        int rowoff = b8 * 16 * ldo4 + pass * 256;
                        asm volatile("" : "+s"(rowoff));
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if constexpr (PL) {
                                const u32x4 o = __builtin_bit_cast(u32x4, old[b8][u]);
                                // (hi + lo is exact in fp32.  fp16 planes: one v_fma_mix_f32 per element forms it straight from the two 16-bit halves -- the
                                //  conversions were 2 of the ~9 vector instructions per element of this un-overlapped epilogue; same bits)
                                const vec4 h4 = __builtin_bit_cast(vec4, u32x2{o[0], o[1]}), l4 = __builtin_bit_cast(vec4, u32x2{o[2], o[3]});
                                f32x4 old4;
                                if constexpr (std::is_same<T, f16>::value) {
                                    old4 = f32x4{mix_add_hh<0, 0>(o[0], o[2]), mix_add_hh<1, 1>(o[0], o[2]), mix_add_hh<0, 0>(o[1], o[3]), mix_add_hh<1, 1>(o[1], o[3])};
                                } else {
                                    old4 = __builtin_convertvector(h4, f32x4) + __builtin_convertvector(l4, f32x4);
                                }
                                v[u] += old4;
                            } else {
                                v[u] += old[b8][u];
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[u]), rsrc_o, offu[b8 & 1][u] + rowoff, 0, 0);
                            }
                        }
                        if constexpr (LNP) {
                            // 16-bit copy of the updated rows + this pass's partial row sums (the LayerNorm that follows is folded into
                            // the next GEMM, which reads the copy as its A operand)
                            f32x2 ss[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const vec4 c = Act<T>::from_f32x4(v[u]);
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c), rsrc_h, (offu[b8 & 1][u] + rowoff) >> 1, 0, 0);
                                if constexpr (PL) {
                                    f32x4 rest;
                                    if constexpr (std::is_same<T, f16>::value) {                  // v - hi, exact: one v_fma_mix_f32 per element
                                        const u32x2 cu = __builtin_bit_cast(u32x2, c);
                                        rest = f32x4{mix_sub_fh<0>(v[u][0], cu[0]), mix_sub_fh<1>(v[u][1], cu[0]), mix_sub_fh<0>(v[u][2], cu[1]), mix_sub_fh<1>(v[u][3], cu[1])};
                                    } else {
                                        rest = v[u] - __builtin_convertvector(c, f32x4);
                                    }
                                    const vec4 cl = Act<T>::from_f32x4(rest);
                                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, cl), rsrc_l, (offu[b8 & 1][u] + rowoff) >> 1, 0, 0);
                                }
                                ss[u] = f32x2{(v[u][0] + v[u][1]) + (v[u][2] + v[u][3]),
                                              fmaf(v[u][0], v[u][0], v[u][1] * v[u][1]) + fmaf(v[u][2], v[u][2], v[u][3] * v[u][3])};
                            }
                            const f32x2 tot = half_wave_sum8(ss, l31);       // row u = 4 bit4 + 2 bit3 + bit2 of l31, complete in every lane
                            const int urow = ((l31 >> 4) & 1) * 4 + ((l31 >> 3) & 1) * 2 + ((l31 >> 2) & 1);
                            const int prow = m0 + wave * 64 + b8 * 16 + 2 * urow + hi;
                            if ((l31 & 3) == 0 && prow < M)
                                *reinterpret_cast<f32x2*>(ep.rowpart + ((long)prow * np_part + tn * 2 + pass) * 2) = tot;
                        }
                        if (pass == 0) load_prev(1, b8);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (pass == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every wave has read pass 0 back before it is overwritten
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            if (ep.xl) {
                if (ep.scale) run(std::true_type{}, std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}, std::true_type{});
            } else if (ep.xh) {
                if (ep.scale) run(std::true_type{}, std::true_type{}, std::false_type{}); else run(std::false_type{}, std::true_type{}, std::false_type{});
            } else {
                if (ep.scale) run(std::true_type{}, std::false_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}, std::false_type{});
            }
            return;
        }
    }
    constexpr int NPASS = F16OUT ? 1 : 2;
    constexpr int JP = FJ / NPASS;                 // column blocks per pass
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass) __syncthreads();
        EpiArgs epv = ep;                           // what the value transform still has to apply
        if (bias_in_acc || lnf) epv.bias = nullptr;
        EpiCols<JP> cols;                           // [jj]: columns n0 + wn*128 + 16 (pass*JP + jj) + 4 kb
        epi_cols_load<EPI>(epv, cols, [&](int jj) { return n0 + wn * 128 + (pass * JP + jj) * 16 + 4 * kb; });
        auto values = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int row = wm * 128 + i * 16 + l15;
#pragma unroll
                for (int jj = 0; jj < JP; jj += 2) {
                    const int j = pass * JP + jj;
                    f32x4 v0 = agpr_read(acc[i][j]), v1 = agpr_read(acc[i][j + 1]);
                    epi_value_pair<EPI, FAST, true>(epv, cols.bias[jj], cols.scale[jj], cols.bias[jj + 1], cols.scale[jj + 1], v0, v1);
                    if constexpr (F16OUT) {
                        const vec4 o0 = Act<T>::from_f32x4(v0), o1 = Act<T>::from_f32x4(v1);
                        // 16-byte chunk of the 512-byte row: wn*16 + 2 j + (kb>>1); the 8-byte half (kb & 1) is XOR-ed with row bit 3
                        const int chunk = wn * 16 + j * 2 + (kb >> 1);
                        const int half = ((kb & 1) ^ ((l15 >> 3) & 1)) * 8;
                        *reinterpret_cast<vec4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4) + half) = o0;
                        *reinterpret_cast<vec4*>(smem + row * 512 + (((chunk + 2) ^ (row & 31)) << 4) + half) = o1;
                    } else {
                        const int chunk = wn * 16 + jj * 4 + kb;       // 16-byte chunk = 4 fp32 columns of this pass's 128
                        *reinterpret_cast<f32x4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4)) = v0;
                        *reinterpret_cast<f32x4*>(smem + row * 512 + (((chunk + 4) ^ (row & 31)) << 4)) = v1;
                    }
                }
            }
        };
        auto values_ln = [&]() {
            if constexpr (LNC && F16OUT) {
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    const int row = wm * 128 + i * 16 + l15;
#pragma unroll
                    for (int j = 0; j < FJ; j += 2) {
                        f32x4 v0 = agpr_read(acc[i][j]), v1 = agpr_read(acc[i][j + 1]);
                        v0 = v0 * rs[i][0] + (cs[j] * rs[i][1] + cb[j]);
                        v1 = v1 * rs[i][0] + (cs[j + 1] * rs[i][1] + cb[j + 1]);
                        epi_value_pair<EPI, true, true>(epv, cb[j], cb[j], cb[j + 1], cb[j + 1], v0, v1);      // activation only
                        const vec4 o0 = Act<T>::from_f32x4(v0), o1 = Act<T>::from_f32x4(v1);
                        const int chunk = wn * 16 + j * 2 + (kb >> 1);
                        const int half = ((kb & 1) ^ ((l15 >> 3) & 1)) * 8;
                        *reinterpret_cast<vec4*>(smem + row * 512 + ((chunk ^ (row & 31)) << 4) + half) = o0;
                        *reinterpret_cast<vec4*>(smem + row * 512 + (((chunk + 2) ^ (row & 31)) << 4) + half) = o1;
                    }
                }
            }
        };
        if (LNC && F16OUT && lnf) values_ln();
        else if (F16OUT && (bias_in_acc || (ep.bias == nullptr && ep.acc_scale == 1.0f))) values(std::true_type{}); else values(std::false_type{});
        __syncthreads();
        // one wave per SIMD: batch 8 rows (reads first, then the stores) so the LDS / L2 latencies overlap
#pragma unroll 1
        for (int b8 = 0; b8 < 4; ++b8) {
            if constexpr (F16OUT) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    v[u] = *reinterpret_cast<const u32x4*>(smem + row * 512 + l31 * 16);
                    if ((u >> 2) & 1) v[u] = u32x4{v[u][2], v[u][3], v[u][0], v[u][1]};      // row bit 3 set: halves stored swapped
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    const int chunk = l31 ^ (row & 31);
                    if (m0 + row < M)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(ep.out) + (long)(m0 + row) * ep.ldo + n0 + chunk * 8) = v[u];
                }
            } else {
                f32x4 v[8], o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    const int chunk = l31 ^ (row & 31);
                    const int n = n0 + (chunk >> 4) * 128 + pass * 64 + (chunk & 15) * 4;
                    v[u] = *reinterpret_cast<const f32x4*>(smem + row * 512 + l31 * 16);
                    if constexpr (EPI == AMDS_EPI_RESIDUAL)
                        o[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ep.out) +
                                                               (long)min(m0 + row, M - 1) * ep.ldo + n);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = wave * 64 + (b8 * 8 + u) * 2 + hi;
                    const int chunk = l31 ^ (row & 31);
                    const int n = n0 + (chunk >> 4) * 128 + pass * 64 + (chunk & 15) * 4;
                    if constexpr (EPI == AMDS_EPI_RESIDUAL) v[u] += o[u];
                    if (m0 + row < M)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (long)(m0 + row) * ep.ldo + n) = v[u];
                }
            }
        }
    }
}

template <typename T, int EPI, bool SPREAD = true, int P3 = 6, int P0 = 6, int PRL = 0, int PRS = 0, bool TN = false>
static int launch_gemm_4w16(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const EpiArgs& ep,
                            hipStream_t st) {
    if constexpr (!epi_is_staged<EPI>() && EPI != AMDS_EPI_SWIGLU) {
        return launch_gemm_4w64<T, EPI>(A, lda, W, ldw, M, N, K, ep, st);
    } else {
        constexpr int LDS = 2 * (256 + 256) * 128;
        auto kern = gemm_4w16_kernel<T, EPI, P3, P0, SPREAD, PRL, PRS, TN>;
        EpiArgs epp = ep;
        if constexpr (PRL + PRS > 0) {      // probe scratch: one region per workgroup, never read by anything real
            static char* scratch = nullptr;
            static size_t scratch_bytes = 0;
            const size_t need = (size_t)cdiv(M, 256) * (N / 256) * (K / 64) * 4096 * (PRL > PRS ? PRL : PRS);
            if (need > scratch_bytes) {
                if (scratch) AMDS_HIP(hipFree(scratch));
                AMDS_HIP(hipMalloc(reinterpret_cast<void**>(&scratch), need));
                scratch_bytes = need;
            }
            epp.pos = reinterpret_cast<const float*>(scratch);
        }
        static bool attr_set = false;
        if (!attr_set) {
            AMDS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr_set = true;
        }
        const int tiles_m = cdiv(M, 256), tiles_n = N / 256;
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, ep.nbatch), dim3(256), LDS, st, reinterpret_cast<const T*>(A), lda,
                           reinterpret_cast<const T*>(W), ldw, M, N, K, epp, tiles_m, tiles_n);
        AMDS_LAUNCH_CHECK("gemm_4w16_kernel");
        return AMDS_OK;
    }
}

}  // namespace amds
