// K loop of the fused qkv + attention kernel ALONE (stamp_amd/csrc/qkv_attn257.hip, G phase: 256 x 192 x D per item, two LDS stages, buffer-form LDS-DMA,
// inline-asm v_mfma_f32_16x16x32), with four waves (one per SIMD, 128 x 96 wave tiles) or eight (two per SIMD, 64 x 96 wave tiles): does a second wave per
// SIMD cover the ~60 issue cycles of every LDS-DMA request (14 per wave and K tile) that cost the four-wave form 0.36 of its time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iinclude -Istamp_amd/csrc tools/ubench/qa_kloop.hip -o tools/ubench/qa_kloop
#include "../../stamp_amd/csrc/common.h"
#include <cstdio>
#include <vector>
namespace amds { thread_local char g_err[512]; void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -2; } }
using namespace amds;

constexpr int XB = 256 * 128, STAGE = (256 + 192) * 128;

template <int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
kloop_kernel(const f16* __restrict__ X, const f16* __restrict__ W, float* __restrict__ sink, int B, int H, int D) {
    typedef f16x8 vec8;
    constexpr int NT = NW * 64, RB = 512 / NW, FI = RB / 16, FJ = 6, NF = FI + FJ, NM = 24, PX = 2048 / NT, PW = 1536 / NT, NP = PX + PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, kb = lane >> 4, nk = D / 64;
    int voffx[PX], voffw[PW];
#pragma unroll
    for (int it = 0; it < PX; ++it) { const int c = it * NT + tid, row = c >> 3, cp = c & 7, scn = cp ^ ((row >> 1) & 7); voffx[it] = (row * D + scn * 8) * 2; }
#pragma unroll
    for (int it = 0; it < PW; ++it) { const int c = it * NT + tid, row = c >> 3, cp = c & 7, scn = cp ^ ((row >> 1) & 7); voffw[it] = (((row >> 6) * D + (row & 63)) * D + scn * 8) * 2; }
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(W), 0, 3 * D * D * 2, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_x = rsrc_w;
    int wsoff = 0;
    auto issue_pieces = [&](int kt, int lo, int hi_) {
        char* st = smem + (kt & 1) * STAGE;
        const int koff = kt * 128;
#pragma unroll
        for (int it = 0; it < NP; ++it)
            if (it >= lo && it < hi_) {
                if (it < PX) bufl16(rsrc_x, st + (it * NT + wave * 64) * 16, voffx[it], koff);
                else bufl16(rsrc_w, st + XB + ((it - PX) * NT + wave * 64) * 16, voffw[it - PX], koff + wsoff);
            }
    };
    const int swz = (l15 >> 1) & 7, a_off = (wm * RB + l15) * 128;
    int w_off[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) w_off[j] = XB + ((j >> 1) * 64 + (2 * wn + (j & 1)) * 16 + l15) * 128;
    vec8 af[2][FI], wf[2][FJ];
    auto load_frags = [&](int kt, int ks, int s, int lo, int hi_) {
        const char* sb = smem + (kt & 1) * STAGE;
        const int co = ((ks * 4 + kb) ^ swz) << 4;
#pragma unroll
        for (int q = 0; q < NF; ++q)
            if (q >= lo && q < hi_) {
                if (q < FI) af[s][q] = *reinterpret_cast<const vec8*>(sb + a_off + q * 16 * 128 + co);
                else wf[s][q - FI] = *reinterpret_cast<const vec8*>(sb + w_off[q - FI] + co);
            }
    };
    f32x4 acc[FI][FJ];
    auto unit = [&](int s, int ih, auto nr_c, int rkt, int rks, int rs_, int rlo, auto nc_c, int ckt, int clo) {
        constexpr int NR = decltype(nr_c)::value, NC = decltype(nc_c)::value;
        constexpr int R_SPAN = NR >= NF ? NF : NM, C_LO = NR >= NF ? NF : 0, C_SPAN = NM - C_LO;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int i = ih * 4 + m / FJ, j = m % FJ;
            if (j < 4) Act<f16>::mfma16_agpr(wf[s][j], af[s][i], acc[i][j]); else Act<f16>::mfma16_agpr(af[s][i], wf[s][j], acc[i][j]);
            if (NR > 0 && m < R_SPAN) { const int r0 = m * NR / R_SPAN, r1 = (m + 1) * NR / R_SPAN; if (r1 > r0) load_frags(rkt, rks, rs_, rlo + r0, rlo + r1); }
            if (NC > 0 && m >= C_LO && m < C_LO + C_SPAN) { const int c0 = (m - C_LO) * NC / C_SPAN, c1 = (m - C_LO + 1) * NC / C_SPAN; if (c1 > c0) issue_pieces(ckt, clo + c0, clo + c1); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    auto k_tile = [&](int kt, auto next_c, auto next2_c) {
        constexpr bool NEXT = decltype(next_c)::value, NEXT2 = decltype(next2_c)::value;
        if constexpr (NW == 4) {
            typedef std::integral_constant<int, 7> I7; typedef std::integral_constant<int, 14> I14; typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5;
            if constexpr (NEXT) unit(0, 0, I7{}, kt, 1, 1, 0, I5{}, kt + 1, 4); else unit(0, 0, I7{}, kt, 1, 1, 0, I0{}, 0, 0);
            if constexpr (NEXT) unit(0, 1, I7{}, kt, 1, 1, 7, I5{}, kt + 1, 9); else unit(0, 1, I7{}, kt, 1, 1, 7, I0{}, 0, 0);
            unit(1, 0, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BAR();
            if constexpr (NEXT2) unit(1, 1, I14{}, kt + 1, 0, 0, 0, I4{}, kt + 2, 0);
            else if constexpr (NEXT) unit(1, 1, I14{}, kt + 1, 0, 0, 0, I0{}, 0, 0);
            else unit(1, 1, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
        } else {      // eight waves: a K tile = two units (k-half 0 on set 0, k-half 1 on set 1), the barrier between them
            typedef std::integral_constant<int, 10> I10; typedef std::integral_constant<int, 3> I3; typedef std::integral_constant<int, 4> I4;
            if constexpr (NEXT) unit(0, 0, I10{}, kt, 1, 1, 0, I4{}, kt + 1, 3); else unit(0, 0, I10{}, kt, 1, 1, 0, I0{}, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BAR();
            if constexpr (NEXT2) unit(1, 0, I10{}, kt + 1, 0, 0, 0, I3{}, kt + 2, 0);
            else if constexpr (NEXT) unit(1, 0, I10{}, kt + 1, 0, 0, 0, I0{}, 0, 0);
            else unit(1, 0, I0{}, 0, 0, 0, 0, I0{}, 0, 0);
        }
    };
    constexpr int P3 = NW == 4 ? 4 : 3;
    const int n_items = B * H;
#pragma unroll 1
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(X + (long)b * 257 * D), 0, 256 * D * 2, 0x00020000);
        wsoff = h * 64 * D * 2;
        issue_pieces(0, 0, NP);
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            if (i == FI - 1) asm volatile("s_nop 7" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
            else asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BAR();
        load_frags(0, 0, 0, 0, NF);
        issue_pieces(1, 0, P3);
        __builtin_amdgcn_sched_barrier(0);
        int kt = 0;
        for (; kt < nk - 2; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
        k_tile(kt++, std::true_type{}, std::false_type{});
        k_tile(kt, std::false_type{}, std::false_type{});
        BAR();
#pragma unroll
        for (int i = 0; i < FI; ++i) {
            if (i == 0) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
            else asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                f32x4 v;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(acc[i][j][0]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(acc[i][j][1]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(acc[i][j][2]));
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(acc[i][j][3]));
                sum += v[0] + v[1] + v[2] + v[3];
            }
        if (sum == 12345.678f) sink[tid] = sum;
    }
}

template <int NW>
static void run(const f16* x, const f16* w, float* sink, int B, int H, int D) {
    auto kern = kloop_kernel<NW>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), 2 * STAGE, 0, x, w, sink, B, H, D);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 2.0 * B * H * 256.0 * 192.0 * D;
        printf("NW=%d launch %d: %.1f us  (%.0f TFLOP/s; bare MFMA time of the K loops at 2.4 GHz: %.0f us)  err=%s\n", NW, r, ms * 1e3, fl / ms / 1e9, fl / 2.5e15 * 1e6, hipGetErrorString(hipGetLastError()));
    }
}

int main() {
    const int B = 1020, H = 16, D = 1024;
    const size_t nx = (size_t)B * 257 * D, nw = (size_t)3 * D * D;
    std::vector<_Float16> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = (_Float16)(((i * 2654435761u) >> 20 & 1023) / 512.0f - 1.0f);
    for (size_t i = 0; i < nw; ++i) hw[i] = (_Float16)((((i * 40503u) >> 7 & 1023) / 512.0f - 1.0f) * 0.03f);
    void *x, *w, *s;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&s, 4096);
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    run<4>((const f16*)x, (const f16*)w, (float*)s, B, H, D);
    run<8>((const f16*)x, (const f16*)w, (float*)s, B, H, D);
    run<4>((const f16*)x, (const f16*)w, (float*)s, B, H, D);
    run<8>((const f16*)x, (const f16*)w, (float*)s, B, H, D);
    return 0;
}
