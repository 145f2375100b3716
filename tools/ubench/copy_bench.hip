// Micro-benchmark: per-CU throughput of global->LDS copies (global_load_lds_dwordx4) and plain 16-B global loads
// for the access shapes a GEMM tile loader produces.  One 512-thread block per CU, each wave keeps `depth` pieces in
// flight (counted vmcnt).  Build: hipcc --offload-arch=gfx950 -O3 copy_bench.hip -o copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// pattern: rows of `row_bytes` contiguous bytes (64, 128, 1024), row pitch `pitch` bytes; a wave instruction covers
// 1024/row_bytes rows.  Each block walks `iters` K-steps over its own 256-row panel (like a GEMM A tile).
template <int ROWB, int MODE>   // MODE 0 = LDS-DMA, 1 = global_load to VGPR (+ dummy use)
__global__ void __launch_bounds__(512) copy_kernel(const char* __restrict__ base, long pitch, int iters, int panels, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int panel = blockIdx.x % panels;
    constexpr int LPR = ROWB / 16;              // lanes per row
    // block tile: 512 rows x ROWB? keep bytes per step per block = 32 KB: rows = 32768/ROWB
    constexpr int ROWS = 32768 / ROWB;
    constexpr int PIECES = 32768 / (512 * 16); // = 4 per thread
    const char* src[PIECES];
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        const int c = it * 512 + tid, row = c / LPR, cp = c % LPR;
        src[it] = base + ((long)panel * ROWS + row) * pitch + cp * 16;
    }
    float acc = 0.f;
    for (int k = 0; k < iters; ++k) {
        char* st = smem + (k & 3) * 32768;
#pragma unroll
        for (int it = 0; it < PIECES; ++it) {
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((gptr_t)(src[it] + (long)k * ROWB), (lptr_t)(st + (it * 512 + wave * 64) * 16), 16, 0, 0);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(src[it] + (long)k * ROWB);
                acc += v.x;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) acc = smem[(tid * 16) & 32767];
    if (acc == 12345.678f) sink[0] = acc;
}

template <int ROWB, int MODE>
double run(const char* buf, long pitch, int iters, int panels, int blocks, float* sink) {
    hipFuncSetAttribute((const void*)copy_kernel<ROWB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    copy_kernel<ROWB, MODE><<<blocks, 512, 131072>>>(buf, pitch, iters, panels, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) copy_kernel<ROWB, MODE><<<blocks, 512, 131072>>>(buf, pitch, iters, panels, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)blocks * iters * 32768.0 * 5 / (ms * 1e-3) / 1e12;
}

int main() {
    const long total = 1L << 30;   // 1 GiB
    char* buf; hipMalloc(&buf, total + (1 << 20)); hipMemset(buf, 1, total);
    float* sink; hipMalloc(&sink, 4);
    const int blocks = 256;
    printf("TB/s aggregate (256 blocks x 512 thr, 32 KB per step per block)\n");
    // distinct panel per block (no sharing) vs 4 blocks sharing a panel (L2 reuse like N-tiles sharing an A panel)
    for (int share = 1; share <= 4; share *= 4) {
        const int panels = blocks / share;
        // pitch 8192 (K=4096 f16), 2048 (K=1024 f16)
        for (long pitch : {8192L, 2048L}) {
            int iters64 = (int)(pitch / 64), iters128 = (int)(pitch / 128);
            printf("share=%d pitch=%ld  64B rows: dma %.2f  vgpr %.2f | 128B rows: dma %.2f vgpr %.2f\n", share, pitch,
                   run<64, 0>(buf, pitch, iters64, panels, blocks, sink), run<64, 1>(buf, pitch, iters64, panels, blocks, sink),
                   run<128, 0>(buf, pitch, iters128, panels, blocks, sink), run<128, 1>(buf, pitch, iters128, panels, blocks, sink));
        }
        printf("share=%d contiguous 1KB rows (pitch 1<<20 / panel walk): dma %.2f vgpr %.2f\n", share,
               run<1024, 0>(buf, 1 << 15, 32, panels, blocks, sink), run<1024, 1>(buf, 1 << 15, 32, panels, blocks, sink));
    }
    return 0;
}
