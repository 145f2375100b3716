// Sustained rate of bare MFMA streams on gfx950: v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16, one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/ubench/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(512) mfma_kernel(float* out, int iters, float seed) {
    f16x8 a, b;
    // pseudo-random operands (a bare stream of CONSTANT operands toggles few wires and under-reads the power)
    unsigned h = (threadIdx.x + blockIdx.x * 977u) * 2654435761u + (unsigned)seed;
    for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; a[i] = (_Float16)(((int)(h >> 16) % 2001 - 1000) * 0.001f); h = h * 1664525u + 1013904223u; b[i] = (_Float16)(((int)(h >> 16) % 2001 - 1000) * 0.001f); }
    if (KIND == 0) {
        f32x16 acc[8];
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 12345.678f) out[threadIdx.x] = s;
    } else {
        f32x4 acc[32];
        for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 32; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
}

static void smi() { if (system("rocm-smi --showclocks --showpower --csv 2>/dev/null | grep card | cut -d, -f6,10") != 0) {} }

template <int KIND>
static void run(const char* name, int threads) {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 20000;                     // 16 x 32x32x16 or 32 x 16x16x32 per iteration = 524288 flop per wave per iteration
    const int waves = threads / 64;
    const double flop = 256.0 * waves * iters * 524288.0;
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(mfma_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
            n += 10;
            hipDeviceSynchronize();
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(mfma_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
        printf("%-28s %d waves/SIMD: %7.0f TFLOP/s   ", name, waves / 4, flop * n / dt / 1e12);
        fflush(stdout);
        smi();                                     // sampled while 40 launches are queued
        hipDeviceSynchronize();
    }
    hipFree(out);
}

int main() {
    run<0>("v_mfma_f32_32x32x16_f16", 256);
    run<1>("v_mfma_f32_16x16x32_f16", 256);
    run<0>("v_mfma_f32_32x32x16_f16", 512);
    run<1>("v_mfma_f32_16x16x32_f16", 512);
    return 0;
}
