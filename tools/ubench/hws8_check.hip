// Checks half_wave_sum8 (stamp_amd/csrc/common.h) against a host reference: hipcc --offload-arch=gfx950 -I include -I stamp_amd/csrc tools/ubench/hws8_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "common.h"
using namespace amds;
__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    f32x2 v[8];
    for (int u = 0; u < 8; ++u) v[u] = f32x2{in[(lane * 8 + u) * 2], in[(lane * 8 + u) * 2 + 1]};
    const f32x2 t = half_wave_sum8(v, l31);
    out[lane * 2] = t[0];
    out[lane * 2 + 1] = t[1];
}
int main() {
    float h[64 * 16], o[128];
    for (int i = 0; i < 64 * 16; ++i) h[i] = (float)((i * 37) % 101) * 0.25f - 7.f;
    float *d, *e;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int l31 = lane & 31, half = lane >> 5;
        const int u = ((l31 >> 4) & 1) * 4 + ((l31 >> 3) & 1) * 2 + ((l31 >> 2) & 1);
        for (int e2 = 0; e2 < 2; ++e2) {
            double s = 0;
            for (int l = 0; l < 32; ++l) s += h[((half * 32 + l) * 8 + u) * 2 + e2];
            if (fabs(s - o[lane * 2 + e2]) > 1e-3) { if (bad < 12) printf("lane %d e %d: got %f want %f\n", lane, e2, o[lane * 2 + e2], s); ++bad; }
        }
    }
    printf("hws8: %d mismatches\n", bad);
    return bad != 0;
}
