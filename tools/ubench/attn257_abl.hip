// Ablations of the T = 257 attention kernel (attention_vit257.hip compiled with -DA7_ABL=<bits>): 1 = no v_exp_f32 in the chunk softmax (an FMA result stands in),
// 2 = no K / V staging loads of the next item, 4 = both.  Prints us per 1020 x 16 launch.  Results are wrong by construction: this is a timing probe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA7_ABL=1 -Iinclude -o attn257_abl1 tools/ubench/attn257_abl.hip
#include "../../stamp_amd/csrc/attention_vit257.hip"
#include <cstdio>
#include <vector>
namespace amds { thread_local char g_err[512]; void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -2; } }
int main() {
    const int B = 1020, H = 16, T = 257;
    const size_t n = (size_t)B * T * 3 * H * 64;
    std::vector<_Float16> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (_Float16)(((i * 2654435761u) >> 20 & 1023) / 512.0f - 1.0f);
    void *q, *o;
    hipMalloc(&q, n * 2); hipMalloc(&o, (size_t)B * T * H * 64 * 2);
    hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
    for (int r = 0; r < 20; ++r) amds::attention_vit257(q, o, B, H, AMDS_F16, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    for (int r = 0; r < 50; ++r) amds::attention_vit257(q, o, B, H, AMDS_F16, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
#ifndef A7_ABL
#define A7_ABL 0
#endif
    printf("A7_ABL=%d: %.1f us per launch\n", A7_ABL, ms / 50 * 1e3);
    return 0;
}
