// Phase timeline of the T = 257 attention kernel (stamp_amd/csrc/attention_vit257.hip compiled with -DA7_TRACE): s_memtime of the 8 waves of
// workgroup 0 at 7 marks per item.  NOTE (since the in-wave software pipeline): a mark is a conditional store, i.e. a branch; the ones around the
// pipelined stages split its basic block and the compiler then sinks vector work across them -- this build runs ~2x slower than the product
// kernel and its chunk-phase numbers no longer describe it.  The timelines in profiles/r02_pmc_attn257_sq.txt were taken before that change.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA7_TRACE -Iinclude -Istamp_amd/csrc tools/ubench/attn257_trace.hip
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
    for (int r = 0; r < 3; ++r) amds::attention_vit257(q, o, B, H, AMDS_F16, 0);
    hipDeviceSynchronize();
    static unsigned long long t[32 * 8 * 8];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(amds::a7_trace), sizeof(t));
    const char* names[6] = {"merge of the previous odd query + descriptor", "chunk loop (4 x QK^T, softmax, PV)", "odd key + normalise + store", "odd query (MFMA partials)", "stage next item into LDS", "final barrier"};
    for (int it = 4; it < 12; ++it) {
        printf("item %2d: total %6llu cycles (wave 0)  |", it, t[((it + 1) * 8 + 0) * 8] - t[(it * 8 + 0) * 8]);
        for (int k = 0; k < 6; ++k) {
            unsigned long long mn = ~0ull, mx = 0;
            for (int w = 0; w < 8; ++w) { const unsigned long long d = t[(it * 8 + k + 1) * 8 + w] - t[(it * 8 + k) * 8 + w]; mn = d < mn ? d : mn; mx = d > mx ? d : mx; }
            printf(" %s %llu-%llu |", k == 0 ? "loads" : k == 1 ? "chunks" : k == 2 ? "oddkey+out" : k == 3 ? "oddquery" : k == 4 ? "stage" : "barrier", mn, mx);
        }
        printf("\n");
    }
    (void)names;
    return 0;
}
