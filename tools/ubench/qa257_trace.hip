// Phase timeline of the fused qkv + attention kernel (stamp_amd/csrc/qkv_attn257.hip compiled with -DQA_TRACE): s_memtime of the four waves of workgroup 0
// at the phase boundaries of its first items.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQA_TRACE -Iinclude -Istamp_amd/csrc tools/ubench/qa257_trace.hip -o tools/ubench/qa257_trace
#include "../../stamp_amd/csrc/qkv_attn257.hip"
#include <cstdio>
#include <vector>
namespace amds { thread_local char g_err[512]; void set_error(const char*, ...) {} int hip_fail(hipError_t, const char*) { return -2; } std::atomic<int> g_prof_any{0};
int prof_begin(int, double, hipStream_t, amds_ctx**) { return -1; } void prof_end(amds_ctx*, int, hipStream_t) {} }
int main() {
    const int B = 1020, H = 16, T = 257, D = 1024;
    const size_t nx = (size_t)B * T * D, nw = (size_t)3 * D * D, nq = (size_t)B * T * 3 * D;
    std::vector<_Float16> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = (_Float16)(((i * 2654435761u) >> 20 & 1023) / 512.0f - 1.0f);
    for (size_t i = 0; i < nw; ++i) hw[i] = (_Float16)((((i * 40503u) >> 7 & 1023) / 512.0f - 1.0f) * 0.03f);
    std::vector<float> hb(3 * D, 0.1f), hrs((size_t)B * T * 2);
    for (size_t i = 0; i < hrs.size(); i += 2) { hrs[i] = 1.0f; hrs[i + 1] = 0.0f; }
    void *x, *w, *b, *cs, *rs, *qt, *o;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&b, 3 * D * 4); hipMalloc(&cs, 3 * D * 4); hipMalloc(&rs, hrs.size() * 4); hipMalloc(&qt, nq * 2); hipMalloc(&o, nx * 2);
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 3 * D * 4, hipMemcpyHostToDevice); hipMemcpy(cs, hb.data(), 3 * D * 4, hipMemcpyHostToDevice);
    hipMemcpy(rs, hrs.data(), hrs.size() * 4, hipMemcpyHostToDevice); hipMemset(qt, 0, nq * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        amds_qkv_attention_vit257(x, w, (const float*)b, (const float*)cs, (const float*)rs, qt, o, B, H, D, AMDS_F16, 0);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); printf("launch %d: %.1f us (traced build)\n", r, ms * 1e3);
    }
    static unsigned long long t[32 * 12 * 4];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(amds::qa_trace), sizeof(t));
    const char* names[6] = {"prologue (first K tile lands)", "K loop", "hand-off (acc -> images)", "barrier", "S phase", "barrier"};
    for (int it = 2; it < 12; ++it) {
        printf("item %2d: total %6llu ticks (wave 0) |", it, t[((it + 1) * 12 + 0) * 4] - t[(it * 12 + 0) * 4]);
        for (int k = 0; k < 6; ++k) {
            unsigned long long mn = ~0ull, mx = 0;
            for (int wv = 0; wv < 4; ++wv) { const unsigned long long d = t[(it * 12 + k + 1) * 4 + wv] - t[(it * 12 + k) * 4 + wv]; mn = d < mn ? d : mn; mx = d > mx ? d : mx; }
            printf(" %s %llu-%llu |", names[k], mn, mx);
        }
        {   // inside the S phase: marks 4 -> 7 (pass 1: maxima) -> 8 (pass 2: exp + P V) -> 9 (normalise + store) -> 5 (odd query)
            const int seq[5] = {4, 7, 8, 9, 5};
            const char* sn[4] = {"S.pass1", "S.pass2", "S.store", "S.oddq"};
            for (int k = 0; k < 4; ++k) {
                unsigned long long mn = ~0ull, mx = 0;
                for (int wv = 0; wv < 4; ++wv) { const unsigned long long d = t[(it * 12 + seq[k + 1]) * 4 + wv] - t[(it * 12 + seq[k]) * 4 + wv]; mn = d < mn ? d : mn; mx = d > mx ? d : mx; }
                printf(" %s %llu-%llu |", sn[k], mn, mx);
            }
        }
        printf("\n");
    }
    return 0;
}
