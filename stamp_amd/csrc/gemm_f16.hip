// f16-operand instantiations of the MFMA GEMM (see gemm_kernel.h)
#include "gemm_8p.h"
#include "gemm_4w.h"
#include "gemm_8p64.h"
#include "gemm_pp.h"
#include "gemm_4w64.h"
#include "gemm_4w16.h"
namespace amds {
AMDS_GEMM_DISPATCH_IMPL(f16)
}

// log buffer for the timeline bit (16) of the 2000-range ablation ids: set it, then call amds_gemm_ablate(2016, ...) with a real bias
static void* g_gemm_debug_log = nullptr;
extern "C" void amds_gemm_debug_log(void* p) { g_gemm_debug_log = p; }

// performance-archaeology entry (f16, EPI_BIAS only): ablated variants of the 8-phase kernel; results are wrong by design
extern "C" int amds_gemm_ablate(int abl, const void* A, long lda, const void* W, long ldw, int M, int N, int K,
                                void* out, long ldo, const float* bias, void* stream) {
    using namespace amds;
    EpiArgs ep; ep.out = out; ep.ldo = ldo; ep.bias = bias; ep.scale = nullptr; const int bits = abl >= 3000 ? 0 : (abl >= 2000 ? abl - 2000 : (abl >= 1000 ? abl - 1000 : abl)); ep.pos = (bits & 16) ? bias : nullptr; if (bits & 16) ep.bias = nullptr; if (abl >= 2000 && (bits & 16)) { if (!g_gemm_debug_log) return AMDS_ERR_INVALID; ep.pos = (const float*)g_gemm_debug_log; ep.bias = bias; } ep.np = ep.T = ep.P = 0; ep.acc_scale = 1.f;
    hipStream_t st = (hipStream_t)stream;
    switch (abl) {
#define C_(x) case x: return launch_gemm_8p_abl<f16, AMDS_EPI_BIAS, x>(A, lda, W, ldw, M, N, K, ep, st);
        C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(12) C_(14) C_(15) C_(16) C_(17) C_(18) C_(32) C_(48) C_(64) C_(80) C_(128) C_(192) C_(208)
#undef C_
    }
    switch (abl - 1000) {     // 1000 + bits: the 4-wave kernel
#define D_(x) case x: return launch_gemm_4w_abl<f16, AMDS_EPI_BIAS, x>(A, lda, W, ldw, M, N, K, ep, st);
        D_(0) D_(1) D_(4) D_(5) D_(8) D_(9) D_(12) D_(13)
#undef D_
    }
    switch (abl - 2000) {     // 2000 + bits: the 4-wave / 128-byte-row kernel
#define E_(x) case x: return launch_gemm_4w64_abl<f16, AMDS_EPI_BIAS, x>(A, lda, W, ldw, M, N, K, ep, st);
        E_(0) E_(1) E_(4) E_(5) E_(8) E_(9) E_(12) E_(13) E_(16)
#undef E_
    }
    return AMDS_ERR_INVALID;
}
