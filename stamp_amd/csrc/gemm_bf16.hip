// bf16-operand instantiations of the MFMA GEMM (see gemm_kernel.h)
#include "gemm_8p64.h"
#include "gemm_4w64.h"
#include "gemm_4w16.h"
namespace amds {
AMDS_GEMM_DISPATCH_IMPL(bf16)
}
