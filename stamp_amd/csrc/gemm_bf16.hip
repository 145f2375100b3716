// bf16-operand instantiations of the MFMA GEMM (see gemm_kernel.h)
#include "gemm_kernel.h"
namespace amds {
AMDS_GEMM_DISPATCH_IMPL(bf16)
}
