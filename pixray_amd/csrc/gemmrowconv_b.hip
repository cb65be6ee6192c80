// gemmrowconv_kernel.h instances: bf16 operands
#include "gemmrowconv_kernel.h"
bool prx_gemmrowconv_launch_b(const prx_gemm_dev::GemmArgs& a, int row_tiles, int n_cu, hipStream_t s) {
    return prx_gemmrow_dev::launch_conv<bf16_t>(a, row_tiles, n_cu, s);
}
