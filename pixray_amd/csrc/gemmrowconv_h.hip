// gemmrowconv_kernel.h instances: IEEE half operands
#include "gemmrowconv_kernel.h"
bool prx_gemmrowconv_launch_h(const prx_gemm_dev::GemmArgs& a, int row_tiles, int n_cu, hipStream_t s) {
    return prx_gemmrow_dev::launch_conv<half_t>(a, row_tiles, n_cu, s);
}
