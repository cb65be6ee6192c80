// gemmrow_kernel.h instances: bf16 operands, fp32 residuals, K <= 320
#include "gemmrow_kernel.h"
bool prx_gemmrow_launch_b10(const prx_gemm_dev::GemmArgs& a, int nt, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    return prx_gemmrow_dev::launch_slab<bf16_t, 1, 10>(a, nt, ksteps, nslab, row_tiles, nchunks, grid, s);
}
