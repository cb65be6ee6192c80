// gemmrow_kernel.h instances: bf16 operands, fp32 residuals, 320 < K <= 640, 80-column slabs
#include "gemmrow_kernel.h"
bool prx_gemmrow_launch_b20(const prx_gemm_dev::GemmArgs& a, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    return prx_gemmrow_dev::launch_slab80_k640<bf16_t, 1>(a, ksteps, nslab, row_tiles, nchunks, grid, s);
}
