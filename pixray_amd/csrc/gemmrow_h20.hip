// gemmrow_kernel.h instances: IEEE half operands, 16-bit residual streams, 320 < K <= 640, 80-column slabs
#include "gemmrow_kernel.h"
bool prx_gemmrow_launch_h20(const prx_gemm_dev::GemmArgs& a, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    return prx_gemmrow_dev::launch_slab80_k640<half_t, 2>(a, ksteps, nslab, row_tiles, nchunks, grid, s);
}
