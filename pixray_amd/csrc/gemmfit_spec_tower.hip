// Fit kernels with a compile-time epilogue (gemmfit_kernel.h FIT_EPI_*): the token-batch tiles of the CLIP tower (row-major
// products, 80-row granular, IEEE-half operands).
#include "gemmfit_kernel.h"

#define TOWER_TILE(...)                                                                   \
    switch (epi) {                                                                        \
        FIT_SPEC_CASE(FIT_EPI_OUT16, __VA_ARGS__, FIT_EPI_OUT16, false)                   \
        FIT_SPEC_CASE(FIT_EPI_RES16, __VA_ARGS__, FIT_EPI_RES16, false)                   \
        FIT_SPEC_CASE(FIT_EPI_GELU, __VA_ARGS__, FIT_EPI_GELU, false)                     \
        FIT_SPEC_CASE(FIT_EPI_DGELU, __VA_ARGS__, FIT_EPI_DGELU, false)                   \
        default: return false;                                                            \
    }

bool prx_gemmfit_launch_spec_tower(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp) {
    if (a.d.a_mode != PRX_A_ROWMAJOR) return false;
    if (bm == 160 && bn == 256) { TOWER_TILE(2, 4, 5, 4, 1) }
    if (bm == 160 && bn == 192) { TOWER_TILE(2, 4, 5, 3, 1) }
    if (bm == 160 && bn == 128) { TOWER_TILE(2, 4, 5, 2, 1) }
    if (bm == 80 && bn == 128) { TOWER_TILE(1, 4, 5, 2, 2) }
    return false;
}
