// Fit kernels with a compile-time epilogue (gemmfit_kernel.h FIT_EPI_*): decoder tiles 32 x 64, 16 x 64, 16 x 32 (implicit 3x3 convolutions and
// row-major products, IEEE-half operands), with and without the fused GroupNorm sums.
#include "gemmfit_kernel.h"

#define DEC_TILE(...)                                                                     \
    switch (epi) {                                                                        \
        FIT_SPEC_CASE(FIT_EPI_OUT16, __VA_ARGS__, FIT_EPI_OUT16, true)                    \
        FIT_SPEC_CASE(FIT_EPI_RES16, __VA_ARGS__, FIT_EPI_RES16, true)                    \
        FIT_SPEC_CASE(FIT_EPI_GN, __VA_ARGS__, FIT_EPI_GN, true)                          \
        FIT_SPEC_CASE(FIT_EPI_RES16_GN, __VA_ARGS__, FIT_EPI_RES16_GN, true)              \
        FIT_SPEC_CASE(FIT_EPI_GNB, __VA_ARGS__, FIT_EPI_GNB, true)                        \
        default: return false;                                                            \
    }

bool prx_gemmfit_launch_spec_dec_c(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp) {
    if (bm == 32 && bn == 64) { DEC_TILE(1, 2, 2, 2, 4) }
    if (bm == 16 && bn == 64) { DEC_TILE(1, 2, 1, 2, 4) }
    if (bm == 16 && bn == 32) { DEC_TILE(1, 1, 1, 2, 8) }
    return false;
}
