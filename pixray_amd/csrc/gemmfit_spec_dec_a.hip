// Fit kernels with a compile-time epilogue (gemmfit_kernel.h FIT_EPI_*): decoder tiles 256 x 128, 128 x 128 (implicit 3x3 convolutions and
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

bool prx_gemmfit_launch_spec_dec_a(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp) {
    if (bm == 256 && bn == 128) { DEC_TILE(4, 2, 4, 4, 1) }
    if (bm == 128 && bn == 128) { DEC_TILE(2, 4, 4, 2, 1) }
    return false;
}
