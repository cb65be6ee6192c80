// Fit kernels with a compile-time epilogue (gemmfit_kernel.h FIT_EPI_*): decoder tiles 128 x 64, 64 x 64 (implicit 3x3 convolutions and
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

bool prx_gemmfit_launch_spec_dec_b(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp) {
    if (bm == 128 && bn == 64) { DEC_TILE(2, 2, 4, 2, 2) }
    if (bm == 64 && bn == 64) { DEC_TILE(2, 2, 2, 2, 2) }
    return false;
}
