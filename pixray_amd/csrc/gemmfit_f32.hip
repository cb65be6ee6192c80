// Fit kernels on fp32 OPERANDS (gemmfit_kernel.h FIT_EPI_F32): the exact mode's decoder products -- implicit 3x3 convolutions and
// row-major products on v_mfma_f32_16x16x4_f32, the tiles of the decoder family.  Why: the like-for-like `ref` precision of the
// bench (the reference's own mix: fp32 VQGAN decoder + fp16 CLIP tower, vqgan.py:124-140 / slip.py:175) spent 8 of its 12.4 ms in
// the 4-wave fp32 kernels (64 x 64 tiles at ~80 TFLOP/s, 128 x 128 at ~98, split-K reduce launches); with 16 x the MFMA time per
// operand byte the ring never starves, so the one-workgroup-per-CU tiles run at the matrix pipes' pace.
#include "gemmfit_kernel.h"

bool prx_gemmfit_launch_f32(const prx_gemm_dev::GemmArgs& a, int bm, int bn, dim3 grid, hipStream_t s, const bf16_t* zp) {
    if (bm == 256 && bn == 128) { launch_fit_f32<4, 2, 4, 4, 1>(a, grid, s, zp); return true; }
    if (bm == 128 && bn == 128) { launch_fit_f32<2, 4, 4, 2, 1>(a, grid, s, zp); return true; }
    if (bm == 128 && bn == 64) { launch_fit_f32<2, 2, 4, 2, 2>(a, grid, s, zp); return true; }
    if (bm == 64 && bn == 64) { launch_fit_f32<2, 2, 2, 2, 2>(a, grid, s, zp); return true; }
    if (bm == 32 && bn == 64) { launch_fit_f32<1, 2, 2, 2, 4>(a, grid, s, zp); return true; }
    if (bm == 16 && bn == 64) { launch_fit_f32<1, 2, 1, 2, 4>(a, grid, s, zp); return true; }
    if (bm == 16 && bn == 32) { launch_fit_f32<1, 1, 1, 2, 8>(a, grid, s, zp); return true; }
    return false;
}
