#pragma once
#include "common.h"
// `f32`: the untyped buffers hold fp32 (else 16-bit); `prec`: PRX_PREC_* where the 16-bit FORMAT matters; `bf16_t*` + `h16`:
// a 16-bit operand buffer holding bf16 (h16 == 0) or IEEE half
int prx_transpose_op(const void* in, int ldin, void* out, int ldout, int R, int C, int f32, hipStream_t s);
int prx_softmax_rows(const float* S, int lds_, float scale, void* P, int ldp, void* PT, int ldpt, int rows,
                     int cols, int prec, hipStream_t s);
int prx_softmax_rows_bwd(const void* P, int ldp, const float* dP, int lddp, float scale, void* dS, int ldds,
                         void* dST, int lddst, int rows, int cols, int prec, hipStream_t s);
// s16: `hi` is a 16-bit stream in the operand format (lean layout); `low` (fp32) may then be null
int prx_upsample2x_bwd(const void* hi, float* low, bf16_t* low_bf16, int NB, int Hl, int Wl, int C, hipStream_t s, int h16 = 0, int s16 = 0);
int prx_nchw_to_nhwc(const float* in, float* out_f32, bf16_t* out_bf16, int NB, int C, int HW, int Cpad, hipStream_t s, int h16 = 0);
int prx_nhwc_to_nchw(const float* in, int ldc, float* out, int NB, int C, int HW, hipStream_t s);
int prx_image_head_fwd(const float* x, int ldc, float* img, int NB, int C, int HW, hipStream_t s);
// gscale (device scalar or null): multiplies the outgoing gradient -- the runner's power-of-two gradient scale in the half
// mode; the ClampWithGrad sign test is unaffected
int prx_image_head_bwd(const float* x, int ldc, const float* gimg, float* dx, bf16_t* dx_bf16, int ldo, int NB, int C,
                       int HW, hipStream_t s, int h16 = 0, const float* gscale = nullptr);
// ... as the im2col matrix [H*W][ldk] of the 3x3 convolution that follows (tap-major, 8 channels per tap, zero padded to ldk): batch 1
int prx_image_head_bwd_im2col(const float* x, int ldc, const float* gimg, bf16_t* col, int ldk, int C, int H, int W, hipStream_t s, int h16 = 0,
                              const float* gscale = nullptr);
// Power-of-two gradient scale of the half (PRX_PREC_F16) mode, chosen on the device from the gradient that enters a runner's
// backward: scale2 = {S, 1/S} with S * max|g| in [2^(T-1), 2^T), T = target_log2 (S = 1 for all-zero / non-finite g).
// part: nparts floats of scratch.  Two tiny launches, no host synchronisation.
int prx_grad_scale(const float* g, size_t n, float* part, int nparts, int target_log2, float* scale2, hipStream_t s);
int prx_adam_clamp(float* z, float* m, float* v, const float* g, const float* zmin, const float* zmax, int hw,
                   size_t n, float lr, float b1, float b2, float eps, int step, hipStream_t s);
int prx_adam_clamp_dev(float* z, float* m, float* v, const float* g, const float* zmin, const float* zmax, int hw,
                       size_t n, const float* hyper, float b1, float b2, float eps, hipStream_t s);
int prx_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s, int h16 = 0);
// the same over several tensors (one common S): part holds count * nparts_each floats
int prx_grad_scale_multi(const float* const* gs, const size_t* ns, int count, float* part, int nparts_each, int target_log2,
                         float* scale2, hipStream_t s);
int prx_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t s);
// x[i] *= *scale with a DEVICE scalar: applies the half mode's power-of-two gradient scale to an fp32 gradient BEFORE it is
// rounded to half (scaling in a GEMM epilogue would come after the operand conversion and lose the small entries to subnormals)
// out16 (optional): the scaled values also as a 16-bit operand (format h16)
int prx_scale_dev(float* x, size_t n, const float* scale, hipStream_t s, bf16_t* out16 = nullptr, int h16 = 0);
