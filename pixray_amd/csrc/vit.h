#pragma once
#include "common.h"
struct PrxVit;
struct GemmCtx;
int prx_pack_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
int prx_pack_transpose_bf16(const float* in, bf16_t* out, int R, int C, hipStream_t s);
// weight packs at operand precision `prec` (PRX_PREC_*; f32: plain copy / fp32 transpose)
int prx_pack_op(const float* in, void* out, size_t n, int prec, hipStream_t s);
int prx_pack_transpose_op(const float* in, void* out, int R, int C, int prec, hipStream_t s);
int prx_vit_create_impl(PrxVit** out, int res, int patch, int width, int layers, int heads, int out_dim, int max_n,
                        int precision, const float* const* w, int n_w, hipStream_t s);
GemmCtx* prx_vit_gemm_ctx_impl(PrxVit* v);
void prx_vit_destroy_impl(PrxVit* v);
int prx_vit_minmax_impl(PrxVit* v, const float* cutouts, int n, float* mm, hipStream_t s);
int prx_vit_forward_impl(PrxVit* v, const float* cutouts, int n, const float* mm, float* embeds, hipStream_t s);
int prx_vit_backward_a_impl(PrxVit* v, const float* cutouts, const float* mm, const float* d_embeds, double* acc,
                            hipStream_t s);
int prx_vit_backward_b_impl(PrxVit* v, const float* cutouts, const float* mm, const double* acc, float* g_cutouts,
                            hipStream_t s);
