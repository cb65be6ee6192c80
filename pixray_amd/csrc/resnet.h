// CLIP ModifiedResNet visual tower runner (RN50x4 of BASELINE.json configs[2]); see resnet.hip.
#pragma once
#include "common.h"

struct PrxResNet;
struct GemmCtx;
GemmCtx* prx_resnet_gemm_ctx_impl(PrxResNet* r);
int prx_resnet_create_impl(PrxResNet** out, int res, int width, const int* layers, int heads, int out_dim, int max_n,
                           int precision, const float* const* w, int n_w, hipStream_t s);
void prx_resnet_destroy_impl(PrxResNet* r);
int prx_resnet_minmax_impl(PrxResNet* r, const float* cutouts, int n, float* mm, hipStream_t s);
int prx_resnet_forward_impl(PrxResNet* r, const float* cutouts, int n, const float* mm, float* embeds, hipStream_t s);
int prx_resnet_backward_a_impl(PrxResNet* r, const float* cutouts, const float* mm, const float* d_embeds, double* acc,
                               hipStream_t s);
int prx_resnet_backward_b_impl(PrxResNet* r, const float* cutouts, const float* mm, const double* acc, float* g_cutouts,
                               hipStream_t s);
