// VGG16 feature extractor of the StyleLoss plugin (reference Losses/StyleLoss.py:24-47) on the MFMA
// implicit-GEMM engine.  See vgg.hip.
#pragma once
#include "common.h"

struct PrxVgg16;
struct GemmCtx;
GemmCtx* prx_vgg16_gemm_ctx_impl(PrxVgg16* v);
int prx_vgg16_create_impl(PrxVgg16** out, const float* const* weights, int n_weights, int max_h, int max_w, int precision, hipStream_t s);
void prx_vgg16_destroy_impl(PrxVgg16* v);
long long prx_vgg16_workspace_bytes_impl(int H, int W, int precision);
int prx_vgg16_feature_shape_impl(int H, int W, int k, int* h, int* w, int* c);
int prx_vgg16_forward_impl(PrxVgg16* v, const float* x, int H, int W, void* workspace, float* const* feats, hipStream_t s);
int prx_vgg16_backward_impl(PrxVgg16* v, int H, int W, const void* workspace, const float* const* g_feats, float* g_x, hipStream_t s);
