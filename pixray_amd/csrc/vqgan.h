#pragma once
#include "common.h"
struct PrxVqgan;
struct GemmCtx;
GemmCtx* prx_vqgan_gemm_ctx_impl(PrxVqgan* v);
int prx_vqgan_create_impl(PrxVqgan** out, int ch, const int* ch_mult, int n_mult, int num_res_blocks, int attn_res,
                          int resolution, int z_channels, int embed_dim, int n_embed, int out_ch, int h0, int w0,
                          int precision, const float* const* w, int n_w, hipStream_t s);
void prx_vqgan_destroy_impl(PrxVqgan* v);
int prx_vqgan_bounds_impl(PrxVqgan* v, float* zmin, float* zmax, hipStream_t s);
int prx_vqgan_synth_impl(PrxVqgan* v, const float* z, float* img, int* indices, int quantize, hipStream_t s);
int prx_vqgan_backward_impl(PrxVqgan* v, const float* g_img, float* dz, hipStream_t s);
long long prx_vqgan_debug_stage_impl(PrxVqgan* v, int stage, float* dst, long long max_floats, hipStream_t s);
