// VqganDrawer.init_from_tensor / reapply_from_tensor / get_z_from_tensor runner (vqgan.py:174-185):
// taming VQModel.encode = Encoder -> quant_conv -> VectorQuantizer (nearest code).  Forward only.
#pragma once
#include "common.h"

struct PrxVqganEnc;
int prx_vqgan_enc_create_impl(PrxVqganEnc** out, int ch, const int* ch_mult, int n_mult, int num_res_blocks, int attn_res,
                              int resolution, int in_ch, int z_channels, int embed_dim, int n_embed, int H, int W,
                              const float* const* w, int n_w, hipStream_t s);
void prx_vqgan_enc_destroy_impl(PrxVqganEnc* e);
int prx_vqgan_encode_impl(PrxVqganEnc* e, const float* img, float* z, float* z_pre, int* indices, hipStream_t s);
