// CLIP_Base.encode_text runner (slip.py:68-70 -> openai/CLIP `CLIP.encode_text` [UPSTREAM clip/model.py]).  Forward only.
#pragma once
#include "common.h"

struct PrxClipText;
int prx_clip_text_create_impl(PrxClipText** out, int vocab, int ctx, int width, int layers, int heads, int out_dim, int max_n,
                              const float* const* w, int n_w, hipStream_t s);
void prx_clip_text_destroy_impl(PrxClipText* t);
int prx_clip_text_encode_impl(PrxClipText* t, const int* tokens, int n, float* embeds, hipStream_t s);
