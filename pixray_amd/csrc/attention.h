#pragma once
#include "common.h"
// 16-bit operand buffers (`bf16_t*`): bf16, or IEEE half when h16 != 0
// qkv: bf16 [N*T, 3C] (q | k | v, head h at columns h*64..), out/dout: bf16 [N*T, C]
int prx_mha_fwd(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s, int h16 = 0);
int prx_mha_bwd(const bf16_t* qkv, const bf16_t* dout, bf16_t* dqkv, int N, int T, int C, int heads, hipStream_t s, int h16 = 0);
// any sequence length (64-token tiles, online softmax); out must be kept for the backward, lse: fp32 [N*heads*T]
int prx_mha_fwd_gen(const bf16_t* qkv, bf16_t* out, float* lse, int N, int T, int C, int heads, hipStream_t s, int h16 = 0);
int prx_mha_bwd_gen(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int N, int T,
                    int C, int heads, hipStream_t s, int h16 = 0);
// forward only, causal mask (CLIP text transformer, context 77)
int prx_mha_fwd_causal(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s, int h16 = 0);
// exact-f32 attention of the PRX_PREC_F32 parity mode (attention_f32.hip): any T, fp32 operands; `out` and `lse`
// ([N*heads*T]) are kept for the backward
int prx_mha_fwd_f32(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, hipStream_t s);
int prx_mha_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int N, int T, int C,
                    int heads, hipStream_t s);
