// CLIP ViT multi-head self-attention (nn.MultiheadAttention inside
// clip.model.ResidualAttentionBlock [UPSTREAM openai/CLIP clip/model.py], call
// site slip.py:65), forward and activation-gradient backward, for sequences of
// T <= 64 tokens with head dim 64 (ViT-B/32: T = 50).  One wave owns one
// (image, head): Q/K/V tiles live in LDS, S = QK^T, P = softmax(S/8) and O = PV
// run on v_mfma_f32_32x32x16_bf16 with the softmax done in the MFMA C layout by
// 32-lane butterflies.  Backward recomputes P and produces dQ/dK/dV in one pass.
//
// The kernels live in attention_kernels.inc, instantiated for both 16-bit operand formats: bf16
// (v_mfma_f32_32x32x16_bf16) and IEEE half (v_mfma_f32_32x32x16_f16, PRX_PREC_F16 -- the arithmetic of the reference's
// fp16 CLIP towers, slip.py:175); the public entry points pick the instantiation from `h16`.
#include "attention.h"
#include <stdlib.h>

// PRX_MHA_TILES=1: route 64 < T <= 512 through the tile kernels as well (A/B measurements; keeps the T > 512 path tested)
static bool prx_mha_force_tiles() {
    static const bool v = [] { const char* e = getenv("PRX_MHA_TILES"); return e && e[0] == '1'; }();
    return v;
}

#define A16_NS att_bf16
#define a16_t bf16_t
#define a16x8 bf16x8
#define a16x4 bf16x4
#define A16_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#include "attention_kernels.inc"
#undef A16_NS
#undef a16_t
#undef a16x8
#undef a16x4
#undef A16_MFMA

#define A16_NS att_f16
#define a16_t half_t
#define a16x8 f16x8
#define a16x4 f16x4
#define A16_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#include "attention_kernels.inc"
#undef A16_NS
#undef a16_t
#undef a16x8
#undef a16x4
#undef A16_MFMA

#define H(p) reinterpret_cast<const half_t*>(p)
#define HM(p) reinterpret_cast<half_t*>(p)
int prx_mha_fwd(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s, int h16) {
    return h16 ? att_f16::mha_fwd(H(qkv), HM(out), N, T, C, heads, s) : att_bf16::mha_fwd(qkv, out, N, T, C, heads, s);
}
int prx_mha_bwd(const bf16_t* qkv, const bf16_t* dout, bf16_t* dqkv, int N, int T, int C, int heads, hipStream_t s, int h16) {
    return h16 ? att_f16::mha_bwd(H(qkv), H(dout), HM(dqkv), N, T, C, heads, s) : att_bf16::mha_bwd(qkv, dout, dqkv, N, T, C, heads, s);
}
int prx_mha_fwd_gen(const bf16_t* qkv, bf16_t* out, float* lse, int N, int T, int C, int heads, hipStream_t s, int h16) {
    return h16 ? att_f16::mha_fwd_gen(H(qkv), HM(out), lse, N, T, C, heads, s) : att_bf16::mha_fwd_gen(qkv, out, lse, N, T, C, heads, s);
}
int prx_mha_fwd_causal(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s, int h16) {
    return h16 ? att_f16::mha_fwd_causal(H(qkv), HM(out), N, T, C, heads, s) : att_bf16::mha_fwd_causal(qkv, out, N, T, C, heads, s);
}
int prx_mha_bwd_gen(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int N, int T,
                    int C, int heads, hipStream_t s, int h16) {
    return h16 ? att_f16::mha_bwd_gen(H(qkv), H(out), H(dout), lse, HM(dqkv), N, T, C, heads, s)
               : att_bf16::mha_bwd_gen(qkv, out, dout, lse, dqkv, N, T, C, heads, s);
}
