// C-ABI kernel-level entry points (tests/ check every kernel alone through these).
#include "common.h"
#include "norms.h"
#include "elementwise.h"
#include "attention.h"
#include "../../include/prx.h"

#define S_(x) ((hipStream_t)(x))
#define B_(x) ((bf16_t*)(x))
#define CB_(x) ((const bf16_t*)(x))

extern "C" {

int prx_k_groupnorm_fwd(const float* x, const float* gamma, const float* beta, double* stats, void* out_bf16,
                        float* out_f32, int NB, int P, int C, int swish, float eps, prx_stream_t s) {
    return prx_groupnorm_fwd(x, gamma, beta, stats, B_(out_bf16), out_f32, NB, P, C, swish, eps, S_(s));
}
int prx_k_groupnorm_bwd(const float* g, const float* x, const float* gamma, const float* beta, const double* fstats,
                        double* bstats, const float* add, float* dx, int NB, int P, int C, int swish, float eps,
                        prx_stream_t s) {
    return prx_groupnorm_bwd(g, x, gamma, beta, fstats, bstats, add, dx, nullptr, NB, P, C, swish, eps, S_(s));
}
int prx_k_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, void* out_bf16,
                        float* out_f32, float* mean, float* rstd, int rows, int C, float eps, prx_stream_t s) {
    return prx_layernorm_fwd(x, ldx, gamma, beta, B_(out_bf16), out_f32, mean, rstd, rows, C, eps, S_(s));
}
int prx_k_layernorm_bwd(const float* g, long long ldg, const float* x, long long ldx, const float* gamma,
                        const float* mean, const float* rstd, const float* add, long long ldadd, float* dx,
                        long long lddx, int rows, int C, prx_stream_t s) {
    return prx_layernorm_bwd(g, ldg, x, ldx, gamma, mean, rstd, add, ldadd, dx, lddx, nullptr, 0, rows, C, S_(s));
}
int prx_k_transpose_bf16(const void* in, int ldin, void* out, int ldout, int R, int C, prx_stream_t s) {
    return prx_transpose_op(in, ldin, out, ldout, R, C, PRX_PREC_BF16, S_(s));
}
int prx_k_softmax_rows(const float* S, int lds_, float scale, void* P, int ldp, void* PT, int ldpt, int rows, int cols,
                       prx_stream_t s) {
    return prx_softmax_rows(S, lds_, scale, P, ldp, PT, ldpt, rows, cols, PRX_PREC_BF16, S_(s));
}
int prx_k_softmax_rows_bwd(const void* P, int ldp, const float* dP, int lddp, float scale, void* dS, int ldds,
                           void* dST, int lddst, int rows, int cols, prx_stream_t s) {
    return prx_softmax_rows_bwd(P, ldp, dP, lddp, scale, dS, ldds, dST, lddst, rows, cols, PRX_PREC_BF16, S_(s));
}
int prx_k_upsample2x_bwd(const float* hi, float* low, int NB, int Hl, int Wl, int C, prx_stream_t s) {
    return prx_upsample2x_bwd(hi, low, nullptr, NB, Hl, Wl, C, S_(s));
}
int prx_k_nchw_to_nhwc(const float* in, float* out_f32, void* out_bf16, int NB, int C, int HW, int Cpad,
                       prx_stream_t s) {
    return prx_nchw_to_nhwc(in, out_f32, B_(out_bf16), NB, C, HW, Cpad, S_(s));
}
int prx_k_nhwc_to_nchw(const float* in, int ldc, float* out, int NB, int C, int HW, prx_stream_t s) {
    return prx_nhwc_to_nchw(in, ldc, out, NB, C, HW, S_(s));
}
int prx_k_image_head_fwd(const float* x, int ldc, float* img, int NB, int C, int HW, prx_stream_t s) {
    return prx_image_head_fwd(x, ldc, img, NB, C, HW, S_(s));
}
int prx_k_image_head_bwd(const float* x, int ldc, const float* gimg, float* dx, void* dx_bf16, int ldo, int NB, int C,
                         int HW, prx_stream_t s) {
    return prx_image_head_bwd(x, ldc, gimg, dx, B_(dx_bf16), ldo, NB, C, HW, S_(s));
}
int prx_k_mha_fwd(const void* qkv, void* out, int N, int T, int C, int heads, prx_stream_t s) {
    return prx_mha_fwd(CB_(qkv), B_(out), N, T, C, heads, S_(s));
}
int prx_k_mha_bwd(const void* qkv, const void* dout, void* dqkv, int N, int T, int C, int heads, prx_stream_t s) {
    return prx_mha_bwd(CB_(qkv), CB_(dout), B_(dqkv), N, T, C, heads, S_(s));
}

int prx_k_mha_fwd_gen(const void* qkv, void* out, float* lse, int N, int T, int C, int heads, prx_stream_t s) {
    return prx_mha_fwd_gen(CB_(qkv), B_(out), lse, N, T, C, heads, S_(s));
}
int prx_k_mha_bwd_gen(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int N, int T, int C,
                      int heads, prx_stream_t s) {
    return prx_mha_bwd_gen(CB_(qkv), CB_(out), CB_(dout), lse, B_(dqkv), N, T, C, heads, S_(s));
}

int prx_k_mha_fwd_f32(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, prx_stream_t s) {
    return prx_mha_fwd_f32(qkv, out, lse, N, T, C, heads, S_(s));
}
int prx_k_mha_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int N, int T, int C,
                      int heads, prx_stream_t s) {
    return prx_mha_bwd_f32(qkv, out, dout, lse, dqkv, N, T, C, heads, S_(s));
}

}  // extern "C"
