#pragma once
#include "common.h"
#define PRX_CUT_DESC_WORDS 36
int prx_pool_fwd(const float* img, float* pooled, int* argmax, const unsigned char* mask, int C, int H, int W, int S, hipStream_t s);
int prx_pool_bwd(const float* g, const int* argmax, const unsigned char* mask, float* gimg, int C, int H, int W, int S, hipStream_t s);
// stage A renders Ha x Wa images from the shared Hs x Ws source; stage B reads them through the descriptor's window
int prx_warp_a_fwd(const float* src, int Hs, int Ws, const double* desc, float* out, int n_cut, int Ha, int Wa, hipStream_t s);
int prx_warp_a_bwd(const float* g, int Hs, int Ws, const double* desc, float* uv, float* gsrc_priv, float* gsrc, int n_cut, int Ha,
                   int Wa, hipStream_t s);   // uv: [n_cut,Ha*Wa,2] scratch
int prx_warp_b_fwd(const float* a, int Ha, int Wa, const double* desc, const float* noise, float* out, int n_cut, int S,
                   hipStream_t s);
int prx_warp_b_bwd(const float* a, int Ha, int Wa, const double* desc, const float* g, float* grgb, float* uv, float* ga, int n_cut,
                   int S, hipStream_t s, float* maps_scratch, size_t maps_scratch_bytes);
// grgb: [n_cut,3,S,S], uv: [n_cut,S*S,2] scratch; maps_scratch: >= 64 bytes per cutout, untouched by anything else until the
// launch has finished (the per-cutout inverse stage maps of the tile-owned scatter)
// bilinear resize of the pooled [C,S,S] image to the canvas aspect [C,Hb,Wb] (pixray.py:468-472) and its gradient
int prx_rescale_fwd(const float* pooled, float* base, int C, int S, int Hb, int Wb, hipStream_t s);
int prx_rescale_bwd(const float* g_base, float* g_pooled, int C, int S, int Hb, int Wb, hipStream_t s);
int prx_minmax(const float* x, size_t n, float* part, int nparts, float* mm, hipStream_t s);
int prx_patchify_fwd(const float* cut, const float* mm, void* A, int prec, int N, int S, int P, int T, hipStream_t s);   // A at operand precision (PRX_PREC_*)
int prx_patchify_bwd_reduce(const float* cut, const float* mm, const float* dA, double* acc, int N, int S, int P, int T,
                            hipStream_t s);
int prx_patchify_bwd_apply(const float* cut, const float* mm, const float* dA, const double* acc, float* gcut, int N,
                           int S, int P, int T, hipStream_t s);
// gradient through slip.py:21-42 (batch-global min/max renorm + mean/std) for an image-layout gradient dY[N][3][S][S]
int prx_preproc_bwd_reduce(const float* cut, const float* mm, const float* dY, double* acc, int N, int S, hipStream_t s);
int prx_preproc_bwd_apply(const float* cut, const float* mm, const float* dY, const double* acc, float* gcut, int N, int S,
                          hipStream_t s);
