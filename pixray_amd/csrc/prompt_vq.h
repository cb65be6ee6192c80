#pragma once
#include "common.h"
int prx_prompt_loss(const float* x, const float* embed, int n, int m, int D, float weight, float stop, float denom,
                    float* rowloss, float* grad, float* loss, unsigned* ticket, hipStream_t s);
int prx_l2norm_fwd(const float* e, float* out, int n, int D, hipStream_t s);
int prx_l2norm_bwd(const float* e, const float* g, float* de, int n, int D, hipStream_t s);
int prx_sqnorm_rows(const float* w, float* out, int rows, int D, hipStream_t s);
// z tokens: x[p][k] = z[k*ch_stride + p*tok_stride]; pmin/pidx: [P][ceil(NC/64)] scratch
int prx_vq_nearest(const float* z, long long tok_stride, long long ch_stride, const float* codebook, const float* cnorm,
                   int P, int NC, int D, float* pmin, int* pidx, int* idx_out, float* zq, hipStream_t s);
