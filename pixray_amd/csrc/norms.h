#pragma once
#include "common.h"
#include <algorithm>
// `bf16_t*` outputs are 16-bit operand buffers: bf16, or IEEE half when h16 != 0 (common.h)
// GroupNorm(32 groups) on NHWC fp32 x[NB][P][C]; stats = double[NB][32][2] workspace (sum, sumsq).
// s16: the stream inputs (x, g, add) are 16-bit tensors in the operand format instead of fp32 (the lean layout, common.h)
int prx_groupnorm_fwd(const void* x, const float* gamma, const float* beta, double* stats, bf16_t* out_bf16,
                      float* out_f32, int NB, int P, int C, int swish, float eps, hipStream_t s, int zero_stats = 1,
                      int stats_ready = 0, int h16 = 0, int s16 = 0);
int prx_groupnorm_bwd(const void* g, const void* x, const float* gamma, const float* beta, const double* fstats,
                      double* bstats, const void* add, float* dx, bf16_t* dx_bf16, int NB, int P, int C, int swish,
                      float eps, hipStream_t s, int zero_stats = 1, int stats_ready = 0, int h16 = 0, int s16 = 0);
// the backward's statistics pass alone (sum dxhat, sum dxhat * xhat per group, accumulated into bstats, which the caller zeroed):
// what a producing GEMM's epilogue does when it can (gemm.h gnb_*); fp32 streams
int prx_groupnorm_bwd_stats(const float* g, const float* x, const float* gamma, const float* beta, const double* fstats,
                            double* bstats, int NB, int P, int C, int swish, float eps, hipStream_t s);
// LayerNorm on rows of width C.
// s16 (forward): x is a 16-bit stream.  s16 (backward), bits: 1 = x, 2 = g, 4 = add are 16-bit streams; dx (fp32) may be null
// when only the 16-bit output is wanted.
int prx_layernorm_fwd(const void* x, long long ldx, const float* gamma, const float* beta, bf16_t* out_bf16,
                      float* out_f32, float* mean, float* rstd, int rows, int C, float eps, hipStream_t s, int h16 = 0, int s16 = 0);
int prx_layernorm_bwd(const void* g, long long ldg, const void* x, long long ldx, const float* gamma,
                      const float* mean, const float* rstd, const void* add, long long ldadd, float* dx,
                      long long lddx, bf16_t* dx_bf16, long long lddxb, int rows, int C, hipStream_t s, int h16 = 0,
                      int add_every = 0, int s16 = 0);   // add_every > 0: `add` is read on the rows that are multiples of it only (it is taken as zero elsewhere)
