// bf16 MFMA GEMM / implicit-GEMM convolution engine (gfx950).
//   C[M,N] = epilogue( A[M,K] * Bt[N,K]^T )
// A is either a row-major matrix (bf16, or f32 converted on load) or the
// implicit im2col view of an NHWC activation for a 3x3/pad-1 convolution
// (optionally reading through a fused nearest-2x upsample).  Bt is always the
// bf16 weight pack, K contiguous.  Accumulation is fp32 on
// v_mfma_f32_32x32x16_bf16.
#pragma once
#include "common.h"

enum { PRX_ACT_NONE = 0, PRX_ACT_QUICKGELU = 1, PRX_ACT_MUL_DQUICKGELU = 2,
       PRX_ACT_RELU = 3,            // max(v, 0) after bias / residual (CLIP ModifiedResNet)
       PRX_ACT_MUL_RELUMASK = 4 };  // v *= (aux > 0): ReLU backward with the forward OUTPUT as aux
enum { PRX_A_ROWMAJOR = 0, PRX_A_CONV3X3 = 1 };

struct GemmDesc {
    // operands
    const void* A = nullptr;   // bf16 or f32 (a_is_f32)
    int a_is_f32 = 0;
    int a_mode = PRX_A_ROWMAJOR;
    int lda = 0;               // row stride (row-major) or pixel stride (conv), elements
    const bf16_t* B = nullptr; // [N, K], ldb
    int ldb = 0;
    int M = 0, N = 0, K = 0;
    // conv geometry: M = NB*H*W output pixels, K = 9*Cin; `up` reads an (H/2)x(W/2) input
    int H = 0, W = 0, Cin = 0, up = 0;
    // epilogue:  v = alpha*acc + bias_n[n] + bias_m[m];  v *= dquickgelu(aux) ; v += resid
    float alpha = 1.f;
    const float* bias_n = nullptr;
    const float* bias_m = nullptr;
    const bf16_t* aux = nullptr; int ldaux = 0;
    const float* resid = nullptr; int ldr = 0;
    int act = PRX_ACT_NONE;
    float* out_f32 = nullptr; int ldc_f32 = 0;
    bf16_t* out_bf16 = nullptr;      // post-activation
    bf16_t* out_bf16_pre = nullptr;  // pre-activation (QUICKGELU only)
    int ldc_bf16 = 0;
    // optional: accumulate GroupNorm statistics of the fp32 output (sum, sum of squares per group of `gn_gs`
    // consecutive columns) into gn_stats[group*2 + {0,1}] (double, pre-zeroed) -- saves the separate stats pass over
    // the conv output.  Needs the vector epilogue (N % 4 == 0, gn_gs % 4 == 0).
    double* gn_stats = nullptr;
    int gn_gs = 0;
    // ... or, when gnb_x is set, the BACKWARD sums of the GroupNorm(+swish) whose output gradient this GEMM produces
    // (batch 1: M pixels): with xhat = (gnb_x - mean) * rstd from gnb_fstats and dxhat = out * swish'(xhat*gamma+beta)
    // * gamma, accumulate (sum dxhat, sum dxhat*xhat) per group into gn_stats -- saves the stats pass of the GroupNorm
    // backward (8 bytes read per element) for 4 bytes read in this epilogue.
    const float* gnb_x = nullptr;        // [M, N] fp32, the GroupNorm's forward input
    const double* gnb_fstats = nullptr;  // its forward sums [32][2]
    const float* gnb_gamma = nullptr;
    const float* gnb_beta = nullptr;
    int gnb_swish = 0;
    float gnb_eps = 1e-6f;
};

// Launch on `stream`.  `ws` is a scratch buffer for split-K partials (may be
// null -> split-K disabled).  Returns 0 or a negative error code.
int prx_gemm_launch(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream);

// Per-launch timing hook used by bench.py's roofline leg (HIP events on the
// launch stream).  When enabled every prx_gemm_launch is bracketed by events.
void prx_gemm_set_variant(int use_glds);   // 1 (default): direct-to-LDS v2 kernel for bf16 A; 0: register-staged v1
void prx_gemm_force_tile(int bm, int bn, int splits);   // tuning override; (0,0,0) restores the heuristic
void prx_gemm_tile_rule_set(int M, int N, int K, int mode, int bm, int bn, int splits);   // per-shape override; bm=0 drops it, M=0 drops all
void prx_gemm_profile_enable(int on);
int prx_gemm_profile_collect(double* total_ms, double* total_flop, long long* launches);
