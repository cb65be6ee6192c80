// MFMA GEMM / implicit-GEMM convolution engine (gfx950).
//   C[M,N] = epilogue( A[M,K] * Bt[N,K]^T )
// A is either a row-major matrix or the implicit im2col view of an NHWC
// activation for a 3x3/pad-1 convolution (optionally reading through a fused
// nearest-2x upsample or a stride-2 window).  Bt is the weight pack, K contiguous.
// Two operand precisions (GemmDesc::f32):
//   0  16-bit operands -- bf16, or IEEE half with GemmDesc::h16 (A may be f32, converted on load) -- fp32 accumulate on
//      v_mfma_f32_32x32x16_{bf16,f16} (same rate) -- the fast path;
//   1  fp32 operands end to end on v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain, the f32 vector rate = 1/16 of
//      the bf16 MFMA rate) -- the exact parity mode.  Every "bf16" pointer below then addresses fp32 data.
#pragma once
#include "common.h"
#include <vector>
#include <mutex>

enum { PRX_ACT_NONE = 0, PRX_ACT_QUICKGELU = 1, PRX_ACT_MUL_DQUICKGELU = 2,
       PRX_ACT_RELU = 3,            // max(v, 0) after bias / residual (CLIP ModifiedResNet)
       PRX_ACT_MUL_RELUMASK = 4,    // v *= (aux > 0): ReLU backward with the forward OUTPUT as aux
       PRX_ACT_RELUMASK_POST = 5 }; // the same mask after the residual add: (acc + resid) * (aux > 0)
enum { PRX_A_ROWMAJOR = 0, PRX_A_CONV3X3 = 1 };

struct GemmDesc {
    // operands
    int f32 = 0;               // 1 = A, B, aux, out_bf16, out_bf16_pre are all fp32 (exact mode)
    int h16 = 0;               // 16-bit operand format when f32 == 0: 0 = bf16, 1 = IEEE half (v_mfma_f32_32x32x16_f16)
    const void* A = nullptr;   // bf16 or f32 (a_is_f32)
    int a_is_f32 = 0;
    int a_mode = PRX_A_ROWMAJOR;
    int lda = 0;               // row stride (row-major) or pixel stride (conv), elements
    const void* B = nullptr;   // [N, K], ldb (operand precision)
    int ldb = 0;
    int M = 0, N = 0, K = 0;
    // conv geometry: M = NB*H*W output pixels, K = 9*Cin; `up` reads an (H/2)x(W/2) input
    int H = 0, W = 0, Cin = 0, up = 0;
    // epilogue:  v = alpha*acc + bias_n[n] + bias_m[m];  v *= dquickgelu(aux) ; v += resid
    float alpha = 1.f;
    const float* alpha_dev = nullptr;   // optional device scalar multiplied into alpha (the half mode's gradient scale, common.h)
    const float* bias_n = nullptr;
    const float* bias_m = nullptr;
    const void* aux = nullptr; int ldaux = 0;   // operand precision
    const float* resid = nullptr; int ldr = 0;
    int act = PRX_ACT_NONE;
    float* out_f32 = nullptr; int ldc_f32 = 0;
    void* out_bf16 = nullptr;        // post-activation, operand precision (the next GEMM's A)
    void* out_bf16_pre = nullptr;    // pre-activation (QUICKGELU only), operand precision
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
    short gnb_swish = 0;
    short row16 = 0;                     // bit 0: `resid` addresses a 16-bit stream in the operand format (the lean layout of the runners: residual /
                                         // feature-map streams kept in IEEE half), bit 1: so does `gnb_x`.  Flags in what used to be half of an int, not a
                                         // second pair of pointers: the descriptor travels by value, and past its round-4 size the 128 x 128 kernels keep
                                         // it on the stack (320 bytes of scratch per lane: tests/test_host_logic.py guards that)
    float gnb_eps = 1e-6f;
};

// Per-handle engine state: tuning overrides and the optional per-launch timing log.  Every runner handle owns one, so two
// handles (or two threads driving different handles) never share mutable state; a null ctx means "built-in heuristics,
// no profiling" and touches nothing mutable.
struct GemmTileRule { int M, N, K, mode, bm, bn, splits; };
struct GemmProfRec { hipEvent_t a, b; double flop; int M, N, K, mode, bm, bn, splits; };
struct GemmCtx {
    int use_glds = 1;                 // 1: direct-to-LDS v2 kernel for bf16 A; 0: register-staged v1 (A/B comparisons)
    int force_bm = 0, force_bn = 0, force_splits = 0, force_stages = 0;
    int xcd_swizzle = 2;              // 0 off, 1 on, 2 narrow row-major problems only
    int conv_c64 = 1;                 // scalar-tap conv gather when Cin % 64 == 0
    int wide_tile = 128;
    int tile8p = 128;            // > 0: 256 x 256 tiles on the 8-phase kernel (gemm8p.hip) are CONSIDERED from this many tiles on (PRX_GEMM_8P;
                                 // gemm.hip plan_8phase then decides by cost: full rounds of 256 tiles on it, the remainder rows on the 4-wave kernels)
    int fit_flags = 1;           // gemmfit.hip A/B switches: bit 0 staggered wave groups (PRX_FIT_FLAGS)
    int n_cu = 0;                // compute units of the device this context launches on (0: not asked yet; planners then assume 256)
    int n_cu_dev = -1;           // ... and the device the count was taken from (re-asked when the context launches on another one)
    int force_fit = 0;           // with force_bm: the tile shapes both kernel families have mean the fit kernel (tests, tools)
    int dbg_only = -2, dbg_count = 0;   // bisection aid (override -13)
    int fit_conv = 1;            // ... for the implicit 3x3 convolutions as well (PRX_FIT_CONV)
    int fit = 1;                 // fit tiles (gemmfit.hip) when their grid fills the chip better (PRX_GEMM_FIT)
    int rowk = 1;                // row-streaming kernels (gemmrow.hip) for skinny-K, very tall row-major problems (PRX_GEMM_ROWK)
    int rowk_min = 24 << 20;     // ... from this many output elements on (PRX_GEMM_ROWK_MIN; override -14: tests run them on small problems)
    std::vector<GemmTileRule> rules;  // per-shape (M, N, K, mode) -> tile / split-K, consulted before the heuristic
    bool prof_on = false;
    std::vector<GemmProfRec> prof;
    std::mutex mu;                    // guards prof / prof_on (collect may run on another thread than the launches)
    GemmCtx();                        // reads the PRX_* tuning environment variables once
};
void prx_gemm_ctx_force_tile(GemmCtx* c, int bm, int bn, int splits);   // (0,0,0) restores the heuristic; bm < 0: switches, see .hip
void prx_gemm_ctx_tile_rule(GemmCtx* c, int M, int N, int K, int mode, int bm, int bn, int splits);   // bm=0 drops it, M=0 drops all
void prx_gemm_ctx_profile_enable(GemmCtx* c, int on);
int prx_gemm_ctx_profile_collect(GemmCtx* c, double* total_ms, double* total_flop, long long* launches);
int prx_gemm_plan_rows_8phase_impl(const GemmCtx* c, int M, int N, int K);   // host-only: the planner's decision for a plain row-major 16-bit product

// Launch on `stream`.  `ws` is a scratch buffer for split-K partials (may be
// null -> split-K disabled).  Returns 0 or a negative error code.
int prx_gemm_launch(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream, GemmCtx* ctx = nullptr);
