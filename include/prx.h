/* prx.h -- C ABI of libprx_hip.so, the MI355X (gfx950) implementation of
 * pixray's per-iteration hot path:
 *
 *   drawer.synth (VQGAN decode) -> MakeCutouts -> CLIP ViT encode_image ->
 *   Prompt spherical-distance loss -> backward to z -> Adam + clip_z
 *
 * The reference (pixray/pixray) has NO FFI: its boundary for this path is the
 * Python plugin surface (DrawingInterface / LossInterface / FilterInterface /
 * Prompt / perceptor.encode_image / MakeCutouts) and every operator below is
 * what PyTorch/cuDNN executes underneath it.  Each entry point cites the
 * reference call site (file:line in /root/reference) whose arithmetic it
 * replaces.  See INTEGRATION.md for the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C: pointers, ints, floats; no torch/HIP C++ types in signatures
 *     (`prx_stream_t` is a hipStream_t passed as an opaque pointer).
 *   - every function returns 0 on success, <0 on error; prx_last_error() gives
 *     a thread-local message.  Nothing throws across the ABI.
 *   - all pointers named d_* / documented "device" are HBM addresses owned by
 *     the caller; the library owns only handles made by prx_*_create.
 *   - all work is enqueued asynchronously on the given stream; no hidden
 *     device synchronisation; handles are re-entrant per handle (one forward
 *     may be in flight per handle until its backward has been enqueued).
 *   - tensors at this boundary are fp32, NCHW / row-major, exactly as the
 *     reference's PyTorch tensors are laid out.  Internal layouts (NHWC bf16
 *     operand packs, fp32 residual streams) are private.
 */
#ifndef PRX_H_
#define PRX_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRX_ABI_VERSION 1

typedef void* prx_stream_t; /* hipStream_t */

const char* prx_last_error(void);
int prx_abi_version(void);
int prx_device_info(int* cu_count, char* arch_name, int arch_name_len);

/* ------------------------------------------------------------------------ */
/* Kernel-level entry points (used by tests/ to check every kernel alone)    */
/* ------------------------------------------------------------------------ */

/* activation / A-operand modes of the GEMM engine */
#define PRX_ACT_NONE 0
#define PRX_ACT_QUICKGELU 1      /* CLIP QuickGELU, x*sigmoid(1.702x)  [UPSTREAM clip/model.py] */
#define PRX_ACT_MUL_DQUICKGELU 2 /* multiply by QuickGELU'(aux) (backward) */
#define PRX_A_ROWMAJOR 0
#define PRX_A_CONV3X3 1          /* implicit im2col of an NHWC tensor, 3x3 pad 1 */

/* C[M,N] = epilogue(alpha * A[M,K] * Bt[N,K]^T); bf16 MFMA, fp32 accumulate.
 * Replaces torch.nn.functional.linear / conv2d under clip.model.* and
 * taming Decoder (call sites slip.py:65, vqgan.py:195). */
typedef struct prx_gemm_args {
    const void* A;      /* device; bf16 (a_is_f32=0) or fp32 converted on load */
    int a_is_f32, a_mode, lda;
    const void* B;      /* device bf16 [N, K], K contiguous */
    int ldb;
    int M, N, K;
    int H, W, Cin, up;  /* conv geometry (a_mode == PRX_A_CONV3X3) */
    float alpha;
    const float* bias_n;
    const float* bias_m;
    const void* aux;    /* bf16 [M, ldaux] */
    int ldaux;
    const float* resid; /* fp32 [M, ldr] */
    int ldr;
    int act;
    float* out_f32; int ldc_f32;
    void* out_bf16; void* out_bf16_pre; int ldc_bf16;
} prx_gemm_args;
int prx_k_gemm(const prx_gemm_args* g, void* ws, size_t ws_bytes, prx_stream_t stream);

/* taming `Normalize` = GroupNorm(32, C, eps 1e-6) (+ swish `nonlinearity`) on an NHWC fp32
 * tensor x[NB][P][C]  [UPSTREAM taming/modules/diffusionmodules/model.py; call site vqgan.py:195].
 * stats: double[NB*64] workspace (sum, sum of squares per group).  bwd returns the activation
 * gradient only (weights are frozen, vqgan.py:125). */
int prx_k_groupnorm_fwd(const float* x, const float* gamma, const float* beta, double* stats, void* out_bf16,
                        float* out_f32, int NB, int P, int C, int swish, float eps, prx_stream_t s);
int prx_k_groupnorm_bwd(const float* g, const float* x, const float* gamma, const float* beta, const double* fstats,
                        double* bstats, const float* add, float* dx, int NB, int P, int C, int swish, float eps,
                        prx_stream_t s);
/* CLIP LayerNorm (eps 1e-5, computed in fp32) [UPSTREAM clip/model.py; call site slip.py:65] */
int prx_k_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, void* out_bf16,
                        float* out_f32, float* mean, float* rstd, int rows, int C, float eps, prx_stream_t s);
int prx_k_layernorm_bwd(const float* g, long long ldg, const float* x, long long ldx, const float* gamma,
                        const float* mean, const float* rstd, const float* add, long long ldadd, float* dx,
                        long long lddx, int rows, int C, prx_stream_t s);
int prx_k_transpose_bf16(const void* in, int ldin, void* out, int ldout, int R, int C, prx_stream_t s);
/* taming AttnBlock softmax over keys (single head) */
int prx_k_softmax_rows(const float* S, int lds_, float scale, void* P, int ldp, void* PT, int ldpt, int rows, int cols,
                       prx_stream_t s);
int prx_k_softmax_rows_bwd(const void* P, int ldp, const float* dP, int lddp, float scale, void* dS, int ldds,
                           void* dST, int lddst, int rows, int cols, prx_stream_t s);
/* backward of taming Upsample's nearest-2x interpolate */
int prx_k_upsample2x_bwd(const float* hi, float* low, int NB, int Hl, int Wl, int C, prx_stream_t s);
int prx_k_nchw_to_nhwc(const float* in, float* out_f32, void* out_bf16, int NB, int C, int HW, int Cpad,
                       prx_stream_t s);
int prx_k_nhwc_to_nchw(const float* in, int ldc, float* out, int NB, int C, int HW, prx_stream_t s);
/* VqganDrawer.synth tail: clamp_with_grad(decode(z_q).add(1).div(2), 0, 1)  (vqgan.py:66-79,195) */
int prx_k_image_head_fwd(const float* x, int ldc, float* img, int NB, int C, int HW, prx_stream_t s);
int prx_k_image_head_bwd(const float* x, int ldc, const float* gimg, float* dx, void* dx_bf16, int ldo, int NB, int C,
                         int HW, prx_stream_t s);
/* nn.MultiheadAttention core of CLIP's ResidualAttentionBlock, T <= 64, head dim 64 */
int prx_k_mha_fwd(const void* qkv, void* out, int N, int T, int C, int heads, prx_stream_t s);
int prx_k_mha_bwd(const void* qkv, const void* dout, void* dqkv, int N, int T, int C, int heads, prx_stream_t s);

/* per-launch GEMM timing (HIP events on the launch stream) for bench.py */
void prx_profile_gemm_enable(int on);
int prx_profile_gemm_collect(double* total_ms, double* total_flop, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* PRX_H_ */
