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

#define PRX_ABI_VERSION 3   /* 3: prx_gemm_args.row16; prx_prompt_loss_fwd_bwd(loss, ticket); 36-word cutout descriptors */

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
#define PRX_ACT_RELU 3           /* max(v, 0) after bias / residual (CLIP ModifiedResNet [UPSTREAM clip/model.py]) */
#define PRX_ACT_MUL_RELUMASK 4   /* multiply by (aux > 0): ReLU backward, aux = the forward output */
#define PRX_ACT_RELUMASK_POST 5  /* the same mask applied AFTER the residual add: (acc + resid) * (aux > 0) -- the gradient
                                    arriving at a Bottleneck's output ReLU, formed in the epilogue of the GEMM that sums it */
#define PRX_A_ROWMAJOR 0
#define PRX_A_CONV3X3 1          /* implicit im2col of an NHWC tensor, 3x3 pad 1 */

/* Operand precision of a runner handle (the `precision` field of the *_config structs below):
 *   PRX_PREC_BF16  GEMM operands (inter-kernel activations, weight packs) bf16, fp32 accumulate on
 *                  v_mfma_f32_32x32x16_bf16; residual streams, norms, softmax statistics, loss, optimiser fp32.
 *   PRX_PREC_F32   every operand fp32 end to end on v_mfma_f32_32x32x2_f32 (exact f32 = what the reference's CPU
 *                  path computes, pixray.py:275-280, vqgan.py:60-79, slip.py:21-66): the parity mode, 1/16 of the rate.
 *   PRX_PREC_F16   the same data flow as PRX_PREC_BF16 with IEEE half operands on v_mfma_f32_32x32x16_f16 (same MFMA rate):
 *                  the arithmetic the reference's CLIP towers run in on a GPU -- `clip.load` keeps fp16 weights and
 *                  activations with fp32 LayerNorm (slip.py:175) -- with 11 significand bits instead
 *                  of bf16's 8.  Conversions saturate at +-65504, and every runner backward runs under a power-of-two
 *                  gradient scale S chosen ON THE DEVICE from the gradient entering it (S * max|g| in [8, 16); no host
 *                  synchronisation; exact, because each backward op is linear in the incoming gradient and ClampWithGrad /
 *                  ReLU masks only read signs) that is removed before anything leaves the handle.  The VGG16 extractor of the StyleLoss plugin has the same three modes; the VQGAN encoder and the CLIP text tower (forward only) stay bf16 / fp32. */
#define PRX_PREC_BF16 0
#define PRX_PREC_F32 1
#define PRX_PREC_F16 2

/* Engine state of ONE handle: tile / split-K overrides and the optional per-launch timing log.  Each runner handle owns
 * one (prx_*_gemm_ctx below); the library has no process-global mutable state. */
typedef struct prx_gemm_ctx prx_gemm_ctx;
prx_gemm_ctx* prx_gemm_ctx_create(void);
void prx_gemm_ctx_destroy(prx_gemm_ctx* c);

/* C[M,N] = epilogue(alpha * A[M,K] * Bt[N,K]^T); MFMA, fp32 accumulate.
 * Replaces torch.nn.functional.linear / conv2d under clip.model.* and
 * taming Decoder (call sites slip.py:65, vqgan.py:195). */
typedef struct prx_gemm_args {
    const void* A;      /* device; bf16 (a_is_f32=0) or fp32 converted on load */
    int a_is_f32, a_mode, lda;
    const void* B;      /* device bf16 [N, K], K contiguous */
    int ldb;
    int M, N, K;
    int H, W, Cin, up;  /* conv geometry (a_mode == PRX_A_CONV3X3): output H x W; up = 0 source H x W, 1 source H/2 x W/2 read through
                         * a nearest-2x upsample, 2 source 2H x 2W read with stride 2 and taming's (0,1,0,1) Downsample padding */
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
    int f32;            /* operand precision, PRX_PREC_*: _F32 = A, B, aux, out_bf16, out_bf16_pre are all fp32 and the product is
                         * exact f32; _F16 = the 16-bit operands are IEEE half; _BF16 (0) = bf16 */
    int row16;          /* 16-bit operand modes: bit 0 = `resid` addresses a 16-bit stream in the operand format (the runners' lean
                         * layout: residual / feature-map streams kept in 16 bits), bit 1 = so does prx_k_gemm_gn's `gnb_x` */
    prx_gemm_ctx* ctx;  /* tuning / timing context or NULL (built-in heuristics) */
} prx_gemm_args;
int prx_k_gemm(const prx_gemm_args* g, void* ws, size_t ws_bytes, prx_stream_t stream);
/* The same product with the decoder's fused GroupNorm sums in the epilogue (taming `Normalize`, 32 groups of gn_gs = N / 32
 * consecutive channels; call site vqgan.py:195): gn_stats (double[64], pre-zeroed, accumulated with atomics) receives per group
 * {sum, sum of squares} of the fp32 output -- or, when gnb_x is given (the forward INPUT [M, N] of the GroupNorm whose output
 * gradient this product is, with its forward sums gnb_fstats and affine parameters), that GroupNorm's backward sums
 * {sum dxhat, sum dxhat * xhat}.  Needs N == 32 * gn_gs, gn_gs a multiple of 4, 16-bit operands. */
int prx_k_gemm_gn(const prx_gemm_args* g, double* gn_stats, int gn_gs, const float* gnb_x, const double* gnb_fstats,
                  const float* gnb_gamma, const float* gnb_beta, int gnb_swish, float gnb_eps, void* ws, size_t ws_bytes,
                  prx_stream_t stream);
/* Diagnostics: how many launches of this process ran a fit or 8-phase kernel whose epilogue is specialised at compile time
 * (csrc/gemmfit_kernel.h FIT_EPI_*, csrc/gemm8p.hip: the descriptor patterns of the two runners, IEEE-half operands) -- tests use it to know that the specialised kernel,
 * not the generic one, produced what they compare. */
long long prx_gemm_fit_spec_launches(void);
/* ... and how many ran the row-streaming kernel (csrc/gemmrow.hip: row-major 16-bit problems with K <= 192, N a multiple of 128 or
 * 160, M N >= 5 Mi -- the ModifiedResNet runner's stage-1 / stage-2 1x1 convolutions; PRX_GEMM_ROWK=0 keeps them on the tiled kernels). */
long long prx_gemm_row_launches(void);

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

/* ------------------------------------------------------------------------ */
/* Path-level operators: the drop-in boundary of the hot path                */
/* ------------------------------------------------------------------------ */

/* --- VqganDrawer (vqgan.py:83-214): synth = vector_quantize (60-64, straight-through 48-58) ->
 *     taming VQModel.decode [UPSTREAM] -> add(1).div(2) -> clamp_with_grad(0,1) (66-79,195).
 * weights[]: fp32 device tensors in taming state-dict order:
 *   quantize.embedding.weight [n_embed, embed_dim];
 *   post_quant_conv.{weight[zc,ed,1,1], bias}; decoder.conv_in.{weight, bias};
 *   mid.block_1 (ResnetBlock), mid.attn_1 (AttnBlock), mid.block_2;
 *   for level = n_mult-1 .. 0: (num_res_blocks+1) x [ResnetBlock (+ AttnBlock at attn_resolution)], then
 *     upsample.conv.{weight,bias} when level != 0;
 *   norm_out.{weight,bias}; conv_out.{weight,bias}.
 *   ResnetBlock = norm1.{w,b}, conv1.{w,b}, norm2.{w,b}, conv2.{w,b} [, nin_shortcut.{w,b} if Cin != Cout]
 *   AttnBlock   = norm.{w,b}, q.{w,b}, k.{w,b}, v.{w,b}, proj_out.{w,b}
 * The caller may free the weight tensors after the stream has drained. */
typedef struct prx_vqgan prx_vqgan;
typedef struct prx_vqgan_config {
    int ch;                /* 128 */
    int ch_mult[8];        /* {1,1,2,2,4} */
    int n_mult;            /* 5 -> f = 16 (DrawingInterface.get_num_resolutions, vqgan.py:187-188) */
    int num_res_blocks;    /* 2 */
    int attn_resolution;   /* 16 */
    int resolution;        /* nominal training resolution of the config, 256 */
    int z_channels;        /* 256 */
    int embed_dim;         /* 256 */
    int n_embed;           /* 16384 */
    int out_ch;            /* 3 */
    int latent_h, latent_w;/* z is [1, z_channels, latent_h, latent_w] */
    int precision;         /* PRX_PREC_F16 | PRX_PREC_BF16 (fast paths) | PRX_PREC_F32 (exact-f32 MFMA parity mode); the encoder ignores it */
} prx_vqgan_config;
int prx_vqgan_create(prx_vqgan** out, const prx_vqgan_config* cfg, const float* const* weights, int n_weights,
                     prx_stream_t s);
void prx_vqgan_destroy(prx_vqgan* h);
/* per-channel codebook min/max used by VqganDrawer.clip_z (vqgan.py:155-158, 202-204) */
int prx_vqgan_z_bounds(prx_vqgan* h, float* zmin, float* zmax, prx_stream_t s);
/* z [1,zc,h,w] -> img [1,out_ch,16h,16w] in [0,1]; indices (optional, int32 [h*w]) = chosen codes;
 * quantize=0 skips the VQ step (decode of an already-quantised latent). */
int prx_vqgan_synth(prx_vqgan* h, const float* z, float* img, int* indices, int quantize, prx_stream_t s);
/* d loss / d img -> d loss / d z of the forward in flight on this handle */
int prx_vqgan_synth_backward(prx_vqgan* h, const float* g_img, float* dz, prx_stream_t s);
/* Diagnostic (tools/debug_repeat.py, tests): copies one intermediate of the last forward, fp32, to dst and returns the
 * number of floats copied (-1: bad stage).  stage -2 quantised latent, -1 conv_in output, 0..n-1 decoder stage outputs,
 * n conv_out output [H*W,4], n+1 the forward GroupNorm sums. */
long long prx_vqgan_debug_stage(prx_vqgan* h, int stage, float* dst, long long max_floats, prx_stream_t stream);

/* --- VqganDrawer.init_from_tensor / reapply_from_tensor / get_z_from_tensor (vqgan.py:174-185):
 *     `z, *_ = model.encode(img)` = taming Encoder -> quant_conv -> VectorQuantizer2 nearest code [UPSTREAM].  Forward only.
 * cfg: the same prx_vqgan_config (out_ch unused; latent_h/latent_w ignored: the latent is H/f x W/f).
 * weights[]: fp32 device tensors in this order:
 *   quantize.embedding.weight [n_embed, embed_dim]; encoder.conv_in.{weight,bias};
 *   for level = 0 .. n_mult-1: num_res_blocks x [ResnetBlock (+ AttnBlock at attn_resolution)], then
 *     downsample.conv.{weight,bias} when level != n_mult-1;
 *   mid.block_1, mid.attn_1, mid.block_2; norm_out.{weight,bias}; conv_out.{weight,bias}; quant_conv.{weight,bias}.
 * img: NCHW [1, in_channels, H, W] fp32 in [-1,1] (pixray.py:718-727); z: NCHW [1, embed_dim, H/f, W/f] = the chosen code
 * vectors; z_pre (optional): the latent before quantisation; indices (optional): int32 [H/f * W/f]. */
typedef struct prx_vqgan_enc prx_vqgan_enc;
int prx_vqgan_enc_create(prx_vqgan_enc** out, const prx_vqgan_config* cfg, int in_channels, int H, int W,
                         const float* const* weights, int n_weights, prx_stream_t s);
void prx_vqgan_enc_destroy(prx_vqgan_enc* h);
int prx_vqgan_encode(prx_vqgan_enc* h, const float* img, float* z, float* z_pre, int* indices, prx_stream_t s);

/* --- VGG16 feature extractor of the StyleLoss plugin (Losses/StyleLoss.py:24-47: torchvision `vgg16().features`, frozen,
 * the nine captured ReLU outputs at features[1,3,6,8,11,13,15,22,29]).  Batch 1, any H x W in [16, max] x [16, max].
 * weights: fp32 device pointers in torchvision order {features.N.weight [Cout,Cin,3,3], features.N.bias} for the 13 convs.
 * x: [3,H,W] fp32, already in the extractor's input space (StyleLoss.py:41-45 normalises outside this call).
 * feats[k] (k = 0..8, or NULL when not wanted): fp32 NHWC [h_k*w_k, C_k], shape from prx_vgg16_feature_shape.
 * workspace: caller-owned device buffer of prx_vgg16_workspace_bytes(H, W) bytes holding this forward's activations; the
 * matching backward reads it, so any number of forward passes can be alive at once.  precision: PRX_PREC_BF16 | PRX_PREC_F16 |
 * PRX_PREC_F32 (the workspace of the exact mode is twice as large; the half mode's backward runs under the power-of-two
 * gradient scale described at PRX_PREC_F16, chosen from max |g_feats| on the device).
 * backward: g_feats[k] fp32 NHWC gradient of feature k (or NULL); g_x: [3,H,W] fp32, overwritten. */
typedef struct prx_vgg16 prx_vgg16;
int prx_vgg16_create(prx_vgg16** out, const float* const* weights, int n_weights, int max_h, int max_w, int precision,
                     prx_stream_t s);
void prx_vgg16_destroy(prx_vgg16* h);
long long prx_vgg16_workspace_bytes(int H, int W, int precision);
int prx_vgg16_feature_shape(int H, int W, int k, int* h, int* w, int* c);
int prx_vgg16_forward(prx_vgg16* h, const float* x, int H, int W, void* workspace, float* const* feats, prx_stream_t s);
int prx_vgg16_backward(prx_vgg16* h, int H, int W, const void* workspace, const float* const* g_feats, float* g_x, prx_stream_t s);

/* --- the fft drawer's spectrum -> image map and its backward (BASELINE.json configs[3]; fftdrawer.py:45-62 `init_from_tensor`,
 * 79-86 `synth`: aphantasia fft_image(sd 0.01, decay_power) + to_valid_rgb(colors), evaluated at `contrast`):
 *   image = sigmoid(colour matrix (irfft2(scale * spectrum, s = (H, W), norm = "ortho") * contrast / std))
 * The inverse real transform runs as exact-f32 GEMMs against twiddle matrices built in float64 at creation (no FFT library;
 * csrc/fft_drawer.hip).  params / g_params: fp32 [3][H][Wf][2] (real, imaginary), Wf = prx_fft_drawer_freq_columns() =
 * W/2 + 1, or W/2 + 2 for an odd W (the surplus column of the lucid frequency helper is ignored and gets a zero gradient).
 * image / g_image: fp32 [3][H][W] in (0,1).  backward differentiates the LAST synth of the handle (it keeps the
 * un-normalised image and its moments).  Asynchronous on `stream`; one handle per canvas size, not shared between threads. */
typedef struct prx_fft_drawer prx_fft_drawer;
prx_fft_drawer* prx_fft_drawer_create(int W, int H, float decay, float colors);
void prx_fft_drawer_destroy(prx_fft_drawer* h);
int prx_fft_drawer_freq_columns(const prx_fft_drawer* h);
int prx_fft_drawer_synth(prx_fft_drawer* h, const float* params, float contrast, float* image, prx_stream_t stream);
int prx_fft_drawer_backward(prx_fft_drawer* h, const float* g_image, float* g_params, prx_stream_t stream);

/* --- STROTSS hyper-column sampling of the StyleLoss plugin (`spatial_feature_extract`, Losses/StyleLoss.py:169-223): n
 * positions, one bilinear sample of each of n_layers NHWC fp32 feature maps per position, concatenated over channels, plus
 * the two coordinate channels -> out [n, ldo] (ldo >= sum(channels) + 2).
 * feats / g_feats / channels: HOST arrays of n_layers device pointers / channel counts.  rows: device int64 [n_layers][4][n],
 * flat row (pixel) index of the four taps; weights: device fp32 [4*n_layers + 2][n], the tap weights, then the x and y
 * coordinate channels.  Same roundings as the composed torch expression (products added left to right).
 * backward: g_feats[l] (or NULL) must be zero-initialised [h_l*w_l, C_l]; adds weight * g_out to the four taps. */
int prx_hypercolumns_fwd(const float* const* feats, const int* channels, int n_layers, const long long* rows, const float* weights,
                         int n, float* out, int ldo, prx_stream_t s);
int prx_hypercolumns_bwd(float* const* g_feats, const int* channels, int n_layers, const long long* rows, const float* weights,
                         int n, const float* g_out, int ldo, prx_stream_t s);

/* --- STROTSS distance arithmetic of the StyleLoss plugin (Losses/StyleLoss.py:225-293), csrc/strotss.hip.  All matrices fp32
 * row-major on the device; the products G = X Y^T are the caller's (plain library GEMMs).
 * xs / ys: the squared row norms `pairwise_distances_cos` / `_sq_l2` take (225-237), the caller's reductions.
 * remd_fwd (`style_loss`, 272-293): G [n, m], xs [n], ys [m] -> stats = {max(rmean, cmean), rmean, cmean} where rmean / cmean are
 *   the means of the row / column minima of M = 1 - (G / |x|) / |y| (+ sqrt(clamp(|x|^2 + |y|^2 - 2 G, 1e-5, 1e5) / d) when l2:
 *   the 3-channel palette term); rowpack [n] / colpack [m]: {ordered value << 32 | position} of each minimum (ties: smallest
 *   position), kept for the backward.
 * remd_bwd: dX [n, d] = g_out[0] * d stats[0] / dX, visiting only the n + m selected pairs (X [n, d], Y [m, d]; Y takes no gradient:
 *   it is the style image's), a row's pairs in chunks of 32 per workgroup, added in ascending column order (no floating-point
 *   atomics).  d <= 4096; workspace: remd_bwd_workspace_bytes(n, m, d) bytes, 16-byte aligned.
 * selfsim_fwd (`content_loss`, 246-265): out[0] = mean |D(X, X) - D(Y, Y)| from Gx = X X^T, Gy = Y Y^T [n, n]; partial: n doubles.
 * selfsim_bwd: Sx / Sy [n, lds] = dL/dG + (dL/dG)^T of each product and cx / cy [n] such that dX = Sx X + cx (.) X (rows scaled),
 *   dY = Sy Y + cy (.) Y. */
long long prx_strotss_remd_bwd_workspace_bytes(int n, int m, int d);
int prx_strotss_remd_fwd(const float* G, int ldg, const float* xs, const float* ys, int n, int m, int l2, int d,
                         unsigned long long* rowpack, unsigned long long* colpack, float* stats, prx_stream_t s);
int prx_strotss_remd_bwd(const float* G, int ldg, const float* X, int ldx, const float* Y, int ldy, int d, const float* xs, const float* ys,
                         const unsigned long long* rowpack, const unsigned long long* colpack, int n, int m, int l2, const float* stats,
                         const float* g_out, void* workspace, long long workspace_bytes, float* dX, int lddx, prx_stream_t s);
int prx_strotss_selfsim_fwd(const float* Gx, int ldgx, const float* xs, const float* Gy, int ldgy, const float* ys, int n,
                            double* partial, float* out, prx_stream_t s);
int prx_strotss_selfsim_bwd(const float* Gx, int ldgx, const float* xs, const float* Gy, int ldgy, const float* ys, int n,
                            const float* g_out, float* Sx, float* Sy, int lds, float* cx, float* cy, prx_stream_t s);

/* --- MakeCutouts.forward (pixray.py:445-511) with explicit randomness.
 * desc: fp64 [n_cut][36] per-cutout descriptor (built by pixray_amd/cutouts.py::build_descriptors):
 *   [0..8] stage-A 3x3, [9..17] stage-B 3x3: kornia's src_norm_trans_dst_norm (normalised destination
 *          coords -> normalised source coords, [0,W-1]->[-1,1] convention),
 *   [18] stage-A mode, [19] stage-B mode (0 copy, 1 zeros, 2 border, 3 reflection, 4 fill, 5 reflection under align_corners=True),
 *   [20] fill gray, [21] jitter on/off, [22] saturation factor, [23] hue shift (rad), [24] saturation-first,
 *   [25] noise factor, [26]/[27] stage-A/B grid flavour = which kornia 0.6.2 call built the sampling grid AND the
 *   align_corners flag that call hands to F.grid_sample (the convention is data, pixray_amd.cutouts.KORNIA_062_CONVENTIONS):
 *   0 warp_perspective(align_corners=False): create_meshgrid + transform_points, sampled with (g+1)*W/2-0.5 (RandomPerspective);
 *   1 warp_affine(align_corners=False): F.affine_grid pixel-centre grid (RandomAffine); 2 warp_affine(align_corners=True):
 *   corner-aligned grid sampled with (g+1)/2*(W-1) (RandomResizedCrop / CenterCrop via crop_by_transform_mat);
 *   3 warp_perspective(align_corners=True) (the cached-transform path, pixray.py:482-485),
 *   [28..31] stage-B source window (x, y, width, height) inside the stage-A image,
 *   [32] noise seed (an integer < 2^53, 0 = none): with `noise` NULL and a non-zero factor [25] the cutout's additive N(0,1)
 *        draws (pixray.py:508-510, randn_like) are generated in the kernel -- Philox4x32-10 keyed by the seed, counter = pixel
 *        index, Box-Muller; [33..35] reserved (zero).
 * Geometry: the canvas is pooled to [3,S,S] (pixray.py:463); on a W != H canvas the reference rescales that to the
 * canvas aspect (pixray.py:468-472): the "base" image [3,Hb,Wb] with Hb == S or Wb == S (Hb = Wb = S on a square
 * canvas, `base` may then be NULL).  Stage A renders [n_cut,3,Hb,Wb] from the base, stage B the S x S cutouts.
 * noise: fp32 [n_cut,3,S,S] explicit N(0,1) draws (parity tests hand the oracle's), or NULL (see [32]).  pooled/argmax/base/stage_a are caller-owned save-for-backward
 * buffers ([3,S,S] f32, [3,S,S] i32, [3,Hb,Wb] f32, [n_cut,3,Hb,Wb] f32).
 * spot_mask (optional): uint8 [3,S,S]; pooled pixels where it is non-zero are set to 0 (spot prompts, pixray.py:453-466). */
int prx_cutouts_forward(const float* img, int H, int W, const double* desc, const float* noise, const unsigned char* spot_mask,
                        int n_cut, int S, int Hb, int Wb, float* pooled, int* argmax, float* base, float* stage_a, float* out,
                        prx_stream_t s);
/* scratch: g_stage_a and g_base_priv are [n_cut,3,Hb,Wb] fp32 each, uv_scratch is [n_cut,Hb*Wb,2] fp32 (the forward's sampling
 * coordinates, recomputed per stage), g_base is [3,Hb,Wb], g_pooled is [3,S,S].  The backward is in
 * gather form (every source pixel sums its contributions in a fixed order): no atomics, bit-reproducible (cf. pixray.py:29). */
int prx_cutouts_backward(const float* g_out, const double* desc, const unsigned char* spot_mask, int n_cut, int S, int Hb, int Wb,
                         int H, int W, const float* stage_a, const int* argmax, float* g_stage_a, float* g_base_priv, float* uv_scratch,
                         float* g_base, float* g_pooled, float* g_img, prx_stream_t s);

/* --- CLIP_Base.encode_image (slip.py:62-66) for a ViT visual tower [UPSTREAM clip/model.py].
 * weights[]: fp32 device tensors in OpenAI state-dict order under `visual.`:
 *   conv1.weight, class_embedding, positional_embedding, ln_pre.{weight,bias},
 *   per layer: ln_1.{weight,bias}, attn.in_proj_{weight,bias}, attn.out_proj.{weight,bias},
 *              ln_2.{weight,bias}, mlp.c_fc.{weight,bias}, mlp.c_proj.{weight,bias},
 *   ln_post.{weight,bias}, proj.
 * The batch-global min/max renorm (slip.py:21-36) couples all cutouts, so the call sequence is split to
 * let a sharded caller all-reduce the two small buffers in between:
 *   minmax(cutouts) -> mm[2] {min,max}      [all-reduce MIN/MAX over ranks]
 *   encode(cutouts, mm) -> embeds [n, output_dim] (unit vectors)
 *   backward_reduce(d_embeds) -> acc[4] doubles {sum g, sum g*y, #min, #max}   [all-reduce SUM]
 *   backward_finish(acc) -> g_cutouts [n,3,R,R] */
typedef struct prx_clip_vit prx_clip_vit;
typedef struct prx_clip_vit_config {
    int input_resolution;  /* 224 */
    int patch_size;        /* 32 */
    int width;             /* 768 */
    int layers;            /* 12 */
    int heads;             /* 12 */
    int output_dim;        /* 512 */
    int max_batch;         /* capacity in cutouts */
    int precision;         /* PRX_PREC_F16 | PRX_PREC_BF16 | PRX_PREC_F32 */
} prx_clip_vit_config;
int prx_clip_vit_create(prx_clip_vit** out, const prx_clip_vit_config* cfg, const float* const* weights, int n_weights,
                        prx_stream_t s);
void prx_clip_vit_destroy(prx_clip_vit* h);
int prx_clip_vit_minmax(prx_clip_vit* h, const float* cutouts, int n, float* mm, prx_stream_t s);
int prx_clip_vit_encode(prx_clip_vit* h, const float* cutouts, int n, const float* mm, float* embeds, prx_stream_t s);
int prx_clip_vit_backward_reduce(prx_clip_vit* h, const float* cutouts, const float* mm, const float* d_embeds,
                                 double* acc, prx_stream_t s);
int prx_clip_vit_backward_finish(prx_clip_vit* h, const float* cutouts, const float* mm, const double* acc,
                                 float* g_cutouts, prx_stream_t s);

/* --- CLIP_Base.encode_image for a ModifiedResNet visual tower (RN50x4, RN50, ...) [UPSTREAM clip/model.py: ModifiedResNet,
 *     Bottleneck, AttentionPool2d].  Same call sequence and min/max contract as the ViT entry points above.
 * weights[]: fp32 device tensors with the eval-mode BatchNorms folded into their convolutions on the host
 *   (pixray_amd/weights.py::fold_clip_resnet_params, from the `visual.*` state dict):
 *   stem conv1 {weight [w/2,3,3,3], bias}, stem conv2 {weight, bias}, stem conv3 {weight, bias};
 *   per Bottleneck: conv1 {weight [planes,in,1,1], bias}, conv2 {weight [planes,planes,3,3], bias}, conv3 {weight
 *   [4*planes,planes,1,1], bias} [, downsample {weight [4*planes,in,1,1], bias}];
 *   attnpool.positional_embedding [(R/32)^2+1, 32w], in_proj {weight [3*32w, 32w] = q|k|v, bias}, c_proj {weight, bias}. */
typedef struct prx_clip_resnet prx_clip_resnet;
typedef struct prx_clip_resnet_config {
    int input_resolution;  /* 288 (RN50x4) */
    int width;             /* 80 */
    int layers[4];         /* {4, 6, 10, 6} */
    int heads;             /* 40 = width * 32 / 64 */
    int output_dim;        /* 640 */
    int max_batch;
    int precision;         /* PRX_PREC_F16 | PRX_PREC_BF16 | PRX_PREC_F32 */
} prx_clip_resnet_config;
int prx_clip_resnet_create(prx_clip_resnet** out, const prx_clip_resnet_config* cfg, const float* const* weights, int n_weights,
                           prx_stream_t s);
void prx_clip_resnet_destroy(prx_clip_resnet* h);
int prx_clip_resnet_minmax(prx_clip_resnet* h, const float* cutouts, int n, float* mm, prx_stream_t s);
int prx_clip_resnet_encode(prx_clip_resnet* h, const float* cutouts, int n, const float* mm, float* embeds, prx_stream_t s);
int prx_clip_resnet_backward_reduce(prx_clip_resnet* h, const float* cutouts, const float* mm, const float* d_embeds,
                                    double* acc, prx_stream_t s);
int prx_clip_resnet_backward_finish(prx_clip_resnet* h, const float* cutouts, const float* mm, const double* acc,
                                    float* g_cutouts, prx_stream_t s);

/* --- CLIP_Base.encode_text (slip.py:68-70) = openai/CLIP `CLIP.encode_text` [UPSTREAM clip/model.py] on token ids:
 *     token_embedding[tokens] + positional_embedding -> causal transformer -> ln_final -> row at argmax(tokens) (the EOT
 *     token) @ text_projection.  Forward only; the result is NOT normalised (as the reference's, slip.py:70).
 * weights[]: fp32 device tensors in state-dict order: token_embedding.weight [vocab, width], positional_embedding
 *   [context, width], transformer.resblocks.{i}.{ln_1.weight, ln_1.bias, attn.in_proj_weight, attn.in_proj_bias,
 *   attn.out_proj.weight, attn.out_proj.bias, ln_2.weight, ln_2.bias, mlp.c_fc.weight, mlp.c_fc.bias, mlp.c_proj.weight,
 *   mlp.c_proj.bias}, ln_final.{weight,bias}, text_projection [width, output_dim].
 * tokens: int32 device [n, context] as `clip.tokenize` returns them (SOT ... EOT, zero padded). */
typedef struct prx_clip_text prx_clip_text;
typedef struct prx_clip_text_config {
    int vocab_size;       /* 49408 */
    int context_length;   /* 77 */
    int width;            /* 512 (ViT-B/32, B/16), 768 (ViT-L/14); heads = width / 64 */
    int layers;           /* 12 */
    int heads;            /* 8 */
    int output_dim;       /* 512 */
    int max_batch;
} prx_clip_text_config;
int prx_clip_text_create(prx_clip_text** out, const prx_clip_text_config* cfg, const float* const* weights, int n_weights,
                         prx_stream_t s);
void prx_clip_text_destroy(prx_clip_text* h);
int prx_clip_text_encode(prx_clip_text* h, const int* tokens, int n, float* embeds, prx_stream_t s);

/* --- Prompt.forward (pixray.py:275-280) fused with its backward.
 * rowloss[i] = sum_j sign(w) * 2*asin(|x^_i - e^_j|/2)^2;  *loss (optional, may be NULL) = |w| * sum(rowloss) / denom, the value
 * Prompt.forward returns, written by the workgroup that finishes last (rows added in a fixed order); `ticket`: one device word
 * that is zero on entry and zero again on exit (needed with `loss`; one per stream that may run this concurrently);
 * grad = d/d input of |w| * mean(max(sign(w) d, stop)) with the mean over `denom` (= global n*m) pairs. */
int prx_prompt_loss_fwd_bwd(const float* input, const float* embed, int n, int m, int D, float weight, float stop,
                            float denom, float* rowloss, float* grad, float* loss, unsigned* ticket, prx_stream_t s);

/* --- optim.Adam([z], lr) step (pixray.py:539,1484-1485) fused with VqganDrawer.clip_z (vqgan.py:202-204).
 * z/exp_avg/exp_avg_sq/grad: [1,C,hw] fp32; zmin/zmax per channel or NULL; step is 1-based. */
int prx_adam_clamp_step(float* z, float* exp_avg, float* exp_avg_sq, const float* grad, const float* zmin,
                        const float* zmax, int hw, size_t n, float lr, float beta1, float beta2, float eps, int step,
                        prx_stream_t s);

/* same step with the step-dependent scalars read from device memory: hyper = {lr / (1 - beta1^t), sqrt(1 - beta2^t)}.
 * Used when the whole iteration is captured in a hipGraph and replayed (kernel arguments are frozen at capture). */
int prx_adam_clamp_step_dev(float* z, float* exp_avg, float* exp_avg_sq, const float* grad, const float* zmin,
                            const float* zmax, int hw, size_t n, const float* hyper, float beta1, float beta2, float eps,
                            prx_stream_t s);

int prx_k_vq_nearest(const float* z, long long tok_stride, long long ch_stride, const float* codebook,
                     const float* cnorm, int P, int NC, int D, float* pmin, int* pidx, int* idx_out, float* zq,
                     prx_stream_t s);
int prx_k_sqnorm_rows(const float* w, float* out, int rows, int D, prx_stream_t s);

/* same attention for any sequence length (ViT-B/16: 197 tokens, ViT-L/14: 257): 64-token tiles, online softmax;
 * `out` and `lse` (fp32 [N*heads*T]) are kept for the backward */
int prx_k_mha_fwd_gen(const void* qkv, void* out, float* lse, int N, int T, int C, int heads, prx_stream_t s);
int prx_k_mha_bwd_gen(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int N, int T, int C,
                      int heads, prx_stream_t s);

/* the exact-f32 attention of the PRX_PREC_F32 mode (fp32 qkv / out / dout / dqkv, any T, head dim 64) */
int prx_k_mha_fwd_f32(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, prx_stream_t s);
int prx_k_mha_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int N, int T, int C,
                      int heads, prx_stream_t s);

/* tuning override of the tile / split-K heuristic of one context: bm,bn in {(128,128),(128,64),(64,64),(256,128)};
 * (0,0,0) = heuristic.  Negative bm selects a switch: (-1,_,v) XCD-aware tile order 0/1/2; (-2,_,n) LDS pipeline depth;
 * (-3,_,v) 1 = direct-to-LDS v2 kernel (default), 0 = register-staged v1; (-5,_,v) scalar-tap conv gather; (-14,_,n) plan the
 * launches of this context for n compute units (it shares the chip with concurrent chains on other streams; 0 = the device's). */
void prx_gemm_tile_override(prx_gemm_ctx* c, int bm, int bn, int splits);
/* the same per problem shape (tools/gemm_rules.py): mode = a_mode + 2*up + 4*a_is_f32; splits 0 = heuristic; bm = 0 drops
 * the rule, M = 0 drops all rules */
void prx_gemm_tile_rule(prx_gemm_ctx* c, int M, int N, int K, int mode, int bm, int bn, int splits);

/* what the launch planner does with a row-major 16-bit C[M,N] = A[M,K] * Bt[N,K]^T (no launch, no device needed): returns the number of
 * leading rows it gives to the 256 x 256 8-phase kernel -- M (all), 0 (none: 4-wave kernels), or a multiple of 256 in between
 * (whole rounds of 256 tiles on the 8-phase kernel, the remaining rows on the 4-wave kernels).  c may be NULL (default tuning). */
int prx_gemm_plan_rows_8phase(prx_gemm_ctx* c, int M, int N, int K);

/* per-launch GEMM timing (HIP events on the launch stream) for bench.py */
void prx_profile_gemm_enable(prx_gemm_ctx* c, int on);
int prx_profile_gemm_collect(prx_gemm_ctx* c, double* total_ms, double* total_flop, long long* launches);

/* the engine context owned by a runner handle (valid for the handle's lifetime; do not destroy) */
prx_gemm_ctx* prx_vqgan_gemm_ctx(prx_vqgan* h);
prx_gemm_ctx* prx_clip_vit_gemm_ctx(prx_clip_vit* h);
prx_gemm_ctx* prx_clip_resnet_gemm_ctx(prx_clip_resnet* h);
prx_gemm_ctx* prx_vgg16_gemm_ctx(prx_vgg16* h);

/* ---- the exchange step of the cutout-sharded iteration ------------------------------------------------
 * The reference has no distributed code; what this replaces is the seam in `train()` between
 * `loss.backward()` and `opt.step()` (pixray.py:1482-1485) where a data-parallel port would all-reduce the gradient.
 * prx_allreduce_grad is a ONE-SHOT DIRECT-WRITE all-reduce (SUM, fp32, in place) over IPC-mapped peer windows: every rank
 * writes its vector straight into each peer's window over xGMI, raises a flag, waits for the peers' flags in its own window
 * and sums the slots in rank order -- bit-identical on every rank, one kernel, no ring (csrc/comm.hip).
 *   prx_comm_create    allocates this rank's window (slots of max_bytes for 2 calls in flight x world ranks)
 *   prx_comm_export    the window's IPC handle, prx_comm_handle_bytes() bytes, to be exchanged by the host side
 *   prx_comm_connect   maps the peers' windows from `world` handles in rank order (own entry ignored)
 *   prx_allreduce_grad n floats, n % 4 == 0, 16-byte aligned, n * 4 <= max_bytes; asynchronous on `stream`
 *   prx_comm_status    0, or 1 + r when a wait for rank r's data timed out (synchronises) */
typedef struct prx_comm prx_comm;
int prx_comm_handle_bytes(void);
int prx_comm_create(prx_comm** out, int rank, int world, size_t max_bytes);
int prx_comm_export(prx_comm* c, void* handle_out);
int prx_comm_connect(prx_comm* c, const void* handles);
int prx_allreduce_grad(prx_comm* c, float* grad, size_t n, prx_stream_t stream);
/* the same exchange for the two scalar-sized collectives of the sharded iteration, so that a C-ABI caller needs nothing else:
 * PRX_COMM_MAX_F32 for the {-min, max} pair of the batch-global renormalisation (slip.py:21-36), PRX_COMM_SUM_F64 for its four
 * backward sums.  n_words counts 4-byte words (a double = 2), a multiple of 4; data 16-byte aligned.  A wait that times out
 * poisons the result with NaN (never a plausible partial sum) and is reported by prx_comm_status.  The kernel's <= 64 blocks
 * must be co-resident (they are on an otherwise idle stream; see csrc/comm.hip). */
#define PRX_COMM_SUM_F32 0
#define PRX_COMM_MAX_F32 1
#define PRX_COMM_SUM_F64 2
int prx_allreduce(prx_comm* c, void* data, size_t n_words, int op, prx_stream_t stream);
int prx_comm_status(prx_comm* c);
void prx_comm_destroy(prx_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* PRX_H_ */
