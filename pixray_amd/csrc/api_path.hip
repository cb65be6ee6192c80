// C-ABI path-level operators: what the Python plugin surface (pixray_amd/*.py) binds.
#include "common.h"
#include "vit.h"
#include "vqgan.h"
#include "vqgan_enc.h"
#include "vgg.h"
#include "clip_text.h"
#include "resnet.h"
#include "cutouts.h"
#include "prompt_vq.h"
#include "elementwise.h"
#include "gemm.h"
#include "../../include/prx.h"

#define S_(x) ((hipStream_t)(x))

extern "C" {

// ---- VQGAN drawer ---------------------------------------------------------------------------
int prx_vqgan_create(prx_vqgan** out, const prx_vqgan_config* c, const float* const* weights, int n_weights,
                     prx_stream_t s) {
    PRX_REQUIRE(out && c && weights, "prx_vqgan_create: null argument");
    PRX_REQUIRE(c->n_mult >= 1 && c->n_mult <= 8, "prx_vqgan_create: bad n_mult %d", c->n_mult);
    return prx_vqgan_create_impl((PrxVqgan**)out, c->ch, c->ch_mult, c->n_mult, c->num_res_blocks, c->attn_resolution,
                                 c->resolution, c->z_channels, c->embed_dim, c->n_embed, c->out_ch, c->latent_h,
                                 c->latent_w, c->precision, weights, n_weights, S_(s));
}
void prx_vqgan_destroy(prx_vqgan* h) { prx_vqgan_destroy_impl((PrxVqgan*)h); }
prx_gemm_ctx* prx_vqgan_gemm_ctx(prx_vqgan* h) { return (prx_gemm_ctx*)prx_vqgan_gemm_ctx_impl((PrxVqgan*)h); }
int prx_vqgan_z_bounds(prx_vqgan* h, float* zmin, float* zmax, prx_stream_t s) {
    PRX_REQUIRE(h, "null handle");
    return prx_vqgan_bounds_impl((PrxVqgan*)h, zmin, zmax, S_(s));
}
int prx_vqgan_synth(prx_vqgan* h, const float* z, float* img, int* indices, int quantize, prx_stream_t s) {
    PRX_REQUIRE(h && z && img, "prx_vqgan_synth: null argument");
    return prx_vqgan_synth_impl((PrxVqgan*)h, z, img, indices, quantize, S_(s));
}
long long prx_vqgan_debug_stage(prx_vqgan* h, int stage, float* dst, long long max_floats, prx_stream_t s) {
    if (!h || !dst) return -1;
    return prx_vqgan_debug_stage_impl((PrxVqgan*)h, stage, dst, max_floats, S_(s));
}
int prx_vqgan_enc_create(prx_vqgan_enc** out, const prx_vqgan_config* c, int in_channels, int H, int W,
                         const float* const* weights, int n_weights, prx_stream_t s) {
    PRX_REQUIRE(out && c && weights, "prx_vqgan_enc_create: null argument");
    PRX_REQUIRE(c->n_mult >= 1 && c->n_mult <= 8, "prx_vqgan_enc_create: bad n_mult %d", c->n_mult);
    return prx_vqgan_enc_create_impl((PrxVqganEnc**)out, c->ch, c->ch_mult, c->n_mult, c->num_res_blocks, c->attn_resolution,
                                     c->resolution, in_channels, c->z_channels, c->embed_dim, c->n_embed, H, W, weights,
                                     n_weights, S_(s));
}
void prx_vqgan_enc_destroy(prx_vqgan_enc* h) { prx_vqgan_enc_destroy_impl((PrxVqganEnc*)h); }

int prx_vgg16_create(prx_vgg16** out, const float* const* weights, int n_weights, int max_h, int max_w, int precision,
                     prx_stream_t s) {
    return prx_vgg16_create_impl((PrxVgg16**)out, weights, n_weights, max_h, max_w, precision, S_(s));
}
void prx_vgg16_destroy(prx_vgg16* h) { prx_vgg16_destroy_impl((PrxVgg16*)h); }
prx_gemm_ctx* prx_vgg16_gemm_ctx(prx_vgg16* h) { return (prx_gemm_ctx*)prx_vgg16_gemm_ctx_impl((PrxVgg16*)h); }
long long prx_vgg16_workspace_bytes(int H, int W, int precision) { return prx_vgg16_workspace_bytes_impl(H, W, precision); }
int prx_vgg16_feature_shape(int H, int W, int k, int* h, int* w, int* c) {
    PRX_REQUIRE(h && w && c, "prx_vgg16_feature_shape: null argument");
    return prx_vgg16_feature_shape_impl(H, W, k, h, w, c);
}
int prx_vgg16_forward(prx_vgg16* h, const float* x, int H, int W, void* workspace, float* const* feats, prx_stream_t s) {
    return prx_vgg16_forward_impl((PrxVgg16*)h, x, H, W, workspace, feats, S_(s));
}
int prx_vgg16_backward(prx_vgg16* h, int H, int W, const void* workspace, const float* const* g_feats, float* g_x, prx_stream_t s) {
    return prx_vgg16_backward_impl((PrxVgg16*)h, H, W, workspace, g_feats, g_x, S_(s));
}
int prx_vqgan_encode(prx_vqgan_enc* h, const float* img, float* z, float* z_pre, int* indices, prx_stream_t s) {
    PRX_REQUIRE(h && img && z, "prx_vqgan_encode: null argument");
    return prx_vqgan_encode_impl((PrxVqganEnc*)h, img, z, z_pre, indices, S_(s));
}
int prx_vqgan_synth_backward(prx_vqgan* h, const float* g_img, float* dz, prx_stream_t s) {
    PRX_REQUIRE(h && g_img && dz, "prx_vqgan_synth_backward: null argument");
    return prx_vqgan_backward_impl((PrxVqgan*)h, g_img, dz, S_(s));
}

// ---- MakeCutouts ------------------------------------------------------------------------------
int prx_cutouts_forward(const float* img, int H, int W, const double* desc, const float* noise, const unsigned char* spot_mask,
                        int n_cut, int S, int Hb, int Wb, float* pooled, int* argmax, float* base, float* stage_a, float* out,
                        prx_stream_t s) {
    PRX_REQUIRE(img && desc && pooled && argmax && stage_a && out, "prx_cutouts_forward: null argument");
    PRX_REQUIRE(Hb >= S && Wb >= S && (Hb == S || Wb == S), "prx_cutouts_forward: base %dx%d must be S=%d on one side", Hb, Wb, S);
    int r;
    if ((r = prx_pool_fwd(img, pooled, argmax, spot_mask, 3, H, W, S, S_(s)))) return r;
    const float* src = pooled;
    if (Hb != S || Wb != S) {
        PRX_REQUIRE(base, "prx_cutouts_forward: a non-square canvas needs the `base` buffer");
        if ((r = prx_rescale_fwd(pooled, base, 3, S, Hb, Wb, S_(s)))) return r;
        src = base;
    }
    if ((r = prx_warp_a_fwd(src, Hb, Wb, desc, stage_a, n_cut, Hb, Wb, S_(s)))) return r;
    return prx_warp_b_fwd(stage_a, Hb, Wb, desc, noise, out, n_cut, S, S_(s));
}
int prx_cutouts_backward(const float* g_out, const double* desc, const unsigned char* spot_mask, int n_cut, int S, int Hb, int Wb,
                         int H, int W, const float* stage_a, const int* argmax, float* g_stage_a, float* g_base_priv, float* uv_scratch,
                         float* g_base, float* g_pooled, float* g_img, prx_stream_t s) {
    PRX_REQUIRE(g_out && desc && stage_a && argmax && g_stage_a && g_base_priv && uv_scratch && g_base && g_pooled && g_img,
                "prx_cutouts_backward: null argument");
    int r;
    // g_base_priv doubles as the [n_cut,3,S,S] scratch of the ColorJitter pull-back (S <= Hb, Wb) before stage A overwrites it
    // g_pooled is written only at the very end of this call: until then it holds the per-cutout stage maps of stage B
    if ((r = prx_warp_b_bwd(stage_a, Hb, Wb, desc, g_out, g_base_priv, uv_scratch, g_stage_a, n_cut, S, S_(s), g_pooled,
                            sizeof(float) * 3 * (size_t)S * S))) return r;
    const bool rect = Hb != S || Wb != S;
    if ((r = prx_warp_a_bwd(g_stage_a, Hb, Wb, desc, uv_scratch, g_base_priv, rect ? g_base : g_pooled, n_cut, Hb, Wb, S_(s)))) return r;
    if (rect && (r = prx_rescale_bwd(g_base, g_pooled, 3, S, Hb, Wb, S_(s)))) return r;
    return prx_pool_bwd(g_pooled, argmax, spot_mask, g_img, 3, H, W, S, S_(s));
}

// ---- CLIP visual tower ------------------------------------------------------------------------
int prx_clip_vit_create(prx_clip_vit** out, const prx_clip_vit_config* c, const float* const* weights, int n_weights,
                        prx_stream_t s) {
    PRX_REQUIRE(out && c && weights, "prx_clip_vit_create: null argument");
    return prx_vit_create_impl((PrxVit**)out, c->input_resolution, c->patch_size, c->width, c->layers, c->heads,
                               c->output_dim, c->max_batch, c->precision, weights, n_weights, S_(s));
}
void prx_clip_vit_destroy(prx_clip_vit* h) { prx_vit_destroy_impl((PrxVit*)h); }
prx_gemm_ctx* prx_clip_vit_gemm_ctx(prx_clip_vit* h) { return (prx_gemm_ctx*)prx_vit_gemm_ctx_impl((PrxVit*)h); }
int prx_clip_vit_minmax(prx_clip_vit* h, const float* cutouts, int n, float* mm, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm, "prx_clip_vit_minmax: null argument");
    return prx_vit_minmax_impl((PrxVit*)h, cutouts, n, mm, S_(s));
}
int prx_clip_vit_encode(prx_clip_vit* h, const float* cutouts, int n, const float* mm, float* embeds, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm && embeds, "prx_clip_vit_encode: null argument");
    return prx_vit_forward_impl((PrxVit*)h, cutouts, n, mm, embeds, S_(s));
}
int prx_clip_vit_backward_reduce(prx_clip_vit* h, const float* cutouts, const float* mm, const float* d_embeds,
                                 double* acc, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm && d_embeds && acc, "prx_clip_vit_backward_reduce: null argument");
    return prx_vit_backward_a_impl((PrxVit*)h, cutouts, mm, d_embeds, acc, S_(s));
}
int prx_clip_vit_backward_finish(prx_clip_vit* h, const float* cutouts, const float* mm, const double* acc,
                                 float* g_cutouts, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm && acc && g_cutouts, "prx_clip_vit_backward_finish: null argument");
    return prx_vit_backward_b_impl((PrxVit*)h, cutouts, mm, acc, g_cutouts, S_(s));
}

// ---- Prompt loss, optimiser -------------------------------------------------------------------
int prx_prompt_loss_fwd_bwd(const float* input, const float* embed, int n, int m, int D, float weight, float stop,
                            float denom, float* rowloss, float* grad, float* loss, unsigned* ticket, prx_stream_t s) {
    PRX_REQUIRE(input && embed && rowloss && grad, "prx_prompt_loss_fwd_bwd: null argument");
    return prx_prompt_loss(input, embed, n, m, D, weight, stop, denom, rowloss, grad, loss, ticket, S_(s));
}
int prx_adam_clamp_step(float* z, float* exp_avg, float* exp_avg_sq, const float* grad, const float* zmin,
                        const float* zmax, int hw, size_t n, float lr, float beta1, float beta2, float eps, int step,
                        prx_stream_t s) {
    PRX_REQUIRE(z && exp_avg && exp_avg_sq && grad, "prx_adam_clamp_step: null argument");
    return prx_adam_clamp(z, exp_avg, exp_avg_sq, grad, zmin, zmax, hw, n, lr, beta1, beta2, eps, step, S_(s));
}

int prx_adam_clamp_step_dev(float* z, float* exp_avg, float* exp_avg_sq, const float* grad, const float* zmin,
                            const float* zmax, int hw, size_t n, const float* hyper, float beta1, float beta2, float eps,
                            prx_stream_t s) {
    PRX_REQUIRE(z && exp_avg && exp_avg_sq && grad && hyper, "prx_adam_clamp_step_dev: null argument");
    return prx_adam_clamp_dev(z, exp_avg, exp_avg_sq, grad, zmin, zmax, hw, n, hyper, beta1, beta2, eps, S_(s));
}

// kernel-level entries for the pieces above (tests)
int prx_k_vq_nearest(const float* z, long long tok_stride, long long ch_stride, const float* codebook,
                     const float* cnorm, int P, int NC, int D, float* pmin, int* pidx, int* idx_out, float* zq,
                     prx_stream_t s) {
    return prx_vq_nearest(z, tok_stride, ch_stride, codebook, cnorm, P, NC, D, pmin, pidx, idx_out, zq, S_(s));
}
int prx_k_sqnorm_rows(const float* w, float* out, int rows, int D, prx_stream_t s) {
    return prx_sqnorm_rows(w, out, rows, D, S_(s));
}

// ---- CLIP text tower ------------------------------------------------------------------------
int prx_clip_text_create(prx_clip_text** out, const prx_clip_text_config* c, const float* const* weights, int n_weights,
                         prx_stream_t s) {
    PRX_REQUIRE(out && c && weights, "prx_clip_text_create: null argument");
    return prx_clip_text_create_impl((PrxClipText**)out, c->vocab_size, c->context_length, c->width, c->layers, c->heads,
                                     c->output_dim, c->max_batch, weights, n_weights, S_(s));
}
void prx_clip_text_destroy(prx_clip_text* h) { prx_clip_text_destroy_impl((PrxClipText*)h); }
int prx_clip_text_encode(prx_clip_text* h, const int* tokens, int n, float* embeds, prx_stream_t s) {
    PRX_REQUIRE(h && tokens && embeds, "prx_clip_text_encode: null argument");
    return prx_clip_text_encode_impl((PrxClipText*)h, tokens, n, embeds, S_(s));
}
// ---- CLIP ModifiedResNet tower -----------------------------------------------------------------
int prx_clip_resnet_create(prx_clip_resnet** out, const prx_clip_resnet_config* c, const float* const* weights, int n_weights,
                           prx_stream_t s) {
    PRX_REQUIRE(out && c && weights, "prx_clip_resnet_create: null argument");
    return prx_resnet_create_impl((PrxResNet**)out, c->input_resolution, c->width, c->layers, c->heads, c->output_dim,
                                  c->max_batch, c->precision, weights, n_weights, S_(s));
}
void prx_clip_resnet_destroy(prx_clip_resnet* h) { prx_resnet_destroy_impl((PrxResNet*)h); }
prx_gemm_ctx* prx_clip_resnet_gemm_ctx(prx_clip_resnet* h) { return (prx_gemm_ctx*)prx_resnet_gemm_ctx_impl((PrxResNet*)h); }
int prx_clip_resnet_minmax(prx_clip_resnet* h, const float* cutouts, int n, float* mm, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm, "prx_clip_resnet_minmax: null argument");
    return prx_resnet_minmax_impl((PrxResNet*)h, cutouts, n, mm, S_(s));
}
int prx_clip_resnet_encode(prx_clip_resnet* h, const float* cutouts, int n, const float* mm, float* embeds, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm && embeds, "prx_clip_resnet_encode: null argument");
    return prx_resnet_forward_impl((PrxResNet*)h, cutouts, n, mm, embeds, S_(s));
}
int prx_clip_resnet_backward_reduce(prx_clip_resnet* h, const float* cutouts, const float* mm, const float* d_embeds,
                                    double* acc, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm && d_embeds && acc, "prx_clip_resnet_backward_reduce: null argument");
    return prx_resnet_backward_a_impl((PrxResNet*)h, cutouts, mm, d_embeds, acc, S_(s));
}
int prx_clip_resnet_backward_finish(prx_clip_resnet* h, const float* cutouts, const float* mm, const double* acc,
                                    float* g_cutouts, prx_stream_t s) {
    PRX_REQUIRE(h && cutouts && mm && acc && g_cutouts, "prx_clip_resnet_backward_finish: null argument");
    return prx_resnet_backward_b_impl((PrxResNet*)h, cutouts, mm, acc, g_cutouts, S_(s));
}
}  // extern "C"
