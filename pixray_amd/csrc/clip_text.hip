// CLIP text tower on MI355X (call sites slip.py:68-70, pixray.py:859-877):
//   x = token_embedding[tokens] + positional_embedding                       [n, ctx, width]
//   12 x ResidualAttentionBlock with the causal mask (LN -> MHA -> +res -> LN -> FC -> QuickGELU -> FC -> +res)
//   x = ln_final(x)[i, argmax_j tokens[i, j]] @ text_projection               (the EOT token has the largest id)
// Forward only: prompt embeddings are computed once before the loop and enter it as constants (pixray.py:873-877).
// Same kernels and layout as the image tower (tokens [n*ctx, width] row-major, fp32 residual stream, bf16 GEMM operands).
#include "clip_text.h"
#include "gemm.h"
#include "norms.h"
#include "attention.h"
#include "elementwise.h"
#include "vit.h"  // prx_pack_* helpers
#include <vector>
#include <memory>

namespace {

// one block per sequence: x[row] = tok_emb[token] + pos[j]; eot[i] = first index of the largest token id
__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ emb,
                                                         const float* __restrict__ pos, float* __restrict__ x,
                                                         int* __restrict__ eot, int ctx, int W, int vocab) {
    const int i = blockIdx.x;
    const int* tk = tokens + (size_t)i * ctx;
    if (threadIdx.x == 0) {
        int best = tk[0], bj = 0;
        for (int j = 1; j < ctx; ++j) if (tk[j] > best) { best = tk[j]; bj = j; }
        eot[i] = bj;
    }
    for (int j = 0; j < ctx; ++j) {
        const int t = min(max(tk[j], 0), vocab - 1);      // the host wrapper rejects out-of-range ids; never read out of bounds
        const float* e = emb + (size_t)t * W;
        float* o = x + ((size_t)i * ctx + j) * W;
        for (int c = threadIdx.x; c < W; c += blockDim.x) o[c] = e[c] + pos[(size_t)j * W + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ eot,
                                                          float* __restrict__ out, int ctx, int W) {
    const int i = blockIdx.x;
    const float* src = x + ((size_t)i * ctx + eot[i]) * W;
    for (int c = threadIdx.x; c < W; c += blockDim.x) out[(size_t)i * W + c] = src[c];
}

struct TextLayer {
    float *ln1_g, *ln1_b, *bqkv, *bo, *ln2_g, *ln2_b, *b1, *b2;
    bf16_t *Wqkv, *Wo, *W1, *W2;
};

}  // namespace

struct PrxClipText {
    int vocab, ctx, width, layers, heads, out_dim, max_n;
    std::vector<void*> allocs;
    float *tok, *pos, *lnf_g, *lnf_b;
    bf16_t* projT;
    std::vector<TextLayer> L;
    float *x, *x_mid, *rows, *mean, *rstd, *ws;
    bf16_t *h, *qkv, *att, *u, *hpost;
    int* eot;
    size_t ws_bytes;
};

namespace {
template <typename Tp>
int talloc(PrxClipText* t, Tp** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(Tp)));
    t->allocs.push_back(q);
    *p = (Tp*)q;
    return 0;
}
#define TALLOC(ptr, count) do { int _r = talloc(t, &(ptr), (count)); if (_r) return _r; } while (0)
int tcopy(PrxClipText* t, float** dst, const float* src, size_t n, hipStream_t s) {
    TALLOC(*dst, n);
    PRX_CHECK_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int tpack(PrxClipText* t, bf16_t** W, const float* src, size_t n, hipStream_t s) {
    TALLOC(*W, n);
    return prx_pack_bf16(src, *W, n, s);
}
int tg(PrxClipText* t, GemmDesc& d, hipStream_t s) { return prx_gemm_launch(d, t->ws, t->ws_bytes, s); }
}  // namespace

// weights: token_embedding.weight [vocab, W], positional_embedding [ctx, W], per layer the 12 tensors of a
// ResidualAttentionBlock in state-dict order, ln_final.{weight,bias}, text_projection [W, out_dim]
int prx_clip_text_create_impl(PrxClipText** out, int vocab, int ctx, int width, int layers, int heads, int out_dim, int max_n,
                              const float* const* w, int n_w, hipStream_t s) {
    PRX_REQUIRE(n_w == 2 + 12 * layers + 3, "clip_text_create: expected %d weight tensors, got %d", 2 + 12 * layers + 3, n_w);
    PRX_REQUIRE(width == heads * 64 && width % 256 == 0 && ctx >= 1 && vocab >= 1 && max_n >= 1, "clip_text_create: unsupported geometry");
    PrxClipText* t = new PrxClipText();
    std::unique_ptr<PrxClipText> guard(t);
    t->vocab = vocab; t->ctx = ctx; t->width = width; t->layers = layers; t->heads = heads; t->out_dim = out_dim; t->max_n = max_n;
    const int W = width;
    int r;
    if ((r = tcopy(t, &t->tok, w[0], (size_t)vocab * W, s))) return r;
    if ((r = tcopy(t, &t->pos, w[1], (size_t)ctx * W, s))) return r;
    t->L.resize(layers);
    for (int l = 0; l < layers; ++l) {
        const float* const* q = w + 2 + 12 * l;
        TextLayer& y = t->L[l];
        if ((r = tcopy(t, &y.ln1_g, q[0], W, s))) return r;
        if ((r = tcopy(t, &y.ln1_b, q[1], W, s))) return r;
        if ((r = tpack(t, &y.Wqkv, q[2], (size_t)3 * W * W, s))) return r;
        if ((r = tcopy(t, &y.bqkv, q[3], 3 * W, s))) return r;
        if ((r = tpack(t, &y.Wo, q[4], (size_t)W * W, s))) return r;
        if ((r = tcopy(t, &y.bo, q[5], W, s))) return r;
        if ((r = tcopy(t, &y.ln2_g, q[6], W, s))) return r;
        if ((r = tcopy(t, &y.ln2_b, q[7], W, s))) return r;
        if ((r = tpack(t, &y.W1, q[8], (size_t)4 * W * W, s))) return r;
        if ((r = tcopy(t, &y.b1, q[9], 4 * W, s))) return r;
        if ((r = tpack(t, &y.W2, q[10], (size_t)4 * W * W, s))) return r;
        if ((r = tcopy(t, &y.b2, q[11], W, s))) return r;
    }
    const float* const* q = w + 2 + 12 * layers;
    if ((r = tcopy(t, &t->lnf_g, q[0], W, s))) return r;
    if ((r = tcopy(t, &t->lnf_b, q[1], W, s))) return r;
    TALLOC(t->projT, (size_t)W * out_dim);                       // x @ proj[W, out]: Bt = proj^T [out, W]
    if ((r = prx_pack_transpose_bf16(q[2], t->projT, W, out_dim, s))) return r;
    const size_t R = (size_t)max_n * ctx;
    TALLOC(t->x, R * W); TALLOC(t->x_mid, R * W); TALLOC(t->rows, (size_t)max_n * W); TALLOC(t->mean, R); TALLOC(t->rstd, R);
    TALLOC(t->h, R * W); TALLOC(t->qkv, R * 3 * W); TALLOC(t->att, R * W); TALLOC(t->u, R * 4 * W); TALLOC(t->hpost, (size_t)max_n * W);
    TALLOC(t->eot, max_n);
    t->ws_bytes = (size_t)16 << 20;
    TALLOC(t->ws, t->ws_bytes / sizeof(float));
    *out = guard.release();
    return 0;
}

void prx_clip_text_destroy_impl(PrxClipText* t) {
    if (!t) return;
    for (void* p : t->allocs) (void)hipFree(p);
    delete t;
}

// tokens: int32 [n, ctx] (what clip.tokenize returns, zero padded after the EOT token) -> embeds fp32 [n, out_dim],
// NOT normalised (CLIP_Base.encode_text returns the raw projection, slip.py:68-70; Prompt normalises, pixray.py:276)
int prx_clip_text_encode_impl(PrxClipText* t, const int* tokens, int n, float* embeds, hipStream_t s) {
    PRX_REQUIRE(n >= 1 && n <= t->max_n, "clip_text: batch %d exceeds handle capacity %d", n, t->max_n);
    const int W = t->width, T = t->ctx, R = n * T;
    int r;
    hipLaunchKernelGGL(text_embed_kernel, dim3(n), dim3(256), 0, s, tokens, t->tok, t->pos, t->x, t->eot, T, W, t->vocab);
    PRX_LAUNCH_CHECK();
    for (int l = 0; l < t->layers; ++l) {
        TextLayer& y = t->L[l];
        if ((r = prx_layernorm_fwd(t->x, W, y.ln1_g, y.ln1_b, t->h, nullptr, t->mean, t->rstd, R, W, 1e-5f, s))) return r;
        {   GemmDesc d; d.A = t->h; d.lda = W; d.B = y.Wqkv; d.ldb = W; d.M = R; d.N = 3 * W; d.K = W;
            d.bias_n = y.bqkv; d.out_bf16 = t->qkv; d.ldc_bf16 = 3 * W;
            if ((r = tg(t, d, s))) return r; }
        if ((r = prx_mha_fwd_causal(t->qkv, t->att, n, T, W, t->heads, s))) return r;
        {   GemmDesc d; d.A = t->att; d.lda = W; d.B = y.Wo; d.ldb = W; d.M = R; d.N = W; d.K = W;
            d.bias_n = y.bo; d.resid = t->x; d.ldr = W; d.out_f32 = t->x_mid; d.ldc_f32 = W;
            if ((r = tg(t, d, s))) return r; }
        if ((r = prx_layernorm_fwd(t->x_mid, W, y.ln2_g, y.ln2_b, t->h, nullptr, t->mean, t->rstd, R, W, 1e-5f, s))) return r;
        {   GemmDesc d; d.A = t->h; d.lda = W; d.B = y.W1; d.ldb = W; d.M = R; d.N = 4 * W; d.K = W;
            d.bias_n = y.b1; d.act = PRX_ACT_QUICKGELU; d.out_bf16 = t->u; d.ldc_bf16 = 4 * W;
            if ((r = tg(t, d, s))) return r; }
        {   GemmDesc d; d.A = t->u; d.lda = 4 * W; d.B = y.W2; d.ldb = 4 * W; d.M = R; d.N = W; d.K = 4 * W;
            d.bias_n = y.b2; d.resid = t->x_mid; d.ldr = W; d.out_f32 = t->x; d.ldc_f32 = W;
            if ((r = tg(t, d, s))) return r; }
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, t->x, t->eot, t->rows, T, W);
    PRX_LAUNCH_CHECK();
    if ((r = prx_layernorm_fwd(t->rows, W, t->lnf_g, t->lnf_b, t->hpost, nullptr, t->mean, t->rstd, n, W, 1e-5f, s))) return r;
    GemmDesc d; d.A = t->hpost; d.lda = W; d.B = t->projT; d.ldb = W; d.M = n; d.N = t->out_dim; d.K = W;
    d.out_f32 = embeds; d.ldc_f32 = t->out_dim;
    return tg(t, d, s);
}
