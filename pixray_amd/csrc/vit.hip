// CLIP visual tower runner (clip.model.VisionTransformer [UPSTREAM openai/CLIP], as called by
// CLIP_Base.encode_image, slip.py:62-66): forward and activation-gradient backward on the
// MFMA engine.  Residual stream and LayerNorm statistics stay fp32; every GEMM operand
// (LN outputs, qkv, attention output, MLP hidden) is bf16; weights are packed once to bf16 in
// both orientations (forward Bt = W[out,in], dgrad Bt = W^T[in,out]) because they are frozen
// (slip.py:176).  With precision == PRX_PREC_F32 the same operand buffers hold fp32, the GEMMs run on
// v_mfma_f32_32x32x2_f32 and attention on the fp32 kernels of attention_f32.hip: the exact parity mode.
#include "vit.h"
#include <stdlib.h>
#include "gemm.h"
#include "norms.h"
#include "attention.h"
#include "cutouts.h"
#include "prompt_vq.h"
#include "elementwise.h"
#include <vector>

namespace {

template <typename TOp>
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float* __restrict__ in, TOp* __restrict__ out,
                                                             int R, int C) {
    // out[c][r] = TOp(in[r][c])
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * R + r] = op_cvt<TOp>(tile[tx][ty + 8 * i]);
    }
}

__global__ __launch_bounds__(256) void add_cls_pos_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                          const float* __restrict__ pos, int N, int T, int W) {
    const size_t total = (size_t)N * T * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const int t = (int)((i / W) % T);
        float v = x[i] + pos[(size_t)t * W + c];
        if (t == 0) v += cls[c];
        x[i] = v;
    }
}

}  // namespace

int prx_pack_op(const float* in, void* out, size_t n, int prec, hipStream_t s) {
    if (!prec_is_f32(prec)) return prx_f32_to_bf16(in, (bf16_t*)out, n, s, prec_is_h16(prec));
    PRX_CHECK_HIP(hipMemcpyAsync(out, in, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int prx_pack_transpose_op(const float* in, void* out, int R, int C, int prec, hipStream_t s) {
    const dim3 grid(ceil_div(C, 32), ceil_div(R, 32));
    PRX_OP_DISPATCH(prec_is_f32(prec), prec_is_h16(prec), TO,
                    hipLaunchKernelGGL(pack_transpose_kernel<TO>, grid, dim3(256), 0, s, in, (TO*)out, R, C));
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_pack_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) { return prx_pack_op(in, out, n, 0, s); }
int prx_pack_transpose_bf16(const float* in, bf16_t* out, int R, int C, hipStream_t s) { return prx_pack_transpose_op(in, out, R, C, 0, s); }

struct VitLayer {
    float *ln1_g, *ln1_b, *bqkv, *bo, *ln2_g, *ln2_b, *b1, *b2;
    void *Wqkv, *WqkvT, *Wo, *WoT, *W1, *W1T, *W2, *W2T;   // operand precision (bf16 | fp32)
    // saved activations (x_in / x_mid: the residual stream, fp32 -- or the 16-bit operand format in the lean layout)
    void *x_in, *x_mid;
    float *mean1, *rstd1, *mean2, *rstd2;
    void *qkv, *t;    // operand precision
    void* o_save;     // attention output, kept for the general-T backward (T > 64) and the fp32 attention
    float* lse;
};

struct PrxVit {
    int res, patch, width, layers, heads, out_dim, T, max_n, KP;
    int prec;         // PRX_PREC_*: element type of every operand buffer below (void*)
    int f32, h16;     // derived: operands are fp32 / the 16-bit operand format is IEEE half
    float* gs;        // half mode: device {S, 1/S} = the power-of-two scale of the backward in flight (common.h) + 64 partials; else null
    GemmCtx gctx;     // this handle's engine state (tile overrides, timing log)
    std::vector<void*> allocs;
    void *Wp, *WpT, *projT, *proj;
    float *cls, *pos, *lnpre_g, *lnpre_b, *lnpost_g, *lnpost_b;
    std::vector<VitLayer> L;
    // workspace
    void *A0, *h, *att_o, *u, *hpost, *dt, *do_, *dqkv, *dx_bf, *dh_bf, *de16;   // operand precision
    float *xpre, *mean_pre, *rstd_pre, *mean_post, *rstd_post, *e, *dx, *dh, *dA0, *de, *dhpost, *mm_part;
    void* x_final;
    // The lean layout (half mode, PRX_LEAN): the residual stream and its gradient live in IEEE half only -- the reference's own
    // GPU arithmetic for this tower (slip.py:175: the CLIP model runs in fp16, residual adds included).  One 16-bit tensor is
    // the saved activation, the LayerNorm input and the residual operand of the next product; dx / dh have no fp32 copy.
    int lean;
    int cls_tail;      // 1 (default): the class-token tail below; PRX_VIT_CLS_TAIL=0 runs the last block on every token row (A/B, bisection)
    float* ws; size_t ws_bytes;
    int cur_n;
    // The class-token tail: only the class token of the LAST block's output is ever read (ln_post on token 0, slip.py:66 / clip
    // VisionTransformer), so that block's out-projection, MLP and their backward run on the n class-token rows (row stride
    // T * width into the same buffers) instead of all n * T token rows: the same values for everything that is read, 3 + 3 of
    // the tower's 48 + 48 wide products reduced to M = n.  K and V of that block still need every token.  (Round-5 A/B on the
    // device: 140.1 -> 141.2 and 138.0 -> 139.5 it/s, profiles/r05_first_call/, r05_timeline/.)
};

namespace {
template <typename Tp>
int dev_alloc(PrxVit* v, Tp** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, count * sizeof(Tp)));
    v->allocs.push_back(q);
    *p = (Tp*)q;
    return 0;
}
#define ALLOC(ptr, count) do { int _r = dev_alloc(v, &(ptr), (count)); if (_r) return _r; } while (0)
int dev_alloc_op(PrxVit* v, void** p, size_t count) {   // `count` operand elements
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, count * op_esz(v->f32)));
    v->allocs.push_back(q);
    *p = q;
    return 0;
}
#define ALLOC_OP(ptr, count) do { int _r = dev_alloc_op(v, &(ptr), (count)); if (_r) return _r; } while (0)

int copy_f32(PrxVit* v, float** dst, const float* src, size_t n, hipStream_t s) {
    ALLOC(*dst, n);
    PRX_CHECK_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int pack_both(PrxVit* v, void** W, void** WT, const float* src, int out, int in, hipStream_t s) {
    ALLOC_OP(*W, (size_t)out * in);
    ALLOC_OP(*WT, (size_t)out * in);
    int r = prx_pack_op(src, *W, (size_t)out * in, v->prec, s);
    if (r) return r;
    return prx_pack_transpose_op(src, *WT, out, in, v->prec, s);
}
}  // namespace

int prx_vit_create_impl(PrxVit** out, int res, int patch, int width, int layers, int heads, int out_dim, int max_n,
                        int precision, const float* const* w, int n_w, hipStream_t s) {
    PRX_REQUIRE(prec_valid(precision), "vit_create: unknown precision %d", precision);
    PRX_REQUIRE(n_w == 5 + 12 * layers + 3, "vit_create: expected %d weight tensors, got %d", 5 + 12 * layers + 3, n_w);
    PRX_REQUIRE(res % patch == 0 && width == heads * 64 && width % 256 == 0, "vit_create: unsupported geometry");
    const int G = res / patch;
    const int T = G * G + 1;
    PrxVit* v = new PrxVit();
    v->res = res; v->patch = patch; v->width = width; v->layers = layers; v->heads = heads; v->out_dim = out_dim;
    v->prec = precision; v->f32 = prec_is_f32(precision); v->h16 = prec_is_h16(precision);
    v->gs = nullptr;
    { const char* e = getenv("PRX_LEAN"); v->lean = (v->h16 && !(e && atoi(e) == 0)) ? 1 : 0; }
    { const char* e = getenv("PRX_VIT_CLS_TAIL"); v->cls_tail = (e && atoi(e) == 0) ? 0 : 1; }
    v->T = T; v->max_n = max_n; v->KP = (3 * patch * patch + 7) / 8 * 8; v->cur_n = 0;   // K padded to x8 (L/14: 588 -> 592)
    const int W = width, KP = v->KP;
    int r;
    {   // conv1.weight [W, 3*P*P] -> zero-padded to KP columns, both orientations
        const int K0 = 3 * patch * patch;
        float* padded;
        ALLOC(padded, (size_t)W * KP);
        PRX_CHECK_HIP(hipMemsetAsync(padded, 0, sizeof(float) * (size_t)W * KP, s));
        PRX_CHECK_HIP(hipMemcpy2DAsync(padded, sizeof(float) * KP, w[0], sizeof(float) * K0, sizeof(float) * K0, W,
                                       hipMemcpyDeviceToDevice, s));
        if ((r = pack_both(v, &v->Wp, &v->WpT, padded, W, KP, s))) return r;
    }
    if ((r = copy_f32(v, &v->cls, w[1], W, s))) return r;
    if ((r = copy_f32(v, &v->pos, w[2], (size_t)T * W, s))) return r;
    if ((r = copy_f32(v, &v->lnpre_g, w[3], W, s))) return r;
    if ((r = copy_f32(v, &v->lnpre_b, w[4], W, s))) return r;
    v->L.resize(layers);
    const size_t R = (size_t)max_n * T;
    for (int l = 0; l < layers; ++l) {
        const float* const* q = w + 5 + 12 * l;
        VitLayer& y = v->L[l];
        if ((r = copy_f32(v, &y.ln1_g, q[0], W, s))) return r;
        if ((r = copy_f32(v, &y.ln1_b, q[1], W, s))) return r;
        if ((r = pack_both(v, &y.Wqkv, &y.WqkvT, q[2], 3 * W, W, s))) return r;
        if ((r = copy_f32(v, &y.bqkv, q[3], 3 * W, s))) return r;
        if ((r = pack_both(v, &y.Wo, &y.WoT, q[4], W, W, s))) return r;
        if ((r = copy_f32(v, &y.bo, q[5], W, s))) return r;
        if ((r = copy_f32(v, &y.ln2_g, q[6], W, s))) return r;
        if ((r = copy_f32(v, &y.ln2_b, q[7], W, s))) return r;
        if ((r = pack_both(v, &y.W1, &y.W1T, q[8], 4 * W, W, s))) return r;
        if ((r = copy_f32(v, &y.b1, q[9], 4 * W, s))) return r;
        if ((r = pack_both(v, &y.W2, &y.W2T, q[10], W, 4 * W, s))) return r;
        if ((r = copy_f32(v, &y.b2, q[11], W, s))) return r;
        if (v->lean) { ALLOC_OP(y.x_in, R * W); ALLOC_OP(y.x_mid, R * W); }
        else { float *a_, *b_; ALLOC(a_, R * W); ALLOC(b_, R * W); y.x_in = a_; y.x_mid = b_; }
        ALLOC(y.mean1, R); ALLOC(y.rstd1, R); ALLOC(y.mean2, R); ALLOC(y.rstd2, R);
        ALLOC_OP(y.qkv, R * 3 * W); ALLOC_OP(y.t, R * 4 * W);
        y.o_save = nullptr; y.lse = nullptr;
        if (T > 64 || v->f32) { ALLOC_OP(y.o_save, R * W); ALLOC(y.lse, (size_t)max_n * heads * T); }
    }
    const float* const* q = w + 5 + 12 * layers;
    if ((r = copy_f32(v, &v->lnpost_g, q[0], W, s))) return r;
    if ((r = copy_f32(v, &v->lnpost_b, q[1], W, s))) return r;
    // proj is [width, out]: forward Bt = proj^T [out, width]; dgrad Bt = proj [width, out]
    if ((r = pack_both(v, &v->proj, &v->projT, q[2], W, out_dim, s))) return r;
    ALLOC_OP(v->A0, R * KP); ALLOC_OP(v->h, R * W); ALLOC_OP(v->att_o, R * W); ALLOC_OP(v->u, R * 4 * W);
    ALLOC_OP(v->hpost, (size_t)max_n * W); ALLOC_OP(v->de16, (size_t)max_n * out_dim); ALLOC_OP(v->dt, R * 4 * W); ALLOC_OP(v->do_, R * W); ALLOC_OP(v->dqkv, R * 3 * W);
    ALLOC(v->xpre, R * W); ALLOC(v->mean_pre, R); ALLOC(v->rstd_pre, R);
    if (v->lean) ALLOC_OP(v->x_final, R * W);
    else { float* a_; ALLOC(a_, R * W); v->x_final = a_; }
    ALLOC(v->mean_post, max_n); ALLOC(v->rstd_post, max_n); ALLOC(v->e, (size_t)max_n * out_dim);
    v->dx = v->dh = nullptr;
    if (!v->lean) { ALLOC(v->dx, R * W); ALLOC(v->dh, R * W); }
    ALLOC(v->dA0, R * KP); ALLOC(v->de, (size_t)max_n * out_dim);
    // bf16 twins of the fp32 gradient streams (the dgrad GEMMs' A operands); the exact mode reads the fp32 streams themselves;
    // in the lean layout they ARE the gradient streams
    if (v->f32) { v->dx_bf = v->dx; v->dh_bf = v->dh; }
    else { ALLOC_OP(v->dx_bf, R * W); ALLOC_OP(v->dh_bf, R * W); }
    ALLOC(v->dhpost, (size_t)max_n * W); ALLOC(v->mm_part, 2 * 1024);
    if (v->h16) ALLOC(v->gs, 2 + 64);
    v->ws_bytes = (size_t)64 << 20;
    ALLOC(v->ws, v->ws_bytes / sizeof(float));
    *out = v;
    return 0;
}

void prx_vit_destroy_impl(PrxVit* v) {
    if (!v) return;
    for (void* p : v->allocs) (void)hipFree(p);
    delete v;
}

static int vit_gemm(PrxVit* v, GemmDesc& d, hipStream_t s) {
    if (v->f32) { d.f32 = 1; d.a_is_f32 = 0; }
    d.h16 = v->h16;
    return prx_gemm_launch(d, v->ws, v->ws_bytes, s, &v->gctx);
}
GemmCtx* prx_vit_gemm_ctx_impl(PrxVit* v) { return v ? &v->gctx : nullptr; }

// LayerNorm whose output is a GEMM operand: bf16, or fp32 in the exact mode (the kernel has both outputs)
static int ln_op(PrxVit* v, const void* x, long long ldx, const float* g, const float* b, void* out, float* mean, float* rstd,
                 int rows, hipStream_t s) {
    return prx_layernorm_fwd(x, ldx, g, b, v->f32 ? nullptr : (bf16_t*)out, v->f32 ? (float*)out : nullptr, mean, rstd, rows,
                             v->width, 1e-5f, s, v->h16, v->lean);
}
// LayerNorm backward producing the fp32 gradient stream + its operand twin (the same buffer in the exact mode)
// `s16`: which of x (1), g (2), add (4) are 16-bit streams (the lean layout; dx is then null and dx_op the only output)
static int ln_bwd_op(PrxVit* v, const void* g, long long ldg, const void* x, long long ldx, const float* gamma, const float* mean,
                     const float* rstd, const void* add, long long ldadd, float* dx, long long lddx, void* dx_op, int rows,
                     hipStream_t s, int add_every = 0, int s16 = 0) {
    return prx_layernorm_bwd(g, ldg, x, ldx, gamma, mean, rstd, add, ldadd, dx, lddx, v->f32 ? nullptr : (bf16_t*)dx_op, lddx, rows,
                             v->width, s, v->h16, add_every, s16);
}

int prx_vit_minmax_impl(PrxVit* v, const float* cutouts, int n, float* mm, hipStream_t s) {
    PRX_REQUIRE(n >= 1 && n <= v->max_n, "vit: batch %d exceeds handle capacity %d", n, v->max_n);
    return prx_minmax(cutouts, (size_t)n * 3 * v->res * v->res, v->mm_part, 1024, mm, s);
}

int prx_vit_forward_impl(PrxVit* v, const float* cutouts, int n, const float* mm, float* embeds, hipStream_t s) {
    PRX_REQUIRE(n >= 1 && n <= v->max_n, "vit: batch %d exceeds handle capacity %d", n, v->max_n);
    const int W = v->width, T = v->T, R = n * T, KP = v->KP;
    int r;
    v->cur_n = n;
    if ((r = prx_patchify_fwd(cutouts, mm, v->A0, v->prec, n, v->res, v->patch, T, s))) return r;
    {   // conv1 (patch embed) as GEMM
        GemmDesc d; d.A = v->A0; d.lda = KP; d.B = v->Wp; d.ldb = KP; d.M = R; d.N = W; d.K = KP;
        d.out_f32 = v->xpre; d.ldc_f32 = W;
        if ((r = vit_gemm(v, d, s))) return r;
    }
    hipLaunchKernelGGL(add_cls_pos_kernel, dim3(2048), dim3(256), 0, s, v->xpre, v->cls, v->pos, n, T, W);
    PRX_LAUNCH_CHECK();
    const int lean = v->lean;
    void* x0 = v->layers > 0 ? v->L[0].x_in : v->x_final;
    if ((r = prx_layernorm_fwd(v->xpre, W, v->lnpre_g, v->lnpre_b, lean ? (bf16_t*)x0 : nullptr, lean ? nullptr : (float*)x0, v->mean_pre,
                               v->rstd_pre, R, W, 1e-5f, s, v->h16))) return r;
    for (int l = 0; l < v->layers; ++l) {
        VitLayer& y = v->L[l];
        void* x_next = (l + 1 < v->layers) ? v->L[l + 1].x_in : v->x_final;
        if ((r = ln_op(v, y.x_in, W, y.ln1_g, y.ln1_b, v->h, y.mean1, y.rstd1, R, s))) return r;
        {   GemmDesc d; d.A = v->h; d.lda = W; d.B = y.Wqkv; d.ldb = W; d.M = R; d.N = 3 * W; d.K = W;
            d.bias_n = y.bqkv; d.out_bf16 = y.qkv; d.ldc_bf16 = 3 * W;
            if ((r = vit_gemm(v, d, s))) return r; }
        const void* att = v->att_o;
        if (v->f32) { if ((r = prx_mha_fwd_f32((const float*)y.qkv, (float*)y.o_save, y.lse, n, T, W, v->heads, s))) return r; att = y.o_save; }
        else if (T <= 64) { if ((r = prx_mha_fwd((const bf16_t*)y.qkv, (bf16_t*)v->att_o, n, T, W, v->heads, s, v->h16))) return r; }
        else { if ((r = prx_mha_fwd_gen((const bf16_t*)y.qkv, (bf16_t*)y.o_save, y.lse, n, T, W, v->heads, s, v->h16))) return r; att = y.o_save; }
        // the class-token tail: rows = the n class tokens, reached through a row stride of T * W in the token-major buffers;
        // LN / MLP intermediates of those rows are stored densely ([n][...]) at the start of their buffers
        const bool tail = v->cls_tail && l == v->layers - 1;
        const int rows = tail ? n : R;
        const int ldt = tail ? T * W : W;            // row stride of the token-major fp32 / 16-bit [R, W] buffers
        {   GemmDesc d; d.A = att; d.lda = ldt; d.B = y.Wo; d.ldb = W; d.M = rows; d.N = W; d.K = W;
            d.bias_n = y.bo; d.ldr = ldt;
            if (lean) { d.resid = (const float*)y.x_in; d.row16 = 1; d.out_bf16 = y.x_mid; d.ldc_bf16 = ldt; }
            else { d.resid = (const float*)y.x_in; d.out_f32 = (float*)y.x_mid; d.ldc_f32 = ldt; }
            if ((r = vit_gemm(v, d, s))) return r; }
        if ((r = ln_op(v, y.x_mid, ldt, y.ln2_g, y.ln2_b, v->h, y.mean2, y.rstd2, rows, s))) return r;
        {   GemmDesc d; d.A = v->h; d.lda = W; d.B = y.W1; d.ldb = W; d.M = rows; d.N = 4 * W; d.K = W;
            d.bias_n = y.b1; d.act = PRX_ACT_QUICKGELU; d.out_bf16 = v->u; d.out_bf16_pre = y.t; d.ldc_bf16 = 4 * W;
            if ((r = vit_gemm(v, d, s))) return r; }
        {   GemmDesc d; d.A = v->u; d.lda = 4 * W; d.B = y.W2; d.ldb = 4 * W; d.M = rows; d.N = W; d.K = 4 * W;
            d.bias_n = y.b2; d.ldr = ldt;
            if (lean) { d.resid = (const float*)y.x_mid; d.row16 = 1; d.out_bf16 = x_next; d.ldc_bf16 = ldt; }
            else { d.resid = (const float*)y.x_mid; d.out_f32 = (float*)x_next; d.ldc_f32 = ldt; }
            if ((r = vit_gemm(v, d, s))) return r; }
    }
    // ln_post on the class token, projection, L2 normalisation (slip.py:66)
    if ((r = ln_op(v, v->x_final, (long long)T * W, v->lnpost_g, v->lnpost_b, v->hpost, v->mean_post, v->rstd_post, n, s))) return r;
    {   GemmDesc d; d.A = v->hpost; d.lda = W; d.B = v->projT; d.ldb = W; d.M = n; d.N = v->out_dim; d.K = W;
        d.out_f32 = v->e; d.ldc_f32 = v->out_dim;
        if ((r = vit_gemm(v, d, s))) return r; }
    return prx_l2norm_fwd(v->e, embeds, n, v->out_dim, s);
}

// Backward part A: from d(embeds) down to the patch-embed input gradient and the reduction the
// batch-global min/max renorm needs (acc[4] doubles, to be summed over ranks when the cutout
// batch is sharded).
int prx_vit_backward_a_impl(PrxVit* v, const float* cutouts, const float* mm, const float* d_embeds, double* acc,
                            hipStream_t s) {
    const int n = v->cur_n;
    PRX_REQUIRE(n >= 1, "vit backward: no forward in flight on this handle");
    const int W = v->width, T = v->T, R = n * T, KP = v->KP;
    int r;
    if ((r = prx_l2norm_bwd(v->e, d_embeds, v->de, n, v->out_dim, s))) return r;
    {   GemmDesc d; d.A = v->de; d.a_is_f32 = 1; d.lda = v->out_dim; d.B = v->proj; d.ldb = v->out_dim;
        d.M = n; d.N = W; d.K = v->out_dim; d.out_f32 = v->dhpost; d.ldc_f32 = W;
        // half mode: everything below runs scaled by a power of two S chosen from max|d e| (exact: the backward is linear in g).
        // S multiplies d e in fp32, BEFORE the GEMM's load converts it to half: entries of d e are ~1e-5 at the headline and
        // shrink with the prompt weight and the world size -- unscaled they would land in half's subnormals or flush to zero
        if (v->h16) {
            if ((r = prx_grad_scale(v->de, (size_t)n * v->out_dim, v->gs + 2, 64, prx_grad_target_log2(), v->gs, s))) return r;
            // ... and leaves as the half operand of this product too (hpost is free by now): a 16-bit A puts the M = n product on a
            // fit tile with K groups (6 us) instead of the register-staged 128 x 64 kernel on 12 workgroups (26 us)
            if ((r = prx_scale_dev(v->de, (size_t)n * v->out_dim, v->gs, s, (bf16_t*)v->de16, v->h16))) return r;
            d.A = v->de16; d.a_is_f32 = 0;
        }
        if ((r = vit_gemm(v, d, s))) return r; }
    // the residual-stream gradient is kept in fp32 (dx) with a bf16 twin (dx_bf) that feeds the dgrad GEMMs.  It enters the last
    // block on the class-token rows only; that block's kernels read those rows alone (the class-token tail) until its ln_1
    // backward, which takes the incoming gradient as zero on every other row (add_every = T) and writes all of them -- so the
    // streams need no clearing (two fills of 14.7 MB per iteration at the headline)
    const int lean = v->lean;
    if (v->layers == 0 || !v->cls_tail) PRX_CHECK_HIP(hipMemsetAsync(lean ? v->dx_bf : (void*)v->dx, 0, (lean ? sizeof(bf16_t) : sizeof(float)) * (size_t)R * W, s));
    // lean layout: dxs / dhs are the 16-bit gradient streams themselves (no fp32 copies exist)
    const void* dxs = lean ? v->dx_bf : (const void*)v->dx;
    const void* dhs = lean ? v->dh_bf : (const void*)v->dh;
    if ((r = ln_bwd_op(v, v->dhpost, W, v->x_final, (long long)T * W, v->lnpost_g, v->mean_post, v->rstd_post,
                       nullptr, 0, v->dx, (long long)T * W, v->dx_bf, n, s, 0, lean ? 1 : 0))) return r;
    for (int l = v->layers - 1; l >= 0; --l) {
        VitLayer& y = v->L[l];
        // the class-token tail (see the forward): the gradient entering the last block is non-zero on the class-token rows only
        const bool tail = v->cls_tail && l == v->layers - 1;
        const int rows = tail ? n : R;
        const int ldt = tail ? T * W : W;
        // MLP: x_next = x_mid + c_proj(quickgelu(c_fc(ln_2(x_mid))))
        {   GemmDesc d; d.A = v->dx_bf; d.lda = ldt; d.B = y.W2T; d.ldb = W; d.M = rows; d.N = 4 * W; d.K = W;
            d.act = PRX_ACT_MUL_DQUICKGELU; d.aux = y.t; d.ldaux = 4 * W; d.out_bf16 = v->dt; d.ldc_bf16 = 4 * W;
            if ((r = vit_gemm(v, d, s))) return r; }
        {   GemmDesc d; d.A = v->dt; d.lda = 4 * W; d.B = y.W1T; d.ldb = 4 * W; d.M = rows; d.N = W; d.K = 4 * W;
            if (lean) { d.out_bf16 = v->dh_bf; d.ldc_bf16 = W; } else { d.out_f32 = v->dh; d.ldc_f32 = W; }
            if ((r = vit_gemm(v, d, s))) return r; }
        if ((r = ln_bwd_op(v, dhs, W, y.x_mid, ldt, y.ln2_g, y.mean2, y.rstd2, dxs, ldt, v->dx, ldt, v->dx_bf, rows, s, 0, lean ? 7 : 0))) return r;
        // attention: x_mid = x_in + out_proj(mha(ln_1(x_in)))
        if (tail)       // d(attention output) is written on the class-token rows only: the other rows must read as zero
            PRX_CHECK_HIP(hipMemsetAsync(v->do_, 0, op_esz(v->f32) * (size_t)R * W, s));
        {   GemmDesc d; d.A = v->dx_bf; d.lda = ldt; d.B = y.WoT; d.ldb = W; d.M = rows; d.N = W; d.K = W;
            d.out_bf16 = v->do_; d.ldc_bf16 = ldt;
            if ((r = vit_gemm(v, d, s))) return r; }
        if (v->f32) { if ((r = prx_mha_bwd_f32((const float*)y.qkv, (const float*)y.o_save, (const float*)v->do_, y.lse, (float*)v->dqkv, n, T, W, v->heads, s))) return r; }
        else if (T <= 64) { if ((r = prx_mha_bwd((const bf16_t*)y.qkv, (const bf16_t*)v->do_, (bf16_t*)v->dqkv, n, T, W, v->heads, s, v->h16))) return r; }
        else { if ((r = prx_mha_bwd_gen((const bf16_t*)y.qkv, (const bf16_t*)y.o_save, (const bf16_t*)v->do_, y.lse, (bf16_t*)v->dqkv, n, T, W, v->heads, s, v->h16))) return r; }
        {   GemmDesc d; d.A = v->dqkv; d.lda = 3 * W; d.B = y.WqkvT; d.ldb = 3 * W; d.M = R; d.N = W; d.K = 3 * W;
            if (lean) { d.out_bf16 = v->dh_bf; d.ldc_bf16 = W; } else { d.out_f32 = v->dh; d.ldc_f32 = W; }
            if ((r = vit_gemm(v, d, s))) return r; }
        if ((r = ln_bwd_op(v, dhs, W, y.x_in, W, y.ln1_g, y.mean1, y.rstd1, dxs, W, v->dx, W, v->dx_bf, R, s, tail ? T : 0, lean ? 7 : 0))) return r;
    }
    // ln_pre backward (in place on dx), then patch-embed dgrad
    if ((r = ln_bwd_op(v, dxs, W, v->xpre, W, v->lnpre_g, v->mean_pre, v->rstd_pre, nullptr, 0, v->dh, W, v->dh_bf, R, s, 0, lean ? 2 : 0))) return r;
    {   GemmDesc d; d.A = v->dh_bf; d.lda = W; d.B = v->WpT; d.ldb = W; d.M = R; d.N = KP; d.K = W;
        d.out_f32 = v->dA0; d.ldc_f32 = KP;
        if (v->h16) d.alpha_dev = v->gs + 1;      // ... and is unscaled (1/S) here, before the (rank-summed) renormalisation sums
        if ((r = vit_gemm(v, d, s))) return r; }
    return prx_patchify_bwd_reduce(cutouts, mm, v->dA0, acc, n, v->res, v->patch, T, s);
}

int prx_vit_backward_b_impl(PrxVit* v, const float* cutouts, const float* mm, const double* acc, float* g_cutouts,
                            hipStream_t s) {
    const int n = v->cur_n;
    PRX_REQUIRE(n >= 1, "vit backward: no forward in flight on this handle");
    int r = prx_patchify_bwd_apply(cutouts, mm, v->dA0, acc, g_cutouts, n, v->res, v->patch, v->T, s);
    return r;
}
