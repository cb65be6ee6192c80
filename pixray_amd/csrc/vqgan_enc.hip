// taming VQModel.encode on MI355X (call sites vqgan.py:174-185):
//   Encoder [UPSTREAM taming/modules/diffusionmodules/model.py]: conv_in, per level {ResnetBlock x n (+AttnBlock at the
//   attention resolution), Downsample = pad (0,1,0,1) + 3x3 stride-2 conv}, mid (Res, Attn, Res), GroupNorm+swish,
//   conv_out; then quant_conv (1x1) and the nearest-code lookup of VectorQuantizer2 (the returned latent IS the code
//   vector: z + (z_q - z).detach()).
// Forward only (the reference never differentiates through encode(): the result becomes the leaf z, vqgan.py:175-176).
// Same data layout and kernels as the decoder runner: NHWC fp32 residual stream, GroupNorm(+swish) writes the bf16
// GEMM operand, every conv is an implicit GEMM on the MFMA engine (the stride-2 convs through the engine's up==2 gather).
#include "vqgan_enc.h"
#include "gemm.h"
#include "norms.h"
#include "elementwise.h"
#include "prompt_vq.h"
#include "vit.h"  // prx_pack_* helpers
#include <vector>
#include <memory>
#include <algorithm>
#include <math.h>

namespace {

// Wf[co][tap*CiP + ci] = w[co][ci][ky][kx], ci padded with zeros to CiP (conv_in: 3 -> 8 input channels)
__global__ __launch_bounds__(256) void pack_conv3x3_fwd_kernel(const float* __restrict__ w, bf16_t* __restrict__ Wf, int Cout,
                                                               int Cin, int CiP) {
    const size_t total = (size_t)Cout * 9 * CiP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CiP);
        const int tap = (int)((i / CiP) % 9);
        const int co = (int)(i / ((size_t)9 * CiP));
        Wf[i] = ci < Cin ? (bf16_t)w[(((size_t)co * Cin + ci) * 3 + tap / 3) * 3 + tap % 3] : (bf16_t)0.f;
    }
}

struct EConv3 { int Cin, CiP, Cout; bf16_t* W; float* b; };
struct EConv1 { int Cin, Cout; bf16_t* W; float* b; };
struct EGN { int C; float *g, *b; };
struct ERes { int Cin, Cout; EGN n1, n2; EConv3 c1, c2; EConv1 sc; bool has_sc; };
struct EAttn { int C; EGN n; EConv1 qkv, proj; };
struct EOp { int kind; int idx; int H, W; };   // 0 res, 1 attn, 2 downsample conv (H, W = output size)

}  // namespace

struct PrxVqganEnc {
    int in_ch, zc, D, NC, H, W, h0, w0;
    std::vector<void*> allocs;
    float *codebook, *cnorm;
    EConv3 conv_in, conv_out; EConv1 quant; EGN norm_out;
    std::vector<ERes> res; std::vector<EAttn> attn; std::vector<EConv3> downs; std::vector<EOp> ops;
    bf16_t *img8, *a, *xb0, *xb1, *qkvb, *tA, *tB, *Pm, *PT, *co_bf;
    float *x0, *x1, *h1, *sc, *S, *hq, *zq, *ws, *pmin;
    int *pidx, *idx;
    double* stats;
    size_t ws_bytes;
};

namespace {
template <typename Tp>
int ealloc(PrxVqganEnc* e, Tp** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(Tp)));
    e->allocs.push_back(q);
    *p = (Tp*)q;
    return 0;
}
#define EALLOC(ptr, count) do { int _r = ealloc(e, &(ptr), (count)); if (_r) return _r; } while (0)
struct ECursor { const float* const* w; int n, pos; };
#define ENEXT(cur, dst) do { PRX_REQUIRE((cur).pos < (cur).n, "vqgan_enc_create: weight list too short"); (dst) = (cur).w[(cur).pos++]; } while (0)

int ecopy(PrxVqganEnc* e, float** dst, const float* src, size_t n, hipStream_t s) {
    EALLOC(*dst, n);
    PRX_CHECK_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int e_gn(PrxVqganEnc* e, EGN& g, int C, ECursor& cur, hipStream_t s) {
    const float *w, *b; ENEXT(cur, w); ENEXT(cur, b);
    g.C = C;
    int r;
    if ((r = ecopy(e, &g.g, w, C, s))) return r;
    return ecopy(e, &g.b, b, C, s);
}
int e_conv3(PrxVqganEnc* e, EConv3& c, int Cin, int Cout, ECursor& cur, hipStream_t s) {
    const float *w, *b; ENEXT(cur, w); ENEXT(cur, b);
    c.Cin = Cin; c.CiP = (Cin + 7) / 8 * 8; c.Cout = Cout;
    EALLOC(c.W, (size_t)Cout * 9 * c.CiP);
    hipLaunchKernelGGL(pack_conv3x3_fwd_kernel, dim3(512), dim3(256), 0, s, w, c.W, Cout, Cin, c.CiP);
    PRX_LAUNCH_CHECK();
    return ecopy(e, &c.b, b, Cout, s);
}
int e_conv1(PrxVqganEnc* e, EConv1& c, int Cin, int Cout, ECursor& cur, hipStream_t s) {
    const float *w, *b; ENEXT(cur, w); ENEXT(cur, b);
    c.Cin = Cin; c.Cout = Cout;
    EALLOC(c.W, (size_t)Cout * Cin);
    int r;
    if ((r = prx_pack_bf16(w, c.W, (size_t)Cout * Cin, s))) return r;
    return ecopy(e, &c.b, b, Cout, s);
}
int e_res(PrxVqganEnc* e, int Cin, int Cout, int H, int W, ECursor& cur, hipStream_t s) {
    ERes rb{};
    rb.Cin = Cin; rb.Cout = Cout; rb.has_sc = Cin != Cout;
    int r;
    if ((r = e_gn(e, rb.n1, Cin, cur, s))) return r;
    if ((r = e_conv3(e, rb.c1, Cin, Cout, cur, s))) return r;
    if ((r = e_gn(e, rb.n2, Cout, cur, s))) return r;
    if ((r = e_conv3(e, rb.c2, Cout, Cout, cur, s))) return r;
    if (rb.has_sc && (r = e_conv1(e, rb.sc, Cin, Cout, cur, s))) return r;
    e->ops.push_back({0, (int)e->res.size(), H, W});
    e->res.push_back(rb);
    return 0;
}
int e_attn(PrxVqganEnc* e, int C, int H, int W, ECursor& cur, hipStream_t s) {
    EAttn ab{};
    ab.C = C;
    int r;
    if ((r = e_gn(e, ab.n, C, cur, s))) return r;
    const float *wq, *bq, *wk, *bk, *wv, *bv;
    ENEXT(cur, wq); ENEXT(cur, bq); ENEXT(cur, wk); ENEXT(cur, bk); ENEXT(cur, wv); ENEXT(cur, bv);
    float *wcat, *bcat;
    EALLOC(wcat, (size_t)3 * C * C); EALLOC(bcat, 3 * C);
    const float* ws_[3] = {wq, wk, wv}; const float* bs_[3] = {bq, bk, bv};
    for (int i = 0; i < 3; ++i) {
        PRX_CHECK_HIP(hipMemcpyAsync(wcat + (size_t)i * C * C, ws_[i], sizeof(float) * C * C, hipMemcpyDeviceToDevice, s));
        PRX_CHECK_HIP(hipMemcpyAsync(bcat + i * C, bs_[i], sizeof(float) * C, hipMemcpyDeviceToDevice, s));
    }
    ab.qkv.Cin = C; ab.qkv.Cout = 3 * C; ab.qkv.b = bcat;
    EALLOC(ab.qkv.W, (size_t)3 * C * C);
    if ((r = prx_pack_bf16(wcat, ab.qkv.W, (size_t)3 * C * C, s))) return r;
    if ((r = e_conv1(e, ab.proj, C, C, cur, s))) return r;
    e->ops.push_back({1, (int)e->attn.size(), H, W});
    e->attn.push_back(ab);
    return 0;
}

int eg(PrxVqganEnc* e, GemmDesc& d, hipStream_t s) { return prx_gemm_launch(d, e->ws, e->ws_bytes, s); }

int conv3(PrxVqganEnc* e, const EConv3& c, const bf16_t* x, int H, int W, int mode, const float* resid, float* out,
          bf16_t* out_bf, hipStream_t s) {
    GemmDesc d; d.A = x; d.a_mode = PRX_A_CONV3X3; d.lda = c.CiP;
    d.B = c.W; d.ldb = 9 * c.CiP; d.M = H * W; d.N = c.Cout; d.K = 9 * c.CiP;
    d.H = H; d.W = W; d.Cin = c.CiP; d.up = mode; d.bias_n = c.b; d.resid = resid; d.ldr = c.Cout;
    d.out_f32 = out; d.ldc_f32 = c.Cout; d.out_bf16 = out_bf; d.ldc_bf16 = c.Cout;
    return eg(e, d, s);
}
int gn(PrxVqganEnc* e, const EGN& g, const float* x, int P, int swish, hipStream_t s) {
    return prx_groupnorm_fwd(x, g.g, g.b, e->stats, e->a, nullptr, 1, P, g.C, swish, 1e-6f, s);
}
}  // namespace

int prx_vqgan_enc_create_impl(PrxVqganEnc** out, int ch, const int* ch_mult, int n_mult, int num_res_blocks, int attn_res,
                              int resolution, int in_ch, int z_channels, int embed_dim, int n_embed, int H, int W,
                              const float* const* w, int n_w, hipStream_t s) {
    const int f = 1 << (n_mult - 1);
    PRX_REQUIRE(H > 0 && W > 0 && H % f == 0 && W % f == 0, "vqgan_enc_create: image %dx%d must be a multiple of %d", H, W, f);
    PRX_REQUIRE(in_ch >= 1 && in_ch <= 8, "vqgan_enc_create: in_channels %d not supported", in_ch);
    PrxVqganEnc* e = new PrxVqganEnc();
    std::unique_ptr<PrxVqganEnc> guard(e);
    e->in_ch = in_ch; e->zc = z_channels; e->D = embed_dim; e->NC = n_embed; e->H = H; e->W = W; e->h0 = H / f; e->w0 = W / f;
    ECursor cur{w, n_w, 0};
    int r;
    const float* cb; ENEXT(cur, cb);
    if ((r = ecopy(e, &e->codebook, cb, (size_t)n_embed * embed_dim, s))) return r;
    EALLOC(e->cnorm, n_embed);
    if ((r = prx_sqnorm_rows(e->codebook, e->cnorm, n_embed, embed_dim, s))) return r;
    if ((r = e_conv3(e, e->conv_in, in_ch, ch, cur, s))) return r;
    int block_in = ch, h = H, wd = W, nominal = resolution;
    size_t maxPC = (size_t)H * W * ch, maxP = 1;
    for (int lvl = 0; lvl < n_mult; ++lvl) {
        const int block_out = ch * ch_mult[lvl];
        for (int b = 0; b < num_res_blocks; ++b) {
            if ((r = e_res(e, block_in, block_out, h, wd, cur, s))) return r;
            maxPC = std::max(maxPC, (size_t)h * wd * std::max(block_in, block_out));
            block_in = block_out;
            if (nominal == attn_res) { if ((r = e_attn(e, block_in, h, wd, cur, s))) return r; maxP = std::max(maxP, (size_t)h * wd); }
        }
        if (lvl != n_mult - 1) {
            EConv3 dc{};
            if ((r = e_conv3(e, dc, block_in, block_in, cur, s))) return r;
            PRX_REQUIRE(block_in % 64 == 0, "vqgan_enc_create: Downsample needs channels %% 64 == 0 (got %d)", block_in);
            h /= 2; wd /= 2; nominal /= 2;
            e->ops.push_back({2, (int)e->downs.size(), h, wd});
            e->downs.push_back(dc);
        }
    }
    if ((r = e_res(e, block_in, block_in, h, wd, cur, s))) return r;
    if ((r = e_attn(e, block_in, h, wd, cur, s))) return r;
    maxP = std::max(maxP, (size_t)h * wd);
    if ((r = e_res(e, block_in, block_in, h, wd, cur, s))) return r;
    if ((r = e_gn(e, e->norm_out, block_in, cur, s))) return r;
    if ((r = e_conv3(e, e->conv_out, block_in, z_channels, cur, s))) return r;
    if ((r = e_conv1(e, e->quant, z_channels, embed_dim, cur, s))) return r;
    PRX_REQUIRE(cur.pos == n_w, "vqgan_enc_create: %d weight tensors given, %d consumed", n_w, cur.pos);
    PRX_REQUIRE(h == e->h0 && wd == e->w0, "vqgan_enc_create: internal size mismatch");
    const size_t P0 = (size_t)e->h0 * e->w0;
    size_t maxAttnC = 1;
    for (auto& ab : e->attn) maxAttnC = std::max(maxAttnC, (size_t)ab.C);
    EALLOC(e->img8, (size_t)H * W * 8);
    EALLOC(e->a, maxPC); EALLOC(e->xb0, maxPC); EALLOC(e->xb1, maxPC);
    EALLOC(e->x0, maxPC); EALLOC(e->x1, maxPC); EALLOC(e->h1, maxPC); EALLOC(e->sc, maxPC);
    const size_t maxP8 = (maxP + 7) & ~(size_t)7;
    EALLOC(e->qkvb, maxP * 3 * maxAttnC); EALLOC(e->tA, maxP8 * maxAttnC); EALLOC(e->tB, maxP * maxAttnC);
    EALLOC(e->Pm, maxP * maxP8); EALLOC(e->PT, maxP * maxP8); EALLOC(e->S, maxP * maxP);
    EALLOC(e->co_bf, P0 * z_channels); EALLOC(e->hq, P0 * embed_dim); EALLOC(e->zq, P0 * embed_dim); EALLOC(e->idx, P0);
    const int ntiles = ceil_div(n_embed, 64);
    EALLOC(e->pmin, P0 * ntiles); EALLOC(e->pidx, P0 * ntiles);
    EALLOC(e->stats, 64);
    e->ws_bytes = (size_t)64 << 20;
    EALLOC(e->ws, e->ws_bytes / sizeof(float));
    *out = guard.release();
    return 0;
}

void prx_vqgan_enc_destroy_impl(PrxVqganEnc* e) {
    if (!e) return;
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

// img: NCHW [1, in_ch, H, W] fp32 in [-1, 1] (what pixray feeds: vqgan.py:174, pixray.py:718-727) ->
// z: NCHW [1, D, h0, w0] = the selected code vectors; z_pre (optional): NCHW pre-quantisation latent; indices (optional)
int prx_vqgan_encode_impl(PrxVqganEnc* e, const float* img, float* z, float* z_pre, int* indices, hipStream_t s) {
    int r;
    const int P = e->H * e->W;
    if ((r = prx_nchw_to_nhwc(img, nullptr, e->img8, 1, e->in_ch, P, 8, s))) return r;
    float* x = e->x0; float* xn = e->x1;
    bf16_t* xb = e->xb0; bf16_t* xbn = e->xb1;
    if ((r = conv3(e, e->conv_in, e->img8, e->H, e->W, 0, nullptr, x, xb, s))) return r;
    for (const EOp& op : e->ops) {
        const int Pc = op.H * op.W;
        if (op.kind == 0) {
            const ERes& b = e->res[op.idx];
            if ((r = gn(e, b.n1, x, Pc, 1, s))) return r;
            if ((r = conv3(e, b.c1, e->a, op.H, op.W, 0, nullptr, e->h1, nullptr, s))) return r;
            const float* resid = x;
            if (b.has_sc) {
                GemmDesc d; d.A = xb; d.lda = b.Cin; d.B = b.sc.W; d.ldb = b.Cin; d.M = Pc; d.N = b.Cout; d.K = b.Cin;
                d.bias_n = b.sc.b; d.out_f32 = e->sc; d.ldc_f32 = b.Cout;
                if ((r = eg(e, d, s))) return r;
                resid = e->sc;
            }
            if ((r = gn(e, b.n2, e->h1, Pc, 1, s))) return r;
            if ((r = conv3(e, b.c2, e->a, op.H, op.W, 0, resid, xn, xbn, s))) return r;
        } else if (op.kind == 1) {
            const EAttn& b = e->attn[op.idx];
            const int C = b.C;
            if ((r = gn(e, b.n, x, Pc, 0, s))) return r;
            {   GemmDesc d; d.A = e->a; d.lda = C; d.B = b.qkv.W; d.ldb = C; d.M = Pc; d.N = 3 * C; d.K = C;
                d.bias_n = b.qkv.b; d.out_bf16 = e->qkvb; d.ldc_bf16 = 3 * C;
                if ((r = eg(e, d, s))) return r; }
            const int P8 = (Pc + 7) & ~7;            // [*, P] operands use a row pitch of round_up(P, 8) with zero pad columns
            if (P8 != Pc) {
                PRX_CHECK_HIP(hipMemsetAsync(e->tA, 0, sizeof(bf16_t) * (size_t)C * P8, s));
                PRX_CHECK_HIP(hipMemsetAsync(e->Pm, 0, sizeof(bf16_t) * (size_t)Pc * P8, s));
            }
            if ((r = prx_transpose_op(e->qkvb + 2 * C, 3 * C, e->tA, P8, Pc, C, PRX_PREC_BF16, s))) return r;      // tA = v^T [C, P8]
            {   GemmDesc d; d.A = e->qkvb; d.lda = 3 * C; d.B = e->qkvb + C; d.ldb = 3 * C; d.M = Pc; d.N = Pc; d.K = C;
                d.out_f32 = e->S; d.ldc_f32 = Pc;
                if ((r = eg(e, d, s))) return r; }
            if ((r = prx_softmax_rows(e->S, Pc, 1.f / sqrtf((float)C), e->Pm, P8, e->PT, P8, Pc, Pc, PRX_PREC_BF16, s))) return r;
            {   GemmDesc d; d.A = e->Pm; d.lda = P8; d.B = e->tA; d.ldb = P8; d.M = Pc; d.N = C; d.K = P8;
                d.out_bf16 = e->tB; d.ldc_bf16 = C;
                if ((r = eg(e, d, s))) return r; }
            {   GemmDesc d; d.A = e->tB; d.lda = C; d.B = b.proj.W; d.ldb = C; d.M = Pc; d.N = C; d.K = C;
                d.bias_n = b.proj.b; d.resid = x; d.ldr = C; d.out_f32 = xn; d.ldc_f32 = C; d.out_bf16 = xbn; d.ldc_bf16 = C;
                if ((r = eg(e, d, s))) return r; }
        } else {
            const EConv3& c = e->downs[op.idx];
            if ((r = conv3(e, c, xb, op.H, op.W, 2, nullptr, xn, xbn, s))) return r;
        }
        std::swap(x, xn); std::swap(xb, xbn);
    }
    const int P0 = e->h0 * e->w0;
    if ((r = gn(e, e->norm_out, x, P0, 1, s))) return r;
    if ((r = conv3(e, e->conv_out, e->a, e->h0, e->w0, 0, nullptr, nullptr, e->co_bf, s))) return r;
    {   GemmDesc d; d.A = e->co_bf; d.lda = e->zc; d.B = e->quant.W; d.ldb = e->zc; d.M = P0; d.N = e->D; d.K = e->zc;
        d.bias_n = e->quant.b; d.out_f32 = e->hq; d.ldc_f32 = e->D;
        if ((r = eg(e, d, s))) return r; }
    if (z_pre && (r = prx_nhwc_to_nchw(e->hq, e->D, z_pre, 1, e->D, P0, s))) return r;
    if ((r = prx_vq_nearest(e->hq, e->D, 1, e->codebook, e->cnorm, P0, e->NC, e->D, e->pmin, e->pidx, e->idx, e->zq, s))) return r;
    if (indices) PRX_CHECK_HIP(hipMemcpyAsync(indices, e->idx, sizeof(int) * P0, hipMemcpyDeviceToDevice, s));
    return prx_nhwc_to_nchw(e->zq, e->D, z, 1, e->D, P0, s);
}
