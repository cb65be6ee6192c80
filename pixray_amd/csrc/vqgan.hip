// VqganDrawer.synth runner (vqgan.py:190-195): vector_quantize (straight-through) ->
// taming VQModel.decode = post_quant_conv + Decoder [UPSTREAM taming-transformers,
// taming/modules/diffusionmodules/model.py] -> (x+1)/2 -> ClampWithGrad, forward and
// activation-gradient backward (weights frozen, vqgan.py:125).
//
// Data layout: every feature map is NHWC ([H*W, C] row-major = a GEMM A matrix), batch 1.  The
// residual stream / conv outputs are fp32 (they are also the saved activations the GroupNorm
// backward needs); GroupNorm+swish writes the bf16 GEMM operand; 3x3 convs (and the nearest-2x
// upsample in front of them) run as implicit GEMMs on the MFMA engine, dgrad uses the
// flipped/transposed weight pack.
#include "vqgan.h"
#include "gemm.h"
#include "norms.h"
#include "elementwise.h"
#include "prompt_vq.h"
#include "vit.h"  // prx_pack_* helpers
#include <vector>
#include <memory>
#include <stdlib.h>

namespace {

// Wf[co][tap*Cin + ci] = w[co][ci][ky][kx]           (forward pack)
// Wd[ci][tap*CoP + co] = w[co][ci][2-ky][2-kx]       (dgrad pack; co padded with zeros to CoP)
template <typename TOp>
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* __restrict__ w, TOp* __restrict__ Wf,
                                                           TOp* __restrict__ Wd, int Cout, int Cin, int CoP) {
    const size_t total_f = (size_t)Cout * 9 * Cin;
    const size_t total_d = (size_t)Cin * 9 * CoP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_f + total_d;
         i += (size_t)gridDim.x * blockDim.x) {
        if (i < total_f) {
            int ci = (int)(i % Cin);
            int tap = (int)((i / Cin) % 9);
            int co = (int)(i / ((size_t)9 * Cin));
            Wf[i] = op_cvt<TOp>(w[(((size_t)co * Cin + ci) * 3 + tap / 3) * 3 + tap % 3]);
        } else {
            size_t j = i - total_f;
            int co = (int)(j % CoP);
            int tap = (int)((j / CoP) % 9);
            int ci = (int)(j / ((size_t)9 * CoP));
            int ky = 2 - tap / 3, kx = 2 - tap % 3;
            Wd[j] = op_cvt<TOp>((co < Cout) ? w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx] : 0.f);
        }
    }
}

__global__ __launch_bounds__(256) void colminmax_kernel(const float* __restrict__ w, float* __restrict__ mn,
                                                        float* __restrict__ mx, int rows, int D) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float a = INFINITY, b = -INFINITY;
    for (int r = 0; r < rows; ++r) { float v = w[(size_t)r * D + c]; a = fminf(a, v); b = fmaxf(b, v); }
    mn[c] = a; mx[c] = b;
}

}  // namespace

static inline int pad8(int P) { return (P + 7) & ~7; }   // row pitch of [*, P] GEMM operands (see zero_if_padded)

// `void*` members are operand-precision buffers: bf16, or fp32 when the handle was created with PRX_PREC_F32
struct Conv3 { int Cin, Cout, CoP; void *Wf, *Wd; float* b; };
struct Conv1 { int Cin, Cout; void *W, *WT; float* b; };
struct GN { int C; float *g, *b; double *stats, *bstats; int id; };

struct ResBlock {
    int Cin, Cout, rh, rw;   // feature-map height x width (pixray sizes need not be square)
    GN n1, n2; Conv3 c1, c2; Conv1 sc; bool has_sc;
    void *x_in, *h1, *scbuf, *out;   // streams: fp32, or the 16-bit operand format in the lean layout (PrxVqgan::lean)
    void *x_in_bf, *out_bf;     // operand twins (only where a GEMM consumes the tensor); the stream itself in the exact mode and in the lean layout
};
struct AttnBlock {
    int C, rh, rw;
    GN n; Conv1 qkv, proj;   // qkv = [3C, C] concatenated q|k|v
    void* x_in; void *qkvb, *Pm, *PT; void* out; void* out_bf;
    void* qkvT;      // [3C, P8]: q^T | k^T | v^T, ONE transpose of the block's [P, 3C] q|k|v tensor in the forward -- the products that
                     // contract over the tokens (P v forward; dS k, dS^T q backward) need that operand token-contiguous
};
struct UpBlock { int C, rh, rw; Conv3 c; void *x_in, *out; void *x_in_bf, *out_bf; };

struct Stage { int kind; int idx; };  // 0 res, 1 attn, 2 up

struct PrxVqgan {
    int zc, D, NC, ch, out_ch, h0, w0, H, W, nstage;
    int prec;         // PRX_PREC_*
    int f32, h16;     // derived: operands are fp32 / the 16-bit operand format is IEEE half
    // The lean layout (half mode, PRX_LEAN): every feature map and every gradient map lives in IEEE half ONLY -- one tensor is
    // the saved activation (GroupNorm backward reads it), the residual operand and the next convolution's A matrix -- instead of an
    // fp32 stream plus a 16-bit twin (40 % of the iteration's HBM writes).  GroupNorm sums are still taken from the fp32
    // accumulators in the producing epilogue.
    int lean;
    float* gs;        // half mode: device {S, 1/S} = the power-of-two scale of the backward in flight (common.h) + 256 partials; else null
    GemmCtx gctx;     // this handle's engine state
    std::vector<void*> allocs;
    float *codebook, *cnorm, *zmin, *zmax;
    Conv1 pq; Conv3 conv_in, conv_out; GN norm_out;
    std::vector<ResBlock> res; std::vector<AttnBlock> attn; std::vector<UpBlock> ups; std::vector<Stage> stages;
    // activations
    float *zq, *y, *dzq; void* h_in; int* idx; void *pqo_bf, *h_in_bf, *dpq_bf;
    float *pmin; int* pidx;
    void* a;               // GN(+swish) operand, max size
    void *tA, *tB, *tC, *tD, *dqkv, *dy8;    // attention temporaries [P*C max], dgrad head input
    void *dycol, *Wd_col;                    // 16-bit modes: the head gradient as the im2col matrix [H*W][128] of conv_out's dgrad and that dgrad's
                                             // weight pack with its rows padded to 128 (elementwise.h prx_image_head_bwd_im2col); else null
    float* S;                // score matrix
    void *g0, *g1, *g2;      // gradient ping-pong streams (max P*C): fp32, 16-bit in the lean layout
    void *g0b, *g1b, *g2b;   // their operand twins (dgrad GEMM operands); aliases of g0..g2 in the exact mode and in the lean layout
    double* all_stats; int n_gn;   // [n_gn][64] forward stats followed by [n_gn][64] backward stats
    float* ws; size_t ws_bytes;
    void* x_last;   // input of norm_out (a stream)
};

namespace {
template <typename Tp>
int dalloc(PrxVqgan* v, Tp** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(Tp)));
    v->allocs.push_back(q);
    *p = (Tp*)q;
    return 0;
}
#define VALLOC(ptr, count) do { int _r = dalloc(v, &(ptr), (count)); if (_r) return _r; } while (0)
int dalloc_op(PrxVqgan* v, void** p, size_t count) {   // `count` operand elements
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * op_esz(v->f32)));
    v->allocs.push_back(q);
    *p = q;
    return 0;
}
#define VALLOC_OP(ptr, count) do { int _r = dalloc_op(v, &(ptr), (count)); if (_r) return _r; } while (0)
int dalloc_stream(PrxVqgan* v, void** p, size_t count) {   // `count` stream elements: fp32, or 16-bit in the lean layout
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * (v->lean ? sizeof(bf16_t) : sizeof(float))));
    v->allocs.push_back(q);
    *p = q;
    return 0;
}
#define VALLOC_S(ptr, count) do { int _r = dalloc_stream(v, &(ptr), (count)); if (_r) return _r; } while (0)

struct WCursor { const float* const* w; int n, pos; };
#define NEXTW(cur, dst) do { PRX_REQUIRE((cur).pos < (cur).n, "vqgan_create: weight list too short"); (dst) = (cur).w[(cur).pos++]; } while (0)

int copyf(PrxVqgan* v, float** dst, const float* src, size_t n, hipStream_t s) {
    VALLOC(*dst, n);
    PRX_CHECK_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int make_gn(PrxVqgan* v, GN& g, int C, WCursor& cur, hipStream_t s) {
    const float *w, *b; NEXTW(cur, w); NEXTW(cur, b);
    g.C = C;
    int r;
    if ((r = copyf(v, &g.g, w, C, s))) return r;
    if ((r = copyf(v, &g.b, b, C, s))) return r;
    g.id = v->n_gn++;
    g.stats = g.bstats = nullptr;   // assigned once every GroupNorm is known
    return 0;
}
int make_conv3(PrxVqgan* v, Conv3& c, int Cin, int Cout, WCursor& cur, hipStream_t s) {
    const float *w, *b; NEXTW(cur, w); NEXTW(cur, b);
    c.Cin = Cin; c.Cout = Cout; c.CoP = (Cout + 7) / 8 * 8;
    // the forward pack has CoP rows too (zero rows beyond Cout, zero bias): N % 8 == 0 puts conv_out (3 channels) on the MFMA tiles
    // with vector epilogues like every other convolution; its fp32 output is [pixels][CoP]
    VALLOC_OP(c.Wf, (size_t)c.CoP * 9 * Cin);
    if (c.CoP != Cout) PRX_CHECK_HIP(hipMemsetAsync(c.Wf, 0, (size_t)c.CoP * 9 * Cin * op_esz(v->f32), s));
    VALLOC_OP(c.Wd, (size_t)Cin * 9 * c.CoP);
    PRX_OP_DISPATCH(v->f32, v->h16, TO,
                    hipLaunchKernelGGL(pack_conv3x3_kernel<TO>, dim3(1024), dim3(256), 0, s, w, (TO*)c.Wf, (TO*)c.Wd, Cout, Cin, c.CoP));
    PRX_LAUNCH_CHECK();
    VALLOC(c.b, c.CoP);
    if (c.CoP != Cout) PRX_CHECK_HIP(hipMemsetAsync(c.b, 0, sizeof(float) * c.CoP, s));
    PRX_CHECK_HIP(hipMemcpyAsync(c.b, b, sizeof(float) * Cout, hipMemcpyDeviceToDevice, s));
    return 0;
}
int make_conv1(PrxVqgan* v, Conv1& c, int Cin, int Cout, WCursor& cur, hipStream_t s) {
    const float *w, *b; NEXTW(cur, w); NEXTW(cur, b);
    c.Cin = Cin; c.Cout = Cout;
    VALLOC_OP(c.W, (size_t)Cout * Cin); VALLOC_OP(c.WT, (size_t)Cout * Cin);
    int r;
    if ((r = prx_pack_op(w, c.W, (size_t)Cout * Cin, v->prec, s))) return r;
    if ((r = prx_pack_transpose_op(w, c.WT, Cout, Cin, v->prec, s))) return r;
    return copyf(v, &c.b, b, Cout, s);
}
int make_res(PrxVqgan* v, int Cin, int Cout, int rh, int rw, WCursor& cur, hipStream_t s) {
    ResBlock rb{};
    rb.Cin = Cin; rb.Cout = Cout; rb.rh = rh; rb.rw = rw; rb.has_sc = Cin != Cout;
    int r;
    if ((r = make_gn(v, rb.n1, Cin, cur, s))) return r;
    if ((r = make_conv3(v, rb.c1, Cin, Cout, cur, s))) return r;
    if ((r = make_gn(v, rb.n2, Cout, cur, s))) return r;
    if ((r = make_conv3(v, rb.c2, Cout, Cout, cur, s))) return r;
    const size_t P = (size_t)rh * rw;
    if (rb.has_sc) {
        if ((r = make_conv1(v, rb.sc, Cin, Cout, cur, s))) return r;
        VALLOC_S(rb.scbuf, P * Cout);
    }
    VALLOC_S(rb.h1, P * Cout); VALLOC_S(rb.out, P * Cout);
    v->stages.push_back({0, (int)v->res.size()});
    v->res.push_back(rb);
    return 0;
}
int make_attn(PrxVqgan* v, int C, int rh, int rw, WCursor& cur, hipStream_t s) {
    AttnBlock ab{};
    ab.C = C; ab.rh = rh; ab.rw = rw;
    int r;
    if ((r = make_gn(v, ab.n, C, cur, s))) return r;
    // q, k, v 1x1 convs are concatenated into one [3C, C] GEMM
    const float *wq, *bq, *wk, *bk, *wv, *bv;
    NEXTW(cur, wq); NEXTW(cur, bq); NEXTW(cur, wk); NEXTW(cur, bk); NEXTW(cur, wv); NEXTW(cur, bv);
    float *wcat, *bcat;
    VALLOC(wcat, (size_t)3 * C * C); VALLOC(bcat, 3 * C);
    const float* ws_[3] = {wq, wk, wv}; const float* bs_[3] = {bq, bk, bv};
    for (int i = 0; i < 3; ++i) {
        PRX_CHECK_HIP(hipMemcpyAsync(wcat + (size_t)i * C * C, ws_[i], sizeof(float) * C * C, hipMemcpyDeviceToDevice, s));
        PRX_CHECK_HIP(hipMemcpyAsync(bcat + i * C, bs_[i], sizeof(float) * C, hipMemcpyDeviceToDevice, s));
    }
    ab.qkv.Cin = C; ab.qkv.Cout = 3 * C; ab.qkv.b = bcat;
    VALLOC_OP(ab.qkv.W, (size_t)3 * C * C); VALLOC_OP(ab.qkv.WT, (size_t)3 * C * C);
    if ((r = prx_pack_op(wcat, ab.qkv.W, (size_t)3 * C * C, v->prec, s))) return r;
    if ((r = prx_pack_transpose_op(wcat, ab.qkv.WT, 3 * C, C, v->prec, s))) return r;
    if ((r = make_conv1(v, ab.proj, C, C, cur, s))) return r;
    const size_t P = (size_t)rh * rw;
    const size_t P8 = (size_t)pad8((int)P);
    VALLOC_OP(ab.qkvb, P * 3 * C); VALLOC_OP(ab.Pm, P * P8); VALLOC_OP(ab.PT, P * P8); VALLOC_S(ab.out, P * C);
    VALLOC_OP(ab.qkvT, (size_t)3 * C * P8);
    PRX_CHECK_HIP(hipMemsetAsync(ab.qkvT, 0, (size_t)3 * C * P8 * op_esz(v->f32), s));
    PRX_CHECK_HIP(hipMemsetAsync(ab.Pm, 0, P * P8 * op_esz(v->f32), s));      // pad columns stay zero: the kernels never write them
    PRX_CHECK_HIP(hipMemsetAsync(ab.PT, 0, P * P8 * op_esz(v->f32), s));
    v->stages.push_back({1, (int)v->attn.size()});
    v->attn.push_back(ab);
    return 0;
}
int make_up(PrxVqgan* v, int C, int rh, int rw, WCursor& cur, hipStream_t s) {
    UpBlock ub{};
    ub.C = C; ub.rh = rh; ub.rw = rw;
    int r;
    if ((r = make_conv3(v, ub.c, C, C, cur, s))) return r;
    VALLOC_S(ub.out, (size_t)rh * rw * C);
    v->stages.push_back({2, (int)v->ups.size()});
    v->ups.push_back(ub);
    return 0;
}
}  // namespace

int prx_vqgan_create_impl(PrxVqgan** out, int ch, const int* ch_mult, int n_mult, int num_res_blocks, int attn_res,
                          int resolution, int z_channels, int embed_dim, int n_embed, int out_ch, int h0, int w0,
                          int precision, const float* const* w, int n_w, hipStream_t s) {
    PRX_REQUIRE(h0 >= 1 && w0 >= 1, "vqgan_create: bad latent size %dx%d", h0, w0);
    PRX_REQUIRE(prec_valid(precision), "vqgan_create: unknown precision %d", precision);
    PRX_REQUIRE(embed_dim == z_channels, "vqgan_create: embed_dim must equal z_channels");
    PrxVqgan* v = new PrxVqgan();
    std::unique_ptr<PrxVqgan> guard(v);
    v->prec = precision; v->f32 = prec_is_f32(precision); v->h16 = prec_is_h16(precision);
    v->gs = nullptr;
    { const char* e = getenv("PRX_LEAN"); v->lean = (v->h16 && !(e && atoi(e) == 0)) ? 1 : 0; }
    v->n_gn = 0; v->zc = z_channels; v->D = embed_dim; v->NC = n_embed; v->ch = ch; v->out_ch = out_ch; v->h0 = h0; v->w0 = w0;
    WCursor cur{w, n_w, 0};
    int r;
    const float* cb; NEXTW(cur, cb);
    if ((r = copyf(v, &v->codebook, cb, (size_t)n_embed * embed_dim, s))) return r;
    VALLOC(v->cnorm, n_embed); VALLOC(v->zmin, embed_dim); VALLOC(v->zmax, embed_dim);
    if ((r = prx_sqnorm_rows(v->codebook, v->cnorm, n_embed, embed_dim, s))) return r;
    hipLaunchKernelGGL(colminmax_kernel, dim3(ceil_div(embed_dim, 256)), dim3(256), 0, s, v->codebook, v->zmin, v->zmax,
                       n_embed, embed_dim);
    PRX_LAUNCH_CHECK();
    if ((r = make_conv1(v, v->pq, embed_dim, z_channels, cur, s))) return r;
    int block_in = ch * ch_mult[n_mult - 1];
    int rh = h0, rw = w0;
    if ((r = make_conv3(v, v->conv_in, z_channels, block_in, cur, s))) return r;
    if ((r = make_res(v, block_in, block_in, rh, rw, cur, s))) return r;
    if ((r = make_attn(v, block_in, rh, rw, cur, s))) return r;
    if ((r = make_res(v, block_in, block_in, rh, rw, cur, s))) return r;
    // taming decides AttnBlock placement from the config's NOMINAL resolution, not the actual latent size
    int nominal = resolution >> (n_mult - 1);
    for (int lvl = n_mult - 1; lvl >= 0; --lvl) {
        int block_out = ch * ch_mult[lvl];
        for (int b = 0; b < num_res_blocks + 1; ++b) {
            if ((r = make_res(v, block_in, block_out, rh, rw, cur, s))) return r;
            block_in = block_out;
            if (nominal == attn_res) { if ((r = make_attn(v, block_in, rh, rw, cur, s))) return r; }
        }
        if (lvl != 0) {
            rh *= 2; rw *= 2; nominal *= 2;
            if ((r = make_up(v, block_in, rh, rw, cur, s))) return r;
        }
    }
    v->H = rh; v->W = rw;
    if ((r = make_gn(v, v->norm_out, block_in, cur, s))) return r;
    if ((r = make_conv3(v, v->conv_out, block_in, out_ch, cur, s))) return r;
    PRX_REQUIRE(cur.pos == n_w, "vqgan_create: %d weight tensors given, %d consumed", n_w, cur.pos);
    // activations / scratch
    const size_t P0 = (size_t)h0 * w0, PH = (size_t)rh * rw;
    size_t maxPC = P0 * (size_t)std::max(z_channels, embed_dim), maxAttnPC = 1;      // conv_in's / post_quant_conv's input gradients land in the gradient streams too
    for (auto& rb : v->res) maxPC = std::max(maxPC, (size_t)rb.rh * rb.rw * std::max(rb.Cin, rb.Cout));
    for (auto& ub : v->ups) maxPC = std::max(maxPC, (size_t)ub.rh * ub.rw * ub.C);
    for (auto& ab : v->attn) maxAttnPC = std::max(maxAttnPC, (size_t)pad8(ab.rh * ab.rw) * (size_t)std::max(ab.C, pad8(ab.rh * ab.rw)));
    VALLOC(v->zq, P0 * embed_dim); VALLOC_OP(v->pqo_bf, P0 * z_channels); VALLOC_OP(v->dpq_bf, P0 * z_channels);
    VALLOC_S(v->h_in, P0 * (size_t)(ch * ch_mult[n_mult - 1]));
    VALLOC(v->y, PH * (size_t)v->conv_out.CoP); VALLOC(v->idx, P0); VALLOC(v->dzq, P0 * embed_dim);
    const int ntiles = ceil_div(n_embed, 64);
    VALLOC(v->pmin, P0 * ntiles); VALLOC(v->pidx, P0 * ntiles);
    VALLOC_OP(v->a, maxPC);
    VALLOC_OP(v->tA, maxAttnPC); VALLOC_OP(v->tB, maxAttnPC); VALLOC_OP(v->tC, maxAttnPC); VALLOC_OP(v->tD, maxAttnPC);
    VALLOC_OP(v->dqkv, maxAttnPC * 3); VALLOC_OP(v->dy8, PH * 8);
    v->dycol = v->Wd_col = nullptr;
    if (!v->f32 && v->conv_out.CoP == 8) {
        const int Cin = v->conv_out.Cin;
        VALLOC_OP(v->dycol, PH * 128); VALLOC_OP(v->Wd_col, (size_t)Cin * 128);
        PRX_CHECK_HIP(hipMemsetAsync(v->Wd_col, 0, (size_t)Cin * 128 * sizeof(bf16_t), s));
        PRX_CHECK_HIP(hipMemcpy2DAsync(v->Wd_col, 128 * sizeof(bf16_t), v->conv_out.Wd, 72 * sizeof(bf16_t), 72 * sizeof(bf16_t), Cin, hipMemcpyDeviceToDevice, s));
    }
    VALLOC(v->S, maxAttnPC);
    VALLOC_S(v->g0, maxPC); VALLOC_S(v->g1, maxPC); VALLOC_S(v->g2, maxPC);
    if (v->f32 || v->lean) { v->g0b = v->g0; v->g1b = v->g1; v->g2b = v->g2; }     // the gradient streams are the dgrad operands
    else { VALLOC_OP(v->g0b, maxPC); VALLOC_OP(v->g1b, maxPC); VALLOC_OP(v->g2b, maxPC); }
    // bf16 twins of stage outputs that feed a GEMM directly (1x1 shortcut or upsample conv of the next stage)
    v->h_in_bf = nullptr;
    for (size_t i = 0; i < v->stages.size(); ++i) {
        const Stage& st = v->stages[i];
        const bool need = (st.kind == 0 && v->res[st.idx].has_sc) || st.kind == 2;
        if (!need) continue;
        void** slot; size_t cnt; void* self;
        if (i == 0) { slot = &v->h_in_bf; cnt = P0 * (size_t)(ch * ch_mult[n_mult - 1]); self = v->h_in; }
        else {
            const Stage& pr = v->stages[i - 1];
            if (pr.kind == 0) { ResBlock& b = v->res[pr.idx]; slot = &b.out_bf; cnt = (size_t)b.rh * b.rw * b.Cout; self = b.out; }
            else if (pr.kind == 1) { AttnBlock& b = v->attn[pr.idx]; slot = &b.out_bf; cnt = (size_t)b.rh * b.rw * b.C; self = b.out; }
            else { UpBlock& b = v->ups[pr.idx]; slot = &b.out_bf; cnt = (size_t)b.rh * b.rw * b.C; self = b.out; }
        }
        if (v->f32 || v->lean) *slot = self;          // exact mode / lean layout: the stage output is the operand
        else VALLOC_OP(*slot, cnt);
    }
    {   // one contiguous stats slab: forward stats of every GroupNorm, then the backward stats
        VALLOC(v->all_stats, (size_t)2 * v->n_gn * 64);
        auto fix = [&](GN& g) { g.stats = v->all_stats + (size_t)g.id * 64; g.bstats = v->all_stats + (size_t)(v->n_gn + g.id) * 64; };
        for (auto& b : v->res) { fix(b.n1); fix(b.n2); }
        for (auto& b : v->attn) fix(b.n);
        fix(v->norm_out);
    }
    v->ws_bytes = (size_t)64 << 20;
    VALLOC(v->ws, v->ws_bytes / sizeof(float));
    if (v->h16) VALLOC(v->gs, 2 + 256);
    *out = guard.release();
    return 0;
}

void prx_vqgan_destroy_impl(PrxVqgan* v) {
    if (!v) return;
    for (void* p : v->allocs) (void)hipFree(p);
    delete v;
}

// the GEMM epilogue can accumulate the next GroupNorm's sums only for power-of-two group sizes >= 4 channels
// (the exact mode too: its fp32-operand fit kernels carry the sums, and the engine falls back to the norm kernels' own
// statistics pass for a product those kernels do not take -- gemm.hip; PRX_F32_GN_FUSE=0: always the separate passes)
static bool fusable(const PrxVqgan* v, int C) {
    static const bool f32_fuse = [] { const char* e = getenv("PRX_F32_GN_FUSE"); return !(e && atoi(e) == 0); }();
    const int gs = C / 32;
    return (!v->f32 || f32_fuse) && C % 32 == 0 && gs >= 4 && (gs & (gs - 1)) == 0;
}

// Token counts of the attention maps (P = h*w of the latent) are arbitrary (pixray sizes are multiples of 16 pixels, so
// e.g. 25x14 = 350 tokens), but GEMM K dimensions and leading dimensions must be multiples of 8: every [*, P] operand is
// laid out with a row pitch of P8 = round_up(P, 8) and zero columns beyond P.
static int zero_if_padded(const PrxVqgan* v, void* buf, size_t rows, int P, hipStream_t s) {
    if (pad8(P) != P) PRX_CHECK_HIP(hipMemsetAsync(buf, 0, rows * (size_t)pad8(P) * op_esz(v->f32), s));
    return 0;
}


// PRX_VQ_TRACE=1 (debugging aid): L2 norm of every stage's output, forward and backward, on stderr
static void vq_trace(const PrxVqgan* v, const char* tag, int i, const void* p_, size_t n, hipStream_t s) {
    static const bool on = getenv("PRX_VQ_TRACE") != nullptr;
    if (!on || !p_ || v->lean) return;             // (fp32 streams only: PRX_LEAN=0 for a trace)
    const float* p = (const float*)p_;
    std::vector<float> h(n);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), p, n * sizeof(float), hipMemcpyDeviceToHost);
    double q = 0, mx = 0;
    for (float x : h) { q += (double)x * x; mx = fabs(x) > mx ? fabs(x) : mx; }
    fprintf(stderr, "[vq] %s %2d n=%zu l2 %.9e max %.6e\n", tag, i, n, sqrt(q), mx);
}

static int vg(PrxVqgan* v, GemmDesc& d, hipStream_t s) {
    if (v->f32) {
        d.f32 = 1; d.a_is_f32 = 0;
        if (d.out_bf16 == (void*)d.out_f32) d.out_bf16 = nullptr;   // the "twin" is the fp32 output itself
    }
    d.h16 = v->h16;
    return prx_gemm_launch(d, v->ws, v->ws_bytes, s, &v->gctx);
}
GemmCtx* prx_vqgan_gemm_ctx_impl(PrxVqgan* v) { return v ? &v->gctx : nullptr; }

// A GEMM's stream output / residual / GroupNorm-input operands in the handle's layout: fp32 stream (+ optional 16-bit twin), or -- lean --
// the one 16-bit tensor
static void out_stream(const PrxVqgan* v, GemmDesc& d, void* stream, int ld, void* twin) {
    if (v->lean) { d.out_bf16 = stream ? stream : twin; d.ldc_bf16 = ld; }
    else { d.out_f32 = (float*)stream; d.ldc_f32 = ld; d.out_bf16 = twin; d.ldc_bf16 = ld; }
}
static void resid_stream(const PrxVqgan* v, GemmDesc& d, const void* r, int ld) {
    if (!r) return;
    d.ldr = ld;
    d.resid = (const float*)r;
    if (v->lean) d.row16 |= 1;
}
// `f32_out`: `out` is an fp32 buffer in every layout (the image head's input), not a stream
static int conv3_fwd(PrxVqgan* v, const Conv3& c, const void* x, bool x_f32, int rh, int rw, bool up, const void* resid,
                     void* out, int ldc, hipStream_t s, void* out_bf = nullptr, const GN* stats_for = nullptr, bool f32_out = false) {
    GemmDesc d; d.A = x; d.a_is_f32 = x_f32; d.a_mode = PRX_A_CONV3X3; d.lda = c.Cin;
    d.B = c.Wf; d.ldb = 9 * c.Cin; d.M = rh * rw; d.N = c.CoP; d.K = 9 * c.Cin;       // CoP == Cout except for conv_out
    d.H = rh; d.W = rw; d.Cin = c.Cin; d.up = up; d.bias_n = c.b;
    resid_stream(v, d, resid, c.Cout);
    if (f32_out) { d.out_f32 = (float*)out; d.ldc_f32 = ldc; }
    else { out_stream(v, d, out, ldc, out_bf); if (!v->lean) d.ldc_bf16 = c.Cout; }
    if (stats_for && stats_for->C == c.Cout && fusable(v, c.Cout)) { d.gn_stats = stats_for->stats; d.gn_gs = c.Cout / 32; }
    return vg(v, d, s);
}
// dgrad of a 3x3 conv: dx[res*res, Cin] = convT(dy[res*res, CoP])
// `gnb` (+ its forward input gnb_x): the GroupNorm whose output gradient this dgrad produces -- its backward sums are
// accumulated in the GEMM epilogue (gemm.h gnb_*), so gn_bwd can skip its statistics pass
static void set_gnb(const PrxVqgan* v, GemmDesc& d, const GN* gnb, const void* gnb_x, int swish) {
    if (!gnb || !fusable(v, gnb->C) || d.N != gnb->C) return;
    d.gn_stats = gnb->bstats; d.gn_gs = gnb->C / 32;
    d.gnb_x = (const float*)gnb_x;
    if (v->lean) d.row16 |= 2;
    d.gnb_fstats = gnb->stats; d.gnb_gamma = gnb->g; d.gnb_beta = gnb->b; d.gnb_swish = swish; d.gnb_eps = 1e-6f;
}
static int conv3_bwd(PrxVqgan* v, const Conv3& c, const void* dy, bool dy_f32, int rh, int rw, void* dx, hipStream_t s,
                     void* dx_bf = nullptr, const GN* gnb = nullptr, const void* gnb_x = nullptr, int gnb_swish = 1) {
    GemmDesc d; d.A = dy; d.a_is_f32 = dy_f32; d.a_mode = PRX_A_CONV3X3; d.lda = c.CoP;
    d.B = c.Wd; d.ldb = 9 * c.CoP; d.M = rh * rw; d.N = c.Cin; d.K = 9 * c.CoP;
    d.H = rh; d.W = rw; d.Cin = c.CoP; d.up = 0;
    out_stream(v, d, dx, c.Cin, dx_bf);
    set_gnb(v, d, gnb, gnb_x, gnb_swish);
    return vg(v, d, s);
}
// first GroupNorm of stage `si` (whose statistics the producer of that stage's input can accumulate in its epilogue)
static const GN* first_norm(const PrxVqgan* v, int si) {
    if (si >= (int)v->stages.size()) return &v->norm_out;
    const Stage& st = v->stages[si];
    if (st.kind == 0) return &v->res[st.idx].n1;
    if (st.kind == 1) return &v->attn[st.idx].n;
    return nullptr;
}
static int gn_fwd(PrxVqgan* v, const GN& g, const void* x, int P, int swish, hipStream_t s, bool stats_ready = false) {
    return prx_groupnorm_fwd(x, g.g, g.b, g.stats, v->f32 ? nullptr : (bf16_t*)v->a, v->f32 ? (float*)v->a : nullptr, 1, P, g.C, swish,
                             1e-6f, s, /*zero_stats=*/0, stats_ready ? 1 : 0, v->h16, v->lean);
}
// dx: the fp32 gradient stream (may be null: only the operand twin is wanted); lean layout: dx_bf IS the stream, no fp32 output
static int gn_bwd(PrxVqgan* v, const GN& g, const void* grad, const void* x, const void* add, void* dx,
                  void* dx_bf, int P, int swish, hipStream_t s, bool stats_ready = false) {
    return prx_groupnorm_bwd(grad, x, g.g, g.b, g.stats, g.bstats, add, v->lean ? nullptr : (float*)dx, v->f32 ? nullptr : (bf16_t*)dx_bf, 1, P, g.C,
                             swish, 1e-6f, s, /*zero_stats=*/0, stats_ready && fusable(v, g.C) ? 1 : 0, v->h16, v->lean);
}

int prx_vqgan_bounds_impl(PrxVqgan* v, float* zmin, float* zmax, hipStream_t s) {
    PRX_CHECK_HIP(hipMemcpyAsync(zmin, v->zmin, sizeof(float) * v->D, hipMemcpyDeviceToDevice, s));
    PRX_CHECK_HIP(hipMemcpyAsync(zmax, v->zmax, sizeof(float) * v->D, hipMemcpyDeviceToDevice, s));
    return 0;
}

// z: NCHW [1, zc, h0, w0] fp32 -> img NCHW [1, out_ch, H, W] in [0,1]; indices (optional) int32 [h0*w0]
int prx_vqgan_synth_impl(PrxVqgan* v, const float* z, float* img, int* indices, int quantize, hipStream_t s) {
    const int P0 = v->h0 * v->w0;
    int r;
    PRX_CHECK_HIP(hipMemsetAsync(v->all_stats, 0, sizeof(double) * (size_t)v->n_gn * 64, s));   // all forward GN stats
    if (quantize) {
        if ((r = prx_vq_nearest(z, 1, P0, v->codebook, v->cnorm, P0, v->NC, v->D, v->pmin, v->pidx, v->idx, v->zq, s))) return r;
        if (indices) PRX_CHECK_HIP(hipMemcpyAsync(indices, v->idx, sizeof(int) * P0, hipMemcpyDeviceToDevice, s));
    } else {
        if ((r = prx_nchw_to_nhwc(z, v->zq, nullptr, 1, v->zc, P0, v->zc, s))) return r;
    }
    {   GemmDesc d; d.A = v->zq; d.a_is_f32 = 1; d.lda = v->D; d.B = v->pq.W; d.ldb = v->D; d.M = P0; d.N = v->zc; d.K = v->D;
        d.bias_n = v->pq.b; d.out_bf16 = v->pqo_bf; d.ldc_bf16 = v->zc;
        if ((r = vg(v, d, s))) return r; }
    // `sr`: the statistics of the GroupNorm that consumes x next were already accumulated by x's producer
    const GN* nx = first_norm(v, 0);
    if ((r = conv3_fwd(v, v->conv_in, v->pqo_bf, false, v->h0, v->w0, false, nullptr, v->h_in, v->conv_in.Cout, s, v->h_in_bf, nx))) return r;
    bool sr = nx && nx->C == v->conv_in.Cout && fusable(v, nx->C);
    void* x = v->h_in;
    void* x_bf = v->h_in_bf;     // operand twin of x (null when no GEMM reads x directly)
    for (int si = 0; si < (int)v->stages.size(); ++si) {
        const Stage& st = v->stages[si];
        nx = first_norm(v, si + 1);
        if (st.kind == 0) {
            ResBlock& b = v->res[st.idx];
            const int P = b.rh * b.rw;
            b.x_in = x; b.x_in_bf = x_bf;
            if ((r = gn_fwd(v, b.n1, x, P, 1, s, sr))) return r;
            if ((r = conv3_fwd(v, b.c1, v->a, false, b.rh, b.rw, false, nullptr, b.h1, b.Cout, s, nullptr, &b.n2))) return r;
            const void* resid = x;
            if (b.has_sc) {
                PRX_REQUIRE(x_bf != nullptr, "vqgan: missing bf16 twin for the shortcut input");
                GemmDesc d; d.A = x_bf; d.lda = b.Cin; d.B = b.sc.W; d.ldb = b.Cin; d.M = P; d.N = b.Cout; d.K = b.Cin;
                d.bias_n = b.sc.b; out_stream(v, d, b.scbuf, b.Cout, nullptr);
                if ((r = vg(v, d, s))) return r;
                resid = b.scbuf;
            }
            if ((r = gn_fwd(v, b.n2, b.h1, P, 1, s, fusable(v, b.Cout)))) return r;
            if ((r = conv3_fwd(v, b.c2, v->a, false, b.rh, b.rw, false, resid, b.out, b.Cout, s, b.out_bf, nx))) return r;
            sr = nx && nx->C == b.Cout && fusable(v, b.Cout);
            x = b.out; x_bf = b.out_bf;
        } else if (st.kind == 1) {
            AttnBlock& b = v->attn[st.idx];
            const int P = b.rh * b.rw, C = b.C;
            b.x_in = x;
            if ((r = gn_fwd(v, b.n, x, P, 0, s, sr))) return r;
            {   GemmDesc d; d.A = v->a; d.lda = C; d.B = b.qkv.W; d.ldb = C; d.M = P; d.N = 3 * C; d.K = C;
                d.bias_n = b.qkv.b; d.out_bf16 = b.qkvb; d.ldc_bf16 = 3 * C;
                if ((r = vg(v, d, s))) return r; }
            const int P8 = pad8(P);
            if ((r = prx_transpose_op(b.qkvb, 3 * C, b.qkvT, P8, P, 3 * C, v->f32, s))) return r;  // q^T | k^T | v^T [3C, P8] (pad columns stay zero)
            {   GemmDesc d; d.A = b.qkvb; d.lda = 3 * C; d.B = op_off(b.qkvb, C, v->f32); d.ldb = 3 * C; d.M = P; d.N = P; d.K = C;
                d.out_f32 = v->S; d.ldc_f32 = P;
                if ((r = vg(v, d, s))) return r; }
            if ((r = prx_softmax_rows(v->S, P, 1.f / sqrtf((float)C), b.Pm, P8, b.PT, P8, P, P, v->prec, s))) return r;
            {   GemmDesc d; d.A = b.Pm; d.lda = P8; d.B = op_off(b.qkvT, (size_t)2 * C * P8, v->f32); d.ldb = P8; d.M = P; d.N = C; d.K = P8;
                d.out_bf16 = v->tB; d.ldc_bf16 = C;
                if ((r = vg(v, d, s))) return r; }
            {   GemmDesc d; d.A = v->tB; d.lda = C; d.B = b.proj.W; d.ldb = C; d.M = P; d.N = C; d.K = C;
                d.bias_n = b.proj.b; resid_stream(v, d, x, C); out_stream(v, d, b.out, C, b.out_bf);
                if (nx && nx->C == C && fusable(v, C)) { d.gn_stats = nx->stats; d.gn_gs = C / 32; }
                if ((r = vg(v, d, s))) return r; }
            sr = nx && nx->C == C && fusable(v, C);
            x = b.out; x_bf = b.out_bf;
        } else {
            UpBlock& b = v->ups[st.idx];
            b.x_in = x; b.x_in_bf = x_bf;
            PRX_REQUIRE(x_bf != nullptr, "vqgan: missing bf16 twin for the upsample input");
            if ((r = conv3_fwd(v, b.c, x_bf, false, b.rh, b.rw, true, nullptr, b.out, b.C, s, b.out_bf, nx))) return r;
            sr = nx && nx->C == b.C && fusable(v, b.C);
            x = b.out; x_bf = b.out_bf;
        }
        {   const int Pq = st.kind == 0 ? v->res[st.idx].rh * v->res[st.idx].rw : st.kind == 1 ? v->attn[st.idx].rh * v->attn[st.idx].rw : v->ups[st.idx].rh * v->ups[st.idx].rw;
            const int Cq = st.kind == 0 ? v->res[st.idx].Cout : st.kind == 1 ? v->attn[st.idx].C : v->ups[st.idx].C;
            vq_trace(v, "fwd", si, x, (size_t)Pq * Cq, s); }
    }
    v->x_last = x;
    const int PH = v->H * v->W;
    if ((r = gn_fwd(v, v->norm_out, x, PH, 1, s, sr))) return r;
    if ((r = conv3_fwd(v, v->conv_out, v->a, false, v->H, v->W, false, nullptr, v->y, v->conv_out.CoP, s, nullptr, nullptr, true))) return r;
    return prx_image_head_fwd(v->y, v->conv_out.CoP, img, 1, v->out_ch, PH, s);
}

// Diagnostic: copy one intermediate of the last forward (fp32) to dst.  stage -2: quantised latent, -1: conv_in output,
// 0..n-1: output of decoder stage i, n: conv_out output [H*W,4], n+1: the forward GroupNorm sums (doubles, as floats).
long long prx_vqgan_debug_stage_impl(PrxVqgan* v, int stage, float* dst, long long max_floats, hipStream_t s) {
    const void* src = nullptr; long long n = 0;
    PRX_REQUIRE(!v->lean || stage == -2 || stage >= (int)v->stages.size(), "vqgan debug_stage: the stage outputs are 16-bit in the lean layout (PRX_LEAN=0 for fp32 streams)");
    const int ns = (int)v->stages.size();
    if (stage == -2) { src = v->zq; n = (long long)v->h0 * v->w0 * v->D; }
    else if (stage == -1) { src = v->h_in; n = (long long)v->h0 * v->w0 * v->conv_in.Cout; }
    else if (stage >= 0 && stage < ns) {
        const Stage& st = v->stages[stage];
        if (st.kind == 0) { const ResBlock& b = v->res[st.idx]; src = b.out; n = (long long)b.rh * b.rw * b.Cout; }
        else if (st.kind == 1) { const AttnBlock& b = v->attn[st.idx]; src = b.out; n = (long long)b.rh * b.rw * b.C; }
        else { const UpBlock& b = v->ups[st.idx]; src = b.out; n = (long long)b.rh * b.rw * b.C; }
    } else if (stage == ns) { src = v->y; n = (long long)v->H * v->W * v->conv_out.CoP; }
    else if (stage == ns + 1) { src = reinterpret_cast<const float*>(v->all_stats); n = (long long)v->n_gn * 64 * 2; }
    else return -1;
    if (n > max_floats) n = max_floats;
    if (hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
    return n;
}

// g_img: NCHW [1, out_ch, H, W] -> dz: NCHW [1, zc, h0, w0] (straight-through over the quantiser)
int prx_vqgan_backward_impl(PrxVqgan* v, const float* g_img, float* dz, hipStream_t s) {
    const int PH = v->H * v->W;
    int r;
    PRX_REQUIRE(v->x_last != nullptr, "vqgan backward: no forward in flight on this handle");
    PRX_CHECK_HIP(hipMemsetAsync(v->all_stats + (size_t)v->n_gn * 64, 0, sizeof(double) * (size_t)v->n_gn * 64, s));
    // half mode: the whole backward runs under a power-of-two scale S chosen from max|dL/d(image)|; ClampWithGrad only reads signs
    if (v->h16 && (r = prx_grad_scale(g_img, (size_t)v->out_ch * PH, v->gs + 2, 256, prx_grad_target_log2(), v->gs, s))) return r;
    struct GB { void* f; void* b; };      // a gradient stream and its operand twin (one tensor in the exact mode and in the lean layout)
    GB g{v->g0, v->g0b}, t1{v->g1, v->g1b}, t2{v->g2, v->g2b};
    if (v->dycol) {
        // conv_out's dgrad as a row-major product over the im2col'ed head gradient (K = 72 padded to 128: two stages of a fit tile)
        if ((r = prx_image_head_bwd_im2col(v->y, v->conv_out.CoP, g_img, (bf16_t*)v->dycol, 128, v->out_ch, v->H, v->W, s, v->h16, v->gs))) return r;
        GemmDesc d; d.A = v->dycol; d.lda = 128; d.B = v->Wd_col; d.ldb = 128; d.M = PH; d.N = v->conv_out.Cin; d.K = 128;
        out_stream(v, d, t1.f, v->conv_out.Cin, nullptr);
        set_gnb(v, d, &v->norm_out, v->x_last, 1);
        if ((r = vg(v, d, s))) return r;
    } else {
        if ((r = prx_image_head_bwd(v->y, v->conv_out.CoP, g_img, v->f32 ? (float*)v->dy8 : nullptr, v->f32 ? nullptr : (bf16_t*)v->dy8, v->conv_out.CoP, 1,
                                    v->out_ch, PH, s, v->h16, v->gs))) return r;
        if ((r = conv3_bwd(v, v->conv_out, v->dy8, false, v->H, v->W, t1.f, s, nullptr, &v->norm_out, v->x_last, 1))) return r;
    }
    if ((r = gn_bwd(v, v->norm_out, t1.f, v->x_last, nullptr, g.f, g.b, PH, 1, s, true))) return r;
    for (int si = (int)v->stages.size() - 1; si >= 0; --si) {
        const Stage& st = v->stages[si];
        if (st.kind == 0) {
            ResBlock& b = v->res[st.idx];
            const int P = b.rh * b.rw;
            if ((r = conv3_bwd(v, b.c2, g.b, false, b.rh, b.rw, t1.f, s, nullptr, &b.n2, b.h1, 1))) return r;     // d a2
            // d h1: only the following dgrad reads it, as a 16-bit operand -- the fp32 copy is not written (in the exact mode the
            // fp32 buffer IS the operand)
            if ((r = gn_bwd(v, b.n2, t1.f, b.h1, nullptr, v->f32 ? t2.f : nullptr, t2.b, P, 1, s, true))) return r;
            if ((r = conv3_bwd(v, b.c1, t2.b, false, b.rh, b.rw, t1.f, s, nullptr, &b.n1, b.x_in, 1))) return r;   // d a1
            const void* add = g.f;
            if (b.has_sc) {
                GemmDesc d; d.A = g.b; d.lda = b.Cout; d.B = b.sc.WT; d.ldb = b.Cout; d.M = P; d.N = b.Cin; d.K = b.Cout;
                out_stream(v, d, t2.f, b.Cin, nullptr);
                if ((r = vg(v, d, s))) return r;
                add = t2.f;
            }
            // dx = GN1_bwd(d a1) + shortcut grad ; written into a buffer that is neither `add` nor `t1`
            GB& dst = (add == g.f) ? t2 : g;
            if ((r = gn_bwd(v, b.n1, t1.f, b.x_in, add, dst.f, dst.b, P, 1, s, true))) return r;
            if (&dst != &g) std::swap(g, t2);
        } else if (st.kind == 1) {
            AttnBlock& b = v->attn[st.idx];
            const int P = b.rh * b.rw, C = b.C;
            {   GemmDesc d; d.A = g.b; d.lda = C; d.B = b.proj.WT; d.ldb = C; d.M = P; d.N = C; d.K = C;
                d.out_bf16 = v->tA; d.ldc_bf16 = C;                                      // tA = d o [P, C]
                if ((r = vg(v, d, s))) return r; }
            {   GemmDesc d; d.A = v->tA; d.lda = C; d.B = op_off(b.qkvb, 2 * C, v->f32); d.ldb = 3 * C; d.M = P; d.N = P; d.K = C;
                d.out_f32 = v->S; d.ldc_f32 = P;                                         // dP = do v^T
                if ((r = vg(v, d, s))) return r; }
            const int P8 = pad8(P);
            if ((r = zero_if_padded(v, v->tB, P, P, s))) return r;
            if ((r = zero_if_padded(v, v->tC, P, P, s))) return r;
            if ((r = prx_softmax_rows_bwd(b.Pm, P8, v->S, P, 1.f / sqrtf((float)C), v->tB, P8, v->tC, P8, P, P, v->prec, s))) return r;  // tB = dS, tC = dS^T
            {   GemmDesc d; d.A = v->tB; d.lda = P8; d.B = op_off(b.qkvT, (size_t)C * P8, v->f32); d.ldb = P8; d.M = P; d.N = C; d.K = P8;
                d.out_bf16 = v->dqkv; d.ldc_bf16 = 3 * C;                               // dq = dS k   (k^T from the forward's transpose)
                if ((r = vg(v, d, s))) return r; }
            {   GemmDesc d; d.A = v->tC; d.lda = P8; d.B = b.qkvT; d.ldb = P8; d.M = P; d.N = C; d.K = P8;
                d.out_bf16 = op_off(v->dqkv, C, v->f32); d.ldc_bf16 = 3 * C;            // dk = dS^T q  (q^T likewise)
                if ((r = vg(v, d, s))) return r; }
            if ((r = zero_if_padded(v, v->tD, C, P, s))) return r;
            if ((r = prx_transpose_op(v->tA, C, v->tD, P8, P, C, v->f32, s))) return r;                // tD = do^T
            {   GemmDesc d; d.A = b.PT; d.lda = P8; d.B = v->tD; d.ldb = P8; d.M = P; d.N = C; d.K = P8;
                d.out_bf16 = op_off(v->dqkv, 2 * C, v->f32); d.ldc_bf16 = 3 * C;        // dv = P^T do
                if ((r = vg(v, d, s))) return r; }
            {   GemmDesc d; d.A = v->dqkv; d.lda = 3 * C; d.B = b.qkv.WT; d.ldb = 3 * C; d.M = P; d.N = C; d.K = 3 * C;
                out_stream(v, d, t1.f, C, nullptr);                                      // d GN(x)
                set_gnb(v, d, &b.n, b.x_in, 0);
                if ((r = vg(v, d, s))) return r; }
            if ((r = gn_bwd(v, b.n, t1.f, b.x_in, g.f, t2.f, t2.b, P, 0, s, true))) return r;
            std::swap(g, t2);
        } else {
            UpBlock& b = v->ups[st.idx];
            if ((r = conv3_bwd(v, b.c, g.b, false, b.rh, b.rw, t1.f, s))) return r;       // d up(x) at high res
            if ((r = prx_upsample2x_bwd(t1.f, v->lean ? nullptr : (float*)t2.f, v->f32 ? nullptr : (bf16_t*)t2.b, 1, b.rh / 2, b.rw / 2, b.C, s, v->h16,
                                        v->lean))) return r;
            std::swap(g, t2);
        }
        {   const int Pq = st.kind == 0 ? v->res[st.idx].rh * v->res[st.idx].rw : st.kind == 1 ? v->attn[st.idx].rh * v->attn[st.idx].rw : v->ups[st.idx].rh * v->ups[st.idx].rw / 4;
            const int Cq = st.kind == 0 ? v->res[st.idx].Cin : st.kind == 1 ? v->attn[st.idx].C : v->ups[st.idx].C;
            vq_trace(v, "bwd", si, g.f, (size_t)Pq * Cq, s); }
    }
    // conv_in, post_quant_conv, straight-through VQ (ReplaceGrad, vqgan.py:48-58)
    if ((r = conv3_bwd(v, v->conv_in, g.b, false, v->h0, v->w0, v->lean ? nullptr : t1.f, s, v->dpq_bf))) return r;
    const int P0 = v->h0 * v->w0;
    {   GemmDesc d; d.A = v->dpq_bf; d.lda = v->zc; d.B = v->pq.WT; d.ldb = v->zc; d.M = P0; d.N = v->D; d.K = v->zc;
        d.out_f32 = v->dzq; d.ldc_f32 = v->D;
        if (v->h16) d.alpha_dev = v->gs + 1;         // unscale: 1/S (exact, a power of two)
        if ((r = vg(v, d, s))) return r; }
    return prx_nhwc_to_nchw(v->dzq, v->D, dz, 1, v->D, P0, s);
}
