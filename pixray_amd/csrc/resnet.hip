// CLIP ModifiedResNet visual tower on MI355X (call site slip.py:62-66 -> `model.encode_image`,
// [UPSTREAM openai/CLIP clip/model.py: ModifiedResNet / Bottleneck / AttentionPool2d]), forward and activation-gradient
// backward (weights frozen, slip.py:176; eval-mode BatchNorm is folded into the convolutions on the host).
//
// Layout: every feature map NHWC = [n*H*W, C] row-major (a GEMM A matrix), bf16 post-ReLU activations (they are both
// the next operand and the ReLU mask of the backward), fp32 block outputs for the identity path.  1x1 convs are plain
// GEMMs, 3x3 convs implicit GEMMs on the MFMA engine with ReLU / ReLU-mask epilogues; the 3-channel stride-2 stem conv
// (0.3 % of the FLOPs) and its input gradient are direct fp32 kernels that also apply slip.py's preprocessing.
#include "resnet.h"
#include "gemm.h"
#include "attention.h"
#include "cutouts.h"
#include "prompt_vq.h"
#include "vit.h"  // prx_pack_* helpers
#include <vector>
#include <memory>
#include <algorithm>

namespace {

__constant__ float r_clip_mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
__constant__ float r_clip_std[3] = {0.26862954f, 0.26130258f, 0.27577711f};

inline int rgrid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 16384); }

// Wf[co][tap*Cin + ci] = w[co][ci][ky][kx];  Wd[ci][tap'*Cout + co] = w[co][ci][2-ky][2-kx]
__global__ __launch_bounds__(256) void rn_pack_conv3x3_kernel(const float* __restrict__ w, bf16_t* __restrict__ Wf, bf16_t* __restrict__ Wd,
                                                              int Cout, int Cin) {
    const size_t total = (size_t)Cout * 9 * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < total) {
            const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 9), co = (int)(i / ((size_t)9 * Cin));
            Wf[i] = (bf16_t)w[(((size_t)co * Cin + ci) * 3 + tap / 3) * 3 + tap % 3];
        } else {
            const size_t j = i - total;
            const int co = (int)(j % Cout), tap = (int)((j / Cout) % 9), ci = (int)(j / ((size_t)9 * Cout));
            Wd[j] = (bf16_t)w[(((size_t)co * Cin + ci) * 3 + (2 - tap / 3)) * 3 + (2 - tap % 3)];
        }
    }
}

// stem conv1: y = relu(conv3x3 stride 2 pad 1 (normalise(cutouts)) + b) -> NHWC bf16 [n, Ho*Wo, Co]
// normalise = slip.py:21-42: (x - min) / (max - min) with the batch-global min / max, then CLIP mean / std
__global__ __launch_bounds__(256) void stem1_fwd_kernel(const float* __restrict__ cut, const float* __restrict__ mm,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        bf16_t* __restrict__ out, int N, int S, int Co) {
    const int So = S / 2;
    const float mn = mm[0], range = mm[1] - mm[0];
    const float inv = range != 0.f ? 1.f / range : 1.f;
    const size_t total = (size_t)N * So * So * Co;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Co);
        const size_t pix = idx / Co;
        const int xo = (int)(pix % So), yo = (int)((pix / So) % So), n = (int)(pix / ((size_t)So * So));
        float acc = b[co];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float im = r_clip_mean[c], is = 1.f / r_clip_std[c];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int y = 2 * yo + ky - 1;
                if (y < 0 || y >= S) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int x = 2 * xo + kx - 1;
                    if (x < 0 || x >= S) continue;
                    const float v = ((cut[(((size_t)n * 3 + c) * S + y) * S + x] - mn) * inv - im) * is;
                    acc += v * w[((co * 3 + c) * 3 + ky) * 3 + kx];
                }
            }
        }
        out[idx] = (bf16_t)fmaxf(acc, 0.f);
    }
}

// dY[n][c][y][x] = sum over (co, ky, kx) with y = 2*yo + ky - 1, x = 2*xo + kx - 1 of g[n][yo][xo][co] * w[co][c][ky][kx]
// (g already masked by the ReLU of the stem conv1 output)
__global__ __launch_bounds__(256) void stem1_bwd_kernel(const bf16_t* __restrict__ g, const float* __restrict__ w, float* __restrict__ dY,
                                                        int N, int S, int Co) {
    const int So = S / 2;
    const size_t total = (size_t)N * 3 * S * S;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % S), y = (int)((idx / S) % S), c = (int)((idx / ((size_t)S * S)) % 3);
        const int n = (int)(idx / ((size_t)3 * S * S));
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int t = y + 1 - ky;
            if (t < 0 || (t & 1)) continue;
            const int yo = t >> 1;
            if (yo >= So) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int u = x + 1 - kx;
                if (u < 0 || (u & 1)) continue;
                const int xo = u >> 1;
                if (xo >= So) continue;
                const bf16_t* gp = g + (((size_t)n * So + yo) * So + xo) * Co;
                for (int co = 0; co < Co; ++co) acc += (float)gp[co] * w[((co * 3 + c) * 3 + ky) * 3 + kx];
            }
        }
        dY[idx] = acc;
    }
}

// 2x2 average pooling of an NHWC bf16 map
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t pix = idx / C;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Ho * Wo));
        const bf16_t* p = x + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + c;
        const float v = ((float)p[0] + (float)p[C]) + ((float)p[(size_t)W * C] + (float)p[(size_t)W * C + C]);
        out[idx] = (bf16_t)(0.25f * v);
    }
}
// backward: dx[n][y][x][c] = 0.25 * g[n][y/2][x/2][c]  (* [mask > 0] when `mask` is given); fp32 and/or bf16 outputs
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ g, const bf16_t* __restrict__ mask,
                                                           float* __restrict__ dx_f32, bf16_t* __restrict__ dx_bf, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * H * W * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t pix = idx / C;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((size_t)H * W));
        float v = 0.25f * g[(((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c];
        if (mask && !((float)mask[idx] > 0.f)) v = 0.f;
        if (dx_f32) dx_f32[idx] = v;
        if (dx_bf) dx_bf[idx] = (bf16_t)v;
    }
}
// g <- g * [out > 0] in place (fp32) and as bf16
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ g, const bf16_t* __restrict__ out, bf16_t* __restrict__ g_bf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = ((float)out[i] > 0.f) ? g[i] : 0.f;
        g[i] = v;
        g_bf[i] = (bf16_t)v;
    }
}
// AttentionPool2d tokens: t[n][0] = mean_p x[n][p] + pos[0]; t[n][1+p] = x[n][p] + pos[1+p]   (x fp32 [n, P, C]) -> bf16
__global__ __launch_bounds__(256) void tokens_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, bf16_t* __restrict__ t,
                                                         int N, int P, int C) {
    const size_t total = (size_t)N * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C), n = (int)(idx / C);
        const float* xp = x + (size_t)n * P * C + c;
        bf16_t* tp = t + (size_t)n * (P + 1) * C + c;
        float sum = 0.f;
        for (int p = 0; p < P; ++p) {
            const float v = xp[(size_t)p * C];
            sum += v;
            tp[(size_t)(p + 1) * C] = (bf16_t)(v + pos[(size_t)(p + 1) * C + c]);
        }
        tp[0] = (bf16_t)(sum / (float)P + pos[c]);
    }
}
// dx[n][p] = dt[n][1+p] + dt[n][0] / P   (fp32 + bf16 twin)
__global__ __launch_bounds__(256) void tokens_bwd_kernel(const float* __restrict__ dt, float* __restrict__ dx, int N, int P, int C) {
    const size_t total = (size_t)N * P * C;
    const float ip = 1.f / (float)P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t np = idx / C;
        const int p = (int)(np % P), n = (int)(np / P);
        const float* base = dt + (size_t)n * (P + 1) * C + c;
        dx[idx] = base[(size_t)(p + 1) * C] + base[0] * ip;
    }
}
// rows of token 0: out[n][c] = t[n][0][c]; and the reverse (zero everywhere else)
__global__ __launch_bounds__(256) void tok0_gather_kernel(const bf16_t* __restrict__ t, bf16_t* __restrict__ out, int N, int T, int C) {
    const size_t total = (size_t)N * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
        out[idx] = t[(idx / C) * (size_t)T * C + idx % C];
}
__global__ __launch_bounds__(256) void tok0_scatter_kernel(const bf16_t* __restrict__ g0, bf16_t* __restrict__ dt, int N, int T, int C) {
    const size_t total = (size_t)N * T * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t nt = idx / C;
        const int tok = (int)(nt % T), n = (int)(nt / T);
        dt[idx] = tok == 0 ? g0[(size_t)n * C + c] : (bf16_t)0.f;
    }
}

struct RConv1 { int Cin, Cout; bf16_t *W, *WT; float* b; };
struct RConv3 { int Cin, Cout; bf16_t *Wf, *Wd; float* b; };
struct RBlock {
    int Cin, planes, Hin, stride; bool has_ds;
    RConv1 c1, c3, ds; RConv3 c2;
    // saved forward activations (bf16 post-ReLU) and the fp32 block output
    bf16_t *a1, *a2, *p2, *xp, *out_bf; float* out_f32;
    const bf16_t* xin_bf; const float* xin_f32;
};

}  // namespace

struct PrxResNet {
    int res, width, heads, out_dim, max_n, C, G, T, cur_n;
    std::vector<void*> allocs;
    float *w1, *b1;                      // stem conv1 (fp32, BN folded)
    RConv3 s2, s3;
    std::vector<RBlock> blocks;
    float* pos; bf16_t *Win, *WinT, *Wc, *WcT; float *bin, *bc;
    // activations
    bf16_t *s1, *s2a, *s3a, *s0_bf; float* s0_f32;     // stem outputs; s0 = pooled stem output (layer1 input)
    bf16_t *tok, *qkv, *att, *o0, *dtok, *dqkv, *do0; float *lse, *e, *de, *dtokf;
    float *gA, *gB, *tf; bf16_t *tb1, *tb2, *gbf;      // backward ping-pong / temporaries
    float *dY, *mm_part, *ws; size_t ws_bytes;
};

namespace {
template <typename Tp>
int ralloc(PrxResNet* r, Tp** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(Tp)));
    r->allocs.push_back(q);
    *p = (Tp*)q;
    return 0;
}
#define RALLOC(ptr, count) do { int _e = ralloc(r, &(ptr), (count)); if (_e) return _e; } while (0)
struct RCur { const float* const* w; int n, pos; };
#define RNEXT(cur, dst) do { PRX_REQUIRE((cur).pos < (cur).n, "resnet_create: weight list too short"); (dst) = (cur).w[(cur).pos++]; } while (0)

int rcopy(PrxResNet* r, float** dst, const float* src, size_t n, hipStream_t s) {
    RALLOC(*dst, n);
    PRX_CHECK_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int mk1(PrxResNet* r, RConv1& c, int Cin, int Cout, RCur& cur, hipStream_t s) {
    const float *w, *b; RNEXT(cur, w); RNEXT(cur, b);
    c.Cin = Cin; c.Cout = Cout;
    RALLOC(c.W, (size_t)Cout * Cin); RALLOC(c.WT, (size_t)Cout * Cin);
    int e;
    if ((e = prx_pack_bf16(w, c.W, (size_t)Cout * Cin, s))) return e;
    if ((e = prx_pack_transpose_bf16(w, c.WT, Cout, Cin, s))) return e;
    return rcopy(r, &c.b, b, Cout, s);
}
int mk3(PrxResNet* r, RConv3& c, int Cin, int Cout, RCur& cur, hipStream_t s) {
    const float *w, *b; RNEXT(cur, w); RNEXT(cur, b);
    c.Cin = Cin; c.Cout = Cout;
    RALLOC(c.Wf, (size_t)Cout * 9 * Cin); RALLOC(c.Wd, (size_t)Cout * 9 * Cin);
    hipLaunchKernelGGL(rn_pack_conv3x3_kernel, dim3(1024), dim3(256), 0, s, w, c.Wf, c.Wd, Cout, Cin);
    PRX_LAUNCH_CHECK();
    return rcopy(r, &c.b, b, Cout, s);
}
int rg(PrxResNet* r, GemmDesc& d, hipStream_t s) { return prx_gemm_launch(d, r->ws, r->ws_bytes, s); }

// 1x1 conv forward / dgrad as GEMMs
int lin(PrxResNet* r, const bf16_t* A, int M, int K, const bf16_t* Bt, int N, const float* bias, const float* resid, int act,
        const bf16_t* aux, float* of, bf16_t* ob, hipStream_t s) {
    GemmDesc d; d.A = A; d.lda = K; d.B = Bt; d.ldb = K; d.M = M; d.N = N; d.K = K;
    d.bias_n = bias; d.resid = resid; d.ldr = N; d.act = act; d.aux = aux; d.ldaux = N;
    d.out_f32 = of; d.ldc_f32 = N; d.out_bf16 = ob; d.ldc_bf16 = N;
    return rg(r, d, s);
}
int conv3(PrxResNet* r, const bf16_t* x, int NB, int H, int Cin, const bf16_t* Bt, int Cout, const float* bias, int act,
          const bf16_t* aux, float* of, bf16_t* ob, hipStream_t s) {
    GemmDesc d; d.A = x; d.a_mode = PRX_A_CONV3X3; d.lda = Cin; d.B = Bt; d.ldb = 9 * Cin; d.M = NB * H * H; d.N = Cout; d.K = 9 * Cin;
    d.H = H; d.W = H; d.Cin = Cin; d.bias_n = bias; d.act = act; d.aux = aux; d.ldaux = Cout;
    d.out_f32 = of; d.ldc_f32 = Cout; d.out_bf16 = ob; d.ldc_bf16 = Cout;
    return rg(r, d, s);
}
}  // namespace

// weights (fp32 device, BatchNorm folded on the host, pixray_amd/weights.py::fold_clip_resnet_params): stem1 {w,b}, stem2,
// stem3, per Bottleneck c1, c2, c3 [, ds], attnpool positional_embedding, in_proj {w [3C,C], b}, c_proj {w, b}
int prx_resnet_create_impl(PrxResNet** out, int res, int width, const int* layers, int heads, int out_dim, int max_n,
                           const float* const* w, int n_w, hipStream_t s) {
    PRX_REQUIRE(res % 32 == 0 && width % 16 == 0 && max_n >= 1, "resnet_create: unsupported geometry (res %d width %d)", res, width);
    PRX_REQUIRE(width * 32 == heads * 64, "resnet_create: the attention pool needs head dim 64 (width %d heads %d)", width, heads);
    PrxResNet* r = new PrxResNet();
    std::unique_ptr<PrxResNet> guard(r);
    r->res = res; r->width = width; r->heads = heads; r->out_dim = out_dim; r->max_n = max_n; r->cur_n = 0;
    r->C = width * 32; r->G = res / 32; r->T = r->G * r->G + 1;
    RCur cur{w, n_w, 0};
    int e;
    {   const float *ww, *bb; RNEXT(cur, ww); RNEXT(cur, bb);
        if ((e = rcopy(r, &r->w1, ww, (size_t)(width / 2) * 27, s))) return e;
        if ((e = rcopy(r, &r->b1, bb, width / 2, s))) return e; }
    if ((e = mk3(r, r->s2, width / 2, width / 2, cur, s))) return e;
    if ((e = mk3(r, r->s3, width / 2, width, cur, s))) return e;
    const size_t N = (size_t)max_n;
    int H = res / 4, inplanes = width;
    size_t maxMC = N * (size_t)(res / 2) * (res / 2) * width;      // largest [M, C] map: stem conv3 output
    for (int li = 0; li < 4; ++li) {
        const int planes = width << li;
        for (int b = 0; b < layers[li]; ++b) {
            RBlock k{};
            k.Cin = inplanes; k.planes = planes; k.Hin = H; k.stride = (li > 0 && b == 0) ? 2 : 1;
            k.has_ds = k.stride > 1 || inplanes != planes * 4;
            if ((e = mk1(r, k.c1, inplanes, planes, cur, s))) return e;
            if ((e = mk3(r, k.c2, planes, planes, cur, s))) return e;
            if ((e = mk1(r, k.c3, planes, planes * 4, cur, s))) return e;
            if (k.has_ds && (e = mk1(r, k.ds, inplanes, planes * 4, cur, s))) return e;
            const size_t Min = N * H * H, Ho = H / k.stride, Mout = N * Ho * Ho;
            RALLOC(k.a1, Min * planes); RALLOC(k.a2, Min * planes);
            k.p2 = k.a2; k.xp = nullptr;
            if (k.stride > 1) { RALLOC(k.p2, Mout * planes); RALLOC(k.xp, Mout * inplanes); }
            RALLOC(k.out_bf, Mout * planes * 4); RALLOC(k.out_f32, Mout * planes * 4);
            maxMC = std::max(maxMC, std::max(Min * (size_t)std::max(inplanes, planes), Mout * (size_t)planes * 4));
            r->blocks.push_back(k);
            inplanes = planes * 4; H = (int)Ho;
        }
    }
    PRX_REQUIRE(inplanes == r->C && H == r->G, "resnet_create: internal geometry mismatch");
    const float *pos, *win, *bin, *wc, *bc;
    RNEXT(cur, pos); RNEXT(cur, win); RNEXT(cur, bin); RNEXT(cur, wc); RNEXT(cur, bc);
    PRX_REQUIRE(cur.pos == n_w, "resnet_create: %d weight tensors given, %d consumed", n_w, cur.pos);
    const int C = r->C, T = r->T;
    if ((e = rcopy(r, &r->pos, pos, (size_t)T * C, s))) return e;
    RALLOC(r->Win, (size_t)3 * C * C); RALLOC(r->WinT, (size_t)3 * C * C);
    if ((e = prx_pack_bf16(win, r->Win, (size_t)3 * C * C, s))) return e;
    if ((e = prx_pack_transpose_bf16(win, r->WinT, 3 * C, C, s))) return e;
    if ((e = rcopy(r, &r->bin, bin, 3 * C, s))) return e;
    RALLOC(r->Wc, (size_t)out_dim * C); RALLOC(r->WcT, (size_t)out_dim * C);
    if ((e = prx_pack_bf16(wc, r->Wc, (size_t)out_dim * C, s))) return e;
    if ((e = prx_pack_transpose_bf16(wc, r->WcT, out_dim, C, s))) return e;
    if ((e = rcopy(r, &r->bc, bc, out_dim, s))) return e;
    const size_t S2 = (size_t)(res / 2) * (res / 2), S4 = (size_t)(res / 4) * (res / 4);
    RALLOC(r->s1, N * S2 * (width / 2)); RALLOC(r->s2a, N * S2 * (width / 2)); RALLOC(r->s3a, N * S2 * width);
    RALLOC(r->s0_bf, N * S4 * width); RALLOC(r->s0_f32, N * S4 * width);
    RALLOC(r->tok, N * T * C); RALLOC(r->qkv, N * T * 3 * C); RALLOC(r->att, N * T * C); RALLOC(r->o0, N * C);
    RALLOC(r->dtok, N * T * C); RALLOC(r->dqkv, N * T * 3 * C); RALLOC(r->do0, N * C); RALLOC(r->dtokf, N * T * C);
    RALLOC(r->lse, N * heads * T); RALLOC(r->e, N * out_dim); RALLOC(r->de, N * out_dim);
    RALLOC(r->gA, maxMC); RALLOC(r->gB, maxMC); RALLOC(r->tf, maxMC);
    RALLOC(r->tb1, maxMC); RALLOC(r->tb2, maxMC); RALLOC(r->gbf, maxMC);
    RALLOC(r->dY, N * 3 * (size_t)res * res); RALLOC(r->mm_part, 2 * 1024);
    r->ws_bytes = (size_t)64 << 20;
    RALLOC(r->ws, r->ws_bytes / sizeof(float));
    *out = guard.release();
    return 0;
}

void prx_resnet_destroy_impl(PrxResNet* r) {
    if (!r) return;
    for (void* p : r->allocs) (void)hipFree(p);
    delete r;
}

int prx_resnet_minmax_impl(PrxResNet* r, const float* cutouts, int n, float* mm, hipStream_t s) {
    PRX_REQUIRE(n >= 1 && n <= r->max_n, "resnet: batch %d exceeds handle capacity %d", n, r->max_n);
    return prx_minmax(cutouts, (size_t)n * 3 * r->res * r->res, r->mm_part, 1024, mm, s);
}

int prx_resnet_forward_impl(PrxResNet* r, const float* cutouts, int n, const float* mm, float* embeds, hipStream_t s) {
    PRX_REQUIRE(n >= 1 && n <= r->max_n, "resnet: batch %d exceeds handle capacity %d", n, r->max_n);
    const int S = r->res, S2 = S / 2, S4 = S / 4, w = r->width, wh = w / 2;
    int e;
    r->cur_n = n;
    hipLaunchKernelGGL(stem1_fwd_kernel, dim3(rgrid((size_t)n * S2 * S2 * wh)), dim3(256), 0, s, cutouts, mm, r->w1, r->b1, r->s1, n, S, wh);
    PRX_LAUNCH_CHECK();
    if ((e = conv3(r, r->s1, n, S2, wh, r->s2.Wf, wh, r->s2.b, PRX_ACT_RELU, nullptr, nullptr, r->s2a, s))) return e;
    if ((e = conv3(r, r->s2a, n, S2, wh, r->s3.Wf, w, r->s3.b, PRX_ACT_RELU, nullptr, nullptr, r->s3a, s))) return e;
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(rgrid((size_t)n * S4 * S4 * w)), dim3(256), 0, s, r->s3a, r->s0_bf, n, S2, S2, w);
    PRX_LAUNCH_CHECK();
    // the identity path of layer1.0 goes through its downsample conv, so no fp32 copy of the stem output is needed
    const bf16_t* x_bf = r->s0_bf; const float* x_f32 = nullptr;
    for (RBlock& k : r->blocks) {
        const int H = k.Hin, Ho = H / k.stride, Min = n * H * H, Mout = n * Ho * Ho, p = k.planes;
        k.xin_bf = x_bf; k.xin_f32 = x_f32;
        if ((e = lin(r, x_bf, Min, k.Cin, k.c1.W, p, k.c1.b, nullptr, PRX_ACT_RELU, nullptr, nullptr, k.a1, s))) return e;
        if ((e = conv3(r, k.a1, n, H, p, k.c2.Wf, p, k.c2.b, PRX_ACT_RELU, nullptr, nullptr, k.a2, s))) return e;
        if (k.stride > 1) {
            hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(rgrid((size_t)Mout * p)), dim3(256), 0, s, k.a2, k.p2, n, H, H, p);
            PRX_LAUNCH_CHECK();
        }
        const float* idn = x_f32;
        if (k.has_ds) {
            const bf16_t* xi = x_bf;
            if (k.stride > 1) {
                hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(rgrid((size_t)Mout * k.Cin)), dim3(256), 0, s, x_bf, k.xp, n, H, H, k.Cin);
                PRX_LAUNCH_CHECK();
                xi = k.xp;
            }
            if ((e = lin(r, xi, Mout, k.Cin, k.ds.W, 4 * p, k.ds.b, nullptr, PRX_ACT_NONE, nullptr, r->tf, nullptr, s))) return e;
            idn = r->tf;
        }
        PRX_REQUIRE(idn != nullptr, "resnet: identity path without an fp32 input");
        if ((e = lin(r, k.p2, Mout, p, k.c3.W, 4 * p, k.c3.b, idn, PRX_ACT_RELU, nullptr, k.out_f32, k.out_bf, s))) return e;
        x_bf = k.out_bf; x_f32 = k.out_f32;
    }
    // attention pool
    const int C = r->C, T = r->T, P = T - 1;
    hipLaunchKernelGGL(tokens_fwd_kernel, dim3(rgrid((size_t)n * C)), dim3(256), 0, s, x_f32, r->pos, r->tok, n, P, C);
    PRX_LAUNCH_CHECK();
    if ((e = lin(r, r->tok, n * T, C, r->Win, 3 * C, r->bin, nullptr, PRX_ACT_NONE, nullptr, nullptr, r->qkv, s))) return e;
    if ((e = prx_mha_fwd_gen(r->qkv, r->att, r->lse, n, T, C, r->heads, s))) return e;
    hipLaunchKernelGGL(tok0_gather_kernel, dim3(rgrid((size_t)n * C)), dim3(256), 0, s, r->att, r->o0, n, T, C);
    PRX_LAUNCH_CHECK();
    if ((e = lin(r, r->o0, n, C, r->Wc, r->out_dim, r->bc, nullptr, PRX_ACT_NONE, nullptr, r->e, nullptr, s))) return e;
    return prx_l2norm_fwd(r->e, embeds, n, r->out_dim, s);
}

// Backward part A: from d(embeds) down to the gradient w.r.t. the normalised image (dY) and the four sums the
// batch-global min/max renorm needs (to be summed over ranks when the cutout batch is sharded)
int prx_resnet_backward_a_impl(PrxResNet* r, const float* cutouts, const float* mm, const float* d_embeds, double* acc,
                               hipStream_t s) {
    const int n = r->cur_n;
    PRX_REQUIRE(n >= 1, "resnet backward: no forward in flight on this handle");
    const int C = r->C, T = r->T, P = T - 1;
    int e;
    if ((e = prx_l2norm_bwd(r->e, d_embeds, r->de, n, r->out_dim, s))) return e;
    // c_proj dgrad: A fp32 -> the register-staged GEMM converts on load
    {   GemmDesc d; d.A = r->de; d.a_is_f32 = 1; d.lda = r->out_dim; d.B = r->WcT; d.ldb = r->out_dim; d.M = n; d.N = C; d.K = r->out_dim;
        d.out_bf16 = r->do0; d.ldc_bf16 = C;
        if ((e = rg(r, d, s))) return e; }
    hipLaunchKernelGGL(tok0_scatter_kernel, dim3(rgrid((size_t)n * T * C)), dim3(256), 0, s, r->do0, r->dtok, n, T, C);
    PRX_LAUNCH_CHECK();
    if ((e = prx_mha_bwd_gen(r->qkv, r->att, r->dtok, r->lse, r->dqkv, n, T, C, r->heads, s))) return e;
    if ((e = lin(r, r->dqkv, n * T, 3 * C, r->WinT, C, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->dtokf, nullptr, s))) return e;
    float* g = r->gA; float* g2 = r->gB;
    hipLaunchKernelGGL(tokens_bwd_kernel, dim3(rgrid((size_t)n * P * C)), dim3(256), 0, s, r->dtokf, g, n, P, C);
    PRX_LAUNCH_CHECK();
    for (int bi = (int)r->blocks.size() - 1; bi >= 0; --bi) {
        RBlock& k = r->blocks[bi];
        const int H = k.Hin, Ho = H / k.stride, Min = n * H * H, Mout = n * Ho * Ho, p = k.planes;
        // through the final ReLU: g (fp32, in place) and its bf16 twin
        hipLaunchKernelGGL(relu_mask_kernel, dim3(rgrid((size_t)Mout * 4 * p)), dim3(256), 0, s, g, k.out_bf, r->gbf, (size_t)Mout * 4 * p);
        PRX_LAUNCH_CHECK();
        // main branch: conv3 (1x1) dgrad [-> avgpool bwd] -> ReLU mask of a2
        if (k.stride > 1) {
            if ((e = lin(r, r->gbf, Mout, 4 * p, k.c3.WT, p, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->tf, nullptr, s))) return e;
            hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(rgrid((size_t)Min * p)), dim3(256), 0, s, r->tf, k.a2, (float*)nullptr, r->tb1, n, H, H, p);
            PRX_LAUNCH_CHECK();
        } else {
            if ((e = lin(r, r->gbf, Mout, 4 * p, k.c3.WT, p, nullptr, nullptr, PRX_ACT_MUL_RELUMASK, k.a2, nullptr, r->tb1, s))) return e;
        }
        // conv2 (3x3) dgrad -> ReLU mask of a1
        if ((e = conv3(r, r->tb1, n, H, p, k.c2.Wd, p, nullptr, PRX_ACT_MUL_RELUMASK, k.a1, nullptr, r->tb2, s))) return e;
        // identity branch
        const float* gid = g;
        if (k.has_ds) {
            if (k.stride > 1) {
                if ((e = lin(r, r->gbf, Mout, 4 * p, k.ds.WT, k.Cin, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->tf, nullptr, s))) return e;
                hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(rgrid((size_t)Min * k.Cin)), dim3(256), 0, s, r->tf, (const bf16_t*)nullptr, g2,
                                   (bf16_t*)nullptr, n, H, H, k.Cin);
                PRX_LAUNCH_CHECK();
                gid = g2;
            } else {
                if ((e = lin(r, r->gbf, Mout, 4 * p, k.ds.WT, k.Cin, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->tf, nullptr, s))) return e;
                gid = r->tf;
            }
        }
        // conv1 (1x1) dgrad + identity gradient -> gradient w.r.t. the block input
        float* dst = (gid == g2) ? g : g2;        // never write over the residual being added
        if (gid == r->tf) dst = g2;
        if ((e = lin(r, r->tb2, Min, p, k.c1.WT, k.Cin, nullptr, gid, PRX_ACT_NONE, nullptr, dst, nullptr, s))) return e;
        if (dst != g) std::swap(g, g2);
    }
    // stem: avgpool -> relu3 mask -> conv3 dgrad -> relu2 mask -> conv2 dgrad -> relu1 mask -> conv1 input gradient
    const int S = r->res, S2 = S / 2, w = r->width, wh = w / 2;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(rgrid((size_t)n * S2 * S2 * w)), dim3(256), 0, s, g, r->s3a, (float*)nullptr, r->tb1, n, S2, S2, w);
    PRX_LAUNCH_CHECK();
    if ((e = conv3(r, r->tb1, n, S2, w, r->s3.Wd, wh, nullptr, PRX_ACT_MUL_RELUMASK, r->s2a, nullptr, r->tb2, s))) return e;
    if ((e = conv3(r, r->tb2, n, S2, wh, r->s2.Wd, wh, nullptr, PRX_ACT_MUL_RELUMASK, r->s1, nullptr, r->tb1, s))) return e;
    hipLaunchKernelGGL(stem1_bwd_kernel, dim3(rgrid((size_t)n * 3 * S * S)), dim3(256), 0, s, r->tb1, r->w1, r->dY, n, S, wh);
    PRX_LAUNCH_CHECK();
    return prx_preproc_bwd_reduce(cutouts, mm, r->dY, acc, n, S, s);
}

int prx_resnet_backward_b_impl(PrxResNet* r, const float* cutouts, const float* mm, const double* acc, float* g_cutouts,
                               hipStream_t s) {
    PRX_REQUIRE(r->cur_n >= 1, "resnet backward: no forward in flight on this handle");
    return prx_preproc_bwd_apply(cutouts, mm, r->dY, acc, g_cutouts, r->cur_n, r->res, s);
}
