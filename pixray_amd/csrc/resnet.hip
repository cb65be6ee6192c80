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
#include "elementwise.h"
#include "cutouts.h"
#include "prompt_vq.h"
#include "vit.h"  // prx_pack_* helpers
#include <cstdlib>
#include <vector>
#include <memory>
#include <algorithm>

namespace {

__constant__ float r_clip_mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
__constant__ float r_clip_std[3] = {0.26862954f, 0.26130258f, 0.27577711f};

inline int rgrid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 16384); }

// Wf[co][tap*Cin + ci] = w[co][ci][ky][kx];  Wd[ci][tap'*Cout + co] = w[co][ci][2-ky][2-kx]
template <typename TOp>
__global__ __launch_bounds__(256) void rn_pack_conv3x3_kernel(const float* __restrict__ w, void* __restrict__ Wf_, void* __restrict__ Wd_,
                                                              int Cout, int Cin) {
    TOp* Wf = reinterpret_cast<TOp*>(Wf_);
    TOp* Wd = reinterpret_cast<TOp*>(Wd_);
    const size_t total = (size_t)Cout * 9 * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < total) {
            const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 9), co = (int)(i / ((size_t)9 * Cin));
            Wf[i] = op_cvt<TOp>(w[(((size_t)co * Cin + ci) * 3 + tap / 3) * 3 + tap % 3]);
        } else {
            const size_t j = i - total;
            const int co = (int)(j % Cout), tap = (int)((j / Cout) % 9), ci = (int)(j / ((size_t)9 * Cout));
            Wd[j] = op_cvt<TOp>(w[(((size_t)co * Cin + ci) * 3 + (2 - tap / 3)) * 3 + (2 - tap % 3)]);
        }
    }
}

// stem conv1: y = relu(conv3x3 stride 2 pad 1 (normalise(cutouts)) + b) -> NHWC bf16 [n, Ho*Wo, Co]
// normalise = slip.py:21-42: (x - min) / (max - min) with the batch-global min / max, then CLIP mean / std.
// One thread per OUTPUT PIXEL: its 27 normalised inputs are loaded once and all CO channels are produced from the weights
// in LDS ([tap][co], read as wave-wide broadcasts), the CO results leave as 16-byte stores.  (One thread per output
// element -- 27 scattered loads and 27 weight loads each, 40 threads re-reading the same inputs -- took 3.0 ms for
// RN50x4 at 128 cutouts; the accumulation order per channel, (c, ky, kx) after the bias, is unchanged.)
template <typename TOp, int CO>
__global__ __launch_bounds__(256) void stem1_fwd_kernel(const float* __restrict__ cut, const float* __restrict__ mm,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        void* __restrict__ out_, int N, int S) {
    __shared__ __attribute__((aligned(16))) float ws[27 * CO];
    __shared__ float bs[CO];
    for (int i = threadIdx.x; i < 27 * CO; i += 256) {
        const int co = i % CO, t = i / CO;               // t = (c*3 + ky)*3 + kx
        ws[i] = w[co * 27 + t];
    }
    for (int i = threadIdx.x; i < CO; i += 256) bs[i] = b[i];
    __syncthreads();
    TOp* out = reinterpret_cast<TOp*>(out_);
    const int So = S / 2;
    const float mn = mm[0], range = mm[1] - mm[0];
    const float inv = range != 0.f ? 1.f / range : 1.f;
    const size_t total = (size_t)N * So * So;
    constexpr bool STAGE = sizeof(TOp) == 2;             // (the fp32 mode's tile would not fit beside the weights: it keeps its direct stores)
    __shared__ __attribute__((aligned(16))) TOp otile[STAGE ? 256 * CO : 4];
    for (size_t pixb = (size_t)blockIdx.x * blockDim.x; pixb < total; pixb += (size_t)gridDim.x * blockDim.x) {      // workgroup-uniform trip count (barriers inside)
        const size_t pix = pixb + threadIdx.x;
        if (pix < total) {
        const int xo = (int)(pix % So), yo = (int)((pix / So) % So), n = (int)(pix / ((size_t)So * So));
        float v[27];                       // taps outside the image stay 0: adding 0 * w leaves the sum as the skipped tap did
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float im = r_clip_mean[c], is = 1.f / r_clip_std[c];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = 2 * yo + ky - 1, x = 2 * xo + kx - 1, t = (c * 3 + ky) * 3 + kx;
                    const bool ok = y >= 0 && y < S && x >= 0 && x < S;
                    // clamped address + select: a branch around each of the 27 loads made them 27 dependent round trips
                    const int yc = y < 0 ? 0 : (y < S ? y : S - 1), xc = x < 0 ? 0 : (x < S ? x : S - 1);
                    const float raw = cut[(((size_t)n * 3 + c) * S + yc) * S + xc];
                    v[t] = ok ? ((raw - mn) * inv - im) * is : 0.f;
                }
        }
        // the CO results of a pixel go to an LDS tile [256 pixels][CO] and leave as 16-byte pieces of the workgroup's CONTIGUOUS output
        // range: written per thread (8 bytes every 2 CO bytes) the stores were partial sectors -- 673 MB of HBM writes for a 212 MB map
        // (profiles/r06_cfg2_pmc_hbm_traffic.csv)
        TOp* orow = STAGE ? otile + threadIdx.x * CO : out + pix * CO;
#pragma unroll 1
        for (int c0 = 0; c0 < CO; c0 += 4) {
            float a[4] = {bs[c0], bs[c0 + 1], bs[c0 + 2], bs[c0 + 3]};
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const float4 w4 = *reinterpret_cast<const float4*>(&ws[t * CO + c0]);
                a[0] += v[t] * w4.x; a[1] += v[t] * w4.y; a[2] += v[t] * w4.z; a[3] += v[t] * w4.w;
            }
            op_st4(orow, (size_t)c0, fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f));
        }
        }
        if constexpr (STAGE) {
        __syncthreads();
        {
            const size_t pix0 = pixb;                                       // first pixel of this workgroup's tile
            const size_t npix = total - pix0 < 256 ? total - pix0 : 256;
            const int pieces = (int)(npix * CO * sizeof(TOp) / 16);           // CO * sizeof(TOp) % 16 == 0
            const float4* src = reinterpret_cast<const float4*>(otile);
            float4* dst = reinterpret_cast<float4*>(out + pix0 * CO);
            for (int i = threadIdx.x; i < pieces; i += 256) dst[i] = src[i];
        }
        __syncthreads();
        }
    }
}

// dY[n][c][y][x] = sum over (co, ky, kx) with y = 2*yo + ky - 1, x = 2*xo + kx - 1 of g[n][yo][xo][co] * w[co][c][ky][kx]
// (g already masked by the ReLU of the stem conv1 output).  One thread per input PIXEL (its three channels): each of the
// (at most four) taps that reach it reads the CO gradients of one output pixel as contiguous vectors, the weights come
// from LDS ([ky][kx][c][co]); the order of the additions per channel, (ky, kx, co), is the one-thread-per-element kernel's
// (which took 7.4 ms for RN50x4 at 128 cutouts: 2-byte loads of 40 values per tap, three threads per pixel re-reading them).
template <typename TOp, int CO>
__global__ __launch_bounds__(256) void stem1_bwd_kernel(const void* __restrict__ g_, const float* __restrict__ w, float* __restrict__ dY,
                                                        int N, int S, const float* __restrict__ oscale_dev) {
    const float oscale = oscale_dev ? *oscale_dev : 1.f;
    __shared__ float ws[27 * CO];
    for (int i = threadIdx.x; i < 27 * CO; i += 256) {
        const int co = i % CO, c = (i / CO) % 3, kx = (i / (3 * CO)) % 3, ky = i / (9 * CO);
        ws[i] = w[((co * 3 + c) * 3 + ky) * 3 + kx];
    }
    __syncthreads();
    const TOp* g = reinterpret_cast<const TOp*>(g_);
    const int So = S / 2;
    const size_t plane = (size_t)S * S, total = (size_t)N * plane;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(pix % S), y = (int)((pix / S) % S), n = (int)(pix / plane);
        float acc[3] = {0.f, 0.f, 0.f};
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
                const TOp* gp = g + (((size_t)n * So + yo) * So + xo) * CO;
                const float* wp = ws + (ky * 3 + kx) * 3 * CO;
#pragma unroll
                for (int c0 = 0; c0 < CO; c0 += 4) {
                    float gv[4];
                    op_ld4(gp, (size_t)c0, gv);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[0] += gv[q] * wp[c0 + q];
                        acc[1] += gv[q] * wp[CO + c0 + q];
                        acc[2] += gv[q] * wp[2 * CO + c0 + q];
                    }
                }
            }
        }
        float* o = dY + (size_t)n * 3 * plane + (size_t)y * S + x;
        o[0] = acc[0] * oscale; o[plane] = acc[1] * oscale; o[2 * plane] = acc[2] * oscale;    // oscale: 1 / gradient scale
    }
}

// 2x2 average pooling of an NHWC map in the operand format; a thread owns 4 consecutive channels of an output pixel (C % 4 == 0:
// 8- / 16-byte accesses -- these passes move up to 1 GB each and were bound by their 2-byte accesses)
template <typename TOp>
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const void* __restrict__ x_, void* __restrict__ out_, int N, int H, int W, int C) {
    const TOp* x = reinterpret_cast<const TOp*>(x_);
    TOp* out = reinterpret_cast<TOp*>(out_);
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        const size_t pix = idx / C4;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Ho * Wo));
        const size_t p = (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + c;
        const float4 a = op_ld4v(x, p), b = op_ld4v(x, p + C), e = op_ld4v(x, p + (size_t)W * C), f = op_ld4v(x, p + (size_t)W * C + C);
        op_st4(out, pix * C + c, 0.25f * ((a.x + b.x) + (e.x + f.x)), 0.25f * ((a.y + b.y) + (e.y + f.y)),
               0.25f * ((a.z + b.z) + (e.z + f.z)), 0.25f * ((a.w + b.w) + (e.w + f.w)));
    }
}
// backward: dx[n][y][x][c] = 0.25 * g[n][y/2][x/2][c]  (* [mask > 0] when `mask` is given); fp32 and/or operand-format outputs.
// A thread owns 4 consecutive channels of one OUTPUT-gradient pixel and writes its four input pixels (g is read once).
template <typename TOp>
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ g, const void* __restrict__ mask_,
                                                           float* __restrict__ dx_f32, void* __restrict__ dx_bf_, int N, int H, int W, int C) {
    const TOp* mask = reinterpret_cast<const TOp*>(mask_);
    TOp* dx_bf = reinterpret_cast<TOp*>(dx_bf_);
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        const size_t pix = idx / C4;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Ho * Wo));
        float4 v = *reinterpret_cast<const float4*>(g + pix * C + c);
        v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t o = (((size_t)n * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1)) * C + c;
            float4 u = v;
            if (mask) {
                const float4 m = op_ld4v(mask, o);
                if (!(m.x > 0.f)) u.x = 0.f;
                if (!(m.y > 0.f)) u.y = 0.f;
                if (!(m.z > 0.f)) u.z = 0.f;
                if (!(m.w > 0.f)) u.w = 0.f;
            }
            if (dx_f32) *reinterpret_cast<float4*>(dx_f32 + o) = u;
            if (dx_bf) op_st4(dx_bf, o, u.x, u.y, u.z, u.w);
        }
    }
}
// g <- g * [out > 0] in place (fp32) and as bf16
template <typename TOp>
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ g, const void* __restrict__ out_, void* __restrict__ g_bf_, size_t n) {
    const TOp* out = reinterpret_cast<const TOp*>(out_);
    TOp* g_bf = reinterpret_cast<TOp*>(g_bf_);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = ((float)out[i] > 0.f) ? g[i] : 0.f;
        g[i] = v;
        g_bf[i] = op_cvt<TOp>(v);
    }
}
// AttentionPool2d tokens: t[n][0] = mean_p x[n][p] + pos[0]; t[n][1+p] = x[n][p] + pos[1+p]   (x fp32 [n, P, C]) -> bf16
template <typename TOp>
__global__ __launch_bounds__(256) void tokens_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, void* __restrict__ t_,
                                                         int N, int P, int C) {
    TOp* t = reinterpret_cast<TOp*>(t_);
    const size_t total = (size_t)N * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C), n = (int)(idx / C);
        const float* xp = x + (size_t)n * P * C + c;
        TOp* tp = t + (size_t)n * (P + 1) * C + c;
        float sum = 0.f;
        for (int p = 0; p < P; ++p) {
            const float v = xp[(size_t)p * C];
            sum += v;
            tp[(size_t)(p + 1) * C] = op_cvt<TOp>(v + pos[(size_t)(p + 1) * C + c]);
        }
        tp[0] = op_cvt<TOp>(sum / (float)P + pos[c]);
    }
}
// dx[n][p] = dt[n][1+p] + dt[n][0] / P   (fp32 + bf16 twin)
template <typename TOp>
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
template <typename TOp>
__global__ __launch_bounds__(256) void tok0_gather_kernel(const void* __restrict__ t_, void* __restrict__ out_, int N, int T, int C) {
    const TOp* t = reinterpret_cast<const TOp*>(t_);
    TOp* out = reinterpret_cast<TOp*>(out_);
    const size_t total = (size_t)N * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
        out[idx] = t[(idx / C) * (size_t)T * C + idx % C];
}
template <typename TOp>
__global__ __launch_bounds__(256) void tok0_scatter_kernel(const void* __restrict__ g0_, void* __restrict__ dt_, int N, int T, int C) {
    const TOp* g0 = reinterpret_cast<const TOp*>(g0_);
    TOp* dt = reinterpret_cast<TOp*>(dt_);
    const size_t total = (size_t)N * T * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t nt = idx / C;
        const int tok = (int)(nt % T), n = (int)(nt / T);
        dt[idx] = tok == 0 ? g0[(size_t)n * C + c] : (TOp)0.f;
    }
}

// `void*` members: operand-precision buffers (bf16 | fp32, PrxResNet::f32)
struct RConv1 { int Cin, Cout; void *W, *WT; float* b; };
struct RConv3 { int Cin, Cout; void *Wf, *Wd; float* b; };
struct RBlock {
    int Cin, planes, Hin, stride; bool has_ds;
    RConv1 c1, c3, ds; RConv3 c2;
    // saved forward activations (bf16 post-ReLU) and the fp32 block output
    void *a1, *a2, *p2, *xp, *out_bf; float* out_f32;
    const void* xin_bf; const float* xin_f32;
};

}  // namespace

struct PrxResNet {
    int res, width, heads, out_dim, max_n, C, G, T, cur_n;
    int prec;         // PRX_PREC_*
    int f32, h16;     // derived: operands are fp32 / the 16-bit operand format is IEEE half
    // The lean layout (half mode, PRX_RN_LEAN, as vit.hip's PRX_LEAN): the residual stream of the Bottleneck stack and its
    // gradient live in IEEE half only -- the reference's own activation precision (CLIP's convert_weights runs the
    // tower in half).  The conv3 / downsample GEMMs then write 2 bytes per element instead of 6 and read a 2-byte
    // identity, the conv1 dgrad reads a 2-byte identity gradient and writes 2 bytes instead of 6: these 1x1 GEMMs are
    // bound by exactly those streams.  The last block keeps its fp32 output (the attention pool's token mean reads it).
    int lean;
    float* gs;        // half mode: device {S, 1/S} = the power-of-two scale of the backward in flight (common.h) + 64 partials; else null
    GemmCtx gctx;     // this handle's engine state
    std::vector<void*> allocs;
    float *w1, *b1;                      // stem conv1 (fp32, BN folded)
    RConv3 s2, s3;
    std::vector<RBlock> blocks;
    float* pos; void *Win, *WinT, *Wc, *WcT; float *bin, *bc;
    // activations
    void *s1, *s2a, *s3a, *s0_bf; float* s0_f32;     // stem outputs; s0 = pooled stem output (layer1 input)
    void *tok, *qkv, *att, *o0, *dtok, *dqkv, *do0; float *lse, *e, *de, *dtokf;
    float *gA, *gB, *tf; void *tb1, *tb2, *gbf;      // backward ping-pong / temporaries
    void *gb2, *gb3;                                 // lean layout: second gradient stream, downsample-branch gradient
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
int ralloc_op(PrxResNet* r, void** p, size_t count) {   // `count` operand elements
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * op_esz(r->f32)));
    r->allocs.push_back(q);
    *p = q;
    return 0;
}
#define RALLOC_OP(ptr, count) do { int _e = ralloc_op(r, &(ptr), (count)); if (_e) return _e; } while (0)
// launch an operand-typed kernel template for this handle's precision
#define RLAUNCH(kernel, total, ...)                                                                                      \
    do {                                                                                                                 \
        PRX_OP_DISPATCH(r->f32, r->h16, TO_, hipLaunchKernelGGL(kernel<TO_>, dim3(rgrid(total)), dim3(256), 0, s, __VA_ARGS__)); \
        PRX_LAUNCH_CHECK();                                                                                              \
    } while (0)
// the stem kernels keep all CO = width/2 output channels of a pixel in one thread: CO is a template parameter
// (RN50: 32, RN50x4: 40, RN50x16: 48, RN50x64: 64)
#define STEM_CASE(kernel, CO_, total, ...)                                                                               \
    case CO_:                                                                                                            \
        PRX_OP_DISPATCH(r->f32, r->h16, TO_, hipLaunchKernelGGL((kernel<TO_, CO_>), dim3(rgrid(total)), dim3(256), 0, s, __VA_ARGS__)); \
        break;
#define STEM_LAUNCH(kernel, total, co, ...)                                                                              \
    do {                                                                                                                 \
        switch (co) {                                                                                                    \
            STEM_CASE(kernel, 8, total, __VA_ARGS__) STEM_CASE(kernel, 16, total, __VA_ARGS__)                           \
            STEM_CASE(kernel, 32, total, __VA_ARGS__) STEM_CASE(kernel, 40, total, __VA_ARGS__)                          \
            STEM_CASE(kernel, 48, total, __VA_ARGS__) STEM_CASE(kernel, 64, total, __VA_ARGS__)                          \
            default: PRX_REQUIRE(false, "resnet stem: width/2 = %d is not one of 8, 16, 32, 40, 48, 64", co);            \
        }                                                                                                                \
        PRX_LAUNCH_CHECK();                                                                                              \
    } while (0)
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
    RALLOC_OP(c.W, (size_t)Cout * Cin); RALLOC_OP(c.WT, (size_t)Cout * Cin);
    int e;
    if ((e = prx_pack_op(w, c.W, (size_t)Cout * Cin, r->prec, s))) return e;
    if ((e = prx_pack_transpose_op(w, c.WT, Cout, Cin, r->prec, s))) return e;
    return rcopy(r, &c.b, b, Cout, s);
}
int mk3(PrxResNet* r, RConv3& c, int Cin, int Cout, RCur& cur, hipStream_t s) {
    const float *w, *b; RNEXT(cur, w); RNEXT(cur, b);
    c.Cin = Cin; c.Cout = Cout;
    RALLOC_OP(c.Wf, (size_t)Cout * 9 * Cin); RALLOC_OP(c.Wd, (size_t)Cout * 9 * Cin);
    RLAUNCH(rn_pack_conv3x3_kernel, (size_t)1024 * 256, w, c.Wf, c.Wd, Cout, Cin);
    return rcopy(r, &c.b, b, Cout, s);
}
int rg(PrxResNet* r, GemmDesc& d, hipStream_t s) {
    if (r->f32) { d.f32 = 1; d.a_is_f32 = 0; }
    d.h16 = r->h16;
    return prx_gemm_launch(d, r->ws, r->ws_bytes, s, &r->gctx);
}

// 1x1 conv forward / dgrad as GEMMs
int lin(PrxResNet* r, const void* A, int M, int K, const void* Bt, int N, const float* bias, const float* resid, int act,
        const void* aux, float* of, void* ob, hipStream_t s, int resid16 = 0) {
    GemmDesc d; d.A = A; d.lda = K; d.B = Bt; d.ldb = K; d.M = M; d.N = N; d.K = K;
    d.bias_n = bias; d.resid = resid; d.ldr = N; d.act = act; d.aux = aux; d.ldaux = N; d.row16 = resid16 ? 1 : 0;
    d.out_f32 = of; d.ldc_f32 = N; d.out_bf16 = ob; d.ldc_bf16 = N;
    return rg(r, d, s);
}
int conv3(PrxResNet* r, const void* x, int NB, int H, int Cin, const void* Bt, int Cout, const float* bias, int act,
          const void* aux, float* of, void* ob, hipStream_t s) {
    GemmDesc d; d.A = x; d.a_mode = PRX_A_CONV3X3; d.lda = Cin; d.B = Bt; d.ldb = 9 * Cin; d.M = NB * H * H; d.N = Cout; d.K = 9 * Cin;
    d.H = H; d.W = H; d.Cin = Cin; d.bias_n = bias; d.act = act; d.aux = aux; d.ldaux = Cout;
    d.out_f32 = of; d.ldc_f32 = Cout; d.out_bf16 = ob; d.ldc_bf16 = Cout;
    return rg(r, d, s);
}
}  // namespace

// weights (fp32 device, BatchNorm folded on the host, pixray_amd/weights.py::fold_clip_resnet_params): stem1 {w,b}, stem2,
// stem3, per Bottleneck c1, c2, c3 [, ds], attnpool positional_embedding, in_proj {w [3C,C], b}, c_proj {w, b}
GemmCtx* prx_resnet_gemm_ctx_impl(PrxResNet* r) { return r ? &r->gctx : nullptr; }

int prx_resnet_create_impl(PrxResNet** out, int res, int width, const int* layers, int heads, int out_dim, int max_n,
                           int precision, const float* const* w, int n_w, hipStream_t s) {
    PRX_REQUIRE(prec_valid(precision), "resnet_create: unknown precision %d", precision);
    PRX_REQUIRE(res % 32 == 0 && width % 16 == 0 && max_n >= 1, "resnet_create: unsupported geometry (res %d width %d)", res, width);
    PRX_REQUIRE(width * 32 == heads * 64, "resnet_create: the attention pool needs head dim 64 (width %d heads %d)", width, heads);
    PrxResNet* r = new PrxResNet();
    std::unique_ptr<PrxResNet> guard(r);
    r->prec = precision; r->f32 = prec_is_f32(precision); r->h16 = prec_is_h16(precision);
    r->gs = nullptr;
    { const char* ev = getenv("PRX_RN_LEAN"); r->lean = (r->h16 && !(ev && atoi(ev) == 0)) ? 1 : 0; }
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
            RALLOC_OP(k.a1, Min * planes); RALLOC_OP(k.a2, Min * planes);
            k.p2 = k.a2; k.xp = nullptr;
            if (k.stride > 1) { RALLOC_OP(k.p2, Mout * planes); RALLOC_OP(k.xp, Mout * inplanes); }
            RALLOC_OP(k.out_bf, Mout * planes * 4);
            k.out_f32 = nullptr;
            if (!r->lean || (li == 3 && b == layers[li] - 1)) RALLOC(k.out_f32, Mout * planes * 4);
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
    RALLOC_OP(r->Win, (size_t)3 * C * C); RALLOC_OP(r->WinT, (size_t)3 * C * C);
    if ((e = prx_pack_op(win, r->Win, (size_t)3 * C * C, r->prec, s))) return e;
    if ((e = prx_pack_transpose_op(win, r->WinT, 3 * C, C, r->prec, s))) return e;
    if ((e = rcopy(r, &r->bin, bin, 3 * C, s))) return e;
    RALLOC_OP(r->Wc, (size_t)out_dim * C); RALLOC_OP(r->WcT, (size_t)out_dim * C);
    if ((e = prx_pack_op(wc, r->Wc, (size_t)out_dim * C, r->prec, s))) return e;
    if ((e = prx_pack_transpose_op(wc, r->WcT, out_dim, C, r->prec, s))) return e;
    if ((e = rcopy(r, &r->bc, bc, out_dim, s))) return e;
    const size_t S2 = (size_t)(res / 2) * (res / 2), S4 = (size_t)(res / 4) * (res / 4);
    RALLOC_OP(r->s1, N * S2 * (width / 2)); RALLOC_OP(r->s2a, N * S2 * (width / 2)); RALLOC_OP(r->s3a, N * S2 * width);
    RALLOC_OP(r->s0_bf, N * S4 * width); RALLOC(r->s0_f32, N * S4 * width);
    RALLOC_OP(r->tok, N * T * C); RALLOC_OP(r->qkv, N * T * 3 * C); RALLOC_OP(r->att, N * T * C); RALLOC_OP(r->o0, N * C);
    RALLOC_OP(r->dtok, N * T * C); RALLOC_OP(r->dqkv, N * T * 3 * C); RALLOC_OP(r->do0, N * C); RALLOC(r->dtokf, N * T * C);
    RALLOC(r->lse, N * heads * T); RALLOC(r->e, N * out_dim); RALLOC(r->de, N * out_dim);
    RALLOC(r->gA, maxMC); RALLOC(r->tf, maxMC);
    r->gB = nullptr; r->gb2 = r->gb3 = nullptr;
    if (r->lean) { RALLOC_OP(r->gb2, maxMC); RALLOC_OP(r->gb3, maxMC); } else RALLOC(r->gB, maxMC);
    RALLOC_OP(r->tb1, maxMC); RALLOC_OP(r->tb2, maxMC); RALLOC_OP(r->gbf, maxMC);
    RALLOC(r->dY, N * 3 * (size_t)res * res); RALLOC(r->mm_part, 2 * 1024);
    if (r->h16) RALLOC(r->gs, 2 + 64);
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
    STEM_LAUNCH(stem1_fwd_kernel, (size_t)n * S2 * S2, wh, cutouts, mm, r->w1, r->b1, r->s1, n, S);
    if ((e = conv3(r, r->s1, n, S2, wh, r->s2.Wf, wh, r->s2.b, PRX_ACT_RELU, nullptr, nullptr, r->s2a, s))) return e;
    if ((e = conv3(r, r->s2a, n, S2, wh, r->s3.Wf, w, r->s3.b, PRX_ACT_RELU, nullptr, nullptr, r->s3a, s))) return e;
    RLAUNCH(avgpool2_fwd_kernel, (size_t)n * S4 * S4 * w / 4, r->s3a, r->s0_bf, n, S2, S2, w);
    // the identity path of layer1.0 goes through its downsample conv, so no fp32 copy of the stem output is needed
    const void* x_bf = r->s0_bf; const float* x_f32 = nullptr;
    const int lean = r->lean;
    for (RBlock& k : r->blocks) {
        const int H = k.Hin, Ho = H / k.stride, Min = n * H * H, Mout = n * Ho * Ho, p = k.planes;
        k.xin_bf = x_bf; k.xin_f32 = x_f32;
        if ((e = lin(r, x_bf, Min, k.Cin, k.c1.W, p, k.c1.b, nullptr, PRX_ACT_RELU, nullptr, nullptr, k.a1, s))) return e;
        if ((e = conv3(r, k.a1, n, H, p, k.c2.Wf, p, k.c2.b, PRX_ACT_RELU, nullptr, nullptr, k.a2, s))) return e;
        if (k.stride > 1) {
            RLAUNCH(avgpool2_fwd_kernel, (size_t)Mout * p / 4, k.a2, k.p2, n, H, H, p);
        }
        // the identity: fp32, or (lean) the 16-bit stream itself -- tb1 holds the downsample output (free in the forward)
        const float* idn = lean ? (const float*)x_bf : x_f32;
        if (k.has_ds) {
            const void* xi = x_bf;
            if (k.stride > 1) {
                RLAUNCH(avgpool2_fwd_kernel, (size_t)Mout * k.Cin / 4, x_bf, k.xp, n, H, H, k.Cin);
                xi = k.xp;
            }
            if ((e = lin(r, xi, Mout, k.Cin, k.ds.W, 4 * p, k.ds.b, nullptr, PRX_ACT_NONE, nullptr, lean ? nullptr : r->tf,
                         lean ? r->tb1 : nullptr, s))) return e;
            idn = lean ? (const float*)r->tb1 : r->tf;
        }
        PRX_REQUIRE(idn != nullptr, "resnet: identity path without an fp32 input");
        if ((e = lin(r, k.p2, Mout, p, k.c3.W, 4 * p, k.c3.b, idn, PRX_ACT_RELU, nullptr, k.out_f32, k.out_bf, s, lean))) return e;
        x_bf = k.out_bf; x_f32 = k.out_f32;
    }
    // attention pool
    const int C = r->C, T = r->T, P = T - 1;
    RLAUNCH(tokens_fwd_kernel, (size_t)n * C, x_f32, r->pos, r->tok, n, P, C);
    if ((e = lin(r, r->tok, n * T, C, r->Win, 3 * C, r->bin, nullptr, PRX_ACT_NONE, nullptr, nullptr, r->qkv, s))) return e;
    if (r->f32) { if ((e = prx_mha_fwd_f32((const float*)r->qkv, (float*)r->att, r->lse, n, T, C, r->heads, s))) return e; }
    else if ((e = prx_mha_fwd_gen((const bf16_t*)r->qkv, (bf16_t*)r->att, r->lse, n, T, C, r->heads, s, r->h16))) return e;
    RLAUNCH(tok0_gather_kernel, (size_t)n * C, r->att, r->o0, n, T, C);
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
        // half mode: the whole backward runs scaled by a power of two S chosen from max|d e|; stem1_bwd unscales
        // (S multiplies d e in fp32 before the load converts it to half: see vit.hip)
        if (r->h16) {
            if ((e = prx_grad_scale(r->de, (size_t)n * r->out_dim, r->gs + 2, 64, prx_grad_target_log2(), r->gs, s))) return e;
            if ((e = prx_scale_dev(r->de, (size_t)n * r->out_dim, r->gs, s))) return e;
        }
        if ((e = rg(r, d, s))) return e; }
    RLAUNCH(tok0_scatter_kernel, (size_t)n * T * C, r->do0, r->dtok, n, T, C);
    if (r->f32) { if ((e = prx_mha_bwd_f32((const float*)r->qkv, (const float*)r->att, (const float*)r->dtok, r->lse, (float*)r->dqkv, n, T, C, r->heads, s))) return e; }
    else if ((e = prx_mha_bwd_gen((const bf16_t*)r->qkv, (const bf16_t*)r->att, (const bf16_t*)r->dtok, r->lse, (bf16_t*)r->dqkv, n, T, C, r->heads, s, r->h16))) return e;
    if ((e = lin(r, r->dqkv, n * T, 3 * C, r->WinT, C, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->dtokf, nullptr, s))) return e;
    float* g = r->gA; float* g2 = r->gB;
    const int lean = r->lean;
    void* gb = r->gbf; void* gb_alt = r->gb2;     // lean layout: the 16-bit gradient stream and the buffer its successor goes to
    RLAUNCH(tokens_bwd_kernel, (size_t)n * P * C, r->dtokf, g, n, P, C);
    for (int bi = (int)r->blocks.size() - 1; bi >= 0; --bi) {
        RBlock& k = r->blocks[bi];
        const int H = k.Hin, Ho = H / k.stride, Min = n * H * H, Mout = n * Ho * Ho, p = k.planes;
        // through the final ReLU: g (fp32, in place) and its operand-precision twin.  Only the last block needs a pass of its
        // own; for every other block the GEMM that formed g (conv1 dgrad + identity gradient of the block after it) applied
        // this block's mask in its epilogue and wrote both copies (PRX_ACT_RELUMASK_POST below): 25 of RN50x4's 26 passes
        // over the residual-stream gradient (12 bytes per element each) are gone.
        if (bi == (int)r->blocks.size() - 1)
            RLAUNCH(relu_mask_kernel, (size_t)Mout * 4 * p, g, k.out_bf, gb, (size_t)Mout * 4 * p);
        // main branch: conv3 (1x1) dgrad [-> avgpool bwd] -> ReLU mask of a2
        if (k.stride > 1) {
            if ((e = lin(r, gb, Mout, 4 * p, k.c3.WT, p, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->tf, nullptr, s))) return e;
            RLAUNCH(avgpool2_bwd_kernel, (size_t)Min * p / 16, r->tf, k.a2, (float*)nullptr, r->tb1, n, H, H, p);
        } else {
            if ((e = lin(r, gb, Mout, 4 * p, k.c3.WT, p, nullptr, nullptr, PRX_ACT_MUL_RELUMASK, k.a2, nullptr, r->tb1, s))) return e;
        }
        // conv2 (3x3) dgrad -> ReLU mask of a1
        if ((e = conv3(r, r->tb1, n, H, p, k.c2.Wd, p, nullptr, PRX_ACT_MUL_RELUMASK, k.a1, nullptr, r->tb2, s))) return e;
        // identity branch
        const float* gid = lean ? (const float*)gb : g;
        if (k.has_ds) {
            if (k.stride > 1) {
                if ((e = lin(r, gb, Mout, 4 * p, k.ds.WT, k.Cin, nullptr, nullptr, PRX_ACT_NONE, nullptr, r->tf, nullptr, s))) return e;
                RLAUNCH(avgpool2_bwd_kernel, (size_t)Min * k.Cin / 16, r->tf, (const void*)nullptr, lean ? (float*)nullptr : g2,
                                   lean ? r->gb3 : (void*)nullptr, n, H, H, k.Cin);
                gid = lean ? (const float*)r->gb3 : g2;
            } else {
                if ((e = lin(r, gb, Mout, 4 * p, k.ds.WT, k.Cin, nullptr, nullptr, PRX_ACT_NONE, nullptr, lean ? nullptr : r->tf,
                             lean ? r->gb3 : nullptr, s))) return e;
                gid = lean ? (const float*)r->gb3 : r->tf;
            }
        }
        if (lean) {
            // conv1 (1x1) dgrad + the 16-bit identity gradient -> the next 16-bit stream (never the buffer being added); block 0
            // hands the stem an fp32 gradient
            if (bi > 0) {
                if ((e = lin(r, r->tb2, Min, p, k.c1.WT, k.Cin, nullptr, gid, PRX_ACT_RELUMASK_POST, r->blocks[bi - 1].out_bf, nullptr,
                             gb_alt, s, 1))) return e;
                std::swap(gb, gb_alt);
            } else if ((e = lin(r, r->tb2, Min, p, k.c1.WT, k.Cin, nullptr, gid, PRX_ACT_NONE, nullptr, g, nullptr, s, 1))) return e;
            continue;
        }
        // conv1 (1x1) dgrad + identity gradient -> gradient w.r.t. the block input
        float* dst = (gid == g2) ? g : g2;        // never write over the residual being added
        if (gid == r->tf) dst = g2;
        // ... masked by the output ReLU of the block below (its out_bf is this block's input), fp32 + operand twin
        if (bi > 0) {
            if ((e = lin(r, r->tb2, Min, p, k.c1.WT, k.Cin, nullptr, gid, PRX_ACT_RELUMASK_POST, r->blocks[bi - 1].out_bf, dst, gb, s))) return e;
        } else if ((e = lin(r, r->tb2, Min, p, k.c1.WT, k.Cin, nullptr, gid, PRX_ACT_NONE, nullptr, dst, nullptr, s))) return e;
        if (dst != g) std::swap(g, g2);
    }
    // stem: avgpool -> relu3 mask -> conv3 dgrad -> relu2 mask -> conv2 dgrad -> relu1 mask -> conv1 input gradient
    const int S = r->res, S2 = S / 2, w = r->width, wh = w / 2;
    RLAUNCH(avgpool2_bwd_kernel, (size_t)n * S2 * S2 * w / 16, g, r->s3a, (float*)nullptr, r->tb1, n, S2, S2, w);
    if ((e = conv3(r, r->tb1, n, S2, w, r->s3.Wd, wh, nullptr, PRX_ACT_MUL_RELUMASK, r->s2a, nullptr, r->tb2, s))) return e;
    if ((e = conv3(r, r->tb2, n, S2, wh, r->s2.Wd, wh, nullptr, PRX_ACT_MUL_RELUMASK, r->s1, nullptr, r->tb1, s))) return e;
    STEM_LAUNCH(stem1_bwd_kernel, (size_t)n * S * S, wh, r->tb1, r->w1, r->dY, n, S, (const float*)(r->h16 ? r->gs + 1 : nullptr));
    return prx_preproc_bwd_reduce(cutouts, mm, r->dY, acc, n, S, s);
}

int prx_resnet_backward_b_impl(PrxResNet* r, const float* cutouts, const float* mm, const double* acc, float* g_cutouts,
                               hipStream_t s) {
    PRX_REQUIRE(r->cur_n >= 1, "resnet backward: no forward in flight on this handle");
    return prx_preproc_bwd_apply(cutouts, mm, r->dY, acc, g_cutouts, r->cur_n, r->res, s);
}
