// VGG16 `features` up to relu5_3 on MI355X, forward and input-gradient backward, for the StyleLoss plugin
// (reference Losses/StyleLoss.py:24-47: torchvision vgg16 features, frozen, the nine captured ReLU outputs
// [1,3,6,8,11,13,15,22,29] = relu1_1 1_2 2_1 2_2 3_1 3_2 3_3 4_3 5_3).  Batch 1, any H x W up to the size given at
// creation (STROTSS runs the extractor on a pyramid of image sizes).
//
// Three operand precisions (prx.h PRX_PREC_*): bf16, IEEE half (with the power-of-two gradient scale of common.h in the
// backward), exact f32.
// Layout: activations are NHWC 16-bit (the implicit-GEMM engine's operand), every conv3x3+bias+ReLU is ONE launch of
// the MFMA engine with the ReLU in its epilogue; captured layers also get their fp32 NHWC feature map written by the
// same epilogue.  The 3-channel input is padded to 8 channels.  2x2/2 max pooling keeps a 2-bit argmax per output.
// Backward per conv, top down: G = (gradient from the layer above, through the pool's argmax when there is one)
// + (the captured feature's gradient), masked by the layer's own ReLU -> bf16; then the dgrad conv (flipped weight pack)
// on the engine.  The forward's activations live in a caller-owned workspace, so several forward passes can be alive
// at once (STROTSS accumulates its loss over many extractor calls before one backward).
#include "vgg.h"
#include "gemm.h"
#include "elementwise.h"
#include <vector>
#include <algorithm>

namespace {

constexpr int NCONV = 13;
constexpr int NFEAT = 9;
const int kCin[NCONV] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
const int kCout[NCONV] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int kStage[NCONV] = {0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4};       // resolution level (input size >> stage)
const int kFeat[NCONV] = {0, 1, 2, 3, 4, 5, 6, -1, -1, 7, -1, -1, 8};     // captured feature index or -1
inline int cpad(int c) { return c < 8 ? 8 : c; }
inline int vgrid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 16384); }

struct VLayout {
    int h[5], w[5];
    size_t x0;                 // padded input, bf16 [H*W*8]
    size_t act[NCONV];         // bf16 post-ReLU activations
    size_t pooled[4];          // bf16 pooled maps feeding stages 1..4
    size_t arg[4];             // uint8 argmax of those pools
    size_t total;
};
VLayout vlayout(int H, int W, int f32) {
    const size_t es = op_esz(f32);       // operand element size: bf16, or fp32 in the exact mode
    VLayout L;
    L.h[0] = H; L.w[0] = W;
    for (int s = 1; s < 5; ++s) { L.h[s] = L.h[s - 1] / 2; L.w[s] = L.w[s - 1] / 2; }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    L.x0 = take((size_t)H * W * 8 * es);
    for (int l = 0; l < NCONV; ++l) L.act[l] = take((size_t)L.h[kStage[l]] * L.w[kStage[l]] * kCout[l] * es);
    const int pool_c[4] = {64, 128, 256, 512};
    for (int p = 0; p < 4; ++p) {
        L.pooled[p] = take((size_t)L.h[p + 1] * L.w[p + 1] * pool_c[p] * es);
        L.arg[p] = take((size_t)L.h[p + 1] * L.w[p + 1] * pool_c[p]);
    }
    L.total = off;
    return L;
}

// Wf[co][tap*CiP + ci] = w[co][ci][ky][kx] (ci zero-padded to CiP);  Wd[ci][tap'*Cout + co] = w[co][ci][2-ky][2-kx]
template <typename TOp>
__global__ __launch_bounds__(256) void vgg_pack_kernel(const float* __restrict__ w, void* __restrict__ Wf_, void* __restrict__ Wd_,
                                                       int Cout, int Cin, int CiP) {
    TOp* Wf = reinterpret_cast<TOp*>(Wf_);
    TOp* Wd = reinterpret_cast<TOp*>(Wd_);
    const size_t total = (size_t)Cout * 9 * CiP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += (size_t)gridDim.x * blockDim.x) {
        if (i < total) {
            const int ci = (int)(i % CiP), tap = (int)((i / CiP) % 9), co = (int)(i / ((size_t)9 * CiP));
            Wf[i] = op_cvt<TOp>(ci < Cin ? w[(((size_t)co * Cin + ci) * 3 + tap / 3) * 3 + tap % 3] : 0.f);
        } else {
            const size_t j = i - total;
            const int co = (int)(j % Cout), tap = (int)((j / Cout) % 9), ci = (int)(j / ((size_t)9 * Cout));
            Wd[j] = op_cvt<TOp>(ci < Cin ? w[(((size_t)co * Cin + ci) * 3 + (2 - tap / 3)) * 3 + (2 - tap % 3)] : 0.f);
        }
    }
}

// x [3][H][W] fp32 -> NHWC bf16 with 8 channels (5 zeros)
template <typename TOp>
__global__ __launch_bounds__(256) void vgg_input_kernel(const float* __restrict__ x, void* __restrict__ out_, int HW) {
    TOp* out = reinterpret_cast<TOp*>(out_);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        op_st4(out, (size_t)p * 8, x[p], x[HW + p], x[2 * (size_t)HW + p], 0.f);
        op_st4(out, (size_t)p * 8 + 4, 0.f, 0.f, 0.f, 0.f);
    }
}
// dgrad of conv1_1 [HW][8] fp32 -> g_x [3][H][W]
// `unscale`: device scalar 1/S of the half mode's gradient scale, or null
__global__ __launch_bounds__(256) void vgg_input_grad_kernel(const float* __restrict__ d, float* __restrict__ gx, int HW,
                                                             const float* __restrict__ unscale) {
    const float u = unscale ? *unscale : 1.f;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        gx[p] = u * d[(size_t)p * 8]; gx[HW + p] = u * d[(size_t)p * 8 + 1]; gx[2 * (size_t)HW + p] = u * d[(size_t)p * 8 + 2];
    }
}

// 2x2 stride-2 max pooling (floor), first maximum wins on ties like torch's max_pool2d backward
template <typename TOp>
__global__ __launch_bounds__(256) void vgg_maxpool_kernel(const void* __restrict__ x_, void* __restrict__ out_, unsigned char* __restrict__ arg,
                                                          int H, int W, int C) {
    const TOp* x = reinterpret_cast<const TOp*>(x_);
    TOp* out = reinterpret_cast<TOp*>(out_);
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)Ho * Wo * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t pix = idx / C;
        const int xo = (int)(pix % Wo), yo = (int)(pix / Wo);
        const TOp* p = x + (((size_t)2 * yo) * W + 2 * xo) * C + c;
        float best = (float)p[0]; int a = 0;
        float v = (float)p[C]; if (v > best) { best = v; a = 1; }
        v = (float)p[(size_t)W * C]; if (v > best) { best = v; a = 2; }
        v = (float)p[(size_t)W * C + C]; if (v > best) { best = v; a = 3; }
        out[idx] = op_cvt<TOp>(best);
        arg[idx] = (unsigned char)a;
    }
}

// gpre = [act > 0] * (above + gcap) as bf16, for one conv layer's output map [H*W][C].
//   above: fp32 gradient w.r.t. what the next conv read -- this map itself (arg == nullptr), or its 2x2 max-pooled
//   version [H/2*W/2][C] routed through `arg`; may be null.  gcap: fp32 gradient of the captured feature; may be null.
//   gscale: device scalar S of the half mode (common.h): the captured gradients enter the backward as S * gcap, in fp32 and
//   BEFORE the conversion to half (`above` already carries S); null otherwise.
template <typename TOp>
__global__ __launch_bounds__(256) void vgg_combine_kernel(const float* __restrict__ above, const unsigned char* __restrict__ arg,
                                                          const float* __restrict__ gcap, const void* __restrict__ act_,
                                                          void* __restrict__ gpre_, int H, int W, int C, const float* __restrict__ gscale) {
    const float S = gscale ? *gscale : 1.f;
    const TOp* act = reinterpret_cast<const TOp*>(act_);
    TOp* gpre = reinterpret_cast<TOp*>(gpre_);
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)H * W * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (above) {
            if (arg) {
                const int c = (int)(idx % C);
                const size_t pix = idx / C;
                const int x = (int)(pix % W), y = (int)(pix / W);
                if ((y >> 1) < Ho && (x >> 1) < Wo) {
                    const size_t o = ((size_t)(y >> 1) * Wo + (x >> 1)) * C + c;
                    if (arg[o] == (unsigned char)((y & 1) * 2 + (x & 1))) v = above[o];
                }
            } else {
                v = above[idx];
            }
        }
        if (gcap) v += S * gcap[idx];
        gpre[idx] = op_cvt<TOp>(((float)act[idx] > 0.f) ? v : 0.f);
    }
}

struct VConv { int Cin, CiP, Cout; void *Wf, *Wd; float* b; };   // weight packs at operand precision

}  // namespace

struct PrxVgg16 {
    int max_h, max_w;
    int prec;         // PRX_PREC_*
    int f32, h16;     // derived: operands are fp32 / the 16-bit operand format is IEEE half
    float* gs;        // half mode: device {S, 1/S} of the backward in flight + partial maxima; else null
    GemmCtx gctx;     // this handle's engine state
    std::vector<void*> allocs;
    VConv conv[NCONV];
    float *dA, *dB, *ws;       // dgrad ping-pong (fp32), split-K scratch
    void* gpre;       // operand precision
    size_t ws_bytes;
};

namespace {
template <typename Tp>
int valloc(PrxVgg16* v, Tp** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(Tp)));
    v->allocs.push_back(q);
    *p = (Tp*)q;
    return 0;
}
#define VALLOC(ptr, count) do { int _e = valloc(v, &(ptr), (count)); if (_e) return _e; } while (0)
int valloc_op(PrxVgg16* v, void** p, size_t count) {
    void* q = nullptr;
    PRX_CHECK_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * op_esz(v->f32)));
    v->allocs.push_back(q);
    *p = q;
    return 0;
}
#define VLAUNCH(kernel, total, ...)                                                                                      \
    PRX_OP_DISPATCH(v->f32, v->h16, TV, hipLaunchKernelGGL(kernel<TV>, dim3(vgrid(total)), dim3(256), 0, s, __VA_ARGS__))

int vconv(PrxVgg16* v, const void* x, int H, int W, int Cin, const void* Bt, int Cout, const float* bias, int act,
          float* of, void* ob, hipStream_t s) {
    GemmDesc d; d.A = x; d.a_mode = PRX_A_CONV3X3; d.lda = Cin; d.B = Bt; d.ldb = 9 * Cin; d.M = H * W; d.N = Cout; d.K = 9 * Cin;
    d.H = H; d.W = W; d.Cin = Cin; d.bias_n = bias; d.act = act;
    d.out_f32 = of; d.ldc_f32 = Cout; d.out_bf16 = ob; d.ldc_bf16 = Cout;
    d.f32 = v->f32; d.h16 = v->h16;
    return prx_gemm_launch(d, v->ws, v->ws_bytes, s, &v->gctx);
}
}  // namespace

long long prx_vgg16_workspace_bytes_impl(int H, int W, int precision) { return (H < 16 || W < 16) ? -1 : (long long)vlayout(H, W, prec_is_f32(precision)).total; }
GemmCtx* prx_vgg16_gemm_ctx_impl(PrxVgg16* v) { return v ? &v->gctx : nullptr; }

int prx_vgg16_feature_shape_impl(int H, int W, int k, int* h, int* w, int* c) {
    PRX_REQUIRE(k >= 0 && k < NFEAT && H >= 16 && W >= 16, "vgg16_feature_shape: feature %d of a %dx%d input", k, H, W);
    const VLayout L = vlayout(H, W, 0);
    for (int l = 0; l < NCONV; ++l)
        if (kFeat[l] == k) { *h = L.h[kStage[l]]; *w = L.w[kStage[l]]; *c = kCout[l]; }
    return 0;
}

// weights: torchvision order, {weight [Cout,Cin,3,3], bias [Cout]} for the 13 convs (fp32, device)
int prx_vgg16_create_impl(PrxVgg16** out, const float* const* weights, int n_weights, int max_h, int max_w, int precision, hipStream_t s) {
    PRX_REQUIRE(prec_valid(precision), "vgg16_create: unknown precision %d", precision);
    PRX_REQUIRE(out && weights && n_weights == 2 * NCONV, "vgg16_create: expected %d weight tensors, got %d", 2 * NCONV, n_weights);
    PRX_REQUIRE(max_h >= 16 && max_w >= 16, "vgg16_create: the input must be at least 16x16 (got %dx%d)", max_h, max_w);
    PrxVgg16* v = new PrxVgg16();
    v->max_h = max_h; v->max_w = max_w; v->prec = precision; v->f32 = prec_is_f32(precision); v->h16 = prec_is_h16(precision);
    v->gs = nullptr;
    auto fail = [&](int e) { prx_vgg16_destroy_impl(v); return e; };
    for (int l = 0; l < NCONV; ++l) {
        VConv& c = v->conv[l];
        c.Cin = kCin[l]; c.CiP = cpad(kCin[l]); c.Cout = kCout[l];
        const size_t n = (size_t)c.Cout * 9 * c.CiP;
        int e;
        if ((e = valloc_op(v, &c.Wf, n)) || (e = valloc_op(v, &c.Wd, n)) || (e = valloc(v, &c.b, (size_t)c.Cout))) return fail(e);
        VLAUNCH(vgg_pack_kernel, (size_t)1024 * 256, weights[2 * l], c.Wf, c.Wd, c.Cout, c.Cin, c.CiP);
        if (hipGetLastError() != hipSuccess) return fail(-1);
        if (hipMemcpyAsync(c.b, weights[2 * l + 1], sizeof(float) * c.Cout, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(-1);
    }
    const size_t big = (size_t)max_h * max_w * 64;
    int e;
    if ((e = valloc(v, &v->dA, big)) || (e = valloc(v, &v->dB, big)) || (e = valloc_op(v, &v->gpre, big))) return fail(e);
    v->ws_bytes = (size_t)64 << 20;
    if ((e = valloc(v, &v->ws, v->ws_bytes / sizeof(float)))) return fail(e);
    if (v->h16 && (e = valloc(v, &v->gs, (size_t)2 + NFEAT * 256))) return fail(e);
    *out = v;
    return 0;
}

void prx_vgg16_destroy_impl(PrxVgg16* v) {
    if (!v) return;
    for (void* p : v->allocs) (void)hipFree(p);
    delete v;
}

// x: [3,H,W] fp32 (already in the extractor's input space); feats[k]: fp32 NHWC [h_k*w_k, C_k] or null (not wanted)
int prx_vgg16_forward_impl(PrxVgg16* v, const float* x, int H, int W, void* workspace, float* const* feats, hipStream_t s) {
    PRX_REQUIRE(v && x && workspace && feats, "vgg16_forward: null argument");
    PRX_REQUIRE(H >= 16 && W >= 16 && H <= v->max_h && W <= v->max_w, "vgg16_forward: input %dx%d outside [16x16, %dx%d]", H, W, v->max_h, v->max_w);
    const VLayout L = vlayout(H, W, v->f32);
    char* base = (char*)workspace;
    void* cur = base + L.x0;
    VLAUNCH(vgg_input_kernel, (size_t)H * W, x, cur, H * W);
    PRX_LAUNCH_CHECK();
    for (int l = 0; l < NCONV; ++l) {
        const VConv& c = v->conv[l];
        const int st = kStage[l];
        if (l > 0 && kStage[l - 1] != st) {      // pool the previous activation
            const int p = st - 1;
            void* pooled = base + L.pooled[p];
            VLAUNCH(vgg_maxpool_kernel, (size_t)L.h[st] * L.w[st] * c.Cin, (const void*)cur, pooled,
                    (unsigned char*)(base + L.arg[p]), L.h[st - 1], L.w[st - 1], c.Cin);
            PRX_LAUNCH_CHECK();
            cur = pooled;
        }
        void* act = base + L.act[l];
        int e = vconv(v, cur, L.h[st], L.w[st], c.CiP, c.Wf, c.Cout, c.b, PRX_ACT_RELU, kFeat[l] >= 0 ? feats[kFeat[l]] : nullptr, act, s);
        if (e) return e;
        cur = act;
    }
    return 0;
}

// g_feats[k]: fp32 NHWC gradient of feature k, or null; g_x: [3,H,W] fp32 (overwritten)
int prx_vgg16_backward_impl(PrxVgg16* v, int H, int W, const void* workspace, const float* const* g_feats, float* g_x, hipStream_t s) {
    PRX_REQUIRE(v && workspace && g_feats && g_x, "vgg16_backward: null argument");
    PRX_REQUIRE(H >= 16 && W >= 16 && H <= v->max_h && W <= v->max_w, "vgg16_backward: input %dx%d outside [16x16, %dx%d]", H, W, v->max_h, v->max_w);
    const VLayout L = vlayout(H, W, v->f32);
    const char* base = (const char*)workspace;
    if (v->h16) {
        // half mode: the whole backward runs under a power-of-two scale S chosen on the device from max |g_feats| (exact: every
        // op below is linear in the incoming gradients, the ReLU masks and pool routes only read the forward); vgg_combine
        // multiplies the captured gradients by S before the conversion to half, vgg_input_grad removes it
        const float* gp[NFEAT]; size_t gn[NFEAT];
        for (int l = 0; l < NCONV; ++l)
            if (kFeat[l] >= 0) { gp[kFeat[l]] = g_feats[kFeat[l]]; gn[kFeat[l]] = (size_t)L.h[kStage[l]] * L.w[kStage[l]] * kCout[l]; }
        bool any = false;
        for (int k = 0; k < NFEAT; ++k) any = any || gp[k] != nullptr;
        if (any) { int e = prx_grad_scale_multi(gp, gn, NFEAT, v->gs + 2, 256, prx_grad_target_log2(), v->gs, s); if (e) return e; }
    }
    const float* above = nullptr;     // gradient w.r.t. the input of conv l+1
    float* bufs[2] = {v->dA, v->dB};
    int flip = 0;
    for (int l = NCONV - 1; l >= 0; --l) {
        const VConv& c = v->conv[l];
        const int st = kStage[l];
        const float* gcap = kFeat[l] >= 0 ? g_feats[kFeat[l]] : nullptr;
        if (!above && !gcap) continue;                           // nothing reaches this layer yet
        const bool pooled_above = above && l + 1 < NCONV && kStage[l + 1] != st;
        const unsigned char* arg = pooled_above ? (const unsigned char*)(base + L.arg[st]) : nullptr;
        const size_t n = (size_t)L.h[st] * L.w[st] * c.Cout;
        VLAUNCH(vgg_combine_kernel, n, above, arg, gcap, (const void*)(base + L.act[l]), v->gpre, L.h[st], L.w[st], c.Cout,
                (const float*)v->gs);
        PRX_LAUNCH_CHECK();
        float* dst = bufs[flip]; flip ^= 1;
        int e = vconv(v, v->gpre, L.h[st], L.w[st], c.Cout, c.Wd, c.CiP, nullptr, PRX_ACT_NONE, dst, nullptr, s);
        if (e) return e;
        above = dst;
    }
    if (above) {
        hipLaunchKernelGGL(vgg_input_grad_kernel, dim3(vgrid((size_t)H * W)), dim3(256), 0, s, above, g_x, H * W,
                           (const float*)(v->gs ? v->gs + 1 : nullptr));
        PRX_LAUNCH_CHECK();
    } else {
        PRX_CHECK_HIP(hipMemsetAsync(g_x, 0, sizeof(float) * 3 * (size_t)H * W, s));
    }
    return 0;
}
