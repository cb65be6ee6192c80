// STROTSS hyper-column sampling of the StyleLoss plugin (`spatial_feature_extract`, Losses/StyleLoss.py:169-223):
// n sampling positions, followed down the VGG16 pyramid, each giving one bilinear sample of every captured feature map;
// the samples of all maps are concatenated over channels (3 + 2*64 + 2*128 + 3*256 + 2*512 = 2179) and the two (finally
// halved) coordinates are appended.  The plugin composes this from 4 index_selects, 4 multiplies and 3 adds per map (and
// their autograd mirror images): ~500 launches per call, 12 calls per iteration -- the host-side launch cost of which
// bounded BASELINE.json configs[3].  Here: one gather launch forward, one scatter launch backward, over all maps.
// The arithmetic keeps the plugin's roundings: each tap is multiplied by its weight and the four products are added left to
// right (no fma contraction), so the value is bit-identical to the composed torch expression.
#include "common.h"

namespace {

constexpr int HC_MAX_LAYERS = 12;
struct HcLayers {
    const float* f[HC_MAX_LAYERS];    // NHWC fp32 [h*w, C]
    float* g[HC_MAX_LAYERS];          // gradient maps (backward), same layout, zero-initialised by the caller
    int C[HC_MAX_LAYERS];
    int off[HC_MAX_LAYERS + 1];       // channel offset of each map in the concatenated column
    int L;
};

// rows: int64 [L][4][n] (flat row index of the four taps), wts: fp32 [4L+2][n] (their weights; then the two coordinates)
__global__ __launch_bounds__(256) void hypercol_fwd_kernel(HcLayers ls, const long long* __restrict__ rows, const float* __restrict__ wts,
                                                           int n, float* __restrict__ out, int ldo) {
    const int i = blockIdx.x;
    const int ctot = ls.off[ls.L];
    for (int c = threadIdx.x; c < ctot + 2; c += 256) {
        if (c >= ctot) { out[(size_t)i * ldo + c] = wts[(size_t)(4 * ls.L + (c - ctot)) * n + i]; continue; }
        int l = 0;
        while (c >= ls.off[l + 1]) ++l;
        const int cl = c - ls.off[l], C = ls.C[l];
        const long long* r = rows + (size_t)l * 4 * n + i;
        const float* w = wts + (size_t)l * 4 * n + i;
        const float* f = ls.f[l];
        const float p0 = __fmul_rn(f[(size_t)r[0] * C + cl], w[0]);
        const float p1 = __fmul_rn(f[(size_t)r[n] * C + cl], w[n]);
        const float p2 = __fmul_rn(f[(size_t)r[2 * (size_t)n] * C + cl], w[2 * (size_t)n]);
        const float p3 = __fmul_rn(f[(size_t)r[3 * (size_t)n] * C + cl], w[3 * (size_t)n]);
        out[(size_t)i * ldo + c] = __fadd_rn(__fadd_rn(__fadd_rn(p0, p1), p2), p3);
    }
}

// d/d(maps): every sample adds weight * column gradient to its four taps (samples may share taps on the coarse maps: fp32
// atomics, as torch's index_add_ does)
__global__ __launch_bounds__(256) void hypercol_bwd_kernel(HcLayers ls, const long long* __restrict__ rows, const float* __restrict__ wts,
                                                           int n, const float* __restrict__ gout, int ldo) {
    const int i = blockIdx.x;
    const int ctot = ls.off[ls.L];
    for (int c = threadIdx.x; c < ctot; c += 256) {
        int l = 0;
        while (c >= ls.off[l + 1]) ++l;
        float* g = ls.g[l];
        if (!g) continue;
        const int cl = c - ls.off[l], C = ls.C[l];
        const long long* r = rows + (size_t)l * 4 * n + i;
        const float* w = wts + (size_t)l * 4 * n + i;
        const float gv = gout[(size_t)i * ldo + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&g[(size_t)r[(size_t)k * n] * C + cl], __fmul_rn(gv, w[(size_t)k * n]));
    }
}

int fill_layers(HcLayers& ls, const float* const* feats, float* const* g_feats, const int* channels, int n_layers) {
    PRX_REQUIRE(n_layers >= 1 && n_layers <= HC_MAX_LAYERS, "hypercolumns: 1..%d feature maps (got %d)", HC_MAX_LAYERS, n_layers);
    ls.L = n_layers;
    ls.off[0] = 0;
    for (int l = 0; l < n_layers; ++l) {
        PRX_REQUIRE(channels[l] >= 1, "hypercolumns: map %d has %d channels", l, channels[l]);
        ls.f[l] = feats ? feats[l] : nullptr;
        ls.g[l] = g_feats ? g_feats[l] : nullptr;
        ls.C[l] = channels[l];
        ls.off[l + 1] = ls.off[l] + channels[l];
    }
    return 0;
}

}  // namespace

extern "C" {

int prx_hypercolumns_fwd(const float* const* feats, const int* channels, int n_layers, const long long* rows, const float* weights,
                         int n, float* out, int ldo, void* stream) {
    HcLayers ls{};
    if (int e = fill_layers(ls, feats, nullptr, channels, n_layers)) return e;
    PRX_REQUIRE(n >= 1 && ldo >= ls.off[n_layers] + 2, "hypercolumns: n=%d ldo=%d (needs >= %d)", n, ldo, ls.off[n_layers] + 2);
    for (int l = 0; l < n_layers; ++l) PRX_REQUIRE(feats[l] != nullptr, "hypercolumns: feature map %d is NULL", l);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hypercol_fwd_kernel, dim3(n), dim3(256), 0, s, ls, rows, weights, n, out, ldo);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_hypercolumns_bwd(float* const* g_feats, const int* channels, int n_layers, const long long* rows, const float* weights,
                         int n, const float* g_out, int ldo, void* stream) {
    HcLayers ls{};
    if (int e = fill_layers(ls, nullptr, g_feats, channels, n_layers)) return e;
    PRX_REQUIRE(n >= 1 && ldo >= ls.off[n_layers] + 2, "hypercolumns bwd: n=%d ldo=%d (needs >= %d)", n, ldo, ls.off[n_layers] + 2);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(hypercol_bwd_kernel, dim3(n), dim3(256), 0, s, ls, rows, weights, n, g_out, ldo);
    PRX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
