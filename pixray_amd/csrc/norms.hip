// GroupNorm(32)+swish (taming Decoder `Normalize` + `nonlinearity`, [UPSTREAM]
// taming/modules/diffusionmodules/model.py, call site vqgan.py:195) and
// LayerNorm (CLIP VisionTransformer ln_pre / ln_1 / ln_2 / ln_post, call site
// slip.py:65) -- forward and activation-gradient backward.  HBM-bound
// wave-primitive kernels: fp32 residual stream in, bf16 GEMM operand out.
#include "norms.h"

namespace {

// ---------------------------------------------------------------------------
// GroupNorm statistics over an NHWC fp32 tensor x[NB][P][C], 32 groups.
// MODE 0: forward sums (sum x, sum x^2)
// MODE 1: backward sums (sum dxhat, sum dxhat*xhat) with dxhat = g * act'(y) * gamma
// Each thread owns one channel quad (float4); a block covers 256/(C/4) pixels
// per step.  Block partials are combined in double and added to
// stats[NB][32][2] (double) with one atomic per (block, group, moment).
// ---------------------------------------------------------------------------
struct GNArgs {
    const float* x;      // [NB][P][C]
    const float* g;      // [NB][P][C] upstream grad (bwd)
    const double* fstats;// forward stats (bwd) [NB][32][2]
    const float* gamma;  // [C]
    const float* beta;   // [C]
    double* stats;       // out [NB][32][2]
    int P, C, swish;
    float eps;
};

__device__ __forceinline__ void gn_mean_rstd(const double* st, double n, float eps, float& mean, float& rstd) {
    double m = st[0] / n;
    double var = st[1] / n - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

__device__ __forceinline__ float swish_grad(float y) {
    float s = sigmoidf_(y);
    return s * (1.f + y * (1.f - s));
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GNArgs a) {
    const int C4 = a.C >> 2;
    const int ppb = 256 / C4;              // pixels per block step
    const int cq = threadIdx.x % C4;
    const int pl = threadIdx.x / C4;
    const int b = blockIdx.y;
    const int gs = a.C / 32;               // channels per group
    const int grp = (cq * 4) / gs;
    const size_t base = (size_t)b * a.P * a.C;
    const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
    const float4* g4 = MODE ? reinterpret_cast<const float4*>(a.g + base) : nullptr;

    float mean = 0.f, rstd = 1.f;
    float4 ga = {1, 1, 1, 1}, be = {0, 0, 0, 0};
    if (MODE) {
        gn_mean_rstd(a.fstats + ((size_t)b * 32 + grp) * 2, (double)a.P * gs, a.eps, mean, rstd);
        ga = reinterpret_cast<const float4*>(a.gamma)[cq];
        be = reinterpret_cast<const float4*>(a.beta)[cq];
    }
    float s0 = 0.f, s1 = 0.f;
    auto accum = [&](const float4& v, const float4& gg) {
        if (MODE == 0) {
            s0 += (v.x + v.y) + (v.z + v.w);
            s1 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
            float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float xh = (xv[i] - mean) * rstd;
                float gy = gv[i];
                if (a.swish) gy *= swish_grad(xh * gav[i] + bev[i]);
                float dxh = gy * gav[i];
                s0 += dxh;
                s1 += dxh * xh;
            }
        }
    };
    {
        const int stride = gridDim.x * ppb;
        int p = blockIdx.x * ppb + pl;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (; p + 3 * stride < a.P; p += 4 * stride) {      // 4 independent loads in flight per thread
            float4 v0 = x4[(size_t)p * C4 + cq], v1 = x4[(size_t)(p + stride) * C4 + cq];
            float4 v2 = x4[(size_t)(p + 2 * stride) * C4 + cq], v3 = x4[(size_t)(p + 3 * stride) * C4 + cq];
            float4 g0 = z4, g1 = z4, g2 = z4, g3 = z4;
            if (MODE) {
                g0 = g4[(size_t)p * C4 + cq]; g1 = g4[(size_t)(p + stride) * C4 + cq];
                g2 = g4[(size_t)(p + 2 * stride) * C4 + cq]; g3 = g4[(size_t)(p + 3 * stride) * C4 + cq];
            }
            accum(v0, g0); accum(v1, g1); accum(v2, g2); accum(v3, g3);
        }
        for (; p < a.P; p += stride) accum(x4[(size_t)p * C4 + cq], MODE ? g4[(size_t)p * C4 + cq] : z4);
    }
    __shared__ float red[256][2];
    red[threadIdx.x][0] = s0;
    red[threadIdx.x][1] = s1;
    __syncthreads();
    if (threadIdx.x < 64) {
        // threads 0..31 -> moment 0 of group t, 32..63 -> moment 1
        const int gidx = threadIdx.x & 31, mom = threadIdx.x >> 5;
        const int q0 = gidx * gs / 4, q1 = (gidx + 1) * gs / 4;
        double acc = 0.0;
        for (int pp = 0; pp < ppb; ++pp)
            for (int q = q0; q < q1; ++q) acc += (double)red[pp * C4 + q][mom];
        atomicAdd(&a.stats[((size_t)b * 32 + gidx) * 2 + mom], acc);
    }
}

// y = gn(x); out = swish ? y*sigmoid(y) : y   -> bf16 (and optionally fp32)
// `xcd`: give workgroup b the xcd_linear(b)-th contiguous slice of the tensor (common.h) instead of a grid-stride comb
__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const GNArgs a, bf16_t* out_bf16, float* out_f32, int NB, int xcd, int h16) {
    const int C4 = a.C >> 2;
    const int gs = a.C / 32;
    const size_t total = (size_t)NB * a.P * C4;
    size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i1 = total, step = (size_t)gridDim.x * blockDim.x;
    if (xcd) {
        const size_t per = (total + gridDim.x - 1) / gridDim.x, lo = per * xcd_linear(blockIdx.x, gridDim.x);
        i0 = lo + threadIdx.x; i1 = lo + per < total ? lo + per : total; step = blockDim.x;
    }
    for (size_t idx = i0; idx < i1; idx += step) {
        const int cq = (int)(idx % C4);
        const int b = (int)(idx / ((size_t)a.P * C4));
        const int grp = (cq * 4) / gs;
        float mean, rstd;
        gn_mean_rstd(a.stats + ((size_t)b * 32 + grp) * 2, (double)a.P * gs, a.eps, mean, rstd);
        float4 v = reinterpret_cast<const float4*>(a.x)[idx];
        float4 ga = reinterpret_cast<const float4*>(a.gamma)[cq];
        float4 be = reinterpret_cast<const float4*>(a.beta)[cq];
        float xv[4] = {v.x, v.y, v.z, v.w};
        float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = (xv[i] - mean) * rstd * gav[i] + bev[i];
            o[i] = a.swish ? y * sigmoidf_(y) : y;
        }
        if (out_bf16) {
            reinterpret_cast<bf16x4*>(out_bf16)[idx] = to_op16x4(o[0], o[1], o[2], o[3], h16);
        }
        if (out_f32) reinterpret_cast<float4*>(out_f32)[idx] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat*xhat)) (+ add)
__global__ __launch_bounds__(256) void gn_apply_bwd_kernel(const GNArgs a, const double* bstats, const float* add,
                                                           float* dx, bf16_t* dx_bf16, int NB, int xcd, int h16) {
    const int C4 = a.C >> 2;
    const int gs = a.C / 32;
    const size_t total = (size_t)NB * a.P * C4;
    const double n = (double)a.P * gs;
    size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, i1 = total, step = (size_t)gridDim.x * blockDim.x;
    if (xcd) {
        const size_t per = (total + gridDim.x - 1) / gridDim.x, lo = per * xcd_linear(blockIdx.x, gridDim.x);
        i0 = lo + threadIdx.x; i1 = lo + per < total ? lo + per : total; step = blockDim.x;
    }
    for (size_t idx = i0; idx < i1; idx += step) {
        const int cq = (int)(idx % C4);
        const int b = (int)(idx / ((size_t)a.P * C4));
        const int grp = (cq * 4) / gs;
        float mean, rstd;
        gn_mean_rstd(a.fstats + ((size_t)b * 32 + grp) * 2, n, a.eps, mean, rstd);
        const float m1 = (float)(bstats[((size_t)b * 32 + grp) * 2 + 0] / n);
        const float m2 = (float)(bstats[((size_t)b * 32 + grp) * 2 + 1] / n);
        float4 v = reinterpret_cast<const float4*>(a.x)[idx];
        float4 gg = reinterpret_cast<const float4*>(a.g)[idx];
        float4 ga = reinterpret_cast<const float4*>(a.gamma)[cq];
        float4 be = reinterpret_cast<const float4*>(a.beta)[cq];
        float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
        float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float xh = (xv[i] - mean) * rstd;
            float gy = gv[i];
            if (a.swish) gy *= swish_grad(xh * gav[i] + bev[i]);
            float dxh = gy * gav[i];
            o[i] = rstd * (dxh - m1 - xh * m2);
        }
        if (add) {
            float4 ad = reinterpret_cast<const float4*>(add)[idx];
            o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
        }
        if (dx) reinterpret_cast<float4*>(dx)[idx] = make_float4(o[0], o[1], o[2], o[3]);      // dx may be null: only the operand twin is wanted
        if (dx_bf16) {
            reinterpret_cast<bf16x4*>(dx_bf16)[idx] = to_op16x4(o[0], o[1], o[2], o[3], h16);
        }
    }
}

// ---------------------------------------------------------------------------
// LayerNorm over rows of width C (C % 256 == 0, C <= 2048): one wave per row.
// ---------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ out_bf16,
                                                     float* __restrict__ out_f32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, int C, long long ldx,
                                                     float eps, int xcd, int h16) {
    const int lane = threadIdx.x & 63;
    const int row = (xcd ? xcd_linear(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 8;  // float4 per lane
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            v[i] = xr[i * 64 + lane];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            ss += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float var = wave_sum(ss) / (float)C;
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const int c4 = i * 64 + lane;
            float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
            float4 be = reinterpret_cast<const float4*>(beta)[c4];
            float o0 = (v[i].x - mean) * rstd * ga.x + be.x;
            float o1 = (v[i].y - mean) * rstd * ga.y + be.y;
            float o2 = (v[i].z - mean) * rstd * ga.z + be.z;
            float o3 = (v[i].w - mean) * rstd * ga.w + be.w;
            if (out_bf16) {
                reinterpret_cast<bf16x4*>(out_bf16 + (size_t)row * C)[c4] = to_op16x4(o0, o1, o2, o3, h16);
            }
            if (out_f32) reinterpret_cast<float4*>(out_f32 + (size_t)row * C)[c4] = make_float4(o0, o1, o2, o3);
        }
}

// dx_out[row] = (add ? add[row] : 0) + rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)),  dxhat = g*gamma
template <int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ g, long long ldg,
                                                     const float* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ add,
                                                     long long ldadd, float* __restrict__ dx, long long lddx,
                                                     bf16_t* __restrict__ dx_bf16, long long lddxb, int rows, int C, int xcd, int h16, int add_every) {
    const int lane = threadIdx.x & 63;
    const int row = (xcd ? xcd_linear(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (add_every > 0 && row % add_every != 0) add = nullptr;      // wave-uniform
    const int nv = C >> 8;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * ldx);
    const float4* gr = reinterpret_cast<const float4*>(g + (size_t)row * ldg);
    float4 xh[MAXV], dh[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const int c4 = i * 64 + lane;
            float4 xv = xr[c4], gv = gr[c4];
            float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
            xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            dh[i] = make_float4(gv.x * ga.x, gv.y * ga.y, gv.z * ga.z, gv.w * ga.w);
            s1 += (dh[i].x + dh[i].y) + (dh[i].z + dh[i].w);
            s2 += (dh[i].x * xh[i].x + dh[i].y * xh[i].y) + (dh[i].z * xh[i].z + dh[i].w * xh[i].w);
        }
    const float m1 = wave_sum(s1) / (float)C;
    const float m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const int c4 = i * 64 + lane;
            float4 o = make_float4(rstd * (dh[i].x - m1 - xh[i].x * m2), rstd * (dh[i].y - m1 - xh[i].y * m2),
                                   rstd * (dh[i].z - m1 - xh[i].z * m2), rstd * (dh[i].w - m1 - xh[i].w * m2));
            if (add) {
                float4 ad = reinterpret_cast<const float4*>(add + (size_t)row * ldadd)[c4];
                o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
            }
            reinterpret_cast<float4*>(dx + (size_t)row * lddx)[c4] = o;
            if (dx_bf16) {
                reinterpret_cast<bf16x4*>(dx_bf16 + (size_t)row * lddxb)[c4] = to_op16x4(o.x, o.y, o.z, o.w, h16);
            }
        }
}

int gn_grid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 4096); }

}  // namespace

int prx_groupnorm_fwd(const float* x, const float* gamma, const float* beta, double* stats, bf16_t* out_bf16,
                      float* out_f32, int NB, int P, int C, int swish, float eps, hipStream_t s, int zero_stats,
                      int stats_ready, int h16) {
    PRX_REQUIRE(C % 32 == 0 && (C / 32) % 4 == 0 && 256 % (C / 4) == 0, "groupnorm: unsupported C=%d", C);
    GNArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.stats = stats; a.P = P; a.C = C; a.swish = swish; a.eps = eps;
    if (!stats_ready) {     // stats_ready: the producing GEMM already accumulated (sum, sumsq) in its epilogue
        if (zero_stats) PRX_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * NB * 64, s));
        const int ppb = 256 / (C / 4);
        int blocks = std::min(ceil_div(P, ppb * 4), 256);   // <= one block per CU: few (contended) double atomics
        hipLaunchKernelGGL(gn_stats_kernel<0>, dim3(blocks, NB), dim3(256), 0, s, a);
        PRX_LAUNCH_CHECK();
    }
    if (out_bf16 || out_f32) {
        hipLaunchKernelGGL(gn_apply_fwd_kernel, dim3(gn_grid((size_t)NB * P * C / 4)), dim3(256), 0, s, a, out_bf16,
                           out_f32, NB, prx_xcd_local(), h16);
        PRX_LAUNCH_CHECK();
    }
    return 0;
}

int prx_groupnorm_bwd(const float* g, const float* x, const float* gamma, const float* beta, const double* fstats,
                      double* bstats, const float* add, float* dx, bf16_t* dx_bf16, int NB, int P, int C, int swish,
                      float eps, hipStream_t s, int zero_stats, int stats_ready, int h16) {
    PRX_REQUIRE(256 % (C / 4) == 0 && (C / 32) % 4 == 0, "groupnorm bwd: unsupported C=%d", C);
    GNArgs a{};
    a.x = x; a.g = g; a.fstats = fstats; a.gamma = gamma; a.beta = beta; a.stats = bstats;
    a.P = P; a.C = C; a.swish = swish; a.eps = eps;
    if (!stats_ready) {      // stats_ready: the GEMM that produced `g` already accumulated the sums in its epilogue (gemm.h gnb_*)
        if (zero_stats) PRX_CHECK_HIP(hipMemsetAsync(bstats, 0, sizeof(double) * NB * 64, s));
        const int ppb = 256 / (C / 4);
        int blocks = std::min(ceil_div(P, ppb * 4), 256);
        hipLaunchKernelGGL(gn_stats_kernel<1>, dim3(blocks, NB), dim3(256), 0, s, a);
        PRX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(gn_apply_bwd_kernel, dim3(gn_grid((size_t)NB * P * C / 4)), dim3(256), 0, s, a, bstats, add, dx,
                       dx_bf16, NB, prx_xcd_local(), h16);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, bf16_t* out_bf16,
                      float* out_f32, float* mean, float* rstd, int rows, int C, float eps, hipStream_t s, int h16) {
    PRX_REQUIRE(C % 256 == 0 && C <= 2048, "layernorm: C must be a multiple of 256 and <= 2048 (C=%d)", C);
    dim3 grid(ceil_div(rows, 4));
    if (C <= 1024)
        hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, s, x, gamma, beta, out_bf16, out_f32, mean, rstd, rows,
                           C, ldx, eps, prx_xcd_local(), h16);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, dim3(256), 0, s, x, gamma, beta, out_bf16, out_f32, mean, rstd, rows,
                           C, ldx, eps, prx_xcd_local(), h16);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_layernorm_bwd(const float* g, long long ldg, const float* x, long long ldx, const float* gamma,
                      const float* mean, const float* rstd, const float* add, long long ldadd, float* dx,
                      long long lddx, bf16_t* dx_bf16, long long lddxb, int rows, int C, hipStream_t s, int h16, int add_every) {
    PRX_REQUIRE(C % 256 == 0 && C <= 2048, "layernorm bwd: C must be a multiple of 256 and <= 2048 (C=%d)", C);
    dim3 grid(ceil_div(rows, 4));
    if (C <= 1024)
        hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), 0, s, g, ldg, x, ldx, gamma, mean, rstd, add, ldadd, dx,
                           lddx, dx_bf16, lddxb, rows, C, prx_xcd_local(), h16, add_every);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, dim3(256), 0, s, g, ldg, x, ldx, gamma, mean, rstd, add, ldadd, dx,
                           lddx, dx_bf16, lddxb, rows, C, prx_xcd_local(), h16, add_every);
    PRX_LAUNCH_CHECK();
    return 0;
}
