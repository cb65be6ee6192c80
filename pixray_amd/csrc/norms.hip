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
    const void* x;       // [NB][P][C] fp32, or the 16-bit operand format when s16 (the lean layout, common.h stream_ld4)
    const void* g;       // [NB][P][C] upstream grad (bwd), same element type as x
    const double* fstats;// forward stats (bwd) [NB][32][2]
    const float* gamma;  // [C]
    const float* beta;   // [C]
    double* stats;       // out [NB][32][2]
    int P, C, swish;
    float eps;
    int s16, h16;        // x / g / add are 16-bit streams; their format (IEEE half or bf16)
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

template <int MODE, bool S16>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GNArgs a) {
    const int C4 = a.C >> 2;
    const int ppb = 256 / C4;              // pixels per block step
    const int cq = threadIdx.x % C4;
    const int pl = threadIdx.x / C4;
    const int b = blockIdx.y;
    const int gs = a.C / 32;               // channels per group
    const int grp = (cq * 4) / gs;
    const size_t base4 = (size_t)b * a.P * C4;
    auto X = [&](size_t i) { return stream_ld4<S16>(a.x, base4 + i, a.h16); };
    auto G = [&](size_t i) { return stream_ld4<S16>(a.g, base4 + i, a.h16); };

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
            float4 v0 = X((size_t)p * C4 + cq), v1 = X((size_t)(p + stride) * C4 + cq);
            float4 v2 = X((size_t)(p + 2 * stride) * C4 + cq), v3 = X((size_t)(p + 3 * stride) * C4 + cq);
            float4 g0 = z4, g1 = z4, g2 = z4, g3 = z4;
            if (MODE) {
                g0 = G((size_t)p * C4 + cq); g1 = G((size_t)(p + stride) * C4 + cq);
                g2 = G((size_t)(p + 2 * stride) * C4 + cq); g3 = G((size_t)(p + 3 * stride) * C4 + cq);
            }
            accum(v0, g0); accum(v1, g1); accum(v2, g2); accum(v3, g3);
        }
        for (; p < a.P; p += stride) accum(X((size_t)p * C4 + cq), MODE ? G((size_t)p * C4 + cq) : z4);
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
//
// Both apply kernels: a thread's channel quad is the same in every iteration (the strides are multiples of 256, and 256 % (C/4)
// == 0), so its group's mean / rstd (float64 divisions and a square root) and its gamma / beta are taken ONCE, not per element --
// with them inside the loop these "HBM-bound" passes were bound by the fp64 arithmetic: halving their bytes (the lean layout)
// moved them by 0-8 % (profiles/r05_timeline/).  Two elements are in flight per thread.
struct GnLane { float mean, rstd, m1, m2; float4 ga, be; int b; };
template <bool BWD>
__device__ __forceinline__ void gn_lane_load(const GNArgs& a, const double* bstats, int b, int cq, GnLane& L) {
    const int gs = a.C / 32, grp = (cq * 4) / gs;
    const double n = (double)a.P * gs;
    gn_mean_rstd((BWD ? a.fstats : a.stats) + ((size_t)b * 32 + grp) * 2, n, a.eps, L.mean, L.rstd);
    if (BWD) {
        L.m1 = (float)(bstats[((size_t)b * 32 + grp) * 2 + 0] / n);
        L.m2 = (float)(bstats[((size_t)b * 32 + grp) * 2 + 1] / n);
    }
    L.b = b;
}

// (32-bit indices: the host checks NB * P * C / 4 < 2^31 -- the 64-bit divisions of the slice arithmetic were a third of a small
// launch; the first elements are requested BEFORE the statistics are fetched and turned into mean / rstd, so the two memory round
// trips of a launch-bound pass overlap instead of following each other)
template <bool S16>
__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const GNArgs a, bf16_t* out_bf16, float* out_f32, int NB, int xcd, int h16) {
    const unsigned C4 = (unsigned)a.C >> 2;
    const unsigned total = (unsigned)NB * (unsigned)a.P * C4, per_b = (unsigned)a.P * C4;
    unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x, i1 = total, step = gridDim.x * blockDim.x;
    if (xcd) {
        const unsigned per = (total + gridDim.x - 1) / gridDim.x, lo = per * xcd_linear(blockIdx.x, gridDim.x);
        i0 = lo + threadIdx.x; i1 = lo + per < total ? lo + per : total; step = blockDim.x;
    }
    if (i0 >= i1) return;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned idx = i0;
    float4 v0 = stream_ld4<S16>(a.x, idx, a.h16);
    float4 v1 = idx + step < i1 ? stream_ld4<S16>(a.x, idx + step, a.h16) : z4;
    const int cq = (int)(i0 % C4);
    GnLane L;
    L.ga = reinterpret_cast<const float4*>(a.gamma)[cq];
    L.be = reinterpret_cast<const float4*>(a.beta)[cq];
    gn_lane_load<false>(a, nullptr, NB == 1 ? 0 : (int)(i0 / per_b), cq, L);
    auto one = [&](unsigned id, const float4& v) {
        if (NB != 1) { const int b = (int)(id / per_b); if (b != L.b) gn_lane_load<false>(a, nullptr, b, cq, L); }
        const float xv[4] = {v.x, v.y, v.z, v.w};
        const float gav[4] = {L.ga.x, L.ga.y, L.ga.z, L.ga.w}, bev[4] = {L.be.x, L.be.y, L.be.z, L.be.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = (xv[i] - L.mean) * L.rstd * gav[i] + bev[i];
            o[i] = a.swish ? y * sigmoidf_(y) : y;
        }
        if (out_bf16) reinterpret_cast<bf16x4*>(out_bf16)[id] = to_op16x4(o[0], o[1], o[2], o[3], h16);
        if (out_f32) reinterpret_cast<float4*>(out_f32)[id] = make_float4(o[0], o[1], o[2], o[3]);
    };
    for (; idx + step < i1; idx += 2 * step) {
        const unsigned nx = idx + 2 * step;                       // the next pair is in flight while this one is finished
        const float4 n0 = nx < i1 ? stream_ld4<S16>(a.x, nx, a.h16) : z4;
        const float4 n1 = nx + step < i1 ? stream_ld4<S16>(a.x, nx + step, a.h16) : z4;
        one(idx, v0); one(idx + step, v1);
        v0 = n0; v1 = n1;
    }
    if (idx < i1) one(idx, v0);
}

// dx = rstd * (dxhat - mean_g(dxhat) - xhat * mean_g(dxhat*xhat)) (+ add)
template <bool S16>
__global__ __launch_bounds__(256) void gn_apply_bwd_kernel(const GNArgs a, const double* bstats, const void* add,
                                                           float* dx, bf16_t* dx_bf16, int NB, int xcd, int h16) {
    const unsigned C4 = (unsigned)a.C >> 2;
    const unsigned total = (unsigned)NB * (unsigned)a.P * C4, per_b = (unsigned)a.P * C4;
    unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x, i1 = total, step = gridDim.x * blockDim.x;
    if (xcd) {
        const unsigned per = (total + gridDim.x - 1) / gridDim.x, lo = per * xcd_linear(blockIdx.x, gridDim.x);
        i0 = lo + threadIdx.x; i1 = lo + per < total ? lo + per : total; step = blockDim.x;
    }
    if (i0 >= i1) return;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned idx = i0;
    // the first pair is requested before the statistics (see the forward kernel)
    const bool two0 = idx + step < i1;
    float4 v0 = stream_ld4<S16>(a.x, idx, a.h16), g0 = stream_ld4<S16>(a.g, idx, a.h16), a0 = add ? stream_ld4<S16>(add, idx, a.h16) : z4;
    float4 v1 = two0 ? stream_ld4<S16>(a.x, idx + step, a.h16) : z4, g1 = two0 ? stream_ld4<S16>(a.g, idx + step, a.h16) : z4;
    float4 a1 = (two0 && add) ? stream_ld4<S16>(add, idx + step, a.h16) : z4;
    const int cq = (int)(i0 % C4);
    GnLane L;
    L.ga = reinterpret_cast<const float4*>(a.gamma)[cq];
    L.be = reinterpret_cast<const float4*>(a.beta)[cq];
    gn_lane_load<true>(a, bstats, NB == 1 ? 0 : (int)(i0 / per_b), cq, L);
    auto one = [&](unsigned id, const float4& v, const float4& gg, const float4& ad) {
        if (NB != 1) { const int b = (int)(id / per_b); if (b != L.b) gn_lane_load<true>(a, bstats, b, cq, L); }
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {gg.x, gg.y, gg.z, gg.w};
        const float gav[4] = {L.ga.x, L.ga.y, L.ga.z, L.ga.w}, bev[4] = {L.be.x, L.be.y, L.be.z, L.be.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float xh = (xv[i] - L.mean) * L.rstd;
            float gy = gv[i];
            if (a.swish) gy *= swish_grad(xh * gav[i] + bev[i]);
            float dxh = gy * gav[i];
            o[i] = L.rstd * (dxh - L.m1 - xh * L.m2);
        }
        o[0] += ad.x; o[1] += ad.y; o[2] += ad.z; o[3] += ad.w;
        if (dx) reinterpret_cast<float4*>(dx)[id] = make_float4(o[0], o[1], o[2], o[3]);      // dx may be null: only the operand twin is wanted
        if (dx_bf16) reinterpret_cast<bf16x4*>(dx_bf16)[id] = to_op16x4(o[0], o[1], o[2], o[3], h16);
    };
    for (; idx + step < i1; idx += 2 * step) {
        const unsigned nx = idx + 2 * step;
        const bool m0 = nx < i1, m1 = nx + step < i1;
        const float4 nv0 = m0 ? stream_ld4<S16>(a.x, nx, a.h16) : z4, ng0 = m0 ? stream_ld4<S16>(a.g, nx, a.h16) : z4;
        const float4 na0 = (m0 && add) ? stream_ld4<S16>(add, nx, a.h16) : z4;
        const float4 nv1 = m1 ? stream_ld4<S16>(a.x, nx + step, a.h16) : z4, ng1 = m1 ? stream_ld4<S16>(a.g, nx + step, a.h16) : z4;
        const float4 na1 = (m1 && add) ? stream_ld4<S16>(add, nx + step, a.h16) : z4;
        one(idx, v0, g0, a0); one(idx + step, v1, g1, a1);
        v0 = nv0; g0 = ng0; a0 = na0; v1 = nv1; g1 = ng1; a1 = na1;
    }
    if (idx < i1) one(idx, v0, g0, a0);
}

// ---------------------------------------------------------------------------
// LayerNorm over rows of width C (C % 256 == 0, C <= 2048): one wave per row.
// ---------------------------------------------------------------------------
template <int MAXV, bool S16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ out_bf16,
                                                     float* __restrict__ out_f32, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int rows, int C, long long ldx,
                                                     float eps, int xcd, int h16) {
    const int lane = threadIdx.x & 63;
    const int row = (xcd ? xcd_linear(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 8;  // float4 per lane
    const size_t xr4 = (size_t)row * ldx / 4;          // ldx % 4 == 0 (checked on the host)
    // every load of the row -- x, gamma, beta -- is requested up front with a CLAMPED vector index (i >= nv re-reads vector 0 and is
    // ignored), not inside `if (i < nv)`: a wave-uniform branch around a load is a join point at which the compiler waits for
    // everything in flight, so the three x vectors of a 768-wide row and then its gamma / beta vectors were six dependent round trips
    // of a kernel that is one round trip + two wave reductions long
    float4 v[MAXV], gav[MAXV], bev[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) v[i] = stream_ld4<S16>(x, xr4 + (i < nv ? i : 0) * 64 + lane, h16);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        gav[i] = reinterpret_cast<const float4*>(gamma)[(i < nv ? i : 0) * 64 + lane];
        bev[i] = reinterpret_cast<const float4*>(beta)[(i < nv ? i : 0) * 64 + lane];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += i < nv ? (v[i].x + v[i].y) + (v[i].z + v[i].w) : 0.f;
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
            const float4 ga = gav[i], be = bev[i];
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
template <int MAXV, int S16>     // S16 bits: 1 = x, 2 = g, 4 = add are 16-bit streams
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ g, long long ldg,
                                                     const void* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const void* __restrict__ add,
                                                     long long ldadd, float* __restrict__ dx, long long lddx,
                                                     bf16_t* __restrict__ dx_bf16, long long lddxb, int rows, int C, int xcd, int h16, int add_every) {
    const int lane = threadIdx.x & 63;
    const int row = (xcd ? xcd_linear(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (add_every > 0 && row % add_every != 0) add = nullptr;      // wave-uniform
    const int nv = C >> 8;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const size_t xr4 = (size_t)row * ldx / 4, gr4 = (size_t)row * ldg / 4;        // leading dimensions are multiples of 4 (host check)
    // all loads of the row up front with clamped vector indices (ln_fwd_kernel's note); `add` is wave-uniform: clamped to x when absent
    float4 xh[MAXV], dh[MAXV], adv[MAXV];
    {
        float4 xv[MAXV], gv[MAXV], ga[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = (i < nv ? i : 0) * 64 + lane;
            xv[i] = stream_ld4<(S16 & 1) != 0>(x, xr4 + c4, h16);
            gv[i] = stream_ld4<(S16 & 2) != 0>(g, gr4 + c4, h16);
            ga[i] = reinterpret_cast<const float4*>(gamma)[c4];
        }
        if (add) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) adv[i] = stream_ld4<(S16 & 4) != 0>(add, (size_t)row * ldadd / 4 + (i < nv ? i : 0) * 64 + lane, h16);
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            xh[i] = make_float4((xv[i].x - mean) * rstd, (xv[i].y - mean) * rstd, (xv[i].z - mean) * rstd, (xv[i].w - mean) * rstd);
            dh[i] = make_float4(gv[i].x * ga[i].x, gv[i].y * ga[i].y, gv[i].z * ga[i].z, gv[i].w * ga[i].w);
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        s1 += i < nv ? (dh[i].x + dh[i].y) + (dh[i].z + dh[i].w) : 0.f;
        s2 += i < nv ? (dh[i].x * xh[i].x + dh[i].y * xh[i].y) + (dh[i].z * xh[i].z + dh[i].w * xh[i].w) : 0.f;
    }
    const float m1 = wave_sum(s1) / (float)C;
    const float m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const int c4 = i * 64 + lane;
            float4 o = make_float4(rstd * (dh[i].x - m1 - xh[i].x * m2), rstd * (dh[i].y - m1 - xh[i].y * m2),
                                   rstd * (dh[i].z - m1 - xh[i].z * m2), rstd * (dh[i].w - m1 - xh[i].w * m2));
            if (add) { o.x += adv[i].x; o.y += adv[i].y; o.z += adv[i].z; o.w += adv[i].w; }
            if (dx) reinterpret_cast<float4*>(dx + (size_t)row * lddx)[c4] = o;
            if (dx_bf16) {
                reinterpret_cast<bf16x4*>(dx_bf16 + (size_t)row * lddxb)[c4] = to_op16x4(o.x, o.y, o.z, o.w, h16);
            }
        }
}

// apply passes: one channel quad per thread up to 2048 workgroups (the small maps are latency-bound: all the parallelism they can get),
// 4+ quads per thread beyond (the per-thread constants above amortise)
int gn_grid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 2048); }

}  // namespace

int prx_groupnorm_fwd(const void* x, const float* gamma, const float* beta, double* stats, bf16_t* out_bf16,
                      float* out_f32, int NB, int P, int C, int swish, float eps, hipStream_t s, int zero_stats,
                      int stats_ready, int h16, int s16) {
    PRX_REQUIRE(C % 32 == 0 && (C / 32) % 4 == 0 && 256 % (C / 4) == 0, "groupnorm: unsupported C=%d", C);
    GNArgs a{};
    a.x = x; a.gamma = gamma; a.beta = beta; a.stats = stats; a.P = P; a.C = C; a.swish = swish; a.eps = eps; a.s16 = s16; a.h16 = h16;
    if (!stats_ready) {     // stats_ready: the producing GEMM already accumulated (sum, sumsq) in its epilogue
        if (zero_stats) PRX_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * NB * 64, s));
        const int ppb = 256 / (C / 4);
        int blocks = std::min(ceil_div(P, ppb * 4), 256);   // <= one block per CU: few (contended) double atomics
        if (s16) hipLaunchKernelGGL((gn_stats_kernel<0, true>), dim3(blocks, NB), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gn_stats_kernel<0, false>), dim3(blocks, NB), dim3(256), 0, s, a);
        PRX_LAUNCH_CHECK();
    }
    if (out_bf16 || out_f32) {
        PRX_REQUIRE((unsigned long long)NB * P * C / 4 < (1ull << 31), "groupnorm: %d x %d x %d elements exceed the 32-bit index range", NB, P, C);
        if (s16) hipLaunchKernelGGL(gn_apply_fwd_kernel<true>, dim3(gn_grid((size_t)NB * P * C / 4)), dim3(256), 0, s, a, out_bf16,
                                    out_f32, NB, prx_xcd_local(), h16);
        else hipLaunchKernelGGL(gn_apply_fwd_kernel<false>, dim3(gn_grid((size_t)NB * P * C / 4)), dim3(256), 0, s, a, out_bf16,
                                out_f32, NB, prx_xcd_local(), h16);
        PRX_LAUNCH_CHECK();
    }
    return 0;
}

int prx_groupnorm_bwd_stats(const float* g, const float* x, const float* gamma, const float* beta, const double* fstats,
                            double* bstats, int NB, int P, int C, int swish, float eps, hipStream_t s) {
    PRX_REQUIRE(256 % (C / 4) == 0 && (C / 32) % 4 == 0, "groupnorm bwd: unsupported C=%d", C);
    GNArgs a{};
    a.x = x; a.g = g; a.fstats = fstats; a.gamma = gamma; a.beta = beta; a.stats = bstats;
    a.P = P; a.C = C; a.swish = swish; a.eps = eps; a.s16 = 0; a.h16 = 0;
    const int ppb = 256 / (C / 4);
    int blocks = std::min(ceil_div(P, ppb * 4), 256);
    hipLaunchKernelGGL((gn_stats_kernel<1, false>), dim3(blocks, NB), dim3(256), 0, s, a);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_groupnorm_bwd(const void* g, const void* x, const float* gamma, const float* beta, const double* fstats,
                      double* bstats, const void* add, float* dx, bf16_t* dx_bf16, int NB, int P, int C, int swish,
                      float eps, hipStream_t s, int zero_stats, int stats_ready, int h16, int s16) {
    PRX_REQUIRE(256 % (C / 4) == 0 && (C / 32) % 4 == 0, "groupnorm bwd: unsupported C=%d", C);
    GNArgs a{};
    a.x = x; a.g = g; a.fstats = fstats; a.gamma = gamma; a.beta = beta; a.stats = bstats;
    a.P = P; a.C = C; a.swish = swish; a.eps = eps; a.s16 = s16; a.h16 = h16;
    if (!stats_ready) {      // stats_ready: the GEMM that produced `g` already accumulated the sums in its epilogue (gemm.h gnb_*)
        if (zero_stats) PRX_CHECK_HIP(hipMemsetAsync(bstats, 0, sizeof(double) * NB * 64, s));
        const int ppb = 256 / (C / 4);
        int blocks = std::min(ceil_div(P, ppb * 4), 256);
        if (s16) hipLaunchKernelGGL((gn_stats_kernel<1, true>), dim3(blocks, NB), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gn_stats_kernel<1, false>), dim3(blocks, NB), dim3(256), 0, s, a);
        PRX_LAUNCH_CHECK();
    }
    PRX_REQUIRE((unsigned long long)NB * P * C / 4 < (1ull << 31), "groupnorm bwd: %d x %d x %d elements exceed the 32-bit index range", NB, P, C);
    if (s16) hipLaunchKernelGGL(gn_apply_bwd_kernel<true>, dim3(gn_grid((size_t)NB * P * C / 4)), dim3(256), 0, s, a, bstats, add, dx,
                                dx_bf16, NB, prx_xcd_local(), h16);
    else hipLaunchKernelGGL(gn_apply_bwd_kernel<false>, dim3(gn_grid((size_t)NB * P * C / 4)), dim3(256), 0, s, a, bstats, add, dx,
                            dx_bf16, NB, prx_xcd_local(), h16);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_layernorm_fwd(const void* x, long long ldx, const float* gamma, const float* beta, bf16_t* out_bf16,
                      float* out_f32, float* mean, float* rstd, int rows, int C, float eps, hipStream_t s, int h16, int s16) {
    PRX_REQUIRE(C % 256 == 0 && C <= 2048 && ldx % 4 == 0, "layernorm: C must be a multiple of 256 and <= 2048 (C=%d), ldx a multiple of 4", C);
    dim3 grid(ceil_div(rows, 4));
#define PRX_LN_FWD(MAXV, S16) hipLaunchKernelGGL((ln_fwd_kernel<MAXV, S16>), grid, dim3(256), 0, s, x, gamma, beta, out_bf16, out_f32, mean, rstd, \
                                                 rows, C, ldx, eps, prx_xcd_local(), h16)
    if (C <= 1024) { if (s16) PRX_LN_FWD(4, true); else PRX_LN_FWD(4, false); }
    else { if (s16) PRX_LN_FWD(8, true); else PRX_LN_FWD(8, false); }
#undef PRX_LN_FWD
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_layernorm_bwd(const void* g, long long ldg, const void* x, long long ldx, const float* gamma,
                      const float* mean, const float* rstd, const void* add, long long ldadd, float* dx,
                      long long lddx, bf16_t* dx_bf16, long long lddxb, int rows, int C, hipStream_t s, int h16, int add_every, int s16) {
    PRX_REQUIRE(C % 256 == 0 && C <= 2048, "layernorm bwd: C must be a multiple of 256 and <= 2048 (C=%d)", C);
    PRX_REQUIRE(ldg % 4 == 0 && ldx % 4 == 0 && ldadd % 4 == 0 && (dx || dx_bf16), "layernorm bwd: leading dimensions must be multiples of 4, one output is needed");
    PRX_REQUIRE(s16 == 0 || s16 == 1 || s16 == 2 || s16 == 7, "layernorm bwd: stream layouts in use are 0, 1 (x), 2 (g), 7 (x, g, add): got %d", s16);
    dim3 grid(ceil_div(rows, 4));
#define PRX_LN_BWD(MAXV, S16) hipLaunchKernelGGL((ln_bwd_kernel<MAXV, S16>), grid, dim3(256), 0, s, g, ldg, x, ldx, gamma, mean, rstd, add, ldadd, dx, \
                                                 lddx, dx_bf16, lddxb, rows, C, prx_xcd_local(), h16, add_every)
#define PRX_LN_BWD_S(MAXV) do { if (s16 == 0) PRX_LN_BWD(MAXV, 0); else if (s16 == 1) PRX_LN_BWD(MAXV, 1); else if (s16 == 2) PRX_LN_BWD(MAXV, 2); \
                                else PRX_LN_BWD(MAXV, 7); } while (0)
    if (C <= 1024) PRX_LN_BWD_S(4); else PRX_LN_BWD_S(8);
#undef PRX_LN_BWD_S
#undef PRX_LN_BWD
    PRX_LAUNCH_CHECK();
    return 0;
}
