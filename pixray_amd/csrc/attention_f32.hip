// Exact-f32 multi-head self-attention for the PRX_PREC_F32 parity mode (same operator as attention.hip:
// nn.MultiheadAttention inside clip.model.ResidualAttentionBlock [UPSTREAM openai/CLIP clip/model.py], call site
// slip.py:65), any sequence length, head dim 64, forward and activation-gradient backward.
//
// This is the checker's arithmetic, not the fast path: fp32 operands, fp32 VALU fma chains, one query (or key) per lane,
// the other side of the product broadcast from LDS in 64-token blocks with an online softmax.  What it must be is
// simple and exact; the bf16 MFMA kernels in attention.hip are measured against it.
#include "attention.h"

namespace {

constexpr int HD = 64;        // head dim
constexpr int BT = 64;        // tokens per LDS block

// stage `rows` x 64 floats of a [T, ld] matrix (rows r0.., columns c0..c0+63) into lds[BT][HD]; rows >= T are zeroed
__device__ __forceinline__ void stage_block(const float* __restrict__ src, long long ld, int r0, int T, float* lds, int lane) {
#pragma unroll 4
    for (int idx = lane; idx < BT * (HD / 4); idx += 64) {
        const int row = idx >> 4, c4 = idx & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + row < T) v = *reinterpret_cast<const float4*>(src + (long long)(r0 + row) * ld + c4 * 4);
        *reinterpret_cast<float4*>(lds + row * HD + c4 * 4) = v;
    }
}

// grid (ceil(T/64), heads, N), one wave; lane = one query
__global__ __launch_bounds__(64) void mha_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                         float* __restrict__ lse, int T, int C, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[BT * HD];
    __shared__ __attribute__((aligned(16))) float Vs[BT * HD];
    __shared__ float Ss[BT * 64];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, n = blockIdx.z;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const long long ld = 3LL * C;
    const float* base = qkv + (long long)n * T * ld + h * HD;
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) v = *reinterpret_cast<const float4*>(base + (long long)t * ld + d);
        q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;
        o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += BT) {
        __syncthreads();
        stage_block(base + C, ld, k0, T, Ks, lane);
        stage_block(base + 2 * C, ld, k0, T, Vs, lane);
        __syncthreads();
        const int nk = min(BT, T - k0);
        float bm = -INFINITY;
        for (int j = 0; j < nk; ++j) {          // scores of this block -> LDS (lane-contiguous, conflict-free)
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(Ks + j * HD + d);
                a = fmaf(q[d], kv.x, a); a = fmaf(q[d + 1], kv.y, a); a = fmaf(q[d + 2], kv.z, a); a = fmaf(q[d + 3], kv.w, a);
            }
            Ss[j * 64 + lane] = a;
            bm = fmaxf(bm, a);
        }
        const float mn = fmaxf(m, bm);
        const float corr = expf(m - mn);      // exp(-inf) = 0 on the first block
        l *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= corr;
        for (int j = 0; j < nk; ++j) {
            const float p = expf(Ss[j * 64 + lane] - mn);
            l += p;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(Vs + j * HD + d);
                o[d] = fmaf(p, vv.x, o[d]); o[d + 1] = fmaf(p, vv.y, o[d + 1]);
                o[d + 2] = fmaf(p, vv.z, o[d + 2]); o[d + 3] = fmaf(p, vv.w, o[d + 3]);
            }
        }
        m = mn;
    }
    if (!live) return;
    const float inv = 1.f / l;
    float* op = out + ((long long)n * T + t) * C + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4)
        *reinterpret_cast<float4*>(op + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    if (lse) lse[((long long)n * heads + h) * T + t] = m + logf(l);
}

// dQ: lane = one query; recomputes P from the saved log-sum-exp.  D = rowsum(dO o O).
__global__ __launch_bounds__(64) void mha_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                            const float* __restrict__ dout, const float* __restrict__ lse,
                                                            float* __restrict__ dqkv, int T, int C, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[BT * HD];
    __shared__ __attribute__((aligned(16))) float Vs[BT * HD];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, n = blockIdx.z;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const long long ld = 3LL * C;
    const float* base = qkv + (long long)n * T * ld + h * HD;
    float q[HD], dO[HD], dq[HD];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), g = v, ov = v;
        if (live) {
            v = *reinterpret_cast<const float4*>(base + (long long)t * ld + d);
            g = *reinterpret_cast<const float4*>(dout + ((long long)n * T + t) * C + h * HD + d);
            ov = *reinterpret_cast<const float4*>(o + ((long long)n * T + t) * C + h * HD + d);
        }
        q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;
        dO[d] = g.x; dO[d + 1] = g.y; dO[d + 2] = g.z; dO[d + 3] = g.w;
        D = fmaf(g.x, ov.x, D); D = fmaf(g.y, ov.y, D); D = fmaf(g.z, ov.z, D); D = fmaf(g.w, ov.w, D);
        dq[d] = dq[d + 1] = dq[d + 2] = dq[d + 3] = 0.f;
    }
    const float L = live ? lse[((long long)n * heads + h) * T + t] : 0.f;
    for (int k0 = 0; k0 < T; k0 += BT) {
        __syncthreads();
        stage_block(base + C, ld, k0, T, Ks, lane);
        stage_block(base + 2 * C, ld, k0, T, Vs, lane);
        __syncthreads();
        const int nk = min(BT, T - k0);
        for (int j = 0; j < nk; ++j) {
            float a = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(Ks + j * HD + d);
                const float4 vv = *reinterpret_cast<const float4*>(Vs + j * HD + d);
                a = fmaf(q[d], kv.x, a); a = fmaf(q[d + 1], kv.y, a); a = fmaf(q[d + 2], kv.z, a); a = fmaf(q[d + 3], kv.w, a);
                dp = fmaf(dO[d], vv.x, dp); dp = fmaf(dO[d + 1], vv.y, dp); dp = fmaf(dO[d + 2], vv.z, dp); dp = fmaf(dO[d + 3], vv.w, dp);
            }
            const float p = expf(a - L);
            const float ds = p * (dp - D);
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(Ks + j * HD + d);
                dq[d] = fmaf(ds, kv.x, dq[d]); dq[d + 1] = fmaf(ds, kv.y, dq[d + 1]);
                dq[d + 2] = fmaf(ds, kv.z, dq[d + 2]); dq[d + 3] = fmaf(ds, kv.w, dq[d + 3]);
            }
        }
    }
    if (!live) return;
    float* dst = dqkv + ((long long)n * T + t) * ld + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4)
        *reinterpret_cast<float4*>(dst + d) = make_float4(dq[d] * scale, dq[d + 1] * scale, dq[d + 2] * scale, dq[d + 3] * scale);
}

// dK, dV: lane = one key; the queries (scaled), their dO, log-sum-exp and D are broadcast from LDS in 64-token blocks
__global__ __launch_bounds__(64) void mha_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                             const float* __restrict__ dout, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, int T, int C, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) float Qs[BT * HD];
    __shared__ __attribute__((aligned(16))) float Gs[BT * HD];
    __shared__ float Ls[BT], Ds[BT];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, n = blockIdx.z;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const long long ld = 3LL * C;
    const float* base = qkv + (long long)n * T * ld + h * HD;
    const float* obase = o + (long long)n * T * C + h * HD;
    const float* gbase = dout + (long long)n * T * C + h * HD;
    float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (live) {
            kv = *reinterpret_cast<const float4*>(base + C + (long long)t * ld + d);
            vv = *reinterpret_cast<const float4*>(base + 2 * C + (long long)t * ld + d);
        }
        k[d] = kv.x; k[d + 1] = kv.y; k[d + 2] = kv.z; k[d + 3] = kv.w;
        v[d] = vv.x; v[d + 1] = vv.y; v[d + 2] = vv.z; v[d + 3] = vv.w;
        dk[d] = dk[d + 1] = dk[d + 2] = dk[d + 3] = 0.f;
        dv[d] = dv[d + 1] = dv[d + 2] = dv[d + 3] = 0.f;
    }
    for (int q0 = 0; q0 < T; q0 += BT) {
        __syncthreads();
        stage_block(base, ld, q0, T, Qs, lane);
        stage_block(gbase, C, q0, T, Gs, lane);
        {
            const int qi = q0 + lane;
            float D = 0.f, L = 0.f;
            if (qi < T) {
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 g = *reinterpret_cast<const float4*>(gbase + (long long)qi * C + d);
                    const float4 ov = *reinterpret_cast<const float4*>(obase + (long long)qi * C + d);
                    D = fmaf(g.x, ov.x, D); D = fmaf(g.y, ov.y, D); D = fmaf(g.z, ov.z, D); D = fmaf(g.w, ov.w, D);
                }
                L = lse[((long long)n * heads + h) * T + qi];
            }
            Ds[lane] = D; Ls[lane] = L;
        }
        __syncthreads();
        const int nq = min(BT, T - q0);
        for (int i = 0; i < nq; ++i) {
            float a = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(Qs + i * HD + d);
                const float4 g = *reinterpret_cast<const float4*>(Gs + i * HD + d);
                a = fmaf(qv.x * scale, k[d], a); a = fmaf(qv.y * scale, k[d + 1], a);
                a = fmaf(qv.z * scale, k[d + 2], a); a = fmaf(qv.w * scale, k[d + 3], a);
                dp = fmaf(g.x, v[d], dp); dp = fmaf(g.y, v[d + 1], dp); dp = fmaf(g.z, v[d + 2], dp); dp = fmaf(g.w, v[d + 3], dp);
            }
            const float p = expf(a - Ls[i]);
            const float ds = p * (dp - Ds[i]);
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 qv = *reinterpret_cast<const float4*>(Qs + i * HD + d);
                const float4 g = *reinterpret_cast<const float4*>(Gs + i * HD + d);
                dv[d] = fmaf(p, g.x, dv[d]); dv[d + 1] = fmaf(p, g.y, dv[d + 1]);
                dv[d + 2] = fmaf(p, g.z, dv[d + 2]); dv[d + 3] = fmaf(p, g.w, dv[d + 3]);
                dk[d] = fmaf(ds, qv.x * scale, dk[d]); dk[d + 1] = fmaf(ds, qv.y * scale, dk[d + 1]);
                dk[d + 2] = fmaf(ds, qv.z * scale, dk[d + 2]); dk[d + 3] = fmaf(ds, qv.w * scale, dk[d + 3]);
            }
        }
    }
    if (!live) return;
    float* dst = dqkv + ((long long)n * T + t) * ld + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        *reinterpret_cast<float4*>(dst + C + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
        *reinterpret_cast<float4*>(dst + 2 * C + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
    }
}

}  // namespace

int prx_mha_fwd_f32(const float* qkv, float* out, float* lse, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(C == heads * 64 && T >= 1, "mha(f32): needs head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    hipLaunchKernelGGL(mha_fwd_f32_kernel, dim3(ceil_div(T, 64), heads, N), dim3(64), 0, s, qkv, out, lse, T, C, heads, 0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_mha_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int N, int T, int C,
                    int heads, hipStream_t s) {
    PRX_REQUIRE(C == heads * 64 && T >= 1, "mha(f32) bwd: needs head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    const dim3 grid(ceil_div(T, 64), heads, N);
    hipLaunchKernelGGL(mha_bwd_dq_f32_kernel, grid, dim3(64), 0, s, qkv, out, dout, lse, dqkv, T, C, heads, 0.125f);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mha_bwd_dkv_f32_kernel, grid, dim3(64), 0, s, qkv, out, dout, lse, dqkv, T, C, heads, 0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}
