// CLIP ViT multi-head self-attention (nn.MultiheadAttention inside
// clip.model.ResidualAttentionBlock [UPSTREAM openai/CLIP clip/model.py], call
// site slip.py:65), forward and activation-gradient backward, for sequences of
// T <= 64 tokens with head dim 64 (ViT-B/32: T = 50).  One wave owns one
// (image, head): Q/K/V tiles live in LDS, S = QK^T, P = softmax(S/8) and O = PV
// run on v_mfma_f32_32x32x16_bf16 with the softmax done in the MFMA C layout by
// 32-lane butterflies.  Backward recomputes P and produces dQ/dK/dV in one pass.
#include "attention.h"

namespace {

constexpr int LD = 72;             // padded LDS row (bf16 elements), 144 B
constexpr int TILE = 64 * LD;      // one [64][72] bf16 buffer

__device__ __forceinline__ float half_max(float v) {  // reduce over the 32 lanes of a half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// acc[mi][nj] += A[(mi*32 + r)][k] * B[(nj*32 + c)][k] over k in [0,64)
__device__ __forceinline__ void mma_64x64x64(const bf16_t* A, const bf16_t* B, f32x16 (&acc)[2][2], int lane) {
    const int fr = lane & 31, fk = 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i] = *reinterpret_cast<const bf16x8*>(&A[(i * 32 + fr) * LD + ks * 16 + fk]);
            b[i] = *reinterpret_cast<const bf16x8*>(&B[(i * 32 + fr) * LD + ks * 16 + fk]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Column swizzle of the transposed images written by load_tile and read by tr_frag: element (d, t) lives at column
// t ^ tr_swz(d).  Unswizzled, the 64 two-byte writes of one instruction land in 8 banks (rows 8 apart are 1152 B = 0 mod
// 128 apart) and the 8-byte fragment reads of rows d and d+16 collide; with bits 4-5 of d moved into bits 2-4 of the
// column both are conflict-free in a 64-bank model (exhaustive search over the linear swizzles; the unswizzled kernel showed
// SQ_LDS_BANK_CONFLICT = 18 % of its wave cycles, profiles/r01_h_pmc_sq_stalls.csv).  Worth 0.02 ms per iteration.
// Only bits >= 2 of t change, so the 4-element groups the reader fetches stay contiguous.
__device__ __forceinline__ int tr_swz(int d) { return (((d >> 4) & 1) * 12) ^ (((d >> 5) & 1) * 16); }

// 128 threads load a [T][64] bf16 tile (row stride ld) into dst[t][d] and optionally dstT[d][t ^ tr_swz(d)]; rows >= T are zero
__device__ __forceinline__ void load_tile(const bf16_t* src, long long ld, int T, bf16_t* dst, bf16_t* dstT, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 128 * i;
        const int row = c >> 3, kc = c & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < T) v = *reinterpret_cast<const bf16x8*>(src + (long long)row * ld + kc * 8);
        if (dst) *reinterpret_cast<bf16x8*>(&dst[row * LD + kc * 8]) = v;
        if (dstT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dstT[(kc * 8 + e) * LD + (row ^ tr_swz(kc * 8))] = v[e];
        }
    }
}

// acc[mi] += A[(mi*32 + r)][k] * B[(nb*32 + c)][k] over k in [0,64): the 64 x 32 column block `nb` of A B^T
__device__ __forceinline__ void mma_64x32x64(const bf16_t* A, const bf16_t* B, int nb, f32x16 (&acc)[2], int lane) {
    const int fr = lane & 31, fk = 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&B[(nb * 32 + fr) * LD + ks * 16 + fk]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&A[(i * 32 + fr) * LD + ks * 16 + fk]);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void zero_acc2(f32x16 (&acc)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}
// write the C-layout rows 32*rb .. 32*rb+31 of a [t][64] result (a[dj] = columns 32dj ..) to global rows < T
__device__ __forceinline__ void store_rows_global(const f32x16 (&a)[2], int rb, bf16_t* out, long long ld, int T, int lane) {
    const int col = lane & 31, r0 = 32 * rb + 4 * (lane >> 5);
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = r0 + (r & 3) + 8 * (r >> 2);
            if (i < T) out[(long long)i * ld + dj * 32 + col] = (bf16_t)a[dj][r];
        }
}

// softmax over j in the C layout; s holds raw q.k sums; returns p in s.
__device__ __forceinline__ void softmax_c_layout(f32x16 (&s)[2][2], float scale, int T, int lane) {
    const int j0 = lane & 31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v0 = (j0 < T) ? s[mi][0][r] * scale : -INFINITY;
            float v1 = (j0 + 32 < T) ? s[mi][1][r] * scale : -INFINITY;
            float mx = half_max(fmaxf(v0, v1));
            float e0 = __expf(v0 - mx), e1 = __expf(v1 - mx);
            float sum = half_sum(e0 + e1);
            float inv = 1.f / sum;
            s[mi][0][r] = e0 * inv;
            s[mi][1][r] = e1 * inv;
        }
}

// write a C-layout 64x64 tile to LDS as dst[i][j] and/or dstT[j][i] (bf16)
__device__ __forceinline__ void store_c_tile(const f32x16 (&a)[2][2], bf16_t* dst, bf16_t* dstT, int lane) {
    const int col = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = mi * 32 + (r & 3) + 8 * (r >> 2) + rb;
                const int j = nj * 32 + col;
                bf16_t v = (bf16_t)a[mi][nj][r];
                if (dst) dst[i * LD + j] = v;
                if (dstT) dstT[j * LD + i] = v;
            }
}

// write a C-layout [t][d] tile to global rows < T
__device__ __forceinline__ void store_c_global(const f32x16 (&a)[2][2], bf16_t* out, long long ld, int T, int lane) {
    const int col = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = mi * 32 + (r & 3) + 8 * (r >> 2) + rb;
                if (i < T) out[(long long)i * ld + nj * 32 + col] = (bf16_t)a[mi][nj][r];
            }
}

// Forward, "swapped" formulation: S^T = K Q^T puts one QUERY per lane column (C layout: lane l holds query l&31 of its
// block, 16 of its keys in registers, the other 16 in lane l^32), so the softmax is an in-register reduction plus ONE
// cross-half exchange per statistic instead of a 5-step butterfly per row; and the normalised P^T accumulators are
// already the A operand of O = P V (rows = queries = this lane, k slots = the keys it holds) -- P never goes through
// LDS.  The MFMA only needs A and B to agree on which key sits in which k slot: slot 8h+s of step (mi, G) is key
// 32mi + 16G + 8(s>>2) + 4h + (s&3), which for the B operand (rows = head-dim d of V^T) is two 8-byte reads of the
// transposed V image.
// Two waves per (image, head): wave w owns queries 32w .. 32w+31 (its column block of S^T, its rows of O).  768
// one-wave workgroups left every CU with 3 waves in flight and each of them latency-bound; the split halves the MFMA
// chain per wave and doubles the waves a CU can interleave.
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int G) {
    bf16x8 f;
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = (bf16_t)a[8 * G + s];
    return f;
}
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* Tt, int dj, int mi, int G, int lane) {
    const int d = dj * 32 + (lane & 31), t0 = 32 * mi + 16 * G + 4 * (lane >> 5), sw = tr_swz(d);
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Tt + d * LD + (t0 ^ sw));
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Tt + d * LD + ((t0 + 8) ^ sw));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// o[dj] += sum over the contraction axis of A (accumulators a[mi], rows = this lane's column index) x B^T image
__device__ __forceinline__ void mma_acc_tr(const f32x16 (&a)[2], const bf16_t* Tt, f32x16 (&o)[2], int lane) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int G = 0; G < 2; ++G) {
            const bf16x8 fa = acc_frag(a[mi], G);
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) o[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, tr_frag(Tt, dj, mi, G, lane), o[dj], 0, 0, 0);
        }
}

__global__ __launch_bounds__(128) void mha_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T,
                                                      int C, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[3 * TILE];
    bf16_t* Qs = smem;
    bf16_t* Ks = smem + TILE;
    bf16_t* Vt = smem + 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // (image, head) from the flat workgroup id through the XCD map (common.h): an XCD works on a contiguous range of images,
    // the same token rows whose qkv the projection GEMM of that XCD just wrote
    const unsigned flat = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
    const unsigned lin = xcd ? xcd_linear(flat, nwg) : flat;
    const int h = (int)(lin % gridDim.x), n = (int)(lin / gridDim.x);
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    load_tile(base, ld, T, Qs, nullptr, tid);
    load_tile(base + C, ld, T, Ks, nullptr, tid);
    load_tile(base + 2 * C, ld, T, nullptr, Vt, tid);
    __syncthreads();
    f32x16 st[2];                          // st[mi]: keys 32mi.., queries 32w..
    zero_acc2(st);
    mma_64x32x64(Ks, Qs, w, st, lane);
    const int hh = lane >> 5;
    {
        float m = -INFINITY;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v = (j < T) ? st[mi][r] * scale : -INFINITY;
                st[mi][r] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(st[mi][r] - m);
                st[mi][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[mi][r] *= inv;
    }
    f32x16 o[2];                           // o[dj]: queries 32w.., head-dim 32dj..
    zero_acc2(o);
    mma_acc_tr(st, Vt, o, lane);
    store_rows_global(o, w, out + (long long)n * T * C + h * 64, C, T, lane);
}

// Backward, same register-resident scheme as the forward, in two orientations:
//   (1) queries on lanes:  P^T = softmax(K Q^T), dP^T = V dO^T, D_i = sum_j P dP, dS^T  ->  dQ = dS K   (A = dS^T accumulators)
//   (2) keys on lanes:     P = exp(Q K^T / 8 - lse_i), dP = dO V^T, dS               ->  dK = dS^T Q, dV = P^T dO
// Orientation 2 recomputes the two score products instead of transposing dS / P through LDS; it gets the per-query
// log-sum-exp and D_i from orientation 1 through two 64-float LDS arrays.  The B operands K^T, Q^T, dO^T are transposed
// LDS images built one after the other in the fifth tile.  Two waves per (image, head), as in the forward: wave w owns
// queries 32w.. in orientation 1 (its rows of dQ) and keys 32w.. in orientation 2 (its rows of dK and dV).
__global__ __launch_bounds__(128) void mha_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                      bf16_t* __restrict__ dqkv, int T, int C, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[5 * TILE];
    __shared__ __attribute__((aligned(16))) float s_lse[64], s_D[64];
    bf16_t* Qs = smem;
    bf16_t* Ks = smem + TILE;
    bf16_t* Vs = smem + 2 * TILE;
    bf16_t* dOs = smem + 3 * TILE;
    bf16_t* Tt = smem + 4 * TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // (image, head) from the flat workgroup id through the XCD map (common.h): an XCD works on a contiguous range of images,
    // the same token rows whose qkv the projection GEMM of that XCD just wrote
    const unsigned flat = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
    const unsigned lin = xcd ? xcd_linear(flat, nwg) : flat;
    const int h = (int)(lin % gridDim.x), n = (int)(lin / gridDim.x);
    const int hh = lane >> 5;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const bf16_t* dobase = dout + (long long)n * T * C + h * 64;
    load_tile(base, ld, T, Qs, nullptr, tid);
    load_tile(base + C, ld, T, Ks, Tt, tid);             // K row-major and K^T
    load_tile(base + 2 * C, ld, T, Vs, nullptr, tid);
    load_tile(dobase, C, T, dOs, nullptr, tid);
    __syncthreads();
    bf16_t* obase = dqkv + (long long)n * T * ld + h * 64;
    {   // ---- orientation 1: [key j][query i], lane column = query 32w + (lane & 31)
        f32x16 pt[2], dpt[2];
        zero_acc2(pt);
        mma_64x32x64(Ks, Qs, w, pt, lane);
        zero_acc2(dpt);
        mma_64x32x64(Vs, dOs, w, dpt, lane);
        float m = -INFINITY;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v = (j < T) ? pt[mi][r] * scale : -INFINITY;
                pt[mi][r] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(pt[mi][r] - m);
                pt[mi][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        float dot = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pt[mi][r] *= inv;
                dot += pt[mi][r] * dpt[mi][r];
            }
        dot += __shfl_xor(dot, 32, 64);
        if (hh == 0) { s_lse[w * 32 + lane] = m + __logf(sum); s_D[w * 32 + lane] = dot; }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) dpt[mi][r] = scale * pt[mi][r] * (dpt[mi][r] - dot);   // dS^T
        f32x16 acc[2];
        zero_acc2(acc);
        mma_acc_tr(dpt, Tt, acc, lane);                  // dQ[i][d] = sum_j dS[i][j] K[j][d]
        store_rows_global(acc, w, obase, ld, T, lane);
    }
    __syncthreads();                                      // s_lse / s_D visible, K^T image free
    load_tile(base, ld, T, nullptr, Tt, tid);             // Q^T
    // ---- orientation 2: [query i][key j], lane column = key 32w + (lane & 31)
    f32x16 p[2], dp[2];
    zero_acc2(p);
    mma_64x32x64(Qs, Ks, w, p, lane);
    zero_acc2(dp);
    mma_64x32x64(dOs, Vs, w, dp, lane);
    const bool key_ok = w * 32 + (lane & 31) < T;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int i0 = mi * 32 + 8 * rg + 4 * hh;
            const float4 l4 = *reinterpret_cast<const float4*>(&s_lse[i0]);
            const float4 d4 = *reinterpret_cast<const float4*>(&s_D[i0]);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rg * 4 + q;
                const float pv = key_ok ? __expf(p[mi][r] * scale - lv[q]) : 0.f;
                p[mi][r] = pv;
                dp[mi][r] = scale * pv * (dp[mi][r] - dv[q]);   // dS
            }
        }
    __syncthreads();                                      // Q^T image complete
    {
        f32x16 acc[2];
        zero_acc2(acc);
        mma_acc_tr(dp, Tt, acc, lane);                    // dK[j][d] = sum_i dS[i][j] Q[i][d]
        store_rows_global(acc, w, obase + C, ld, T, lane);
    }
    __syncthreads();
    load_tile(dobase, C, T, nullptr, Tt, tid);             // dO^T
    __syncthreads();
    {
        f32x16 acc[2];
        zero_acc2(acc);
        mma_acc_tr(p, Tt, acc, lane);                     // dV[j][d] = sum_i P[i][j] dO[i][d]
        store_rows_global(acc, w, obase + 2 * C, ld, T, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// General sequence length (T > 64: ViT-B/16 has 197 tokens, ViT-L/14 257), head dim 64: flash-style tiling over
// 64-token tiles with an online softmax.  One wave per (64-query tile, head, image).
//   forward : O = softmax(QK^T/8) V, also writes LSE_i = m_i + log(l_i) (fp32) for the backward
//   backward: dQ kernel (one wave per query tile, loops over key tiles) and dK/dV kernel (one wave per key tile,
//             loops over query tiles); both recompute P from LSE.  D_i = sum_d dO_id * O_id is recomputed per tile.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_rows(const bf16_t* src, long long ld, int t0, int T, bf16_t* dst, bf16_t* dstT, int lane) {
    // rows t0 .. t0+63 of a [T][64] matrix (zero beyond T)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        const int row = c >> 3, kc = c & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + row < T) v = *reinterpret_cast<const bf16x8*>(src + (long long)(t0 + row) * ld + kc * 8);
        if (dst) *reinterpret_cast<bf16x8*>(&dst[row * LD + kc * 8]) = v;
        if (dstT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dstT[(kc * 8 + e) * LD + row] = v[e];
        }
    }
}

// CAUSAL: CLIP's text transformer mask (key j visible to query i iff j <= i); K tiles entirely in the future are skipped
template <bool CAUSAL>
__global__ __launch_bounds__(64) void mha_fwd_gen_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         float* __restrict__ lse, int T, int C, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE];
    bf16_t* Qs = smem; bf16_t* Ks = smem + TILE; bf16_t* Vt = smem + 2 * TILE; bf16_t* Ps = smem + 3 * TILE;
    const int lane = threadIdx.x;
    const int qt = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const int q0 = qt * 64;
    load_rows(base, ld, q0, T, Qs, nullptr, lane);
    f32x16 o[2][2];
    zero_acc(o);
    float m_run[2][16], l_run[2][16];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { m_run[mi][r] = -INFINITY; l_run[mi][r] = 0.f; }
    const int j0l = lane & 31;
    const int k_end = CAUSAL ? min(T, q0 + 64) : T;
    for (int k0 = 0; k0 < k_end; k0 += 64) {
        __syncthreads();
        load_rows(base + C, ld, k0, T, Ks, nullptr, lane);
        load_rows(base + 2 * C, ld, k0, T, nullptr, Vt, lane);
        __syncthreads();
        f32x16 s[2][2];
        zero_acc(s);
        mma_64x64x64(Qs, Ks, s, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = q0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);     // this accumulator's query row
                const int lim = CAUSAL ? min(T, qi + 1) : T;
                const float v0 = (k0 + j0l < lim) ? s[mi][0][r] * scale : -INFINITY;
                const float v1 = (k0 + j0l + 32 < lim) ? s[mi][1][r] * scale : -INFINITY;
                const float mx = fmaxf(m_run[mi][r], half_max(fmaxf(v0, v1)));
                const float alpha = __expf(m_run[mi][r] - mx);          // 0 on the first tile (m = -inf)
                const float e0 = __expf(v0 - mx), e1 = __expf(v1 - mx);
                l_run[mi][r] = l_run[mi][r] * alpha + half_sum(e0 + e1);
                m_run[mi][r] = mx;
                s[mi][0][r] = e0; s[mi][1][r] = e1;
                o[mi][0][r] *= alpha; o[mi][1][r] *= alpha;
            }
        store_c_tile(s, Ps, nullptr, lane);
        __syncthreads();
        mma_64x64x64(Ps, Vt, o, lane);
    }
    const int col = lane & 31, rb = 4 * (lane >> 5);
    bf16_t* obase = out + (long long)n * T * C + h * 64;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = q0 + mi * 32 + (r & 3) + 8 * (r >> 2) + rb;
            if (i < T) {
                const float inv = 1.f / l_run[mi][r];
                obase[(long long)i * C + col] = (bf16_t)(o[mi][0][r] * inv);
                obase[(long long)i * C + 32 + col] = (bf16_t)(o[mi][1][r] * inv);
                if (col == 0 && lse) lse[((long long)n * heads + h) * T + i] = m_run[mi][r] + __logf(l_run[mi][r]);
            }
        }
}

// P (C layout, rows = this wave's queries) from LSE, and dS = scale * P o (dP - D)
__device__ __forceinline__ void probs_from_lse(f32x16 (&s)[2][2], const float (&lse_r)[2][16], float scale, int k0, int T, int lane) {
    const int j0l = lane & 31;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[mi][0][r] = (k0 + j0l < T) ? __expf(s[mi][0][r] * scale - lse_r[mi][r]) : 0.f;
            s[mi][1][r] = (k0 + j0l + 32 < T) ? __expf(s[mi][1][r] * scale - lse_r[mi][r]) : 0.f;
        }
}

// per-row quantities of the 64 query rows starting at q0 in the C layout: LSE and D = rowsum(dO o O)
__device__ __forceinline__ void row_stats(const float* lse_h, const bf16_t* o_h, const bf16_t* do_h, long long ldo, int q0, int T,
                                          float (&lse_r)[2][16], float (&d_r)[2][16], int lane) {
    const int rb = 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = q0 + mi * 32 + (r & 3) + 8 * (r >> 2) + rb;
            float dsum = 0.f, l = 0.f;
            if (i < T) {
                l = lse_h[i];
                // 32 lanes of the half-wave share row i: each handles 2 of the 64 columns
                const int c = (lane & 31) * 2;
                dsum = (float)do_h[(long long)i * ldo + c] * (float)o_h[(long long)i * ldo + c] +
                       (float)do_h[(long long)i * ldo + c + 1] * (float)o_h[(long long)i * ldo + c + 1];
            }
            d_r[mi][r] = half_sum(dsum);
            lse_r[mi][r] = (i < T) ? l : INFINITY;     // exp(x - inf) = 0 for padded query rows
        }
}

__global__ __launch_bounds__(64) void mha_bwd_dq_gen_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, int T, int C, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[5 * TILE];
    bf16_t* Qs = smem; bf16_t* dOs = smem + TILE; bf16_t* Ks = smem + 2 * TILE; bf16_t* Vs = smem + 3 * TILE; bf16_t* Kt = smem + 4 * TILE;
    const int lane = threadIdx.x;
    const int qt = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const bf16_t* dob = dout + (long long)n * T * C + h * 64;
    const bf16_t* ob = o + (long long)n * T * C + h * 64;
    const int q0 = qt * 64;
    load_rows(base, ld, q0, T, Qs, nullptr, lane);
    load_rows(dob, C, q0, T, dOs, nullptr, lane);
    float lse_r[2][16], d_r[2][16];
    row_stats(lse + ((long long)n * heads + h) * T, ob, dob, C, q0, T, lse_r, d_r, lane);
    f32x16 dq[2][2];
    zero_acc(dq);
    for (int k0 = 0; k0 < T; k0 += 64) {
        __syncthreads();
        load_rows(base + C, ld, k0, T, Ks, Kt, lane);
        load_rows(base + 2 * C, ld, k0, T, Vs, nullptr, lane);
        __syncthreads();
        f32x16 p[2][2], dp[2][2];
        zero_acc(p);
        mma_64x64x64(Qs, Ks, p, lane);
        probs_from_lse(p, lse_r, scale, k0, T, lane);
        zero_acc(dp);
        mma_64x64x64(dOs, Vs, dp, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dp[mi][0][r] = scale * p[mi][0][r] * (dp[mi][0][r] - d_r[mi][r]);
                dp[mi][1][r] = scale * p[mi][1][r] * (dp[mi][1][r] - d_r[mi][r]);
            }
        __syncthreads();
        store_c_tile(dp, Ks, nullptr, lane);         // Ks <- dS [i][j]  (K itself is no longer needed, K^T is)
        __syncthreads();
        mma_64x64x64(Ks, Kt, dq, lane);              // dQ += dS K
    }
    store_c_global(dq, dqkv + (long long)n * T * ld + h * 64 + (long long)q0 * ld, ld, T - q0, lane);
}

__global__ __launch_bounds__(64) void mha_bwd_dkv_gen_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                             const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, int T, int C, int heads, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[6 * TILE];
    bf16_t* Ks = smem; bf16_t* Vs = smem + TILE; bf16_t* Qs = smem + 2 * TILE; bf16_t* dOs = smem + 3 * TILE;
    bf16_t* Qt = smem + 4 * TILE; bf16_t* dOt = smem + 5 * TILE;
    const int lane = threadIdx.x;
    const int kt = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const bf16_t* dob = dout + (long long)n * T * C + h * 64;
    const bf16_t* ob = o + (long long)n * T * C + h * 64;
    const int k0 = kt * 64;
    load_rows(base + C, ld, k0, T, Ks, nullptr, lane);
    load_rows(base + 2 * C, ld, k0, T, Vs, nullptr, lane);
    f32x16 dk[2][2], dv[2][2];
    zero_acc(dk); zero_acc(dv);
    for (int q0 = 0; q0 < T; q0 += 64) {
        __syncthreads();
        load_rows(base, ld, q0, T, Qs, Qt, lane);
        load_rows(dob, C, q0, T, dOs, dOt, lane);
        __syncthreads();
        float lse_r[2][16], d_r[2][16];
        row_stats(lse + ((long long)n * heads + h) * T, ob, dob, C, q0, T, lse_r, d_r, lane);
        f32x16 p[2][2], dp[2][2];
        zero_acc(p);
        mma_64x64x64(Qs, Ks, p, lane);
        probs_from_lse(p, lse_r, scale, k0, T, lane);
        zero_acc(dp);
        mma_64x64x64(dOs, Vs, dp, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dp[mi][0][r] = scale * p[mi][0][r] * (dp[mi][0][r] - d_r[mi][r]);
                dp[mi][1][r] = scale * p[mi][1][r] * (dp[mi][1][r] - d_r[mi][r]);
            }
        __syncthreads();
        store_c_tile(p, nullptr, Qs, lane);          // Qs  <- P^T  [j][i]
        store_c_tile(dp, nullptr, dOs, lane);        // dOs <- dS^T [j][i]
        __syncthreads();
        mma_64x64x64(Qs, dOt, dv, lane);             // dV += P^T dO
        mma_64x64x64(dOs, Qt, dk, lane);             // dK += dS^T Q
    }
    bf16_t* ob2 = dqkv + (long long)n * T * ld + h * 64 + (long long)k0 * ld;
    store_c_global(dk, ob2 + C, ld, T - k0, lane);
    store_c_global(dv, ob2 + 2 * C, ld, T - k0, lane);
}


// ---------------------------------------------------------------------------------------------------------------
// 64 < T <= 512 (ViT-B/16: 197, ViT-L/14: 257, RN50x4 attention pool: 82): one WORKGROUP per (image, head) with one
// wave per 32 tokens, the register-resident scheme of the T <= 64 kernels above (scores with the owned tokens on the
// lane columns, so that the probabilities are already the A operand of the second product) streamed over 128-token LDS
// chunks of the other operand.  The tile kernels further below (one wave per 64-query tile, every wave re-staging and
// re-transposing K / V for itself, 36-55 KB of LDS per wave) ran ViT-L/14 at 256 cutouts at 1.5 / 1.9 / 6.0 ms per
// layer (forward / dQ / dK+dV: 2-3 waves per CU); they remain for T > 512 and for the causal text tower.
//   forward : online softmax per owned query (running max / sum per lane); the rescale factors reach the O accumulators
//             (queries on register rows) through a 32-float LDS line per wave
//   backward: dQ kernel (owned queries, chunks of K / V / K^T) and dK+dV kernel (owned keys, chunks of Q / dO / Q^T /
//             dO^T with the per-query LSE and D = rowsum(dO o O) staged beside them); both recompute P from the LSE
// ---------------------------------------------------------------------------------------------------------------
constexpr int CH = 128;             // tokens per LDS chunk (four 32-token MFMA blocks)
constexpr int TRS = 132;            // row stride (bf16) of a transposed [64][CH] image: 66 dwords = 2 mod 64 banks, so the 32
                                    // rows d .. d+31 of one 8-byte fragment read fall into 32 distinct bank pairs

// rows t0 .. t0+CH-1 of a [T][64] matrix (zero beyond T) -> row-major image and / or transposed image
__device__ __forceinline__ void stage_chunk(const bf16_t* src, long long ld, int t0, int T, bf16_t* rm, bf16_t* tr, int tid, int nthr) {
    for (int c = tid; c < CH * 8; c += nthr) {
        const int row = c >> 3, kc = c & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t0 + row < T) v = *reinterpret_cast<const bf16x8*>(src + (long long)(t0 + row) * ld + kc * 8);
        if (rm) *reinterpret_cast<bf16x8*>(&rm[row * LD + kc * 8]) = v;
        if (tr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) tr[(kc * 8 + e) * TRS + row] = v[e];
        }
    }
}
// the four k16 fragments of token `t` (one token per lane column, zero beyond T): the B operand of "chunk rows x owned tokens"
__device__ __forceinline__ void own_frags(const bf16_t* src, long long ld, int t, int T, bf16x8 (&f)[4], int lane) {
    const int fk = 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (t < T) f[ks] = *reinterpret_cast<const bf16x8*>(src + (long long)t * ld + ks * 16 + fk);
    }
}
// acc = rows 32sb.. of the row-major chunk image (A) x owned tokens (B): [chunk token][owned token]
__device__ __forceinline__ void mma_chunk_own(const bf16_t* A, int sb, const bf16x8 (&b)[4], f32x16& acc, int lane) {
    const int fr = lane & 31, fk = 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&A[(sb * 32 + fr) * LD + ks * 16 + fk]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ks], acc, 0, 0, 0);
    }
}
__device__ __forceinline__ bf16x8 tr_frag_chunk(const bf16_t* Tt, int dj, int sb, int G, int lane) {
    const int d = dj * 32 + (lane & 31), t0 = 32 * sb + 16 * G + 4 * (lane >> 5);
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Tt + d * TRS + t0);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Tt + d * TRS + t0 + 8);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// o[dj] += (accumulators a: rows = chunk tokens 32sb.., lane column = owned token) as A  x  transposed chunk image
__device__ __forceinline__ void mma_acc_tr_chunk(const f32x16& a, const bf16_t* Tt, int sb, f32x16 (&o)[2], int lane) {
#pragma unroll
    for (int G = 0; G < 2; ++G) {
        const bf16x8 fa = acc_frag(a, G);
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) o[dj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, tr_frag_chunk(Tt, dj, sb, G, lane), o[dj], 0, 0, 0);
    }
}
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// multiply the register rows of o (row of register r: (r&3) + 8(r>>2) + 4(lane>>5)) by line[row]
__device__ __forceinline__ void scale_rows(f32x16 (&o)[2], const float* line, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const float4 a4 = *reinterpret_cast<const float4*>(&line[8 * rg + 4 * hh]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[0][rg * 4 + q] *= av[q]; o[1][rg * 4 + q] *= av[q]; }
    }
}
__device__ __forceinline__ void block_head(int xcd, int& h, int& n) {
    const unsigned flat = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
    const unsigned lin = xcd ? xcd_linear(flat, nwg) : flat;
    h = (int)(lin % gridDim.x); n = (int)(lin / gridDim.x);
}

constexpr int WPB = 5;               // waves per workgroup: a head with more 32-token blocks is split over gridDim.z workgroups
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void mha_fwd_blk_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse, int T, int C, int heads, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[CH * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * TRS];
    __shared__ __attribute__((aligned(16))) float s_line[WPB][32];
    const int tid = threadIdx.x, lane = tid & 63, w = blockIdx.z * (blockDim.x >> 6) + (tid >> 6), hh = lane >> 5;
    const bool active = 32 * w < T;                        // the last workgroup of a head may carry a wave without tokens
    int h, n;
    block_head(xcd, h, n);
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const int q = 32 * w + (lane & 31);                    // this lane's query
    bf16x8 qf[4];
    own_frags(base, ld, q, T, qf, lane);
    float m = -INFINITY, l = 0.f;                           // running max (base-2 domain) and sum of this lane's query
    const float sl2 = scale * 1.4426950408889634f;
    f32x16 o[2];
    zero16(o[0]); zero16(o[1]);
    float* line = s_line[tid >> 6];
    for (int c0 = 0; c0 < T; c0 += CH) {
        __syncthreads();
        stage_chunk(base + C, ld, c0, T, Ks, nullptr, tid, blockDim.x);
        stage_chunk(base + 2 * C, ld, c0, T, nullptr, Vt, tid, blockDim.x);
        __syncthreads();
        const int nsb = active ? min(CH / 32, (T - c0 + 31) / 32) : 0;
        for (int sb = 0; sb < nsb; ++sb) {
            f32x16 st;                                      // [key 32sb..][query]
            zero16(st);
            mma_chunk_own(Ks, sb, qf, st, lane);
            // scores in the base-2 domain (one multiply by scale * log2(e), then v_exp_f32 directly); only the last 32-key
            // block of a head can hold keys beyond T (wave-uniform test)
            float bm = -INFINITY;
            const bool partial = c0 + sb * 32 + 32 > T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = st[r] * sl2;
                if (partial) {
                    const int j = c0 + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    v = (j < T) ? v : -INFINITY;
                }
                st[r] = v;
                bm = fmaxf(bm, v);
            }
            bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
            const float mn = fmaxf(m, bm);                  // finite: every 32-key block that is visited has a valid key
            const float alpha = __builtin_amdgcn_exp2f(m - mn);     // 0 on the first block
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[r] - mn);
                st[r] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 32, 64);
            l = l * alpha + sum;
            m = mn;
            if (__any(alpha != 1.f)) {                      // the running maximum of some query moved: rescale its O row
                if (hh == 0) line[lane] = alpha;
                __builtin_amdgcn_wave_barrier();
                scale_rows(o, line, lane);
                __builtin_amdgcn_wave_barrier();
            }
            mma_acc_tr_chunk(st, Vt, sb, o, lane);          // O[query][d] += P[query][key] V[key][d]
        }
    }
    if (!active) return;
    if (hh == 0) {
        line[lane] = 1.f / l;
        if (q < T && lse) lse[((long long)n * heads + h) * T + q] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
    }
    __builtin_amdgcn_wave_barrier();
    scale_rows(o, line, lane);
    store_rows_global(o, w, out + (long long)n * T * C + h * 64, C, T, lane);
}

__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void mha_bwd_dq_blk_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                              const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                              bf16_t* __restrict__ dqkv, int T, int C, int heads, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[CH * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[CH * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[64 * TRS];
    const int tid = threadIdx.x, lane = tid & 63, w = blockIdx.z * (blockDim.x >> 6) + (tid >> 6), hh = lane >> 5;
    const bool active = 32 * w < T;
    int h, n;
    block_head(xcd, h, n);
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const bf16_t* dob = dout + (long long)n * T * C + h * 64;
    const bf16_t* ob = o + (long long)n * T * C + h * 64;
    const int q = 32 * w + (lane & 31);
    bf16x8 qf[4], dof[4];
    own_frags(base, ld, q, T, qf, lane);
    own_frags(dob, C, q, T, dof, lane);
    float lse_q = INFINITY, D_q = 0.f;                      // exp(x - inf) = 0 for padded queries
    if (q < T) {
        lse_q = lse[((long long)n * heads + h) * T + q];
        float dsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                       // this half-wave's 32 of the 64 head-dim columns
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(dob + (long long)q * C + hh * 32 + i * 8);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(ob + (long long)q * C + hh * 32 + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += (float)a[e] * (float)b[e];
        }
        D_q = dsum;
    }
    D_q += __shfl_xor(D_q, 32, 64);
    const float sl2 = scale * 1.4426950408889634f, lse2 = lse_q * 1.4426950408889634f;     // base-2 domain
    f32x16 acc[2];
    zero16(acc[0]); zero16(acc[1]);
    for (int c0 = 0; c0 < T; c0 += CH) {
        __syncthreads();
        stage_chunk(base + C, ld, c0, T, Ks, Kt, tid, blockDim.x);
        stage_chunk(base + 2 * C, ld, c0, T, Vs, nullptr, tid, blockDim.x);
        __syncthreads();
        const int nsb = active ? min(CH / 32, (T - c0 + 31) / 32) : 0;
        for (int sb = 0; sb < nsb; ++sb) {
            f32x16 pt, dpt;                                 // [key][query]
            zero16(pt); zero16(dpt);
            mma_chunk_own(Ks, sb, qf, pt, lane);            // S^T  = K Q^T
            mma_chunk_own(Vs, sb, dof, dpt, lane);          // dP^T = V dO^T
            const bool partial = c0 + sb * 32 + 32 > T;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(pt[r] * sl2 - lse2);
                if (partial) {
                    const int j = c0 + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    p = (j < T) ? p : 0.f;
                }
                dpt[r] = scale * p * (dpt[r] - D_q);        // dS^T
            }
            mma_acc_tr_chunk(dpt, Kt, sb, acc, lane);       // dQ[query][d] += dS[query][key] K[key][d]
        }
    }
    if (active) store_rows_global(acc, w, dqkv + (long long)n * T * ld + h * 64, ld, T, lane);
}

__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(3, 3))) void mha_bwd_dkv_blk_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, int T, int C, int heads, float scale, int xcd) {
    __shared__ __attribute__((aligned(16))) bf16_t Qs[CH * LD];
    __shared__ __attribute__((aligned(16))) bf16_t dOs[CH * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Qt[64 * TRS];
    __shared__ __attribute__((aligned(16))) bf16_t dOt[64 * TRS];
    __shared__ __attribute__((aligned(16))) float s_lse[CH], s_D[CH];
    const int tid = threadIdx.x, lane = tid & 63, w = blockIdx.z * (blockDim.x >> 6) + (tid >> 6), hh = lane >> 5;
    const bool active = 32 * w < T;
    int h, n;
    block_head(xcd, h, n);
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const bf16_t* dob = dout + (long long)n * T * C + h * 64;
    const bf16_t* ob = o + (long long)n * T * C + h * 64;
    const float* lse_h = lse + ((long long)n * heads + h) * T;
    const int key = 32 * w + (lane & 31);
    const bool key_ok = key < T;
    const float sl2 = scale * 1.4426950408889634f;
    bf16x8 kf[4], vf[4];
    own_frags(base + C, ld, key, T, kf, lane);
    own_frags(base + 2 * C, ld, key, T, vf, lane);
    f32x16 dk[2], dv[2];
    zero16(dk[0]); zero16(dk[1]); zero16(dv[0]); zero16(dv[1]);
    for (int c0 = 0; c0 < T; c0 += CH) {
        __syncthreads();
        stage_chunk(base, ld, c0, T, Qs, Qt, tid, blockDim.x);
        // dO chunk, and beside it D_i = sum_d dO_id O_id (eight lanes share a row: one 8-column piece each)
        for (int c = tid; c < CH * 8; c += blockDim.x) {
            const int row = c >> 3, kc = c & 7;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            float part = 0.f;
            if (c0 + row < T) {
                v = *reinterpret_cast<const bf16x8*>(dob + (long long)(c0 + row) * C + kc * 8);
                const bf16x8 ov = *reinterpret_cast<const bf16x8*>(ob + (long long)(c0 + row) * C + kc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) part += (float)v[e] * (float)ov[e];
            }
            *reinterpret_cast<bf16x8*>(&dOs[row * LD + kc * 8]) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) dOt[(kc * 8 + e) * TRS + row] = v[e];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (kc == 0) {
                s_D[row] = part;
                s_lse[row] = (c0 + row < T) ? lse_h[c0 + row] * 1.4426950408889634f : INFINITY;      // base-2 domain
            }
        }
        __syncthreads();
        const int nsb = active ? min(CH / 32, (T - c0 + 31) / 32) : 0;
        for (int sb = 0; sb < nsb; ++sb) {
            f32x16 p, dp;                                   // [query 32sb..][key]
            zero16(p); zero16(dp);
            mma_chunk_own(Qs, sb, kf, p, lane);             // S  = Q K^T
            mma_chunk_own(dOs, sb, vf, dp, lane);           // dP = dO V^T
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int i0 = sb * 32 + 8 * rg + 4 * hh;
                const float4 l4 = *reinterpret_cast<const float4*>(&s_lse[i0]);
                const float4 d4 = *reinterpret_cast<const float4*>(&s_D[i0]);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int r = rg * 4 + qq;
                    const float pv = key_ok ? __builtin_amdgcn_exp2f(p[r] * sl2 - lv[qq]) : 0.f;
                    p[r] = pv;
                    dp[r] = scale * pv * (dp[r] - dvv[qq]);   // dS
                }
            }
            mma_acc_tr_chunk(dp, Qt, sb, dk, lane);         // dK[key][d] += dS[query][key] Q[query][d]
            mma_acc_tr_chunk(p, dOt, sb, dv, lane);         // dV[key][d] += P[query][key] dO[query][d]
        }
    }
    if (!active) return;
    bf16_t* obase = dqkv + (long long)n * T * ld + h * 64;
    store_rows_global(dk, w, obase + C, ld, T, lane);
    store_rows_global(dv, w, obase + 2 * C, ld, T, lane);
}

}  // namespace

// PRX_MHA_TILES=1: route 64 < T <= 512 through the tile kernels as well (A/B measurements; keeps the T > 512 path tested)
static bool prx_mha_force_tiles() {
    static const bool v = [] { const char* e = getenv("PRX_MHA_TILES"); return e && e[0] == '1'; }();
    return v;
}

int prx_mha_fwd(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(T <= 64 && C == heads * 64, "mha: needs T <= 64 and head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(heads, N), dim3(128), 0, s, qkv, out, T, C, 0.125f, prx_xcd_local());
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_mha_bwd(const bf16_t* qkv, const bf16_t* dout, bf16_t* dqkv, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(T <= 64 && C == heads * 64, "mha bwd: needs T <= 64 and head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    hipLaunchKernelGGL(mha_bwd_kernel, dim3(heads, N), dim3(128), 0, s, qkv, dout, dqkv, T, C, 0.125f, prx_xcd_local());
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_mha_fwd_gen(const bf16_t* qkv, bf16_t* out, float* lse, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(C == heads * 64 && T >= 1, "mha(gen): needs head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    if (T <= 512 && !prx_mha_force_tiles()) {       // workgroups of <= 5 waves per (image, head), one wave per 32 tokens
        const int nw = ceil_div(T, 32), nsplit = ceil_div(nw, 5), wpb = ceil_div(nw, nsplit);
        hipLaunchKernelGGL(mha_fwd_blk_kernel, dim3(heads, N, nsplit), dim3(64 * wpb), 0, s, qkv, out, lse, T, C, heads, 0.125f, prx_xcd_local());
        PRX_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(mha_fwd_gen_kernel<false>, dim3(ceil_div(T, 64), heads, N), dim3(64), 0, s, qkv, out, lse, T, C, heads, 0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_mha_fwd_causal(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(C == heads * 64 && T >= 1, "mha(causal): needs head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    hipLaunchKernelGGL(mha_fwd_gen_kernel<true>, dim3(ceil_div(T, 64), heads, N), dim3(64), 0, s, qkv, out, (float*)nullptr, T, C, heads,
                       0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_mha_bwd_gen(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int N, int T,
                    int C, int heads, hipStream_t s) {
    PRX_REQUIRE(C == heads * 64 && T >= 1, "mha(gen) bwd: needs head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    if (T <= 512 && !prx_mha_force_tiles()) {
        const int nw = ceil_div(T, 32), nsplit = ceil_div(nw, 5), wpb = ceil_div(nw, nsplit);
        const dim3 g(heads, N, nsplit), b(64 * wpb);
        const int xcd = prx_xcd_local();
        hipLaunchKernelGGL(mha_bwd_dq_blk_kernel, g, b, 0, s, qkv, out, dout, lse, dqkv, T, C, heads, 0.125f, xcd);
        PRX_LAUNCH_CHECK();
        hipLaunchKernelGGL(mha_bwd_dkv_blk_kernel, g, b, 0, s, qkv, out, dout, lse, dqkv, T, C, heads, 0.125f, xcd);
        PRX_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(ceil_div(T, 64), heads, N);
    hipLaunchKernelGGL(mha_bwd_dq_gen_kernel, grid, dim3(64), 0, s, qkv, out, dout, lse, dqkv, T, C, heads, 0.125f);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mha_bwd_dkv_gen_kernel, grid, dim3(64), 0, s, qkv, out, dout, lse, dqkv, T, C, heads, 0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}
