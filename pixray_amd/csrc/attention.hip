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

// load a [T][64] bf16 tile (row stride ld) into dst[t][d] and optionally dstT[d][t]; rows >= T are zero
__device__ __forceinline__ void load_tile(const bf16_t* src, long long ld, int T, bf16_t* dst, bf16_t* dstT, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        const int row = c >> 3, kc = c & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < T) v = *reinterpret_cast<const bf16x8*>(src + (long long)row * ld + kc * 8);
        if (dst) *reinterpret_cast<bf16x8*>(&dst[row * LD + kc * 8]) = v;
        if (dstT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dstT[(kc * 8 + e) * LD + row] = v[e];
        }
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

__global__ __launch_bounds__(64) void mha_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T,
                                                     int C, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE];
    bf16_t* Qs = smem;
    bf16_t* Ks = smem + TILE;
    bf16_t* Vt = smem + 2 * TILE;
    bf16_t* Ps = smem + 3 * TILE;
    const int lane = threadIdx.x;
    const int h = blockIdx.x, n = blockIdx.y;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    load_tile(base, ld, T, Qs, nullptr, lane);
    load_tile(base + C, ld, T, Ks, nullptr, lane);
    load_tile(base + 2 * C, ld, T, nullptr, Vt, lane);
    __syncthreads();
    f32x16 s[2][2];
    zero_acc(s);
    mma_64x64x64(Qs, Ks, s, lane);
    softmax_c_layout(s, scale, T, lane);
    store_c_tile(s, Ps, nullptr, lane);
    __syncthreads();
    f32x16 o[2][2];
    zero_acc(o);
    mma_64x64x64(Ps, Vt, o, lane);
    store_c_global(o, out + (long long)n * T * C + h * 64, C, T, lane);
}

__global__ __launch_bounds__(64) void mha_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                     bf16_t* __restrict__ dqkv, int T, int C, float scale) {
    // Four [64][72] bf16 LDS tiles (36.9 KB -> 4 workgroups per CU, all 768 (image, head) pairs resident at once):
    //   phase 1: T0=Q T1=K T2=V T3=dO           -> S = QK^T, P (registers), dP = dO V^T, dS (registers)
    //   phase 2: T0=dS T1=dS^T T2=P^T, T3 = K^T / Q^T / dO^T in turn (re-read from L2) -> dQ, dK, dV
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE];
    bf16_t* T0 = smem;
    bf16_t* T1 = smem + TILE;
    bf16_t* T2 = smem + 2 * TILE;
    bf16_t* T3 = smem + 3 * TILE;
    const int lane = threadIdx.x;
    const int h = blockIdx.x, n = blockIdx.y;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    const bf16_t* dobase = dout + (long long)n * T * C + h * 64;
    load_tile(base, ld, T, T0, nullptr, lane);
    load_tile(base + C, ld, T, T1, nullptr, lane);
    load_tile(base + 2 * C, ld, T, T2, nullptr, lane);
    load_tile(dobase, C, T, T3, nullptr, lane);
    __syncthreads();

    f32x16 p[2][2], dp[2][2];
    zero_acc(p);
    mma_64x64x64(T0, T1, p, lane);
    softmax_c_layout(p, scale, T, lane);
    zero_acc(dp);
    mma_64x64x64(T3, T2, dp, lane);
    // dS = scale * P o (dP - rowsum(P o dP))
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float dot = half_sum(p[mi][0][r] * dp[mi][0][r] + p[mi][1][r] * dp[mi][1][r]);
            dp[mi][0][r] = scale * p[mi][0][r] * (dp[mi][0][r] - dot);
            dp[mi][1][r] = scale * p[mi][1][r] * (dp[mi][1][r] - dot);
        }
    __syncthreads();  // all reads of Q/K/V/dO tiles done
    store_c_tile(dp, T0, T1, lane);        // T0 <- dS, T1 <- dS^T
    store_c_tile(p, nullptr, T2, lane);    // T2 <- P^T
    load_tile(base + C, ld, T, nullptr, T3, lane);      // T3 <- K^T
    __syncthreads();

    bf16_t* obase = dqkv + (long long)n * T * ld + h * 64;
    f32x16 acc[2][2];
    zero_acc(acc);
    mma_64x64x64(T0, T3, acc, lane);       // dQ = dS K
    store_c_global(acc, obase, ld, T, lane);
    __syncthreads();
    load_tile(base, ld, T, nullptr, T3, lane);           // T3 <- Q^T
    __syncthreads();
    zero_acc(acc);
    mma_64x64x64(T1, T3, acc, lane);       // dK = dS^T Q
    store_c_global(acc, obase + C, ld, T, lane);
    __syncthreads();
    load_tile(dobase, C, T, nullptr, T3, lane);          // T3 <- dO^T
    __syncthreads();
    zero_acc(acc);
    mma_64x64x64(T2, T3, acc, lane);       // dV = P^T dO
    store_c_global(acc, obase + 2 * C, ld, T, lane);
}

}  // namespace

int prx_mha_fwd(const bf16_t* qkv, bf16_t* out, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(T <= 64 && C == heads * 64, "mha: needs T <= 64 and head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(heads, N), dim3(64), 0, s, qkv, out, T, C, 0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_mha_bwd(const bf16_t* qkv, const bf16_t* dout, bf16_t* dqkv, int N, int T, int C, int heads, hipStream_t s) {
    PRX_REQUIRE(T <= 64 && C == heads * 64, "mha bwd: needs T <= 64 and head dim 64 (T=%d C=%d heads=%d)", T, C, heads);
    hipLaunchKernelGGL(mha_bwd_kernel, dim3(heads, N), dim3(64), 0, s, qkv, dout, dqkv, T, C, 0.125f);
    PRX_LAUNCH_CHECK();
    return 0;
}
