// Prompt spherical-distance loss (pixray.py:268-280, ReplaceGrad 249-259) fused forward +
// analytic backward, and the VQGAN nearest-codebook quantiser (vector_quantize, vqgan.py:60-64)
// in exact fp32.
#include "prompt_vq.h"
#include <algorithm>

namespace {

// One wave per input row i.  rowloss[i] = sum_j sign(w) * 2*asin(|xn_i - en_j|/2)^2   (forward value, no stop)
// grad[i][:] = d/dx_i of  |w| * mean_ij max(sign(w)*d_ij, stop)   with mean over `denom` pairs.
__device__ __forceinline__ float prompt_loss_row(const float* __restrict__ x, const float* __restrict__ embed, int i, int lane,
                                                int n, int m, int D, float weight, float stop, float denom,
                                                float* __restrict__ rowloss, float* __restrict__ grad) {
    constexpr int MAXE = 16;  // D <= 1024
    const int ne = D / 64;
    float xv[MAXE], gacc[MAXE];
    float ss = 0.f;
    // loads with a clamped element index, outside `if (e < ne)`: a branch around each load made them D / 64 dependent round trips
    // (norms.hip ln_fwd_kernel's note); the arithmetic keeps its order
#pragma unroll
    for (int e = 0; e < MAXE; ++e) xv[e] = x[(size_t)i * D + (e < ne ? e : 0) * 64 + lane];
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
        if (e < ne) { ss += xv[e] * xv[e]; gacc[e] = 0.f; }
    const float xnorm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);      // F.normalize eps
    const float ixn = 1.f / xnorm;
    const float sgn = (weight > 0.f) ? 1.f : ((weight < 0.f) ? -1.f : 0.f);
    float loss = 0.f;
    for (int j = 0; j < m; ++j) {
        float ev[MAXE];
        float es = 0.f;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) ev[e] = embed[(size_t)j * D + (e < ne ? e : 0) * 64 + lane];
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (e < ne) es += ev[e] * ev[e];
        const float ien = 1.f / fmaxf(sqrtf(wave_sum(es)), 1e-12f);
        float r2 = 0.f;
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (e < ne) { float u = xv[e] * ixn - ev[e] * ien; r2 += u * u; }
        const float r = sqrtf(wave_sum(r2));
        const float a = asinf(fminf(r * 0.5f, 1.f));
        const float d = 2.f * a * a * sgn;
        loss += d;
        // gradient of max(d, stop): 1 above, 1/2 on a tie, 0 below (torch.maximum)
        float mask = (d > stop) ? 1.f : ((d == stop) ? 0.5f : 0.f);
        // dd/dr = 2a / sqrt(1 - r^2/4);  dr/dxn = u / r
        float coef = 0.f;
        if (r > 1e-20f) coef = sgn * mask * (2.f * a / sqrtf(fmaxf(1.f - 0.25f * r * r, 1e-20f))) / r;
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (e < ne) gacc[e] += coef * (xv[e] * ixn - ev[e] * ien);
    }
    // through F.normalize: g_x = (g - xn (xn.g)) / |x|
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
        if (e < ne) dot += gacc[e] * xv[e] * ixn;
    dot = wave_sum(dot);
    const float sc = fabsf(weight) / denom;
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
        if (e < ne) grad[(size_t)i * D + e * 64 + lane] = sc * (gacc[e] - xv[e] * ixn * dot) * ixn;
    if (lane == 0) rowloss[i] = loss;
    return loss;
}

// One wave per row, four rows per workgroup.  The scalar the loop adds up -- loss = |w| * sum(rowloss) / denom, pixray.py:280 --
// leaves the same launch (it used to take a torch reduction and a scalar multiply per Prompt and iteration): the workgroup that
// finishes last (a wrapping ticket counter the caller hands over zeroed; it is zero again once every workgroup drew) adds the row values in a fixed order.
__global__ __launch_bounds__(256) void prompt_loss_kernel(const float* __restrict__ x, const float* __restrict__ embed,
                                                          int n, int m, int D, float weight, float stop, float denom,
                                                          float* __restrict__ rowloss, float* __restrict__ grad, float* __restrict__ loss_out,
                                                          unsigned* __restrict__ ticket) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i < n) prompt_loss_row(x, embed, i, lane, n, m, D, weight, stop, denom, rowloss, grad);
    if (!loss_out) return;
    __shared__ unsigned last;
    __threadfence();                       // this workgroup's row values are visible device-wide before its ticket is drawn
    __syncthreads();
    // atomicInc wraps to zero when the last ticket is drawn: the word is zero again on exit WITHOUT a second store, so a launch
    // never depends on an earlier one having run to its last line
    if (threadIdx.x == 0) last = atomicInc(ticket, gridDim.x - 1) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();
    float t = 0.f;
    for (int r = lane; r < n; r += 64) t += *reinterpret_cast<const volatile float*>(rowloss + r);      // written by other workgroups: read past this CU's L1
    t = wave_sum(t);
    if (lane == 0) *loss_out = t * (fabsf(weight) / denom);
}

// e_hat = e/|e|  (slip.py:66) and its backward  de = (g - e_hat (e_hat.g))/|e|
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ e, float* __restrict__ out, int n, int D) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { float v = e[(size_t)i * D + c]; ss += v * v; }
    const float inv = 1.f / sqrtf(wave_sum(ss));
    for (int c = lane; c < D; c += 64) out[(size_t)i * D + c] = e[(size_t)i * D + c] * inv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ e, const float* __restrict__ g,
                                                         float* __restrict__ de, int n, int D) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < D; c += 64) {
        float v = e[(size_t)i * D + c];
        ss += v * v; dot += v * g[(size_t)i * D + c];
    }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float inv = 1.f / sqrtf(ss);
    for (int c = lane; c < D; c += 64) {
        float v = e[(size_t)i * D + c];
        de[(size_t)i * D + c] = (g[(size_t)i * D + c] - v * dot / ss) * inv;
    }
}

// ---------------------------------------------------------------------------------------------
// VQ: d[p][c] = (|x_p|^2 + |c|^2) - 2 x_p.c ; argmin over c (first index on ties) ; z_q = codebook[idx]
// Tokens are read from the NCHW z: x[p][k] = z[k*P + p].  fp32 FMA tiles: 64 tokens x 64 codes per block.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_rows_kernel(const float* __restrict__ w, float* __restrict__ out, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) { float v = w[(size_t)i * D + c]; ss += v * v; }
    ss = wave_sum(ss);
    if (lane == 0) out[i] = ss;
}

// 64 tokens x 128 codes per workgroup, 4 tokens x 8 codes per thread on packed fp32 FMAs (v_pk_fma_f32: two codes per lane and
// instruction); every (token, code) product is still ONE chain of fused multiply-adds over k = 0 .. D - 1, so distances -- and the
// selected codes -- are bit for bit those of the scalar formulation this replaces (91 -> see profiles; fp32 VALU bound).
typedef float vq_f32x2 __attribute__((ext_vector_type(2)));
constexpr int VQ_CODES = 128, VQ_CPAD = VQ_CODES + 4;
__global__ __launch_bounds__(256) void vq_dist_kernel(const float* __restrict__ z, long long tok_stride,
                                                      long long ch_stride, const float* __restrict__ codebook,
                                                      const float* __restrict__ cnorm, int P, int NC, int D,
                                                      float* __restrict__ pmin, int* __restrict__ pidx) {
    __shared__ __attribute__((aligned(16))) float Xs[16][64];
    __shared__ __attribute__((aligned(16))) float Cs[16][VQ_CPAD];
    __shared__ float red_v[64][16];
    __shared__ int red_i[64][16];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;   // tx: code quads tx and 16 + tx of the tile, ty: token quad
    const int p0 = blockIdx.y * 64, c0 = blockIdx.x * VQ_CODES;
    vq_f32x2 acc[4][4];                       // [token][code pair]: codes tx*4 + {0,1}, {2,3}, 64 + tx*4 + {0,1}, {2,3}
    float xn[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = vq_f32x2{0.f, 0.f};
    // the operands of K step k0 + 16 are requested (clamped addresses + select: no branch around a load) before the products of step
    // k0: as `load -> LDS -> barrier -> products` per step the kernel paid one exposed global round trip for each of its D / 16 steps
    float xr[4];
    float4 cr[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {        // 64 tokens x 16 k (strided source: NCHW or NHWC)
            const int e = tid + 256 * r;     // 0..1023
            const int kk = e >> 6, p = p0 + (e & 63);
            const float v = z[(long long)(k0 + kk) * ch_stride + (long long)(p < P ? p : P - 1) * tok_stride];
            xr[r] = (p < P) ? v : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {        // 128 codes x 16 k (rows of the codebook: k contiguous, 16-byte loads)
            const int e = tid + 256 * r;     // 0..511
            const int c = c0 + (e >> 2), q = e & 3;
            const float4 v = *reinterpret_cast<const float4*>(codebook + (size_t)(c < NC ? c : NC - 1) * D + k0 + 4 * q);
            cr[r] = (c < NC) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < D; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            Xs[e >> 6][e & 63] = xr[r];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = tid + 256 * r;
            const int cc = e >> 2, q = e & 3;
            Cs[4 * q + 0][cc] = cr[r].x; Cs[4 * q + 1][cc] = cr[r].y; Cs[4 * q + 2][cc] = cr[r].z; Cs[4 * q + 3][cc] = cr[r].w;
        }
        __syncthreads();
        if (k0 + 16 < D) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 xa = *reinterpret_cast<const float4*>(&Xs[kk][ty * 4]);
            const float4 ca = *reinterpret_cast<const float4*>(&Cs[kk][tx * 4]);
            const float4 cb = *reinterpret_cast<const float4*>(&Cs[kk][64 + tx * 4]);
            const vq_f32x2 cp[4] = {vq_f32x2{ca.x, ca.y}, vq_f32x2{ca.z, ca.w}, vq_f32x2{cb.x, cb.y}, vq_f32x2{cb.z, cb.w}};
            const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xn[i] = fmaf(xv[i], xv[i], xn[i]);
                const vq_f32x2 xx = vq_f32x2{xv[i], xv[i]};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_elementwise_fma(xx, cp[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float best = INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 8; ++j) {          // ascending code index: the first minimum wins
            const int c = c0 + (j < 4 ? tx * 4 + j : 64 + tx * 4 + (j - 4));
            if (c < NC) {
                const float a = (j & 1) ? acc[i][j >> 1].y : acc[i][j >> 1].x;
                float d = (xn[i] + cnorm[c]) - 2.f * a;
                if (d < best) { best = d; bi = c; }
            }
        }
        red_v[ty * 4 + i][tx] = best;
        red_i[ty * 4 + i][tx] = bi;
    }
    __syncthreads();
    if (tid < 64) {
        float best = red_v[tid][0]; int bi = red_i[tid][0];
        for (int j = 1; j < 16; ++j) {         // a thread's codes are not contiguous: ties go to the smaller index explicitly
            const float v = red_v[tid][j]; const int id = red_i[tid][j];
            if (v < best || (v == best && id < bi)) { best = v; bi = id; }
        }
        int p = p0 + tid;
        if (p < P) { pmin[(size_t)p * gridDim.x + blockIdx.x] = best; pidx[(size_t)p * gridDim.x + blockIdx.x] = bi; }
    }
}

__global__ __launch_bounds__(256) void vq_select_kernel(const float* __restrict__ pmin, const int* __restrict__ pidx,
                                                        int ntiles, const float* __restrict__ codebook, int D,
                                                        int* __restrict__ idx_out, float* __restrict__ zq, int P) {
    const int p = blockIdx.x;
    __shared__ float sv[256];
    __shared__ int si[256];
    float best = INFINITY; int bi = 0x7fffffff;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        float v = pmin[(size_t)p * ntiles + t]; int id = pidx[(size_t)p * ntiles + t];
        if (v < best || (v == best && id < bi)) { best = v; bi = id; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            float v = sv[threadIdx.x + o]; int id = si[threadIdx.x + o];
            if (v < sv[threadIdx.x] || (v == sv[threadIdx.x] && id < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = id; }
        }
        __syncthreads();
    }
    const int sel = si[0];
    if (threadIdx.x == 0 && idx_out) idx_out[p] = sel;
    for (int c = threadIdx.x; c < D; c += blockDim.x) zq[(size_t)p * D + c] = codebook[(size_t)sel * D + c];
}

}  // namespace

int prx_prompt_loss(const float* x, const float* embed, int n, int m, int D, float weight, float stop, float denom,
                    float* rowloss, float* grad, float* loss, unsigned* ticket, hipStream_t s) {
    PRX_REQUIRE(D % 64 == 0 && D <= 1024, "prompt_loss: D must be a multiple of 64 and <= 1024 (D=%d)", D);
    PRX_REQUIRE(!loss || ticket, "prompt_loss: the scalar result needs a zeroed ticket word");
    hipLaunchKernelGGL(prompt_loss_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, x, embed, n, m, D, weight, stop, denom,
                       rowloss, grad, loss, ticket);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_l2norm_fwd(const float* e, float* out, int n, int D, hipStream_t s) {
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, e, out, n, D);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_l2norm_bwd(const float* e, const float* g, float* de, int n, int D, hipStream_t s) {
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, s, e, g, de, n, D);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_sqnorm_rows(const float* w, float* out, int rows, int D, hipStream_t s) {
    hipLaunchKernelGGL(sqnorm_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w, out, rows, D);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_vq_nearest(const float* z, long long tok_stride, long long ch_stride, const float* codebook, const float* cnorm,
                   int P, int NC, int D, float* pmin, int* pidx, int* idx_out, float* zq, hipStream_t s) {
    PRX_REQUIRE(D % 16 == 0, "vq: D %% 16 != 0");
    PRX_REQUIRE(((uintptr_t)codebook & 15) == 0, "vq: the codebook must be 16-byte aligned");
    const int ntiles = ceil_div(NC, VQ_CODES);      // <= ceil(NC / 64): the scratch contract of prompt_vq.h still covers it
    hipLaunchKernelGGL(vq_dist_kernel, dim3(ntiles, ceil_div(P, 64)), dim3(256), 0, s, z, tok_stride, ch_stride, codebook,
                       cnorm, P, NC, D, pmin, pidx);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(vq_select_kernel, dim3(P), dim3(256), 0, s, pmin, pidx, ntiles, codebook, D, idx_out, zq, P);
    PRX_LAUNCH_CHECK();
    return 0;
}
