// STROTSS distance arithmetic of the StyleLoss plugin (Losses/StyleLoss.py:225-293): what the plugin composes from ~40
// elementwise / reduction launches over [1024 x 5000] and [1024 x 1024] distance matrices per evaluation (and their
// autograd mirror images, including a DENSE 22 GFLOP product against a gradient matrix that has one non-zero per row and per
// column), 12 evaluations per iteration of BASELINE.json configs[3].  Here, per evaluation:
//   * relaxed EMD (`style_loss`, 272-293): one pass over the product matrix G = X Y^T turns it into cosine (+ L2 for the
//     3-channel palette term) distances and keeps the row and column minima WITH their positions (64-bit packed
//     {ordered value, index} + atomicMin: independent of the order blocks arrive in, ties go to the smallest index); the
//     backward visits only the n + m selected pairs -- one workgroup per row of X, pairs in a fixed order, no atomics;
//   * self-similarity (`content_loss`, 246-265): mean |D(X,X) - D(Y,Y)| in one pass over both product matrices; the backward
//     writes the symmetrised d/dG of both (so the caller needs ONE product per operand) and the per-row norm coefficients.
// The products themselves stay plain library GEMMs (fp32).  HBM-bound byte work: coalesced row-major reads, one read of each
// matrix per pass.  Roundings follow the plugin's expressions ((G / |x|) / |y|, correctly rounded sqrt and division).
#include "common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned order_key(float f) {             // monotone float -> unsigned
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// `pairwise_distances_cos` (225-230) [+ sqrt of `pairwise_distances_sq_l2` (232-243) when L2]; in_clamp: the clamp passed s through
template <bool L2>
__device__ __forceinline__ float distance(float g, float xs, float xn, float ys, float yn, float dch, float* l2_out, bool* in_clamp) {
    float v = __fsub_rn(1.f, (g / xn) / yn);
    if (L2) {
        const float s = __fsub_rn(__fadd_rn(xs, ys), __fmul_rn(2.f, g));
        const float c = fminf(fmaxf(s, 1e-5f), 1e5f);
        const float l2 = sqrtf(c / dch);
        if (l2_out) *l2_out = l2;
        if (in_clamp) *in_clamp = s >= 1e-5f && s <= 1e5f;
        v = __fadd_rn(v, l2);
    }
    return v;
}

constexpr int MIN_ROWS = 32;     // rows of G per block of the minima pass

// grid (ceil(m / 256), ceil(n / MIN_ROWS)); thread = one column, MIN_ROWS rows
template <bool L2>
__global__ __launch_bounds__(256) void cosdist_minima_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ xs,
                                                             const float* __restrict__ ys, int n, int m, float dch,
                                                             u64* __restrict__ rowpack, u64* __restrict__ colpack) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * MIN_ROWS;
    const bool jin = j < m;
    const float ysj = jin ? ys[j] : 1.f;
    const float yn = sqrtf(ysj);
    u64 cbest = ~0ull;
    for (int r = 0; r < MIN_ROWS; ++r) {
        const int i = i0 + r;
        if (i >= n) break;                                   // uniform
        const float xsi = xs[i], xn = sqrtf(xsi);
        u64 p = ~0ull;
        if (jin) {
            const float v = distance<L2>(G[(size_t)i * ldg + j], xsi, xn, ysj, yn, dch, nullptr, nullptr);
            const u64 key = (u64)order_key(v) << 32;
            p = key | (unsigned)j;
            const u64 pc = key | (unsigned)i;
            cbest = pc < cbest ? pc : cbest;
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const u64 o = __shfl_xor(p, off);
            p = o < p ? o : p;
        }
        if ((threadIdx.x & 63) == 0 && p != ~0ull) atomicMin(&rowpack[i], p);
    }
    if (jin && cbest != ~0ull) atomicMin(&colpack[j], cbest);
}

__device__ __forceinline__ double block_sum(double v, double* sh) {      // 256 threads; result in every thread
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// stats: {max(rmean, cmean), rmean, cmean}
__global__ __launch_bounds__(256) void remd_finalize_kernel(const u64* __restrict__ rowpack, const u64* __restrict__ colpack, int n, int m,
                                                            float* __restrict__ stats) {
    __shared__ double sh[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)key_value((unsigned)(rowpack[i] >> 32));
    for (int j = threadIdx.x; j < m; j += 256) b += (double)key_value((unsigned)(colpack[j] >> 32));
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0) {
        const float rm = (float)(a / n), cm = (float)(b / m);
        stats[0] = fmaxf(rm, cm); stats[1] = rm; stats[2] = cm;
    }
}

constexpr int REMD_DMAX = 4096;          // channels per column (LDS row accumulator)
constexpr int REMD_CHUNK = 64 * 256;     // columns per pass of the match scan (256 ballot words)
constexpr int REMD_PAIRS = 32;           // selected pairs per workgroup of the backward

// The backward is dX[i, :] = sum over the selected pairs (i, j) of coef(i, j) y_j + b(i, j) x_i: the row minimum of row i (weight wr)
// and every column whose minimum sits in row i (weight wc).  In feature space the column minima are very unevenly spread
// (a few rows of X are the nearest neighbour of thousands of style columns), so a row's pairs are cut into chunks of
// REMD_PAIRS, one workgroup each: count -> scan -> chunk sums -> (rows with several chunks) ordered reduction.  No
// floating-point atomics anywhere: a row's pairs are always added in ascending column order, chunk by chunk.
struct RemdW { float wr, wc; };
__device__ __forceinline__ RemdW remd_weights(const float* stats, const float* gout, int n, int m) {
    const float rm = stats[1], cm = stats[2], g = gout[0];
    // torch.max(a, b) of two scalars (291): the gradient goes to the larger one, half to each on a tie
    RemdW w;
    w.wr = g * (rm > cm ? 1.f : (rm == cm ? 0.5f : 0.f)) / (float)n;
    w.wc = g * (cm > rm ? 1.f : (rm == cm ? 0.5f : 0.f)) / (float)m;
    return w;
}

// cnt[i] = number of columns whose minimum sits in row i (0 when the column branch carries no gradient)
__global__ __launch_bounds__(256) void remd_count_kernel(const u64* __restrict__ colpack, int n, int m, const float* __restrict__ stats,
                                                         const float* __restrict__ gout, int* __restrict__ cnt) {
    __shared__ int sh[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int c = 0;
    if (remd_weights(stats, gout, n, m).wc != 0.f)
        for (int j = threadIdx.x; j < m; j += 256) c += (int)(unsigned)(colpack[j] & 0xffffffffull) == i;
#pragma unroll
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) sh[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[i] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// chunk_start[i] = first chunk of row i (every row has at least one), chunk_start[n] = number of chunks; one workgroup
__global__ __launch_bounds__(1024) void remd_scan_kernel(const int* __restrict__ cnt, int n, int* __restrict__ chunk_start) {
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int v = i < n ? max(1, (cnt[i] + REMD_PAIRS - 1) / REMD_PAIRS) : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = sh[threadIdx.x], base = carry;
        if (i < n) chunk_start[i] = base + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_start[n] = carry;
}

// workgroup b = chunk b: pairs [c P, (c + 1) P) of its row's ascending column list (+ the row minimum's pair in chunk 0)
template <bool L2>
__global__ __launch_bounds__(256) void remd_chunk_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                                         const float* __restrict__ Y, int ldy, int d, const float* __restrict__ xs,
                                                         const float* __restrict__ ys, const u64* __restrict__ rowpack,
                                                         const u64* __restrict__ colpack, int n, int m, const float* __restrict__ stats,
                                                         const float* __restrict__ gout, const int* __restrict__ chunk_start,
                                                         float* __restrict__ partial, float* __restrict__ dX, int lddx) {
    __shared__ float acc[REMD_DMAX];
    __shared__ u64 masks[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (b >= chunk_start[n]) return;
    int lo_i = 0, hi_i = n - 1;                              // the row of chunk b: the last i with chunk_start[i] <= b
    while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i + 1) >> 1;
        if (chunk_start[mid] <= b) lo_i = mid; else hi_i = mid - 1;
    }
    const int i = lo_i, c = b - chunk_start[i], nchunks = chunk_start[i + 1] - chunk_start[i];
    const RemdW w = remd_weights(stats, gout, n, m);
    const float dch = (float)d;
    const float xsi = xs[i], xn = sqrtf(xsi);
    const float* x = X + (size_t)i * ldx;
    for (int ch = tid; ch < d; ch += 256) acc[ch] = 0.f;     // each thread owns its channels: no barrier between pairs
    float bsum = 0.f;                                        // coefficient of x itself (the same in every thread)
    auto pair = [&](int j, float wt) {
        const float ysj = ys[j], yn = sqrtf(ysj);
        const float gij = G[(size_t)i * ldg + j];
        // M = 1 - G / (|x| |y|):  dM/dx = -y / (|x| |y|) + G x / (|x|^3 |y|)
        float ay = -wt / (xn * yn);
        float bx = wt * gij / (xsi * xn * yn);
        if (L2) {
            float l2; bool in;
            distance<true>(gij, xsi, xn, ysj, yn, dch, &l2, &in);
            if (in) { const float e = wt / (dch * l2); bx += e; ay -= e; }    // d sqrt(s / d)/dx = (x - y) / (d sqrt(s / d))
        }
        bsum += bx;
        const float* y = Y + (size_t)j * ldy;
        for (int ch = tid; ch < d; ch += 256) acc[ch] = fmaf(ay, y[ch], acc[ch]);
    };
    if (c == 0 && w.wr != 0.f) pair((int)(unsigned)(rowpack[i] & 0xffffffffull), w.wr);
    if (w.wc != 0.f) {
        const int lo = c * REMD_PAIRS, hi = lo + REMD_PAIRS;
        int rank = 0;                                        // matches seen so far (uniform)
        for (int j0 = 0; j0 < m && rank < hi; j0 += REMD_CHUNK) {
            __syncthreads();
            const int nw = min((m - j0 + 63) >> 6, 256);
            for (int q = wave; q < nw; q += 4) {
                const int j = j0 + q * 64 + lane;
                const bool hit = j < m && (int)(unsigned)(colpack[j] & 0xffffffffull) == i;
                const u64 bal = __ballot(hit);
                if (lane == 0) masks[q] = bal;
            }
            __syncthreads();
            for (int q = 0; q < nw && rank < hi; ++q) {      // uniform: every thread walks the same bits in the same order
                u64 bal = masks[q];
                const int pc = __popcll(bal);
                if (rank + pc <= lo) { rank += pc; continue; }
                while (bal && rank < hi) {
                    const int k = __ffsll((long long)bal) - 1;
                    bal &= bal - 1;
                    if (rank >= lo) pair(j0 + q * 64 + k, w.wc);
                    ++rank;
                }
            }
        }
    }
    float* out = nchunks == 1 ? dX + (size_t)i * lddx : partial + (size_t)b * d;
    for (int ch = tid; ch < d; ch += 256) out[ch] = fmaf(bsum, x[ch], acc[ch]);
}

// rows cut into several chunks: the chunk sums in order
__global__ __launch_bounds__(256) void remd_reduce_kernel(const int* __restrict__ chunk_start, const float* __restrict__ partial, int d,
                                                          float* __restrict__ dX, int lddx) {
    const int i = blockIdx.x;
    const int c0 = chunk_start[i], c1 = chunk_start[i + 1];
    if (c1 - c0 <= 1) return;
    for (int ch = threadIdx.x; ch < d; ch += 256) {
        float v = partial[(size_t)c0 * d + ch];
        for (int c = c0 + 1; c < c1; ++c) v += partial[(size_t)c * d + ch];
        dX[(size_t)i * lddx + ch] = v;
    }
}

// ---- self-similarity ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cosd(float g, float an_i, float an_j) { return __fsub_rn(1.f, (g / an_i) / an_j); }

// one workgroup per row: partial[i] = sum_j |Dx(i, j) - Dy(i, j)|
__global__ __launch_bounds__(256) void selfsim_fwd_kernel(const float* __restrict__ Gx, int ldgx, const float* __restrict__ xs,
                                                          const float* __restrict__ Gy, int ldgy, const float* __restrict__ ys, int n,
                                                          double* __restrict__ partial) {
    __shared__ double sh[4];
    const int i = blockIdx.x;
    const float ax = sqrtf(xs[i]), ay = sqrtf(ys[i]);
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float mx = cosd(Gx[(size_t)i * ldgx + j], ax, sqrtf(xs[j]));
        const float my = cosd(Gy[(size_t)i * ldgy + j], ay, sqrtf(ys[j]));
        s += (double)fabsf(__fsub_rn(mx, my));
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) partial[i] = s;
}
__global__ __launch_bounds__(256) void selfsim_finalize_kernel(const double* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = (float)(s / ((double)n * (double)n));
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// one workgroup per row i: Sx[i, j] = dL/dGx[i, j] + dL/dGx[j, i] (same for y), cx[i] = (dL/d|x_i|) / |x_i| (same for y), so that
// dX = Sx X + cx (.) X.  Reads row i (coalesced) and column i (strided; the matrices are L2-sized) of both products.
__global__ __launch_bounds__(256) void selfsim_bwd_kernel(const float* __restrict__ Gx, int ldgx, const float* __restrict__ xs,
                                                          const float* __restrict__ Gy, int ldgy, const float* __restrict__ ys, int n,
                                                          const float* __restrict__ gout, float* __restrict__ Sx, float* __restrict__ Sy,
                                                          int lds_, float* __restrict__ cx, float* __restrict__ cy) {
    __shared__ double sh[4];
    const int i = blockIdx.x;
    const float k = gout[0] / ((float)n * (float)n);
    const float xsi = xs[i], ysi = ys[i], ax = sqrtf(xsi), ay = sqrtf(ysi);
    double nx = 0.0, ny = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float axj = sqrtf(xs[j]), ayj = sqrtf(ys[j]);
        const float gx_r = Gx[(size_t)i * ldgx + j], gy_r = Gy[(size_t)i * ldgy + j];       // D(i, j): i in the row role
        const float gx_c = Gx[(size_t)j * ldgx + i], gy_c = Gy[(size_t)j * ldgy + i];       // D(j, i): i in the column role
        const float s_r = k * sgn(__fsub_rn(cosd(gx_r, ax, axj), cosd(gy_r, ay, ayj)));
        const float s_c = k * sgn(__fsub_rn(cosd(gx_c, axj, ax), cosd(gy_c, ayj, ay)));
        const float ix = 1.f / (ax * axj), iy = 1.f / (ay * ayj);
        Sx[(size_t)i * lds_ + j] = -(s_r + s_c) * ix;
        Sy[(size_t)i * lds_ + j] = (s_r + s_c) * iy;
        // D = 1 - G / (a_i a_j):  dD/da_i = G / (a_i^2 a_j), from both roles
        nx += (double)((s_r * gx_r + s_c * gx_c) * ix);
        ny += (double)((s_r * gy_r + s_c * gy_c) * iy);
    }
    nx = block_sum(nx, sh);
    ny = block_sum(ny, sh);
    if (threadIdx.x == 0) { cx[i] = (float)(nx / (double)xsi); cy[i] = (float)(-ny / (double)ysi); }
}

}  // namespace

extern "C" {

long long prx_strotss_remd_bwd_workspace_bytes(int n, int m, int d) {
    if (n < 1 || m < 1 || d < 1) return -1;
    const long long chunks = (long long)n + (m + REMD_PAIRS - 1) / REMD_PAIRS;
    return (long long)sizeof(int) * (2ll * n + 2) + (long long)sizeof(float) * chunks * d + 64;
}

int prx_strotss_remd_fwd(const float* G, int ldg, const float* xs, const float* ys, int n, int m, int l2, int d,
                         unsigned long long* rowpack, unsigned long long* colpack, float* stats, hipStream_t s) {
    PRX_REQUIRE(G && xs && ys && rowpack && colpack && stats, "strotss remd: NULL argument");
    PRX_REQUIRE(n >= 1 && m >= 1 && ldg >= m && d >= 1, "strotss remd: bad shape n=%d m=%d ldg=%d d=%d", n, m, ldg, d);
    PRX_CHECK_HIP(hipMemsetAsync(rowpack, 0xff, sizeof(u64) * (size_t)n, s));
    PRX_CHECK_HIP(hipMemsetAsync(colpack, 0xff, sizeof(u64) * (size_t)m, s));
    const dim3 grid((m + 255) / 256, (n + MIN_ROWS - 1) / MIN_ROWS);
    if (l2) hipLaunchKernelGGL(cosdist_minima_kernel<true>, grid, dim3(256), 0, s, G, ldg, xs, ys, n, m, (float)d, rowpack, colpack);
    else    hipLaunchKernelGGL(cosdist_minima_kernel<false>, grid, dim3(256), 0, s, G, ldg, xs, ys, n, m, (float)d, rowpack, colpack);
    hipLaunchKernelGGL(remd_finalize_kernel, dim3(1), dim3(256), 0, s, rowpack, colpack, n, m, stats);
    PRX_CHECK_HIP(hipGetLastError());
    return 0;
}

int prx_strotss_remd_bwd(const float* G, int ldg, const float* X, int ldx, const float* Y, int ldy, int d, const float* xs, const float* ys,
                         const unsigned long long* rowpack, const unsigned long long* colpack, int n, int m, int l2, const float* stats,
                         const float* g_out, void* workspace, long long workspace_bytes, float* dX, int lddx, hipStream_t s) {
    PRX_REQUIRE(G && X && Y && xs && ys && rowpack && colpack && stats && g_out && dX && workspace, "strotss remd bwd: NULL argument");
    PRX_REQUIRE(n >= 1 && m >= 1 && d >= 1 && d <= REMD_DMAX && ldg >= m && ldx >= d && ldy >= d && lddx >= d,
                "strotss remd bwd: bad shape n=%d m=%d d=%d (d <= %d)", n, m, d, REMD_DMAX);
    PRX_REQUIRE(workspace_bytes >= prx_strotss_remd_bwd_workspace_bytes(n, m, d) && ((uintptr_t)workspace & 15) == 0,
                "strotss remd bwd: workspace of %lld bytes, 16-byte aligned, needed", prx_strotss_remd_bwd_workspace_bytes(n, m, d));
    int* cnt = (int*)workspace;
    int* chunk_start = cnt + n;
    float* partial = (float*)(((uintptr_t)(chunk_start + n + 1) + 15) & ~(uintptr_t)15);
    const int max_chunks = n + (m + REMD_PAIRS - 1) / REMD_PAIRS;
    hipLaunchKernelGGL(remd_count_kernel, dim3(n), dim3(256), 0, s, colpack, n, m, stats, g_out, cnt);
    hipLaunchKernelGGL(remd_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, n, chunk_start);
    if (l2) hipLaunchKernelGGL(remd_chunk_kernel<true>, dim3(max_chunks), dim3(256), 0, s, G, ldg, X, ldx, Y, ldy, d, xs, ys, rowpack, colpack,
                               n, m, stats, g_out, chunk_start, partial, dX, lddx);
    else    hipLaunchKernelGGL(remd_chunk_kernel<false>, dim3(max_chunks), dim3(256), 0, s, G, ldg, X, ldx, Y, ldy, d, xs, ys, rowpack, colpack,
                               n, m, stats, g_out, chunk_start, partial, dX, lddx);
    hipLaunchKernelGGL(remd_reduce_kernel, dim3(n), dim3(256), 0, s, chunk_start, partial, d, dX, lddx);
    PRX_CHECK_HIP(hipGetLastError());
    return 0;
}

int prx_strotss_selfsim_fwd(const float* Gx, int ldgx, const float* xs, const float* Gy, int ldgy, const float* ys, int n,
                            double* partial, float* out, hipStream_t s) {
    PRX_REQUIRE(Gx && xs && Gy && ys && partial && out, "strotss selfsim: NULL argument");
    PRX_REQUIRE(n >= 1 && ldgx >= n && ldgy >= n, "strotss selfsim: bad shape n=%d", n);
    hipLaunchKernelGGL(selfsim_fwd_kernel, dim3(n), dim3(256), 0, s, Gx, ldgx, xs, Gy, ldgy, ys, n, partial);
    hipLaunchKernelGGL(selfsim_finalize_kernel, dim3(1), dim3(256), 0, s, partial, n, out);
    PRX_CHECK_HIP(hipGetLastError());
    return 0;
}

int prx_strotss_selfsim_bwd(const float* Gx, int ldgx, const float* xs, const float* Gy, int ldgy, const float* ys, int n,
                            const float* g_out, float* Sx, float* Sy, int lds, float* cx, float* cy, hipStream_t s) {
    PRX_REQUIRE(Gx && xs && Gy && ys && g_out && Sx && Sy && cx && cy, "strotss selfsim bwd: NULL argument");
    PRX_REQUIRE(n >= 1 && ldgx >= n && ldgy >= n && lds >= n, "strotss selfsim bwd: bad shape n=%d", n);
    hipLaunchKernelGGL(selfsim_bwd_kernel, dim3(n), dim3(256), 0, s, Gx, ldgx, xs, Gy, ldgy, ys, n, g_out, Sx, Sy, lds, cx, cy);
    PRX_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
