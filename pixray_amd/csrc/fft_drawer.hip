// The fft drawer's spectrum -> image map (BASELINE.json configs[3]; /root/reference/fftdrawer.py:45-62, 79-86:
// aphantasia's fft_image(sd = 0.01, decay_power) + to_valid_rgb(colors = 1.5), evaluated with contrast = 0.9) and its
// backward, on MI355X, without an FFT library:
//
//     image = irfft2(scale * spectrum, s = (H, W), norm = "ortho");  image *= contrast / std(image);
//     rgb   = sigmoid(colour matrix applied to image)
//
// The inverse real transform of a 3 x H x (W/2+1) spectrum is two small DENSE contractions (a few GFLOP at 512 x 512), so it
// runs as exact-f32 GEMMs on the engine (gemm.h, v_mfma_f32_32x32x2_f32: an fmaf chain per output, fp32 end to end) against
// twiddle matrices built once per canvas size in float64 on the host:
//
//   x[c,h,w] = Re sum_v g_v e^{i th(v,w)} sum_u S[c,u,v] e^{i ph(u,h)} / sqrt(HW)      g_v = 1 (v = 0, and v = W/2 when W is even), else 2
//
//   pack    Bt1[(c, part', u)][(v, part)]   part' = 0: ( re, im) * scale[u,v];  part' = 1: (-im, re) * scale[u,v]   (i * S)
//   GEMM 1  C1[w][(c, part', u)] = sum_{(v,part)} A1[w][(v,part)] Bt1[..][(v,part)],  A1[w][(v,0)] = g_v cos th / sqrt(HW),
//           A1[w][(v,1)] = -g_v sin th / sqrt(HW)      ->  C1[w][(c,0,u)] = Zr[c,u,w],  C1[w][(c,1,u)] = -Zi[c,u,w]
//   GEMM 2  (per channel)  x_c[h][w] = sum_{(part',u)} T2[h][(part',u)] C1[w][(c,part',u)],  T2 = [cos ph | sin ph]
//   tail    a = contrast / std (unbiased, over all 3HW values; sums in float64), y = a x, z_d = sum_c y_c M[c][d], rgb = sigmoid(z)
//
// Backward = the transposed chain: tail backward (two passes: the std couples every element through sum(gy x)), writing dx
// TRANSPOSED [c][w][h] so that both backward GEMMs find their contraction index contiguous; GEMM 2' per channel
// dC1T[(c,k)][w] = T2T[k][h] dxT_c[w][h]; GEMM 1' dBt1[n][(v,part)] = dC1T[n][w] A1T[(v,part)][w]; unpack folds the two
// copies of the spectrum and the scale back into d(spectrum).
// Every row stride is padded to 4 floats (the engine's 16-byte operand chunks); padding columns of the operands are zero.
#include "gemm.h"
#include "../../include/prx.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
inline int up4(int v) { return (v + 3) & ~3; }
inline int fgrid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 8192); }

// params [3][H][Wf][2] (Wf >= Wh columns; the surplus one of an odd width is ignored) -> Bt1 [3 * HP2][K1p], zero padded
__global__ __launch_bounds__(256) void fft_pack_kernel(const float* __restrict__ prm, const float* __restrict__ scale, float* __restrict__ bt1,
                                                       int H, int Wf, int Wh, int HP2, int K1p) {
    const size_t total = (size_t)3 * HP2 * K1p;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % K1p);
        const size_t row = i / K1p;
        const int k = (int)(row % HP2), c = (int)(row / HP2);
        float v = 0.f;
        if (k < 2 * H && kk < 2 * Wh) {
            const int rot = k / H, u = k - rot * H, col = kk >> 1, part = kk & 1;
            const float* p = prm + (((size_t)c * H + u) * Wf + col) * 2;
            const float s = scale[(size_t)u * Wh + col];
            v = rot == 0 ? p[part] * s : (part == 0 ? -p[1] * s : p[0] * s);
        }
        bt1[i] = v;
    }
}

// Moments in float64 with a FIXED summation order (the image's std scales every pixel and every gradient entry: the drawer is
// bit-reproducible run to run, like the rest of the path): every block writes its three partial sums, one block adds them up
// part[b][0] = sum x, part[b][1] = sum x^2, part[b][2] = sum x * y (when y is given) over block b's grid-stride share
constexpr int FFT_MOM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void fft_moments_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, double* __restrict__ part) {
    __shared__ double red[3][4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        s0 += v; s1 += v * v;
        if (y) s2 += v * (double)y[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        part[(size_t)blockIdx.x * 3 + k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    }
}
// acc[k] = sum_b part[b][k], one block: thread t adds blocks t, t + 256, ... in that order, then a fixed butterfly and wave order
__global__ __launch_bounds__(256) void fft_moments_final_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ acc) {
    __shared__ double red[3][4];
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += part[(size_t)b * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s[0]; red[1][wave] = s[1]; red[2][wave] = s[2]; }
    __syncthreads();
    if (threadIdx.x < 3) acc[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

struct FftStd { float mean, sigma; };
__device__ __forceinline__ FftStd fft_std(const double* acc, size_t n) {
    const double mean = acc[0] / (double)n;
    const double var = (acc[1] - (double)n * mean * mean) / (double)(n - 1);         // torch.std: unbiased
    return FftStd{(float)mean, (float)sqrt(var > 0.0 ? var : 0.0)};
}

// x [3][P] -> rgb [3][P] = sigmoid(sum_c (contrast / sigma) x_c M[c][d]),  P = H * W;  cm: colour matrix [c][d], row-major
__global__ __launch_bounds__(256) void fft_tail_fwd_kernel(const float* __restrict__ x, const double* __restrict__ acc, float contrast,
                                                           const float* __restrict__ cm, float* __restrict__ rgb, size_t P) {
    const FftStd st = fft_std(acc, 3 * P);
    const float a = contrast / st.sigma;
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = cm[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
        const float y0 = a * x[i], y1 = a * x[P + i], y2 = a * x[2 * P + i];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float z = y0 * m[d] + y1 * m[3 + d] + y2 * m[6 + d];
            rgb[d * P + i] = 1.f / (1.f + __expf(-z));
        }
    }
}

// pass 1 of the tail backward: gy_c = sum_d g_d s_d (1 - s_d) M[c][d] (s recomputed from x), stored
__global__ __launch_bounds__(256) void fft_tail_bwd_gy_kernel(const float* __restrict__ x, const float* __restrict__ g, const double* __restrict__ acc,
                                                              float contrast, const float* __restrict__ cm, float* __restrict__ gy, size_t P) {
    const FftStd st = fft_std(acc, 3 * P);
    const float a = contrast / st.sigma;
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = cm[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
        const float y0 = a * x[i], y1 = a * x[P + i], y2 = a * x[2 * P + i];
        float gz[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float z = y0 * m[d] + y1 * m[3 + d] + y2 * m[6 + d];
            const float s = 1.f / (1.f + __expf(-z));
            gz[d] = g[d * P + i] * s * (1.f - s);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) gy[c * P + i] = gz[0] * m[3 * c] + gz[1] * m[3 * c + 1] + gz[2] * m[3 * c + 2];
    }
}

// pass 2: dx = a gy - (a / sigma) S (x - mean) / ((n - 1) sigma),  S = sum gy x (bacc[2]);  written transposed: dxT[c][w][h], row stride HP
__global__ __launch_bounds__(256) void fft_tail_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ gy, const double* __restrict__ acc,
                                                              const double* __restrict__ bacc, float contrast, float* __restrict__ dxT,
                                                              int H, int W, int HP) {
    const size_t P = (size_t)H * W, n = 3 * P;
    const FftStd st = fft_std(acc, n);
    const float a = contrast / st.sigma;
    const float k = (float)((double)a * bacc[2] / ((double)st.sigma * (double)st.sigma * (double)(n - 1)));
    const size_t total = (size_t)3 * W * HP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % HP);
        const size_t r = i / HP;
        const int w = (int)(r % W), c = (int)(r / W);
        float v = 0.f;
        if (h < H) {
            const size_t src = (size_t)c * P + (size_t)h * W + w;
            v = a * gy[src] - k * (x[src] - st.mean);
        }
        dxT[i] = v;
    }
}

// dP [3 * HP2][K1p] -> d(spectrum) [3][H][Wf][2]: both copies of the spectrum (identity and i *) fold back, times the scale
__global__ __launch_bounds__(256) void fft_unpack_kernel(const float* __restrict__ dP, const float* __restrict__ scale, float* __restrict__ gprm,
                                                         int H, int Wf, int Wh, int HP2, int K1p) {
    const size_t total = (size_t)3 * H * Wf;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % Wf);
        const size_t r = i / Wf;
        const int u = (int)(r % H), c = (int)(r / H);
        float gre = 0.f, gim = 0.f;
        if (v < Wh) {
            const float* p0 = dP + ((size_t)c * HP2 + u) * K1p + 2 * v;            // part' = 0 row: (re, im)
            const float* p1 = dP + ((size_t)c * HP2 + H + u) * K1p + 2 * v;        // part' = 1 row: (-im, re)
            const float s = scale[(size_t)u * Wh + v];
            gre = (p0[0] + p1[1]) * s;
            gim = (p0[1] - p1[0]) * s;
        }
        gprm[2 * i] = gre;
        gprm[2 * i + 1] = gim;
    }
}
}  // namespace

struct prx_fft_drawer {
    int W, H, Wf, Wh, HP, HP2, WP, K1p;
    float* scale = nullptr;      // [H][Wh]
    float* A1 = nullptr;         // [W][K1p]
    float* A1T = nullptr;        // [K1p][WP]
    float* T2 = nullptr;         // [H][HP2]
    float* T2T = nullptr;        // [HP2][HP]
    float* cm = nullptr;         // colour matrix [c][d]
    float* bt1 = nullptr;        // [3 HP2][K1p]
    float* c1 = nullptr;         // [W][3 HP2]
    float* x = nullptr;          // [3][H][W], kept from synth for the backward
    float* gy = nullptr;         // [3][H][W]
    float* dxT = nullptr;        // [3][W][HP]
    float* dc1t = nullptr;       // [3 HP2][WP]
    float* dP = nullptr;         // [3 HP2][K1p]
    double* acc = nullptr;       // [6]: forward sums (0..2), backward sums (3..5); then [FFT_MOM_BLOCKS][3] block partials
    float* ws = nullptr;         // split-K workspace of the engine
    size_t ws_bytes = 0;
    float contrast = 0.9f;
    GemmCtx gctx;
};

namespace {
int upload(float** dst, const std::vector<float>& v) {
    PRX_CHECK_HIP(hipMalloc(dst, v.size() * sizeof(float)));
    PRX_CHECK_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
int gemm_f32(prx_fft_drawer* h, const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* out, int ldc, hipStream_t s) {
    GemmDesc d;
    d.f32 = 1;
    d.A = A; d.lda = lda; d.B = B; d.ldb = ldb;
    d.M = M; d.N = N; d.K = K;
    d.out_f32 = out; d.ldc_f32 = ldc;
    return prx_gemm_launch(d, h->ws, h->ws_bytes, s, &h->gctx);
}
}  // namespace

extern "C" {

prx_fft_drawer* prx_fft_drawer_create(int W, int H, float decay, float colors) {
    if (W < 2 || H < 2) { prx_set_error("fft drawer: canvas %d x %d too small", W, H); return nullptr; }
    prx_fft_drawer* h = new prx_fft_drawer();
    h->W = W; h->H = H;
    h->Wh = W / 2 + 1;
    h->Wf = W / 2 + (W % 2 ? 2 : 1);        // the lucid frequency helper keeps one surplus column for an odd width
    h->HP = up4(H); h->HP2 = up4(2 * H); h->WP = up4(W); h->K1p = up4(2 * h->Wh);
    const int Wh = h->Wh, HP = h->HP, HP2 = h->HP2, WP = h->WP, K1p = h->K1p;
    const double two_pi = 6.283185307179586476925286766559;
    // scale[u][v] = sqrt(W H) / max(|f|, 1 / max(W, H)) ^ decay
    std::vector<float> scale((size_t)H * Wh);
    for (int u = 0; u < H; ++u) {
        const double fy = (double)(u < (H + 1) / 2 ? u : u - H) / H;
        for (int v = 0; v < Wh; ++v) {
            const double fx = (double)(v < (W + 1) / 2 ? v : v - W) / W;
            const double f = std::max(std::sqrt(fx * fx + fy * fy), 1.0 / std::max(W, H));
            scale[(size_t)u * Wh + v] = (float)(std::sqrt((double)W * H) / std::pow(f, (double)decay));
        }
    }
    std::vector<float> A1((size_t)W * K1p, 0.f), A1T((size_t)K1p * WP, 0.f), T2((size_t)H * HP2, 0.f), T2T((size_t)HP2 * HP, 0.f);
    const double nrm = 1.0 / std::sqrt((double)W * H);
    for (int w = 0; w < W; ++w)
        for (int v = 0; v < Wh; ++v) {
            const double g = (v == 0 || (W % 2 == 0 && v == W / 2)) ? 1.0 : 2.0;
            const double th = two_pi * (double)(((long long)v * w) % W) / W;
            const float c = (float)(g * std::cos(th) * nrm), s = (float)(-g * std::sin(th) * nrm);
            A1[(size_t)w * K1p + 2 * v] = c; A1[(size_t)w * K1p + 2 * v + 1] = s;
            A1T[(size_t)(2 * v) * WP + w] = c; A1T[(size_t)(2 * v + 1) * WP + w] = s;
        }
    for (int y = 0; y < H; ++y)
        for (int u = 0; u < H; ++u) {
            const double ph = two_pi * (double)(((long long)u * y) % H) / H;
            const float c = (float)std::cos(ph), s = (float)std::sin(ph);
            T2[(size_t)y * HP2 + u] = c; T2[(size_t)y * HP2 + H + u] = s;
            T2T[(size_t)u * HP + y] = c; T2T[(size_t)(H + u) * HP + y] = s;
        }
    // colour matrix (aphantasia to_valid_rgb): color_correlation_svd_sqrt / (colors, 1, 1), normalised by its largest column norm, transposed
    const double base[3][3] = {{0.26, 0.09, 0.02}, {0.27, 0.00, -0.05}, {0.27, -0.09, 0.03}};
    double m[3][3], mx = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = base[i][j] / (j == 0 ? (double)colors : 1.0);
    for (int j = 0; j < 3; ++j) mx = std::max(mx, std::sqrt(m[0][j] * m[0][j] + m[1][j] * m[1][j] + m[2][j] * m[2][j]));
    std::vector<float> cm(9);
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) cm[3 * c + d] = (float)(m[d][c] / mx);       // [c][d] = M^T[c][d]: rgb_d = sum_c y_c M[d][c]
    bool ok = upload(&h->scale, scale) == 0 && upload(&h->A1, A1) == 0 && upload(&h->A1T, A1T) == 0 && upload(&h->T2, T2) == 0 &&
              upload(&h->T2T, T2T) == 0 && upload(&h->cm, cm) == 0;
    auto dev = [&](float** p, size_t n) { if (ok && hipMalloc(p, n * sizeof(float)) != hipSuccess) ok = false; };
    dev(&h->bt1, (size_t)3 * HP2 * K1p); dev(&h->c1, (size_t)W * 3 * HP2); dev(&h->x, (size_t)3 * H * W); dev(&h->gy, (size_t)3 * H * W);
    dev(&h->dxT, (size_t)3 * W * HP); dev(&h->dc1t, (size_t)3 * HP2 * WP); dev(&h->dP, (size_t)3 * HP2 * K1p);
    h->ws_bytes = (size_t)32 << 20;
    dev(&h->ws, h->ws_bytes / sizeof(float));
    if (ok && hipMalloc(&h->acc, (6 + 3 * FFT_MOM_BLOCKS) * sizeof(double)) != hipSuccess) ok = false;
    // padding rows of dC1T are an A operand of the last backward GEMM: never written, so zeroed once
    if (ok && hipMemset(h->dc1t, 0, (size_t)3 * HP2 * WP * sizeof(float)) != hipSuccess) ok = false;
    if (!ok) { prx_set_error("fft drawer: device allocation failed"); prx_fft_drawer_destroy(h); return nullptr; }
    return h;
}

void prx_fft_drawer_destroy(prx_fft_drawer* h) {
    if (!h) return;
    float* bufs[] = {h->scale, h->A1, h->A1T, h->T2, h->T2T, h->cm, h->bt1, h->c1, h->x, h->gy, h->dxT, h->dc1t, h->dP, h->ws};
    for (float* b : bufs)
        if (b) (void)hipFree(b);
    if (h->acc) (void)hipFree(h->acc);
    delete h;
}

int prx_fft_drawer_freq_columns(const prx_fft_drawer* h) { return h ? h->Wf : 0; }

int prx_fft_drawer_synth(prx_fft_drawer* h, const float* params, float contrast, float* image, prx_stream_t stream) {
    PRX_REQUIRE(h && params && image, "fft drawer synth: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int W = h->W, H = h->H, HP2 = h->HP2, K1p = h->K1p;
    const size_t P = (size_t)H * W;
    h->contrast = contrast;
    hipLaunchKernelGGL(fft_pack_kernel, dim3(fgrid((size_t)3 * HP2 * K1p)), dim3(256), 0, s, params, h->scale, h->bt1, H, h->Wf, h->Wh, HP2, K1p);
    PRX_LAUNCH_CHECK();
    int rc = gemm_f32(h, h->A1, K1p, h->bt1, K1p, W, 3 * HP2, K1p, h->c1, 3 * HP2, s);
    if (rc) return rc;
    for (int c = 0; c < 3; ++c) {
        rc = gemm_f32(h, h->T2, HP2, h->c1 + (size_t)c * HP2, 3 * HP2, H, W, HP2, h->x + (size_t)c * P, W, s);
        if (rc) return rc;
    }
    const int mb = std::min(fgrid(3 * P), FFT_MOM_BLOCKS);
    hipLaunchKernelGGL(fft_moments_kernel, dim3(mb), dim3(256), 0, s, h->x, (const float*)nullptr, 3 * P, h->acc + 6);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fft_moments_final_kernel, dim3(1), dim3(256), 0, s, h->acc + 6, mb, h->acc);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fft_tail_fwd_kernel, dim3(fgrid(P)), dim3(256), 0, s, h->x, h->acc, contrast, h->cm, image, P);
    PRX_LAUNCH_CHECK();
    return 0;
}

int prx_fft_drawer_backward(prx_fft_drawer* h, const float* g_image, float* g_params, prx_stream_t stream) {
    PRX_REQUIRE(h && g_image && g_params, "fft drawer backward: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int W = h->W, H = h->H, HP = h->HP, HP2 = h->HP2, WP = h->WP, K1p = h->K1p;
    const size_t P = (size_t)H * W;
    hipLaunchKernelGGL(fft_tail_bwd_gy_kernel, dim3(fgrid(P)), dim3(256), 0, s, h->x, g_image, h->acc, h->contrast, h->cm, h->gy, P);
    PRX_LAUNCH_CHECK();
    const int mb = std::min(fgrid(3 * P), FFT_MOM_BLOCKS);
    hipLaunchKernelGGL(fft_moments_kernel, dim3(mb), dim3(256), 0, s, h->gy, h->x, 3 * P, h->acc + 6);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fft_moments_final_kernel, dim3(1), dim3(256), 0, s, h->acc + 6, mb, h->acc + 3);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(fft_tail_bwd_dx_kernel, dim3(fgrid((size_t)3 * W * HP)), dim3(256), 0, s, h->x, h->gy, h->acc, h->acc + 3, h->contrast,
                       h->dxT, H, W, HP);
    PRX_LAUNCH_CHECK();
    for (int c = 0; c < 3; ++c) {        // dC1T[(c,k)][w] = sum_h T2T[k][h] dxT_c[w][h]
        int rc = gemm_f32(h, h->T2T, HP, h->dxT + (size_t)c * W * HP, HP, 2 * H, W, HP, h->dc1t + (size_t)c * HP2 * WP, WP, s);
        if (rc) return rc;
    }
    int rc = gemm_f32(h, h->dc1t, WP, h->A1T, WP, 3 * HP2, K1p, WP, h->dP, K1p, s);      // dBt1[n][kk] = sum_w dC1T[n][w] A1T[kk][w]
    if (rc) return rc;
    hipLaunchKernelGGL(fft_unpack_kernel, dim3(fgrid((size_t)3 * H * h->Wf)), dim3(256), 0, s, h->dP, h->scale, g_params, H, h->Wf, h->Wh, HP2, K1p);
    PRX_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
