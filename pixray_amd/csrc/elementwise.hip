// Small HBM-bound helpers of the decoder / optimiser path: layout changes,
// row softmax (decoder AttnBlock), nearest-2x upsample backward, the
// (x+1)/2 + ClampWithGrad image head (vqgan.py:66-79,195), Adam + clip_z
// (pixray.py:539,1484-1487; vqgan.py:202-204).
#include "elementwise.h"
#include <algorithm>

namespace {

inline int ew_grid(size_t total, int per = 256) { return (int)std::min<size_t>((total + per - 1) / per, 8192); }

// out[c][r] = in[r][c]   (operand precision: bf16 or fp32), tiled through LDS
template <typename T>
__global__ __launch_bounds__(256) void transpose_op_kernel(const T* __restrict__ in, int ldin,
                                                           T* __restrict__ out, int ldout, int R, int C) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? in[(size_t)r * ldin + c] : (T)0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * ldout + r] = tile[tx][ty + 8 * i];
    }
}

// P = softmax(scale * S) per row; one wave per row; writes P (and optionally P^T) at operand precision
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int lds_, float scale,
                                                           T* __restrict__ P, int ldp, T* __restrict__ PT,
                                                           int ldpt, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + (size_t)row * lds_;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, s[c] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(s[c] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < cols; c += 64) {
        T p = op_cvt<T>(__expf(s[c] * scale - mx) * inv);
        P[(size_t)row * ldp + c] = p;
        if (PT) PT[(size_t)c * ldpt + row] = p;
    }
}

// dS = scale * P o (dP - rowsum(dP o P));  writes dS and dS^T at operand precision
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ P, int ldp,
                                                               const float* __restrict__ dP, int lddp, float scale,
                                                               T* __restrict__ dS, int ldds,
                                                               T* __restrict__ dST, int lddst, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64)
        dot += (float)P[(size_t)row * ldp + c] * dP[(size_t)row * lddp + c];
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) {
        float p = (float)P[(size_t)row * ldp + c];
        T v = op_cvt<T>(scale * p * (dP[(size_t)row * lddp + c] - dot));
        dS[(size_t)row * ldds + c] = v;
        if (dST) dST[(size_t)c * lddst + row] = v;
    }
}

// low[b][y][x][c] = sum over the 2x2 children of hi (NHWC fp32) -- backward of nearest-2x upsample
template <bool S16>       // S16: `hi` is a 16-bit stream in the operand format (the lean layout)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const void* __restrict__ hi, float* __restrict__ low,
                                                             bf16_t* __restrict__ low_bf16, int NB, int Hl, int Wl,
                                                             int C, int h16) {
    const int C4 = C >> 2;
    const size_t total = (size_t)NB * Hl * Wl * C4;
    const int Wh = Wl * 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        int cq = (int)(idx % C4);
        size_t pix = idx / C4;
        int x = (int)(pix % Wl);
        size_t t = pix / Wl;
        int y = (int)(t % Hl);
        int b = (int)(t / Hl);
        size_t p00 = (((size_t)b * Hl * 2 + 2 * y) * Wh + 2 * x) * C4 + cq;
        float4 a = stream_ld4<S16>(hi, p00, h16), bb = stream_ld4<S16>(hi, p00 + C4, h16), c = stream_ld4<S16>(hi, p00 + (size_t)Wh * C4, h16),
               d = stream_ld4<S16>(hi, p00 + (size_t)Wh * C4 + C4, h16);
        float4 o = make_float4((a.x + bb.x) + (c.x + d.x), (a.y + bb.y) + (c.y + d.y), (a.z + bb.z) + (c.z + d.z),
                               (a.w + bb.w) + (c.w + d.w));
        if (low) reinterpret_cast<float4*>(low)[idx] = o;
        if (low_bf16) {
            reinterpret_cast<bf16x4*>(low_bf16)[idx] = to_op16x4(o.x, o.y, o.z, o.w, h16);
        }
    }
}

// NCHW fp32 -> NHWC (fp32 and/or bf16), channels padded with zeros up to Cpad
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out_f32,
                                                           bf16_t* __restrict__ out_bf16, int NB, int C, int HW,
                                                           int Cpad, int h16) {
    const size_t total = (size_t)NB * HW * Cpad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(idx % Cpad);
        size_t t = idx / Cpad;
        int p = (int)(t % HW);
        int b = (int)(t / HW);
        float v = c < C ? in[((size_t)b * C + c) * HW + p] : 0.f;
        if (out_f32) out_f32[idx] = v;
        if (out_bf16) out_bf16[idx] = to_op16(v, h16);
    }
}

// NHWC fp32 (channel stride ldc) -> NCHW fp32
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int ldc,
                                                           float* __restrict__ out, int NB, int C, int HW) {
    const size_t total = (size_t)NB * C * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        size_t t = idx / HW;
        int c = (int)(t % C);
        int b = (int)(t / C);
        out[idx] = in[((size_t)b * HW + p) * ldc + c];
    }
}

// image head: img = clamp((x+1)/2, 0, 1)  (x is NHWC with channel stride ldc, img is NCHW)
__global__ __launch_bounds__(256) void image_head_fwd_kernel(const float* __restrict__ x, int ldc,
                                                             float* __restrict__ img, int NB, int C, int HW) {
    const size_t total = (size_t)NB * C * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        int p = (int)(idx % HW);
        size_t t = idx / HW;
        int c = (int)(t % C);
        int b = (int)(t / C);
        float v = (x[((size_t)b * HW + p) * ldc + c] + 1.f) * 0.5f;
        img[idx] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// ClampWithGrad backward (vqgan.py:76-79): g passes where g*(u - clamp(u)) >= 0, u=(x+1)/2, then *0.5.
// Output dx in NHWC (fp32, channel stride ldo, zero-padded) and bf16 copy for the dgrad conv.
__global__ __launch_bounds__(256) void image_head_bwd_kernel(const float* __restrict__ x, int ldc,
                                                             const float* __restrict__ gimg, float* __restrict__ dx,
                                                             bf16_t* __restrict__ dx_bf16, int ldo, int NB, int C,
                                                             int HW, int h16, const float* __restrict__ gscale_dev) {
    const float gscale = gscale_dev ? *gscale_dev : 1.f;
    const size_t total = (size_t)NB * HW * ldo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        int c = (int)(idx % ldo);
        size_t t = idx / ldo;
        int p = (int)(t % HW);
        int b = (int)(t / HW);
        float o = 0.f;
        if (c < C) {
            float u = (x[((size_t)b * HW + p) * ldc + c] + 1.f) * 0.5f;
            float g = gimg[((size_t)b * C + c) * HW + p];
            float uc = fminf(fmaxf(u, 0.f), 1.f);
            o = (g * (u - uc) >= 0.f) ? (0.5f * gscale) * g : 0.f;      // gscale: power-of-two gradient scale of the half mode
        }
        if (dx) dx[idx] = o;
        if (dx_bf16) dx_bf16[idx] = to_op16(o, h16);
    }
}

// The same, written as the im2col matrix of the 3x3 convolution that follows (the dgrad of conv_out: 8 gradient channels in, K = 72):
// col[p][tap * 8 + c] = dy[(y + ky - 1, x + kx - 1)][c] (zero outside the image), columns 72 .. ldk - 1 zero.  With K = 72 the
// implicit-convolution kernels run their generic per-element gather (54 us for 1.2 GFLOP at 256^2); as a row-major product over
// this matrix it is a two-stage fit-tile launch.  One thread per (pixel, tap): 8 halves = one 16-byte store.  Batch 1.
__global__ __launch_bounds__(256) void image_head_bwd_im2col_kernel(const float* __restrict__ x, int ldc, const float* __restrict__ gimg,
                                                                    bf16_t* __restrict__ col, int ldk, int C, int H, int W, int h16,
                                                                    const float* __restrict__ gscale_dev) {
    const float gscale = gscale_dev ? *gscale_dev : 1.f;
    const int slots = ldk / 8;                      // 16-byte slots per row: 9 taps, then zero padding
    const size_t HW = (size_t)H * W, total = HW * slots;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(idx % slots);
        const size_t p = idx / slots;
        const int py = (int)(p / W), px = (int)(p - (size_t)py * W);
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = to_op16(0.f, h16);
        if (tap < 9) {
            const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const size_t q = (size_t)yy * W + xx;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c < C) {
                        const float u = (x[q * ldc + c] + 1.f) * 0.5f;
                        const float g = gimg[(size_t)c * HW + q];
                        const float uc = fminf(fmaxf(u, 0.f), 1.f);
                        o[c] = to_op16((g * (u - uc) >= 0.f) ? (0.5f * gscale) * g : 0.f, h16);
                    }
                }
            }
        }
        reinterpret_cast<bf16x8*>(col)[idx] = o;
    }
}

// Adam (torch.optim.Adam semantics, amsgrad off, weight_decay 0) fused with clip_z.
// z is NCHW [1, C, HW]; zmin/zmax per channel (may be null -> no clamp).
__global__ __launch_bounds__(256) void adam_clamp_kernel(float* __restrict__ z, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ g,
                                                         const float* __restrict__ zmin, const float* __restrict__ zmax,
                                                         int hw, size_t n, float lr, float b1, float b2, float eps,
                                                         float bc1, float bc2_sqrt, const float* __restrict__ hyper) {
    // hyper (optional, device): {lr / bias_correction1, sqrt(bias_correction2)} -- lets a captured hipGraph be
    // replayed with the step-dependent scalars updated in place
    if (hyper) { lr = hyper[0]; bc1 = 1.f; bc2_sqrt = hyper[1]; }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        float mi = b1 * m[i] + (1.f - b1) * gi;          // exp_avg.lerp_(grad, 1-beta1)
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float denom = sqrtf(vi) / bc2_sqrt + eps;
        float zi = z[i] - (lr / bc1) * (mi / denom);
        if (zmin) {
            int c = (int)(i / hw);
            zi = fminf(fmaxf(zi, zmin[c]), zmax[c]);
        }
        z[i] = zi;
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                          size_t n, int h16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = to_op16(in[i], h16);
}

// ---- power-of-two gradient scale of the half (PRX_PREC_F16) mode: scale2 = {S, 1/S} with S * max|g| in [2^(T-1), 2^T) --------
__global__ __launch_bounds__(256) void amax_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(g[i]));
    m = wave_max(m);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}
__global__ __launch_bounds__(256) void grad_scale_final_kernel(const float* __restrict__ part, int nparts, int target_log2,
                                                               float* __restrict__ scale2) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) m = fmaxf(m, part[i]);
    m = wave_max(m);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        float S = 1.f;
        if (m > 0.f && m < INFINITY) {            // NaN / inf / all-zero gradients: leave them unscaled
            int e;
            (void)frexpf(m, &e);                   // m = f * 2^e, f in [0.5, 1)
            int k = target_log2 - e;
            k = k < -40 ? -40 : (k > 60 ? 60 : k);
            S = ldexpf(1.f, k);
        }
        scale2[0] = S;
        scale2[1] = 1.f / S;                       // exact: a power of two
    }
}

// x *= *scale (a device scalar; the half mode's power-of-two gradient scale, so the product is exact)
// out16 (optional): the scaled values in the 16-bit operand format as well -- the next GEMM's A operand (a 16-bit A is what the
// fit tiles take; an fp32 A goes to the register-staged kernels)
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, size_t n, const float* __restrict__ scale, bf16_t* __restrict__ out16, int h16) {
    const float sc = *scale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i] * sc;
        x[i] = v;
        if (out16) out16[i] = to_op16(v, h16);
    }
}

__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

}  // namespace

int prx_transpose_op(const void* in, int ldin, void* out, int ldout, int R, int C, int f32, hipStream_t s) {
    dim3 grid(ceil_div(C, 32), ceil_div(R, 32));
    if (f32) hipLaunchKernelGGL(transpose_op_kernel<float>, grid, dim3(256), 0, s, (const float*)in, ldin, (float*)out, ldout, R, C);
    else     hipLaunchKernelGGL(transpose_op_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, ldin, (bf16_t*)out, ldout, R, C);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_softmax_rows(const float* S, int lds_, float scale, void* P, int ldp, void* PT, int ldpt, int rows,
                     int cols, int prec, hipStream_t s) {
    PRX_OP_DISPATCH(prec_is_f32(prec), prec_is_h16(prec), T,
                    hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, S, lds_, scale, (T*)P, ldp,
                                       (T*)PT, ldpt, rows, cols));
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_softmax_rows_bwd(const void* P, int ldp, const float* dP, int lddp, float scale, void* dS, int ldds,
                         void* dST, int lddst, int rows, int cols, int prec, hipStream_t s) {
    PRX_OP_DISPATCH(prec_is_f32(prec), prec_is_h16(prec), T,
                    hipLaunchKernelGGL(softmax_rows_bwd_kernel<T>, dim3(ceil_div(rows, 4)), dim3(256), 0, s, (const T*)P, ldp, dP, lddp,
                                       scale, (T*)dS, ldds, (T*)dST, lddst, rows, cols));
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_upsample2x_bwd(const void* hi, float* low, bf16_t* low_bf16, int NB, int Hl, int Wl, int C, hipStream_t s, int h16, int s16) {
    PRX_REQUIRE(C % 4 == 0 && (low || low_bf16), "upsample2x_bwd: C %% 4 != 0, or no output");
    if (s16) hipLaunchKernelGGL(upsample2x_bwd_kernel<true>, dim3(ew_grid((size_t)NB * Hl * Wl * C / 4)), dim3(256), 0, s, hi, low, low_bf16,
                                NB, Hl, Wl, C, h16);
    else hipLaunchKernelGGL(upsample2x_bwd_kernel<false>, dim3(ew_grid((size_t)NB * Hl * Wl * C / 4)), dim3(256), 0, s, hi, low, low_bf16,
                            NB, Hl, Wl, C, h16);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_nchw_to_nhwc(const float* in, float* out_f32, bf16_t* out_bf16, int NB, int C, int HW, int Cpad,
                     hipStream_t s, int h16) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((size_t)NB * HW * Cpad)), dim3(256), 0, s, in, out_f32,
                       out_bf16, NB, C, HW, Cpad, h16);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_nhwc_to_nchw(const float* in, int ldc, float* out, int NB, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((size_t)NB * HW * C)), dim3(256), 0, s, in, ldc, out, NB, C,
                       HW);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_image_head_fwd(const float* x, int ldc, float* img, int NB, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(image_head_fwd_kernel, dim3(ew_grid((size_t)NB * HW * C)), dim3(256), 0, s, x, ldc, img, NB, C,
                       HW);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_grad_scale(const float* g, size_t n, float* part, int nparts, int target_log2, float* scale2, hipStream_t s) {
    PRX_REQUIRE(nparts >= 1 && nparts <= 1024, "grad_scale: 1..1024 partials");
    const int blocks = (int)std::min<size_t>((n + 1023) / 1024 + 1, (size_t)nparts);
    hipLaunchKernelGGL(amax_partial_kernel, dim3(blocks), dim3(256), 0, s, g, n, part);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(grad_scale_final_kernel, dim3(1), dim3(256), 0, s, part, blocks, target_log2, scale2);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_grad_scale_multi(const float* const* gs, const size_t* ns, int count, float* part, int nparts_each, int target_log2,
                         float* scale2, hipStream_t s) {
    PRX_REQUIRE(count >= 1 && nparts_each >= 1 && (long long)count * nparts_each <= 4096, "grad_scale_multi: bad partial layout");
    int used = 0;
    for (int i = 0; i < count; ++i) {
        if (!gs[i] || ns[i] == 0) continue;
        const int blocks = (int)std::min<size_t>((ns[i] + 1023) / 1024 + 1, (size_t)nparts_each);
        hipLaunchKernelGGL(amax_partial_kernel, dim3(blocks), dim3(256), 0, s, gs[i], ns[i], part + used);
        PRX_LAUNCH_CHECK();
        used += blocks;
    }
    PRX_REQUIRE(used > 0, "grad_scale_multi: no gradient tensor");
    hipLaunchKernelGGL(grad_scale_final_kernel, dim3(1), dim3(256), 0, s, part, used, target_log2, scale2);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_scale_dev(float* x, size_t n, const float* scale, hipStream_t s, bf16_t* out16, int h16) {
    hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, s, x, n, scale, out16, h16);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_image_head_bwd(const float* x, int ldc, const float* gimg, float* dx, bf16_t* dx_bf16, int ldo, int NB, int C,
                       int HW, hipStream_t s, int h16, const float* gscale) {
    hipLaunchKernelGGL(image_head_bwd_kernel, dim3(ew_grid((size_t)NB * HW * ldo)), dim3(256), 0, s, x, ldc, gimg, dx,
                       dx_bf16, ldo, NB, C, HW, h16, gscale);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_image_head_bwd_im2col(const float* x, int ldc, const float* gimg, bf16_t* col, int ldk, int C, int H, int W, hipStream_t s, int h16,
                              const float* gscale) {
    PRX_REQUIRE(C <= 8 && ldk % 8 == 0 && ldk >= 72, "image_head_bwd_im2col: at most 8 channels, row pitch a multiple of 8 >= 72");
    hipLaunchKernelGGL(image_head_bwd_im2col_kernel, dim3(ew_grid((size_t)H * W * (ldk / 8))), dim3(256), 0, s, x, ldc, gimg, col, ldk, C, H, W,
                       h16, gscale);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_adam_clamp(float* z, float* m, float* v, const float* g, const float* zmin, const float* zmax, int hw,
                   size_t n, float lr, float b1, float b2, float eps, int step, hipStream_t s) {
    PRX_REQUIRE(step >= 1, "adam: step must be >= 1");
    const double bc1 = 1.0 - pow((double)b1, step);
    const double bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_clamp_kernel, dim3(ew_grid(n)), dim3(256), 0, s, z, m, v, g, zmin, zmax, hw, n, lr, b1, b2,
                       eps, (float)bc1, (float)sqrt(bc2), (const float*)nullptr);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_adam_clamp_dev(float* z, float* m, float* v, const float* g, const float* zmin, const float* zmax, int hw,
                       size_t n, const float* hyper, float b1, float b2, float eps, hipStream_t s) {
    PRX_REQUIRE(hyper != nullptr, "adam: null hyper buffer");
    hipLaunchKernelGGL(adam_clamp_kernel, dim3(ew_grid(n)), dim3(256), 0, s, z, m, v, g, zmin, zmax, hw, n, 0.f, b1, b2,
                       eps, 1.f, 1.f, hyper);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s, int h16) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, s, in, out, n, h16);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, s, a, b, out, n);
    PRX_LAUNCH_CHECK();
    return 0;
}
