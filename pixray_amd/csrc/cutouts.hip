// MakeCutouts (pixray.py:400-511) as coalesced HBM gather kernels, forward and backward,
// plus the batch-global min/max renorm + CLIP mean/std + patchify that
// CLIP_Base.preprocess (slip.py:21-42,52-60) feeds to the visual tower.
//
//   pool      : (AdaptiveAvgPool2d + AdaptiveMaxPool2d)/2 of the synthesised image, computed ONCE
//               (the reference recomputes it per cutout, pixray.py:461-463)
//   stage A   : per-cutout bilinear warp of the pooled image (zoom: perspective with
//               reflection/border padding; wide: affine with gray fill)
//   stage B   : per-cutout bilinear warp of stage A (zoom: resized crop, zeros padding; wide:
//               perspective with gray fill) fused with ColorJitter (HSV saturation/hue) and the
//               additive noise.
// Each geometric stage is described by a 3x3 matrix taking a destination pixel (x, y, 1) to the
// source sampling position in F.grid_sample's unnormalised pixel coordinates; the host
// (pixray_amd/cutouts.py) folds kornia's normalisation conventions into it.
// Backward is in GATHER form: every source pixel sums its own contributions in a fixed order (no atomics, bit-
// reproducible -- the reference's grid_sampler_2d_backward is the non-determinism pixray.py:29 complains about); the
// ColorJitter Jacobian is obtained with forward-mode duals.
#include "cutouts.h"
#include <algorithm>
#include <stdlib.h>
#include <stdio.h>

namespace {

constexpr int DESC_WORDS = 36;
// descriptor word offsets (all stored as fp64; the two 3x3 matrices map NORMALISED destination coordinates to
// NORMALISED source coordinates exactly as kornia builds them, so sampling positions round like the oracle's)
enum { D_M1 = 0, D_M2 = 9, D_MODE1 = 18, D_MODE2 = 19, D_FILL = 20, D_JIT = 21, D_SAT = 22, D_HUE = 23,
       D_SATFIRST = 24, D_NOISE = 25, D_GRID1 = 26, D_GRID2 = 27,
       D_WOX = 28, D_WOY = 29, D_WW = 30, D_WH = 31,     // stage-B source window inside the stage-A image (x, y, width, height)
       D_SEED = 32 };   // != 0 and no noise tensor given: the additive N(0,1) draws of this cutout come from Philox4x32-10 keyed by it (words 33-35 spare)
// grid flavour of a stage = which kornia 0.6.2 call built its sampling grid AND the align_corners flag that call passed to
// F.grid_sample (the convention is part of the descriptor, not an assumption of the kernel):
//   GRID_MESH       warp_perspective(align_corners=False): create_meshgrid + transform_points, sampled with (g+1)*W/2 - 0.5
//                   (RandomPerspective's default flag, pixray.py:333-334,363)
//   GRID_AFFINE     warp_affine(align_corners=False): F.affine_grid's pixel-centre base grid (RandomAffine's default, pixray.py:349)
//   GRID_AFFINE_AC  warp_affine(align_corners=True): F.affine_grid's corner-aligned base grid linspace(-1,1), sampled with
//                   (g+1)/2*(W-1) -- what crop_by_transform_mat passes for RandomResizedCrop / CenterCrop, whose flag
//                   defaults to True (pixray.py:415,433 pass none)
//   GRID_MESH_AC    warp_perspective(align_corners=True), the function default: the cached-transform path (pixray.py:482-485)
enum { GRID_MESH = 0, GRID_AFFINE = 1, GRID_AFFINE_AC = 2, GRID_MESH_AC = 3 };
enum { MODE_IDENT = 0, MODE_ZEROS = 1, MODE_BORDER = 2, MODE_REFLECT = 3, MODE_FILL = 4,
       MODE_REFLECT_AC = 5 };   // reflection as F.grid_sample does it with align_corners=True (the cached-transform path)

inline int ew_grid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 16384); }

// ------------------------------------------------------------------ pooling
__device__ __forceinline__ int win_start(int i, int in, int out) { return (int)(((long long)i * in) / out); }
__device__ __forceinline__ int win_end(int i, int in, int out) { return (int)((((long long)(i + 1)) * in + out - 1) / out); }

// `mask` (optional, uint8 [C,S,S]): spot prompts (pixray.py:453-466) zero the pooled cutout where the mask is set
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ img, float* __restrict__ pooled,
                                                       int* __restrict__ argmax, const unsigned char* __restrict__ mask,
                                                       int C, int H, int W, int S) {
    const int total = C * S * S;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int x = idx % S, y = (idx / S) % S, c = idx / (S * S);
        const int y0 = win_start(y, H, S), y1 = win_end(y, H, S);
        const int x0 = win_start(x, W, S), x1 = win_end(x, W, S);
        float sum = 0.f, mx = -INFINITY;
        int am = y0 * W + x0;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                float v = img[((size_t)c * H + yy) * W + xx];
                sum += v;
                if (v > mx || v != v) { mx = v; am = yy * W + xx; }   // first max wins (torch CPU scan order)
            }
        pooled[idx] = (mask && mask[idx]) ? 0.f : 0.5f * (sum / (float)((y1 - y0) * (x1 - x0)) + mx);
        argmax[idx] = am;
    }
}

// Backward of the pooling in GATHER form (one thread per image pixel, fixed summation order -> bit-reproducible; the
// scatter form needed float atomics because adaptive windows overlap whenever H is not a multiple of S): the pooled cells
// whose window [win_start, win_end) contains row yy are a run of at most a few consecutive y around yy*S/H.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ g, const int* __restrict__ argmax,
                                                       const unsigned char* __restrict__ mask, float* __restrict__ gimg, int C, int H,
                                                       int W, int S) {
    const int total = C * H * W;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int xx = idx % W, yy = (idx / W) % H, c = idx / (H * W);
        int ya = (int)(((long long)yy * S) / H), xa = (int)(((long long)xx * S) / W);
        while (ya > 0 && win_end(ya - 1, H, S) > yy) --ya;          // first cell whose window reaches down to yy
        while (xa > 0 && win_end(xa - 1, W, S) > xx) --xa;
        float acc = 0.f;
        for (int y = ya; y < S && win_start(y, H, S) <= yy; ++y) {
            const int y0 = win_start(y, H, S), y1 = win_end(y, H, S);
            if (yy >= y1) continue;
            for (int x = xa; x < S && win_start(x, W, S) <= xx; ++x) {
                const int x0 = win_start(x, W, S), x1 = win_end(x, W, S);
                if (xx >= x1) continue;
                const int cell = (c * S + y) * S + x;
                if (mask && mask[cell]) continue;                   // masked pooled pixels are constants
                const float gv = 0.5f * g[cell];
                acc += gv / (float)((y1 - y0) * (x1 - x0));
                if (argmax[cell] == yy * W + xx) acc += gv;
            }
        }
        gimg[idx] = acc;
    }
}

// ------------------------------------------------------------------ bilinear sampling
struct Taps {
    int x0, y0;          // north-west tap
    float wx, wy;        // weight of the east / south taps
    bool vx0, vx1, vy0, vy1;
};

__device__ __forceinline__ float reflect_coord(float in, float size) {
    // F.grid_sample reflection, align_corners=False: reflect about [-0.5, size-0.5]
    const float mn = -0.5f, span = size;
    in = fabsf(in - mn);
    float extra = fmodf(in, span);
    int flips = (int)floorf(in / span);
    return (flips & 1) ? (span - extra + mn) : (extra + mn);
}

__device__ __forceinline__ float reflect_coord_ac(float in, float size) {
    // F.grid_sample reflection, align_corners=True: reflect about the pixel centres [0, size-1]
    const float span = size - 1.f;
    if (span <= 0.f) return 0.f;
    in = fabsf(in);
    float extra = fmodf(in, span);
    int flips = (int)floorf(in / span);
    return (flips & 1) ? (span - extra) : extra;
}

__device__ __forceinline__ Taps make_taps(float u, float v, int W, int H, int mode) {
    if (mode == MODE_BORDER) {
        u = fminf(fmaxf(u, 0.f), (float)(W - 1));
        v = fminf(fmaxf(v, 0.f), (float)(H - 1));
    } else if (mode == MODE_REFLECT) {
        u = fminf(fmaxf(reflect_coord(u, (float)W), 0.f), (float)(W - 1));
        v = fminf(fmaxf(reflect_coord(v, (float)H), 0.f), (float)(H - 1));
    } else if (mode == MODE_REFLECT_AC) {
        u = fminf(fmaxf(reflect_coord_ac(u, (float)W), 0.f), (float)(W - 1));
        v = fminf(fmaxf(reflect_coord_ac(v, (float)H), 0.f), (float)(H - 1));
    }
    Taps t;
    float fx = floorf(u), fy = floorf(v);
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.wx = u - fx; t.wy = v - fy;
    t.vx0 = t.x0 >= 0 && t.x0 < W;
    t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    t.vy0 = t.y0 >= 0 && t.y0 < H;
    t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    return t;
}

// the tap set of an exact copy of pixel (x, y): weight 1 on the north-west tap, the other three masked off
__device__ __forceinline__ Taps ident_taps(int x, int y) {
    Taps t;
    t.x0 = x; t.y0 = y; t.wx = 0.f; t.wy = 0.f;
    t.vx0 = true; t.vy0 = true; t.vx1 = false; t.vy1 = false;
    return t;
}

// torch.linspace(-1, 1, n) in fp32 (ATen's symmetric two-sided formula)
__device__ __forceinline__ float linspace_pm1(int i, int n) {
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? (-1.f + step * (float)i) : (1.f - step * (float)(n - 1 - i));
}

// Source sampling position (F.grid_sample pixel coordinates) of destination pixel (x, y).
// GRID_MESH*: kornia create_meshgrid + transform_points (fp64) -> fp32 grid (warp_perspective);
// GRID_AFFINE*: F.affine_grid's fp32 base grid times the fp32 theta (warp_affine).  The fp32 grid value is
// then unnormalised the way ATen does for the stage's align_corners flag.
__device__ __forceinline__ void project(const double* m, int gtype, int x, int y, int Wd, int Hd, int Ws, int Hs,
                                        float& u, float& v) {
    float gx, gy;
    if (gtype == GRID_MESH || gtype == GRID_MESH_AC) {
        const double xn = ((double)x / (double)(Wd - 1) - 0.5) * 2.0;
        const double yn = ((double)y / (double)(Hd - 1) - 0.5) * 2.0;
        const double X = m[0] * xn + m[1] * yn + m[2];
        const double Y = m[3] * xn + m[4] * yn + m[5];
        const double Z = m[6] * xn + m[7] * yn + m[8];
        const double sc = (fabs(Z) > 1e-8) ? 1.0 / Z : 1.0;
        gx = (float)(X * sc); gy = (float)(Y * sc);
    } else {
        float xb = linspace_pm1(x, Wd), yb = linspace_pm1(y, Hd);      // affine_grid(align_corners=True): corner-aligned
        if (gtype == GRID_AFFINE) {                                      // align_corners=False: pixel centres
            xb = (xb * (float)(Wd - 1)) / (float)Wd;
            yb = (yb * (float)(Hd - 1)) / (float)Hd;
        }
        gx = (float)((double)xb * m[0] + (double)yb * m[1] + m[2]);
        gy = (float)((double)xb * m[3] + (double)yb * m[4] + m[5]);
    }
    if (gtype >= GRID_AFFINE_AC) {          // ATen grid_sampler_unnormalize, align_corners=True
        u = ((gx + 1.f) / 2.f) * (float)(Ws - 1);
        v = ((gy + 1.f) / 2.f) * (float)(Hs - 1);
    } else {                                // align_corners=False
        u = (gx + 1.f) * ((float)Ws * 0.5f) - 0.5f;
        v = (gy + 1.f) * ((float)Hs * 0.5f) - 0.5f;
    }
}

// sample one channel plane; returns value and the coverage (sum of in-bounds weights)
__device__ __forceinline__ float sample_plane(const float* __restrict__ p, int W, const Taps& t) {
    float v00 = (t.vx0 && t.vy0) ? p[t.y0 * W + t.x0] : 0.f;
    float v01 = (t.vx1 && t.vy0) ? p[t.y0 * W + t.x0 + 1] : 0.f;
    float v10 = (t.vx0 && t.vy1) ? p[(t.y0 + 1) * W + t.x0] : 0.f;
    float v11 = (t.vx1 && t.vy1) ? p[(t.y0 + 1) * W + t.x0 + 1] : 0.f;
    // ATen's operation order (GridSamplerKernel.cpp, bilinear): the four weights first (nw = s*e, ne = s*w, sw = n*e,
    // se = n*w with e = 1-w, s = 1-n), then nw_val*nw + ne_val*ne + sw_val*sw + se_val*se left to right -- so that flat
    // regions (the exact 0/1 plateaus ClampWithGrad leaves) round to the same fp32 value as the oracle's and ties
    // between channels, which the HSV jitter's Jacobian is discontinuous at, break the same way
    const float e = 1.f - t.wx, s_ = 1.f - t.wy;
    return v00 * (s_ * e) + v01 * (s_ * t.wx) + v10 * (t.wy * e) + v11 * (t.wy * t.wx);
}
__device__ __forceinline__ float coverage(const Taps& t) {
    float c = 0.f;
    if (t.vx0 && t.vy0) c += (1.f - t.wx) * (1.f - t.wy);
    if (t.vx1 && t.vy0) c += t.wx * (1.f - t.wy);
    if (t.vx0 && t.vy1) c += (1.f - t.wx) * t.wy;
    if (t.vx1 && t.vy1) c += t.wx * t.wy;
    return c;
}
// gray fill of the uncovered part of the bilinear footprint (kornia's `padding_mode="fill"`: zeros padding + (1 - warp(ones)) * fill).
// Branch-free on purpose: hipcc miscompiled the `mode == MODE_FILL ? ... : 0` form of this inside warp_b_fwd_kernel (the
// divergent select clobbered live tap-pointer registers; MODE_ZEROS was fine, MODE_FILL returned garbage).
__device__ __forceinline__ float fill_term(const Taps& t, int mode, const double* d) {
    const float fillv = (mode == MODE_FILL) ? (float)d[D_FILL] : 0.f;
    return (1.f - coverage(t)) * fillv;
}
// ------------------------------------------------------------------ ColorJitter (HSV) with duals
template <int ND>
struct Dual {
    float v;
    float d[ND > 0 ? ND : 1];
};
template <int ND> __device__ __forceinline__ Dual<ND> cst(float c) {
    Dual<ND> r; r.v = c;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = 0.f;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator+(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator-(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator*(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> operator/(const Dual<ND>& a, const Dual<ND>& b) {
    Dual<ND> r; r.v = a.v / b.v;   // true (correctly rounded) division: mirrors torch's op sequence
    const float ib = 1.f / b.v;    // the derivative parts need no bit-parity with anything: one reciprocal, ND multiplies
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> addc(const Dual<ND>& a, float c) { Dual<ND> r = a; r.v += c; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> divc(const Dual<ND>& a, float c) {
    Dual<ND> r; r.v = a.v / c;
    const float ic = 1.f / c;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * ic;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> mulc(const Dual<ND>& a, float c) {
    Dual<ND> r; r.v = a.v * c;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * c;
    return r;
}

constexpr float TWO_PI_F = 6.283185307179586f;

template <int ND>
__device__ __forceinline__ void rgb_to_hsv_d(const Dual<ND> (&rgb)[3], Dual<ND>& H, Dual<ND>& S, Dual<ND>& V) {
    // kornia.color.rgb_to_hsv (eps 1e-8); first max / first min on ties (torch CPU reduction order)
    int imax = 0; Dual<ND> mx = rgb[0];
    if (rgb[1].v > mx.v) { mx = rgb[1]; imax = 1; }
    if (rgb[2].v > mx.v) { mx = rgb[2]; imax = 2; }
    Dual<ND> mn = rgb[0];
    if (rgb[1].v < mn.v) mn = rgb[1];
    if (rgb[2].v < mn.v) mn = rgb[2];
    Dual<ND> delta = mx - mn;
    V = mx;
    S = delta / addc(mx, 1e-8f);
    Dual<ND> dc = (delta.v == 0.f) ? cst<ND>(1.f) : delta;
    Dual<ND> rc = mx - rgb[0], gc = mx - rgb[1], bc = mx - rgb[2];
    Dual<ND> h;
    if (imax == 0) h = (bc - gc) / dc;
    else if (imax == 1) h = ((rc - bc) + mulc(dc, 2.f)) / dc;
    else h = ((gc - rc) + mulc(dc, 4.f)) / dc;
    h = divc(h, 6.f);
    h = addc(h, -floorf(h.v));          // python-style % 1.0
    H = mulc(h, TWO_PI_F);
}

template <int ND>
__device__ __forceinline__ void hsv_to_rgb_d(const Dual<ND>& H, const Dual<ND>& S, const Dual<ND>& V, Dual<ND> (&rgb)[3]) {
    Dual<ND> h6 = mulc(divc(H, TWO_PI_F), 6.f);
    float fl = floorf(h6.v);
    int hi = (int)fl % 6; if (hi < 0) hi += 6;
    // f = ((h*6) % 6) - hi  with python-style %
    float m6 = h6.v - 6.f * floorf(h6.v / 6.f);
    Dual<ND> f = h6; f.v = m6 - (float)hi;
    Dual<ND> one = cst<ND>(1.f);
    Dual<ND> p = V * (one - S);
    Dual<ND> q = V * (one - f * S);
    Dual<ND> t = V * (one - (one - f) * S);
    switch (hi) {
        case 0: rgb[0] = V; rgb[1] = t; rgb[2] = p; break;
        case 1: rgb[0] = q; rgb[1] = V; rgb[2] = p; break;
        case 2: rgb[0] = p; rgb[1] = V; rgb[2] = t; break;
        case 3: rgb[0] = p; rgb[1] = q; rgb[2] = V; break;
        case 4: rgb[0] = t; rgb[1] = p; rgb[2] = V; break;
        default: rgb[0] = V; rgb[1] = p; rgb[2] = q; break;
    }
}

template <int ND>
__device__ __forceinline__ void adjust_sat_d(Dual<ND> (&rgb)[3], float factor) {
    Dual<ND> H, S, V;
    rgb_to_hsv_d<ND>(rgb, H, S, V);
    S = mulc(S, factor);
    if (S.v < 0.f) S = cst<ND>(0.f); else if (S.v > 1.f) S = cst<ND>(1.f);
    hsv_to_rgb_d<ND>(H, S, V, rgb);
}
template <int ND>
__device__ __forceinline__ void adjust_hue_d(Dual<ND> (&rgb)[3], float shift_rad) {
    Dual<ND> H, S, V;
    rgb_to_hsv_d<ND>(rgb, H, S, V);
    float hv = H.v + shift_rad;
    H.v = fmodf(hv, TWO_PI_F);           // torch.fmod (C semantics)
    hsv_to_rgb_d<ND>(H, S, V, rgb);
}
template <int ND>
__device__ __forceinline__ void jitter_d(Dual<ND> (&rgb)[3], float sat, float hue_rad, bool sat_first) {
    if (sat_first) { adjust_sat_d<ND>(rgb, sat); adjust_hue_d<ND>(rgb, hue_rad); }
    else { adjust_hue_d<ND>(rgb, hue_rad); adjust_sat_d<ND>(rgb, sat); }
}

// ------------------------------------------------------------------ warp stages
// Stage A: out[n][c][y][x] (Ha x Wa) from the shared source src[c][Hs][Ws].  On a square canvas all three sizes are S x S;
// on a W != H canvas (pixray.py:468-472) the source is the pooled image rescaled to the canvas aspect and stage A keeps
// that size.
__global__ __launch_bounds__(256) void warp_a_fwd_kernel(const float* __restrict__ src, int Hs, int Ws,
                                                         const double* __restrict__ desc, float* __restrict__ out,
                                                         int n_cut, int Ha, int Wa) {
    const size_t plane = (size_t)Ha * Wa;
    const size_t total = (size_t)n_cut * plane;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wa), y = (int)((idx / Wa) % Ha), n = (int)(idx / plane);
        const double* d = desc + (size_t)n * DESC_WORDS;
        const int mode = (int)d[D_MODE1];
        float* o = out + ((size_t)n * 3) * plane + (size_t)y * Wa + x;
        if (mode == MODE_IDENT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[(size_t)c * plane] = src[(size_t)c * Hs * Ws + (size_t)y * Ws + x];
            continue;
        }
        float u, v;
        project(d + D_M1, (int)d[D_GRID1], x, y, Wa, Ha, Ws, Hs, u, v);
        Taps t = make_taps(u, v, Ws, Hs, mode);
        const float fillc = fill_term(t, mode, d);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(size_t)c * plane] = sample_plane(src + (size_t)c * Hs * Ws, Ws, t) + fillc;
    }
}

// ---- backward in GATHER form ----------------------------------------------------------------------------------------
// The reference differentiates its warps with grid_sampler_2d_backward: float atomics, run-to-run different bits
// (pixray.py:29 documents the non-determinism).  Here every SOURCE pixel collects its own contributions in a fixed order:
// bit-reproducible, no atomics, no memsets.  For source pixel s the destination pixels whose bilinear footprint touches
// it lie in the pre-image of a 2x2 source square -- under zero / fill padding one square, under border padding a half
// strip for border pixels, under reflection one square per mirror image.  Each pre-image is a quadrilateral of the
// destination plane (the stage map is a homography); its bounding box is enumerated and every candidate is judged by the
// forward's OWN raw sampling coordinate (the uv map a destination-parallel pre-pass writes with the same fp64 -> fp32
// steps), so taps and weights are bit-for-bit the forward's and the gradient is its exact adjoint, only summed in a
// different order than ATen does.
// Uniqueness: the raw-coordinate rectangles of one source pixel are disjoint and a candidate is counted in the rectangle
// that contains its raw coordinate; completeness: acceptance is by the exact taps, the rectangles only bound the search.
struct StageMap {
    float Pi[9];     // raw source coordinate (u, v, 1) -> destination pixel, approximate (fp32): bounding boxes only
    float ulo, uhi, vlo, vhi;   // raw-coordinate range of the destination image (+- 1)
};

// pixel-space matrix of a stage: A_src(flavour) * M * A_dst(flavour), all fp64
__device__ void stage_matrix(const double* m, int gtype, int Wd, int Hd, int Ws, int Hs, double* P) {
    // destination pixel -> normalised destination coordinate
    double ax, bx, ay, by;
    if (gtype == GRID_AFFINE) { ax = 2.0 / Wd; bx = 1.0 / Wd - 1.0; ay = 2.0 / Hd; by = 1.0 / Hd - 1.0; }
    else { ax = 2.0 / (Wd - 1); bx = -1.0; ay = 2.0 / (Hd - 1); by = -1.0; }
    double T[9];
    for (int r = 0; r < 3; ++r) {
        T[r * 3 + 0] = m[r * 3 + 0] * ax;
        T[r * 3 + 1] = m[r * 3 + 1] * ay;
        T[r * 3 + 2] = m[r * 3 + 0] * bx + m[r * 3 + 1] * by + m[r * 3 + 2];
    }
    if (gtype == GRID_AFFINE || gtype == GRID_AFFINE_AC) { T[6] = 0.0; T[7] = 0.0; T[8] = 1.0; }    // warp_affine drops the last row
    // normalised source coordinate -> raw pixel coordinate
    double sx, ox, sy, oy;
    if (gtype >= GRID_AFFINE_AC) { sx = 0.5 * (Ws - 1); ox = sx; sy = 0.5 * (Hs - 1); oy = sy; }
    else { sx = 0.5 * Ws; ox = sx - 0.5; sy = 0.5 * Hs; oy = sy - 0.5; }
    for (int c = 0; c < 3; ++c) {
        P[c] = sx * T[c] + ox * T[6 + c];
        P[3 + c] = sy * T[3 + c] + oy * T[6 + c];
        P[6 + c] = T[6 + c];
    }
}
__device__ void invert3(const double* a, double* o) {
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    const double id = det != 0.0 ? 1.0 / det : 0.0;
    o[0] = c0 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}
// thread 0 of the block builds the stage map of its cutout in LDS
__device__ void build_stage_map(StageMap& sm, const double* m, int gtype, int Wd, int Hd, int Ws, int Hs) {
    double P[9], Pi[9];
    stage_matrix(m, gtype, Wd, Hd, Ws, Hs, P);
    invert3(P, Pi);
    float ulo = INFINITY, uhi = -INFINITY, vlo = INFINITY, vhi = -INFINITY;
    for (int k = 0; k < 4; ++k) {
        const double x = (k & 1) ? (double)(Wd - 1) : 0.0, y = (k & 2) ? (double)(Hd - 1) : 0.0;
        const double z = P[6] * x + P[7] * y + P[8];
        const double u = (P[0] * x + P[1] * y + P[2]) / z, v = (P[3] * x + P[4] * y + P[5]) / z;
        ulo = fminf(ulo, (float)u); uhi = fmaxf(uhi, (float)u); vlo = fminf(vlo, (float)v); vhi = fmaxf(vhi, (float)v);
    }
    // normalise so that the projective denominator is O(1) and positive inside the image
    const double zi = Pi[6] * (0.5 * (ulo + uhi)) + Pi[7] * (0.5 * (vlo + vhi)) + Pi[8];
    const double si = zi != 0.0 ? 1.0 / zi : 1.0;
    for (int k = 0; k < 9; ++k) sm.Pi[k] = (float)(Pi[k] * si);
    sm.ulo = ulo - 1.5f; sm.uhi = uhi + 1.5f; sm.vlo = vlo - 1.5f; sm.vhi = vhi + 1.5f;
}

// Raw-coordinate intervals [a, b) whose padded coordinate can produce a tap on source index s (size W), clipped to
// [lo, hi].  Returns the count (<= MAXI); intervals come out sorted, merged and pairwise disjoint.
constexpr int MAXI = 3;
__device__ __forceinline__ int tap_intervals(int mode, int s, int W, float lo, float hi, float* ia, float* ib) {
    const float eps = 2e-3f;       // slack against the fp32 rounding of the forward's own coordinate arithmetic
    const float fs = (float)s, fW = (float)W;
    // the direct image: every mode has it (the upright copy of the source inside its own domain)
    float a = fs - 1.f - eps, b = fs + 1.f + eps;
    if (mode == MODE_BORDER) {                      // clamping folds everything beyond the edge onto the edge pixel
        if (s == 0) a = -INFINITY;
        if (s == W - 1) b = INFINITY;
    }
    a = fmaxf(a, lo); b = fminf(b, hi);
    int n = 0;
    if (b > a) { ia[0] = a; ib[0] = b; n = 1; }
    if (mode != MODE_REFLECT && mode != MODE_REFLECT_AC) return n;
    // first-order mirror images (reflection about the low and the high edge); a raw range that reaches past them
    // (|excursion| > one image size: never with pixray's distortion scales) falls back to the whole range
    float m0a, m0b, m1a, m1b, dlo, dhi;
    if (mode == MODE_REFLECT) {                     // align_corners=False: mirrors about -0.5 and W-0.5
        m0a = -fs - 2.f; m0b = -fs; m1a = 2.f * fW - fs - 2.f; m1b = 2.f * fW - fs;
        dlo = -fW - 0.5f; dhi = 2.f * fW - 0.5f;
    } else {                                        // align_corners=True: mirrors about 0 and W-1
        const float span = fW - 1.f;
        m0a = -fs - 1.f; m0b = -fs + 1.f; m1a = 2.f * span - fs - 1.f; m1b = 2.f * span - fs + 1.f;
        dlo = -span; dhi = 2.f * span;
    }
    if (lo < dlo || hi > dhi) { ia[0] = lo; ib[0] = hi; return 1; }
    // the three images can touch (s near an edge: the slack makes neighbours overlap) -> merge while inserting, in order
    m0a = fmaxf(m0a - eps, lo); m0b = fminf(m0b + eps, hi);
    m1a = fmaxf(m1a - eps, lo); m1b = fminf(m1b + eps, hi);
    int cnt = 0;
    float ra[3], rb[3];
    if (m0b > m0a) { ra[cnt] = m0a; rb[cnt] = m0b; ++cnt; }                     // lowest
    if (n) {
        if (cnt && ia[0] <= rb[cnt - 1]) rb[cnt - 1] = fmaxf(rb[cnt - 1], ib[0]);
        else { ra[cnt] = ia[0]; rb[cnt] = ib[0]; ++cnt; }
    }
    if (m1b > m1a) {
        if (cnt && m1a <= rb[cnt - 1]) rb[cnt - 1] = fmaxf(rb[cnt - 1], m1b);
        else { ra[cnt] = m1a; rb[cnt] = m1b; ++cnt; }
    }
    for (int i = 0; i < cnt; ++i) { ia[i] = ra[i]; ib[i] = rb[i]; }
    return cnt;
}

// bounding box (inclusive, clipped to the destination image) of the pre-image of the raw rectangle [ua,ub) x [va,vb)
__device__ __forceinline__ bool preimage_box(const StageMap& sm, float ua, float ub, float va, float vb, int Wd, int Hd,
                                             int& x0, int& x1, int& y0, int& y1) {
    float xl = INFINITY, xh = -INFINITY, yl = INFINITY, yh = -INFINITY;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float u = (k & 1) ? ub : ua, v = (k & 2) ? vb : va;
        const float z = sm.Pi[6] * u + sm.Pi[7] * v + sm.Pi[8];
        if (!(z > 0.05f)) { bad = true; continue; }
        const float iz = 1.f / z;
        const float x = (sm.Pi[0] * u + sm.Pi[1] * v + sm.Pi[2]) * iz, y = (sm.Pi[3] * u + sm.Pi[4] * v + sm.Pi[5]) * iz;
        xl = fminf(xl, x); xh = fmaxf(xh, x); yl = fminf(yl, y); yh = fmaxf(yh, y);
    }
    if (bad) { x0 = 0; x1 = Wd - 1; y0 = 0; y1 = Hd - 1; return true; }        // near the horizon: search everything
    // the fp32 inverse map is good to a small fraction of a pixel (it only ignores the forward's fp32 grid rounding)
    x0 = max((int)floorf(xl - 0.25f), 0); x1 = min((int)ceilf(xh + 0.25f), Wd - 1);
    y0 = max((int)floorf(yl - 0.25f), 0); y1 = min((int)ceilf(yh + 0.25f), Hd - 1);
    return x0 <= x1 && y0 <= y1;
}

// what one stage's gather needs to know
struct GatherStage {
    const double* m; int gtype, mode;
    int Wd, Hd;            // destination plane
    int Ws, Hs;            // source image the taps index (the stage-B window, or the whole stage-A source)
    const float* g;        // destination-side gradient planes of this cutout, [3][Hd][Wd]
    const float2* uv;      // the forward's raw source coordinate of every destination pixel of this cutout, [Hd][Wd] (uv_kernel)
    unsigned long long* dbg = nullptr;   // PRX_CUTOUT_DBG=1: {candidates visited, rectangles, candidates inside their rectangle, with a tap in the tile}
};

// LDS staging of one block's candidates: neighbouring source pixels share almost all of their destination candidates, so
// the block loads the bounding box of its whole source tile's (direct) pre-image ONCE, coalesced -- raw coordinates and the
// three gradient planes -- and the per-pixel enumeration then reads LDS.  Candidates outside the staged box (mirror images,
// border strips, boxes too large for the buffer) fall back to HBM / L2 reads.
constexpr int TILE_W = 16;           // a block owns a TILE_W x TILE_W tile of source pixels
constexpr int STAGE_CAP = 2304;       // destination pixels (48 x 48): 18 KB of coordinates + 27 KB of gradients
struct TileStage {
    int x0, y0, w, h;                 // staged destination box (w == 0: nothing staged)
    float2 uv[STAGE_CAP];
    float g[3][STAGE_CAP];
};

// the forward's padded sampling coordinate along one axis (what make_taps does before it takes the floor)
__device__ __forceinline__ float pad_coord(float u, int W, int mode) {
    if (mode == MODE_BORDER) return fminf(fmaxf(u, 0.f), (float)(W - 1));
    if (mode == MODE_REFLECT) return fminf(fmaxf(reflect_coord(u, (float)W), 0.f), (float)(W - 1));
    if (mode == MODE_REFLECT_AC) return fminf(fmaxf(reflect_coord_ac(u, (float)W), 0.f), (float)(W - 1));
    return u;
}
// bilinear weight of source index s for the padded coordinate t, with the forward's own roundings: the tap pair is
// (floor t, floor t + 1) with weights (1 - (t - floor t), t - floor t)
__device__ __forceinline__ float tap_weight(float t, int s) {
    const float d = t - (float)s;
    if (d >= 0.f) return d < 1.f ? 1.f - d : 0.f;                 // floor t == s      : weight 1 - (t - s)
    return d > -1.f ? t - (float)(s - 1) : 0.f;                    // floor t == s - 1  : weight t - (s - 1)
}

// contribution of destination pixel (x, y) to source pixel (sx, sy), if its raw coordinate lies in [ua,ub) x [va,vb).
// The raw coordinate comes from the uv map a destination-parallel pre-pass wrote with the forward's own `project` (fp64
// homography -> fp32 grid -> unnormalise), so the weights below are bit-for-bit the forward's (y-weight * x-weight, as
// sample_plane multiplies them).
__device__ __forceinline__ void gather_candidate(const GatherStage& st, const TileStage& ts, int x, int y, int sx, int sy,
                                                 float ua, float ub, float va, float vb, float (&acc)[3]) {
    const int lx = x - ts.x0, ly = y - ts.y0;
    const bool in_lds = lx >= 0 && lx < ts.w && ly >= 0 && ly < ts.h;
    const int li = ly * ts.w + lx;
    const size_t o = (size_t)y * st.Wd + x;
    const float2 q = in_lds ? ts.uv[li] : st.uv[o];
    if (!(q.x >= ua && q.x < ub && q.y >= va && q.y < vb)) return;
    const float wx = tap_weight(pad_coord(q.x, st.Ws, st.mode), sx);
    const float wy = tap_weight(pad_coord(q.y, st.Hs, st.mode), sy);
    const float w = wy * wx;
    if (w == 0.f) return;
    if (in_lds) {
        acc[0] += ts.g[0][li] * w; acc[1] += ts.g[1][li] * w; acc[2] += ts.g[2][li] * w;
    } else {
        const size_t plane = (size_t)st.Hd * st.Wd;
        acc[0] += st.g[o] * w; acc[1] += st.g[plane + o] * w; acc[2] += st.g[2 * plane + o] * w;
    }
}

// All contributions to source pixel (sx, sy).  `coop`: the 64 lanes of the wave share the enumeration of ONE pixel (every
// lane passes the same sx, sy) and the partial sums are combined by a fixed butterfly; otherwise the lane works alone.
constexpr int HEAVY = 96;        // candidates above which a pixel is handed to the whole wave (border strips, corners)
template <bool COOP>
__device__ __forceinline__ int gather_pixel(const GatherStage& st, const StageMap& sm, const TileStage& ts, int sx, int sy,
                                            float (&acc)[3], int budget) {
    float xa[MAXI], xb[MAXI], ya[MAXI], yb[MAXI];
    const int nx = tap_intervals(st.mode, sx, st.Ws, sm.ulo, sm.uhi, xa, xb);
    const int ny = tap_intervals(st.mode, sy, st.Hs, sm.vlo, sm.vhi, ya, yb);
    const int lane = threadIdx.x & 63;
    int total = 0;
    // Only border padding creates unbounded rectangles (an edge pixel owns the whole strip of padded area beside it); under
    // reflection / zeros / fill every rectangle is a 2 x 2 source square, so the counting pass is skipped there.
    const bool may_be_heavy = st.mode == MODE_BORDER;
    if (!COOP && (may_be_heavy || (nx == 1 && ny == 1))) {       // first pass: how much work is it?
        if (nx == 1 && ny == 1) {                  // the common case (interior pixel, or zero / fill padding): one box, used directly
            int x0, x1, y0, y1;
            if (!preimage_box(sm, xa[0], xb[0], ya[0], yb[0], st.Wd, st.Hd, x0, x1, y0, y1)) return 0;
            total = (x1 - x0 + 1) * (y1 - y0 + 1);
            if (total > budget && may_be_heavy) return total;
            if (!may_be_heavy) total = 0;              // a large magnification is uniform over the wave: every lane works alone
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x) gather_candidate(st, ts, x, y, sx, sy, xa[0], xb[0], ya[0], yb[0], acc);
            return total;
        }
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                int x0, x1, y0, y1;
                if (preimage_box(sm, xa[i], xb[i], ya[j], yb[j], st.Wd, st.Hd, x0, x1, y0, y1)) total += (x1 - x0 + 1) * (y1 - y0 + 1);
            }
        if (total > budget) return total;          // too much for one lane: the caller schedules a cooperative pass
    }
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            int x0, x1, y0, y1;
            if (!preimage_box(sm, xa[i], xb[i], ya[j], yb[j], st.Wd, st.Hd, x0, x1, y0, y1)) continue;
            const int bw = x1 - x0 + 1, cnt = bw * (y1 - y0 + 1);
            if (COOP) {
                for (int k = lane; k < cnt; k += 64)
                    gather_candidate(st, ts, x0 + k % bw, y0 + k / bw, sx, sy, xa[i], xb[i], ya[j], yb[j], acc);
            } else {
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x) gather_candidate(st, ts, x, y, sx, sy, xa[i], xb[i], ya[j], yb[j], acc);
            }
        }
    if (COOP) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    }
    return total;
}

// one block = a 16 x 16 tile of SOURCE pixels of one cutout; out-of-tile / out-of-window pixels idle
// (tx0, ty0): source coordinates of the tile's first pixel (may be negative / beyond the image at the window edges)
__device__ __forceinline__ void gather_tile(const GatherStage& st, const StageMap& sm, TileStage& ts, int tx0, int ty0, int sx, int sy,
                                            bool live, float (&out)[3]) {
    // stage the pre-image of the whole tile (its direct image: source square [tx0-1, tx0+TILE_W+1) etc.)
    if (threadIdx.x == 0) {
        int x0, x1, y0, y1;
        ts.w = 0; ts.h = 0; ts.x0 = 0; ts.y0 = 0;
        const float ua = fmaxf((float)tx0 - 1.f, sm.ulo), ub = fminf((float)(tx0 + TILE_W) + 1.f, sm.uhi);
        const float va = fmaxf((float)ty0 - 1.f, sm.vlo), vb = fminf((float)(ty0 + TILE_W) + 1.f, sm.vhi);
        if (ub > ua && vb > va && preimage_box(sm, ua, ub, va, vb, st.Wd, st.Hd, x0, x1, y0, y1) &&
            (x1 - x0 + 1) * (y1 - y0 + 1) <= STAGE_CAP) {
            ts.x0 = x0; ts.y0 = y0; ts.w = x1 - x0 + 1; ts.h = y1 - y0 + 1;
        }
    }
    __syncthreads();
    {
        const int cnt = ts.w * ts.h;
        const size_t plane = (size_t)st.Hd * st.Wd;
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const size_t o = (size_t)(ts.y0 + k / ts.w) * st.Wd + (ts.x0 + k % ts.w);
            ts.uv[k] = st.uv[o];
            ts.g[0][k] = st.g[o]; ts.g[1][k] = st.g[plane + o]; ts.g[2][k] = st.g[2 * plane + o];
        }
    }
    __syncthreads();
    float acc[3] = {0.f, 0.f, 0.f};
    int work = 0;
    if (live) work = gather_pixel<false>(st, sm, ts, sx, sy, acc, HEAVY);
    // pixels that were too heavy for one lane (work > HEAVY: nothing accumulated yet): the wave takes them one by one
    unsigned long long heavy = __ballot(live && work > HEAVY);
    while (heavy) {
        const int src = __ffsll((long long)heavy) - 1;
        heavy &= heavy - 1;
        const int hx = __shfl(sx, src, 64), hy = __shfl(sy, src, 64);
        float a2[3] = {0.f, 0.f, 0.f};
        gather_pixel<true>(st, sm, ts, hx, hy, a2, 0);
        if ((int)(threadIdx.x & 63) == src) { acc[0] = a2[0]; acc[1] = a2[1]; acc[2] = a2[2]; }
    }
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
}

// ---- backward, tile-owned scatter form (round 3) ----------------------------------------------------------------------
// The per-PIXEL gather above spends ~1-2 k instructions of pre-image geometry per source pixel before it touches a candidate
// (0.54 ms per iteration at the headline: 10x its HBM bound).  Same ownership, coarser granularity: a workgroup owns a
// 16 x 16 tile of SOURCE pixels, works out the raw-coordinate rectangles / destination boxes ONCE for the tile, and then runs
// DESTINATION-parallel over the candidates -- each lane takes one destination pixel, rebuilds the forward's own four taps
// (make_taps on the stored raw coordinate: same roundings, exact adjoint) and adds the taps that fall inside the tile into an
// LDS accumulator.  No global atomics, no memsets, every gradient element written exactly once, and still bit-reproducible:
//   * candidates are visited in a fixed order (rectangle by rectangle, k = thread id + 256 j);
//   * each of the block's 4 waves has its OWN accumulator plane, so no two waves ever add to the same word; within a wave the
//     additions of one ds_add_f32 instruction are serialised by the LDS in a fixed lane order, and instructions retire in
//     program order;
//   * the four planes are summed in wave order at the end.
// Uniqueness / completeness as in the gather: the tile's raw rectangles are pairwise disjoint and a candidate is counted in
// the rectangle that contains its raw coordinate; acceptance is by the exact taps, the boxes only bound the search.
constexpr int MAXR = 3;
// raw-coordinate intervals whose padded coordinate can produce a tap on a source index in [s0, s1] (tile range, clipped to
// the image), clipped to [lo, hi]: the tile-level version of tap_intervals
__device__ __forceinline__ int tile_intervals(int mode, int s0, int s1, int W, float lo, float hi, float* ia, float* ib) {
    const float eps = 2e-3f;
    const float f0 = (float)s0, f1 = (float)s1, fW = (float)W;
    float a = f0 - 1.f - eps, b = f1 + 1.f + eps;
    if (mode == MODE_BORDER) {
        if (s0 == 0) a = -INFINITY;
        if (s1 == W - 1) b = INFINITY;
    }
    a = fmaxf(a, lo); b = fminf(b, hi);
    int n = 0;
    if (b > a) { ia[0] = a; ib[0] = b; n = 1; }
    if (mode != MODE_REFLECT && mode != MODE_REFLECT_AC) return n;
    float m0a, m0b, m1a, m1b, dlo, dhi;
    if (mode == MODE_REFLECT) {                     // mirrors about -0.5 and W - 0.5: t -> -1 - t, t -> 2W - 1 - t
        m0a = -f1 - 2.f; m0b = -f0; m1a = 2.f * fW - f1 - 2.f; m1b = 2.f * fW - f0;
        dlo = -fW - 0.5f; dhi = 2.f * fW - 0.5f;
    } else {                                        // mirrors about 0 and W - 1
        const float span = fW - 1.f;
        m0a = -f1 - 1.f; m0b = -f0 + 1.f; m1a = 2.f * span - f1 - 1.f; m1b = 2.f * span - f0 + 1.f;
        dlo = -span; dhi = 2.f * span;
    }
    if (lo < dlo || hi > dhi) { ia[0] = lo; ib[0] = hi; return 1; }       // beyond the first-order mirrors: search the whole range
    m0a = fmaxf(m0a - eps, lo); m0b = fminf(m0b + eps, hi);
    m1a = fmaxf(m1a - eps, lo); m1b = fminf(m1b + eps, hi);
    int cnt = 0;
    float ra[3], rb[3];
    if (m0b > m0a) { ra[cnt] = m0a; rb[cnt] = m0b; ++cnt; }
    if (n) {
        if (cnt && ia[0] <= rb[cnt - 1]) rb[cnt - 1] = fmaxf(rb[cnt - 1], ib[0]);
        else { ra[cnt] = ia[0]; rb[cnt] = ib[0]; ++cnt; }
    }
    if (m1b > m1a) {
        if (cnt && m1a <= rb[cnt - 1]) rb[cnt - 1] = fmaxf(rb[cnt - 1], m1b);
        else { ra[cnt] = m1a; rb[cnt] = m1b; ++cnt; }
    }
    for (int i = 0; i < cnt; ++i) { ia[i] = ra[i]; ib[i] = rb[i]; }
    return cnt;
}

struct TileScatter {
    int nrect;
    float ua[MAXR * MAXR], ub[MAXR * MAXR], va[MAXR * MAXR], vb[MAXR * MAXR];     // disjoint raw rectangles of the tile
    int x0[MAXR * MAXR], y0[MAXR * MAXR], bw[MAXR * MAXR], cnt[MAXR * MAXR];      // their destination boxes
    float acc[4][3][TILE_W * TILE_W];                                                 // one accumulator plane set per wave
};

// one candidate: destination pixel with raw coordinate q and gradient (g0, g1, g2) -> LDS accumulator of the tile
__device__ __forceinline__ int scatter_candidate(const GatherStage& st, float* __restrict__ a0, float* __restrict__ a1, float* __restrict__ a2,
                                                 int tx0, int ty0, float ua, float ub, float va, float vb, float2 q, float g0, float g1, float g2) {
    if (!(q.x >= ua && q.x < ub && q.y >= va && q.y < vb)) return 0;
    const Taps t = make_taps(q.x, q.y, st.Ws, st.Hs, st.mode);          // the forward's own taps
    const int lx = t.x0 - tx0, ly = t.y0 - ty0;                          // north-west tap inside the tile?
    const bool cx0 = t.vx0 && lx >= 0 && lx < TILE_W, cx1 = t.vx1 && lx + 1 >= 0 && lx + 1 < TILE_W;
    const bool cy0 = t.vy0 && ly >= 0 && ly < TILE_W, cy1 = t.vy1 && ly + 1 >= 0 && ly + 1 < TILE_W;
    if (!((cx0 || cx1) && (cy0 || cy1))) return 1;
    const float e = 1.f - t.wx, s_ = 1.f - t.wy;                        // sample_plane's weights, same products
    const float w00 = s_ * e, w01 = s_ * t.wx, w10 = t.wy * e, w11 = t.wy * t.wx;
    const int p00 = ly * TILE_W + lx;
    if (cx0 && cy0) { atomicAdd(&a0[p00], g0 * w00); atomicAdd(&a1[p00], g1 * w00); atomicAdd(&a2[p00], g2 * w00); }
    if (cx1 && cy0) { atomicAdd(&a0[p00 + 1], g0 * w01); atomicAdd(&a1[p00 + 1], g1 * w01); atomicAdd(&a2[p00 + 1], g2 * w01); }
    if (cx0 && cy1) { atomicAdd(&a0[p00 + TILE_W], g0 * w10); atomicAdd(&a1[p00 + TILE_W], g1 * w10); atomicAdd(&a2[p00 + TILE_W], g2 * w10); }
    if (cx1 && cy1) { atomicAdd(&a0[p00 + TILE_W + 1], g0 * w11); atomicAdd(&a1[p00 + TILE_W + 1], g1 * w11); atomicAdd(&a2[p00 + TILE_W + 1], g2 * w11); }
    return 2;
}

// all contributions to the 16 x 16 source tile at (tx0, ty0) (source-window coordinates); thread t returns the sums of its own
// source pixel (tx0 + (t & 15), ty0 + (t >> 4)).  Must be called by all 256 threads of the block.
__device__ __forceinline__ void scatter_tile(const GatherStage& st, const StageMap& sm, TileScatter& ts, int tx0, int ty0, float (&out)[3]) {
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) {
        int n = 0;
        const int sx0 = max(tx0, 0), sx1 = min(tx0 + TILE_W - 1, st.Ws - 1);
        const int sy0 = max(ty0, 0), sy1 = min(ty0 + TILE_W - 1, st.Hs - 1);
        if (sx0 <= sx1 && sy0 <= sy1) {
            float xa[MAXR], xb[MAXR], ya[MAXR], yb[MAXR];
            const int nx = tile_intervals(st.mode, sx0, sx1, st.Ws, sm.ulo, sm.uhi, xa, xb);
            const int ny = tile_intervals(st.mode, sy0, sy1, st.Hs, sm.vlo, sm.vhi, ya, yb);
            for (int j = 0; j < ny; ++j)
                for (int i = 0; i < nx; ++i) {
                    int x0, x1, y0, y1;
                    if (!preimage_box(sm, xa[i], xb[i], ya[j], yb[j], st.Wd, st.Hd, x0, x1, y0, y1)) continue;
                    ts.ua[n] = xa[i]; ts.ub[n] = xb[i]; ts.va[n] = ya[j]; ts.vb[n] = yb[j];
                    ts.x0[n] = x0; ts.y0[n] = y0; ts.bw[n] = x1 - x0 + 1; ts.cnt[n] = (x1 - x0 + 1) * (y1 - y0 + 1);
                    ++n;
                }
        }
        ts.nrect = n;
        if (st.dbg) {
            unsigned long long tot = 0;
            for (int r = 0; r < n; ++r) tot += (unsigned long long)ts.cnt[r];
            atomicAdd(&st.dbg[0], tot);
            atomicAdd(&st.dbg[1], (unsigned long long)n);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int w = 0; w < 4; ++w) ts.acc[w][c][tid] = 0.f;
    __syncthreads();
    const size_t plane = (size_t)st.Hd * st.Wd;
    const int nrect = ts.nrect;
    float* a0 = ts.acc[wave][0]; float* a1 = ts.acc[wave][1]; float* a2 = ts.acc[wave][2];
    constexpr int U = 4;                 // candidates per thread per trip: their 16 loads are in flight together
    int n_in = 0, n_tap = 0;
    for (int r = 0; r < nrect; ++r) {
        const float ua = ts.ua[r], ub = ts.ub[r], va = ts.va[r], vb = ts.vb[r];
        const int bx0 = ts.x0[r], by0 = ts.y0[r], bw = ts.bw[r], cnt = ts.cnt[r];
        const float ibw = 1.f / (float)bw;
        for (int k0 = tid; k0 < cnt; k0 += 256 * U) {
            float2 q[U]; float g0[U], g1[U], g2[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int k = k0 + 256 * j;
                q[j] = make_float2(-INFINITY, -INFINITY);         // fails every rectangle test
                g0[j] = g1[j] = g2[j] = 0.f;
                if (k < cnt) {
                    int ky = (int)((float)k * ibw);                // k / bw without the integer division (cnt < 2^23; corrected below)
                    int kx = k - ky * bw;
                    if (kx < 0) { --ky; kx += bw; } else if (kx >= bw) { ++ky; kx -= bw; }
                    const size_t o = (size_t)(by0 + ky) * st.Wd + (bx0 + kx);
                    q[j] = st.uv[o];
                    g0[j] = st.g[o]; g1[j] = st.g[plane + o]; g2[j] = st.g[2 * plane + o];
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int code = scatter_candidate(st, a0, a1, a2, tx0, ty0, ua, ub, va, vb, q[j], g0[j], g1[j], g2[j]);
                n_in += code >= 1; n_tap += code == 2;
            }
        }
    }
    if (st.dbg) { atomicAdd(&st.dbg[2], (unsigned long long)n_in); atomicAdd(&st.dbg[3], (unsigned long long)n_tap); }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = ((ts.acc[0][c][tid] + ts.acc[1][c][tid]) + ts.acc[2][c][tid]) + ts.acc[3][c][tid];
}

// ---- the same, ONE WAVE per tile ---------------------------------------------------------------------------------------
// Measured (profiles/r03_cfg1_kernel_stats.csv): with 256-thread blocks the two scatter kernels still took 170-190 us, the
// same with and without precomputed stage maps and unrolled loads -- a tile has only ~350 candidates (1.4 trips of 256
// lanes), so a block's life is its serial chain: descriptor words -> rectangles by one lane -> barrier -> loads -> LDS adds ->
// barrier -> store, ~10 us, with 6 blocks resident per CU.  One wave per tile quadruples the tiles in flight per CU, needs
// no workgroup barrier at all (the wave's own LDS traffic is ordered), computes the <= 9 rectangle boxes on 9 lanes in
// parallel, and is reproducible by construction: one accumulator, fixed candidate order, in-order LDS.
struct WaveScatter {
    float ua[MAXR * MAXR], ub[MAXR * MAXR], va[MAXR * MAXR], vb[MAXR * MAXR];
    int x0[MAXR * MAXR], y0[MAXR * MAXR], bw[MAXR * MAXR], cnt[MAXR * MAXR];
    float acc[3][TILE_W * TILE_W];
};

// lane l returns the sums of source pixels l, l + 64, l + 128, l + 192 of the tile (row-major 16 x 16)
__device__ __forceinline__ void scatter_tile_wave(const GatherStage& st, const StageMap& sm, WaveScatter& ts, int tx0, int ty0, float (&out)[4][3]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) ts.acc[c][lane + 64 * j] = 0.f;
    if (lane < MAXR * MAXR) {                    // rectangle (i, j) = (lane % 3, lane / 3) on its own lane
        int cnt = 0;
        const int sx0 = max(tx0, 0), sx1 = min(tx0 + TILE_W - 1, st.Ws - 1);
        const int sy0 = max(ty0, 0), sy1 = min(ty0 + TILE_W - 1, st.Hs - 1);
        if (sx0 <= sx1 && sy0 <= sy1) {
            float xa[MAXR], xb[MAXR], ya[MAXR], yb[MAXR];
            const int nx = tile_intervals(st.mode, sx0, sx1, st.Ws, sm.ulo, sm.uhi, xa, xb);
            const int ny = tile_intervals(st.mode, sy0, sy1, st.Hs, sm.vlo, sm.vhi, ya, yb);
            const int i = lane % MAXR, j = lane / MAXR;
            if (i < nx && j < ny) {
                // static selection (no dynamically indexed register arrays)
                const float ua = i == 0 ? xa[0] : (i == 1 ? xa[1] : xa[2]), ub = i == 0 ? xb[0] : (i == 1 ? xb[1] : xb[2]);
                const float va = j == 0 ? ya[0] : (j == 1 ? ya[1] : ya[2]), vb = j == 0 ? yb[0] : (j == 1 ? yb[1] : yb[2]);
                int x0, x1, y0, y1;
                if (preimage_box(sm, ua, ub, va, vb, st.Wd, st.Hd, x0, x1, y0, y1)) {
                    ts.ua[lane] = ua; ts.ub[lane] = ub; ts.va[lane] = va; ts.vb[lane] = vb;
                    ts.x0[lane] = x0; ts.y0[lane] = y0; ts.bw[lane] = x1 - x0 + 1;
                    cnt = (x1 - x0 + 1) * (y1 - y0 + 1);
                }
            }
        }
        ts.cnt[lane] = cnt;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's LDS writes above are visible to its reads below
    const size_t plane = (size_t)st.Hd * st.Wd;
    constexpr int U = 4;
    for (int r = 0; r < MAXR * MAXR; ++r) {
        const int cnt = ts.cnt[r];
        if (cnt == 0) continue;
        const float ua = ts.ua[r], ub = ts.ub[r], va = ts.va[r], vb = ts.vb[r];
        const int bx0 = ts.x0[r], by0 = ts.y0[r], bw = ts.bw[r];
        const float ibw = 1.f / (float)bw;
        for (int k0 = lane; k0 < cnt; k0 += 64 * U) {
            float2 q[U]; float g0[U], g1[U], g2[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int k = k0 + 64 * j;
                q[j] = make_float2(-INFINITY, -INFINITY);
                g0[j] = g1[j] = g2[j] = 0.f;
                if (k < cnt) {
                    int ky = (int)((float)k * ibw);
                    int kx = k - ky * bw;
                    if (kx < 0) { --ky; kx += bw; } else if (kx >= bw) { ++ky; kx -= bw; }
                    const size_t o = (size_t)(by0 + ky) * st.Wd + (bx0 + kx);
                    q[j] = st.uv[o];
                    g0[j] = st.g[o]; g1[j] = st.g[plane + o]; g2[j] = st.g[2 * plane + o];
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) scatter_candidate(st, ts.acc[0], ts.acc[1], ts.acc[2], tx0, ty0, ua, ub, va, vb, q[j], g0[j], g1[j], g2[j]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[j][c] = ts.acc[c][lane + 64 * j];
}

// the stage map of every cutout, once per stage (it is the same for all tiles of a cutout; built from fp64 it costs a single
// lane ~3 us, which every one of the 196 tile blocks of a cutout used to spend on its own): maps[n] for stage 1 / 2
__global__ void stage_map_kernel(const double* __restrict__ desc, int stage, StageMap* __restrict__ maps, int n_cut, int Wd, int Hd, int Ws, int Hs) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_cut) return;
    const double* d = desc + (size_t)n * DESC_WORDS;
    StageMap sm{};
    if ((int)d[stage == 1 ? D_MODE1 : D_MODE2] != MODE_IDENT) {
        if (stage == 2) { Ws = (int)d[D_WW]; Hs = (int)d[D_WH]; }
        build_stage_map(sm, d + (stage == 1 ? D_M1 : D_M2), (int)d[stage == 1 ? D_GRID1 : D_GRID2], Wd, Hd, Ws, Hs);
    }
    maps[n] = sm;
}

// destination-parallel pre-pass of a gather stage: the raw source coordinate of every destination pixel, exactly as the
// forward computed it (stage 1: D_M1 / D_GRID1 / D_MODE1 on the Ha x Wa stage-A plane; stage 2: the stage-B words on S x S)
__global__ __launch_bounds__(256) void uv_kernel(const double* __restrict__ desc, int stage, float2* __restrict__ uv, int Wd, int Hd,
                                                 int Ws, int Hs) {
    const int n = blockIdx.y;
    const double* d = desc + (size_t)n * DESC_WORDS;
    if ((int)d[stage == 1 ? D_MODE1 : D_MODE2] == MODE_IDENT) return;
    const double* m = d + (stage == 1 ? D_M1 : D_M2);
    const int gtype = (int)d[stage == 1 ? D_GRID1 : D_GRID2];
    if (stage == 2) { Ws = (int)d[D_WW]; Hs = (int)d[D_WH]; }
    const size_t plane = (size_t)Hd * Wd;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < plane; pix += (size_t)gridDim.x * blockDim.x) {
        float u, v;
        project(m, gtype, (int)(pix % Wd), (int)(pix / Wd), Wd, Hd, Ws, Hs, u, v);
        uv[(size_t)n * plane + pix] = make_float2(u, v);
    }
}

// Stage A backward: g[n][3][Ha][Wa] -> per-cutout private source-gradient planes gsrc[n][3][Hs][Ws] (every element written;
// summed over n afterwards by reduce_planes_kernel in a fixed order)
__global__ __launch_bounds__(256) void warp_a_bwd_kernel(const float* __restrict__ g, int Hs, int Ws,
                                                         const double* __restrict__ desc, const float2* __restrict__ uv,
                                                         float* __restrict__ gsrc, int n_cut, int Ha, int Wa) {
    __shared__ StageMap sm;
    __shared__ TileStage ts;
    const int tiles = (Ws + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int tx0 = (blockIdx.x % tiles) * TILE_W, ty0 = (blockIdx.x / tiles) * TILE_W;
    const int sx = tx0 + (threadIdx.x & 15);
    const int sy = ty0 + (threadIdx.x >> 4);
    const bool live = sx < Ws && sy < Hs;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE1];
    const size_t plane = (size_t)Ha * Wa, splane = (size_t)Hs * Ws;
    const float* gi = g + (size_t)n * 3 * plane;
    float* gs = gsrc + (size_t)n * 3 * splane + (size_t)sy * Ws + sx;
    if (mode == MODE_IDENT) {      // 1:1 copy (Ha x Wa == Hs x Ws)
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gs[(size_t)c * splane] = gi[(size_t)c * plane + (size_t)sy * Wa + sx];
        }
        return;
    }
    if (threadIdx.x == 0) build_stage_map(sm, d + D_M1, (int)d[D_GRID1], Wa, Ha, Ws, Hs);
    __syncthreads();
    GatherStage st{d + D_M1, (int)d[D_GRID1], mode, Wa, Ha, Ws, Hs, gi, uv + (size_t)n * plane};
    float o[3];
    gather_tile(st, sm, ts, tx0, ty0, sx, sy, live, o);
    if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gs[(size_t)c * splane] = o[c];
    }
}

// the same in the tile-owned scatter form (scatter_tile)
__global__ __launch_bounds__(256) void warp_a_bwd2_kernel(const float* __restrict__ g, int Hs, int Ws,
                                                          const double* __restrict__ desc, const float2* __restrict__ uv,
                                                          const StageMap* __restrict__ maps, float* __restrict__ gsrc, int n_cut, int Ha, int Wa,
                                                          unsigned long long* dbg) {
    __shared__ TileScatter ts;
    const int tiles = (Ws + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int tx0 = (blockIdx.x % tiles) * TILE_W, ty0 = (blockIdx.x / tiles) * TILE_W;
    const int sx = tx0 + (threadIdx.x & 15);
    const int sy = ty0 + (threadIdx.x >> 4);
    const bool live = sx < Ws && sy < Hs;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE1];
    const size_t plane = (size_t)Ha * Wa, splane = (size_t)Hs * Ws;
    const float* gi = g + (size_t)n * 3 * plane;
    float* gs = gsrc + (size_t)n * 3 * splane + (size_t)sy * Ws + sx;
    if (mode == MODE_IDENT) {
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gs[(size_t)c * splane] = gi[(size_t)c * plane + (size_t)sy * Wa + sx];
        }
        return;
    }
    const StageMap sm = maps[n];          // uniform address: scalar loads
    GatherStage st{d + D_M1, (int)d[D_GRID1], mode, Wa, Ha, Ws, Hs, gi, uv + (size_t)n * plane, dbg};
    float o[3];
    scatter_tile(st, sm, ts, tx0, ty0, o);
    if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gs[(size_t)c * splane] = o[c];
    }
}

// one wave per tile (scatter_tile_wave)
__global__ __launch_bounds__(64) void warp_a_bwd3_kernel(const float* __restrict__ g, int Hs, int Ws,
                                                         const double* __restrict__ desc, const float2* __restrict__ uv,
                                                         const StageMap* __restrict__ maps, float* __restrict__ gsrc, int n_cut, int Ha, int Wa) {
    __shared__ WaveScatter ts;
    const int tiles = (Ws + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int tx0 = (blockIdx.x % tiles) * TILE_W, ty0 = (blockIdx.x / tiles) * TILE_W;
    const int lane = threadIdx.x;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE1];
    const size_t plane = (size_t)Ha * Wa, splane = (size_t)Hs * Ws;
    const float* gi = g + (size_t)n * 3 * plane;
    float* gs = gsrc + (size_t)n * 3 * splane;
    if (mode == MODE_IDENT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = lane + 64 * j, sx = tx0 + (p & 15), sy = ty0 + (p >> 4);
            if (sx < Ws && sy < Hs)
#pragma unroll
                for (int c = 0; c < 3; ++c) gs[(size_t)c * splane + (size_t)sy * Ws + sx] = gi[(size_t)c * plane + (size_t)sy * Wa + sx];
        }
        return;
    }
    const StageMap sm = maps[n];
    GatherStage st{d + D_M1, (int)d[D_GRID1], mode, Wa, Ha, Ws, Hs, gi, uv + (size_t)n * plane};
    float o[4][3];
    scatter_tile_wave(st, sm, ts, tx0, ty0, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = lane + 64 * j, sx = tx0 + (p & 15), sy = ty0 + (p >> 4);
        if (sx < Ws && sy < Hs)
#pragma unroll
            for (int c = 0; c < 3; ++c) gs[(size_t)c * splane + (size_t)sy * Ws + sx] = o[j][c];
    }
}

// out[i] = sum_n planes[n][i]
__global__ __launch_bounds__(256) void reduce_planes_kernel(const float* __restrict__ planes, float* __restrict__ out,
                                                            int n, size_t plane_elems) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane_elems; i += (size_t)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += planes[(size_t)k * plane_elems + i];
        out[i] = acc;
    }
}

// Stage B (+ ColorJitter + noise): out[n] (S x S) from this cutout's stage-A image a[n] (Ha x Wa), read through the
// descriptor's source window (x0, y0, w, h): the whole image on a square canvas and for the zoom set; for the wide set
// on a W != H canvas the centred S x S crop the reference takes (CenterCrop, pixray.py:433) -- out-of-window taps are
// out-of-image taps, exactly as if the crop had been materialised.
struct SrcWin { int ox, oy, ww, wh; };
__device__ __forceinline__ SrcWin src_window(const double* d) {
    SrcWin q; q.ox = (int)d[D_WOX]; q.oy = (int)d[D_WOY]; q.ww = (int)d[D_WW]; q.wh = (int)d[D_WH];
    return q;
}

// The additive noise of pixray.py:508-510 (`batch + fac * randn_like(batch)`) drawn inside the kernel when the caller hands no
// noise tensor: Philox4x32-10 (Salmon et al., SC'11; the generator behind torch.randn on a GPU) keyed by the cutout's seed word,
// counter = the pixel index, two Box-Muller pairs per call -> the pixel's three channels.  Saves the randn launch, its 12 bytes
// per pixel of stores and their re-read here.
__device__ __forceinline__ void philox_normal3(unsigned long long seed, unsigned pix, float (&z)[3]) {
    unsigned c0 = pix, c1 = 0u, c2 = 0x5bd1e995u, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.f / 16777216.f), u1 = (float)(c1 >> 8) * (1.f / 16777216.f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.f / 16777216.f), u3 = (float)(c3 >> 8) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    float s0, q0, s1, q1;
    sincosf(6.283185307179586f * u1, &s0, &q0);
    sincosf(6.283185307179586f * u3, &s1, &q1);
    z[0] = r0 * q0; z[1] = r0 * s0; z[2] = r1 * q1;
    (void)s1;
}

__global__ __launch_bounds__(256) void warp_b_fwd_kernel(const float* __restrict__ a, int Ha, int Wa, const double* __restrict__ desc,
                                                         const float* __restrict__ noise, float* __restrict__ out,
                                                         int n_cut, int S) {
    const size_t total = (size_t)n_cut * S * S;
    const size_t plane = (size_t)S * S, aplane = (size_t)Ha * Wa;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % S), y = (int)((idx / S) % S), n = (int)(idx / plane);
        const double* d = desc + (size_t)n * DESC_WORDS;
        const int mode = (int)d[D_MODE2];
        const SrcWin q = src_window(d);
        const float* an = a + (size_t)n * 3 * aplane + (size_t)q.oy * Wa + q.ox;     // window origin
        const size_t pix = (size_t)y * S + x;
        Dual<0> rgb[3];
        Taps t = ident_taps(x, y);                 // MODE_IDENT: one tap of weight 1 (same code path, no divergent select)
        if (mode != MODE_IDENT) {
            float u, v;
            project(d + D_M2, (int)d[D_GRID2], x, y, S, S, q.ww, q.wh, u, v);
            t = make_taps(u, v, q.ww, q.wh, mode);
        }
        const float fillc = fill_term(t, mode, d);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c].v = sample_plane(an + c * aplane, Wa, t) + fillc;
        if (d[D_JIT] != 0.0) jitter_d<0>(rgb, (float)d[D_SAT], (float)d[D_HUE], d[D_SATFIRST] != 0.0);
        const float nf = (float)d[D_NOISE];
        float* o = out + (size_t)n * 3 * plane + pix;
        float zn[3] = {0.f, 0.f, 0.f};
        if (!noise && nf != 0.f && d[D_SEED] != 0.0) philox_normal3((unsigned long long)d[D_SEED], (unsigned)pix, zn);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = rgb[c].v;
            if (noise) v += nf * noise[(size_t)n * 3 * plane + c * plane + pix];
            else v += nf * zn[c];
            o[c * plane] = v;
        }
    }
}

// Stage B backward, pass 1 (destination-parallel): the gradient w.r.t. the SAMPLED rgb of every output pixel, i.e. the
// incoming gradient pulled back through the ColorJitter (forward-mode duals of kornia's rgb -> hsv -> rgb round trips).
// Cutouts without jitter are skipped (pass 2 reads their incoming gradient directly).
__global__ __launch_bounds__(256) void warp_b_jac_kernel(const float* __restrict__ a, int Ha, int Wa, const double* __restrict__ desc,
                                                         const float* __restrict__ g, float* __restrict__ grgb, int n_cut, int S) {
    const size_t plane = (size_t)S * S, aplane = (size_t)Ha * Wa;
    const int n = blockIdx.y;
    const double* d = desc + (size_t)n * DESC_WORDS;
    if (d[D_JIT] == 0.0) return;
    const int mode = (int)d[D_MODE2];
    const SrcWin q = src_window(d);
    const float* an = a + (size_t)n * 3 * aplane + (size_t)q.oy * Wa + q.ox;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < plane; pix += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(pix % S), y = (int)(pix / S);
        Taps t = ident_taps(x, y);
        if (mode != MODE_IDENT) {
            float u, v;
            project(d + D_M2, (int)d[D_GRID2], x, y, S, S, q.ww, q.wh, u, v);
            t = make_taps(u, v, q.ww, q.wh, mode);
        }
        const float fillc = fill_term(t, mode, d);
        Dual<3> rgb[3];
        float gin[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rgb[c] = cst<3>(sample_plane(an + c * aplane, Wa, t) + fillc);
            rgb[c].d[c] = 1.f;
            gin[c] = g[(size_t)n * 3 * plane + c * plane + pix];
        }
        jitter_d<3>(rgb, (float)d[D_SAT], (float)d[D_HUE], d[D_SATFIRST] != 0.0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            grgb[(size_t)n * 3 * plane + i * plane + pix] = gin[0] * rgb[0].d[i] + gin[1] * rgb[1].d[i] + gin[2] * rgb[2].d[i];
    }
}

// pass 2 (source-parallel gather): ga[n][3][Ha][Wa], every element written (zero outside the stage-B source window)
__global__ __launch_bounds__(256) void warp_b_bwd_kernel(int Ha, int Wa, const double* __restrict__ desc, const float* __restrict__ g,
                                                         const float* __restrict__ grgb, const float2* __restrict__ uv,
                                                         float* __restrict__ ga, int n_cut, int S) {
    __shared__ StageMap sm;
    __shared__ TileStage ts;
    const size_t plane = (size_t)S * S, aplane = (size_t)Ha * Wa;
    const int tiles = (Wa + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int ax0 = (blockIdx.x % tiles) * TILE_W, ay0 = (blockIdx.x / tiles) * TILE_W;
    const int ax = ax0 + (threadIdx.x & 15);      // stage-A image coordinates
    const int ay = ay0 + (threadIdx.x >> 4);
    const bool inimg = ax < Wa && ay < Ha;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE2];
    const SrcWin q = src_window(d);
    const int sx = ax - q.ox, sy = ay - q.oy;                                // coordinates inside the source window
    const bool live = inimg && sx >= 0 && sx < q.ww && sy >= 0 && sy < q.wh;
    const float* gi = (d[D_JIT] != 0.0 ? grgb : g) + (size_t)n * 3 * plane;
    float* go = ga + (size_t)n * 3 * aplane + (size_t)ay * Wa + ax;
    float o[3] = {0.f, 0.f, 0.f};
    if (mode == MODE_IDENT) {      // output pixel (x, y) copies window pixel (x, y)
        if (live && sx < S && sy < S) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = gi[(size_t)c * plane + (size_t)sy * S + sx];
        }
    } else {
        if (threadIdx.x == 0) build_stage_map(sm, d + D_M2, (int)d[D_GRID2], S, S, q.ww, q.wh);
        __syncthreads();
        GatherStage st{d + D_M2, (int)d[D_GRID2], mode, S, S, q.ww, q.wh, gi, uv + (size_t)n * plane};
        gather_tile(st, sm, ts, ax0 - q.ox, ay0 - q.oy, sx, sy, live, o);
    }
    if (inimg) {
#pragma unroll
        for (int c = 0; c < 3; ++c) go[(size_t)c * aplane] = o[c];
    }
}

// pass 2 in the tile-owned scatter form (scatter_tile)
__global__ __launch_bounds__(256) void warp_b_bwd2_kernel(int Ha, int Wa, const double* __restrict__ desc, const float* __restrict__ g,
                                                          const float* __restrict__ grgb, const float2* __restrict__ uv,
                                                          const StageMap* __restrict__ maps, float* __restrict__ ga, int n_cut, int S,
                                                          unsigned long long* dbg) {
    __shared__ TileScatter ts;
    const size_t plane = (size_t)S * S, aplane = (size_t)Ha * Wa;
    const int tiles = (Wa + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int ax0 = (blockIdx.x % tiles) * TILE_W, ay0 = (blockIdx.x / tiles) * TILE_W;
    const int ax = ax0 + (threadIdx.x & 15);
    const int ay = ay0 + (threadIdx.x >> 4);
    const bool inimg = ax < Wa && ay < Ha;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE2];
    const SrcWin q = src_window(d);
    const int sx = ax - q.ox, sy = ay - q.oy;
    const bool live = inimg && sx >= 0 && sx < q.ww && sy >= 0 && sy < q.wh;
    const float* gi = (d[D_JIT] != 0.0 ? grgb : g) + (size_t)n * 3 * plane;
    float* go = ga + (size_t)n * 3 * aplane + (size_t)ay * Wa + ax;
    float o[3] = {0.f, 0.f, 0.f};
    if (mode == MODE_IDENT) {
        if (live && sx < S && sy < S) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = gi[(size_t)c * plane + (size_t)sy * S + sx];
        }
    } else {
        const StageMap sm = maps[n];
        GatherStage st{d + D_M2, (int)d[D_GRID2], mode, S, S, q.ww, q.wh, gi, uv + (size_t)n * plane, dbg ? dbg + 4 : nullptr};
        scatter_tile(st, sm, ts, ax0 - q.ox, ay0 - q.oy, o);
        if (!live) { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; }
    }
    if (inimg) {
#pragma unroll
        for (int c = 0; c < 3; ++c) go[(size_t)c * aplane] = o[c];
    }
}

// one wave per tile (scatter_tile_wave)
__global__ __launch_bounds__(64) void warp_b_bwd3_kernel(int Ha, int Wa, const double* __restrict__ desc, const float* __restrict__ g,
                                                         const float* __restrict__ grgb, const float2* __restrict__ uv,
                                                         const StageMap* __restrict__ maps, float* __restrict__ ga, int n_cut, int S) {
    __shared__ WaveScatter ts;
    const size_t plane = (size_t)S * S, aplane = (size_t)Ha * Wa;
    const int tiles = (Wa + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int ax0 = (blockIdx.x % tiles) * TILE_W, ay0 = (blockIdx.x / tiles) * TILE_W;
    const int lane = threadIdx.x;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE2];
    const SrcWin q = src_window(d);
    const float* gi = (d[D_JIT] != 0.0 ? grgb : g) + (size_t)n * 3 * plane;
    float* go = ga + (size_t)n * 3 * aplane;
    float o[4][3];
    if (mode == MODE_IDENT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = lane + 64 * j, ax = ax0 + (p & 15), ay = ay0 + (p >> 4), sx = ax - q.ox, sy = ay - q.oy;
            const bool live = ax < Wa && ay < Ha && sx >= 0 && sx < q.ww && sy >= 0 && sy < q.wh && sx < S && sy < S;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[j][c] = live ? gi[(size_t)c * plane + (size_t)sy * S + sx] : 0.f;
        }
    } else {
        const StageMap sm = maps[n];
        GatherStage st{d + D_M2, (int)d[D_GRID2], mode, S, S, q.ww, q.wh, gi, uv + (size_t)n * plane};
        scatter_tile_wave(st, sm, ts, ax0 - q.ox, ay0 - q.oy, o);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = lane + 64 * j, ax = ax0 + (p & 15), ay = ay0 + (p >> 4), sx = ax - q.ox, sy = ay - q.oy;
        if (!(ax < Wa && ay < Ha)) continue;
        const bool live = sx >= 0 && sx < q.ww && sy >= 0 && sy < q.wh;
#pragma unroll
        for (int c = 0; c < 3; ++c) go[(size_t)c * aplane + (size_t)ay * Wa + ax] = live ? o[j][c] : 0.f;
    }
}

// F.interpolate(bilinear, align_corners=False) of the pooled image [C,S,S] to the canvas aspect [C,Hb,Wb]
// (kornia.geometry.transform.rescale, pixray.py:468-472) and its gradient
__device__ __forceinline__ void resize_taps(int o, int in, int outn, int& i0, int& i1, float& w1) {
    float src = ((float)o + 0.5f) * ((float)in / (float)outn) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    w1 = src - (float)i0;
}
__global__ __launch_bounds__(256) void rescale_fwd_kernel(const float* __restrict__ p, float* __restrict__ out, int C, int S, int Hb, int Wb) {
    const size_t total = (size_t)C * Hb * Wb;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wb), y = (int)((idx / Wb) % Hb), c = (int)(idx / ((size_t)Hb * Wb));
        int x0, x1, y0, y1; float wx, wy;
        resize_taps(x, S, Wb, x0, x1, wx); resize_taps(y, S, Hb, y0, y1, wy);
        const float* q = p + (size_t)c * S * S;
        out[idx] = (1.f - wy) * ((1.f - wx) * q[y0 * S + x0] + wx * q[y0 * S + x1]) + wy * ((1.f - wx) * q[y1 * S + x0] + wx * q[y1 * S + x1]);
    }
}
// gather form of the rescale backward (one thread per pooled pixel; fixed order -> bit-reproducible): destination columns
// whose two taps include source column sx are those with source coordinate in (sx-1, sx+1), a contiguous run
__device__ __forceinline__ void rescale_run(int s, int in, int outn, int& o0, int& o1) {
    const float sc = (float)outn / (float)in;
    o0 = max((int)floorf(((float)s - 1.f + 0.5f) * sc - 0.5f) - 1, 0);
    o1 = min((int)ceilf(((float)s + 1.f + 0.5f) * sc - 0.5f) + 1, outn - 1);
}
__global__ __launch_bounds__(256) void rescale_bwd_kernel(const float* __restrict__ g, float* __restrict__ gp, int C, int S, int Hb, int Wb) {
    const size_t total = (size_t)C * S * S;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int sx = (int)(idx % S), sy = (int)((idx / S) % S), c = (int)(idx / ((size_t)S * S));
        int xa, xb, ya, yb;
        rescale_run(sx, S, Wb, xa, xb); rescale_run(sy, S, Hb, ya, yb);
        const float* q = g + (size_t)c * Hb * Wb;
        float acc = 0.f;
        for (int y = ya; y <= yb; ++y) {
            int y0, y1; float wy;
            resize_taps(y, S, Hb, y0, y1, wy);
            const float cy = (y0 == sy ? (1.f - wy) : 0.f) + (y1 == sy ? wy : 0.f);
            if (cy == 0.f) continue;
            for (int x = xa; x <= xb; ++x) {
                int x0, x1; float wx;
                resize_taps(x, S, Wb, x0, x1, wx);
                // the forward's weights: (1-wy)(1-wx), (1-wy)wx, wy(1-wx), wy wx on (y0,x0), (y0,x1), (y1,x0), (y1,x1)
                float w = 0.f;
                if (y0 == sy && x0 == sx) w += (1.f - wy) * (1.f - wx);
                if (y0 == sy && x1 == sx) w += (1.f - wy) * wx;
                if (y1 == sy && x0 == sx) w += wy * (1.f - wx);
                if (y1 == sy && x1 == sx) w += wy * wx;
                if (w != 0.f) acc += q[(size_t)y * Wb + x] * w;
            }
        }
        gp[idx] = acc;
    }
}

// ------------------------------------------------------------------ batch min/max + CLIP normalise + patchify
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, size_t n,
                                                             float* __restrict__ part) {
    float mn = INFINITY, mx = -INFINITY;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)x) & 15) == 0) {           // 16-byte loads, two in flight per lane (one pass over 38.5 MB at the headline: HBM-bound)
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const size_t n4 = n >> 2;
        size_t i = tid;
        for (; i + nthr < n4; i += 2 * nthr) {
            const float4 a = x4[i], b = x4[i + nthr];
            mn = fminf(fminf(mn, fminf(a.x, a.y)), fminf(fminf(a.z, a.w), fminf(fminf(b.x, b.y), fminf(b.z, b.w))));
            mx = fmaxf(fmaxf(mx, fmaxf(a.x, a.y)), fmaxf(fmaxf(a.z, a.w), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
        }
        for (; i < n4; i += nthr) {
            const float4 a = x4[i];
            mn = fminf(fminf(mn, fminf(a.x, a.y)), fminf(a.z, a.w));
            mx = fmaxf(fmaxf(mx, fmaxf(a.x, a.y)), fmaxf(a.z, a.w));
        }
        for (size_t j = (n4 << 2) + tid; j < n; j += nthr) { const float v = x[j]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    } else {
        for (size_t i = tid; i < n; i += nthr) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    mn = wave_min(mn); mx = wave_max(mx);
    __shared__ float s[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[w] = mn; s[4 + w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
        part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
    }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, int nparts,
                                                           float* __restrict__ mm) {
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    __shared__ float s[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[w] = mn; s[4 + w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[0] = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
        mm[1] = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
    }
}

__constant__ float c_clip_mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
__constant__ float c_clip_std[3] = {0.26862954f, 0.26130258f, 0.27577711f};

// A[(n*T + 1 + gy*G + gx)][c*P*P + py*P + px] = ((x - mn)/(mx - mn) - mean_c)/std_c ; row n*T is zero (class token).
// Kp = 3*P*P rounded up to a multiple of 8 (zero columns): ViT-L/14 has 588 -> 592.
// TOp: operand precision of A (bf16, or fp32 in the exact mode)
template <typename TOp>
__global__ __launch_bounds__(256) void patchify_fwd_kernel(const float* __restrict__ cut, const float* __restrict__ mm,
                                                           TOp* __restrict__ A, int N, int S, int P, int T, int Kp) {
    const int G = S / P;
    const int K = 3 * P * P;
    const int K8 = Kp / 8;
    const size_t total = (size_t)N * T * K8;
    const float mn = mm[0];
    const float range = mm[1] - mm[0];
    const float inv = range != 0.f ? 1.f / range : 1.f;
    const bool fast = (P % 8) == 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k8 = (int)(idx % K8);
        const size_t row = idx / K8;
        const int tok = (int)(row % T), n = (int)(row / T);
        float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (tok > 0) {
            const int gy = (tok - 1) / G, gx = (tok - 1) % G;
            if (fast) {
                const int k = k8 * 8;
                const int c = k / (P * P), py = (k / P) % P, px = k % P;
                const float* src = cut + (((size_t)n * 3 + c) * S + gy * P + py) * S + gx * P + px;
                const float im = c_clip_mean[c], is = 1.f / c_clip_std[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = (((src[e] - mn) * inv) - im) * is;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k8 * 8 + e;
                    if (k < K) {
                        const int c = k / (P * P), py = (k / P) % P, px = k % P;
                        const float v = cut[(((size_t)n * 3 + c) * S + gy * P + py) * S + gx * P + px];
                        r[e] = (((v - mn) * inv) - c_clip_mean[c]) / c_clip_std[c];
                    }
                }
            }
        }
        op_st4(A, idx * 8, r[0], r[1], r[2], r[3]);
        op_st4(A, idx * 8 + 4, r[4], r[5], r[6], r[7]);
    }
}

// Reduction for the min/max renorm backward: acc = {sum g_y, sum g_y*y, count(x==min), count(x==max)} (double)
__global__ __launch_bounds__(256) void patchify_bwd_reduce_kernel(const float* __restrict__ cut,
                                                                  const float* __restrict__ mm,
                                                                  const float* __restrict__ dA, double* __restrict__ acc,
                                                                  int N, int S, int P, int T, int K) {
    const int G = S / P;
    const size_t total = (size_t)N * 3 * S * S;
    const float mn = mm[0], mx = mm[1];
    const float range = mx - mn;
    const float inv = range != 0.f ? 1.f / range : 1.f;
    double s1 = 0, s2 = 0, cmin = 0, cmax = 0;
    // 32-bit index arithmetic (the host checks total < 2^31): the size_t div/mod chain of the first version cost more than
    // the memory traffic
    const unsigned S2 = (unsigned)S * S, total32 = (unsigned)total, stride = gridDim.x * blockDim.x;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total32; idx += stride) {
        const unsigned pl = idx / S2, rem = idx - pl * S2;
        const int y = (int)(rem / (unsigned)S), x = (int)(rem - (unsigned)y * S);
        const int n = (int)(pl / 3u), c = (int)(pl - 3u * n);
        const int ty = y / P, tx = x / P;
        const int tok = 1 + ty * G + tx;
        const int k = c * P * P + (y - ty * P) * P + (x - tx * P);
        const float gy = dA[((size_t)n * T + tok) * K + k] / c_clip_std[c];
        const float xv = cut[idx];
        s1 += gy; s2 += gy * ((xv - mn) * inv);
        cmin += (xv == mn); cmax += (xv == mx);
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); cmin = wave_sum_d(cmin); cmax = wave_sum_d(cmax);
    __shared__ double red[4][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = s1; red[wv][1] = s2; red[wv][2] = cmin; red[wv][3] = cmax; }
    __syncthreads();
    if (threadIdx.x < 4)
        atomicAdd(&acc[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// g_cut = g_y/range + [x==min]*gmin/cnt_min + [x==max]*gmax/cnt_max   (slip.py:21-36 backward)
__global__ __launch_bounds__(256) void patchify_bwd_apply_kernel(const float* __restrict__ cut,
                                                                 const float* __restrict__ mm,
                                                                 const float* __restrict__ dA,
                                                                 const double* __restrict__ acc, float* __restrict__ gcut,
                                                                 int N, int S, int P, int T, int K) {
    const int G = S / P;
    const size_t total = (size_t)N * 3 * S * S;
    const float mn = mm[0], mx = mm[1];
    const float range = mx - mn;
    const bool live = range != 0.f;
    const float inv = live ? 1.f / range : 1.f;
    const float gmin = live ? (float)((acc[1] - acc[0]) * inv / fmax(acc[2], 1.0)) : 0.f;
    const float gmax = live ? (float)(-acc[1] * inv / fmax(acc[3], 1.0)) : 0.f;
    const unsigned S2 = (unsigned)S * S, total32 = (unsigned)total, stride = gridDim.x * blockDim.x;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total32; idx += stride) {
        const unsigned pl = idx / S2, rem = idx - pl * S2;
        const int y = (int)(rem / (unsigned)S), x = (int)(rem - (unsigned)y * S);
        const int n = (int)(pl / 3u), c = (int)(pl - 3u * n);
        const int ty = y / P, tx = x / P;
        const int tok = 1 + ty * G + tx;
        const int k = c * P * P + (y - ty * P) * P + (x - tx * P);
        const float gy = dA[((size_t)n * T + tok) * K + k] / c_clip_std[c];
        const float xv = cut[idx];
        float g = gy * inv;
        if (xv == mn) g += gmin;
        if (xv == mx) g += gmax;
        gcut[idx] = g;
    }
}

}  // namespace

// PRX_CUTOUT_BWD=gather: the round-2 per-pixel gather kernels instead of the tile-owned scatter (A/B measurements)
static bool cutout_bwd_gather() {
    static const bool v = [] { const char* e = getenv("PRX_CUTOUT_BWD"); return e && e[0] == 'g'; }();
    return v;
}

// PRX_CUTOUT_DBG=1: count the scatter's work (candidates visited / inside their rectangle / contributing) per stage and print
// the totals when the process exits (diagnostic; adds global atomics, so do not time with it on)
static unsigned long long* cutout_dbg() {
    static unsigned long long* buf = [] () -> unsigned long long* {
        const char* e = getenv("PRX_CUTOUT_DBG");
        if (!(e && e[0] == '1')) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, 8 * sizeof(unsigned long long)) != hipSuccess || hipMemset(p, 0, 8 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        static unsigned long long* keep = (unsigned long long*)p;
        atexit([] {
            unsigned long long h[8];
            if (hipMemcpy(h, keep, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[prx cutout scatter] stage A: visited %llu rects %llu in-rect %llu contributing %llu | stage B: visited %llu rects %llu in-rect %llu contributing %llu\n",
                        h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        });
        return keep;
    }();
    return buf;
}

// The scatter runs one tile per 256-thread workgroup (per-wave accumulator planes) by default; PRX_CUTOUT_BWD=wave selects the
// one-wave-per-tile form.  Measured at the headline (profiles/r03_cfg1_kernel_stats.csv, r03b): stage B / stage A
// 187 / 170 us (workgroup), 209 / 184 us (wave), 150 / 255 us (round-2 gather): neither more tiles in flight nor the removed
// barriers and serial set-up moved it -- what is left is the ~12 conflicting ds_add_f32 per candidate (4 taps x 3 channels,
// neighbouring destination pixels share taps), i.e. the LDS atomic rate.
static bool cutout_bwd_block() {
    static const bool v = [] { const char* e = getenv("PRX_CUTOUT_BWD"); return !(e && e[0] == 'w'); }();
    return v;
}

int prx_pool_fwd(const float* img, float* pooled, int* argmax, const unsigned char* mask, int C, int H, int W, int S, hipStream_t s) {
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(ew_grid((size_t)C * S * S)), dim3(256), 0, s, img, pooled, argmax, mask, C, H, W, S);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_pool_bwd(const float* g, const int* argmax, const unsigned char* mask, float* gimg, int C, int H, int W, int S, hipStream_t s) {
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_grid((size_t)C * H * W)), dim3(256), 0, s, g, argmax, mask, gimg, C, H, W, S);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_warp_a_fwd(const float* src, int Hs, int Ws, const double* desc, float* out, int n_cut, int Ha, int Wa, hipStream_t s) {
    hipLaunchKernelGGL(warp_a_fwd_kernel, dim3(ew_grid((size_t)n_cut * Ha * Wa)), dim3(256), 0, s, src, Hs, Ws, desc, out,
                       n_cut, Ha, Wa);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_warp_a_bwd(const float* g, int Hs, int Ws, const double* desc, float* uv, float* gsrc_priv, float* gsrc, int n_cut, int Ha,
                   int Wa, hipStream_t s) {
    // uv: [n_cut][Ha*Wa][2] scratch
    // gsrc_priv: [n_cut][3][Hs][Ws] per-cutout private planes (scratch, every element written); gsrc: [3][Hs][Ws] their sum
    const int tx = (Ws + TILE_W - 1) / TILE_W, ty = (Hs + TILE_W - 1) / TILE_W;
    hipLaunchKernelGGL(uv_kernel, dim3(std::min(ew_grid((size_t)Ha * Wa), 64), n_cut), dim3(256), 0, s, desc, 1, (float2*)uv, Wa, Ha, Ws, Hs);
    PRX_LAUNCH_CHECK();
    if (cutout_bwd_gather())
        hipLaunchKernelGGL(warp_a_bwd_kernel, dim3(tx * ty, n_cut), dim3(256), 0, s, g, Hs, Ws, desc, (const float2*)uv, gsrc_priv, n_cut,
                           Ha, Wa);
    else {
        // the per-cutout stage maps (16 floats each) live in `gsrc` until reduce_planes_kernel overwrites it with the result
        PRX_REQUIRE((size_t)n_cut * sizeof(StageMap) <= (size_t)3 * Hs * Ws * sizeof(float), "warp_a_bwd: too many cutouts for the stage-map scratch");
        hipLaunchKernelGGL(stage_map_kernel, dim3(ceil_div(n_cut, 64)), dim3(64), 0, s, desc, 1, (StageMap*)gsrc, n_cut, Wa, Ha, Ws, Hs);
        PRX_LAUNCH_CHECK();
        if (cutout_bwd_block())
            hipLaunchKernelGGL(warp_a_bwd2_kernel, dim3(tx * ty, n_cut), dim3(256), 0, s, g, Hs, Ws, desc, (const float2*)uv, (const StageMap*)gsrc,
                               gsrc_priv, n_cut, Ha, Wa, cutout_dbg());
        else
            hipLaunchKernelGGL(warp_a_bwd3_kernel, dim3(tx * ty, n_cut), dim3(64), 0, s, g, Hs, Ws, desc, (const float2*)uv, (const StageMap*)gsrc,
                               gsrc_priv, n_cut, Ha, Wa);
    }
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_planes_kernel, dim3(ew_grid((size_t)3 * Hs * Ws)), dim3(256), 0, s, gsrc_priv, gsrc, n_cut,
                       (size_t)3 * Hs * Ws);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_warp_b_fwd(const float* a, int Ha, int Wa, const double* desc, const float* noise, float* out, int n_cut, int S,
                   hipStream_t s) {
    hipLaunchKernelGGL(warp_b_fwd_kernel, dim3(ew_grid((size_t)n_cut * S * S)), dim3(256), 0, s, a, Ha, Wa, desc, noise, out,
                       n_cut, S);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_warp_b_bwd(const float* a, int Ha, int Wa, const double* desc, const float* g, float* grgb, float* uv, float* ga, int n_cut,
                   int S, hipStream_t s, float* maps_scratch, size_t maps_scratch_bytes) {
    // grgb: [n_cut][3][S][S] scratch (the gradient pulled back through the ColorJitter); uv: [n_cut][S*S][2] scratch;
    // ga: [n_cut][3][Ha][Wa], every element written
    hipLaunchKernelGGL(uv_kernel, dim3(std::min(ew_grid((size_t)S * S), 64), n_cut), dim3(256), 0, s, desc, 2, (float2*)uv, S, S, 0, 0);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(warp_b_jac_kernel, dim3(std::min(ew_grid((size_t)S * S), 64), n_cut), dim3(256), 0, s, a, Ha, Wa, desc, g, grgb,
                       n_cut, S);
    PRX_LAUNCH_CHECK();
    const int tx = (Wa + TILE_W - 1) / TILE_W, ty = (Ha + TILE_W - 1) / TILE_W;
    if (cutout_bwd_gather())
        hipLaunchKernelGGL(warp_b_bwd_kernel, dim3(tx * ty, n_cut), dim3(256), 0, s, Ha, Wa, desc, g, grgb, (const float2*)uv, ga, n_cut, S);
    else {
        // maps_scratch: any buffer of >= n_cut stage maps that nothing else touches until this launch has finished
        PRX_REQUIRE(maps_scratch != nullptr && (size_t)n_cut * sizeof(StageMap) <= maps_scratch_bytes, "warp_b_bwd: stage-map scratch too small");
        hipLaunchKernelGGL(stage_map_kernel, dim3(ceil_div(n_cut, 64)), dim3(64), 0, s, desc, 2, (StageMap*)maps_scratch, n_cut, S, S, 0, 0);
        PRX_LAUNCH_CHECK();
        if (cutout_bwd_block())
            hipLaunchKernelGGL(warp_b_bwd2_kernel, dim3(tx * ty, n_cut), dim3(256), 0, s, Ha, Wa, desc, g, grgb, (const float2*)uv,
                               (const StageMap*)maps_scratch, ga, n_cut, S, cutout_dbg());
        else
            hipLaunchKernelGGL(warp_b_bwd3_kernel, dim3(tx * ty, n_cut), dim3(64), 0, s, Ha, Wa, desc, g, grgb, (const float2*)uv,
                               (const StageMap*)maps_scratch, ga, n_cut, S);
    }
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_rescale_fwd(const float* pooled, float* base, int C, int S, int Hb, int Wb, hipStream_t s) {
    hipLaunchKernelGGL(rescale_fwd_kernel, dim3(ew_grid((size_t)C * Hb * Wb)), dim3(256), 0, s, pooled, base, C, S, Hb, Wb);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_rescale_bwd(const float* g_base, float* g_pooled, int C, int S, int Hb, int Wb, hipStream_t s) {
    hipLaunchKernelGGL(rescale_bwd_kernel, dim3(ew_grid((size_t)C * S * S)), dim3(256), 0, s, g_base, g_pooled, C, S, Hb, Wb);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_minmax(const float* x, size_t n, float* part, int nparts, float* mm, hipStream_t s) {
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts), dim3(256), 0, s, x, n, part);
    PRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, s, part, nparts, mm);
    PRX_LAUNCH_CHECK();
    return 0;
}
static inline int patch_kp(int P) { return (3 * P * P + 7) / 8 * 8; }

int prx_patchify_fwd(const float* cut, const float* mm, void* A, int prec, int N, int S, int P, int T, hipStream_t s) {
    PRX_REQUIRE(S % P == 0 && T == (S / P) * (S / P) + 1, "patchify: bad geometry S=%d P=%d T=%d", S, P, T);
    const int Kp = patch_kp(P);
    const dim3 grid(ew_grid((size_t)N * T * Kp / 8));
    PRX_OP_DISPATCH(prec_is_f32(prec), prec_is_h16(prec), TO,
                    hipLaunchKernelGGL(patchify_fwd_kernel<TO>, grid, dim3(256), 0, s, cut, mm, (TO*)A, N, S, P, T, Kp));
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_patchify_bwd_reduce(const float* cut, const float* mm, const float* dA, double* acc, int N, int S, int P, int T,
                            hipStream_t s) {
    PRX_REQUIRE((size_t)N * 3 * S * S < ((size_t)1 << 31), "patchify backward: %d cutouts of %dx%d exceed the 32-bit index range", N, S, S);
    PRX_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 4, s));
    hipLaunchKernelGGL(patchify_bwd_reduce_kernel, dim3(std::min(ew_grid((size_t)N * 3 * S * S), 2048)), dim3(256), 0, s,
                       cut, mm, dA, acc, N, S, P, T, patch_kp(P));
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_patchify_bwd_apply(const float* cut, const float* mm, const float* dA, const double* acc, float* gcut, int N,
                           int S, int P, int T, hipStream_t s) {
    hipLaunchKernelGGL(patchify_bwd_apply_kernel, dim3(ew_grid((size_t)N * 3 * S * S)), dim3(256), 0, s, cut, mm, dA, acc,
                       gcut, N, S, P, T, patch_kp(P));
    PRX_LAUNCH_CHECK();
    return 0;
}

// The same two passes for a gradient held in IMAGE layout dY[N][3][S][S] (CLIP ModifiedResNet: the stem convolution
// consumes the normalised image directly): the patch kernels with one "patch" = the whole image (P = S, one token per
// image); their token index is 1-based, hence the pointer shifted back by one row of K = 3*S*S.
int prx_preproc_bwd_reduce(const float* cut, const float* mm, const float* dY, double* acc, int N, int S, hipStream_t s) {
    PRX_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 4, s));
    PRX_REQUIRE((size_t)N * 3 * S * S < ((size_t)1 << 31), "preprocessing backward: %d cutouts of %dx%d exceed the 32-bit index range", N, S, S);
    const int K = 3 * S * S;
    hipLaunchKernelGGL(patchify_bwd_reduce_kernel, dim3(std::min(ew_grid((size_t)N * 3 * S * S), 2048)), dim3(256), 0, s,
                       cut, mm, dY - K, acc, N, S, S, 1, K);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_preproc_bwd_apply(const float* cut, const float* mm, const float* dY, const double* acc, float* gcut, int N, int S,
                          hipStream_t s) {
    const int K = 3 * S * S;
    hipLaunchKernelGGL(patchify_bwd_apply_kernel, dim3(ew_grid((size_t)N * 3 * S * S)), dim3(256), 0, s, cut, mm, dY - K, acc,
                       gcut, N, S, S, 1, K);
    PRX_LAUNCH_CHECK();
    return 0;
}
