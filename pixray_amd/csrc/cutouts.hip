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
// Backward scatters with fp32 atomics (as the reference's grid_sampler_2d_backward does,
// pixray.py:29); the ColorJitter Jacobian is obtained with forward-mode duals.
#include "cutouts.h"
#include <algorithm>

namespace {

constexpr int DESC_WORDS = 32;
// descriptor word offsets (all stored as fp64; the two 3x3 matrices map NORMALISED destination coordinates to
// NORMALISED source coordinates exactly as kornia builds them, so sampling positions round like the oracle's)
enum { D_M1 = 0, D_M2 = 9, D_MODE1 = 18, D_MODE2 = 19, D_FILL = 20, D_JIT = 21, D_SAT = 22, D_HUE = 23,
       D_SATFIRST = 24, D_NOISE = 25, D_GRID1 = 26, D_GRID2 = 27,
       D_WOX = 28, D_WOY = 29, D_WW = 30, D_WH = 31 };   // stage-B source window inside the stage-A image (x, y, width, height)
enum { GRID_MESH = 0, GRID_AFFINE = 1 };
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

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ g, const int* __restrict__ argmax,
                                                       const unsigned char* __restrict__ mask, float* __restrict__ gimg, int C, int H,
                                                       int W, int S) {
    const int total = C * S * S;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int x = idx % S, y = (idx / S) % S, c = idx / (S * S);
        const int y0 = win_start(y, H, S), y1 = win_end(y, H, S);
        const int x0 = win_start(x, W, S), x1 = win_end(x, W, S);
        if (mask && mask[idx]) continue;            // masked pooled pixels are constants
        const float gv = 0.5f * g[idx];
        const float ga = gv / (float)((y1 - y0) * (x1 - x0));
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) atomicAdd(&gimg[((size_t)c * H + yy) * W + xx], ga);
        atomicAdd(&gimg[(size_t)c * H * W + argmax[idx]], gv);
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

// Source sampling position (F.grid_sample pixel coordinates, align_corners=False) of destination pixel
// (x, y).  GRID_MESH: kornia create_meshgrid + transform_points (fp64) -> fp32 grid (warp_perspective);
// GRID_AFFINE: F.affine_grid's fp32 base grid times the fp32 theta (warp_affine).  The fp32 grid value is
// then unnormalised the way ATen does: (g + 1) * (size / 2) - 0.5.
__device__ __forceinline__ void project(const double* m, int gtype, int x, int y, int Wd, int Hd, int Ws, int Hs,
                                        float& u, float& v) {
    float gx, gy;
    if (gtype == GRID_MESH) {
        const double xn = ((double)x / (double)(Wd - 1) - 0.5) * 2.0;
        const double yn = ((double)y / (double)(Hd - 1) - 0.5) * 2.0;
        const double X = m[0] * xn + m[1] * yn + m[2];
        const double Y = m[3] * xn + m[4] * yn + m[5];
        const double Z = m[6] * xn + m[7] * yn + m[8];
        const double sc = (fabs(Z) > 1e-8) ? 1.0 / Z : 1.0;
        gx = (float)(X * sc); gy = (float)(Y * sc);
    } else {
        const float xb = (linspace_pm1(x, Wd) * (float)(Wd - 1)) / (float)Wd;
        const float yb = (linspace_pm1(y, Hd) * (float)(Hd - 1)) / (float)Hd;
        gx = (float)((double)xb * m[0] + (double)yb * m[1] + m[2]);
        gy = (float)((double)xb * m[3] + (double)yb * m[4] + m[5]);
    }
    u = (gx + 1.f) * ((float)Ws * 0.5f) - 0.5f;
    v = (gy + 1.f) * ((float)Hs * 0.5f) - 0.5f;
}

// sample one channel plane; returns value and the coverage (sum of in-bounds weights)
__device__ __forceinline__ float sample_plane(const float* __restrict__ p, int W, const Taps& t) {
    float v00 = (t.vx0 && t.vy0) ? p[t.y0 * W + t.x0] : 0.f;
    float v01 = (t.vx1 && t.vy0) ? p[t.y0 * W + t.x0 + 1] : 0.f;
    float v10 = (t.vx0 && t.vy1) ? p[(t.y0 + 1) * W + t.x0] : 0.f;
    float v11 = (t.vx1 && t.vy1) ? p[(t.y0 + 1) * W + t.x0 + 1] : 0.f;
    return v00 * (1.f - t.wx) * (1.f - t.wy) + v01 * t.wx * (1.f - t.wy) + v10 * (1.f - t.wx) * t.wy + v11 * t.wx * t.wy;
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
__device__ __forceinline__ void scatter_plane(float* __restrict__ p, int W, const Taps& t, float g) {
    if (t.vx0 && t.vy0) atomicAdd(&p[t.y0 * W + t.x0], g * (1.f - t.wx) * (1.f - t.wy));
    if (t.vx1 && t.vy0) atomicAdd(&p[t.y0 * W + t.x0 + 1], g * t.wx * (1.f - t.wy));
    if (t.vx0 && t.vy1) atomicAdd(&p[(t.y0 + 1) * W + t.x0], g * (1.f - t.wx) * t.wy);
    if (t.vx1 && t.vy1) atomicAdd(&p[(t.y0 + 1) * W + t.x0 + 1], g * t.wx * t.wy);
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
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
template <int ND> __device__ __forceinline__ Dual<ND> addc(const Dual<ND>& a, float c) { Dual<ND> r = a; r.v += c; return r; }
template <int ND> __device__ __forceinline__ Dual<ND> divc(const Dual<ND>& a, float c) {
    Dual<ND> r; r.v = a.v / c;
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] / c;
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

// ---- backward scatter with LDS tile staging -----------------------------------------------------------------
// A block owns a 16x16 tile of destination pixels of ONE cutout.  Its bilinear taps land in a compact window of
// the source plane, so they are accumulated with ds_add_f32 into an LDS window (WIN x WIN x 3) anchored at the
// block's minimum tap, and the window is flushed once to HBM (one atomic per touched source pixel instead of
// four per destination pixel).  Taps that fall outside the window (reflection wrap-around, extreme
// perspective) go straight to HBM atomics, so the result never depends on the window size.
constexpr int TILE_W = 16;
constexpr int WIN = 48;

struct ScatterWin {
    float acc[3][WIN][WIN];
    int org[2];
};

__device__ __forceinline__ void win_begin(ScatterWin& w, bool has_taps, const Taps& t) {
    if (threadIdx.x == 0) { w.org[0] = 0x7fffffff; w.org[1] = 0x7fffffff; }
    for (int i = threadIdx.x; i < 3 * WIN * WIN; i += 256) (&w.acc[0][0][0])[i] = 0.f;
    __syncthreads();
    if (has_taps) {
        atomicMin(&w.org[0], max(t.x0, 0));
        atomicMin(&w.org[1], max(t.y0, 0));
    }
    __syncthreads();
}
__device__ __forceinline__ void win_add(ScatterWin& w, float* __restrict__ plane, int W, int c, int x, int y, float v) {
    const int lx = x - w.org[0], ly = y - w.org[1];
    if (lx >= 0 && lx < WIN && ly >= 0 && ly < WIN) atomicAdd(&w.acc[c][ly][lx], v);
    else atomicAdd(&plane[y * W + x], v);
}
__device__ __forceinline__ void win_scatter(ScatterWin& w, float* __restrict__ plane, int W, int c, const Taps& t, float g) {
    if (t.vx0 && t.vy0) win_add(w, plane, W, c, t.x0, t.y0, g * (1.f - t.wx) * (1.f - t.wy));
    if (t.vx1 && t.vy0) win_add(w, plane, W, c, t.x0 + 1, t.y0, g * t.wx * (1.f - t.wy));
    if (t.vx0 && t.vy1) win_add(w, plane, W, c, t.x0, t.y0 + 1, g * (1.f - t.wx) * t.wy);
    if (t.vx1 && t.vy1) win_add(w, plane, W, c, t.x0 + 1, t.y0 + 1, g * t.wx * t.wy);
}
// flush the window into the 3 planes at `base` (a Hs x Ws image with row pitch `pitch` and plane stride `pstride`)
__device__ __forceinline__ void win_flush(ScatterWin& w, float* __restrict__ base, int Hs, int Ws, int pitch, size_t pstride) {
    __syncthreads();
    const int ox = w.org[0], oy = w.org[1];
    if (ox == 0x7fffffff) return;
    for (int i = threadIdx.x; i < 3 * WIN * WIN; i += 256) {
        const float v = (&w.acc[0][0][0])[i];
        if (v == 0.f) continue;
        const int c = i / (WIN * WIN), r = (i / WIN) % WIN, q = i % WIN;
        const int x = ox + q, y = oy + r;
        if (x < Ws && y < Hs) atomicAdd(&base[(size_t)c * pstride + (size_t)y * pitch + x], v);
    }
}

// Stage A backward: g[n][3][Ha][Wa] -> per-cutout private source-gradient planes gsrc[n][3][Hs][Ws]
// (summed over n afterwards by reduce_planes_kernel: no cross-cutout atomic contention on the shared image)
__global__ __launch_bounds__(256) void warp_a_bwd_kernel(const float* __restrict__ g, int Hs, int Ws,
                                                         const double* __restrict__ desc, float* __restrict__ gsrc,
                                                         int n_cut, int Ha, int Wa) {
    __shared__ ScatterWin w;
    const int tiles = (Wa + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int x = (blockIdx.x % tiles) * TILE_W + (threadIdx.x & 15);
    const int y = (blockIdx.x / tiles) * TILE_W + (threadIdx.x >> 4);
    const bool valid = x < Wa && y < Ha;
    const size_t plane = (size_t)Ha * Wa;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE1];
    float* gs = gsrc + (size_t)n * 3 * Hs * Ws;
    const float* gi = g + ((size_t)n * 3) * plane + (size_t)y * Wa + x;
    if (mode == MODE_IDENT) {      // 1:1 copy: each source pixel has exactly one writer
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gs[(size_t)c * Hs * Ws + (size_t)y * Ws + x] = gi[(size_t)c * plane];
        }
        return;
    }
    Taps t{};
    if (valid) {
        float u, v;
        project(d + D_M1, (int)d[D_GRID1], x, y, Wa, Ha, Ws, Hs, u, v);
        t = make_taps(u, v, Ws, Hs, mode);
    }
    win_begin(w, valid && (t.vx0 || t.vx1) && (t.vy0 || t.vy1), t);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) win_scatter(w, gs + (size_t)c * Hs * Ws, Ws, c, t, gi[(size_t)c * plane]);
    }
    win_flush(w, gs, Hs, Ws, Ws, (size_t)Hs * Ws);
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
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = rgb[c].v;
            if (noise) v += nf * noise[(size_t)n * 3 * plane + c * plane + pix];
            o[c * plane] = v;
        }
    }
}

__global__ __launch_bounds__(256) void warp_b_bwd_kernel(const float* __restrict__ a, int Ha, int Wa, const double* __restrict__ desc,
                                                         const float* __restrict__ g, float* __restrict__ ga, int n_cut,
                                                         int S) {
    __shared__ ScatterWin w;
    const size_t plane = (size_t)S * S, aplane = (size_t)Ha * Wa;
    const int tiles = (S + TILE_W - 1) / TILE_W;
    const int n = blockIdx.y;
    const int x = (blockIdx.x % tiles) * TILE_W + (threadIdx.x & 15);
    const int y = (blockIdx.x / tiles) * TILE_W + (threadIdx.x >> 4);
    const bool valid = x < S && y < S;
    const double* d = desc + (size_t)n * DESC_WORDS;
    const int mode = (int)d[D_MODE2];
    const SrcWin q = src_window(d);
    const float* an = a + (size_t)n * 3 * aplane + (size_t)q.oy * Wa + q.ox;
    float* gan = ga + (size_t)n * 3 * aplane + (size_t)q.oy * Wa + q.ox;
    const size_t pix = (size_t)y * S + x;
    Taps t = ident_taps(valid ? x : 0, valid ? y : 0);
    float grgb[3] = {0.f, 0.f, 0.f};
    if (valid) {
        float gin[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gin[c] = g[(size_t)n * 3 * plane + c * plane + pix];
        grgb[0] = gin[0]; grgb[1] = gin[1]; grgb[2] = gin[2];
        if (mode != MODE_IDENT) {
            float u, v;
            project(d + D_M2, (int)d[D_GRID2], x, y, S, S, q.ww, q.wh, u, v);
            t = make_taps(u, v, q.ww, q.wh, mode);
        }
        if (d[D_JIT] != 0.0) {
            Dual<3> rgb[3];
            const float fillc = fill_term(t, mode, d);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rgb[c] = cst<3>(sample_plane(an + c * aplane, Wa, t) + fillc);
                rgb[c].d[c] = 1.f;
            }
            jitter_d<3>(rgb, (float)d[D_SAT], (float)d[D_HUE], d[D_SATFIRST] != 0.0);
#pragma unroll
            for (int i = 0; i < 3; ++i) grgb[i] = gin[0] * rgb[0].d[i] + gin[1] * rgb[1].d[i] + gin[2] * rgb[2].d[i];
        }
    }
    if (mode == MODE_IDENT) {      // 1:1: one writer per source pixel (the rest of the stage-A gradient stays zero)
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gan[c * aplane + (size_t)y * Wa + x] = grgb[c];
        }
        return;
    }
    win_begin(w, valid && (t.vx0 || t.vx1) && (t.vy0 || t.vy1), t);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) win_scatter(w, gan + c * aplane, Wa, c, t, grgb[c]);
    }
    win_flush(w, gan, q.wh, q.ww, Wa, aplane);
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
__global__ __launch_bounds__(256) void rescale_bwd_kernel(const float* __restrict__ g, float* __restrict__ gp, int C, int S, int Hb, int Wb) {
    const size_t total = (size_t)C * Hb * Wb;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wb), y = (int)((idx / Wb) % Hb), c = (int)(idx / ((size_t)Hb * Wb));
        int x0, x1, y0, y1; float wx, wy;
        resize_taps(x, S, Wb, x0, x1, wx); resize_taps(y, S, Hb, y0, y1, wy);
        float* q = gp + (size_t)c * S * S;
        const float v = g[idx];
        atomicAdd(&q[y0 * S + x0], v * (1.f - wy) * (1.f - wx)); atomicAdd(&q[y0 * S + x1], v * (1.f - wy) * wx);
        atomicAdd(&q[y1 * S + x0], v * wy * (1.f - wx)); atomicAdd(&q[y1 * S + x1], v * wy * wx);
    }
}

// ------------------------------------------------------------------ batch min/max + CLIP normalise + patchify
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, size_t n,
                                                             float* __restrict__ part) {
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
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

int prx_pool_fwd(const float* img, float* pooled, int* argmax, const unsigned char* mask, int C, int H, int W, int S, hipStream_t s) {
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(ew_grid((size_t)C * S * S)), dim3(256), 0, s, img, pooled, argmax, mask, C, H, W, S);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_pool_bwd(const float* g, const int* argmax, const unsigned char* mask, float* gimg, int C, int H, int W, int S, hipStream_t s) {
    PRX_CHECK_HIP(hipMemsetAsync(gimg, 0, sizeof(float) * C * H * W, s));
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_grid((size_t)C * S * S)), dim3(256), 0, s, g, argmax, mask, gimg, C, H, W, S);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_warp_a_fwd(const float* src, int Hs, int Ws, const double* desc, float* out, int n_cut, int Ha, int Wa, hipStream_t s) {
    hipLaunchKernelGGL(warp_a_fwd_kernel, dim3(ew_grid((size_t)n_cut * Ha * Wa)), dim3(256), 0, s, src, Hs, Ws, desc, out,
                       n_cut, Ha, Wa);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_warp_a_bwd(const float* g, int Hs, int Ws, const double* desc, float* gsrc_priv, float* gsrc, int n_cut, int Ha, int Wa,
                   hipStream_t s) {
    // gsrc_priv: [n_cut][3][Hs][Ws] per-cutout private planes (scratch); gsrc: [3][Hs][Ws] their sum
    PRX_CHECK_HIP(hipMemsetAsync(gsrc_priv, 0, sizeof(float) * (size_t)n_cut * 3 * Hs * Ws, s));
    const int tx = (Wa + TILE_W - 1) / TILE_W, ty = (Ha + TILE_W - 1) / TILE_W;
    hipLaunchKernelGGL(warp_a_bwd_kernel, dim3(tx * ty, n_cut), dim3(256), 0, s, g, Hs, Ws, desc, gsrc_priv, n_cut, Ha, Wa);
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
int prx_warp_b_bwd(const float* a, int Ha, int Wa, const double* desc, const float* g, float* ga, int n_cut, int S, hipStream_t s) {
    PRX_CHECK_HIP(hipMemsetAsync(ga, 0, sizeof(float) * (size_t)n_cut * 3 * Ha * Wa, s));
    const int tiles = (S + TILE_W - 1) / TILE_W;
    hipLaunchKernelGGL(warp_b_bwd_kernel, dim3(tiles * tiles, n_cut), dim3(256), 0, s, a, Ha, Wa, desc, g, ga, n_cut, S);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_rescale_fwd(const float* pooled, float* base, int C, int S, int Hb, int Wb, hipStream_t s) {
    hipLaunchKernelGGL(rescale_fwd_kernel, dim3(ew_grid((size_t)C * Hb * Wb)), dim3(256), 0, s, pooled, base, C, S, Hb, Wb);
    PRX_LAUNCH_CHECK();
    return 0;
}
int prx_rescale_bwd(const float* g_base, float* g_pooled, int C, int S, int Hb, int Wb, hipStream_t s) {
    PRX_CHECK_HIP(hipMemsetAsync(g_pooled, 0, sizeof(float) * (size_t)C * S * S, s));
    hipLaunchKernelGGL(rescale_bwd_kernel, dim3(ew_grid((size_t)C * Hb * Wb)), dim3(256), 0, s, g_base, g_pooled, C, S, Hb, Wb);
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

int prx_patchify_fwd(const float* cut, const float* mm, void* A, int f32, int N, int S, int P, int T, hipStream_t s) {
    PRX_REQUIRE(S % P == 0 && T == (S / P) * (S / P) + 1, "patchify: bad geometry S=%d P=%d T=%d", S, P, T);
    const int Kp = patch_kp(P);
    const dim3 grid(ew_grid((size_t)N * T * Kp / 8));
    if (f32) hipLaunchKernelGGL(patchify_fwd_kernel<float>, grid, dim3(256), 0, s, cut, mm, (float*)A, N, S, P, T, Kp);
    else     hipLaunchKernelGGL(patchify_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, cut, mm, (bf16_t*)A, N, S, P, T, Kp);
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
