// Device-side pieces shared by the GEMM kernels of gemm.hip and gemm8p.hip: the launch argument block and the fused epilogue
// (bias, residual, QuickGELU / ReLU forms, operand-precision twins, GroupNorm statistics) -- see gemm.h for the semantics.
#pragma once
#include "gemm.h"
#include <type_traits>

namespace prx_gemm_dev {

struct GemmArgs {
    GemmDesc d;
    int tiles_m, tiles_n, splits, kt_per_split, kt_total;
    int vec_epi;   // 1: epilogue may use 4-wide vector accesses (alignment verified on the host)
    int xcd_swizzle;
    float* ws;
};

__device__ __forceinline__ float quickgelu_f(float t) { return t * sigmoidf_(1.702f * t); }
__device__ __forceinline__ float dquickgelu_f(float t) {
    float s = sigmoidf_(1.702f * t);
    return s * (1.f + 1.702f * t * (1.f - s));
}

// ---- fused GroupNorm-backward sums (GemmDesc::gnb_*) --------------------------------------------------------------
struct GnbConst { float mean, rstd; float4 ga, be; };
__device__ __forceinline__ GnbConst gnb_load(const GemmDesc& d, int col) {
    GnbConst c;
    const int g = col / d.gn_gs;
    const double n = (double)d.M * d.gn_gs;
    const double m = d.gnb_fstats[g * 2] / n;
    double var = d.gnb_fstats[g * 2 + 1] / n - m * m;
    if (var < 0) var = 0;
    c.mean = (float)m;
    c.rstd = (float)(1.0 / sqrt(var + (double)d.gnb_eps));
    c.ga = *reinterpret_cast<const float4*>(d.gnb_gamma + col);
    c.be = *reinterpret_cast<const float4*>(d.gnb_beta + col);
    return c;
}
__device__ __forceinline__ void gnb_accum(const GemmDesc& d, const GnbConst& c, int row, int col, const float4& o, float& s0, float& s1) {
    const float4 x = *reinterpret_cast<const float4*>(d.gnb_x + (size_t)row * d.N + col);
    const float xv[4] = {x.x, x.y, x.z, x.w}, gv[4] = {o.x, o.y, o.z, o.w};
    const float gav[4] = {c.ga.x, c.ga.y, c.ga.z, c.ga.w}, bev[4] = {c.be.x, c.be.y, c.be.z, c.be.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - c.mean) * c.rstd;
        float gy = gv[i];
        if (d.gnb_swish) {
            const float y = xh * gav[i] + bev[i];
            const float sg = sigmoidf_(y);
            gy *= sg * (1.f + y * (1.f - sg));
        }
        const float dxh = gy * gav[i];
        s0 += dxh;
        s1 += dxh * xh;
    }
}

// TOp = element type of the operand-precision pointers (aux, out_bf16, out_bf16_pre): bf16_t, or float in the exact mode
template <typename TOp>
__device__ __forceinline__ void epilogue_store(const GemmDesc& d, int row, int col, float v) {
    const TOp* aux = reinterpret_cast<const TOp*>(d.aux);
    v *= d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    if (d.bias_n) v += d.bias_n[col];
    if (d.bias_m) v += d.bias_m[row];
    if (d.act == PRX_ACT_MUL_DQUICKGELU) v *= dquickgelu_f(op_ld(aux, (size_t)row * d.ldaux + col));
    const bool masked = (d.act == PRX_ACT_MUL_RELUMASK || d.act == PRX_ACT_RELUMASK_POST) && !(op_ld(aux, (size_t)row * d.ldaux + col) > 0.f);
    if (masked && d.act == PRX_ACT_MUL_RELUMASK) v = 0.f;
    if (d.resid) v += d.resid[(size_t)row * d.ldr + col];
    if (masked && d.act == PRX_ACT_RELUMASK_POST) v = 0.f;          // the mask after the residual add
    if (d.act == PRX_ACT_RELU) v = fmaxf(v, 0.f);
    if (d.act == PRX_ACT_QUICKGELU) {
        const TOp pre = op_cvt<TOp>(v);     // the saved pre-activation is what the backward differentiates: activate its rounded value
        if (d.out_bf16_pre) reinterpret_cast<TOp*>(d.out_bf16_pre)[(size_t)row * d.ldc_bf16 + col] = pre;
        v = quickgelu_f((float)pre);
    }
    if (d.out_f32) d.out_f32[(size_t)row * d.ldc_f32 + col] = v;
    if (d.out_bf16) op_st(reinterpret_cast<TOp*>(d.out_bf16), (size_t)row * d.ldc_bf16 + col, v);
}

// 4 consecutive columns at once (all pointers / leading dimensions checked 16-byte friendly by the host)
template <typename TOp>
__device__ __forceinline__ float4 epilogue_store4(const GemmDesc& d, int row, int col, float4 v) {
    const TOp* aux = reinterpret_cast<const TOp*>(d.aux);
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;     // uniform address: a scalar load
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    if (d.bias_n) {
        const float4 b = *reinterpret_cast<const float4*>(d.bias_n + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (d.bias_m) { const float b = d.bias_m[row]; v.x += b; v.y += b; v.z += b; v.w += b; }
    if (d.act == PRX_ACT_MUL_DQUICKGELU) {
        float t[4];
        op_ld4(aux, (size_t)row * d.ldaux + col, t);
        v.x *= dquickgelu_f(t[0]); v.y *= dquickgelu_f(t[1]); v.z *= dquickgelu_f(t[2]); v.w *= dquickgelu_f(t[3]);
    }
    // ReLU backward: the mask (aux > 0) multiplies the product (MUL_RELUMASK) or the product + residual (RELUMASK_POST)
    float4 keep = make_float4(1.f, 1.f, 1.f, 1.f);
    if (d.act == PRX_ACT_MUL_RELUMASK || d.act == PRX_ACT_RELUMASK_POST) {
        float t[4];
        op_ld4(aux, (size_t)row * d.ldaux + col, t);
        keep = make_float4(t[0] > 0.f ? 1.f : 0.f, t[1] > 0.f ? 1.f : 0.f, t[2] > 0.f ? 1.f : 0.f, t[3] > 0.f ? 1.f : 0.f);
    }
    if (d.act == PRX_ACT_MUL_RELUMASK) {
        if (keep.x == 0.f) v.x = 0.f;
        if (keep.y == 0.f) v.y = 0.f;
        if (keep.z == 0.f) v.z = 0.f;
        if (keep.w == 0.f) v.w = 0.f;
    }
    if (d.resid) {
        const float4 r = *reinterpret_cast<const float4*>(d.resid + (size_t)row * d.ldr + col);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        if (d.act == PRX_ACT_RELUMASK_POST) {
            if (keep.x == 0.f) v.x = 0.f;
            if (keep.y == 0.f) v.y = 0.f;
            if (keep.z == 0.f) v.z = 0.f;
            if (keep.w == 0.f) v.w = 0.f;
        }
    }
    if (d.act == PRX_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (d.act == PRX_ACT_QUICKGELU) {
        const TOp p0 = op_cvt<TOp>(v.x), p1 = op_cvt<TOp>(v.y), p2 = op_cvt<TOp>(v.z), p3 = op_cvt<TOp>(v.w);
        if (d.out_bf16_pre)
            op_st4(reinterpret_cast<TOp*>(d.out_bf16_pre), (size_t)row * d.ldc_bf16 + col, (float)p0, (float)p1, (float)p2, (float)p3);
        v.x = quickgelu_f((float)p0); v.y = quickgelu_f((float)p1); v.z = quickgelu_f((float)p2); v.w = quickgelu_f((float)p3);
    }
    if (d.out_f32) *reinterpret_cast<float4*>(d.out_f32 + (size_t)row * d.ldc_f32 + col) = v;
    if (d.out_bf16) op_st4(reinterpret_cast<TOp*>(d.out_bf16), (size_t)row * d.ldc_bf16 + col, v.x, v.y, v.z, v.w);
    return v;
}


}  // namespace prx_gemm_dev

// gemm8p.hip: the 256 x 256 8-phase kernel (row-major 16-bit operands, K % 128 == 0, no fused GroupNorm statistics)
bool prx_gemm8p_eligible(const GemmDesc& d);
void prx_gemm8p_launch(const prx_gemm_dev::GemmArgs& a, dim3 grid, hipStream_t s);      // grid = (tiles, splits), kt_per_split even
