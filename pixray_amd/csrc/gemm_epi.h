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
    int fit_flags;   // gemmfit.hip: bit 0 = staggered wave groups; bits 2, 3 = timing experiments (no epilogue / no main loop); bit 4 = column-major tile order; bit 6 = generic (run-time) epilogue only
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
template <typename TOp>
__device__ __forceinline__ void gnb_accum(const GemmDesc& d, const GnbConst& c, int row, int col, const float4& o, float& s0, float& s1) {
    const float4 x4 = (d.row16 & 2) ? op_ld4v(reinterpret_cast<const TOp*>(d.gnb_x), (size_t)row * d.N + col)
                                    : *reinterpret_cast<const float4*>(d.gnb_x + (size_t)row * d.N + col);
    const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, gv[4] = {o.x, o.y, o.z, o.w};
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
    if (d.resid) v += d.resid[(size_t)row * d.ldr + col];       // (fp32 residual only: 16-bit streams -- row16 -- need the vector epilogue, checked on the host;
                                                                 // this scalar form must stay small enough to unroll, or the accumulators it indexes go to scratch)
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

// The VALUE part of the epilogue on 4 consecutive columns, operands already in registers: alpha, bias_n[col..], bias_m[row]
// (zeros when absent), dQuickGELU / ReLU-mask forms on `aux`, residual, ReLU, QuickGELU.  `pre` receives the (operand-rounded)
// pre-activation of PRX_ACT_QUICKGELU.  has_resid: d.resid != nullptr (its 4 values are in `res`).
template <typename TOp>
__device__ __forceinline__ float4 epilogue_math4(int act, float alpha, float4 v, const float4& bias, float bias_m, const float (&aux)[4],
                                                 bool has_resid, const float4& res, float4& pre) {
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
    v.x += bias_m; v.y += bias_m; v.z += bias_m; v.w += bias_m;
    if (act == PRX_ACT_MUL_DQUICKGELU) {
        v.x *= dquickgelu_f(aux[0]); v.y *= dquickgelu_f(aux[1]); v.z *= dquickgelu_f(aux[2]); v.w *= dquickgelu_f(aux[3]);
    }
    // ReLU backward: the mask (aux > 0) multiplies the product (MUL_RELUMASK) or the product + residual (RELUMASK_POST)
    const bool masked = act == PRX_ACT_MUL_RELUMASK || act == PRX_ACT_RELUMASK_POST;
    const bool k0 = !masked || aux[0] > 0.f, k1 = !masked || aux[1] > 0.f, k2 = !masked || aux[2] > 0.f, k3 = !masked || aux[3] > 0.f;
    if (act == PRX_ACT_MUL_RELUMASK) {
        if (!k0) v.x = 0.f;
        if (!k1) v.y = 0.f;
        if (!k2) v.z = 0.f;
        if (!k3) v.w = 0.f;
    }
    if (has_resid) {
        v.x += res.x; v.y += res.y; v.z += res.z; v.w += res.w;
        if (act == PRX_ACT_RELUMASK_POST) {
            if (!k0) v.x = 0.f;
            if (!k1) v.y = 0.f;
            if (!k2) v.z = 0.f;
            if (!k3) v.w = 0.f;
        }
    }
    if (act == PRX_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (act == PRX_ACT_QUICKGELU) {
        // the saved pre-activation is what the backward differentiates: activate its rounded value
        pre = make_float4((float)op_cvt<TOp>(v.x), (float)op_cvt<TOp>(v.y), (float)op_cvt<TOp>(v.z), (float)op_cvt<TOp>(v.w));
        v.x = quickgelu_f(pre.x); v.y = quickgelu_f(pre.y); v.z = quickgelu_f(pre.z); v.w = quickgelu_f(pre.w);
    }
    return v;
}

// ... with the operand loads (all pointers / leading dimensions checked 16-byte friendly by the host)
template <typename TOp>
__device__ __forceinline__ float4 epilogue_value4(const GemmDesc& d, int row, int col, float4 v, float4& pre) {
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;     // uniform address: a scalar load
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias_n) bias = *reinterpret_cast<const float4*>(d.bias_n + col);
    const float bias_m = d.bias_m ? d.bias_m[row] : 0.f;
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.act == PRX_ACT_MUL_DQUICKGELU || d.act == PRX_ACT_MUL_RELUMASK || d.act == PRX_ACT_RELUMASK_POST)
        ax = op_ld4v(reinterpret_cast<const TOp*>(d.aux), (size_t)row * d.ldaux + col);
    const float aux[4] = {ax.x, ax.y, ax.z, ax.w};
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.resid) {
        if (d.row16 & 1) res = op_ld4v(reinterpret_cast<const TOp*>(d.resid), (size_t)row * d.ldr + col);
        else res = *reinterpret_cast<const float4*>(d.resid + (size_t)row * d.ldr + col);
    }
    return epilogue_math4<TOp>(d.act, alpha, v, bias, bias_m, aux, d.resid != nullptr, res, pre);
}

template <typename TOp>
__device__ __forceinline__ float4 epilogue_store4(const GemmDesc& d, int row, int col, float4 v) {
    float4 pre;
    v = epilogue_value4<TOp>(d, row, col, v, pre);
    if (d.act == PRX_ACT_QUICKGELU && d.out_bf16_pre)
        op_st4(reinterpret_cast<TOp*>(d.out_bf16_pre), (size_t)row * d.ldc_bf16 + col, pre.x, pre.y, pre.z, pre.w);
    if (d.out_f32) *reinterpret_cast<float4*>(d.out_f32 + (size_t)row * d.ldc_f32 + col) = v;
    if (d.out_bf16) op_st4(reinterpret_cast<TOp*>(d.out_bf16), (size_t)row * d.ldc_bf16 + col, v.x, v.y, v.z, v.w);
    return v;
}

// 8 consecutive columns (16-bit operand types only; the host checked col % 8 == 0-friendly pointers: 16-byte aligned 16-bit
// outputs with ldc_bf16 % 8 == 0): the 16-bit outputs leave as ONE 16-byte store per lane -- the store tail of a one-round
// kernel is bound by store instructions, not bytes (cdna_hip_programming.md T21)
template <typename T16>
__device__ __forceinline__ void epilogue_store8(const GemmDesc& d, int row, int col, float4 v0, float4 v1) {
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    float4 p0, p1;
    v0 = epilogue_value4<T16>(d, row, col, v0, p0);
    v1 = epilogue_value4<T16>(d, row, col + 4, v1, p1);
    if (d.act == PRX_ACT_QUICKGELU && d.out_bf16_pre) {
        t16x8 r;
        r[0] = op_cvt<T16>(p0.x); r[1] = op_cvt<T16>(p0.y); r[2] = op_cvt<T16>(p0.z); r[3] = op_cvt<T16>(p0.w);
        r[4] = op_cvt<T16>(p1.x); r[5] = op_cvt<T16>(p1.y); r[6] = op_cvt<T16>(p1.z); r[7] = op_cvt<T16>(p1.w);
        *reinterpret_cast<t16x8*>(reinterpret_cast<T16*>(d.out_bf16_pre) + (size_t)row * d.ldc_bf16 + col) = r;
    }
    if (d.out_f32) {
        float* o = d.out_f32 + (size_t)row * d.ldc_f32 + col;
        *reinterpret_cast<float4*>(o) = v0;
        *reinterpret_cast<float4*>(o + 4) = v1;
    }
    if (d.out_bf16) {
        t16x8 r;
        r[0] = op_cvt<T16>(v0.x); r[1] = op_cvt<T16>(v0.y); r[2] = op_cvt<T16>(v0.z); r[3] = op_cvt<T16>(v0.w);
        r[4] = op_cvt<T16>(v1.x); r[5] = op_cvt<T16>(v1.y); r[6] = op_cvt<T16>(v1.z); r[7] = op_cvt<T16>(v1.w);
        *reinterpret_cast<t16x8*>(reinterpret_cast<T16*>(d.out_bf16) + (size_t)row * d.ldc_bf16 + col) = r;
    }
}


}  // namespace prx_gemm_dev

// gemm8p.hip: the 256 x 256 8-phase kernel (row-major 16-bit operands, K % 128 == 0, no fused GroupNorm statistics)
bool prx_gemm8p_eligible(const GemmDesc& d);
void prx_gemm8p_launch(const prx_gemm_dev::GemmArgs& a, dim3 grid, hipStream_t s);      // grid = (tiles, splits), kt_per_split even
// gemmrow.hip: the row-streaming kernels for skinny K (K <= 320; N % 160 == 0, N % 128 == 0 or N % 80 == 0; M N >= 5 Mi): weights resident in LDS
bool prx_gemmrow_eligible(const GemmDesc& d);
int prx_gemmrow_launch(const prx_gemm_dev::GemmArgs& a, int n_cu, hipStream_t s);
long long prx_gemmrow_launches();
// gemmfit.hip: tiles whose count matches the chip (row-major 16-bit operands or implicit 3x3 convolutions with Cin % 64 == 0,
// K % (64 ks) == 0, 16-byte-friendly epilogue operands, no split-K across workgroups)
bool prx_gemmfit_tile(int bm, int bn, int* ks);
bool prx_gemmfit_eligible(const GemmDesc& d, int bm, int bn);
void prx_gemmfit_plan(const GemmDesc& d, int n_cu, int* bm, int* bn);   // *bm = 0: leave it to the 4-wave kernels
int prx_gemmfit_launch(const prx_gemm_dev::GemmArgs& a, int bm, int bn, dim3 grid, hipStream_t s);   // grid = (tiles, 1)
