// Shared device/host helpers for the gfx950 kernels of the pixray hot path.
// Everything here is written for CDNA4 (wave = 64 lanes, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PRX_WAVE 64

// ---- operand precision of a runner handle ------------------------------------
// PRX_PREC_BF16: GEMM operands (activations handed from kernel to kernel, weight packs) are bf16, MFMA 32x32x16 bf16.
// PRX_PREC_F32 : the same buffers hold fp32 and every contraction runs on v_mfma_f32_32x32x2_f32 (exact f32, the
//                f32 vector rate) -- the parity mode the bf16 numbers are measured against.
// Operand buffers are untyped (`void*`); kernels that read or write them are instantiated for both element types.
// PRX_PREC_F16 : 16-bit operands in IEEE half (v_mfma_f32_32x32x16_f16, the bf16 MFMA rate) -- the arithmetic the reference's
//                CLIP towers run in on a GPU (clip.load keeps fp16 weights / activations, slip.py:175).  11 significand bits
//                instead of bf16's 8; the narrow exponent is handled by (a) saturating conversions and (b) a power-of-two
//                gradient scale per runner backward (exact: every backward op is linear in the incoming gradient).
// Inside the library a handle carries two flags derived from its precision: `f32` (operand buffers hold fp32) and `h16`
// (the 16-bit operand format is half instead of bf16).  16-bit buffers are typed `bf16_t*` as raw 16-bit carriers; only
// conversions and the MFMA opcode depend on `h16`.
enum { PRX_PREC_BF16 = 0, PRX_PREC_F32 = 1, PRX_PREC_F16 = 2 };
static inline int prec_is_f32(int precision) { return precision == PRX_PREC_F32; }
static inline int prec_is_h16(int precision) { return precision == PRX_PREC_F16; }
static inline bool prec_valid(int precision) { return precision >= PRX_PREC_BF16 && precision <= PRX_PREC_F16; }
static inline size_t op_esz(int f32) { return f32 ? 4 : 2; }
static inline void* op_off(void* p, size_t elems, int f32) { return p ? (char*)p + elems * op_esz(f32) : nullptr; }
static inline const void* op_off(const void* p, size_t elems, int f32) { return p ? (const char*)p + elems * op_esz(f32) : nullptr; }

// ---- error plumbing (never throw across the C ABI) -------------------------
void prx_set_error(const char* fmt, ...);
#define PRX_CHECK_HIP(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            prx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                          hipGetErrorString(_e));                                  \
            return -1;                                                             \
        }                                                                          \
    } while (0)
#define PRX_REQUIRE(cond, ...)                                                     \
    do {                                                                           \
        if (!(cond)) {                                                             \
            prx_set_error(__VA_ARGS__);                                            \
            return -2;                                                             \
        }                                                                          \
    } while (0)
#define PRX_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t _e = hipGetLastError();                                         \
        if (_e != hipSuccess) {                                                    \
            prx_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,        \
                          hipGetErrorString(_e));                                  \
            return -1;                                                             \
        }                                                                          \
    } while (0)

// ---- bf16 conversions (round-to-nearest-even, matches torch .to(bfloat16)) --
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }

__device__ __forceinline__ bf16x8 pack_bf16x8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    return r;
}

// ---- 16-bit operand carrier with a run-time format (h16: 0 = bf16, 1 = IEEE half) ----------------------------------
// For the HBM-bound kernels (norms, elementwise, layout passes) the format is a wave-uniform kernel argument: one extra
// conversion + select per element is free there, and it avoids a second instantiation of every kernel.
#define PRX_F16_MAX 65504.f
__device__ __forceinline__ half_t f32_to_f16_sat(float v) {      // saturating (NaN passes through): no inf in the pipeline
    return (half_t)__builtin_fminf(__builtin_fmaxf(v, -PRX_F16_MAX), PRX_F16_MAX);
}
__device__ __forceinline__ bf16_t to_op16(float v, int h16) {
    if (h16) return __builtin_bit_cast(bf16_t, f32_to_f16_sat(v));
    return (bf16_t)v;
}
__device__ __forceinline__ float from_op16(bf16_t b, int h16) {
    if (h16) return (float)__builtin_bit_cast(half_t, b);
    return (float)b;
}
__device__ __forceinline__ bf16x4 to_op16x4(float a, float b, float c, float d, int h16) {
    bf16x4 r;
    r[0] = to_op16(a, h16); r[1] = to_op16(b, h16); r[2] = to_op16(c, h16); r[3] = to_op16(d, h16);
    return r;
}

// "stream" tensors of the runners (residual streams, feature maps, their gradients): fp32, or -- s16 -- the handle's 16-bit
// operand format (the lean layout of the half mode: one 16-bit tensor is both the saved activation and the next GEMM's operand).
// i4: index in units of 4 elements
// (S16 is a template argument: a run-time test between the loads of one iteration would serialise their round trips)
template <bool S16>
__device__ __forceinline__ float4 stream_ld4(const void* p, size_t i4, int h16) {
    if constexpr (S16) {
        const bf16x4 t = reinterpret_cast<const bf16x4*>(p)[i4];
        return make_float4(from_op16(t[0], h16), from_op16(t[1], h16), from_op16(t[2], h16), from_op16(t[3], h16));
    } else {
        return reinterpret_cast<const float4*>(p)[i4];
    }
}

// operand element access, generic over {bf16_t, half_t, float}
template <typename T> __device__ __forceinline__ float op_ld(const T* p, size_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void op_st(T* p, size_t i, float v) { p[i] = (T)v; }
template <typename T> __device__ __forceinline__ void op_ld4(const T* p, size_t i, float* v);
template <> __device__ __forceinline__ void op_ld4<bf16_t>(const bf16_t* p, size_t i, float* v) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p + i);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <> __device__ __forceinline__ void op_ld4<float>(const float* p, size_t i, float* v) {
    const float4 t = *reinterpret_cast<const float4*>(p + i);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void op_st<half_t>(half_t* p, size_t i, float v) { p[i] = f32_to_f16_sat(v); }
template <> __device__ __forceinline__ void op_ld4<half_t>(const half_t* p, size_t i, float* v) {
    const f16x4 t = *reinterpret_cast<const f16x4*>(p + i);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
// ... returning the four values (no local array at the call site: a `float v[4]` filled through a pointer inside a conditional
// branch kept every such array of the 128 x 128 half kernels' epilogues in scratch)
template <typename T> __device__ __forceinline__ float4 op_ld4v(const T* p, size_t i);
template <> __device__ __forceinline__ float4 op_ld4v<bf16_t>(const bf16_t* p, size_t i) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p + i);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
template <> __device__ __forceinline__ float4 op_ld4v<half_t>(const half_t* p, size_t i) {
    const f16x4 t = *reinterpret_cast<const f16x4*>(p + i);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
template <> __device__ __forceinline__ float4 op_ld4v<float>(const float* p, size_t i) { return *reinterpret_cast<const float4*>(p + i); }
template <typename T> __device__ __forceinline__ void op_st4(T* p, size_t i, float a, float b, float c, float d);
template <> __device__ __forceinline__ void op_st4<half_t>(half_t* p, size_t i, float a, float b, float c, float d) {
    f16x4 r;
    r[0] = f32_to_f16_sat(a); r[1] = f32_to_f16_sat(b); r[2] = f32_to_f16_sat(c); r[3] = f32_to_f16_sat(d);
    *reinterpret_cast<f16x4*>(p + i) = r;
}
// conversion to the operand element type (saturating for half)
template <typename T> __device__ __forceinline__ T op_cvt(float v) { return (T)v; }
template <> __device__ __forceinline__ half_t op_cvt<half_t>(float v) { return f32_to_f16_sat(v); }

// host-side dispatch of a kernel templated on the operand element type: f32 -> float, else h16 -> half_t, else bf16_t
#define PRX_OP_DISPATCH(f32_, h16_, T, ...)                                   \
    do {                                                                      \
        if (f32_) { using T = float; __VA_ARGS__; }                           \
        else if (h16_) { using T = half_t; __VA_ARGS__; }                     \
        else { using T = bf16_t; __VA_ARGS__; }                               \
    } while (0)
template <> __device__ __forceinline__ void op_st4<bf16_t>(bf16_t* p, size_t i, float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (bf16_t)a; r[1] = (bf16_t)b; r[2] = (bf16_t)c; r[3] = (bf16_t)d;
    *reinterpret_cast<bf16x4*>(p + i) = r;
}
template <> __device__ __forceinline__ void op_st4<float>(float* p, size_t i, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p + i) = make_float4(a, b, c, d);
}

// ---- wave / block reductions -----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- XCD-aware work split of streaming kernels -------------------------------------------------------------------------
// MI355X has 8 XCDs with a private 4 MB L2 each, and workgroup b is observed to run on XCD b % 8 (used for speed only:
// nothing below depends on it for correctness).  The GEMM engine can give every XCD a contiguous eighth of the tile rows
// (gemm.hip, xcd_swizzle); when the streaming kernels that produce its A operand and consume its output (GroupNorm /
// LayerNorm apply passes) split their index space the same way, an activation row is written and re-read through the SAME
// L2 instead of crossing the fabric on every kernel boundary.  xcd_linear() is the shared block -> slot map (bijective for
// any grid size); PRX_XCD_LOCAL=0 switches the elementwise side off for A/B runs.
__device__ __forceinline__ unsigned xcd_linear(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, local = bid >> 3, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}
int prx_xcd_local();      // process-wide constant read once from the environment (api_core.hip)
// The runner backwards of the half (PRX_PREC_F16) mode run under a power-of-two scale S chosen on the device per backward
// (elementwise.h prx_grad_scale) such that S * max|incoming gradient| lies in [2^(T-1), 2^T); T = PRX_GRAD_TARGET_LOG2,
// default 4: 2^12 of headroom below the half maximum for growth inside the backward, full half precision down to
// entries 2^-18 of the largest one, subnormal tails below that (api_core.hip).
int prx_grad_target_log2();

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }
