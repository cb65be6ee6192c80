// Shared device/host helpers for the gfx950 kernels of the pixray hot path.
// Everything here is written for CDNA4 (wave = 64 lanes, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PRX_WAVE 64

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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }
