// hipemu: a HOST stand-in for <hip/hip_runtime.h>, so that the non-MFMA kernels of pixray_amd/csrc compile with the host
// clang++ and run on the CPU, one workgroup at a time, every work-item a fiber (tools/hipemu/hipemu.cpp).  TEST
// INFRASTRUCTURE ONLY: it exists so that the kernels' SOURCE can be checked against the oracle in this GPU-less container
// (tests/test_emu_cpu.py); nothing in the product loads it, and it says nothing about performance.
//
// Model: wave = 64 consecutive work-items of a workgroup; __syncthreads and the wave collectives (__shfl*, __ballot, ...)
// suspend the calling fiber until every live work-item of the workgroup / wave has arrived (convergent use, as on the
// hardware).  __shared__ objects are function-local statics (workgroups run one after another).  Atomics are plain
// read-modify-writes (one OS thread).  "Device memory" is host memory: hipMalloc = malloc, streams are ignored, every launch
// is synchronous.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- vector types (the members the kernels use) -------------------------------------------------------------------------
#define HIPEMU_VEC2(T, N) struct N { T x, y; }; static inline N make_##N(T x, T y) { return N{x, y}; }
#define HIPEMU_VEC3(T, N) struct N { T x, y, z; }; static inline N make_##N(T x, T y, T z) { return N{x, y, z}; }
#define HIPEMU_VEC4(T, N) struct alignas(4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T)) N { T x, y, z, w; }; \
    static inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
HIPEMU_VEC3(float, float3)
HIPEMU_VEC4(float, float4)
struct alignas(8) int2 { int x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
HIPEMU_VEC3(int, int3)
HIPEMU_VEC4(int, int4)
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
HIPEMU_VEC4(unsigned, uint4)
struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
template <typename T, unsigned N> struct HIP_vector_type;
template <> struct HIP_vector_type<float, 2u> : float2 {};

// ---- runtime ------------------------------------------------------------------------------------------------------------
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3,
                     hipMemcpyDefault = 4 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    int clockRate;
    int warpSize;
};
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
// device allocations carry a 4 KB red zone on both sides; every launch end and every hipFree checks them (hipemu.cpp), so a
// kernel that writes past a workspace is named at the launch that did it instead of corrupting a neighbour silently
namespace hipemu { void* dev_alloc(size_t n); void dev_free(void* p); }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)hipemu::dev_alloc(n); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void* p) { hipemu::dev_free(p); return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu (host)"); strcpy(p->gcnArchName, "host");
    p->multiProcessorCount = 256; p->warpSize = 64; p->sharedMemPerBlock = 160 * 1024; p->totalGlobalMem = (size_t)1 << 34;
    return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                                         hipStream_t) {
    for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
// inter-process memory handles: not emulated (pixray_amd/csrc/comm.hip compiles, its exchange cannot be set up)
struct hipIpcMemHandle_t { char reserved[64]; };
enum { hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1 };
static inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorInvalidValue; }
static inline hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorInvalidValue; }
static inline hipError_t hipIpcCloseMemHandle(void*) { return hipErrorInvalidValue; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx g_tid, g_bid;
extern dim3 g_bdim, g_gdim;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void syncthreads();
// wave collectives: deposit `bytes` (<= DEPOSIT) of this lane's value, wait for the live lanes of the wave, then read lanes' deposits
constexpr int DEPOSIT = 128, RESULT = 64;
void wave_exchange(const void* mine, size_t bytes);
// the same, plus `reduce` run ONCE for the wave over all deposits (mask: which lanes are live); returns this lane's RESULT bytes
typedef void (*WaveReduce)(unsigned char (*deposits)[DEPOSIT], unsigned long long mask, unsigned char (*results)[RESULT]);
const void* wave_collective(const void* mine, size_t bytes, WaveReduce reduce);
const void* wave_slot(int lane);          // a lane's deposit of the last exchange (nullptr: that lane is not live)
int lane_id();
void set_strict_barrier(bool on);
void set_reverse_order(bool on);
void dma_issue(const void* src, void* dst, int bytes);
void waitcnt_vm(int n);
void wave_sync();                     // all live lanes of the wave arrive (lockstep points: /*hipemu:wave_sync*/ markers in the sources)
unsigned long long wave_live_mask();
}  // namespace hipemu
#define threadIdx hipemu::g_tid
#define blockIdx hipemu::g_bid
#define blockDim hipemu::g_bdim
#define gridDim hipemu::g_gdim
#define warpSize 64
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::waitcnt_vm(0); hipemu::syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_barrier() hipemu::syncthreads()
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_logf(x) log2f(x)
#define HIPEMU_ASM(...) ((void)0)          /* the emulated build rewrites `asm volatile(` (all of them s_waitcnt) to this */

template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    hipemu::wave_exchange(&v, sizeof(T));
    const int lane = hipemu::lane_id();
    const int s = (lane & ~(width - 1)) | (src & (width - 1));
    const void* p = hipemu::wave_slot(s);
    T r = v;
    if (p) memcpy(&r, p, sizeof(T));
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    hipemu::wave_exchange(&v, sizeof(T));
    const int lane = hipemu::lane_id();
    const int s = lane ^ mask;
    const void* p = ((s & ~(width - 1)) == (lane & ~(width - 1))) ? hipemu::wave_slot(s) : nullptr;
    T r = v;
    if (p) memcpy(&r, p, sizeof(T));
    return r;
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    hipemu::wave_exchange(&v, sizeof(T));
    const int lane = hipemu::lane_id();
    const int s = lane + (int)d;
    const void* p = ((s & ~(width - 1)) == (lane & ~(width - 1))) ? hipemu::wave_slot(s) : nullptr;
    T r = v;
    if (p) memcpy(&r, p, sizeof(T));
    return r;
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    hipemu::wave_exchange(&v, sizeof(T));
    const int lane = hipemu::lane_id();
    const int s = lane - (int)d;
    const void* p = (s >= 0 && (s & ~(width - 1)) == (lane & ~(width - 1))) ? hipemu::wave_slot(s) : nullptr;
    T r = v;
    if (p) memcpy(&r, p, sizeof(T));
    return r;
}
static inline unsigned long long __ballot(int pred) {
    const int v = pred != 0;
    hipemu::wave_exchange(&v, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const void* p = hipemu::wave_slot(l);
        if (p && *(const int*)p) m |= 1ull << l;
    }
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(pred) == hipemu::wave_live_mask(); }
static inline unsigned long long __activemask() { return hipemu::wave_live_mask(); }

// ---- atomics (single OS thread: plain read-modify-write) ------------------------------------------------------------------
template <typename T, typename U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = o + (T)v; return o; }
static inline unsigned atomicInc(unsigned* p, unsigned wrap) { unsigned o = *p; *p = o >= wrap ? 0u : o + 1u; return o; }
template <typename T, typename U> static inline T atomicSub(T* p, U v) { T o = *p; *p = o - (T)v; return o; }
template <typename T, typename U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <typename T, typename U> static inline T atomicOr(T* p, U v) { T o = *p; *p = o | (T)v; return o; }
template <typename T, typename U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = o & (T)v; return o; }
template <typename T, typename U, typename V> static inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))

// ---- math / bit intrinsics -------------------------------------------------------------------------------------------------
#define __expf(x) expf(x)      /* glibc declares these names itself */
#define __logf(x) logf(x)
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
static inline float __fdividef(float a, float b) { return a / b; }
// the *_rn forms are single correctly rounded operations that the compiler must not contract into an fma
__attribute__((optnone)) static float __fadd_rn(float a, float b) { return a + b; }
__attribute__((optnone)) static float __fsub_rn(float a, float b) { return a - b; }
__attribute__((optnone)) static float __fmul_rn(float a, float b) { return a * b; }
__attribute__((optnone)) static float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
__attribute__((optnone)) static double __dadd_rn(double a, double b) { return a + b; }
__attribute__((optnone)) static double __dmul_rn(double a, double b) { return a * b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
template <typename T> static inline T __ldg(const T* p) { return *p; }
using std::max;
using std::min;

// ---- LDS DMA: global_load_lds_dwordx4 -- every lane moves `size` bytes from ITS global address to (wave-uniform LDS base) +
// lane * size (+ the instruction offset).  Two completion models (HIPEMU_DMA):
//   deferred (default)  the bytes are read at issue and LAND only when the issuing wave's counted wait retires them
//                       (hipemu::waitcnt_vm(n): all but the newest n of this lane's DMA instructions; __syncthreads() implies
//                       n = 0, as HIP's fence does; the raw s_barrier does NOT; the end of the kernel does) -- the latest legal
//                       completion: code that reads a stage before it waited for it sees stale LDS;
//   eager               they land at issue -- the earliest legal completion: code that overwrites a stage somebody still
//                       reads is caught.
// vmcnt also counts ordinary vector loads / stores on the hardware; the K loops that use counted waits issue nothing else
// between them, so counting DMA instructions is exact there.
#define __builtin_amdgcn_global_load_lds(gptr, lptr, size, offset, aux) \
    hipemu::dma_issue((const void*)(gptr), (char*)(lptr) + hipemu::lane_id() * (size) + (offset), (size))

// ---- MFMA (wave-wide matrix instructions) as wave collectives, CDNA3/4 register layouts:
//   32x32xK (K = 16 16-bit / 2 f32): A lane l holds row l % 32, k-slice l / 32;  B lane l holds column l % 32, k-slice l / 32;
//       D/C lane l holds column l % 32, rows 8 (i / 4) + 4 (l / 32) + i % 4 for i = 0 .. 15
//   16x16x32: A lane l holds row l % 16, k = 8 (l / 16) .. + 7;  B column l % 16, same k;  D/C column l % 16, rows 4 (l / 16) + i
namespace hipemu {
// D = C + A B for the whole wave at once (run by the scheduler when the last live lane has arrived)
template <int MN, int KSL, typename V8>
static void mfma_16bit_reduce(unsigned char (*dep)[DEPOSIT], unsigned long long mask, unsigned char (*res)[RESULT]) {
    constexpr int K = 8 * KSL, NOUT = MN * MN / 64;
    struct Dep { V8 a, b; float c[NOUT]; };
    static float A[MN][K], B[K][MN], D[MN][MN];
    for (int l = 0; l < 64; ++l) {
        const Dep* d = (const Dep*)dep[l];
        const bool live = (mask >> l) & 1;
        const int rc = l % MN, ks = l / MN;
        for (int e = 0; e < 8; ++e) {
            A[rc][8 * ks + e] = live ? (float)d->a[e] : 0.f;
            B[8 * ks + e][rc] = live ? (float)d->b[e] : 0.f;
        }
        for (int i = 0; i < NOUT; ++i) {
            const int m = MN == 32 ? 8 * (i / 4) + 4 * ks + i % 4 : 4 * ks + i;
            D[m][rc] = live ? d->c[i] : 0.f;
        }
    }
    for (int m = 0; m < MN; ++m)
        for (int k = 0; k < K; ++k) {
            const float a = A[m][k];
            for (int n = 0; n < MN; ++n) D[m][n] += a * B[k][n];
        }
    for (int l = 0; l < 64; ++l) {
        float* out = (float*)res[l];
        const int rc = l % MN, ks = l / MN;
        for (int i = 0; i < NOUT; ++i) {
            const int m = MN == 32 ? 8 * (i / 4) + 4 * ks + i % 4 : 4 * ks + i;
            out[i] = D[m][rc];
        }
    }
}
template <int MN, int KSL, typename V8, typename ACC>   // KSL: k-slices (lane groups), 8 elements each
static inline ACC mfma_16bit(V8 a, V8 b, ACC c) {
    constexpr int NOUT = MN * MN / 64;
    struct Dep { V8 a, b; float c[NOUT]; } mine;
    static_assert(sizeof(Dep) <= DEPOSIT && NOUT * 4 <= RESULT, "deposit / result slots too small");
    mine.a = a; mine.b = b;
    for (int i = 0; i < NOUT; ++i) mine.c[i] = c[i];
    const float* r = (const float*)wave_collective(&mine, sizeof(Dep), &mfma_16bit_reduce<MN, KSL, V8>);
    for (int i = 0; i < NOUT; ++i) c[i] = r[i];
    return c;
}
}  // namespace hipemu
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_16bit<32, 2>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu::mfma_16bit<32, 2>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu::mfma_16bit<16, 4>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu::mfma_16bit<16, 4>((a), (b), (c))
namespace hipemu {
static void mfma_32x32x2_f32_reduce(unsigned char (*dep)[DEPOSIT], unsigned long long mask, unsigned char (*res)[RESULT]) {
    struct Dep { float a, b; float c[16]; };
    static float A[32][2], B[2][32], D[32][32];
    for (int l = 0; l < 64; ++l) {
        const Dep* d = (const Dep*)dep[l];
        const bool live = (mask >> l) & 1;
        const int rc = l % 32, k = l / 32;
        A[rc][k] = live ? d->a : 0.f;
        B[k][rc] = live ? d->b : 0.f;
        for (int i = 0; i < 16; ++i) D[8 * (i / 4) + 4 * k + i % 4][rc] = live ? d->c[i] : 0.f;
    }
    for (int m = 0; m < 32; ++m)
        for (int k = 0; k < 2; ++k)
            for (int n = 0; n < 32; ++n) D[m][n] = fmaf(A[m][k], B[k][n], D[m][n]);      // one fused multiply-add per k, k ascending
    for (int l = 0; l < 64; ++l) {
        float* out = (float*)res[l];
        const int rc = l % 32, k = l / 32;
        for (int i = 0; i < 16; ++i) out[i] = D[8 * (i / 4) + 4 * k + i % 4][rc];
    }
}
template <typename ACC> static inline ACC mfma_32x32x2_f32(float a, float b, ACC c) {
    struct Dep { float a, b; float c[16]; } mine;
    mine.a = a; mine.b = b;
    for (int i = 0; i < 16; ++i) mine.c[i] = c[i];
    const float* r = (const float*)wave_collective(&mine, sizeof(Dep), &mfma_32x32x2_f32_reduce);
    for (int i = 0; i < 16; ++i) c[i] = r[i];
    return c;
}
}  // namespace hipemu
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2_f32((a), (b), (c))
namespace hipemu {
//   16x16x4 f32: A lane l holds row l % 16, k = l / 16;  B column l % 16, same k;  D/C column l % 16, rows 4 (l / 16) + i
static void mfma_16x16x4_f32_reduce(unsigned char (*dep)[DEPOSIT], unsigned long long mask, unsigned char (*res)[RESULT]) {
    struct Dep { float a, b; float c[4]; };
    static float A[16][4], B[4][16], D[16][16];
    for (int l = 0; l < 64; ++l) {
        const Dep* d = (const Dep*)dep[l];
        const bool live = (mask >> l) & 1;
        const int rc = l % 16, k = l / 16;
        A[rc][k] = live ? d->a : 0.f;
        B[k][rc] = live ? d->b : 0.f;
        for (int i = 0; i < 4; ++i) D[4 * k + i][rc] = live ? d->c[i] : 0.f;
    }
    for (int m = 0; m < 16; ++m)
        for (int k = 0; k < 4; ++k)
            for (int n = 0; n < 16; ++n) D[m][n] = fmaf(A[m][k], B[k][n], D[m][n]);      // one fused multiply-add per k, k ascending
    for (int l = 0; l < 64; ++l) {
        float* out = (float*)res[l];
        const int rc = l % 16, k = l / 16;
        for (int i = 0; i < 4; ++i) out[i] = D[4 * k + i][rc];
    }
}
template <typename ACC> static inline ACC mfma_16x16x4_f32(float a, float b, ACC c) {
    struct Dep { float a, b; float c[4]; } mine;
    mine.a = a; mine.b = b;
    for (int i = 0; i < 4; ++i) mine.c[i] = c[i];
    const float* r = (const float*)wave_collective(&mine, sizeof(Dep), &mfma_16x16x4_f32_reduce);
    for (int i = 0; i < 4; ++i) c[i] = r[i];
    return c;
}
}  // namespace hipemu
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_16x16x4_f32((a), (b), (c))
