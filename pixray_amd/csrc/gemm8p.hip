// Large-problem GEMM kernel of the engine (gemm.h): 256 x 256 tiles on the 8-phase main loop of gemm8p.h, row-major
// 16-bit operands (bf16 or IEEE half), the engine's fused epilogue.  Selected by prx_gemm_launch for problems that fill the
// chip with 256 x 256 tiles (ViT-B/16 / ViT-L/14 at 128-256 cutouts, the wide ViT-B/32 products); everything else stays on
// the 4-wave kernels of gemm.hip.
#include "gemm_epi.h"
#include "gemm8p.h"
#include <atomic>
#include <stdlib.h>

namespace {
using namespace prx_gemm_dev;

// ---- the tower's epilogues at compile time (round 6; gemmfit_kernel.h FIT_EPI_OUT16 / RES16 / GELU / DGELU, IEEE-half operands) -----
// The generic path below calls epilogue_store4 once per 4 columns: 32 times per lane, each with its own bias / residual / aux
// loads BEHIND the previous call's stores (a load cannot be hoisted over a store that may alias it) -- 32 serial global round
// trips per wave, uniform branches over every fused feature, correctly rounded divisions in the sigmoids.  Here: 8 columns per
// lane (16-byte accesses of the 16-bit streams), the row operand of the NEXT 32-row block in flight while the current one is
// finished, everything the descriptor decides a template argument, v_exp + v_rcp sigmoid.  Arithmetic term by term as the
// generic epilogue's (bit-identical where no sigmoid is evaluated).
enum { G8_EPI_GENERIC = 0, G8_EPI_OUT16 = 1, G8_EPI_RES16 = 2, G8_EPI_GELU = 3, G8_EPI_DGELU = 4 };

__device__ __forceinline__ float g8_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <int EPI>
__device__ __forceinline__ void g8_finish_spec(const GemmArgs& p, bf16_t* lds, f32x16 (&acc)[4][2], int tm, int tn) {
    typedef __attribute__((ext_vector_type(8))) half_t h16x8;
    constexpr bool RESID = EPI == G8_EPI_RES16, DGELU = EPI == G8_EPI_DGELU, GELU = EPI == G8_EPI_GELU;
    constexpr bool ROWOP = RESID || DGELU;
    const GemmDesc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int rbase = tm * 256 + wr * 128, cbase = tn * 256 + wc * 64;
    constexpr int LDW = 64;                                           // no padding: 16 lanes read one 256-byte row, the two write halves hit disjoint lane groups
    float* const stage = reinterpret_cast<float*>(lds) + wave * (32 * LDW);
    const int srow = 4 * (lane >> 5), scol = lane & 31;               // accumulator layout: rows (r & 3) + 8 (r >> 2) + srow, column j * 32 + scol
    const int lr0 = lane >> 3, lc = (lane & 7) * 8;                   // output layout: 8 lanes per row, 8 rows per pass
    const int col = cbase + lc;
    const bool col_ok = col < d.N;                                    // N % 8 == 0 (checked on the host)
    const int colc = col_ok ? col : 0;
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d.bias_n) {
        const float4 b0 = *reinterpret_cast<const float4*>(d.bias_n + colc), b1 = *reinterpret_cast<const float4*>(d.bias_n + colc + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    const half_t* const rowp = RESID ? reinterpret_cast<const half_t*>(d.resid) : reinterpret_cast<const half_t*>(d.aux);
    const size_t ldrow = RESID ? (size_t)d.ldr : (size_t)d.ldaux;
    half_t* const out = reinterpret_cast<half_t*>(d.out_bf16);
    half_t* const outp = reinterpret_cast<half_t*>(d.out_bf16_pre);
    const int mlast = d.M - 1;
    uint4 pf[2][4];
    auto fetch = [&](int I, uint4 (&r)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            int row = rbase + I * 32 + ps * 8 + lr0;
            row = row < mlast ? row : mlast;
            r[ps] = *reinterpret_cast<const uint4*>(rowp + ((size_t)row * ldrow + colc));
        }
    };
    if constexpr (ROWOP) fetch(0, pf[0]);
#define G8_SPEC_BLOCK(I)                                                                                                \
    do {                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                   \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                              \
                stage[((r & 3) + 8 * (r >> 2) + srow) * LDW + j * 32 + scol] = acc[I][j][r];                             \
        if constexpr (ROWOP) { if ((I) < 3) fetch((I) + 1, pf[((I) + 1) & 1]); }                                        \
        /*hipemu:wave_sync*/                                                                                            \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                              \
            const int lr = ps * 8 + lr0, row = rbase + (I) * 32 + lr;                                                   \
            const bool ok = col_ok && row < d.M;                                                                        \
            const float4 s0 = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc]);                                  \
            const float4 s1 = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc + 4]);                              \
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};                                              \
            float x[8];                                                                                                 \
            if constexpr (ROWOP) {                                                                                      \
                const h16x8 rx = __builtin_bit_cast(h16x8, pf[(I) & 1][ps]);                                            \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) x[e] = (float)rx[e];                                      \
            }                                                                                                           \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                             \
                v[e] *= alpha;                                                                                          \
                v[e] += bias[e];                                                                                        \
                if constexpr (DGELU) {                                                                                  \
                    const float sg = g8_sigmoid(1.702f * x[e]);                                                         \
                    v[e] *= sg * (1.f + 1.702f * x[e] * (1.f - sg));                                                    \
                }                                                                                                       \
                if constexpr (RESID) v[e] += x[e];                                                                      \
            }                                                                                                           \
            const size_t oo = (size_t)row * d.ldc_bf16 + col;                                                           \
            if constexpr (GELU) {                                                                                       \
                h16x8 q;                                                                                                \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                         \
                    q[e] = f32_to_f16_sat(v[e]);                                                                        \
                    const float t = (float)q[e];                                                                        \
                    v[e] = t * g8_sigmoid(1.702f * t);                                                                  \
                }                                                                                                       \
                if (ok) *reinterpret_cast<h16x8*>(outp + oo) = q;                                                       \
            }                                                                                                           \
            h16x8 q2;                                                                                                   \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) q2[e] = f32_to_f16_sat(v[e]);                                 \
            if (ok) *reinterpret_cast<h16x8*>(out + oo) = q2;                                                           \
        }                                                                                                               \
        /*hipemu:wave_sync*/                                                                                            \
    } while (0)
    G8_SPEC_BLOCK(0); G8_SPEC_BLOCK(1); G8_SPEC_BLOCK(2); G8_SPEC_BLOCK(3);
#undef G8_SPEC_BLOCK
}

template <typename T16, int EPI = G8_EPI_GENERIC>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[G8_LDS_ELEMS];      // the only __shared__ object (gemm8p.h)
    const GemmDesc& d = p.d;
    int bid = blockIdx.x;
    if (p.xcd_swizzle) bid = (int)xcd_linear(bid, gridDim.x);
    const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
    const int split = blockIdx.y;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(p.kt_total, kt0 + p.kt_per_split);            // both even (checked on the host)

    f32x16 acc[4][2];
    g8_mainloop<T16>(reinterpret_cast<const bf16_t*>(d.A) + (size_t)kt0 * G8_BK, d.lda,
                     reinterpret_cast<const bf16_t*>(d.B) + (size_t)kt0 * G8_BK, d.ldb, d.M, d.N, kt1 - kt0, tm, tn, lds, acc);

    // ---- epilogue: every wave stages its 32 x 64 slabs row-major through its OWN LDS region (no workgroup barrier: a wave's
    // ds_write -> ds_read of the same bytes is ordered by the LDS queue), so that bias / residual / aux loads and all stores
    // are 8-16-byte accesses of 4 consecutive columns per lane.  (The /*hipemu:wave_sync*/ comments mark where the lanes of a wave
    // exchange data relying on lockstep execution: tools/hipemu turns them into fiber synchronisation, the GPU build sees nothing.)
    if constexpr (EPI != G8_EPI_GENERIC) {
        g8_finish_spec<EPI>(p, lds, acc, tm, tn);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int rbase = tm * 256 + wr * 128, cbase = tn * 256 + wc * 64;
    if (p.vec_epi) {
        constexpr int LDW = 64 + 4;                                   // padded row (floats), 16-byte aligned rows
        float* stage = reinterpret_cast<float*>(lds) + wave * (32 * LDW);
        const int lr0 = lane >> 4, lc = (lane & 15) * 4;
        const int srow = 4 * (lane >> 5), scol = lane & 31;
        // one 32-row block of the wave tile; a macro with a LITERAL block index so that the accumulator array is only ever
        // indexed with constants (a rolled loop over the blocks makes hipcc park all 128 accumulators in scratch)
#define G8_EPI_BLOCK(I)                                                                                                 \
        do {                                                                                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                               \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                          \
                    stage[((r & 3) + 8 * (r >> 2) + srow) * LDW + j * 32 + scol] = acc[I][j][r];                         \
            /*hipemu:wave_sync*/                                                                                        \
            _Pragma("unroll") for (int rr = 0; rr < 32; rr += 4) {                                                      \
                const int lr = rr + lr0;                                                                                \
                const float4 v = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc]);                               \
                const int row = rbase + (I) * 32 + lr, col = cbase + lc;                                                \
                if (row < d.M && col < d.N) {                                                                           \
                    if (p.splits > 1) *reinterpret_cast<float4*>(&p.ws[((size_t)split * d.M + row) * d.N + col]) = v;   \
                    else epilogue_store4<T16>(d, row, col, v);                                                          \
                }                                                                                                       \
            }                                                                                                           \
            /*hipemu:wave_sync*/                                                                                        \
        } while (0)
        G8_EPI_BLOCK(0); G8_EPI_BLOCK(1); G8_EPI_BLOCK(2); G8_EPI_BLOCK(3);
#undef G8_EPI_BLOCK
        return;
    }
    const int row0 = rbase + 4 * (lane >> 5), col0 = cbase + (lane & 31);
#define G8_EPI_SCALAR(I)                                                                                                \
    do {                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
            const int col = col0 + j * 32;                                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                            \
                const int row = row0 + (I) * 32 + (r & 3) + 8 * (r >> 2);                                               \
                if (row < d.M && col < d.N) {                                                                           \
                    if (p.splits > 1) p.ws[((size_t)split * d.M + row) * d.N + col] = acc[I][j][r];                     \
                    else epilogue_store<T16>(d, row, col, acc[I][j][r]);                                                \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
    G8_EPI_SCALAR(0); G8_EPI_SCALAR(1); G8_EPI_SCALAR(2); G8_EPI_SCALAR(3);
#undef G8_EPI_SCALAR
}
}  // namespace

bool prx_gemm8p_eligible(const GemmDesc& d) {
    return !d.f32 && !d.a_is_f32 && d.a_mode == PRX_A_ROWMAJOR && d.K % 128 == 0 && d.gn_stats == nullptr;
}

int prx_gemmfit_epi_kind(const GemmDesc& d);             // gemmfit.hip: the compile-time epilogue a descriptor is an instance of (1 .. 4: the tower's)
extern std::atomic<long long> g_prx_gemm8p_spec_launches;
std::atomic<long long> g_prx_gemm8p_spec_launches{0};

void prx_gemm8p_launch(const prx_gemm_dev::GemmArgs& a, dim3 grid, hipStream_t s) {
    // the tower's descriptor patterns on kernels with a compile-time epilogue (bit 6 of the fit switch word keeps the generic one:
    // A/B runs); split-K launches write raw partials and stay generic
    static const bool spec_on = [] { const char* e = getenv("PRX_G8_SPEC"); return !(e && atoi(e) == 0); }();
    if (spec_on && !(a.fit_flags & 64) && a.d.h16 && a.splits == 1 && a.vec_epi && a.d.N % 8 == 0 && (a.d.ldc_bf16 % 8) == 0) {
        const int k = prx_gemmfit_epi_kind(a.d);
        const bool al = (((uintptr_t)a.d.out_bf16 | (uintptr_t)a.d.out_bf16_pre | (uintptr_t)a.d.resid | (uintptr_t)a.d.aux) & 15) == 0 &&
                        (!a.d.resid || a.d.ldr % 8 == 0) && (!a.d.aux || a.d.ldaux % 8 == 0);
        if (al && k >= G8_EPI_OUT16 && k <= G8_EPI_DGELU) {
            g_prx_gemm8p_spec_launches.fetch_add(1, std::memory_order_relaxed);
            switch (k) {
            case G8_EPI_OUT16: hipLaunchKernelGGL((gemm8p_kernel<half_t, G8_EPI_OUT16>), grid, dim3(512), 0, s, a); return;
            case G8_EPI_RES16: hipLaunchKernelGGL((gemm8p_kernel<half_t, G8_EPI_RES16>), grid, dim3(512), 0, s, a); return;
            case G8_EPI_GELU:  hipLaunchKernelGGL((gemm8p_kernel<half_t, G8_EPI_GELU>), grid, dim3(512), 0, s, a); return;
            default:           hipLaunchKernelGGL((gemm8p_kernel<half_t, G8_EPI_DGELU>), grid, dim3(512), 0, s, a); return;
            }
        }
    }
    if (a.d.h16) hipLaunchKernelGGL(gemm8p_kernel<half_t>, grid, dim3(512), 0, s, a);
    else         hipLaunchKernelGGL(gemm8p_kernel<bf16_t>, grid, dim3(512), 0, s, a);
}
