// Large-problem GEMM kernel of the engine (gemm.h): 256 x 256 tiles on the 8-phase main loop of gemm8p.h, row-major
// 16-bit operands (bf16 or IEEE half), the engine's fused epilogue.  Selected by prx_gemm_launch for problems that fill the
// chip with 256 x 256 tiles (ViT-B/16 / ViT-L/14 at 128-256 cutouts, the wide ViT-B/32 products); everything else stays on
// the 4-wave kernels of gemm.hip.
#include "gemm_epi.h"
#include "gemm8p.h"

namespace {
using namespace prx_gemm_dev;

template <typename T16>
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

void prx_gemm8p_launch(const prx_gemm_dev::GemmArgs& a, dim3 grid, hipStream_t s) {
    if (a.d.h16) hipLaunchKernelGGL(gemm8p_kernel<half_t>, grid, dim3(512), 0, s, a);
    else         hipLaunchKernelGGL(gemm8p_kernel<bf16_t>, grid, dim3(512), 0, s, a);
}
