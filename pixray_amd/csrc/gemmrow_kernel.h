// Row-streaming GEMM for skinny K (gfx950):  C[M, N] = epilogue(A[M, K] . B[N, K]^T),  K <= 320, M in the 10^5 .. 10^6 range.
//
// The 1x1 convolutions of the ModifiedResNet runner's first two stages (resnet.hip: conv3 / downsample forward, conv1 dgrad;
// RN50x4 at 128 cutouts of 288^2: M = 663 552 or 165 888 rows, K = 80 / 160, N = 320 / 640) move 1 - 1.4 GB per launch for
// 34 - 68 GFLOP: they are streaming operations.  On the tiled MFMA kernels they ran at 1.4 - 1.8 TB/s (tools/gemm_shapes.py,
// profiles/r06_cfg2_gemm_shapes_*.txt) -- a 128 x 128 tile with two K tiles is all epilogue, and that epilogue goes through an LDS
// staging pass and run-time operand switches.  This kernel treats them as what they are:
//
//   * the WEIGHTS of one column slab (NW = 128 or 160 columns x K) stay in LDS for the life of the workgroup; workgroups are
//     persistent (one or two per CU) and walk a contiguous chunk of 16-row tiles;
//   * a wave owns 16 rows x NW columns: the activations go global -> registers as MFMA operands directly (16-byte loads, next
//     tile's loads in flight during this tile), no LDS ring, no barrier in the loop;
//   * the MFMA runs TRANSPOSED (weights as the A operand, activations as the B operand of v_mfma_f32_16x16x32), so a lane ends
//     up with 4 CONSECUTIVE output columns of one row per accumulator; the slab rows sit in LDS in an order that makes the two
//     accumulators of a column pair adjacent: 8 consecutive columns per lane = one 16-byte store / residual load / mask load, and
//     the epilogue needs no staging pass;
//   * the epilogue operands of a tile (residual, ReLU mask) are requested before its MFMAs; the activation, the operand formats
//     and the residual kind are template arguments;
//   * column slabs of the same rows run on the same XCD (blockIdx -> (xcd, slab, chunk)), so the activations are fetched from HBM
//     once and from that XCD's L2 by the other slabs.
//
// Bit-compatible with the tiled kernels' epilogue (gemm_epi.h epilogue_math4 is the arithmetic); the K sum runs in one MFMA
// chain per output instead of per-K-tile partial chains, so products differ by fp32 summation order only.
#pragma once
#include "gemm.h"
#include "gemm_epi.h"
#include <type_traits>

namespace prx_gemmrow_dev {
using namespace prx_gemm_dev;

constexpr int GR_WAVES = 8;          // waves per workgroup (all on one weight slab)

template <typename T16>
__device__ __forceinline__ f32x4 gr_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (std::is_same<T16, half_t>::value)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <typename T16>
__device__ __forceinline__ float4 gr_unpack4(const bf16x4& v) {
    typedef __attribute__((ext_vector_type(4))) T16 t16x4;
    const t16x4 t = __builtin_bit_cast(t16x4, v);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
template <typename T16>
__device__ __forceinline__ void gr_unpack(const bf16x8& v, float4& lo, float4& hi) {
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    const t16x8 t = __builtin_bit_cast(t16x8, v);
    lo = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    hi = make_float4((float)t[4], (float)t[5], (float)t[6], (float)t[7]);
}
__device__ __forceinline__ bf16x8 gr_zero8() {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
    return v;
}

// NT: 16-column MFMA tiles of a slab (NW = 16 NT columns): NT / 2 column PAIRS (a lane owns 8 consecutive columns of each) and, for odd
//     NT, one lone tile at the end (4 consecutive columns per lane) -- N = 80 is 2 pairs + 1
// KSM: K steps of 32 the kernel has registers and LDS for (6: K <= 192, the next tile's activations prefetched; 10: K <= 320 and
//     20: K <= 640, loaded at the start of their own tile; 20 exists for 80-column slabs only: 104 KB of weights)
// RES: 0 none, 1 fp32 residual, 2 residual in the 16-bit operand format (GemmDesc::row16 bit 0)
template <typename T16, int ACT, int RES, int NT, int KSM>
__global__ __launch_bounds__(GR_WAVES * 64) void gemmrow_kernel(GemmArgs a, int ksteps, int nslab, int row_tiles, int nchunks) {
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    typedef __attribute__((ext_vector_type(4))) T16 t16x4;
    constexpr int NW = NT * 16, NP = NT / 2;
    constexpr bool ODD = (NT & 1) != 0, PREF = KSM <= 6;
    constexpr bool HAS_AUX = ACT == PRX_ACT_MUL_RELUMASK || ACT == PRX_ACT_RELUMASK_POST;
    constexpr int GR_LD = KSM * 32 + 8;          // LDS row stride of the slab at the largest K (elements)
    __shared__ __attribute__((aligned(16))) bf16_t Bs[NW * GR_LD];
    __shared__ __attribute__((aligned(16))) float bias_s[NW];
    const GemmDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = ksteps * 32 + 8;              // slab row stride: (ld / 2) / 4 is odd -> the 8 rows of a ds_read_b128 phase cover all banks
    // workgroup -> (XCD, column slab, row chunk): consecutive workgroup ids go to consecutive XCDs, so the slabs of a chunk are `8` apart
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slab = idx % nslab, chunk = (idx / nslab) * 8 + xcd;
    const int n0 = slab * NW;

    // ---- the slab: LDS row r = (pair q, accumulator h, MFMA row rr) holds weight row n0 + 32 q + 8 (rr / 4) + 4 h + rr % 4; the rows
    // of a lone last tile are in their natural order
    {
        const int kch = ksteps * 4;              // 16-byte chunks per row, zero beyond K
        const bf16_t* Bg = reinterpret_cast<const bf16_t*>(d.B);
        for (int i = tid; i < NW * kch; i += GR_WAVES * 64) {
            const int r = i / kch, c = i - r * kch;
            const int q = r >> 5, h = (r >> 4) & 1, rr = r & 15;
            const int n = n0 + ((ODD && r >= NP * 32) ? r : q * 32 + (rr >> 2) * 8 + 4 * h + (rr & 3));
            bf16x8 v = gr_zero8();
            if (c * 8 < d.K) v = *reinterpret_cast<const bf16x8*>(Bg + (size_t)n * d.ldb + c * 8);
            *reinterpret_cast<bf16x8*>(Bs + r * ld + c * 8) = v;
        }
        for (int i = tid; i < NW; i += GR_WAVES * 64) bias_s[i] = d.bias_n ? d.bias_n[n0 + i] : 0.f;
    }
    __syncthreads();

    const int m_l = lane & 15, kg = lane >> 4;
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    const int per_chunk = (row_tiles + nchunks - 1) / nchunks;
    const int t_begin = chunk * per_chunk;
    const int t_end = t_begin + per_chunk < row_tiles ? t_begin + per_chunk : row_tiles;
    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(d.A);
    const bf16_t* wrow = Bs + m_l * ld + kg * 8;         // + (tile * 16) * ld + ks * 32
    const int ccol = kg * 8;                             // this lane's 8 columns inside a pair
    const int tcol = NP * 32 + kg * 4;                   // ... and its 4 columns of the lone tile, from the slab's first column

    bf16x8 acur[KSM], anxt[PREF ? KSM : 1];
    auto load_a = [&](int t, auto& fr) {
        int row = t * 16 + m_l;
        row = row < d.M ? row : d.M - 1;
        const bf16_t* p = Ag + (size_t)row * d.lda + kg * 8;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            if (ks < ksteps) {
                bf16x8 v = gr_zero8();
                if (ks * 32 + kg * 8 < d.K) v = *reinterpret_cast<const bf16x8*>(p + ks * 32);
                fr[ks] = v;
            }
        }
    };

    int t = t_begin + wave;
    if (PREF && t < t_end) load_a(t, acur);
    for (; t < t_end; t += GR_WAVES) {
        const int row = t * 16 + m_l;
        const bool live = row < d.M;
        const int rowc = live ? row : d.M - 1;
        if constexpr (!PREF) load_a(t, acur);
        // epilogue operands of this tile, then (PREF) the activations of the next one
        bf16x8 aux16[HAS_AUX && NP ? NP : 1];
        bf16x8 res16[RES == 2 && NP ? NP : 1];
        float4 res32[RES == 1 && NP ? NP : 1][2];
        bf16x4 auxt, rest;
        float4 rest32 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (HAS_AUX) {
            const T16* p = reinterpret_cast<const T16*>(d.aux) + (size_t)rowc * d.ldaux + n0;
#pragma unroll
            for (int q = 0; q < NP; ++q) aux16[q] = *reinterpret_cast<const bf16x8*>(p + q * 32 + ccol);
            if constexpr (ODD) auxt = *reinterpret_cast<const bf16x4*>(p + tcol);
        }
        if constexpr (RES == 2) {
            const T16* p = reinterpret_cast<const T16*>(d.resid) + (size_t)rowc * d.ldr + n0;
#pragma unroll
            for (int q = 0; q < NP; ++q) res16[q] = *reinterpret_cast<const bf16x8*>(p + q * 32 + ccol);
            if constexpr (ODD) rest = *reinterpret_cast<const bf16x4*>(p + tcol);
        }
        if constexpr (RES == 1) {
            const float* p = d.resid + (size_t)rowc * d.ldr + n0;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                res32[q][0] = *reinterpret_cast<const float4*>(p + q * 32 + ccol);
                res32[q][1] = *reinterpret_cast<const float4*>(p + q * 32 + ccol + 4);
            }
            if constexpr (ODD) rest32 = *reinterpret_cast<const float4*>(p + tcol);
        }
        if constexpr (PREF) { if (t + GR_WAVES < t_end) load_a(t + GR_WAVES, anxt); }

        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks) {
            if (ks < ksteps) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8*>(wrow + (j * 16) * ld + ks * 32);
                    acc[j] = gr_mfma<T16>(w, acur[ks], acc[j]);
                }
            }
        }

        // ---- epilogue: 8 consecutive columns per lane and pair (4 of the lone tile) ----------------------------------------------
        float4 pre;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int col = n0 + q * 32 + ccol;
            const float4 b0 = *reinterpret_cast<const float4*>(bias_s + q * 32 + ccol);
            const float4 b1 = *reinterpret_cast<const float4*>(bias_s + q * 32 + ccol + 4);
            float4 x0 = z4, x1 = z4, r0 = z4, r1 = z4;
            if constexpr (HAS_AUX) gr_unpack<T16>(aux16[q], x0, x1);
            if constexpr (RES == 2) gr_unpack<T16>(res16[q], r0, r1);
            if constexpr (RES == 1) { r0 = res32[q][0]; r1 = res32[q][1]; }
            const float ax0[4] = {x0.x, x0.y, x0.z, x0.w}, ax1[4] = {x1.x, x1.y, x1.z, x1.w};
            const f32x4 &c0 = acc[2 * q], &c1 = acc[2 * q + 1];
            const float4 v0 = epilogue_math4<T16>(ACT, alpha, make_float4(c0[0], c0[1], c0[2], c0[3]), b0, 0.f, ax0, RES != 0, r0, pre);
            const float4 v1 = epilogue_math4<T16>(ACT, alpha, make_float4(c1[0], c1[1], c1[2], c1[3]), b1, 0.f, ax1, RES != 0, r1, pre);
            if (live) {
                if (d.out_f32) {
                    float* o = d.out_f32 + (size_t)row * d.ldc_f32 + col;
                    *reinterpret_cast<float4*>(o) = v0;
                    *reinterpret_cast<float4*>(o + 4) = v1;
                }
                if (d.out_bf16) {
                    t16x8 o;
                    o[0] = op_cvt<T16>(v0.x); o[1] = op_cvt<T16>(v0.y); o[2] = op_cvt<T16>(v0.z); o[3] = op_cvt<T16>(v0.w);
                    o[4] = op_cvt<T16>(v1.x); o[5] = op_cvt<T16>(v1.y); o[6] = op_cvt<T16>(v1.z); o[7] = op_cvt<T16>(v1.w);
                    *reinterpret_cast<t16x8*>(reinterpret_cast<T16*>(d.out_bf16) + (size_t)row * d.ldc_bf16 + col) = o;
                }
            }
        }
        if constexpr (ODD) {
            const int col = n0 + tcol;
            const float4 b0 = *reinterpret_cast<const float4*>(bias_s + tcol);
            float4 x0 = z4, r0 = z4;
            if constexpr (HAS_AUX) x0 = gr_unpack4<T16>(auxt);
            if constexpr (RES == 2) r0 = gr_unpack4<T16>(rest);
            if constexpr (RES == 1) r0 = rest32;
            const float ax0[4] = {x0.x, x0.y, x0.z, x0.w};
            const f32x4& c0 = acc[NT - 1];
            const float4 v0 = epilogue_math4<T16>(ACT, alpha, make_float4(c0[0], c0[1], c0[2], c0[3]), b0, 0.f, ax0, RES != 0, r0, pre);
            if (live) {
                if (d.out_f32) *reinterpret_cast<float4*>(d.out_f32 + (size_t)row * d.ldc_f32 + col) = v0;
                if (d.out_bf16) {
                    t16x4 o;
                    o[0] = op_cvt<T16>(v0.x); o[1] = op_cvt<T16>(v0.y); o[2] = op_cvt<T16>(v0.z); o[3] = op_cvt<T16>(v0.w);
                    *reinterpret_cast<t16x4*>(reinterpret_cast<T16*>(d.out_bf16) + (size_t)row * d.ldc_bf16 + col) = o;
                }
            }
        }
        if constexpr (PREF) {
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) acur[ks] = anxt[ks];
        }
    }
}

// the (activation, residual) patterns that exist as kernels, per slab shape: the runner's conv3 / downsample forward and conv1 dgrad
// (pairs only, N % 128 == 0 or N % 160 == 0), its conv1 forward / conv3 dgrad (N = 80: no residual) and the first block's conv1
// dgrad (N = 80: residual, no activation)
template <typename T16, int RES, int NT, int KSM>
inline bool launch_instance(const GemmArgs& a, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
#define GR_CASE(ACT_)                                                                                                                   \
    case ACT_:                                                                                                                          \
        hipLaunchKernelGGL((gemmrow_kernel<T16, ACT_, RES, NT, KSM>), dim3(grid), dim3(GR_WAVES * 64), 0, s, a, ksteps, nslab, row_tiles, nchunks); \
        return true;
    if constexpr (RES == 0) {
        switch (a.d.act) {
            GR_CASE(PRX_ACT_NONE) GR_CASE(PRX_ACT_RELU)
            case PRX_ACT_MUL_RELUMASK:
                if constexpr ((NT & 1) != 0) {
                    hipLaunchKernelGGL((gemmrow_kernel<T16, PRX_ACT_MUL_RELUMASK, RES, NT, KSM>), dim3(grid), dim3(GR_WAVES * 64), 0, s, a, ksteps, nslab, row_tiles, nchunks);
                    return true;
                }
                return false;
            default: return false;
        }
    } else if constexpr ((NT & 1) == 0) {
        switch (a.d.act) {
            GR_CASE(PRX_ACT_NONE) GR_CASE(PRX_ACT_RELU) GR_CASE(PRX_ACT_RELUMASK_POST)
            default: return false;
        }
    } else {
        switch (a.d.act) {
            GR_CASE(PRX_ACT_NONE) GR_CASE(PRX_ACT_RELU) GR_CASE(PRX_ACT_RELUMASK_POST)
            default: return false;
        }
    }
    return false;
#undef GR_CASE
}
// K in (320, 640]: 80-column slabs only (the weights of a wider slab do not fit the LDS)
template <typename T16, int RES_ON>
inline bool launch_slab80_k640(const GemmArgs& a, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    return a.d.resid != nullptr ? launch_instance<T16, RES_ON, 5, 20>(a, ksteps, nslab, row_tiles, nchunks, grid, s)
                                : launch_instance<T16, 0, 5, 20>(a, ksteps, nslab, row_tiles, nchunks, grid, s);
}
template <typename T16, int RES_ON, int KSM>
inline bool launch_slab(const GemmArgs& a, int nt, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    const bool res = a.d.resid != nullptr;
    if (nt == 10) return res ? launch_instance<T16, RES_ON, 10, KSM>(a, ksteps, nslab, row_tiles, nchunks, grid, s)
                             : launch_instance<T16, 0, 10, KSM>(a, ksteps, nslab, row_tiles, nchunks, grid, s);
    if (nt == 8) return res ? launch_instance<T16, RES_ON, 8, KSM>(a, ksteps, nslab, row_tiles, nchunks, grid, s)
                            : launch_instance<T16, 0, 8, KSM>(a, ksteps, nslab, row_tiles, nchunks, grid, s);
    if (nt == 5) return res ? launch_instance<T16, RES_ON, 5, KSM>(a, ksteps, nslab, row_tiles, nchunks, grid, s)
                            : launch_instance<T16, 0, 5, KSM>(a, ksteps, nslab, row_tiles, nchunks, grid, s);
    return false;
}
}  // namespace prx_gemmrow_dev

// one translation unit per (operand format, K range): gemmrow_h6.hip, gemmrow_h10.hip, gemmrow_b6.hip, gemmrow_b10.hip
bool prx_gemmrow_launch_h6(const prx_gemm_dev::GemmArgs& a, int nt, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s);
bool prx_gemmrow_launch_h10(const prx_gemm_dev::GemmArgs& a, int nt, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s);
bool prx_gemmrow_launch_b6(const prx_gemm_dev::GemmArgs& a, int nt, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s);
bool prx_gemmrow_launch_b10(const prx_gemm_dev::GemmArgs& a, int nt, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s);
bool prx_gemmrow_launch_h20(const prx_gemm_dev::GemmArgs& a, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s);
bool prx_gemmrow_launch_b20(const prx_gemm_dev::GemmArgs& a, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s);
