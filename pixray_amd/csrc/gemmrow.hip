// Row-streaming GEMM for skinny K (gfx950):  C[M, N] = epilogue(A[M, K] . B[N, K]^T),  K <= 192, M in the 10^5 .. 10^6 range.
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
#include "gemm.h"
#include "gemm_epi.h"
#include <type_traits>
#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace {
using namespace prx_gemm_dev;

constexpr int GR_WAVES = 8;          // waves per workgroup (all on one weight slab)
constexpr int GR_MAXKS = 6;          // K steps of 32: K <= 192
constexpr int GR_LD = GR_MAXKS * 32 + 8;     // LDS row stride of the slab at the largest K (elements)

std::atomic<long long> g_row_launches{0};

template <typename T16>
__device__ __forceinline__ f32x4 gr_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (std::is_same<T16, half_t>::value)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <typename T16>
__device__ __forceinline__ void gr_unpack(const bf16x8& v, float4& lo, float4& hi) {
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    const t16x8 t = __builtin_bit_cast(t16x8, v);
    lo = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    hi = make_float4((float)t[4], (float)t[5], (float)t[6], (float)t[7]);
}

// RES: 0 none, 1 fp32 residual, 2 residual in the 16-bit operand format (GemmDesc::row16 bit 0)
template <typename T16, int ACT, int RES, int NP>
__global__ __launch_bounds__(GR_WAVES * 64) void gemmrow_kernel(GemmArgs a, int ksteps, int nslab, int row_tiles, int nchunks) {
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    constexpr int NW = NP * 32;
    constexpr bool HAS_AUX = ACT == PRX_ACT_MUL_RELUMASK || ACT == PRX_ACT_RELUMASK_POST;
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

    // ---- the slab: LDS row r = (pair q, accumulator t, MFMA row rr) holds weight row n0 + 32 q + 8 (rr / 4) + 4 t + rr % 4 ----------
    {
        const int kch = ksteps * 4;              // 16-byte chunks per row, zero beyond K
        const bf16_t* Bg = reinterpret_cast<const bf16_t*>(d.B);
        for (int i = tid; i < NW * kch; i += GR_WAVES * 64) {
            const int r = i / kch, c = i - r * kch;
            const int q = r >> 5, t = (r >> 4) & 1, rr = r & 15;
            const int n = n0 + q * 32 + (rr >> 2) * 8 + 4 * t + (rr & 3);
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
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
    const bf16_t* wrow = Bs + m_l * ld + kg * 8;         // + (q * 32 + t * 16) * ld + ks * 32
    const int ccol = kg * 8;                             // this lane's 8 columns inside a pair

    bf16x8 acur[GR_MAXKS], anxt[GR_MAXKS];
    auto load_a = [&](int t, bf16x8 (&fr)[GR_MAXKS]) {
        int row = t * 16 + m_l;
        row = row < d.M ? row : d.M - 1;
        const bf16_t* p = Ag + (size_t)row * d.lda + kg * 8;
#pragma unroll
        for (int ks = 0; ks < GR_MAXKS; ++ks) {
            if (ks < ksteps) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
                if (ks * 32 + kg * 8 < d.K) v = *reinterpret_cast<const bf16x8*>(p + ks * 32);
                fr[ks] = v;
            }
        }
    };

    int t = t_begin + wave;
    if (t < t_end) load_a(t, acur);
    for (; t < t_end; t += GR_WAVES) {
        const int row = t * 16 + m_l;
        const bool live = row < d.M;
        const int rowc = live ? row : d.M - 1;
        // epilogue operands of this tile, then the activations of the next one
        bf16x8 aux16[HAS_AUX ? NP : 1];
        bf16x8 res16[RES == 2 ? NP : 1];
        float4 res32[RES == 1 ? NP : 1][2];
        if constexpr (HAS_AUX) {
            const T16* p = reinterpret_cast<const T16*>(d.aux) + (size_t)rowc * d.ldaux + n0 + ccol;
#pragma unroll
            for (int q = 0; q < NP; ++q) aux16[q] = *reinterpret_cast<const bf16x8*>(p + q * 32);
        }
        if constexpr (RES == 2) {
            const T16* p = reinterpret_cast<const T16*>(d.resid) + (size_t)rowc * d.ldr + n0 + ccol;
#pragma unroll
            for (int q = 0; q < NP; ++q) res16[q] = *reinterpret_cast<const bf16x8*>(p + q * 32);
        }
        if constexpr (RES == 1) {
            const float* p = d.resid + (size_t)rowc * d.ldr + n0 + ccol;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                res32[q][0] = *reinterpret_cast<const float4*>(p + q * 32);
                res32[q][1] = *reinterpret_cast<const float4*>(p + q * 32 + 4);
            }
        }
        if (t + GR_WAVES < t_end) load_a(t + GR_WAVES, anxt);

        f32x4 acc[NP][2];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][h][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < GR_MAXKS; ++ks) {
            if (ks < ksteps) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wrow + (q * 32 + h * 16) * ld + ks * 32);
                        acc[q][h] = gr_mfma<T16>(w, acur[ks], acc[q][h]);
                    }
            }
        }

        // ---- epilogue: 8 consecutive columns per lane and pair -----------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int col = n0 + q * 32 + ccol;
            const float4 b0 = *reinterpret_cast<const float4*>(bias_s + q * 32 + ccol);
            const float4 b1 = *reinterpret_cast<const float4*>(bias_s + q * 32 + ccol + 4);
            float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, r0 = x0, r1 = x0, pre;
            if constexpr (HAS_AUX) gr_unpack<T16>(aux16[q], x0, x1);
            if constexpr (RES == 2) gr_unpack<T16>(res16[q], r0, r1);
            if constexpr (RES == 1) { r0 = res32[q][0]; r1 = res32[q][1]; }
            const float ax0[4] = {x0.x, x0.y, x0.z, x0.w}, ax1[4] = {x1.x, x1.y, x1.z, x1.w};
            const float4 v0 = epilogue_math4<T16>(ACT, alpha, make_float4(acc[q][0][0], acc[q][0][1], acc[q][0][2], acc[q][0][3]), b0, 0.f,
                                                  ax0, RES != 0, r0, pre);
            const float4 v1 = epilogue_math4<T16>(ACT, alpha, make_float4(acc[q][1][0], acc[q][1][1], acc[q][1][2], acc[q][1][3]), b1, 0.f,
                                                  ax1, RES != 0, r1, pre);
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
#pragma unroll
        for (int ks = 0; ks < GR_MAXKS; ++ks) acur[ks] = anxt[ks];
    }
}

int slab_pairs(int N) { return N % 160 == 0 ? 5 : (N % 128 == 0 ? 4 : 0); }

template <typename T16, int ACT, int RES>
void launch_np(const GemmArgs& a, int np, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    if (np == 5) hipLaunchKernelGGL((gemmrow_kernel<T16, ACT, RES, 5>), dim3(grid), dim3(GR_WAVES * 64), 0, s, a, ksteps, nslab, row_tiles, nchunks);
    else hipLaunchKernelGGL((gemmrow_kernel<T16, ACT, RES, 4>), dim3(grid), dim3(GR_WAVES * 64), 0, s, a, ksteps, nslab, row_tiles, nchunks);
}
template <typename T16, int RES>
void launch_act(const GemmArgs& a, int np, int ksteps, int nslab, int row_tiles, int nchunks, int grid, hipStream_t s) {
    switch (a.d.act) {
        case PRX_ACT_NONE: launch_np<T16, PRX_ACT_NONE, RES>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s); break;
        case PRX_ACT_RELU: launch_np<T16, PRX_ACT_RELU, RES>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s); break;
        default:
            if constexpr (RES != 0) launch_np<T16, PRX_ACT_RELUMASK_POST, RES>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s);
            break;
    }
}
}  // namespace

// The residual kinds that exist as kernels: the half mode's 16-bit stream (lean layout) and bf16's fp32 residual; plus "none" for both.
bool prx_gemmrow_eligible(const GemmDesc& d) {
    static const int on = [] { const char* e = getenv("PRX_GEMM_ROWK"); return e ? atoi(e) : 1; }();
    if (!on || d.f32 || d.a_is_f32 || d.a_mode != PRX_A_ROWMAJOR) return false;
    if (d.K > GR_MAXKS * 32 || d.K % 8 != 0 || slab_pairs(d.N) == 0) return false;
    if ((long long)d.M * d.N < (5ll << 20)) return false;                    // enough 16 x NW wave tiles for 2 048+ waves
    if (d.bias_m || d.gn_stats || d.gnb_x || d.out_bf16_pre) return false;
    if (d.act != PRX_ACT_NONE && d.act != PRX_ACT_RELU && d.act != PRX_ACT_RELUMASK_POST) return false;
    if (d.act == PRX_ACT_RELUMASK_POST && !(d.resid && d.aux)) return false;
    if (d.resid && ((d.row16 & 1) != 0) != (d.h16 != 0)) return false;       // half: 16-bit residual streams; bf16: fp32 residuals
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al16(d.A) || !al16(d.B) || d.lda % 8 != 0 || d.ldb % 8 != 0) return false;
    if (d.resid && (!al16(d.resid) || d.ldr % 8 != 0)) return false;
    if (d.aux && d.act == PRX_ACT_RELUMASK_POST && (!al16(d.aux) || d.ldaux % 8 != 0)) return false;
    if (d.out_f32 && (!al16(d.out_f32) || d.ldc_f32 % 4 != 0)) return false;
    if (d.out_bf16 && (!al16(d.out_bf16) || d.ldc_bf16 % 8 != 0)) return false;
    return d.out_f32 || d.out_bf16;
}

void prx_gemmrow_launch(const prx_gemm_dev::GemmArgs& a, int n_cu, hipStream_t s) {
    const GemmDesc& d = a.d;
    const int np = slab_pairs(d.N), nw = np * 32, nslab = d.N / nw;
    const int ksteps = (d.K + 31) / 32;
    const int row_tiles = (d.M + 15) / 16;
    // one workgroup of 8 waves per CU (the kernels hold 134 - 218 registers: two waves per SIMD); the grid is a whole number of
    // (8 XCDs x nslab) groups
    static const int per_cu = [] { const char* e = getenv("PRX_GEMM_ROWK_WGS"); return e ? std::max(1, atoi(e)) : 1; }();
    const int group = 8 * nslab;
    int grid = std::max(1, (per_cu * (n_cu > 0 ? n_cu : 256)) / group) * group;
    const int nchunks = (grid / group) * 8;
    if (d.h16) {
        if (d.resid) launch_act<half_t, 2>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s);
        else launch_act<half_t, 0>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s);
    } else {
        if (d.resid) launch_act<bf16_t, 1>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s);
        else launch_act<bf16_t, 0>(a, np, ksteps, nslab, row_tiles, nchunks, grid, s);
    }
    g_row_launches.fetch_add(1, std::memory_order_relaxed);
}

long long prx_gemmrow_launches() { return g_row_launches.load(std::memory_order_relaxed); }
extern "C" long long prx_gemm_row_launches(void) { return prx_gemmrow_launches(); }
