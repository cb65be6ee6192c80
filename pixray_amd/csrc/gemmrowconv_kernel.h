// Row-streaming IMPLICIT 3x3 convolution for few channels (gfx950): out[pixel, N] = epilogue(sum over 9 taps x Cin of x . w), Cin and N
// in {40, 80}, or both 160 -- the ModifiedResNet runner's stem convolutions and stage-1 / stage-2 conv2 (RN50x4 at 128 cutouts of 288^2:
// 2.65 M / 0.66 M / 0.17 M pixels, K = 360 / 720 / 1440) and their dgrads.  With Cin % 64 != 0 the tiled kernels gather these operands 16 bytes at a time with a
// tap decode per chunk and fill 128-wide tiles 31 - 62 %: 230 - 290 TFLOP/s.  Same scheme as gemmrow_kernel.h:
//
//   * ALL the weights (N x 9 Cin: 29 - 115 KB) stay in LDS for the life of the (persistent, one per CU) workgroup;
//   * a wave owns 16 consecutive pixels x N channels; per K step of 32 a lane loads the 16 bytes (8 channels of one tap) its MFMA
//     operand slot needs straight from the NHWC map (taps outside the image: zeros), all K steps of a tile in flight at once;
//   * transposed MFMA: 8 (lone tile: 4) consecutive output channels per lane, epilogue from registers -- bias + ReLU forward, the
//     saved activation's ReLU mask backward.
//
// The K index is tap * Cin + ci (resnet.hip rn_pack_conv3x3_kernel: forward and flipped dgrad weights share it), so 16-byte chunk c
// of a K row is tap c / (Cin / 8), channels 8 (c % (Cin / 8)) ..: never straddling a tap.
#pragma once
#include "gemmrow_kernel.h"

namespace prx_gemmrow_dev {

// NT: 16-column tiles of a slab (5: 80 columns as 2 pairs + a lone tile; 3: 40 columns as 1 pair + the first half of a lone tile)
// CIN8: Cin / 8
// RING: 0 = all K steps of a tile live in registers (9 Cin <= 736); else the K loop runs over a ring of RING operand fragments, each
//       refilled as soon as its MFMAs are issued (Cin = 160: 45 K steps; the weights of a 40-column slab are 139 KB, N = 160 is four
//       slabs whose workgroups share an XCD and so the activations in its L2)
// WAVES: waves per workgroup (16 for the ring kernels: ~110 registers, and the L2 -> CU path they are bound by wants the loads of
//       more waves in flight)
template <typename T16, int ACT, int NT, int CIN8, int RING, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemmrowconv_kernel(GemmArgs a, int row_tiles, int nchunks, int nslab) {
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    typedef __attribute__((ext_vector_type(4))) T16 t16x4;
    constexpr int NW = NT * 16, NP = NT / 2, NV = NT == 5 ? 80 : 40;      // slab rows in LDS; pairs; columns that exist
    constexpr int KCH = 9 * CIN8;                          // 16-byte chunks of a K row
    constexpr int KS0 = (KCH + 3) / 4, KS = RING == 0 ? KS0 : (KS0 + RING - 1) / RING * RING;      // K steps of 32 (a whole number of rings: zero chunks)
    constexpr int LD = KS * 32 + 8;
    constexpr bool HAS_AUX = ACT == PRX_ACT_MUL_RELUMASK;
    static_assert((NT & 1) == 1 && ((LD / 2) / 4) % 2 == 1, "pairs + a lone tile; conflict-free slab stride");
    static_assert(RING == 0 || KS % RING == 0, "the ring divides the K steps");
    __shared__ __attribute__((aligned(16))) bf16_t Bs[NW * LD];
    __shared__ __attribute__((aligned(16))) float bias_s[NW];
    const GemmDesc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (XCD, column slab, pixel chunk): the slabs of a chunk are 8 workgroup ids apart, i.e. on one XCD
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slab = idx % nslab, chunk = (idx / nslab) * 8 + xcd;
    const int n0 = slab * NV;

    // ---- the weights: LDS row r = (pair q, accumulator h, MFMA row rr) holds weight row 32 q + 8 (rr / 4) + 4 h + rr % 4, the lone
    // tile's rows in their natural order; rows >= N and chunks >= 9 Cin / 8 are zeros
    {
        const bf16_t* Bg = reinterpret_cast<const bf16_t*>(d.B);
        for (int i = tid; i < NW * KS * 4; i += WAVES * 64) {
            const int r = i / (KS * 4), c = i - r * (KS * 4);
            const int q = r >> 5, h = (r >> 4) & 1, rr = r & 15;
            const int n = r >= NP * 32 ? r : q * 32 + (rr >> 2) * 8 + 4 * h + (rr & 3);
            bf16x8 v = gr_zero8();
            if (c < KCH && n < NV) v = *reinterpret_cast<const bf16x8*>(Bg + (size_t)(n0 + n) * d.ldb + c * 8);
            *reinterpret_cast<bf16x8*>(Bs + r * LD + c * 8) = v;
        }
        for (int i = tid; i < NW; i += WAVES * 64) bias_s[i] = (d.bias_n && i < NV) ? d.bias_n[n0 + i] : 0.f;
    }
    __syncthreads();

    const int m_l = lane & 15, kg = lane >> 4;
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    const int per_chunk = (row_tiles + nchunks - 1) / nchunks;
    const int t_begin = chunk * per_chunk;
    const int t_end = t_begin + per_chunk < row_tiles ? t_begin + per_chunk : row_tiles;
    const char* Ag = reinterpret_cast<const char*>(d.A);
    const bf16_t* wrow = Bs + m_l * LD + kg * 8;
    const int ccol = kg * 8, tcol = NP * 32 + kg * 4;
    const bool tail_ok = tcol < NV;                       // 40-column slabs: the lone tile's columns 40 .. 47 do not exist
    const int H = d.H, W = d.W, pstride = d.lda * 2;       // bytes between pixels

    // per K step: this lane's chunk as (byte offset from the centre pixel) | tap  (tap 15: beyond K)
    auto chunk_code = [&](int ks) {
        const int c = ks * 4 + kg;
        const int tap = c / CIN8, ch = (c - tap * CIN8) * 8;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        return c < KCH ? (((dy * W + dx) * pstride + ch * 2) | tap) : 15;
    };
    int koff[RING == 0 ? KS : 1];
    if constexpr (RING == 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) koff[ks] = chunk_code(ks);
    }

    for (int t = t_begin + wave; t < t_end; t += WAVES) {
        const int pix = t * 16 + m_l;
        const bool live = pix < d.M;
        const int pc = live ? pix : d.M - 1;
        const int x = pc % W, y = (pc / W) % H;
        unsigned vm = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            vm |= (unsigned)(yy >= 0 && yy < H && xx >= 0 && xx < W) << tap;
        }
        const char* pa = Ag + (size_t)pc * pstride;
        auto load_frag = [&](int ko) {
            bf16x8 v = gr_zero8();
            if ((vm >> (ko & 15)) & 1u) v = *reinterpret_cast<const bf16x8*>(pa + (ko & ~15));
            return v;
        };
        bf16x8 afr[RING == 0 ? KS : RING];
        if constexpr (RING == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) afr[ks] = load_frag(koff[ks]);
        } else {
#pragma unroll
            for (int j = 0; j < RING; ++j) afr[j] = load_frag(chunk_code(j));
        }
        bf16x8 aux16[HAS_AUX ? NP : 1];
        bf16x4 auxt;
        if constexpr (HAS_AUX) {
            const T16* p = reinterpret_cast<const T16*>(d.aux) + (size_t)pc * d.ldaux + n0;
#pragma unroll
            for (int q = 0; q < NP; ++q) aux16[q] = *reinterpret_cast<const bf16x8*>(p + q * 32 + ccol);
            if (tail_ok) auxt = *reinterpret_cast<const bf16x4*>(p + tcol);
        }

        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        if constexpr (RING == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8*>(wrow + (j * 16) * LD + ks * 32);
                    acc[j] = gr_mfma<T16>(w, afr[ks], acc[j]);
                }
                // keep the weight reads of later K steps behind this step's MFMAs: hoisted as far as the scheduler likes they overflow
                // the register file (all 9 Cin / 32 activation fragments of the tile are live here)
                if ((ks & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int g = 0; g < KS; g += RING) {
#pragma unroll
                for (int r = 0; r < RING; ++r) {
                    const int ks = g + r;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wrow + (j * 16) * LD + ks * 32);
                        acc[j] = gr_mfma<T16>(w, afr[r], acc[j]);
                    }
                    if (ks + RING < KS) afr[r] = load_frag(chunk_code(ks + RING));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        float4 pre;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int col = n0 + q * 32 + ccol;
            const float4 b0 = *reinterpret_cast<const float4*>(bias_s + q * 32 + ccol);
            const float4 b1 = *reinterpret_cast<const float4*>(bias_s + q * 32 + ccol + 4);
            float4 x0 = z4, x1 = z4;
            if constexpr (HAS_AUX) gr_unpack<T16>(aux16[q], x0, x1);
            const float ax0[4] = {x0.x, x0.y, x0.z, x0.w}, ax1[4] = {x1.x, x1.y, x1.z, x1.w};
            const f32x4 &c0 = acc[2 * q], &c1 = acc[2 * q + 1];
            const float4 v0 = epilogue_math4<T16>(ACT, alpha, make_float4(c0[0], c0[1], c0[2], c0[3]), b0, 0.f, ax0, false, z4, pre);
            const float4 v1 = epilogue_math4<T16>(ACT, alpha, make_float4(c1[0], c1[1], c1[2], c1[3]), b1, 0.f, ax1, false, z4, pre);
            if (live) {
                if (d.out_f32) {
                    float* o = d.out_f32 + (size_t)pix * d.ldc_f32 + col;
                    *reinterpret_cast<float4*>(o) = v0;
                    *reinterpret_cast<float4*>(o + 4) = v1;
                }
                if (d.out_bf16) {
                    t16x8 o;
                    o[0] = op_cvt<T16>(v0.x); o[1] = op_cvt<T16>(v0.y); o[2] = op_cvt<T16>(v0.z); o[3] = op_cvt<T16>(v0.w);
                    o[4] = op_cvt<T16>(v1.x); o[5] = op_cvt<T16>(v1.y); o[6] = op_cvt<T16>(v1.z); o[7] = op_cvt<T16>(v1.w);
                    *reinterpret_cast<t16x8*>(reinterpret_cast<T16*>(d.out_bf16) + (size_t)pix * d.ldc_bf16 + col) = o;
                }
            }
        }
        if (tail_ok) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias_s + tcol);
            float4 x0 = z4;
            if constexpr (HAS_AUX) x0 = gr_unpack4<T16>(auxt);
            const float ax0[4] = {x0.x, x0.y, x0.z, x0.w};
            const f32x4& c0 = acc[NT - 1];
            const float4 v0 = epilogue_math4<T16>(ACT, alpha, make_float4(c0[0], c0[1], c0[2], c0[3]), b0, 0.f, ax0, false, z4, pre);
            if (live) {
                if (d.out_f32) *reinterpret_cast<float4*>(d.out_f32 + (size_t)pix * d.ldc_f32 + n0 + tcol) = v0;
                if (d.out_bf16) {
                    t16x4 o;
                    o[0] = op_cvt<T16>(v0.x); o[1] = op_cvt<T16>(v0.y); o[2] = op_cvt<T16>(v0.z); o[3] = op_cvt<T16>(v0.w);
                    *reinterpret_cast<t16x4*>(reinterpret_cast<T16*>(d.out_bf16) + (size_t)pix * d.ldc_bf16 + n0 + tcol) = o;
                }
            }
        }
    }
}

template <typename T16, int ACT>
inline bool launch_conv_act(const GemmArgs& a, int row_tiles, int n_cu, hipStream_t s) {
    const int N = a.d.N, Cin = a.d.Cin;
    // one persistent workgroup per CU; the grid is a whole number of (8 XCDs x nslab) groups
#define GRC_CASE(N_, CIN_, NT_, RING_, WAVES_)                                                                                            \
    if (N == N_ && Cin == CIN_) {                                                                                                \
        const int nslab = N_ / (NT_ == 5 ? 80 : 40), group = 8 * nslab;                                                          \
        const int grid = (n_cu / group > 0 ? n_cu / group : 1) * group, nchunks = (grid / group) * 8;                            \
        hipLaunchKernelGGL((gemmrowconv_kernel<T16, ACT, NT_, CIN_ / 8, RING_, WAVES_>), dim3(grid), dim3(WAVES_ * 64), 0, s, a, row_tiles, nchunks, nslab); \
        return true;                                                                                                             \
    }
    // (measured: the ring + 16 waves loses on Cin = 80 -- N = 80: 125 -> 133 us, N = 40: 371 -> 466 us -- where a tile's 23 fragments fit the
    // registers; 16 waves of the register form win on the stem's 40 -> 40: 239 -> 176 us)
    GRC_CASE(80, 80, 5, 0, 8) GRC_CASE(80, 40, 5, 0, 8) GRC_CASE(40, 80, 3, 0, 8) GRC_CASE(40, 40, 3, 0, 16) GRC_CASE(160, 160, 3, 9, 16)
#undef GRC_CASE
    return false;
}
template <typename T16>
inline bool launch_conv(const GemmArgs& a, int row_tiles, int n_cu, hipStream_t s) {
    if (a.d.act == PRX_ACT_RELU) return launch_conv_act<T16, PRX_ACT_RELU>(a, row_tiles, n_cu, s);
    if (a.d.act == PRX_ACT_MUL_RELUMASK) return launch_conv_act<T16, PRX_ACT_MUL_RELUMASK>(a, row_tiles, n_cu, s);
    return false;
}
}  // namespace prx_gemmrow_dev

bool prx_gemmrowconv_launch_h(const prx_gemm_dev::GemmArgs& a, int row_tiles, int n_cu, hipStream_t s);
bool prx_gemmrowconv_launch_b(const prx_gemm_dev::GemmArgs& a, int row_tiles, int n_cu, hipStream_t s);
