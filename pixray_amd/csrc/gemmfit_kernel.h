// "Fit-tile" member of the GEMM engine (gemm.h): 16-bit operands, tiles whose COUNT matches the chip (one workgroup per CU),
// row-major products and implicit 3x3 convolutions.
//
// Why: the hot products of the headline iteration are small.  The ViT-B/32 tower at 64 cutouts is M = 64 * 50 = 3200 token
// rows; the VQGAN decoder is batch 1, M = 256 ... 65 536 pixels by N = 128 ... 512 channels.  With the power-of-two tiles of
// gemm.hip such a product has 150 / 300 / 600 tiles for 256 CUs, or 8 - 64 tiles that need split-K across workgroups plus a
// reduce launch: either 40 % of the chip idles, or a second mostly empty round runs, or partial sums make a round trip through
// HBM; and the small tiles that do fill the chip move twice the L2->LDS bytes per flop (64 B/clk/CU on that path).  Here the
// tile is chosen so that the grid is ~240 - 256 workgroups, and when that makes the tile small the K loop is split over the
// wave groups of ONE workgroup (KS) and summed through LDS -- no workspace, no second launch:
//     ViT-B/32, M = 3200:  N = 3072 -> 160 x 256 (240 tiles)   N = 2304 -> 160 x 192 (240)   N = 768 -> 80 x 128, KS 2 (240)
//     decoder 256^2 level: 65 536 x 128 -> 256 x 128 (256)      128^2: 128 x 128 / 128 x 64 KS 2     64^2: 64 x 64 KS 2
//                  32^2:   32 x 64 KS 4 / 16 x 64 KS 4           16^2:  256 x 512 x 4608 -> 16 x 32, KS 8 (256 tiles)
// The same tiles fit the sharded batches (32 / 16 / 8 cutouts: M = 1600 / 800 / 400 are multiples of 80).
//
// Structure: 8 waves (two per SIMD), 16x16x32 MFMAs, wave tile (16 FM) x (16 FN).  Operands HBM -> LDS by
// global_load_lds_dwordx4 into a 3-deep ring of stages (a stage = KS consecutive K tiles of BK = 64, one per wave group) with
// counted s_waitcnt vmcnt and one raw s_barrier per stage: two stages of DMA stay in flight across the barrier.  The two halves
// of the workgroup issue their DMA at opposite ends of a stage (staggered wave groups, see the main loop).  Swizzle as in
// gemm.hip (LDS chunk c of row r holds source chunk c ^ ((r >> 1) & 7), applied on the DMA source address and on the fragment
// read; conflict-free for the 16x16x32 operand layout too).  Rows >= M / columns >= N are clamped on the source side (row-major)
// or read the zero page (convolution: as its padding does).  Implicit convolution (CONV): Cin % 64 == 0, so a K tile lies inside
// ONE filter tap, which is wave-uniform per DMA piece; the per-lane gather address is row term[ky] + column term[kx] from six
// values computed once (gemm.hip's C64 scheme), optionally through the fused nearest-2x upsample.
// Epilogue: the engine's (gemm_epi.h) on 8 consecutive columns per lane, staged per wave through LDS, 16-byte accesses, its HBM
// reads prefetched; the next GroupNorm's sums or a GroupNorm-backward's sums (GemmDesc::gn_stats / gnb_*) in a fixed summation
// order up to the final fp64 atomics.
//
// Requirements (prx_gemmfit_eligible): 16-bit A (row-major, or NHWC with Cin % 64 == 0 and up in {0, 1}), K % (64 KS) == 0,
// N % 8 == 0 with 16-byte-friendly epilogue operands, at most one of {residual, aux, GroupNorm-backward input} (so not
// PRX_ACT_RELUMASK_POST, which reads a residual and a mask), no split-K across workgroups.
#pragma once
#include "gemm_epi.h"
#include <type_traits>
#include <algorithm>
#include <stdlib.h>
#include <string>
#include <string.h>
#include <stdio.h>

const bf16_t* prx_gemm_zero_page();       // gemm.hip: 256 bytes of zeros on the current device
// the kernels with a compile-time epilogue, one translation unit per tile family (build time): true when the unit has a
// kernel for (tile, epilogue kind) and launched it
int prx_gemmfit_epi_kind(const GemmDesc& d);
bool prx_gemmfit_launch_spec_tower(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp);
bool prx_gemmfit_launch_spec_dec_a(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp);
bool prx_gemmfit_launch_spec_dec_b(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp);
bool prx_gemmfit_launch_spec_dec_c(const prx_gemm_dev::GemmArgs& a, int bm, int bn, int epi, dim3 grid, hipStream_t s, const bf16_t* zp);
bool prx_gemmfit_launch_f32(const prx_gemm_dev::GemmArgs& a, int bm, int bn, dim3 grid, hipStream_t s, const bf16_t* zp);      // gemmfit_f32.hip

namespace {
using namespace prx_gemm_dev;

constexpr int FIT_BK = 64;
constexpr int FIT_STAGES = 3;
typedef const __attribute__((address_space(1))) void* fit_gptr;
typedef __attribute__((address_space(3))) void* fit_lptr;

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

// Diagnostic build only (-DPRX_FIT_TRACE, tools/fit_trace.py): every wave keeps s_memtime stamps of its phases in scalar
// registers and writes them to the caller's workspace at the very end ([workgroup][wave][8] 64-bit ticks): 0 entry, 1 DMA
// coordinates ready, 2 first stage landed, 3 K loop done, 4 K groups summed, 5 epilogue issued, 6 its stores acknowledged.
#ifdef PRX_FIT_TRACE
#define FIT_TRACE_ARG , unsigned long long (&tr)[8]
#define FIT_TRACE_PASS , tr
#define FIT_TRACE(slot) do { tr[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FIT_TRACE_ARG
#define FIT_TRACE_PASS
#define FIT_TRACE(slot) do {} while (0)
#endif

template <typename T16>
__device__ __forceinline__ f32x4 fit_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (std::is_same<T16, half_t>::value)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One fp32 fragment (16 bytes of a ring row) through a __restrict__ parameter: inlining gives the read a scoped-noalias tag, which is
// what tells the compiler's wait-count insertion that it does not alias the LDS-DMA writes in flight (the ring discipline
// guarantees it: a stage is read only after the counted wait + barrier that retired its DMA).  Without the tag every fragment
// read of the fp32 kernels was preceded by s_waitcnt vmcnt(0) -- the stage just issued awaited in full, each stage.
__device__ __forceinline__ float4 fit_frag_f32(const bf16_t* __restrict__ p) {
    return __builtin_bit_cast(float4, *reinterpret_cast<const bf16x8*>(p));
}

// GroupNorm-backward sums on prefetched x (gemm_epi.h gnb_accum with the load taken out)
template <bool FAST = false>
__device__ __forceinline__ void fit_gnb_accum(const GemmDesc& d, const GnbConst& c, const float4& x, const float4& o, float& s0, float& s1) {
    const float xv[4] = {x.x, x.y, x.z, x.w}, gv[4] = {o.x, o.y, o.z, o.w};
    const float gav[4] = {c.ga.x, c.ga.y, c.ga.z, c.ga.w}, bev[4] = {c.be.x, c.be.y, c.be.z, c.be.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - c.mean) * c.rstd;
        float gy = gv[i];
        if (d.gnb_swish) {
            const float y = xh * gav[i] + bev[i];
            const float sg = FAST ? __builtin_amdgcn_rcpf(1.f + __expf(-y)) : sigmoidf_(y);     // FAST: v_exp + v_rcp (the specialised epilogues)
            gy *= sg * (1.f + y * (1.f - sg));
        }
        const float dxh = gy * gav[i];
        s0 += dxh;
        s1 += dxh * xh;
    }
}

// The part of a fit kernel behind its K loop, shared by the ring kernel and the streaming kernel: the K groups' partial sums
// through LDS, then the engine's epilogue.  `fl`: the workgroup's LDS as floats (fit_scratch_floats<...>() of them), free of
// any other use; every wave of the workgroup calls this.
template <int WGM, int WGN, int FM, int FN, int KS>
constexpr int fit_scratch_floats() {        // K-group dump + two staging slabs per wave (the specialised epilogues double-buffer) + GroupNorm partials
    return (KS > 1 ? WGM * WGN * KS * FM * FN * 256 : 0) + 2 * WGM * WGN * KS * 16 * (16 * FN + 4) + KS * WGM * (16 * FN * WGN / 2);
}

// ---- K groups: every wave dumps the 16-row slabs it does not OWN (slab i belongs to group i % KS); the owner adds the
// other groups' partials in group order.  One barrier, no workspace, a fixed summation order.
template <int WGM, int WGN, int FM, int FN, int KS>
__device__ __forceinline__ void fit_ksum(float* const fl, f32x4 (&acc)[FM][FN], int kg, int wt, int lane) {
    constexpr int NWT = WGM * WGN;
    if constexpr (KS > 1) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if (i % KS == kg) continue;
#pragma unroll
            for (int j = 0; j < FN; ++j)
                *reinterpret_cast<f32x4*>(fl + ((kg * NWT + wt) * FM * FN + i * FN + j) * 256 + lane * 4) = acc[i][j];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if (i % KS != kg) continue;
#pragma unroll
            for (int g = 0; g < KS; ++g) {
                if (g == kg) continue;
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] += *reinterpret_cast<const f32x4*>(fl + ((g * NWT + wt) * FM * FN + i * FN + j) * 256 + lane * 4);
            }
        }
    }
}

// ---- GroupNorm sums of a block tile: lanes of one column quad by a fixed butterfly, the waves that share the columns through
// one LDS slot each summed in a fixed order, then one fp64 atomic per group and moment (the only order-dependent step, ~1e-16
// relative).  gpart: [KS * WGM][BN / 4 quads][2] floats of LDS nobody else uses.
template <int WGM, int WGN, int FN, int KS>
__device__ __forceinline__ void fit_stats_tail(const GemmDesc& d, float* const gpart, float gsa0, float gsa1, float gsb0, float gsb1,
                                               int tn, int kg, int wm, int wn, int lane) {
    constexpr int TN = 16 * FN, BN = TN * WGN, LPR = TN / 8;
    if constexpr ((LPR & (LPR - 1)) == 0) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            gsa0 += __shfl_xor(gsa0, o, 64); gsa1 += __shfl_xor(gsa1, o, 64);
            gsb0 += __shfl_xor(gsb0, o, 64); gsb1 += __shfl_xor(gsb1, o, 64);
        }
        if (lane < LPR) {
            float* const w_ = gpart + ((kg * WGM + wm) * (BN / 4) + wn * (TN / 4) + lane * 2) * 2;
            w_[0] = gsa0; w_[1] = gsa1; w_[2] = gsb0; w_[3] = gsb1;
        }
        __syncthreads();
        const int qpg = d.gn_gs >> 2;                  // quads per group
        const int ngrp = BN / d.gn_gs;                 // groups covered by this block tile
        if (tid < ngrp * 2) {
            const int gl = tid >> 1, mom = tid & 1;
            const int gcol = tn * BN + gl * d.gn_gs;
            if (gcol < d.N) {
                double a2 = 0.0;
                for (int w2 = 0; w2 < KS * WGM; ++w2)
                    for (int q = 0; q < qpg; ++q) a2 += (double)gpart[(w2 * (BN / 4) + gl * qpg + q) * 2 + mom];
                atomicAdd(&d.gn_stats[(size_t)(gcol / d.gn_gs) * 2 + mom], a2);
            }
        }
    }
}
template <int WGM, int WGN, int FM, int FN, int KS, typename T16>
__device__ __forceinline__ void fit_finish(const GemmArgs& p, float* const fl, f32x4 (&acc)[FM][FN], int tm, int tn, int wave, int kg,
                                           int wt, int wm, int wn, int lane FIT_TRACE_ARG) {
    constexpr int NWT = WGM * WGN, NW = NWT * KS;
    constexpr int BM = 16 * FM * WGM, BN = 16 * FN * WGN, TN = 16 * FN;
    const GemmDesc& d = p.d;
    const int tid = threadIdx.x;
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int DUMP = KS > 1 ? NW * FM * FN * 256 : 0;          // floats
    fit_ksum<WGM, WGN, FM, FN, KS>(fl, acc, kg, wt, lane);

    FIT_TRACE(4);
    // ---- epilogue: per wave, one 16-row slab at a time through a private LDS slab; 8 consecutive columns per lane, so the
    // 16-bit outputs leave as 16-byte stores (the store tail of a one-round kernel is bound by store INSTRUCTIONS:
    // cdna_hip_programming.md T21).  Everything the epilogue READS from HBM (residual, aux or GroupNorm-input rows, bias) is
    // fetched for the whole wave tile BEFORE the first store: the compiler cannot move a load above an earlier store that may
    // alias it, and with one workgroup per CU nothing else hides a chain of dependent load round trips.
    typedef __attribute__((ext_vector_type(8))) T16 t16x8;
    constexpr int LDW = TN + 4;                 // padded row (floats), rows stay 16-byte aligned
    constexpr int LPR = TN / 8;                 // lanes per row
    constexpr int RPP = 64 / LPR;               // rows per pass
    constexpr int NPASS = (16 + RPP - 1) / RPP;
    float* const stage = fl + DUMP + wave * (16 * LDW);
    const int rbase = tm * BM + wm * (16 * FM), cbase = tn * BN + wn * TN;
    const int lr0 = lane / LPR, lc = (lane - lr0 * LPR) * 8;
    const int col = cbase + lc;
    const bool col_ok = lane < RPP * LPR && col < d.N;        // N % 8 == 0: a lane's 8 columns are in range together
    const int act = d.act;
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    const bool has_resid = d.resid != nullptr;
    const bool row16 = has_resid ? (d.row16 & 1) != 0 : (d.row16 & 2) != 0;      // the row operand (residual / GroupNorm input) is a 16-bit stream
    const bool need_aux = act == PRX_ACT_MUL_DQUICKGELU || act == PRX_ACT_MUL_RELUMASK || act == PRX_ACT_RELUMASK_POST;
    // the 80-row-granular tiles are the token-batch (ViT tower) tiles: no GroupNorm there, and their epilogue is compiled without the
    // statistics code (the fit kernels' epilogues are sensitive to every register and branch: profiles/r05_ln_fold/)
    constexpr bool STATS = BM % 80 != 0;
    const bool do_stats = STATS && d.gn_stats != nullptr;
    const bool gnb = do_stats && d.gnb_x != nullptr;
    float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
    if (d.bias_n && col_ok) {
        bias0 = *reinterpret_cast<const float4*>(d.bias_n + col);
        bias1 = *reinterpret_cast<const float4*>(d.bias_n + col + 4);
    }
    GnbConst gc0{}, gc1{};
    if (gnb && col_ok) { gc0 = gnb_load(d, col); gc1 = gnb_load(d, col + 4); }
    // slabs are handled FMC at a time: the prefetch of a chunk is (2 x 16 bytes + 1) registers per slab and pass on top of the
    // accumulators, and the wide wave tiles (16+ fragments) have no room for all of them at once
    constexpr int FMC = FM * FN > 16 ? 3 : FM;
    float gsa0 = 0.f, gsa1 = 0.f, gsb0 = 0.f, gsb1 = 0.f;    // GroupNorm sums of this lane's two column quads
#pragma unroll
    for (int i0 = 0; i0 < FM; i0 += FMC) {
        uint4 pf[FMC][NPASS][2];                // residual / GroupNorm input (2 x 16 bytes) or aux (16 bytes) of this lane's 8 columns
        float pbm[FMC][NPASS];
#pragma unroll
        for (int ic = 0; ic < FMC; ++ic) {
            const int i = i0 + ic;
            const bool keep = i < FM && (KS == 1 || i % KS == kg);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int lr = ps * RPP + lr0, row = rbase + i * 16 + lr;
                const bool ok = keep && col_ok && lr < 16 && row < d.M;
                pf[ic][ps][0] = pf[ic][ps][1] = uint4{0u, 0u, 0u, 0u};
                pbm[ic][ps] = 0.f;
                if (ok) {
                    if (row16) {            // 8 columns = one 16-byte load; widened to fp32 in place below
                        const T16* r_ = has_resid ? reinterpret_cast<const T16*>(d.resid) + (size_t)row * d.ldr + col
                                                  : reinterpret_cast<const T16*>(d.gnb_x) + (size_t)row * d.N + col;
                        pf[ic][ps][0] = *reinterpret_cast<const uint4*>(r_);
                    } else if (has_resid || gnb) {
                        const float* r_ = has_resid ? d.resid + (size_t)row * d.ldr + col : d.gnb_x + (size_t)row * d.N + col;
                        pf[ic][ps][0] = *reinterpret_cast<const uint4*>(r_);
                        pf[ic][ps][1] = *reinterpret_cast<const uint4*>(r_ + 4);
                    } else if (need_aux) {
                        pf[ic][ps][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T16*>(d.aux) + (size_t)row * d.ldaux + col);
                    }
                    if (d.bias_m) pbm[ic][ps] = d.bias_m[row];
                }
            }
        }
#pragma unroll
        for (int ic = 0; ic < FMC; ++ic) {
            const int i = i0 + ic;
            if (i >= FM) continue;
            if (KS > 1 && i % KS != kg) continue;
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(4 * kq + r) * LDW + j * 16 + l15] = acc[i][j][r];
            /*hipemu:wave_sync*/                    // the slab is exchanged between the lanes of ONE wave, which runs in lockstep (the CPU emulation of tools/hipemu synchronises its fibers at these markers)
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int lr = ps * RPP + lr0, row = rbase + i * 16 + lr;
                if (!(col_ok && lr < 16 && row < d.M)) continue;
                float4 v0 = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc]);
                float4 v1 = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc + 4]);
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
                float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, p0, p1;
                if (row16) {
                    const t16x8 rx = __builtin_bit_cast(t16x8, pf[ic][ps][0]);
                    pf[ic][ps][0] = __builtin_bit_cast(uint4, make_float4((float)rx[0], (float)rx[1], (float)rx[2], (float)rx[3]));
                    pf[ic][ps][1] = __builtin_bit_cast(uint4, make_float4((float)rx[4], (float)rx[5], (float)rx[6], (float)rx[7]));
                }
                if (has_resid) {
                    r0 = __builtin_bit_cast(float4, pf[ic][ps][0]);
                    r1 = __builtin_bit_cast(float4, pf[ic][ps][1]);
                } else if (need_aux) {
                    const t16x8 ax = __builtin_bit_cast(t16x8, pf[ic][ps][0]);
                    a0[0] = (float)ax[0]; a0[1] = (float)ax[1]; a0[2] = (float)ax[2]; a0[3] = (float)ax[3];
                    a1[0] = (float)ax[4]; a1[1] = (float)ax[5]; a1[2] = (float)ax[6]; a1[3] = (float)ax[7];
                }
                v0 = epilogue_math4<T16>(act, alpha, v0, bias0, pbm[ic][ps], a0, has_resid, r0, p0);
                v1 = epilogue_math4<T16>(act, alpha, v1, bias1, pbm[ic][ps], a1, has_resid, r1, p1);
                if (act == PRX_ACT_QUICKGELU && d.out_bf16_pre) {
                    t16x8 q;
                    q[0] = op_cvt<T16>(p0.x); q[1] = op_cvt<T16>(p0.y); q[2] = op_cvt<T16>(p0.z); q[3] = op_cvt<T16>(p0.w);
                    q[4] = op_cvt<T16>(p1.x); q[5] = op_cvt<T16>(p1.y); q[6] = op_cvt<T16>(p1.z); q[7] = op_cvt<T16>(p1.w);
                    *reinterpret_cast<t16x8*>(reinterpret_cast<T16*>(d.out_bf16_pre) + (size_t)row * d.ldc_bf16 + col) = q;
                }
                if (d.out_f32) {
                    float* o = d.out_f32 + (size_t)row * d.ldc_f32 + col;
                    *reinterpret_cast<float4*>(o) = v0;
                    *reinterpret_cast<float4*>(o + 4) = v1;
                }
                if (d.out_bf16) {
                    t16x8 q;
                    q[0] = op_cvt<T16>(v0.x); q[1] = op_cvt<T16>(v0.y); q[2] = op_cvt<T16>(v0.z); q[3] = op_cvt<T16>(v0.w);
                    q[4] = op_cvt<T16>(v1.x); q[5] = op_cvt<T16>(v1.y); q[6] = op_cvt<T16>(v1.z); q[7] = op_cvt<T16>(v1.w);
                    *reinterpret_cast<t16x8*>(reinterpret_cast<T16*>(d.out_bf16) + (size_t)row * d.ldc_bf16 + col) = q;
                }
                if (gnb) {
                    fit_gnb_accum(d, gc0, __builtin_bit_cast(float4, pf[ic][ps][0]), v0, gsa0, gsa1);
                    fit_gnb_accum(d, gc1, __builtin_bit_cast(float4, pf[ic][ps][1]), v1, gsb0, gsb1);
                } else if (do_stats) {
                    gsa0 += (v0.x + v0.y) + (v0.z + v0.w);
                    gsa1 += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w);
                    gsb0 += (v1.x + v1.y) + (v1.z + v1.w);
                    gsb1 += (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
                }
            }
            /*hipemu:wave_sync*/                    // every lane has read the slab before the next one is written over it
        }
    }
    FIT_TRACE(5);
    if (do_stats) fit_stats_tail<WGM, WGN, FN, KS>(d, fl + DUMP + 2 * NW * (16 * LDW), gsa0, gsa1, gsb0, gsb1, tn, kg, wm, wn, lane);
}

// ---- epilogues specialised at compile time (IEEE-half operands, the product default) ---------------------------------------
// Round 6: the phase stamps of tools/fit_trace.py show the generic epilogue above ISSUE-bound, not byte-bound -- 14 us for FC1's
// tile, 11 us for the QKV tile whose only work is a bias and a conversion, the same from a warm instruction cache and into
// L2-resident lines, with the last store acknowledged 0.1 us after it was issued: ~1500 executed instructions per wave at ~10
// cycles each -- uniform branches over every fused feature the descriptor could carry, zero-initialised operand registers per
// pass, correctly rounded divisions inside the sigmoids, a serial write-slab / read-slab / store chain.  The hot descriptor
// patterns of the two runners get their own kernels instead: everything the descriptor decides is a template argument, operand
// rows are fetched for the whole wave tile with clamped (always valid) addresses and no branches, the staging slab is double
// buffered so that slab i + 1 is on its way into LDS while slab i is read back, offsets are 32-bit from uniform bases, and the
// sigmoid is v_exp + v_rcp (1 ulp each; gated by the same parity tests).  Arithmetic is otherwise the generic epilogue's, term
// by term, so results are bit-identical wherever no sigmoid is involved.
enum { FIT_EPI_GENERIC = 0,
       FIT_EPI_OUT16,      // alpha, bias_n -> 16-bit out                                  (QKV, the tower's plain dgrads, stat-less convolutions)
       FIT_EPI_RES16,      // ... + 16-bit residual stream                                   (proj, FC2, shortcut adds)
       FIT_EPI_GELU,       // bias_n, QuickGELU -> 16-bit pre-activation + 16-bit out        (FC1)
       FIT_EPI_DGELU,      // * QuickGELU'(16-bit aux) -> 16-bit out                         (W2^T dgrad)
       FIT_EPI_GN,         // OUT16 + the next GroupNorm's sums
       FIT_EPI_RES16_GN,   // RES16 + the next GroupNorm's sums
       FIT_EPI_GNB,        // OUT16 + a GroupNorm-backward's sums on its 16-bit input stream
       FIT_EPI_F32,        // fp32 OPERANDS (the exact mode, v_mfma_f32_16x16x4_f32): alpha, bias_n, fp32 residual -> fp32 out(s)
       FIT_EPI_COUNT };

__device__ __forceinline__ float fit_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <int WGM, int WGN, int FM, int FN, int KS, int EPI>
__device__ __forceinline__ void fit_finish_spec(const GemmArgs& p, float* const fl, f32x4 (&acc)[FM][FN], int tm, int tn, int wave, int kg,
                                                int wt, int wm, int wn, int lane FIT_TRACE_ARG) {
    typedef __attribute__((ext_vector_type(8))) half_t h16x8;
    constexpr int NWT = WGM * WGN, NW = NWT * KS;
    constexpr int BM = 16 * FM * WGM, BN = 16 * FN * WGN, TN = 16 * FN;
    constexpr bool RESID = EPI == FIT_EPI_RES16 || EPI == FIT_EPI_RES16_GN;
    constexpr bool DGELU = EPI == FIT_EPI_DGELU, GELU = EPI == FIT_EPI_GELU, GNB = EPI == FIT_EPI_GNB;
    constexpr bool ROWOP = RESID || DGELU || GNB;                   // one 16-bit row operand: 16 bytes per lane, slab and pass
    constexpr bool STATS = EPI == FIT_EPI_GN || EPI == FIT_EPI_RES16_GN || GNB;
    static_assert(!STATS || BM % 80 != 0, "GroupNorm sums belong to the decoder's tiles");
    const GemmDesc& d = p.d;
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int DUMP = KS > 1 ? NW * FM * FN * 256 : 0;          // floats
    fit_ksum<WGM, WGN, FM, FN, KS>(fl, acc, kg, wt, lane);
    FIT_TRACE(4);

    constexpr int LDW = TN + 4, LPR = TN / 8, RPP = 64 / LPR, NPASS = (16 + RPP - 1) / RPP;
    constexpr int NOWN = (FM + KS - 1) / KS;                        // slabs a wave owns at most
    constexpr int NBUF = NOWN > 1 ? 2 : 1, SLAB = 16 * LDW;
    float* const stage = fl + DUMP + wave * (2 * SLAB);
    const int rbase = tm * BM + wm * (16 * FM), cbase = tn * BN + wn * TN;
    const int lr0 = lane / LPR, lc = (lane - lr0 * LPR) * 8;
    const int col = cbase + lc;
    const bool col_ok = lane < RPP * LPR && col < d.N;             // N % 8 == 0: a lane's 8 columns are in range together
    const int colc = col_ok ? col : 0;                              // loads always address valid memory; stores are masked
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d.bias_n) {
        const float4 b0 = *reinterpret_cast<const float4*>(d.bias_n + colc), b1 = *reinterpret_cast<const float4*>(d.bias_n + colc + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    GnbConst gc0{}, gc1{};
    if constexpr (GNB) { gc0 = gnb_load(d, colc); gc1 = gnb_load(d, colc + 4); }
    const half_t* const rowp = RESID ? reinterpret_cast<const half_t*>(d.resid) : DGELU ? reinterpret_cast<const half_t*>(d.aux)
                                                                                      : reinterpret_cast<const half_t*>(d.gnb_x);
    const unsigned ldrow = RESID ? (unsigned)d.ldr : DGELU ? (unsigned)d.ldaux : (unsigned)d.N;
    half_t* const out = reinterpret_cast<half_t*>(d.out_bf16);
    half_t* const outp = reinterpret_cast<half_t*>(d.out_bf16_pre);
    const unsigned ldo = (unsigned)d.ldc_bf16;
    const int mlast = d.M - 1;

    // ---- the row operand of the whole wave tile, before the first store (a load cannot be hoisted over a store that may alias it)
    uint4 pf[ROWOP ? FM : 1][NPASS];
    if constexpr (ROWOP) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if (KS > 1 && i % KS != kg) continue;                  // wave-uniform
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                int lr = ps * RPP + lr0;
                lr = lr < 15 ? lr : 15;
                int row = rbase + i * 16 + lr;
                row = row < mlast ? row : mlast;
                pf[i][ps] = *reinterpret_cast<const uint4*>(rowp + ((unsigned)row * ldrow + (unsigned)colc));
            }
        }
    }

    float gsa0 = 0.f, gsa1 = 0.f, gsb0 = 0.f, gsb1 = 0.f;           // GroupNorm sums of this lane's two column quads
    auto put = [&](int i, float* st) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(4 * kq + r) * LDW + j * 16 + l15] = acc[i][j][r];
    };
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (KS > 1 && i % KS != kg) continue;                      // wave-uniform
        float* const st = stage + ((i / KS) % NBUF) * SLAB;
        if (i < KS) put(i, st);                                     // the wave's first slab; every later one was written an iteration ahead
        if (i + KS < FM) put(i + KS, stage + (((i + KS) / KS) % NBUF) * SLAB);
        /*hipemu:wave_sync*/                    // the slab is exchanged between the lanes of ONE wave, which runs in lockstep
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + lr0, row = rbase + i * 16 + lr;
            const bool ok = col_ok && lr < 16 && row < d.M;
            const int lrc = lr < 15 ? lr : 15;
            const float4 s0 = *reinterpret_cast<const float4*>(&st[lrc * LDW + lc]);
            const float4 s1 = *reinterpret_cast<const float4*>(&st[lrc * LDW + lc + 4]);
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float x[8];                                             // the row operand widened to fp32
            if constexpr (ROWOP) {
                const h16x8 rx = __builtin_bit_cast(h16x8, pf[i][ps]);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (float)rx[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] *= alpha;
                v[e] += bias[e];
                if constexpr (DGELU) {
                    const float sg = fit_sigmoid(1.702f * x[e]);
                    v[e] *= sg * (1.f + 1.702f * x[e] * (1.f - sg));
                }
                if constexpr (RESID) v[e] += x[e];
            }
            const unsigned oo = (unsigned)row * ldo + (unsigned)col;
            if constexpr (GELU) {
                h16x8 q;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    q[e] = f32_to_f16_sat(v[e]);                    // the saved pre-activation is what the backward differentiates: activate its rounded value
                    const float t = (float)q[e];
                    v[e] = t * fit_sigmoid(1.702f * t);
                }
                if (ok) *reinterpret_cast<h16x8*>(outp + oo) = q;
            }
            {
                h16x8 q;
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e] = f32_to_f16_sat(v[e]);
                if (ok) *reinterpret_cast<h16x8*>(out + oo) = q;
            }
            if constexpr (GNB) {
                if (ok) {
                    fit_gnb_accum<true>(d, gc0, make_float4(x[0], x[1], x[2], x[3]), make_float4(v[0], v[1], v[2], v[3]), gsa0, gsa1);
                    fit_gnb_accum<true>(d, gc1, make_float4(x[4], x[5], x[6], x[7]), make_float4(v[4], v[5], v[6], v[7]), gsb0, gsb1);
                }
            } else if constexpr (STATS) {
                if (ok) {
                    gsa0 += (v[0] + v[1]) + (v[2] + v[3]);
                    gsa1 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    gsb0 += (v[4] + v[5]) + (v[6] + v[7]);
                    gsb1 += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
                }
            }
        }
        /*hipemu:wave_sync*/                    // every lane has read the slab before it is written over (two slabs later)
    }
    FIT_TRACE(5);
    if constexpr (STATS) fit_stats_tail<WGM, WGN, FN, KS>(d, fl + DUMP + 2 * NW * SLAB, gsa0, gsa1, gsb0, gsb1, tn, kg, wm, wn, lane);
}

// ---- the epilogue of the fp32-OPERAND kernels (FIT_EPI_F32: the exact mode's decoder products) ------------------------------
// alpha, bias_n, an fp32 residual, one or two fp32 outputs (`out_f32`, and `out_bf16` -- which addresses fp32 data in this
// mode, gemm.h -- when the caller wants the value in a second buffer).  Same staging as the specialised 16-bit epilogues; a lane's
// 8 columns are two 16-byte accesses; the residual rows of slab i + 1 are fetched while slab i is finished.
template <int WGM, int WGN, int FM, int FN, int KS>
__device__ __forceinline__ void fit_finish_f32(const GemmArgs& p, float* const fl, f32x4 (&acc)[FM][FN], int tm, int tn, int wave, int kg,
                                               int wt, int wm, int wn, int lane FIT_TRACE_ARG) {
    constexpr int NWT = WGM * WGN, NW = NWT * KS;
    constexpr int BM = 16 * FM * WGM, BN = 16 * FN * WGN, TN = 16 * FN;
    const GemmDesc& d = p.d;
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int DUMP = KS > 1 ? NW * FM * FN * 256 : 0;          // floats
    fit_ksum<WGM, WGN, FM, FN, KS>(fl, acc, kg, wt, lane);
    FIT_TRACE(4);
    constexpr int LDW = TN + 4, LPR = TN / 8, RPP = 64 / LPR, NPASS = (16 + RPP - 1) / RPP;
    constexpr int NOWN = (FM + KS - 1) / KS;
    constexpr int NBUF = NOWN > 1 ? 2 : 1, SLAB = 16 * LDW;
    float* const stage = fl + DUMP + wave * (2 * SLAB);
    const int rbase = tm * BM + wm * (16 * FM), cbase = tn * BN + wn * TN;
    const int lr0 = lane / LPR, lc = (lane - lr0 * LPR) * 8;
    const int col = cbase + lc;
    const bool col_ok = lane < RPP * LPR && col < d.N;
    const int colc = col_ok ? col : 0;
    const float alpha = d.alpha_dev ? d.alpha * *d.alpha_dev : d.alpha;
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d.bias_n) {
        const float4 b0 = *reinterpret_cast<const float4*>(d.bias_n + colc), b1 = *reinterpret_cast<const float4*>(d.bias_n + colc + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    // wave-uniform switches: a residual, the next GroupNorm's sums of the output, or a GroupNorm-backward's sums (whose fp32
    // input rows take the residual's place as the row operand; never both).  These kernels run tens of microseconds on the
    // matrix pipes, so the fused sums are run-time branches here, not further instances
    const bool do_stats = d.gn_stats != nullptr;
    const bool gnb = do_stats && d.gnb_x != nullptr;
    const bool has_resid = d.resid != nullptr && !gnb;
    const bool has_row = has_resid || gnb;
    const float* const rowp = gnb ? d.gnb_x : d.resid;
    const int ldrow = gnb ? d.N : d.ldr;
    GnbConst gc0{}, gc1{};
    if (gnb) { gc0 = gnb_load(d, colc); gc1 = gnb_load(d, colc + 4); }
    float gsa0 = 0.f, gsa1 = 0.f, gsb0 = 0.f, gsb1 = 0.f;
    const int mlast = d.M - 1;
    float* const o32 = d.out_f32;
    float* const o2 = reinterpret_cast<float*>(d.out_bf16);
    auto fetch = [&](int i, float4 (&r)[NPASS][2]) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            int lr = ps * RPP + lr0;
            lr = lr < 15 ? lr : 15;
            int row = rbase + i * 16 + lr;
            row = row < mlast ? row : mlast;
            const float* q = rowp + ((size_t)row * ldrow + colc);
            r[ps][0] = *reinterpret_cast<const float4*>(q);
            r[ps][1] = *reinterpret_cast<const float4*>(q + 4);
        }
    };
    auto put = [&](int i, float* st) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(4 * kq + r) * LDW + j * 16 + l15] = acc[i][j][r];
    };
    float4 rnext[NPASS][2];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) rnext[ps][0] = rnext[ps][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (KS > 1 && i % KS != kg) continue;                      // wave-uniform
        float* const st = stage + ((i / KS) % NBUF) * SLAB;
        if (i < KS) { put(i, st); if (has_row) fetch(i, rnext); }
        float4 rcur[NPASS][2];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) { rcur[ps][0] = rnext[ps][0]; rcur[ps][1] = rnext[ps][1]; }
        if (i + KS < FM) {
            put(i + KS, stage + (((i + KS) / KS) % NBUF) * SLAB);
            if (has_row) fetch(i + KS, rnext);
        }
        /*hipemu:wave_sync*/
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + lr0, row = rbase + i * 16 + lr;
            const bool ok = col_ok && lr < 16 && row < d.M;
            const int lrc = lr < 15 ? lr : 15;
            const float4 s0 = *reinterpret_cast<const float4*>(&st[lrc * LDW + lc]);
            const float4 s1 = *reinterpret_cast<const float4*>(&st[lrc * LDW + lc + 4]);
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float x[8] = {rcur[ps][0].x, rcur[ps][0].y, rcur[ps][0].z, rcur[ps][0].w, rcur[ps][1].x, rcur[ps][1].y, rcur[ps][1].z, rcur[ps][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] *= alpha;
                v[e] += bias[e];
                if (!gnb) v[e] += x[e];                              // zeros without a residual
            }
            if (ok) {
                if (o32) {
                    float* q = o32 + ((size_t)row * d.ldc_f32 + col);
                    *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (o2) {
                    float* q = o2 + ((size_t)row * d.ldc_bf16 + col);
                    *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (gnb) {
                    fit_gnb_accum(d, gc0, make_float4(x[0], x[1], x[2], x[3]), make_float4(v[0], v[1], v[2], v[3]), gsa0, gsa1);
                    fit_gnb_accum(d, gc1, make_float4(x[4], x[5], x[6], x[7]), make_float4(v[4], v[5], v[6], v[7]), gsb0, gsb1);
                } else if (do_stats) {
                    gsa0 += (v[0] + v[1]) + (v[2] + v[3]);
                    gsa1 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    gsb0 += (v[4] + v[5]) + (v[6] + v[7]);
                    gsb1 += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
                }
            }
        }
        /*hipemu:wave_sync*/
    }
    FIT_TRACE(5);
    if (do_stats) fit_stats_tail<WGM, WGN, FN, KS>(d, fl + DUMP + 2 * NW * SLAB, gsa0, gsa1, gsb0, gsb1, tn, kg, wm, wn, lane);
}

// WGM x WGN waves per K group, KS K groups; wave tile (16 FM) x (16 FN); block tile BM x BN = (16 FM WGM) x (16 FN WGN).
// EPI: FIT_EPI_GENERIC, or the epilogue specialisation this instance was compiled for (half operands only).
template <int WGM, int WGN, int FM, int FN, int KS, bool CONV, typename T16, int EPI = FIT_EPI_GENERIC>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void gemmfit_kernel(const GemmArgs p, const bf16_t* __restrict__ zero_page) {
    constexpr int NWT = WGM * WGN, NW = NWT * KS;
    constexpr int BM = 16 * FM * WGM, BN = 16 * FN * WGN;
    typedef T16 E;                                       // operand element: bf16_t / half_t, or float (the exact mode: FIT_EPI_F32)
    constexpr bool F32 = std::is_same<T16, float>::value;
    constexpr int BKE = 128 / (int)sizeof(E);            // elements of one 128-byte K-tile row: 64 (16-bit) or 32 (fp32)
    constexpr int CH = 16 / (int)sizeof(E);              // elements of one 16-byte chunk
    static_assert(F32 == (EPI == FIT_EPI_F32), "the fp32-operand kernels have their own epilogue");
    constexpr int SUB = (BM + BN) * BKE;                 // elements of one K tile (A rows, then B rows)
    constexpr int STAGE = KS * SUB;                      // elements of one ring stage
    constexpr int NA = BM / 8, NB = BN / 8, NPS = NA + NB, NP = KS * NPS;   // DMA pieces (8 rows x 128 B) per stage
    constexpr int PW = (NP + NW - 1) / NW;               // pieces per wave per stage (the last ones may be duplicates)
    constexpr int TN = 16 * FN;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows must be whole DMA pieces");
    static_assert(FIT_STAGES * STAGE * (int)sizeof(E) <= 160 * 1024, "ring exceeds the LDS");
    static_assert(NW >= 2 && NW % 2 == 0, "the stagger splits the workgroup in two halves");

    __shared__ __attribute__((aligned(16))) bf16_t lds_raw[FIT_STAGES * STAGE * (int)(sizeof(E) / 2)];     // the only __shared__ object
    E* const lds = reinterpret_cast<E*>(lds_raw);     // the only __shared__ object

    const GemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NWT;                           // K group of this wave
    const int wt = wave - kg * NWT;
    const int wm = wt / WGN, wn = wt - wm * WGN;

#ifdef PRX_FIT_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tr_real0 = __builtin_amdgcn_s_memrealtime();      // the constant 100 MHz clock, to calibrate the stamps' tick
#endif
    FIT_TRACE(0);
    int bid = blockIdx.x;
    if (p.xcd_swizzle) bid = (int)xcd_linear(bid, gridDim.x);
    // tile order: consecutive tiles (an XCD owns a contiguous range of them) share their A row panel (row-major order) or, for
    // weight-heavy problems (N > M: the decoder's 16^2 / 32^2 convolutions), their B column panel (bit 4: column-major order),
    // so that the larger operand is fetched into ONE L2 instead of all eight
    int tm, tn;
    if (p.fit_flags & 16) { tn = bid / p.tiles_m; tm = bid - tn * p.tiles_m; }
    else                  { tm = bid / p.tiles_n; tn = bid - tm * p.tiles_n; }

    // K is cut into KS contiguous ranges, one per K group (group g: K tiles [g nkg, (g + 1) nkg)): consecutive stages of a group
    // walk consecutive K tiles, so an implicit convolution changes its tap only every Cin / 64 stages
    const int nkg = p.kt_total / KS;
    // ---- DMA coordinates: slot j of wave w moves piece min(w + NW j, NP - 1) of every stage ----------------------------
    const E* const Ap = reinterpret_cast<const E*>(d.A);
    const E* const Bp = reinterpret_cast<const E*>(d.B);
    const int lrow = lane >> 3, cpos = lane & 7;
    unsigned voff[PW];                                    // per-lane ELEMENT offset from the operand base at K tile 0 (conv A: the chunk only)
    int pieceA[PW], pieceOff[PW];                         // wave-uniform: operand select, LDS element offset inside a stage
    // implicit convolution, per lane and slot: source row term of the centre tap (c_r1) and the two row deltas packed as 16-bit halves
    // (c_rd: tap row 0 low, tap row 2 high), source column of the centre tap (c_c1), and one word of flags (c_ok): bit ky / bit 3 + kx =
    // tap row / column inside the image, bit 6 = the left tap's source column is one less, bit 7 = the right tap's is one more
    // (with the fused nearest-2x upsample neighbouring taps can share a source column)
    int c_r1[CONV ? PW : 1], c_rd[CONV ? PW : 1], c_c1[CONV ? PW : 1], c_ok[CONV ? PW : 1];
    int s_tap[CONV ? PW : 1], s_c0[CONV ? PW : 1];        // wave-uniform: (tap, first channel) of the slot's K tile in the NEXT stage to issue
    int s_cur[CONV ? PW : 1];                             // wave-uniform: the tap c_off was computed for (-1: none yet)
    int c_off[CONV ? PW : 1];                             // per lane: element offset of the tap's source pixel + the lane's chunk, < 0: padding
    // pixel index -> (image, y, x): shifts when the map is a power of two on both counts (every level of a square power-of-two
    // canvas: wave-uniform test), two integer divisions per lane and A piece otherwise
    const int hw_ = CONV ? d.H * d.W : 1, w_ = CONV ? d.W : 1;
    const bool pow2_map = CONV && ((hw_ & (hw_ - 1)) | (w_ & (w_ - 1))) == 0;
    const int sh_hw = __builtin_ctz((unsigned)hw_ | 0x80000000u), sh_w = __builtin_ctz((unsigned)w_ | 0x80000000u);
    // the weight rows of stage 0 first: an implicit convolution's A slots cost a pixel decode and a dozen selects per lane before
    // their first address exists, and the B pieces (2/3 of a small tile's bytes) are on their way before that arithmetic starts.
    // (A slot's pieces complete in issue order per stage: the counted waits of the main loop still see stage 0 before stage 1.)
    const int nk = (p.fit_flags & 8) ? 0 : nkg;                  // bit 3 (timing experiments only): no main loop
    if constexpr (CONV) {
        if (0 < nk) {
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                int pc = wave + NW * j;
                pc = pc < NP ? pc : NP - 1;
                const int sub = pc / NPS, q = pc - sub * NPS;
                if (q < NA) continue;                            // wave-uniform
                const int r = (q - NA) * 8 + lrow;
                const int chunk = cpos ^ ((r >> 1) & 7);
                int g = tn * BN + r;
                g = g < d.N ? g : d.N - 1;
                const unsigned vo = (unsigned)g * (unsigned)d.ldb + (unsigned)(sub * nkg * BKE + chunk * CH);
                __builtin_amdgcn_global_load_lds((fit_gptr)(Bp + vo), (fit_lptr)(lds + pc * (8 * BKE)), 16, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        int pc = wave + NW * j;
        pc = pc < NP ? pc : NP - 1;
        const int sub = pc / NPS, q = pc - sub * NPS;
        const bool isA = q < NA;
        const int r = (isA ? q : q - NA) * 8 + lrow;      // row inside the A (B) tile
        const int chunk = cpos ^ ((r >> 1) & 7);
        pieceA[j] = isA;
        pieceOff[j] = pc * (8 * BKE);
        if (isA) {
            const int g = tm * BM + r;
            if constexpr (CONV) {
                voff[j] = (unsigned)(chunk * CH);
                const int Hs = d.up == 1 ? (d.H >> 1) : d.H, Ws = d.up == 1 ? (d.W >> 1) : d.W;
                const int hw = hw_;
                int b, rem, y, x;
                if (pow2_map) { b = g >> sh_hw; rem = g & (hw - 1); y = rem >> sh_w; x = rem & (w_ - 1); }
                else { b = g / hw; rem = g - b * hw; y = rem / d.W; x = rem - y * d.W; }
                int okm = 0, ro[3], co[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int yy = y + t - 1, xx = x + t - 1;
                    if (g < d.M && yy >= 0 && yy < d.H) okm |= 1 << t;
                    if (xx >= 0 && xx < d.W) okm |= 8 << t;
                    ro[t] = (b * Hs + (d.up ? (yy >> 1) : yy)) * Ws;
                    co[t] = d.up ? (xx >> 1) : xx;
                }
                if (co[0] != co[1]) okm |= 64;
                if (co[2] != co[1]) okm |= 128;
                c_ok[j] = okm;
                c_r1[j] = ro[1]; c_rd[j] = ((ro[0] - ro[1]) & 0xffff) | ((ro[2] - ro[1]) << 16);
                c_c1[j] = co[1];
                s_tap[j] = (sub * nkg * BKE) / d.Cin;
                s_c0[j] = sub * nkg * BKE - s_tap[j] * d.Cin;
                s_cur[j] = -1; c_off[j] = -1;
            } else {
                const int gc = g < d.M ? g : d.M - 1;
                voff[j] = (unsigned)gc * (unsigned)d.lda + (unsigned)(sub * nkg * BKE + chunk * CH);
            }
        } else {
            int g = tn * BN + r;
            g = g < d.N ? g : d.N - 1;
            voff[j] = (unsigned)g * (unsigned)d.ldb + (unsigned)(sub * nkg * BKE + chunk * CH);
            if constexpr (CONV) { c_ok[j] = 0; c_r1[j] = c_rd[j] = c_c1[j] = 0; s_tap[j] = s_c0[j] = 0; s_cur[j] = -1; c_off[j] = -1; }
        }
    }
    // stages are issued in K order, exactly once each; a_only: the A slots alone (the prologue's stage 0 of a convolution, whose B
    // pieces left before the decode)
    auto issue = [&](int it, int stage, bool a_only) {
        const size_t kel = (size_t)it * BKE;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (a_only && !pieceA[j]) continue;                           // wave-uniform
            const E* src;
            if (CONV && pieceA[j]) {
                const int tap = s_tap[j];
                if (tap != s_cur[j]) {                       // wave-uniform: a new tap every Cin / 64 stages
                    const int ky = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0), kx = tap - 3 * ky;
                    const int my0 = -(int)(ky == 0), my2 = -(int)(ky == 2), mx0 = -(int)(kx == 0), mx2 = -(int)(kx == 2);
                    const int ro = c_r1[j] + (my0 & ((c_rd[j] << 16) >> 16)) + (my2 & (c_rd[j] >> 16));
                    const int co = c_c1[j] - (mx0 & ((c_ok[j] >> 6) & 1)) + (mx2 & ((c_ok[j] >> 7) & 1));
                    const bool ok = ((c_ok[j] >> ky) & (c_ok[j] >> (3 + kx)) & 1) != 0;
                    c_off[j] = ok ? (ro + co) * d.lda + (int)voff[j] : -1;      // M * lda < 2^31 (eligibility)
                    s_cur[j] = tap;
                }
                src = c_off[j] >= 0 ? Ap + (size_t)(unsigned)(c_off[j] + s_c0[j]) : reinterpret_cast<const E*>(zero_page);
                s_c0[j] += BKE;
                if (s_c0[j] >= d.Cin) { s_c0[j] -= d.Cin; ++s_tap[j]; }
            } else {
                src = (pieceA[j] ? Ap : Bp) + kel + voff[j];
            }
            __builtin_amdgcn_global_load_lds((fit_gptr)src, (fit_lptr)(lds + stage * STAGE + pieceOff[j]), 16, 0, 0);
        }
    };

    // ---- fragment coordinates (16x16x32: lane -> row lane & 15, k = 8 (lane >> 4) .. + 7 of the 32-wide step) ------------
    const int l15 = lane & 15, kq = lane >> 4, fkey = (l15 >> 1) & 7;
    const int koff0 = (kq ^ fkey) * CH, koff1 = ((4 + kq) ^ fkey) * CH;
    const int a_el = kg * SUB + (wm * (16 * FM) + l15) * BKE;                 // + fm * 16 * BK
    const int b_el = kg * SUB + BM * BKE + (wn * TN + l15) * BKE;          // + fn * 16 * BK

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int stage) {
        if constexpr (F32) {
            // fp32 operands on v_mfma_f32_16x16x4_f32 (exact: an fmaf chain; 32 cycles per SIMD, the fp32 vector rate): a lane's
            // 16-byte chunk holds four consecutive k of its row; MFMA s takes component s of every lane, i.e. the k set
            // {s, 4 + s, 8 + s, 12 + s} of the 16 k two chunk groups cover -- A and B agree on it, so the sum over k is complete
            // (fragments are read as 16-byte vectors of the ring's 16-bit carrier type and re-typed in registers: read through
            // a float-typed pointer, the compiler orders the reads behind the LDS-DMA just issued with s_waitcnt vmcnt(0) -- the
            // whole stage latency exposed, every stage; tools/disasm.sh shows the difference)
            const bf16_t* const As = reinterpret_cast<const bf16_t*>(lds) + 2 * (stage * STAGE + a_el);
            const bf16_t* const Bs = reinterpret_cast<const bf16_t*>(lds) + 2 * (stage * STAGE + b_el);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ko = 2 * (h ? koff1 : koff0);
                float4 af[FM], bf[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[j] = fit_frag_f32(Bs + j * (32 * BKE) + ko);
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = fit_frag_f32(As + i * (32 * BKE) + ko);
                // component-major: the four MFMAs of one accumulator are FM * FN issues apart (a dependent 16x16x4 waits 40 cycles,
                // an independent one issues every 32)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j) {
                            const float av = c == 0 ? af[i].x : c == 1 ? af[i].y : c == 2 ? af[i].z : af[i].w;
                            const float bv = c == 0 ? bf[j].x : c == 1 ? bf[j].y : c == 2 ? bf[j].z : bf[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);       // DS reads: this half's fragments
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * FM * FN, 0);   // its MFMAs
            }
        } else {
        const E* const As = lds + stage * STAGE + a_el;
        const E* const Bs = lds + stage * STAGE + b_el;
        bf16x8 af0[FM], bf0[FN], af1[FM], bf1[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) bf0[j] = *reinterpret_cast<const bf16x8*>(Bs + j * (16 * BKE) + koff0);
#pragma unroll
        for (int i = 0; i < FM; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(As + i * (16 * BKE) + koff0);
#pragma unroll
        for (int j = 0; j < FN; ++j) bf1[j] = *reinterpret_cast<const bf16x8*>(Bs + j * (16 * BKE) + koff1);
#pragma unroll
        for (int i = 0; i < FM; ++i) af1[i] = *reinterpret_cast<const bf16x8*>(As + i * (16 * BKE) + koff1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = fit_mfma<T16>(af0[i], bf0[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = fit_mfma<T16>(af1[i], bf1[j], acc[i][j]);
        // The stage's schedule is pinned to "first k-step's fragment reads | its MFMAs with the second k-step's reads issued under
        // them | second k-step's MFMAs": left to itself the compiler re-uses fragment registers and waits lgkmcnt(0) seven times
        // per stage (18 ds_read_b128 / 40 MFMAs for FC1's tile); pinned, one LDS latency is exposed per stage.  (The compiler
        // waits lgkmcnt(0), not a count, in front of the first MFMA, so the second k-step's reads are placed BEHIND the first
        // MFMA, where that wait no longer covers them.)  Round-5 A/B on the device: engine 4.90 -> 4.76 ms per iteration,
        // profiles/r05_first_call/.
        __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);       // DS reads: the first k-step's fragments
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // the first MFMA (and the wait in front of it)
        __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);       // DS reads: the second k-step's fragments
        __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - 1, 0);   // the rest of the first k-step's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, FM * FN, 0);       // the second k-step's
    }
    };

    // ---- main loop: ring of 3 stages, counted waits (PW DMA instructions per wave per stage) ---------------------------
    // Staggered wave groups (fit_flags bit 0): waves [0, NW/2) issue their share of stage T + 2 BEFORE computing stage T, waves
    // [NW/2, NW) -- their SIMD partners -- AFTER it.  Issuing a stage costs a wave about as long as computing one (the DMA
    // instructions queue behind the CU's 64 B/clk load path), so with every wave in the same phase the matrix pipes idle
    // while all eight issue; staggered, one wave of a SIMD computes while the other issues.  The late group's pieces of
    // stage T + 2 are issued after every wave passed barrier T (stage T - 1 was read before it: WAR), and are waited for by
    // the same counted wait in front of barrier T + 2.
    const bool late = (p.fit_flags & 1) && wave >= NW / 2;
    FIT_TRACE(1);
    if (0 < nk) issue(0, 0, CONV);
    if (1 < nk) issue(1, 1, false);
#ifdef PRX_FIT_TRACE
    // loop experiments of the diagnostic build (PRX_FIT_TRACE_LOOP): bit 8 = no MFMAs / fragment reads (DMA, waits and barriers
    // only), bit 9 = no DMA after the first two stages (fragment reads + MFMAs on whatever the ring holds)
    const bool x_nomma = (p.fit_flags & 256) != 0, x_nodma = (p.fit_flags & 512) != 0;
#define FIT_ISSUE(T, ST) do { if (!x_nodma) issue(T, ST, false); } while (0)
#define FIT_COMPUTE(ST) do { if (!x_nomma) compute(ST); } while (0)
#else
#define FIT_ISSUE(T, ST) issue(T, ST, false)
#define FIT_COMPUTE(ST) compute(ST)
#endif
#define FIT_STEP(T, ST)                                                                                                 \
    do {                                                                                                                \
        if ((T) + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");                                    \
        else              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_s_barrier();       /* stage T landed for every wave; everyone is done reading stage T - 1 */   \
        if ((T) == 0) FIT_TRACE(2);                                                                                     \
        if (!late && (T) + 2 < nk) FIT_ISSUE((T) + 2, ((ST) + 2) % 3);                                                  \
        FIT_COMPUTE(ST);                                                                                                \
        if (late && (T) + 2 < nk) FIT_ISSUE((T) + 2, ((ST) + 2) % 3);                                                   \
    } while (0)
    int t = 0;
    for (; t + 3 <= nk; t += 3) { FIT_STEP(t, 0); FIT_STEP(t + 1, 1); FIT_STEP(t + 2, 2); }
    if (t < nk) { FIT_STEP(t, 0); ++t; }
    if (t < nk) { FIT_STEP(t, 1); ++t; }
#undef FIT_STEP
#undef FIT_ISSUE
#undef FIT_COMPUTE
    FIT_TRACE(3);
    __builtin_amdgcn_s_barrier();           // the ring is dead: LDS is reused below

    if (p.fit_flags & 4) {                  // bit 2 (timing experiments only): no epilogue -- keep the accumulators alive
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    static_assert(fit_scratch_floats<WGM, WGN, FM, FN, KS>() * 4 <= FIT_STAGES * STAGE * (int)sizeof(E), "epilogue scratch exceeds the ring");
#ifdef PRX_FIT_TRACE
    // experiment (PRX_FIT_TRACE_REP=1, KS == 1 tiles): the epilogue a second time from the SAME code addresses -- the first pass
    // runs it from a cold instruction cache, the second from a warm one; the second pass's stamps replace slots 1 (begin) and 2 (end)
    const int nrep = (KS == 1 && (p.fit_flags & 128)) ? 2 : 1;
    for (int rep = 0; rep < nrep; ++rep) {
        if (rep) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[1] = __builtin_amdgcn_s_memtime(); }
        if constexpr (EPI == FIT_EPI_GENERIC) fit_finish<WGM, WGN, FM, FN, KS, T16>(p, reinterpret_cast<float*>(lds), acc, tm, tn, wave, kg, wt, wm, wn, lane FIT_TRACE_PASS);
        else if constexpr (EPI == FIT_EPI_F32) fit_finish_f32<WGM, WGN, FM, FN, KS>(p, reinterpret_cast<float*>(lds), acc, tm, tn, wave, kg, wt, wm, wn, lane FIT_TRACE_PASS);
        else fit_finish_spec<WGM, WGN, FM, FN, KS, EPI>(p, reinterpret_cast<float*>(lds), acc, tm, tn, wave, kg, wt, wm, wn, lane FIT_TRACE_PASS);
        if (rep) { tr[2] = tr[5]; }
        else if (nrep > 1) { tr[3] = tr[4]; tr[7] = tr[5]; }
    }
    if (nrep > 1) { const unsigned long long b2 = tr[1], e2 = tr[2]; tr[4] = tr[3]; tr[5] = tr[7]; tr[1] = b2; tr[2] = e2; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FIT_TRACE(6);
    if (p.ws && lane < 8) {
        unsigned long long v = tr[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) v = lane == i ? tr[i] : v;
        if (lane == 7) v = ((__builtin_amdgcn_s_memrealtime() - tr_real0) << 32) | (unsigned)bid;   // wave lifetime in 10 ns ticks, tile index
        reinterpret_cast<unsigned long long*>(p.ws)[((size_t)blockIdx.x * NW + wave) * 8 + lane] = v;
    }
#else
    if constexpr (EPI == FIT_EPI_GENERIC) fit_finish<WGM, WGN, FM, FN, KS, T16>(p, reinterpret_cast<float*>(lds), acc, tm, tn, wave, kg, wt, wm, wn, lane);
    else if constexpr (EPI == FIT_EPI_F32) fit_finish_f32<WGM, WGN, FM, FN, KS>(p, reinterpret_cast<float*>(lds), acc, tm, tn, wave, kg, wt, wm, wn, lane);
    else fit_finish_spec<WGM, WGN, FM, FN, KS, EPI>(p, reinterpret_cast<float*>(lds), acc, tm, tn, wave, kg, wt, wm, wn, lane);
#endif
}

template <int WGM, int WGN, int FM, int FN, int KS, bool HAS_CONV = true>
void launch_fit(const GemmArgs& a, dim3 grid, hipStream_t s, const bf16_t* zp) {
    constexpr int threads = 64 * WGM * WGN * KS;
    const bool conv = a.d.a_mode == PRX_A_CONV3X3;
    if constexpr (HAS_CONV) {
        if (conv) {
            if (a.d.h16) hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, true, half_t>), grid, dim3(threads), 0, s, a, zp);
            else         hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, true, bf16_t>), grid, dim3(threads), 0, s, a, zp);
            return;
        }
    }
    if (a.d.h16) hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, false, half_t>), grid, dim3(threads), 0, s, a, zp);
    else         hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, false, bf16_t>), grid, dim3(threads), 0, s, a, zp);
}
// a kernel with a compile-time epilogue (half operands): row-major or convolution instance by the descriptor
template <int WGM, int WGN, int FM, int FN, int KS, int EPI, bool HAS_CONV>
void launch_fit_spec(const GemmArgs& a, dim3 grid, hipStream_t s, const bf16_t* zp) {
    constexpr int threads = 64 * WGM * WGN * KS;
    if constexpr (HAS_CONV) {
        if (a.d.a_mode == PRX_A_CONV3X3) {
            hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, true, half_t, EPI>), grid, dim3(threads), 0, s, a, zp);
            return;
        }
    }
    hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, false, half_t, EPI>), grid, dim3(threads), 0, s, a, zp);
}

// the fp32-operand instance of a tile (exact mode)
template <int WGM, int WGN, int FM, int FN, int KS>
void launch_fit_f32(const GemmArgs& a, dim3 grid, hipStream_t s, const bf16_t* zp) {
    constexpr int threads = 64 * WGM * WGN * KS;
    if (a.d.a_mode == PRX_A_CONV3X3) hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, true, float, FIT_EPI_F32>), grid, dim3(threads), 0, s, a, zp);
    else hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, false, float, FIT_EPI_F32>), grid, dim3(threads), 0, s, a, zp);
}
}  // namespace

// one tile shape of a specialised unit: dispatch on the epilogue kind over the kinds the unit instantiates
#define FIT_SPEC_CASE(EPI_, ...) case EPI_: launch_fit_spec<__VA_ARGS__>(a, grid, s, zp); return true;
