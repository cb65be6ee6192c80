// "Fit-tile" member of the GEMM engine (gemm.h): row-major 16-bit operands, tiles whose count matches the chip.
//
// Why: the ViT-B/32 tower at the headline's 64 cutouts is M = 64 * 50 = 3200 token rows.  With the power-of-two tiles of
// gemm.hip a product has 150 / 300 / 600 tiles for 256 CUs (128x128 / 128x64 / 64x64): either 40 % of the chip idles or a
// second, mostly empty round runs, and the small tiles that do fill the chip move twice the L2->LDS bytes per flop
// (64 B/clk/CU on that path: a 64x64 tile is bound by it at half the MFMA rate).  M = 3200 = 20 * 160 = 40 * 80, so:
//     N = 3072 (FC1, W2^T dgrad, patch dgrad)   160 x 256 tiles   20 x 12 = 240 workgroups
//     N = 2304 (QKV)                            160 x 192 tiles   20 x 12 = 240
//     N =  768 (proj, FC2, the other dgrads)     80 x 128 tiles   40 x  6 = 240, K split over two wave groups
// i.e. one workgroup per CU on 94 % of the chip, 49 - 98 MFMA flops per L2->LDS byte.  The same tiles fit the sharded
// batches (32 / 16 / 8 cutouts: M = 1600 / 800 / 400 are multiples of 80).
//
// Structure: 8 waves (two per SIMD, so one wave's DMA issue / LDS reads overlap its partner's MFMAs), 16x16x32 MFMAs
// (80 = 5 x 16 rows per wave), wave tile 80 x (16 FN).  Operands HBM -> LDS by global_load_lds_dwordx4 into a 3-deep ring
// of K tiles (BK = 64; KS = 2: a stage holds two consecutive K tiles, one per wave group) with counted s_waitcnt vmcnt and
// one raw s_barrier per stage: two stages of DMA stay in flight across the barrier.  Swizzle as in gemm.hip (LDS chunk c of row r
// holds source chunk c ^ ((r >> 1) & 7), applied on the DMA source address and on the fragment read; conflict-free for the
// 16x16x32 operand layout too: a ds_read_b128 lane group covers rows {0-3, 12-15} of one chunk and rows {4-11} of the next).
// Rows >= M / columns >= N are clamped on the source side.  The epilogue is the engine's (gemm_epi.h), staged per wave
// through LDS so that all global accesses are 16 bytes (8 consecutive columns per lane), its HBM reads prefetched.
//
// Requirements (checked by prx_gemmfit_eligible): row-major 16-bit A, K % (64 KS) == 0, N % 8 == 0 with 16-byte-friendly
// epilogue operands, no split-K, no fused GroupNorm sums.
#include "gemm_epi.h"
#include <type_traits>

namespace {
using namespace prx_gemm_dev;

constexpr int FIT_BK = 64;
constexpr int FIT_STAGES = 3;
typedef const __attribute__((address_space(1))) void* fit_gptr;
typedef __attribute__((address_space(3))) void* fit_lptr;

template <typename T16>
__device__ __forceinline__ f32x4 fit_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (std::is_same<T16, half_t>::value)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// WGM x WGN waves per K group, KS K groups; wave tile (16 FM) x (16 FN); block tile BM x BN = (16 FM WGM) x (16 FN WGN).
template <int WGM, int WGN, int FM, int FN, int KS, typename T16>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void gemmfit_kernel(const GemmArgs p) {
    constexpr int NWT = WGM * WGN, NW = NWT * KS;
    constexpr int BM = 16 * FM * WGM, BN = 16 * FN * WGN;
    constexpr int SUB = (BM + BN) * FIT_BK;              // elements of one K tile (A rows, then B rows)
    constexpr int STAGE = KS * SUB;                      // elements of one ring stage
    constexpr int NA = BM / 8, NB = BN / 8, NPS = NA + NB, NP = KS * NPS;   // DMA pieces (8 rows x 128 B) per stage
    constexpr int PW = (NP + NW - 1) / NW;               // pieces per wave per stage (the last ones may be duplicates)
    constexpr int TN = 16 * FN;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows must be whole DMA pieces");
    static_assert(FIT_STAGES * STAGE * 2 <= 160 * 1024, "ring exceeds the LDS");

    __shared__ __attribute__((aligned(16))) bf16_t lds[FIT_STAGES * STAGE];     // the only __shared__ object

    const GemmDesc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NWT;                           // K group of this wave
    const int wt = wave - kg * NWT;
    const int wm = wt / WGN, wn = wt - wm * WGN;

    int bid = blockIdx.x;
    if (p.xcd_swizzle) bid = (int)xcd_linear(bid, gridDim.x);
    const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;

    // ---- DMA coordinates: slot j of this wave moves piece min(wave + NW j, NP - 1) of every stage -----------------------
    const char* const Abase = reinterpret_cast<const char*>(d.A);
    const char* const Bbase = reinterpret_cast<const char*>(d.B);
    const int lrow = lane >> 3, cpos = lane & 7;
    unsigned voff[PW];                                    // per-lane byte offset from the operand base at K tile 0
    int pieceA[PW], pieceOff[PW];                         // wave-uniform: operand select, LDS element offset inside a stage
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        int pc = wave + NW * j;
        pc = pc < NP ? pc : NP - 1;
        const int sub = pc / NPS, q = pc - sub * NPS;
        const bool isA = q < NA;
        const int r = (isA ? q : q - NA) * 8 + lrow;      // row inside the A (B) tile
        const int chunk = cpos ^ ((r >> 1) & 7);
        pieceA[j] = isA;
        pieceOff[j] = pc * (8 * FIT_BK);
        if (isA) {
            int g = tm * BM + r;
            g = g < d.M ? g : d.M - 1;
            voff[j] = ((unsigned)g * (unsigned)d.lda + (unsigned)(sub * FIT_BK + chunk * 8)) * 2u;
        } else {
            int g = tn * BN + r;
            g = g < d.N ? g : d.N - 1;
            voff[j] = ((unsigned)g * (unsigned)d.ldb + (unsigned)(sub * FIT_BK + chunk * 8)) * 2u;
        }
    }
    auto issue = [&](int it, int stage) {
        const size_t kbytes = (size_t)it * (FIT_BK * KS * 2);
        const char* const a_ = Abase + kbytes;
        const char* const b_ = Bbase + kbytes;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const char* const s_ = pieceA[j] ? a_ : b_;
            __builtin_amdgcn_global_load_lds((fit_gptr)(s_ + voff[j]), (fit_lptr)(lds + stage * STAGE + pieceOff[j]), 16, 0, 0);
        }
    };

    // ---- fragment coordinates (16x16x32: lane -> row lane & 15, k = 8 (lane >> 4) .. + 7 of the 32-wide step) ------------
    const int l15 = lane & 15, kq = lane >> 4, fkey = (l15 >> 1) & 7;
    const int koff0 = ((kq ^ fkey) << 3), koff1 = (((4 + kq) ^ fkey) << 3);
    const int a_el = kg * SUB + (wm * (16 * FM) + l15) * FIT_BK;                 // + fm * 16 * BK
    const int b_el = kg * SUB + BM * FIT_BK + (wn * TN + l15) * FIT_BK;          // + fn * 16 * BK

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // both 32-wide k steps' fragments are fetched up front into two register sets (the waits the compiler then places are
    // counted lgkmcnt: the first MFMAs start when their operands arrive, the second set lands under them)
    auto compute = [&](int stage) {
        const bf16_t* const As = lds + stage * STAGE + a_el;
        const bf16_t* const Bs = lds + stage * STAGE + b_el;
        bf16x8 af0[FM], bf0[FN], af1[FM], bf1[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) bf0[j] = *reinterpret_cast<const bf16x8*>(Bs + j * (16 * FIT_BK) + koff0);
#pragma unroll
        for (int i = 0; i < FM; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(As + i * (16 * FIT_BK) + koff0);
#pragma unroll
        for (int j = 0; j < FN; ++j) bf1[j] = *reinterpret_cast<const bf16x8*>(Bs + j * (16 * FIT_BK) + koff1);
#pragma unroll
        for (int i = 0; i < FM; ++i) af1[i] = *reinterpret_cast<const bf16x8*>(As + i * (16 * FIT_BK) + koff1);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = fit_mfma<T16>(af0[i], bf0[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = fit_mfma<T16>(af1[i], bf1[j], acc[i][j]);
    };

    // ---- main loop: ring of 3 stages, counted waits (PW DMA instructions per wave per stage) ---------------------------
    // Staggered wave groups (fit_flags bit 0): waves [0, NW/2) issue their share of stage T + 2 BEFORE computing stage T, waves
    // [NW/2, NW) -- their SIMD partners -- AFTER it.  Issuing a stage costs a wave about as long as computing one (the DMA
    // instructions queue behind the CU's 64 B/clk load path), so with every wave in the same phase the matrix pipes idle
    // while all eight issue; staggered, one wave of a SIMD computes while the other issues.  The late group's pieces of
    // stage T + 2 are issued after every wave passed barrier T (stage T - 1 was read before it: WAR), and are waited for by
    // the same counted wait in front of barrier T + 2.
    const int nk = (p.fit_flags & 8) ? 0 : p.kt_total / KS;      // bit 3 (timing experiments only): no main loop
    const bool late = (p.fit_flags & 1) && wave >= NW / 2;
    if (0 < nk) issue(0, 0);
    if (1 < nk) issue(1, 1);
#define FIT_STEP(T, ST)                                                                                                 \
    do {                                                                                                                \
        if ((T) + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");                                    \
        else              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_s_barrier();       /* stage T landed for every wave; everyone is done reading stage T - 1 */   \
        if (!late && (T) + 2 < nk) issue((T) + 2, ((ST) + 2) % 3);                                                      \
        compute(ST);                                                                                                    \
        if (late && (T) + 2 < nk) issue((T) + 2, ((ST) + 2) % 3);                                                       \
    } while (0)
    int t = 0;
    for (; t + 3 <= nk; t += 3) { FIT_STEP(t, 0); FIT_STEP(t + 1, 1); FIT_STEP(t + 2, 2); }
    if (t < nk) { FIT_STEP(t, 0); ++t; }
    if (t < nk) { FIT_STEP(t, 1); ++t; }
#undef FIT_STEP
    __builtin_amdgcn_s_barrier();           // the ring is dead: LDS is reused below

    if (p.fit_flags & 4) {                  // bit 2 (timing experiments only): no epilogue -- keep the accumulators alive
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // ---- K groups: partner sums through LDS.  Group 0 keeps row fragments [0, FM0), group 1 the rest -------------------
    float* const fl = reinterpret_cast<float*>(lds);
    constexpr int FM0 = KS == 2 ? (FM + 1) / 2 : FM;
    constexpr int DUMP = KS == 2 ? NWT * FM * FN * 256 : 0;        // floats: every wave dumps the fragments it does not keep
    if constexpr (KS == 2) {
        float* const mine = fl + wt * (FM * FN * 256);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const bool keep = (kg == 0) == (i < FM0);
            if (!keep) {
#pragma unroll
                for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(mine + (i * FN + j) * 256 + lane * 4) = acc[i][j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const bool keep = (kg == 0) == (i < FM0);
            if (keep) {
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(mine + (i * FN + j) * 256 + lane * 4);
            }
        }
    }

    // ---- epilogue: per wave, one 16-row fragment row at a time through a private LDS slab; 8 consecutive columns per lane, so
    // the 16-bit outputs leave as 16-byte stores (the store tail of a one-round kernel is bound by store INSTRUCTIONS:
    // cdna_hip_programming.md T21).  Everything the epilogue READS from HBM (residual or aux rows, bias) is fetched for the whole
    // wave tile BEFORE the first store: the compiler cannot move a load above an earlier store that may alias it, and with one
    // workgroup per CU nothing else hides a chain of FM x NPASS dependent load round trips (measured: 19 of FC1's 32 us).
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
    const bool need_aux = act == PRX_ACT_MUL_DQUICKGELU || act == PRX_ACT_MUL_RELUMASK || act == PRX_ACT_RELUMASK_POST;
    float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
    if (d.bias_n && col_ok) {
        bias0 = *reinterpret_cast<const float4*>(d.bias_n + col);
        bias1 = *reinterpret_cast<const float4*>(d.bias_n + col + 4);
    }
    uint4 pf[FM][NPASS][2];                     // residual (2 x 16 bytes) or aux (16 bytes) of this lane's 8 columns
    float pbm[FM][NPASS];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const bool keep = KS == 1 || ((kg == 0) == (i < FM0));
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + lr0, row = rbase + i * 16 + lr;
            const bool ok = keep && col_ok && lr < 16 && row < d.M;
            pf[i][ps][0] = pf[i][ps][1] = uint4{0u, 0u, 0u, 0u};
            pbm[i][ps] = 0.f;
            if (ok) {
                if (has_resid) {
                    const float* r_ = d.resid + (size_t)row * d.ldr + col;
                    pf[i][ps][0] = *reinterpret_cast<const uint4*>(r_);
                    pf[i][ps][1] = *reinterpret_cast<const uint4*>(r_ + 4);
                } else if (need_aux) {
                    pf[i][ps][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T16*>(d.aux) + (size_t)row * d.ldaux + col);
                }
                if (d.bias_m) pbm[i][ps] = d.bias_m[row];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        if (KS == 2 && ((kg == 0) != (i < FM0))) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(4 * kq + r) * LDW + j * 16 + l15] = acc[i][j][r];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int lr = ps * RPP + lr0, row = rbase + i * 16 + lr;
            if (!(col_ok && lr < 16 && row < d.M)) continue;
            float4 v0 = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc]);
            float4 v1 = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc + 4]);
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, p0, p1;
            if (has_resid) {
                r0 = __builtin_bit_cast(float4, pf[i][ps][0]);
                r1 = __builtin_bit_cast(float4, pf[i][ps][1]);
            } else if (need_aux) {
                const t16x8 ax = __builtin_bit_cast(t16x8, pf[i][ps][0]);
                a0[0] = (float)ax[0]; a0[1] = (float)ax[1]; a0[2] = (float)ax[2]; a0[3] = (float)ax[3];
                a1[0] = (float)ax[4]; a1[1] = (float)ax[5]; a1[2] = (float)ax[6]; a1[3] = (float)ax[7];
            }
            v0 = epilogue_math4<T16>(act, alpha, v0, bias0, pbm[i][ps], a0, has_resid, r0, p0);
            v1 = epilogue_math4<T16>(act, alpha, v1, bias1, pbm[i][ps], a1, has_resid, r1, p1);
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
        }
    }
}

template <int WGM, int WGN, int FM, int FN, int KS>
void launch_fit(const GemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr int threads = 64 * WGM * WGN * KS;
    if (a.d.h16) hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, half_t>), grid, dim3(threads), 0, s, a);
    else         hipLaunchKernelGGL((gemmfit_kernel<WGM, WGN, FM, FN, KS, bf16_t>), grid, dim3(threads), 0, s, a);
}
}  // namespace

// The tile shapes this kernel exists in: (160, 256), (160, 192), (80, 128).  ks = K groups of that tile.
bool prx_gemmfit_tile(int bm, int bn, int* ks) {
    int k = 0;
    if (bm == 160 && bn == 256) k = 1;
    else if (bm == 160 && bn == 192) k = 1;
    else if (bm == 80 && bn == 128) k = 2;
    if (ks) *ks = k;
    return k != 0;
}
bool prx_gemmfit_eligible(const GemmDesc& d, int bm, int bn) {
    int ks = 0;
    if (!prx_gemmfit_tile(bm, bn, &ks)) return false;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };       // null passes
    // the epilogue handles 8 consecutive columns per lane with 16-byte accesses
    const bool epi_ok = d.N % 8 == 0 && al16(d.bias_n) && al16(d.resid) && al16(d.aux) && al16(d.out_f32) && al16(d.out_bf16) &&
                        al16(d.out_bf16_pre) && (!d.resid || d.ldr % 4 == 0) && (!d.aux || d.ldaux % 8 == 0) &&
                        (!d.out_f32 || d.ldc_f32 % 4 == 0) && ((!d.out_bf16 && !d.out_bf16_pre) || d.ldc_bf16 % 8 == 0);
    return !d.f32 && !d.a_is_f32 && d.a_mode == PRX_A_ROWMAJOR && d.K % (FIT_BK * ks) == 0 && d.gn_stats == nullptr && epi_ok &&
           (unsigned long long)d.M * d.lda < (1ull << 31) && (unsigned long long)d.N * d.ldb < (1ull << 31);
}
void prx_gemmfit_launch(const prx_gemm_dev::GemmArgs& a, int bm, int bn, dim3 grid, hipStream_t s) {
    if (bm == 160 && bn == 256) launch_fit<2, 4, 5, 4, 1>(a, grid, s);
    else if (bm == 160 && bn == 192) launch_fit<2, 4, 5, 3, 1>(a, grid, s);
    else launch_fit<1, 4, 5, 2, 2>(a, grid, s);
}
