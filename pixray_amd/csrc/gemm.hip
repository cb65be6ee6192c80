// bf16 MFMA GEMM / implicit-GEMM conv engine for gfx950 (see gemm.h).
//
// Structure (round 1): 256-thread workgroup = 4 waves in a 2x2 grid, block tile
// BMxBN in {128x128, 128x64, 64x64}, BK = 64.  Operands are staged
// global -> VGPR -> LDS (the A loader applies the im2col / upsample / f32->bf16
// transforms, which is why this is not an LDS-DMA path), LDS rows are padded to
// 144 B so the ds_read_b128 fragment reads of a 16-lane group hit 16 distinct
// 16-B slots, and every wave issues v_mfma_f32_32x32x16_bf16 on (BM/64)x(BN/64)
// accumulator tiles.  The next K tile's global loads are issued before the
// MFMAs of the current one.
#include "gemm_epi.h"
#include "norms.h"
#include <vector>
#include <mutex>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 64;
constexpr int LDS_LD = BK + 8;  // elements; 144-byte rows

using namespace prx_gemm_dev;

// 16-bit fragments travel as raw `bf16x8` whatever the format; T16 (bf16_t | half_t) picks the conversion and the opcode
template <typename T16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (std::is_same<T16, half_t>::value)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <typename T16, typename TA>
__device__ __forceinline__ bf16x8 load8(const TA* p) {
    if constexpr (std::is_same<TA, float>::value) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
        typedef __attribute__((ext_vector_type(8))) T16 t16x8;
        t16x8 r;
        r[0] = op_cvt<T16>(a[0]); r[1] = op_cvt<T16>(a[1]); r[2] = op_cvt<T16>(a[2]); r[3] = op_cvt<T16>(a[3]);
        r[4] = op_cvt<T16>(b[0]); r[5] = op_cvt<T16>(b[1]); r[6] = op_cvt<T16>(b[2]); r[7] = op_cvt<T16>(b[3]);
        return __builtin_bit_cast(bf16x8, r);
    } else {
        return *reinterpret_cast<const bf16x8*>(p);
    }
}

template <int BM, int BN, typename TA, int AMODE, typename T16 = bf16_t>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int A_CH = BM * 8 / 256;  // 16-byte chunks per thread per K tile
    constexpr int B_CH = BN * 8 / 256;
    constexpr int MT = BM / 64;         // 32x32 MFMA tiles per wave along M
    constexpr int NT = BN / 64;

    __shared__ __attribute__((aligned(16))) bf16_t As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[BN * LDS_LD];

    const GemmDesc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int bid = blockIdx.x;
    const int tm = bid / p.tiles_n;
    const int tn = bid - tm * p.tiles_n;
    const int split = blockIdx.y;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(p.kt_total, kt0 + p.kt_per_split);

    const TA* __restrict__ Ap = reinterpret_cast<const TA*>(d.A);
    const bf16_t* __restrict__ Bp = reinterpret_cast<const bf16_t*>(d.B);

    // ---- per-thread loader coordinates (fixed across the K loop) ------------
    int a_row[A_CH];          // row within the tile
    long long a_base[A_CH];   // row-major: element offset of the row; conv: unused
    int a_y[A_CH], a_x[A_CH], a_b[A_CH];
    bool a_ok[A_CH];
    const int kc = tid & 7;   // which 8-element chunk of the 64-wide K tile
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int row = (tid >> 3) + 32 * i;
        a_row[i] = row;
        int gm = tm * BM + row;
        a_ok[i] = gm < d.M;
        if (AMODE == PRX_A_ROWMAJOR) {
            a_base[i] = (long long)gm * d.lda;
            a_y[i] = a_x[i] = a_b[i] = 0;
        } else {
            int hw = d.H * d.W;
            int b = gm / hw;
            int rem = gm - b * hw;
            int y = rem / d.W;
            a_b[i] = b; a_y[i] = y; a_x[i] = rem - y * d.W;
            a_base[i] = 0;
        }
    }
    int b_row[B_CH];
    long long b_base[B_CH];
    bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int row = (tid >> 3) + 32 * i;
        b_row[i] = row;
        int gn = tn * BN + row;
        b_ok[i] = gn < d.N;
        b_base[i] = (long long)gn * d.ldb;
    }

    bf16x8 a_reg[A_CH], b_reg[B_CH];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // (always_inline on every kernel lambda: they capture the by-value descriptor by reference, and ONE call the inliner declines
    // puts the whole descriptor on the stack -- 320 bytes of scratch per lane, tests/test_host_logic.py guards it)
    auto load_tiles = [&](int kt) __attribute__((always_inline)) {
        const int k = kt * BK + kc * 8;
        const bool k_ok = k < d.K;
        if (AMODE == PRX_A_ROWMAJOR) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                a_reg[i] = (a_ok[i] && k_ok) ? load8<T16, TA>(Ap + a_base[i] + k) : zero8;
        } else {
            int tap = k / d.Cin;
            int c = k - tap * d.Cin;
            int ky = tap / 3;
            int kx = tap - 3 * ky;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                int yy = a_y[i] + ky - 1, xx = a_x[i] + kx - 1;
                bool ok = a_ok[i] && k_ok && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
                long long pix;
                if (d.up) pix = ((long long)a_b[i] * (d.H >> 1) + (yy >> 1)) * (d.W >> 1) + (xx >> 1);
                else      pix = ((long long)a_b[i] * d.H + yy) * d.W + xx;
                a_reg[i] = ok ? load8<T16, TA>(Ap + pix * d.lda + c) : zero8;
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            b_reg[i] = (b_ok[i] && k_ok) ? *reinterpret_cast<const bf16x8*>(Bp + b_base[i] + k) : zero8;
    };
    auto store_tiles = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            *reinterpret_cast<bf16x8*>(&As[a_row[i] * LDS_LD + kc * 8]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            *reinterpret_cast<bf16x8*>(&Bs[b_row[i] * LDS_LD + kc * 8]) = b_reg[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles();
    }
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = 8 * (lane >> 5);
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (more) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(
                    &As[(wm * (BM / 2) + i * 32 + frag_row) * LDS_LD + ks * 16 + frag_k]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(
                    &Bs[(wn * (BN / 2) + j * 32 + frag_row) * LDS_LD + ks * 16 + frag_k]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma16<T16>(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue -----------------------------------------------------------
    const int row0 = tm * BM + wm * (BM / 2) + 4 * (lane >> 5);
    const int col0 = tn * BN + wn * (BN / 2) + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = col0 + j * 32;
            if (col >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= d.M) continue;
                if (p.splits > 1)
                    p.ws[((size_t)split * d.M + row) * d.N + col] = acc[i][j][r];
                else
                    epilogue_store<T16>(d, row, col, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Exact-f32 variant (GemmDesc::f32): A and B stay fp32 from HBM to the matrix core, v_mfma_f32_32x32x2_f32
// (D = fma chain over k, bit-for-bit f32; 64 cycles per instruction = the f32 vector rate, 157 TFLOP/s peak).
// Same structure as the register-staged kernel above with BK = 32 floats (128-byte rows, 144-byte LDS pitch): one
// ds_read_b128 per fragment fetches 4 consecutive k for the lane's row; the lower half-wave (k slot 0 of the
// instruction) holds k = 8s..8s+3 and the upper half (k slot 1) k = 8s+4..8s+7, so four MFMAs consume the 8 k of
// step s -- the assignment of k values to the instruction's two slots is free as long as A and B agree.
// This is the parity mode, not the fast path: it is MFMA-bound at 1/16 of the bf16 rate by construction.
// ---------------------------------------------------------------------------------------------
constexpr int BKF = 32;
constexpr int LDS_LDF = BKF + 4;   // floats; 144-byte rows

template <int BM, int BN, int AMODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
    constexpr int A_CH = BM * 8 / 256;  // 16-byte chunks (4 floats) per thread per K tile
    constexpr int B_CH = BN * 8 / 256;
    constexpr int MT = BM / 64;
    constexpr int NT = BN / 64;

    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LDF];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LDF];

    const GemmDesc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int bid = blockIdx.x;
    const int tm = bid / p.tiles_n;
    const int tn = bid - tm * p.tiles_n;
    const int split = blockIdx.y;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(p.kt_total, kt0 + p.kt_per_split);

    const float* __restrict__ Ap = reinterpret_cast<const float*>(d.A);
    const float* __restrict__ Bp = reinterpret_cast<const float*>(d.B);

    int a_row[A_CH];
    long long a_base[A_CH];
    int a_y[A_CH], a_x[A_CH], a_b[A_CH];
    bool a_ok[A_CH];
    const int kc = tid & 7;   // which 4-float chunk of the 32-wide K tile
    // source geometry of the implicit conv (see the v2 kernel): up == 1 nearest-2x upsampled, up == 2 stride-2 window
    const int Hs = d.up == 1 ? (d.H >> 1) : (d.up == 2 ? 2 * d.H : d.H);
    const int Ws = d.up == 1 ? (d.W >> 1) : (d.up == 2 ? 2 * d.W : d.W);
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int row = (tid >> 3) + 32 * i;
        a_row[i] = row;
        const int gm = tm * BM + row;
        a_ok[i] = gm < d.M;
        if (AMODE == PRX_A_ROWMAJOR) {
            a_base[i] = (long long)gm * d.lda;
            a_y[i] = a_x[i] = a_b[i] = 0;
        } else {
            const int hw = d.H * d.W;
            const int b = gm / hw;
            const int rem = gm - b * hw;
            const int y = rem / d.W;
            a_b[i] = b; a_y[i] = y; a_x[i] = rem - y * d.W;
            a_base[i] = 0;
        }
    }
    int b_row[B_CH];
    long long b_base[B_CH];
    bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int row = (tid >> 3) + 32 * i;
        b_row[i] = row;
        const int gn = tn * BN + row;
        b_ok[i] = gn < d.N;
        b_base[i] = (long long)gn * d.ldb;
    }

    f32x4 a_reg[A_CH], b_reg[B_CH];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](int kt) __attribute__((always_inline)) {
        const int k = kt * BKF + kc * 4;
        const bool k_ok = k < d.K;
        if (AMODE == PRX_A_ROWMAJOR) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                a_reg[i] = (a_ok[i] && k_ok) ? *reinterpret_cast<const f32x4*>(Ap + a_base[i] + k) : zero4;
        } else {
            const int tap = k / d.Cin;
            const int c = k - tap * d.Cin;
            const int ky = tap / 3;
            const int kx = tap - 3 * ky;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                int yy, xx;
                bool ok = a_ok[i] && k_ok;
                if (d.up == 2) {          // taming Downsample: pad (0,1,0,1), stride 2
                    yy = 2 * a_y[i] + ky; xx = 2 * a_x[i] + kx;
                    ok = ok && yy < Hs && xx < Ws;
                } else {
                    yy = a_y[i] + ky - 1; xx = a_x[i] + kx - 1;
                    ok = ok && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
                    if (d.up == 1) { yy >>= 1; xx >>= 1; }
                }
                const long long pix = ((long long)a_b[i] * Hs + yy) * Ws + xx;
                a_reg[i] = ok ? *reinterpret_cast<const f32x4*>(Ap + pix * d.lda + c) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            b_reg[i] = (b_ok[i] && k_ok) ? *reinterpret_cast<const f32x4*>(Bp + b_base[i] + k) : zero4;
    };
    auto store_tiles = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            *reinterpret_cast<f32x4*>(&As[a_row[i] * LDS_LDF + kc * 4]) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            *reinterpret_cast<f32x4*>(&Bs[b_row[i] * LDS_LDF + kc * 4]) = b_reg[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles();
    }
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = 4 * (lane >> 5);
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (more) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BKF / 8; ++ks) {
            f32x4 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&As[(wm * (BM / 2) + i * 32 + frag_row) * LDS_LDF + ks * 8 + frag_k]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * (BN / 2) + j * 32 + frag_row) * LDS_LDF + ks * 8 + frag_k]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    const int row0 = tm * BM + wm * (BM / 2) + 4 * (lane >> 5);
    const int col0 = tn * BN + wn * (BN / 2) + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = col0 + j * 32;
            if (col >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= d.M) continue;
                if (p.splits > 1)
                    p.ws[((size_t)split * d.M + row) * d.N + col] = acc[i][j][r];
                else
                    epilogue_store<float>(d, row, col, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// v2: operands go HBM -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write), LDS is
// double-buffered with ONE barrier per K tile, and the unpadded [rows][64] bf16 tile is XOR-swizzled on the
// *source* side (the DMA destination is lane-linear): LDS chunk position c of row r holds global chunk
// c ^ ((r >> 1) & 7), which makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte slots.
// Out-of-range rows / K tails / conv zero padding read from a zero page.  bf16 A only.
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// NWM = waves along M (2 -> 4 waves / 256 threads, 4 -> 8 waves / 512 threads for the 256-row tiles); two waves along N.
template <int BM, int BN, int AMODE, int STAGES, bool C64 = false, int NWM = 2, typename T16 = bf16_t>
__global__ __launch_bounds__(NWM * 128) void gemm_glds_kernel(const GemmArgs p, const bf16_t* __restrict__ zero_page) {
    constexpr int NW = NWM * 2;
    constexpr int A_IN = BM / (8 * NW);   // DMA instructions (8 rows x 128 B each) per wave per K tile
    constexpr int B_IN = BN / (8 * NW);
    constexpr int MT = BM / (32 * NWM);
    constexpr int NT = BN / 64;
    static_assert(A_IN >= 1 && B_IN >= 1 && MT >= 1 && NT >= 1, "tile too small for the wave grid");
    constexpr int TILE = (BM + BN) * BK;   // bf16 elements per stage

    __shared__ __attribute__((aligned(16))) bf16_t lds[STAGES * TILE];

    const GemmDesc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order (T1): workgroup b runs on XCD b % 8, each with a private 4 MB L2.  Give every XCD a
    // contiguous range of tiles (bijective for any tile count) so the A row panels / B column panels it touches are a
    // 1/8 slice of the operands instead of all of them.
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, xcd = bid & 7, local = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tm = bid / p.tiles_n;
    const int tn = bid - tm * p.tiles_n;
    const int split = blockIdx.y;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(p.kt_total, kt0 + p.kt_per_split);

    const bf16_t* __restrict__ Ap = reinterpret_cast<const bf16_t*>(d.A);
    const bf16_t* __restrict__ Bp = reinterpret_cast<const bf16_t*>(d.B);

    // ---- per-lane DMA coordinates (fixed across the K loop) -------------------
    const int lrow = lane >> 3, cpos = lane & 7;
    long long a_base[A_IN];
    int a_sw[A_IN], a_y[A_IN], a_x[A_IN], a_b[A_IN];
    bool a_ok[A_IN];
#pragma unroll
    for (int i = 0; i < A_IN; ++i) {
        const int row = (wave * A_IN + i) * 8 + lrow;
        a_sw[i] = cpos ^ ((row >> 1) & 7);
        const int gm = tm * BM + row;
        a_ok[i] = gm < d.M;
        if (AMODE == PRX_A_ROWMAJOR) {
            a_base[i] = (long long)gm * d.lda;
            a_y[i] = a_x[i] = a_b[i] = 0;
        } else {
            const int hw = d.H * d.W;
            const int b = gm / hw;
            const int rem = gm - b * hw;
            const int y = rem / d.W;
            a_b[i] = b; a_y[i] = y; a_x[i] = rem - y * d.W;
            a_base[i] = 0;
        }
    }
    // C64 (implicit conv with Cin % 64 == 0, i.e. every K tile lies inside ONE filter tap): the tap is wave-uniform,
    // so the per-lane gather address is "row term[ky] + column term[kx]" picked from six values computed once here.
    // Without this the tap/bounds/pixel arithmetic (two integer divisions per DMA instruction) costs ~10x the VALU
    // time of the MFMAs it feeds and the conv kernels are VALU-bound.
    // term(ky) = mid + (ky == 0 ? d0 : 0) + (ky == 2 ? d2 : 0), evaluated with scalar masks (no per-lane selects or
    // indexed register arrays, which hipcc would spill to scratch)
    int c_r1[C64 ? A_IN : 1], c_rd0[C64 ? A_IN : 1], c_rd2[C64 ? A_IN : 1];
    int c_c1[C64 ? A_IN : 1], c_cd0[C64 ? A_IN : 1], c_cd2[C64 ? A_IN : 1], c_ok[C64 ? A_IN : 1];
    if constexpr (C64) {
        // source map: up == 1 nearest-2x upsampled (H/2 x W/2), up == 2 stride-2 "down" conv with taming's (0,1,0,1)
        // padding (source 2H x 2W, tap t reads 2y+t / 2x+t, only the bottom/right edge is padded)
        const int Hs = d.up == 1 ? (d.H >> 1) : (d.up == 2 ? 2 * d.H : d.H);
        const int Ws = d.up == 1 ? (d.W >> 1) : (d.up == 2 ? 2 * d.W : d.W);
#pragma unroll
        for (int i = 0; i < A_IN; ++i) {
            int okm = 0, ro[3], co[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (d.up == 2) {
                    const int yy = 2 * a_y[i] + t, xx = 2 * a_x[i] + t;
                    if (a_ok[i] && yy < Hs) okm |= 1 << t;
                    if (xx < Ws) okm |= 8 << t;
                    ro[t] = (a_b[i] * Hs + yy) * Ws;
                    co[t] = xx;
                } else {
                    const int yy = a_y[i] + t - 1, xx = a_x[i] + t - 1;
                    if (a_ok[i] && yy >= 0 && yy < d.H) okm |= 1 << t;
                    if (xx >= 0 && xx < d.W) okm |= 8 << t;
                    ro[t] = (a_b[i] * Hs + (d.up ? (yy >> 1) : yy)) * Ws;
                    co[t] = d.up ? (xx >> 1) : xx;
                }
            }
            c_ok[i] = okm;
            c_r1[i] = ro[1]; c_rd0[i] = ro[0] - ro[1]; c_rd2[i] = ro[2] - ro[1];
            c_c1[i] = co[1]; c_cd0[i] = co[0] - co[1]; c_cd2[i] = co[2] - co[1];
        }
    }
    int c_tap = 0, c_c0 = 0;            // running (tap, first channel) of the next K tile to issue (C64)
    if constexpr (C64) { c_tap = (kt0 * BK) / d.Cin; c_c0 = kt0 * BK - c_tap * d.Cin; }

    long long b_base[B_IN];
    int b_sw[B_IN];
    bool b_ok[B_IN];
#pragma unroll
    for (int i = 0; i < B_IN; ++i) {
        const int row = (wave * B_IN + i) * 8 + lrow;
        b_sw[i] = cpos ^ ((row >> 1) & 7);
        const int gn = tn * BN + row;
        b_ok[i] = gn < d.N;
        b_base[i] = (long long)gn * d.ldb;
    }

    const long long zoff = zero_page - Ap;
    auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
        bf16_t* As = lds + buf * TILE;
        bf16_t* Bs = As + BM * BK;
        if constexpr (C64) {
            // tiles are issued in K order, so (tap, c0) advance incrementally; all of this is scalar
            const int ky = c_tap >= 6 ? 2 : (c_tap >= 3 ? 1 : 0);
            const int kx = c_tap - 3 * ky;
            const int my0 = -(int)(ky == 0), my2 = -(int)(ky == 2), mx0 = -(int)(kx == 0), mx2 = -(int)(kx == 2);
#pragma unroll
            for (int i = 0; i < A_IN; ++i) {
                const int ro = c_r1[i] + (my0 & c_rd0[i]) + (my2 & c_rd2[i]);
                const int co = c_c1[i] + (mx0 & c_cd0[i]) + (mx2 & c_cd2[i]);
                const bool ok = ((c_ok[i] >> ky) & (c_ok[i] >> (3 + kx)) & 1) != 0;
                const bf16_t* src = ok ? Ap + (long long)(ro + co) * d.lda + (c_c0 + a_sw[i] * 8) : zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (wave * A_IN + i) * 8 * BK), 16, 0, 0);
            }
            c_c0 += BK;
            if (c_c0 >= d.Cin) { c_c0 = 0; ++c_tap; }
        } else
#pragma unroll
        for (int i = 0; i < A_IN; ++i) {
            const int k = kt * BK + a_sw[i] * 8;
            const bf16_t* src = zero_page;
            if (AMODE == PRX_A_ROWMAJOR) {
                // a select, not a branch: written as nested ifs hipcc guarded each of these loads with its own saveexec /
                // execz block in the K loop (the B loads below, written as a select, got two v_cndmask each)
                const long long off = (a_ok[i] && k < d.K) ? a_base[i] + k : zoff;     // zoff: the zero page, relative to A
                src = Ap + off;
            } else {
                const int tap = k / d.Cin;
                const int c = k - tap * d.Cin;
                const int ky = tap / 3;
                const int kx = tap - 3 * ky;
                const int yy = a_y[i] + ky - 1, xx = a_x[i] + kx - 1;
                if (a_ok[i] && k < d.K && yy >= 0 && yy < d.H && xx >= 0 && xx < d.W) {
                    long long pix;
                    if (d.up) pix = ((long long)a_b[i] * (d.H >> 1) + (yy >> 1)) * (d.W >> 1) + (xx >> 1);
                    else      pix = ((long long)a_b[i] * d.H + yy) * d.W + xx;
                    src = Ap + pix * d.lda + c;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (wave * A_IN + i) * 8 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IN; ++i) {
            const int k = kt * BK + b_sw[i] * 8;
            const bf16_t* src = (b_ok[i] && k < d.K) ? (Bp + b_base[i] + k) : zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (wave * B_IN + i) * 8 * BK), 16, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int khalf = lane >> 5;
    int a_off[MT], a_key[MT], b_off[NT], b_key[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * (BM / NWM) + i * 32 + frag_row;
        a_off[i] = r * BK; a_key[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int r = wn * (BN / 2) + j * 32 + frag_row;
        b_off[j] = r * BK; b_key[j] = (r >> 1) & 7;
    }
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const bf16_t* As = lds + buf * TILE;
        const bf16_t* Bs = As + BM * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kc = ks * 2 + khalf;
            bf16x8 af[MT], bfr[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(As + a_off[i] + ((kc ^ a_key[i]) << 3));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + b_off[j] + ((kc ^ b_key[j]) << 3));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma16<T16>(af[i], bfr[j], acc[i][j]);
        }
    };

    if constexpr (STAGES == 2) {
        // double buffer, one barrier per K tile (hipcc drains the DMA queue -- vmcnt(0) -- in front of the barrier)
        if (kt0 < kt1) issue(kt0, 0);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            if (kt + 1 < kt1) issue(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    } else {
        // STAGES-deep ring with COUNTED waits: STAGES-1 K tiles of DMA stay in flight across the (raw) barrier, so a
        // workgroup that is alone on its CU no longer pays one L2 round trip per K tile.  Each wave issues
        // LPT = A_IN + B_IN DMA instructions per tile; "vmcnt(LPT*(STAGES-2))" = everything but the newest
        // STAGES-2 tiles has landed.
        constexpr int LPT = A_IN + B_IN;
        const int nk = kt1 - kt0;
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) issue(kt0 + t, t);
        for (int t = 0; t < nk; ++t) {
            if (t + STAGES - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * (STAGES - 2)) : "memory");
            else                     asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // tile t complete for every wave; everyone is done reading tile t-1
            if (t + STAGES - 1 < nk) issue(kt0 + t + STAGES - 1, (t + STAGES - 1) % STAGES);
            compute(t % STAGES);
        }
        __builtin_amdgcn_s_barrier();          // LDS is reused by the epilogue
    }

    if (p.vec_epi) {
        // Epilogue through LDS: the MFMA C layout holds one column per lane (2- or 4-byte scattered stores);
        // staging each wave's 32 x (BN/2) slab row-major lets every lane handle 4 consecutive columns, so bias /
        // residual / aux loads and all stores are 8-16-byte coalesced accesses.
        constexpr int CW = BN / 2;            // columns of the wave tile
        constexpr int LDW = CW + 4;           // padded row (floats), keeps rows 16-byte aligned
        constexpr int LPR = CW / 4;           // lanes per row
        constexpr int RPP = 64 / LPR;         // rows per pass
        float* stage = reinterpret_cast<float*>(lds) + wave * (32 * LDW);
        float* gacc = reinterpret_cast<float*>(lds) + NW * (32 * LDW);      // [BN/4][2] per-column-quad partial sums
        const bool do_stats = d.gn_stats != nullptr && p.splits == 1;
        if (do_stats && tid < BN / 2) gacc[tid] = 0.f;
        float gs0 = 0.f, gs1 = 0.f;
        const int rbase = tm * BM + wm * (BM / NWM);
        const int cbase = tn * BN + wn * (BN / 2);
        const bool gnb = do_stats && d.gnb_x != nullptr;
        GnbConst gc{};
        if (gnb && cbase + (lane % LPR) * 4 < d.N) gc = gnb_load(d, cbase + (lane % LPR) * 4);   // this lane's column quad is fixed
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDW + j * 32 + (lane & 31)] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 32; rr += RPP) {
                const int lr = rr + lane / LPR;
                const int lc = (lane % LPR) * 4;
                const float4 v = *reinterpret_cast<const float4*>(&stage[lr * LDW + lc]);
                const int row = rbase + i * 32 + lr, col = cbase + lc;
                if (row < d.M && col < d.N) {
                    if (p.splits > 1)
                        *reinterpret_cast<float4*>(&p.ws[((size_t)split * d.M + row) * d.N + col]) = v;
                    else {
                        const float4 o = epilogue_store4<T16>(d, row, col, v);
                        if (gnb) gnb_accum<T16>(d, gc, row, col, o, gs0, gs1);
                        else {
                            gs0 += (o.x + o.y) + (o.z + o.w);
                            gs1 += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (do_stats) {
            // lanes with the same column quad (lane % LPR) -> one value, then the block's two M-waves via LDS
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) { gs0 += __shfl_xor(gs0, o, 64); gs1 += __shfl_xor(gs1, o, 64); }
            if (lane < LPR) {
                const int q = wn * LPR + lane;            // column quad within the block tile
                atomicAdd(&gacc[q * 2], gs0);
                atomicAdd(&gacc[q * 2 + 1], gs1);
            }
            __syncthreads();
            const int qpg = d.gn_gs >> 2;                  // quads per group
            const int ngrp = BN / d.gn_gs;                 // groups covered by this block tile
            if (tid < ngrp * 2) {
                const int gl = tid >> 1, mom = tid & 1;
                const int col = tn * BN + gl * d.gn_gs;
                if (col < d.N) {
                    double a2 = 0.0;
                    for (int q = 0; q < qpg; ++q) a2 += (double)gacc[(gl * qpg + q) * 2 + mom];
                    atomicAdd(&d.gn_stats[(size_t)(col / d.gn_gs) * 2 + mom], a2);
                }
            }
        }
        return;
    }
    const int row0 = tm * BM + wm * (BM / NWM) + 4 * (lane >> 5);
    const int col0 = tn * BN + wn * (BN / 2) + (lane & 31);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = col0 + j * 32;
            if (col >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= d.M) continue;
                if (p.splits > 1)
                    p.ws[((size_t)split * d.M + row) * d.N + col] = acc[i][j][r];
                else
                    epilogue_store<T16>(d, row, col, acc[i][j][r]);
            }
        }
}

template <typename TOp>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const GemmDesc& d = p.d;
    const size_t total = (size_t)d.M * d.N;
    if (p.vec_epi) {
        // 32 groups x {sum, sumsq} for this block, in DOUBLE: the order of the LDS atomics varies from run to run, and a
        // float accumulator would make the statistics (hence a few bf16 roundings downstream, which the decoder then
        // amplifies to its bf16 noise floor) irreproducible; double keeps the order effect ~1e-16
        __shared__ double gacc[64];
        const bool do_stats = d.gn_stats != nullptr;
        if (do_stats && threadIdx.x < 64) gacc[threadIdx.x] = 0.0;
        if (do_stats) __syncthreads();
        const int run = do_stats ? (d.gn_gs >> 2) : 1;      // consecutive lanes (4 columns each) sharing a group: 1, 2, 4, ...
        const size_t total4 = total >> 2;
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        const size_t base = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        const size_t iters = (total4 + stride - 1) / stride;        // uniform trip count: the shuffles below need full waves
        for (size_t it = 0; it < iters; ++it) {
            const size_t i4 = base + it * stride;
            const bool live = i4 < total4;
            float s0 = 0.f, s1 = 0.f;
            int col = 0;
            if (live) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                // four partials in flight at a time (a plain loop compiles to load / s_waitcnt vmcnt(0) / add per split: one L2
                // round trip per partial); the additions keep their order, so the sum is bit-identical to the serial loop
                const float4* part = reinterpret_cast<const float4*>(p.ws) + i4;
                const size_t pstride = total >> 2;
                int s = 0;
                for (; s + 4 <= p.splits; s += 4) {
                    const float4 w0 = part[(size_t)s * pstride], w1 = part[(size_t)(s + 1) * pstride];
                    const float4 w2 = part[(size_t)(s + 2) * pstride], w3 = part[(size_t)(s + 3) * pstride];
                    v.x += w0.x; v.y += w0.y; v.z += w0.z; v.w += w0.w;
                    v.x += w1.x; v.y += w1.y; v.z += w1.z; v.w += w1.w;
                    v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w;
                    v.x += w3.x; v.y += w3.y; v.z += w3.z; v.w += w3.w;
                }
                for (; s < p.splits; ++s) {
                    const float4 w = part[(size_t)s * pstride];
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                const size_t idx = i4 << 2;
                const int row = (int)(idx / d.N);
                col = (int)(idx - (size_t)row * d.N);
                const float4 o = epilogue_store4<TOp>(d, row, col, v);
                if (do_stats && d.gnb_x) gnb_accum<TOp>(d, gnb_load(d, col), row, col, o, s0, s1);
                else {
                    s0 = (o.x + o.y) + (o.z + o.w);
                    s1 = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                }
            }
            if (do_stats) {
                // lanes of one group are an aligned run (N and the block offset are multiples of gn_gs): fixed-order butterfly
                for (int o = 1; o < run && o < 64; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
                if (live && (threadIdx.x & (run - 1)) == 0) {
                    const int g = col / d.gn_gs;
                    atomicAdd(&gacc[g * 2], (double)s0);
                    atomicAdd(&gacc[g * 2 + 1], (double)s1);
                }
            }
        }
        if (do_stats) {
            __syncthreads();
            if (threadIdx.x < 64 && gacc[threadIdx.x] != 0.0) atomicAdd(&d.gn_stats[threadIdx.x], gacc[threadIdx.x]);
        }
        return;
    }
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.ws[(size_t)s * total + idx];
        int row = (int)(idx / d.N);
        int col = (int)(idx - (size_t)row * d.N);
        epilogue_store<TOp>(d, row, col, v);
    }
}

template <int BM, int BN, typename T16>
void launch_cfg_t(const GemmArgs& a, dim3 grid, hipStream_t s) {
    const GemmDesc& d = a.d;
    if (d.a_mode == PRX_A_ROWMAJOR) {
        if (d.a_is_f32) hipLaunchKernelGGL((gemm_kernel<BM, BN, float, PRX_A_ROWMAJOR, T16>), grid, dim3(256), 0, s, a);
        else            hipLaunchKernelGGL((gemm_kernel<BM, BN, bf16_t, PRX_A_ROWMAJOR, T16>), grid, dim3(256), 0, s, a);
    } else {
        if (d.a_is_f32) hipLaunchKernelGGL((gemm_kernel<BM, BN, float, PRX_A_CONV3X3, T16>), grid, dim3(256), 0, s, a);
        else            hipLaunchKernelGGL((gemm_kernel<BM, BN, bf16_t, PRX_A_CONV3X3, T16>), grid, dim3(256), 0, s, a);
    }
}
template <int BM, int BN>
void launch_cfg(const GemmArgs& a, dim3 grid, hipStream_t s) {
    if (a.d.h16) launch_cfg_t<BM, BN, half_t>(a, grid, s);
    else         launch_cfg_t<BM, BN, bf16_t>(a, grid, s);
}
template <int BM, int BN>
void launch_f32(const GemmArgs& a, dim3 grid, hipStream_t s) {
    if (a.d.a_mode == PRX_A_ROWMAJOR) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, PRX_A_ROWMAJOR>), grid, dim3(256), 0, s, a);
    else                              hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, PRX_A_CONV3X3>), grid, dim3(256), 0, s, a);
}

template <int BM, int BN, int STAGES, typename T16>
void launch_glds_t(const GemmArgs& a, dim3 grid, hipStream_t s, const bf16_t* zero_page, bool c64) {
    if (a.d.a_mode == PRX_A_ROWMAJOR)
        hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, PRX_A_ROWMAJOR, STAGES, false, 2, T16>), grid, dim3(256), 0, s, a, zero_page);
    else if (c64)
        hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, PRX_A_CONV3X3, STAGES, true, 2, T16>), grid, dim3(256), 0, s, a, zero_page);
    else
        hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, PRX_A_CONV3X3, STAGES, false, 2, T16>), grid, dim3(256), 0, s, a, zero_page);
}
template <int BM, int BN, int STAGES>
void launch_glds_s(const GemmArgs& a, dim3 grid, hipStream_t s, const bf16_t* zero_page, bool c64) {
    if (a.d.h16) launch_glds_t<BM, BN, STAGES, half_t>(a, grid, s, zero_page, c64);
    else         launch_glds_t<BM, BN, STAGES, bf16_t>(a, grid, s, zero_page, c64);
}
template <int BM, int BN>
void launch_glds(const GemmArgs& a, dim3 grid, hipStream_t s, const bf16_t* zero_page, int stages, bool c64) {
    if (stages >= 4 && (BM + BN) * BK * 2 * 4 <= 160 * 1024) launch_glds_s<BM, BN, 4>(a, grid, s, zero_page, c64);
    else if (stages >= 3) launch_glds_s<BM, BN, 3>(a, grid, s, zero_page, c64);
    else launch_glds_s<BM, BN, 2>(a, grid, s, zero_page, c64);
}

// A 256-byte page of zeros per device (source of the DMA for out-of-range / padded operand chunks).  Created once per
// device and never written again: a cache, not state.
std::mutex g_zero_mu;
bf16_t* g_zero_page[16] = {nullptr};

const bf16_t* zero_page_for_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(g_zero_mu);
    if (!g_zero_page[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
        g_zero_page[dev] = (bf16_t*)p;
    }
    return g_zero_page[dev];
}

}  // namespace
const bf16_t* prx_gemm_zero_page() { return zero_page_for_current_device(); }
namespace {

int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

}  // namespace

// ---- per-handle engine state (gemm.h) -------------------------------------------------------------------------------
GemmCtx::GemmCtx() {
    // tuning A/B switches (tools/gemm_bench.py, tools/gemm_tune.py); read once per context
    force_stages = env_int("PRX_GEMM_STAGES", 0);
    xcd_swizzle = env_int("PRX_XCD_SWIZZLE", 2);
    conv_c64 = env_int("PRX_CONV_C64", 1);
    wide_tile = env_int("PRX_WIDE_TILE", 128);
    tile8p = env_int("PRX_GEMM_8P", 128);
    fit = env_int("PRX_GEMM_FIT", 1);
    fit_flags = env_int("PRX_FIT_FLAGS", 1);
    fit_conv = env_int("PRX_FIT_CONV", 1);
    use_glds = env_int("PRX_GEMM_V1", 0) ? 0 : 1;
    rowk = env_int("PRX_GEMM_ROWK", 1);
    rowk_min = env_int("PRX_GEMM_ROWK_MIN", 24 << 20);
}

void prx_gemm_ctx_tile_rule(GemmCtx* c, int M, int N, int K, int mode, int bm, int bn, int splits) {
    if (!c) return;
    if (M <= 0) { c->rules.clear(); return; }
    for (size_t i = 0; i < c->rules.size(); ++i)
        if (c->rules[i].M == M && c->rules[i].N == N && c->rules[i].K == K && c->rules[i].mode == mode) { c->rules.erase(c->rules.begin() + i); break; }
    // a rule names a tile of the 4-wave / 8-phase families; anything else (e.g. a profile record's "fit tile" code bm + 1000 fed
    // back verbatim) matches no kernel and is dropped here instead of at the launch
    const bool known = (bm == 64 && bn == 64) || (bm == 128 && (bn == 64 || bn == 128)) || (bm == 256 && (bn == 128 || bn == 256));
    if (bm > 0 && known) c->rules.push_back({M, N, K, mode, bm, bn, splits});
}

void prx_gemm_ctx_force_tile(GemmCtx* c, int bm, int bn, int splits) {
    if (!c) return;
    if (bm == -1) { c->xcd_swizzle = splits; return; }   // (-1, x, on/off): toggle the XCD-aware tile order
    if (bm == -2) { c->force_stages = splits; return; }  // (-2, x, n): LDS pipeline depth (0 = heuristic)
    if (bm == -3) { c->use_glds = splits; return; }      // (-3, x, on/off): direct-to-LDS v2 kernel vs register-staged v1
    if (bm == -5) { c->conv_c64 = splits; return; }      // (-5, x, on/off): scalar-tap conv gather (Cin % 64 == 0)
    if (bm == -6) { c->tile8p = splits; return; }        // (-6, x, n): 256 x 256 8-phase tiles from n tiles on (0 = never)
    if (bm == -7) { c->fit = splits; return; }           // (-7, x, on/off): fit tiles (gemmfit.hip)
    if (bm == -8) { c->fit_flags = splits; return; }     // (-8, x, bits): fit kernel switches (bit 0: staggered wave groups)
    if (bm == -12) { c->force_fit = splits; return; }    // (-12, x, on/off): a forced 128 x 128 / 128 x 64 / 64 x 64 / 256 x 128 tile means the fit kernel of that shape
    if (bm == -9) { c->fit_conv = splits; return; }      // (-9, x, on/off): fit tiles for the implicit convolutions too
    if (bm == -14) { c->rowk_min = splits > 0 ? splits : (24 << 20); return; }     // (-14, x, n): row-streaming kernels from n output elements on (0 restores the default)
    if (bm == -13) { c->dbg_only = splits; c->dbg_count = 0; return; }   // (-13, x, i): bisection aid -- only the i-th fit convolution (-1: all, counting; -2: off)
    if (bm < 0) return;
    c->force_bm = bm; c->force_bn = bn; c->force_splits = splits;
}

void prx_gemm_ctx_profile_enable(GemmCtx* c, int on) {
    if (!c) return;
    std::lock_guard<std::mutex> lk(c->mu);
    c->prof_on = on != 0;
}

int prx_gemm_ctx_profile_collect(GemmCtx* c, double* total_ms, double* total_flop, long long* launches) {
    double ms = 0, fl = 0;
    long long n = 0;
    if (c) {
        std::lock_guard<std::mutex> lk(c->mu);
        // PRX_GEMM_PROFILE_DUMP=<path>: one CSV row per launch (tools/gemm_shapes.py aggregates them)
        FILE* dump = getenv("PRX_GEMM_PROFILE_DUMP") ? fopen(getenv("PRX_GEMM_PROFILE_DUMP"), "a") : nullptr;
        for (auto& r : c->prof) {
            float t = 0;
            if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) {
                if (dump) fclose(dump);
                return -1;
            }
            ms += t; fl += r.flop;
            if (dump) fprintf(dump, "%d,%d,%d,%d,%d,%d,%d,%.2f\n", r.M, r.N, r.K, r.mode, r.bm, r.bn, r.splits, t * 1e3);
            (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        }
        if (dump) fclose(dump);
        n = (long long)c->prof.size();
        c->prof.clear();
    }
    if (total_ms) *total_ms = ms;
    if (total_flop) *total_flop = fl;
    if (launches) *launches = n;
    return 0;
}

// ---- 256 x 256 8-phase tiles: when, and on how much of M ---------------------------------------------------------------
// One workgroup per CU (128 KB of LDS), so tiles run in ROUNDS of 256: a problem of R full rounds + a few tiles pays a whole
// extra round for the remainder (ViT-L/14 at 256 cutouts: M = 65 792 = 257 row tiles -> 1 028 tiles = 4.02 rounds).  The plan
// gives the full rounds to the 8-phase kernel and hands the last few row tiles to the 4-wave kernels as a second launch
// (rows are independent; every row-indexed pointer of the descriptor is offset).  Costs from tools/micro/gemm8p.hip and
// tools/lib_gemm_compare.py on MI355X (profiles/r03_lib_gemm_compare_8phase.txt): a round costs ~1.7 us per K tile + ~11 us
// (epilogue included); the 4-wave kernels run ~650 TFLOP/s marginal + ~8 us.
struct Plan8p { int main_rows; };     // rows [0, main_rows) on the 8-phase kernel (0: not at all), the rest on the 4-wave kernels
static Plan8p plan_8phase(const GemmDesc& d, const GemmCtx& cx) {
    Plan8p p{0};
    if (cx.tile8p <= 0 || !prx_gemm8p_eligible(d)) return p;
    const int tm = ceil_div(d.M, 256), tn = ceil_div(d.N, 256), tiles = tm * tn;
    if (tiles < cx.tile8p) return p;
    const int n_cu = cx.n_cu > 0 ? cx.n_cu : 256;       // the launching context's device (256 on MI355X; the host-only planner query assumes it)
    const double t_round = 1.7 * (d.K / 64) + 11.0;                                    // us
    auto t_4wave = [&](int rows) { return rows <= 0 ? 0.0 : 2.0 * rows * (double)d.N * d.K / 650e6 + 8.0; };   // us
    const int rounds_all = ceil_div(tiles, n_cu);
    double best = rounds_all * t_round;
    p.main_rows = d.M;
    const int full = tiles / n_cu;                                                      // whole rounds available
    if (full >= 1 && tiles % n_cu != 0) {
        const int rows8 = std::min(tm, (full * n_cu) / tn);                             // row tiles that fit `full` rounds
        const int m_main = std::min(d.M, rows8 * 256);
        const double t = full * t_round + t_4wave(d.M - m_main);
        if (m_main > 0 && m_main < d.M && t < 0.9 * best) { best = t; p.main_rows = m_main; }
    }
    if (t_4wave(d.M) < best) p.main_rows = 0;                                           // the 4-wave kernels win outright
    return p;
}
// the same descriptor restricted to rows [r0, r0 + rows) (row-major A only)
static GemmDesc rows_of(const GemmDesc& d, int r0, int rows) {
    GemmDesc s = d;
    const size_t esz = d.f32 ? 4 : 2;
    s.M = rows;
    s.A = (const char*)d.A + (size_t)r0 * d.lda * (d.a_is_f32 ? 4 : esz);
    if (d.bias_m) s.bias_m = d.bias_m + r0;
    if (d.aux) s.aux = (const char*)d.aux + (size_t)r0 * d.ldaux * esz;
    if (d.resid) s.resid = (d.row16 & 1) ? (const float*)((const char*)d.resid + (size_t)r0 * d.ldr * esz) : d.resid + (size_t)r0 * d.ldr;
    if (d.out_f32) s.out_f32 = d.out_f32 + (size_t)r0 * d.ldc_f32;
    if (d.out_bf16) s.out_bf16 = (char*)d.out_bf16 + (size_t)r0 * d.ldc_bf16 * esz;
    if (d.out_bf16_pre) s.out_bf16_pre = (char*)d.out_bf16_pre + (size_t)r0 * d.ldc_bf16 * esz;
    return s;
}

static int gemm_launch_one(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream, GemmCtx* ctx, int use8p);
// (the 8-wave 256 x 128 member of this family -- PRX_BIG_TILE, measured neutral in round 2 and never the default -- was removed in
// round 5: the descriptor it takes by value had outgrown its register budget (320 bytes of scratch per lane); 256 x 128 is a fit tile)
static bool fourwave_tile(int bm, int bn) { return (bm == 128 && bn == 128) || (bm == 128 && bn == 64) || (bm == 64 && bn == 64); }

int prx_gemm_plan_rows_8phase_impl(const GemmCtx* c, int M, int N, int K) {
    static const GemmCtx k_default;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    GemmDesc d;
    d.a_mode = PRX_A_ROWMAJOR; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K;
    return plan_8phase(d, c ? *c : k_default).main_rows;
}

int prx_gemm_launch(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream, GemmCtx* ctx) {
    static const GemmCtx k_default;      // immutable: heuristics only
    int cur_dev = -1;
    if (ctx && (ctx->n_cu == 0 || (hipGetDevice(&cur_dev) == hipSuccess && cur_dev != ctx->n_cu_dev))) {
        // the planners count tiles against the LAUNCHING device's CUs (cost constants stay MI355X's); re-asked when the context
        // launches on another device than the one the count was taken from
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            { ctx->n_cu = cus; ctx->n_cu_dev = dev; }
        else
            ctx->n_cu = 256;
    }
    const GemmCtx& cx = ctx ? *ctx : k_default;
    bool forced = cx.force_bm != 0;
    if (!forced && !cx.rules.empty()) {          // a per-shape rule overrides the plan for THAT shape only (in-pipeline sweeps)
        const int mode = d.a_mode + 2 * d.up + 4 * d.a_is_f32;
        for (const GemmTileRule& r : cx.rules) forced = forced || (r.M == d.M && r.N == d.N && r.K == d.K && r.mode == mode);
    }
    if (!forced && d.M > 0 && d.N > 0 && d.K > 0) {
        if (cx.rowk && (long long)d.M * d.N >= cx.rowk_min && prx_gemmrow_eligible(d)) return gemm_launch_one(d, ws, ws_bytes, stream, ctx, 2);     // skinny K, very tall: the row-streaming kernel (gemmrow.hip)
        const Plan8p p = plan_8phase(d, cx);
        if (p.main_rows >= d.M) return gemm_launch_one(d, ws, ws_bytes, stream, ctx, 1);
        if (p.main_rows > 0) {
            int r = gemm_launch_one(rows_of(d, 0, p.main_rows), ws, ws_bytes, stream, ctx, 1);
            if (r) return r;
            return gemm_launch_one(rows_of(d, p.main_rows, d.M - p.main_rows), ws, ws_bytes, stream, ctx, 0);
        }
    }
    return gemm_launch_one(d, ws, ws_bytes, stream, ctx, 0);
}

static int gemm_launch_one(const GemmDesc& d, float* ws, size_t ws_bytes, hipStream_t stream, GemmCtx* ctx, int use8p) {
    static const GemmCtx k_default;      // immutable: heuristics only
    const GemmCtx& cx = ctx ? *ctx : k_default;
    PRX_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    PRX_REQUIRE(!(d.f32 && d.h16), "gemm: f32 and h16 are exclusive operand formats");
    const int kq = d.f32 ? 4 : 8;        // elements per 16-byte operand chunk
    PRX_REQUIRE(d.K % kq == 0 && d.ldb % kq == 0, "gemm: K (%d) and ldb (%d) must be multiples of %d", d.K, d.ldb, kq);
    PRX_REQUIRE(d.lda % kq == 0, "gemm: lda (%d) must be a multiple of %d", d.lda, kq);
    PRX_REQUIRE(((uintptr_t)d.A & 15) == 0 && ((uintptr_t)d.B & 15) == 0, "gemm: operands must be 16-byte aligned");
    const bool c64 = d.a_mode == PRX_A_CONV3X3 && d.Cin % BK == 0 && cx.conv_c64 != 0;
    if (d.a_mode == PRX_A_CONV3X3) {
        PRX_REQUIRE(d.Cin % kq == 0 && d.K == 9 * d.Cin, "gemm/conv: need Cin %% %d == 0 and K == 9*Cin (Cin=%d K=%d)", kq, d.Cin, d.K);
        PRX_REQUIRE(d.H > 0 && d.W > 0 && d.M % (d.H * d.W) == 0, "gemm/conv: M (%d) must be NB*H*W (%dx%d)", d.M, d.H, d.W);
        PRX_REQUIRE(d.up != 1 || (d.H % 2 == 0 && d.W % 2 == 0), "gemm/conv: upsample needs even H, W");
        PRX_REQUIRE(d.up >= 0 && d.up <= 2, "gemm/conv: up must be 0 (none), 1 (nearest-2x upsampled source) or 2 (stride-2 source)");
        PRX_REQUIRE(d.up != 2 || d.f32 || (c64 && !d.a_is_f32 && cx.use_glds),
                    "gemm/conv: the stride-2 gather needs a bf16 operand with Cin %% 64 == 0 (Cin=%d)", d.Cin);
    }
    PRX_REQUIRE((d.act != PRX_ACT_MUL_DQUICKGELU && d.act != PRX_ACT_MUL_RELUMASK && d.act != PRX_ACT_RELUMASK_POST) || d.aux,
                "gemm: MUL_DQUICKGELU / MUL_RELUMASK / RELUMASK_POST need aux");
    PRX_REQUIRE(d.act != PRX_ACT_RELUMASK_POST || d.resid, "gemm: RELUMASK_POST masks product + residual: it needs resid");
    // exact mode: the fused GroupNorm statistics live in the fp32-operand fit kernels (gemmfit_f32.hip); a problem those do not take
    // runs without them here and gets the statistics from the norm kernels' own pass right after (gemm_launch_one below)
    PRX_REQUIRE(!d.f32 || d.row16 == 0, "gemm: 16-bit residual / GroupNorm-input streams (row16) belong to the 16-bit operand modes");

    // ---- tile / split-K selection (tools/gemm_tune.py sweeps; MI355X: 256 CUs) -------------------------------
    // Score each tile shape by how well its tile count fills whole "rounds" of resident blocks, weighted by the
    // per-tile efficiency (bigger wave tiles do more MFMA per LDS byte).
    const int n_cu = cx.n_cu > 0 ? cx.n_cu : 256;
    int BM = 128, BN = 128;
    auto ntiles = [&](int bm, int bn) { return ceil_div(d.M, bm) * ceil_div(d.N, bn); };
    {
        struct Cand { int bm, bn, per_cu; double eff; };
        const Cand cands[3] = {{128, 128, 2, 1.0}, {128, 64, 3, 0.85}, {64, 64, 5, 0.75}};
        double best = -1.0;
        for (const Cand& c : cands) {
            if (d.N <= 64 && c.bn > 64) continue;
            const int t = ntiles(c.bm, c.bn);
            const int slots = n_cu * c.per_cu;
            const double fill = (double)t / ((double)ceil_div(t, slots) * slots);
            const double score = fill * c.eff;
            if (score > best) { best = score; BM = c.bm; BN = c.bn; }
        }
        // wide row-major problems (FC1 / W2^T: M=3200, N=3072): the fill model over-rates 64x64 (LDS-bound tiles);
        // sweeps give 128x64 28 us vs 64x64 31 us vs 128x128 31.5 us
        if (cx.wide_tile == 128 && d.a_mode == PRX_A_ROWMAJOR && d.N >= 2048 && BM == 64 && ntiles(128, 128) > 2 * n_cu) { BM = 128; BN = 64; }
    }
    // very large problems (ViT-L/14 at 256 cutouts: M = 65 792): the 8-wave 256 x 128 tile, when it still fills the chip
    // several times over (A/B switch, off by default: see DESIGN.md section 6)
    // fit tiles (gemmfit.hip): one workgroup per CU when a tile grid matches the chip (M = 3200: 240 tiles; the decoder's
    // batch-1 convolutions: K split over the wave groups of a workgroup instead of over workgroups + a reduce launch)
    // (three tile shapes -- 128 x 128, 128 x 64, 64 x 64 -- exist in BOTH kernel families: `fit_tile` says which one is meant)
    bool fit_tile = false;
    if (cx.fit && !use8p && (d.a_mode == PRX_A_ROWMAJOR || cx.fit_conv)) {
        int fbm = 0, fbn = 0;
        prx_gemmfit_plan(d, n_cu, &fbm, &fbn);
        if (fbm && ctx && d.a_mode != PRX_A_ROWMAJOR && cx.dbg_only >= -1) {       // bisection aid: only the dbg_only-th fit convolution since the last reset
            const int idx = ctx->dbg_count++;
            if (cx.dbg_only >= 0 && idx != cx.dbg_only) fbm = 0;
        }
        if (fbm) { BM = fbm; BN = fbn; fit_tile = true; }
    }
    const bool rowk = use8p == 2;
    if (use8p == 1) { BM = 256; BN = 256; fit_tile = false; }          // planned by plan_8phase (prx_gemm_launch)
    if (rowk) { BM = 16; BN = (d.K > 320 || d.a_mode == PRX_A_CONV3X3) ? 80 : (d.N % 160 == 0 ? 160 : (d.N % 128 == 0 ? 128 : 80)); fit_tile = false; }
    // a forced tile selects the fit kernel when the override says so (cx.force_fit), or when only that family has the shape
    if (cx.force_bm) { BM = cx.force_bm; BN = cx.force_bn; fit_tile = prx_gemmfit_tile(BM, BN, nullptr) && (cx.force_fit || !fourwave_tile(BM, BN)); }
    int rule_splits = 0;
    if (!cx.rules.empty()) {
        const int mode = d.a_mode + 2 * d.up + 4 * d.a_is_f32;
        for (const GemmTileRule& r : cx.rules)
            if (r.M == d.M && r.N == d.N && r.K == d.K && r.mode == mode) {
                BM = r.bm; BN = r.bn; rule_splits = r.splits;
                fit_tile = prx_gemmfit_tile(BM, BN, nullptr) && !fourwave_tile(BM, BN);
            }
    }
    if (BM == 256 && BN == 256 && !prx_gemm8p_eligible(d)) { BM = 128; BN = 128; }     // row-major 16-bit operands, K % 128 == 0 only
    if (fit_tile && !prx_gemmfit_eligible(d, BM, BN)) { fit_tile = false; if (!fourwave_tile(BM, BN)) { BM = 128; BN = 128; } }
    if (d.f32 && BM == 256 && !fit_tile) BM = 128;    // the exact mode's 4-wave family has three tiles only (256 x 128 is a fit tile)
    if (!fit_tile && BM == 256 && BN == 128) BM = 128;     // 256 x 128 exists as a fit tile only
    const int bk = d.f32 ? BKF : BK;
    GemmArgs a;
    a.d = d;
    a.tiles_m = ceil_div(d.M, BM);
    a.tiles_n = ceil_div(d.N, BN);
    a.kt_total = ceil_div(d.K, bk);
    int tiles = a.tiles_m * a.tiles_n;
    int splits = 1;
    if (ws && tiles <= n_cu / 2 && a.kt_total >= 16 && !(BM == 256 && BN == 256) && !fit_tile && !rowk) {
        // few tiles, long K (the 16x16 / 32x32 decoder convs): aim at ~320 blocks, >= 4 K tiles per split
        splits = std::max(1, std::min(std::min((320 + tiles / 2) / tiles, a.kt_total / 4), 32));
        while (splits > 1 && (size_t)splits * d.M * d.N * sizeof(float) > ws_bytes) --splits;
    }
    auto al = [](const void* p, size_t a_) { return p == nullptr || ((uintptr_t)p % a_) == 0; };
    const size_t opa = d.f32 ? 16 : 8;   // alignment of a 4-element operand-precision access
    a.vec_epi = (d.N % 4 == 0) && al(d.bias_n, 16) && al(d.resid, (d.row16 & 1) ? 8 : 16) && (d.resid == nullptr || d.ldr % 4 == 0) &&
                al(d.aux, opa) && (d.aux == nullptr || d.ldaux % 4 == 0) && al(d.out_f32, 16) &&
                (d.out_f32 == nullptr || d.ldc_f32 % 4 == 0) && al(d.out_bf16, opa) && al(d.out_bf16_pre, opa) &&
                ((d.out_bf16 == nullptr && d.out_bf16_pre == nullptr) || d.ldc_bf16 % 4 == 0) && al(ws, 16);
    PRX_REQUIRE(d.row16 == 0 || a.vec_epi, "gemm: 16-bit residual / GroupNorm-input streams need the vector epilogue (N %% 4 == 0, aligned operands)");
    if (fit_tile && !a.vec_epi) {               // the fit kernel has the vector epilogue only
        BM = 128; BN = 128; fit_tile = false;
        a.tiles_m = ceil_div(d.M, BM); a.tiles_n = ceil_div(d.N, BN); tiles = a.tiles_m * a.tiles_n;
    }
    if ((cx.force_splits > 0 || rule_splits > 0) && ws && !fit_tile) {
        splits = std::min(rule_splits > 0 ? rule_splits : cx.force_splits, a.kt_total);
        while (splits > 1 && (size_t)splits * d.M * d.N * sizeof(float) > ws_bytes) --splits;
    }
    // measured (tools/gemm_tune.py xcd + bench.py A/B): +10-19% on the row-major N=768 ViT GEMMs, neutral-to-negative
    // on the implicit-conv shapes -> mode 2 (default) applies it to narrow row-major problems only
    a.xcd_swizzle = tiles >= 16 && (cx.xcd_swizzle == 1 || (cx.xcd_swizzle == 2 && d.a_mode == PRX_A_ROWMAJOR && d.N <= 1024));
    // with the streaming producers / consumers split the same way (common.h, PRX_XCD_LOCAL): every tiled launch keeps an
    // XCD on a contiguous eighth of the tile rows, so activation rows stay in one L2 across kernel boundaries
    if (cx.xcd_swizzle == 3) a.xcd_swizzle = tiles >= 16;
    // 8-phase tiles (tools/micro/gemm8p.hip): +11 % at M = 65 792, N = 1024 and at M = 25 216, N = 3072; -2 % at 8192^3
    if (BM == 256 && BN == 256 && cx.xcd_swizzle == 2) a.xcd_swizzle = tiles >= 512 && d.N <= 4096;
    if (fit_tile && cx.xcd_swizzle == 2) a.xcd_swizzle = tiles >= 16;
    if (d.gnb_x) {
        PRX_REQUIRE(d.gn_stats && d.gnb_fstats && d.gnb_gamma && d.gnb_beta && (d.out_f32 || d.out_bf16) && d.act == PRX_ACT_NONE,
                    "gemm: fused GroupNorm-backward sums need gn_stats, gnb_fstats, gnb_gamma, gnb_beta and a plain output");
        PRX_REQUIRE(((uintptr_t)d.gnb_x % 16) == 0 && ((uintptr_t)d.gnb_gamma % 16) == 0 && ((uintptr_t)d.gnb_beta % 16) == 0,
                    "gemm: gnb operands must be 16-byte aligned");
    }
    if (d.f32 && d.gn_stats && !fit_tile) {
        // the 4-wave fp32 kernels have no statistics epilogue: the product without them, then the norm kernels' statistics pass over
        // the result (what the runner would have launched had it not asked for the fusion)
        PRX_REQUIRE(d.out_f32 && d.ldc_f32 == d.N && d.N == 32 * d.gn_gs, "gemm: fp32 GroupNorm statistics need a dense fp32 output with N == 32 * gn_gs");
        GemmDesc d2 = d;
        d2.gn_stats = nullptr; d2.gnb_x = nullptr; d2.gnb_fstats = nullptr; d2.gnb_gamma = d2.gnb_beta = nullptr;
        int r = gemm_launch_one(d2, ws, ws_bytes, stream, ctx, use8p);
        if (r) return r;
        if (d.gnb_x) return prx_groupnorm_bwd_stats(d.out_f32, d.gnb_x, d.gnb_gamma, d.gnb_beta, d.gnb_fstats, d.gn_stats, 1, d.M, d.N, d.gnb_swish, d.gnb_eps, stream);
        return prx_groupnorm_fwd(d.out_f32, nullptr, nullptr, d.gn_stats, nullptr, nullptr, 1, d.M, d.N, 0, 1e-6f, stream, /*zero_stats=*/0, /*stats_ready=*/0);
    }
    if (d.gn_stats) {
        PRX_REQUIRE(a.vec_epi && d.gn_gs >= 4 && d.gn_gs % 4 == 0 && d.N == 32 * d.gn_gs && !d.a_is_f32 && (cx.use_glds || d.f32),
                    "gemm: fused GroupNorm statistics need the v2 kernel's vector epilogue and N == 32 * gn_gs");
    }
    a.fit_flags = fit_tile ? (cx.fit_flags & (15 | 64)) : (cx.fit_flags & 64);   // gemmfit.hip A/B switches (PRX_FIT_FLAGS); bit 6: generic epilogues only
    if (fit_tile && (cx.fit_flags & 32) == 0 && d.N > d.M) a.fit_flags |= 16;      // weight-heavy: column-major tile order (bit 5 of the switch word turns it off)
    a.kt_per_split = ceil_div(a.kt_total, splits);
    if (BM == 256 && BN == 256) a.kt_per_split = (a.kt_per_split + 1) & ~1;      // the 8-phase loop body covers two K tiles
    splits = ceil_div(a.kt_total, a.kt_per_split);
    a.splits = splits;
    a.ws = ws;

    GemmProfRec rec{};
    bool prof = false;
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        prof = ctx->prof_on;
    }
    if (prof) {
        PRX_CHECK_HIP(hipEventCreate(&rec.a));
        PRX_CHECK_HIP(hipEventCreate(&rec.b));
        rec.flop = 2.0 * d.M * d.N * d.K;
        rec.M = d.M; rec.N = d.N; rec.K = d.K; rec.mode = d.a_mode + 2 * d.up + 4 * d.a_is_f32 + 8 * d.f32; rec.bm = fit_tile ? BM + 1000 : (rowk ? BM + 2000 : BM); rec.bn = BN; rec.splits = splits;      // + 1000: the fit kernel of that tile shape
        PRX_CHECK_HIP(hipEventRecord(rec.a, stream));
    }

    dim3 grid(tiles, splits);
    if (rowk) {
        int e = prx_gemmrow_launch(a, n_cu, stream);
        if (e) return e;
    } else if (fit_tile && d.f32) {
        int e = prx_gemmfit_launch(a, BM, BN, grid, stream);      // fp32-operand fit kernels (gemmfit_f32.hip)
        if (e) return e;
    } else if (d.f32) {
        if (BM == 128 && BN == 128) launch_f32<128, 128>(a, grid, stream);
        else if (BM == 128 && BN == 64) launch_f32<128, 64>(a, grid, stream);
        else launch_f32<64, 64>(a, grid, stream);
    } else if (BM == 256 && BN == 256) {
        prx_gemm8p_launch(a, grid, stream);
    } else if (fit_tile) {
        int e = prx_gemmfit_launch(a, BM, BN, grid, stream);
        if (e) return e;
    } else if (!d.a_is_f32 && cx.use_glds) {
        const bf16_t* zp = zero_page_for_current_device();
        PRX_REQUIRE(zp != nullptr, "gemm: could not allocate the zero page");
        // LDS pipeline depth, tuned IN the iteration (tools/gemm_shapes.py), not on hot-cache microbenchmarks: every
        // weight matrix is touched once per iteration (520 MB of packs > the 256 MB MALL), so each K tile of B comes
        // from HBM and a 2-deep ring exposes that latency once per K tile.  A third stage on the 64x64 tiles (48 KB,
        // still 3 workgroups/CU) gives -30 % on the 64^2 decoder convs and -8..-20 % on the other 64x64 launches; on
        // the 128-wide tiles it halves the occupancy and loses 15-25 %.
        int stages = (BM == 64 && BN == 64) ? 3 : 2;
        if (cx.force_stages) stages = cx.force_stages;
        if (BM == 128 && BN == 128) launch_glds<128, 128>(a, grid, stream, zp, stages, c64);
        else if (BM == 128 && BN == 64) launch_glds<128, 64>(a, grid, stream, zp, stages, c64);
        else launch_glds<64, 64>(a, grid, stream, zp, stages, c64);
    } else if (BM == 128 && BN == 128) launch_cfg<128, 128>(a, grid, stream);
    else if (BM == 128 && BN == 64) launch_cfg<128, 64>(a, grid, stream);
    else launch_cfg<64, 64>(a, grid, stream);
    PRX_LAUNCH_CHECK();
    if (splits > 1) {
        size_t total = (size_t)d.M * d.N;
        int blocks = (int)std::min<size_t>((total + 255) / 256, 2048);
        PRX_OP_DISPATCH(d.f32, d.h16, T, hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3(blocks), dim3(256), 0, stream, a));
        PRX_LAUNCH_CHECK();
    }
    if (prof) {
        PRX_CHECK_HIP(hipEventRecord(rec.b, stream));
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->prof.push_back(rec);
    }
    return 0;
}
