// 256 x 256 x 64 "8-phase" MFMA main loop for gfx950 (16-bit operands, fp32 accumulate), the K loop of the large-problem
// GEMM kernels of gemm.hip.  Structure after /opt/skills/guides/cdna_hip_programming.md section 5 ("The 256^2 8-phase
// template"), re-derived here for v_mfma_f32_32x32x16 and this engine's operand layout (A[M,K], Bt[N,K], K contiguous):
//
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns a 128 x 64 slab of C = 4 x 2 accumulators of 32 x 32 (128
//     registers), worked through as four 64 x 32 QUADRANTS per K tile: (m0,n0) (m0,n1) (m1,n1) (m1,n0).  One phase = one
//     quadrant = 8 MFMAs (256 matrix-pipe cycles); 4 phases per K tile, the loop body covers 2 K tiles (8 phases).
//   * LDS: 2 buffers x {A0, A1, B0, B1} half-tiles of 128 rows x 64 k (16 KB each) = 128 KB (slot order: G8_SLOT_*).  A half-tile is NOT a
//     contiguous 128-row block of the operand: A0 holds the m0 rows of BOTH wave rows (tile rows 0..63 and 128..191), A1
//     the m1 rows; B0 / B1 the n0 / n1 columns of all four wave columns.  That way a half-tile is read in exactly one
//     phase (A0 in phase 1, B1 in phase 2, A1 in phase 3, and the B0 of the NEXT tile in phase 4, whose fragments then stay in
//     registers for phases 1 and 4 of that tile) and its slot can be refilled early -- the whole point of the schedule.
//   * Operands go HBM -> LDS by global_load_lds_dwordx4 (2 instructions per wave per half-tile), one half-tile per
//     phase, each staged 5 phases before its (single) read phase -- table in front of G8_BODY.  Counted waits only:
//     s_waitcnt vmcnt(8) in every phase (everything but the FOUR newest half-tiles has landed, which is exactly the one
//     read in the next phase), never vmcnt(0) in the steady state; raw s_barrier (a __syncthreads() would drain the queue).
//   * The two wave rows run staggered by one barrier: while waves 0-3 issue their MFMA cluster, waves 4-7 (their SIMD
//     partners) issue fragment reads + DMA, and vice versa -- the matrix pipe of every SIMD alternates between its two
//     waves.  s_setprio(1) around the clusters.
//   * Hazards (section 5 of the guide, "Read a staged buffer one phase AFTER the wait that retires it"):
//       RAW  a half-tile is read >= 1 phase after the counted wait + barrier that retires it;
//       WAR  a slot is restaged >= 3 phases after its last ds_read (>= 2 is the requirement with the staggered wave rows).
//   * Swizzle: the DMA destination is lane-linear, so LDS chunk position c of row r holds source chunk c ^ ((r >> 1) & 7)
//     (gemm.hip's scheme: every 16-lane ds_read_b128 group hits 16 distinct 16-byte slots), applied on the source
//     address and on the fragment read.
//
// Requirements: K % 128 == 0 (an even number of K tiles), operands 16-byte aligned, lda / ldb multiples of 8.  Rows >= M
// and columns >= N are CLAMPED on the source side (they produce garbage in accumulator entries that are never stored).
#pragma once
#include "common.h"
#include <type_traits>

constexpr int G8_BK = 64;
constexpr int G8_HALF = 128 * G8_BK;             // elements of one half-tile slot
constexpr int G8_LDS_ELEMS = 2 * 4 * G8_HALF;    // 128 KB

// LDS slot order: [A0 A1 of buffer 0 | A0 A1 of buffer 1 | B0 B1 of buffer 0 | B0 B1 of buffer 1].  All A slots lie in the first 64 KB
// and all B slots in the second, so every fragment read is "one of 4 (A) / 4 (B) per-lane base registers + a 16-bit immediate"
// (ds_read offsets are unsigned 16-bit: with the buffers in separate 64 KB halves each buffer needed its own base registers,
// which pushed the kernel over the 256-register budget and the compiler spilled them -- scratch reloads in the K loop).
#define G8_SLOT_A(buf, h) ((buf) * 2 + (h))
#define G8_SLOT_B(buf, h) (4 + (buf) * 2 + (h))

typedef const __attribute__((address_space(1))) void* g8_gptr;
typedef __attribute__((address_space(3))) void* g8_lptr;

template <typename T16>
__device__ __forceinline__ f32x16 g8_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (std::is_same<T16, half_t>::value)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// acc[i][j]: rows wr*128 + i*32 .., columns wc*64 + j*32 .. of the 256 x 256 tile (wr = wave >> 2, wc = wave & 3), in the
// 32x32 MFMA C layout.  `lds` must be the kernel's ONLY __shared__ object (a second one makes hipcc drain the DMA queue
// in front of every ds_read); on return every wave has passed a final barrier and the LDS is free for the epilogue.
template <typename T16>
__device__ __forceinline__ void g8_mainloop(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                            int M, int N, int nk, int tm, int tn, bf16_t* lds, f32x16 (&acc)[4][2]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- DMA coordinates: instruction i of a half-tile writes slot rows i*64 + wave*8 .. +7, lane -> (row, chunk) ----
    const int srow = wave * 8 + (lane >> 3);                       // 0..63
    const int schunk = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;       // source k offset of this lane's 16 bytes
    unsigned voffA[2][2], voffB[2][2];                             // [instruction][half], bytes from the tile base
    const int mrem = M - 1 - tm * 256, nrem = N - 1 - tn * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int lr = i * 128 + h * 64 + srow;                                      // A: slot rows 0..63 = wave row 0, 64.. = wave row 1
            lr = lr < mrem ? lr : mrem;
            voffA[i][h] = ((unsigned)lr * (unsigned)lda + (unsigned)schunk) * 2u;
            int lc = (i * 2 + (srow >> 5)) * 64 + h * 32 + (srow & 31);            // B: slot rows 32*wc' .. = wave column wc'
            lc = lc < nrem ? lc : nrem;
            voffB[i][h] = ((unsigned)lc * (unsigned)ldb + (unsigned)schunk) * 2u;
        }
    const char* Abase = reinterpret_cast<const char*>(A + (size_t)tm * 256 * lda);
    const char* Bbase = reinterpret_cast<const char*>(B + (size_t)tn * 256 * ldb);
    bf16_t* const wslot = lds + wave * (8 * G8_BK);                // this wave's 8 rows inside instruction 0 of a slot

#define G8_STAGE_A(buf, h, kt)                                                                                          \
    do {                                                                                                                \
        const char* s_ = Abase + (size_t)(kt) * (G8_BK * 2);                                                            \
        __builtin_amdgcn_global_load_lds((g8_gptr)(s_ + voffA[0][h]), (g8_lptr)(wslot + G8_SLOT_A(buf, h) * G8_HALF), 16, 0, 0);             \
        __builtin_amdgcn_global_load_lds((g8_gptr)(s_ + voffA[1][h]), (g8_lptr)(wslot + G8_SLOT_A(buf, h) * G8_HALF + 64 * G8_BK), 16, 0, 0); \
    } while (0)
#define G8_STAGE_B(buf, h, kt)                                                                                          \
    do {                                                                                                                \
        const char* s_ = Bbase + (size_t)(kt) * (G8_BK * 2);                                                            \
        __builtin_amdgcn_global_load_lds((g8_gptr)(s_ + voffB[0][h]), (g8_lptr)(wslot + G8_SLOT_B(buf, h) * G8_HALF), 16, 0, 0);             \
        __builtin_amdgcn_global_load_lds((g8_gptr)(s_ + voffB[1][h]), (g8_lptr)(wslot + G8_SLOT_B(buf, h) * G8_HALF + 64 * G8_BK), 16, 0, 0); \
    } while (0)

    // ---- fragment coordinates -------------------------------------------------------------------------------------
    const int l31 = lane & 31, khalf = lane >> 5, fkey = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + khalf) ^ fkey) * 8;
    const bf16_t* const arow = lds + (wr * 64 + l31) * G8_BK;      // + rb * 32 rows
    const bf16_t* const brow = lds + (wc * 32 + l31) * G8_BK;

    bf16x8 afr[2][4], bn1[4], bn0e[4], bn0o[4];     // A fragments (m0 / m1 in turn), B n1, B n0 of the even / odd K tile
#define G8_READ_A(buf, h)                                                                                               \
    do {                                                                                                                \
        _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                             \
            _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                         \
                afr[rb_][ks_] = *reinterpret_cast<const bf16x8*>(arow + G8_SLOT_A(buf, h) * G8_HALF + rb_ * (32 * G8_BK) + koff[ks_]); \
    } while (0)
#define G8_READ_B(buf, h, dst)                                                                                          \
    do {                                                                                                                \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                             \
            dst[ks_] = *reinterpret_cast<const bf16x8*>(brow + G8_SLOT_B(buf, h) * G8_HALF + koff[ks_]);            \
    } while (0)
#define G8_MMA(mh, nh, bfr)                                                                                             \
    do {                                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                             \
            _Pragma("unroll") for (int rb_ = 0; rb_ < 2; ++rb_)                                                         \
                acc[(mh) * 2 + rb_][nh] = g8_mfma<T16>(afr[rb_][ks_], bfr[ks_], acc[(mh) * 2 + rb_][nh]);               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)
#define G8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_LGKM(n) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_VM(n) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// one phase: fragment reads + one half-tile of DMA + the counted wait (L section) | barrier | 8 MFMAs (C section) | barrier
#define G8_PHASE(READS, STAGE, WAIT, MMA)                                                                               \
    do { READS; STAGE; WAIT; G8_BAR(); G8_LGKM(0); MMA; G8_BAR(); } while (0)

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: K tile 0 in buffer 0 and B0, A0 of K tile 1 in buffer 1 issued; B0 / A0 of tile 0 landed, its n0 B
    // fragments in registers.  (Steady state: FOUR half-tiles in flight behind every wait.)
    G8_STAGE_B(0, 0, 0); G8_STAGE_A(0, 0, 0); G8_STAGE_B(0, 1, 0); G8_STAGE_A(0, 1, 0);
    G8_STAGE_B(1, 0, 1); G8_STAGE_A(1, 0, 1);
    G8_VM(8);
    G8_BAR();
    G8_READ_B(0, 0, bn0e);
    G8_LGKM(0);
    if (wr == 1) G8_BAR();          // the stagger: waves 4-7 run one barrier behind waves 0-3

    // One loop body = K tiles t (buffer 0, phases 1-4) and t + 1 (buffer 1, phases 5-8).  Per phase: which half-tile is read
    // (it was staged 5 phases earlier and retired by the PREVIOUS phase's wait), which one is staged (its slot was last read
    // >= 3 phases ago), and vmcnt(8) = everything but the four newest half-tiles has landed.
    //   ph  reads            stages         |  ph  reads            stages
    //   1   A0(t)            B1(t+1)        |  5   A0(t+1)          B1(t+2)
    //   2   B1(t)            A1(t+1)        |  6   B1(t+1)          A1(t+2)
    //   3   A1(t)            B0(t+2)        |  7   A1(t+1)          B0(t+3)
    //   4   B0(t+1) -> bn0o  A0(t+2)        |  8   B0(t+2) -> bn0e  A0(t+3)
    // (12 / 4 / 8 / 0 fragment reads per phase in the textbook order become 8 / 4 / 8 / 4: the n0 B fragments of the NEXT tile
    // are fetched in the otherwise read-free fourth phase into a second register set.)
    // STG = false: the last body -- only tile t + 1 is still completed, the waits drain the queue.
#define G8_BODY(STG)                                                                                                    \
    do {                                                                                                                \
        G8_PHASE(G8_READ_A(0, 0),       G8_STAGE_B(1, 1, t + 1),                 G8_VM(8),                     G8_MMA(0, 0, bn0e)); \
        G8_PHASE(G8_READ_B(0, 1, bn1),  G8_STAGE_A(1, 1, t + 1),                 G8_VM(8),                     G8_MMA(0, 1, bn1));  \
        G8_PHASE(G8_READ_A(0, 1),       if (STG) G8_STAGE_B(0, 0, t + 2),        if (STG) G8_VM(8); else G8_VM(6), G8_MMA(1, 1, bn1));  \
        G8_PHASE(G8_READ_B(1, 0, bn0o), if (STG) G8_STAGE_A(0, 0, t + 2),        if (STG) G8_VM(8); else G8_VM(4), G8_MMA(1, 0, bn0e)); \
        G8_PHASE(G8_READ_A(1, 0),       if (STG) G8_STAGE_B(0, 1, t + 2),        if (STG) G8_VM(8); else G8_VM(2), G8_MMA(0, 0, bn0o)); \
        G8_PHASE(G8_READ_B(1, 1, bn1),  if (STG) G8_STAGE_A(0, 1, t + 2),        if (STG) G8_VM(8); else G8_VM(0), G8_MMA(0, 1, bn1));  \
        G8_PHASE(G8_READ_A(1, 1),       if (STG) G8_STAGE_B(1, 0, t + 3),        if (STG) G8_VM(8),            G8_MMA(1, 1, bn1));  \
        G8_PHASE(if (STG) G8_READ_B(0, 0, bn0e), if (STG) G8_STAGE_A(1, 0, t + 3), if (STG) G8_VM(8),          G8_MMA(1, 0, bn0o)); \
    } while (0)

    int t = 0;
    for (; t + 2 < nk; t += 2) G8_BODY(true);
    G8_BODY(false);
    if (wr == 0) G8_BAR();          // re-align the two wave rows: every wave has now executed the same number of barriers

#undef G8_PHASE
#undef G8_BODY
#undef G8_VM
#undef G8_LGKM
#undef G8_BAR
#undef G8_MMA
#undef G8_READ_B
#undef G8_READ_A
#undef G8_STAGE_B
#undef G8_STAGE_A
}
