// Host side of the row-streaming GEMM kernels (gemmrow_kernel.h): eligibility, slab / grid plan, dispatch to the instance units.
#include "gemmrowconv_kernel.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace {
std::atomic<long long> g_row_launches{0};
// 16-column tiles per slab: 160-wide slabs (5 pairs), 128-wide (4 pairs), or 80 = 2 pairs + a lone tile
int slab_tiles(int N) { return N % 160 == 0 ? 10 : (N % 128 == 0 ? 8 : (N % 80 == 0 ? 5 : 0)); }
}  // namespace

// The (activation, residual) patterns that exist as kernels (gemmrow_kernel.h launch_instance): the half mode's 16-bit residual
// stream (lean layout) and bf16's fp32 residual, or none.  (Whether a problem is LARGE enough is the caller's rule: GemmCtx::rowk_min.)
bool prx_gemmrow_eligible(const GemmDesc& d) {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!d.f32 && !d.a_is_f32 && d.a_mode == PRX_A_CONV3X3) {
        // implicit 3x3 convolutions with Cin, N in {40, 80} or Cin = N = 160 (gemmrowconv_kernel.h): bias + ReLU, or the ReLU mask of `aux`
        const bool small = (d.Cin == 40 || d.Cin == 80) && (d.N == 40 || d.N == 80);
        if (d.up != 0 || !(small || (d.Cin == 160 && d.N == 160)) || d.K != 9 * d.Cin) return false;
        if (d.bias_m || d.gn_stats || d.gnb_x || d.out_bf16_pre || d.resid) return false;
        if (d.act != PRX_ACT_RELU && d.act != PRX_ACT_MUL_RELUMASK) return false;
        if (d.act == PRX_ACT_MUL_RELUMASK && (!d.aux || !al16(d.aux) || d.ldaux % 8 != 0)) return false;
        if (!al16(d.A) || !al16(d.B) || d.lda % 8 != 0 || d.ldb % 8 != 0) return false;
        if ((size_t)d.M * d.lda * 2 >= ((size_t)1 << 31)) return false;          // 32-bit tap offsets inside one image are always fine; keep the map addressable
        if (d.out_f32 && (!al16(d.out_f32) || d.ldc_f32 % 4 != 0)) return false;
        if (d.out_bf16 && (!al16(d.out_bf16) || d.ldc_bf16 % 8 != 0)) return false;
        return d.out_f32 || d.out_bf16;
    }
    if (d.f32 || d.a_is_f32 || d.a_mode != PRX_A_ROWMAJOR) return false;
    const int nt = d.K > 320 ? (d.N % 80 == 0 ? 5 : 0) : slab_tiles(d.N);      // K in (320, 640]: 80-column slabs only
    if (d.K > 640 || d.K % 8 != 0 || nt == 0) return false;
    if (d.bias_m || d.gn_stats || d.gnb_x || d.out_bf16_pre) return false;
    const bool odd = (nt & 1) != 0;
    if (odd) {      // 80-column slabs: conv1 forward / conv3 dgrad (no residual) and every residual pattern of the wider slabs
        if (d.act != PRX_ACT_NONE && d.act != PRX_ACT_RELU && d.act != PRX_ACT_MUL_RELUMASK && d.act != PRX_ACT_RELUMASK_POST) return false;
        if (d.resid && d.act == PRX_ACT_MUL_RELUMASK) return false;
        if (d.act == PRX_ACT_RELUMASK_POST && !d.resid) return false;
    } else {
        if (d.act != PRX_ACT_NONE && d.act != PRX_ACT_RELU && d.act != PRX_ACT_RELUMASK_POST) return false;
        if (d.act == PRX_ACT_RELUMASK_POST && !d.resid) return false;
    }
    const bool has_aux = d.act == PRX_ACT_MUL_RELUMASK || d.act == PRX_ACT_RELUMASK_POST;
    if (has_aux && !d.aux) return false;
    if (d.resid && ((d.row16 & 1) != 0) != (d.h16 != 0)) return false;       // half: 16-bit residual streams; bf16: fp32 residuals
    if (!al16(d.A) || !al16(d.B) || d.lda % 8 != 0 || d.ldb % 8 != 0) return false;
    if (d.resid && (!al16(d.resid) || d.ldr % 8 != 0)) return false;
    if (has_aux && (!al16(d.aux) || d.ldaux % 8 != 0)) return false;
    if (d.out_f32 && (!al16(d.out_f32) || d.ldc_f32 % 4 != 0)) return false;
    if (d.out_bf16 && (!al16(d.out_bf16) || d.ldc_bf16 % 8 != 0)) return false;
    return d.out_f32 || d.out_bf16;
}

int prx_gemmrow_launch(const prx_gemm_dev::GemmArgs& a, int n_cu, hipStream_t s) {
    const GemmDesc& d = a.d;
    if (d.a_mode == PRX_A_CONV3X3) {
        // one persistent workgroup per CU, each on a contiguous run of 16-pixel tiles (neighbouring image rows: the 9 taps of a pixel
        // are fetched from HBM once and from the L2 of the chunk's XCD after that... as far as the round-robin of workgroups allows)
        const int row_tiles = (d.M + 15) / 16, cus = n_cu > 0 ? n_cu : 256;
        const bool ok = d.h16 ? prx_gemmrowconv_launch_h(a, row_tiles, cus, s) : prx_gemmrowconv_launch_b(a, row_tiles, cus, s);
        PRX_REQUIRE(ok, "gemmrow: no convolution instance for N %d, Cin %d, act %d (eligibility and instances disagree)", d.N, d.Cin, d.act);
        g_row_launches.fetch_add(1, std::memory_order_relaxed);
        return 0;
    }
    const int ksteps = (d.K + 31) / 32;
    const int nt = ksteps > 10 ? 5 : slab_tiles(d.N), nslab = d.N / (nt * 16);
    const int row_tiles = (d.M + 15) / 16;
    // one workgroup of 8 waves per CU (the kernels hold 130 - 220 registers: two waves per SIMD); the grid is a whole number of
    // (8 XCDs x nslab) groups
    static const int per_cu = [] { const char* e = getenv("PRX_GEMM_ROWK_WGS"); return e ? std::max(1, atoi(e)) : 1; }();
    const int group = 8 * nslab;
    // (the 80-column slabs: <= 128 registers and 32 / 52 KB of LDS -- two workgroups per CU keep more of their long activation rows in flight)
    const int wgs = (nt == 5 && ksteps <= 10) ? 2 * per_cu : per_cu;
    const int grid = std::max(1, (wgs * (n_cu > 0 ? n_cu : 256)) / group) * group;
    const int nchunks = (grid / group) * 8;
    bool ok;
    if (d.h16) ok = ksteps <= 6 ? prx_gemmrow_launch_h6(a, nt, ksteps, nslab, row_tiles, nchunks, grid, s)
                  : (ksteps <= 10 ? prx_gemmrow_launch_h10(a, nt, ksteps, nslab, row_tiles, nchunks, grid, s)
                                  : prx_gemmrow_launch_h20(a, ksteps, nslab, row_tiles, nchunks, grid, s));
    else ok = ksteps <= 6 ? prx_gemmrow_launch_b6(a, nt, ksteps, nslab, row_tiles, nchunks, grid, s)
            : (ksteps <= 10 ? prx_gemmrow_launch_b10(a, nt, ksteps, nslab, row_tiles, nchunks, grid, s)
                            : prx_gemmrow_launch_b20(a, ksteps, nslab, row_tiles, nchunks, grid, s));
    PRX_REQUIRE(ok, "gemmrow: no kernel instance for act %d, residual %d, N %d (eligibility and instances disagree)", d.act, d.resid != nullptr, d.N);
    g_row_launches.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

long long prx_gemmrow_launches() { return g_row_launches.load(std::memory_order_relaxed); }
extern "C" long long prx_gemm_row_launches(void) { return prx_gemmrow_launches(); }
