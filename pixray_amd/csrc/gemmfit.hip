// Fit-tile member of the GEMM engine: host side (tile table, eligibility, planner, launch dispatch) and the GENERIC kernels --
// every tile shape with the engine's run-time epilogue, both 16-bit operand formats.  The device code lives in gemmfit_kernel.h;
// the kernels whose epilogue is specialised at compile time are instantiated in gemmfit_spec_*.hip.
#include "gemmfit_kernel.h"
#include <atomic>

static std::atomic<long long> g_spec_launches{0};       // launches on kernels with a compile-time epilogue (prx_gemm_fit_spec_launches)

namespace {
// the tile shapes this kernel exists in
struct FitTile { int bm, bn, ks, tn; double eff; };  // tn: wave-tile width; eff: relative efficiency at full occupancy (planner weight)
const FitTile kFitTiles[] = {
    {160, 256, 1, 64, 1.00}, {160, 192, 1, 48, 0.97}, {256, 128, 1, 64, 0.95}, {160, 128, 1, 32, 0.90}, {128, 128, 1, 32, 0.85},
    {80, 128, 2, 32, 0.85}, {128, 64, 2, 32, 0.75}, {64, 64, 2, 32, 0.60}, {32, 64, 4, 32, 0.45}, {16, 64, 4, 32, 0.35}, {16, 32, 8, 32, 0.25},
    {256, 16, 1, 16, 0.20},      // few output channels over many pixels (the decoder's conv_out: 128 -> 3, padded to 8): never the planner's general choice
};
const FitTile* fit_tile(int bm, int bn) {
    for (const FitTile& t : kFitTiles)
        if (t.bm == bm && t.bn == bn) return &t;
    return nullptr;
}
}  // namespace

bool prx_gemmfit_tile(int bm, int bn, int* ks) {
    const FitTile* t = fit_tile(bm, bn);
    if (ks) *ks = t ? t->ks : 0;
    return t != nullptr;
}
bool prx_gemmfit_eligible(const GemmDesc& d, int bm, int bn) {
    const FitTile* ft = fit_tile(bm, bn);
    if (!ft) return false;
    const int ks = ft->ks;
    const int wave_tn = ft->tn;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };       // null passes
    if (d.f32) {
        // the exact mode's fp32-operand kernels (FIT_EPI_F32: the decoder tiles): K tiles of 32 floats, a plain epilogue -- alpha,
        // bias_n, fp32 residual, fp32 output(s); everything else stays with the 4-wave fp32 kernels
        static const bool on = [] { const char* e = getenv("PRX_FIT_F32"); return !(e && atoi(e) == 0); }();
        if (!on || bm % 80 == 0 || (bm == 256 && bn == 16)) return false;
        bool stats32 = !d.gnb_x || d.gn_stats;
        if (d.gn_stats) {
            const int lpr = wave_tn / 8;
            stats32 = (lpr & (lpr - 1)) == 0 && d.gn_gs >= 4 && d.gn_gs % 4 == 0 && d.N == 32 * d.gn_gs && bn % d.gn_gs == 0 && al16(d.gnb_x) &&
                      (!d.gnb_x || (!d.resid && d.gnb_fstats && d.gnb_gamma && d.gnb_beta && al16(d.gnb_gamma) && al16(d.gnb_beta)));
        }
        const bool epi = d.N % 8 == 0 && d.act == PRX_ACT_NONE && !d.bias_m && !d.aux && stats32 && !d.out_bf16_pre && d.row16 == 0 &&
                         (d.out_f32 || d.out_bf16) && al16(d.bias_n) && al16(d.resid) && al16(d.out_f32) && al16(d.out_bf16) &&
                         (!d.resid || d.ldr % 4 == 0) && (!d.out_f32 || d.ldc_f32 % 4 == 0) && (!d.out_bf16 || d.ldc_bf16 % 4 == 0);
        const bool a_ok32 = d.a_mode == PRX_A_ROWMAJOR
                                ? (unsigned long long)d.M * d.lda < (1ull << 31)
                                : (d.a_mode == PRX_A_CONV3X3 && d.Cin % 32 == 0 && d.K == 9 * d.Cin && (d.up == 0 || d.up == 1) && d.H > 0 && d.W > 0 &&
                                   d.M % (d.H * d.W) == 0 && (unsigned long long)d.M * d.lda < (1ull << 31));
        return !d.a_is_f32 && epi && a_ok32 && d.K % (32 * ks) == 0 && d.lda % 4 == 0 && d.ldb % 4 == 0 && d.M >= 1 &&
               (unsigned long long)d.N * d.ldb < (1ull << 31);
    }
    // the epilogue handles 8 consecutive columns per lane with 16-byte accesses
    const bool epi_ok = d.N % 8 == 0 && al16(d.bias_n) && al16(d.resid) && al16(d.aux) && al16(d.out_f32) && al16(d.out_bf16) &&
                        al16(d.out_bf16_pre) && (!d.resid || d.ldr % ((d.row16 & 1) ? 8 : 4) == 0) && (!d.aux || d.ldaux % 8 == 0) &&
                        (!d.out_f32 || d.ldc_f32 % 4 == 0) && ((!d.out_bf16 && !d.out_bf16_pre) || d.ldc_bf16 % 8 == 0);
    // GroupNorm sums: the lanes of a column quad are reduced by a butterfly (16 FN / 8 lanes per row: a power of two), a
    // group is a whole number of quads inside the block tile, and the GroupNorm-backward input has the output's layout
    bool stats_ok = true;
    if (d.gn_stats) {
        const int lpr = wave_tn / 8;
        stats_ok = bm % 80 != 0 && (lpr & (lpr - 1)) == 0 && d.gn_gs >= 4 && d.gn_gs % 4 == 0 && d.N == 32 * d.gn_gs && bn % d.gn_gs == 0 && al16(d.gnb_x) &&
                   (!d.gnb_x || (d.gnb_fstats && d.gnb_gamma && d.gnb_beta && al16(d.gnb_gamma) && al16(d.gnb_beta)));
    }
    const bool a_ok = d.a_mode == PRX_A_ROWMAJOR
                          ? (unsigned long long)d.M * d.lda < (1ull << 31)
                          : (d.a_mode == PRX_A_CONV3X3 && bm % 80 != 0 &&          // the 80-row-granular tiles are the token-batch (row-major) ones
                             d.Cin % FIT_BK == 0 && d.K == 9 * d.Cin && (d.up == 0 || d.up == 1) && d.H > 0 && d.W > 0 &&
                             d.M % (d.H * d.W) == 0 && (unsigned long long)d.M * d.lda < (1ull << 31));
    // the epilogue prefetches ONE row operand per output row into shared registers: residual, aux, or the GroupNorm input
    const bool has_res = d.resid != nullptr, has_gnb = d.gnb_x != nullptr;
    const bool one_operand = !(has_res && d.aux) && !(has_gnb && (has_res || d.aux));
    return !d.f32 && !d.a_is_f32 && a_ok && d.K % (FIT_BK * ks) == 0 && epi_ok && stats_ok && one_operand && d.M >= 1 &&
           (unsigned long long)d.N * d.ldb < (1ull << 31);
}
// planner: the fit tile (if any) whose grid fills the chip best; *bm = 0 when the 4-wave kernels should keep the problem.
void prx_gemmfit_plan(const GemmDesc& d, int n_cu, int* bm, int* bn) {
    *bm = *bn = 0;
    double best = 0.0;
    auto consider = [&](int tbm, int tbn, double eff) {
        if (d.N < tbn && !(tbn <= 64)) return;
        if (!prx_gemmfit_eligible(d, tbm, tbn)) return;
        const int tm_ = ceil_div(d.M, tbm), tn_ = ceil_div(d.N, tbn);
        const int wgs = tm_ * tn_;
        if (wgs > 2 * n_cu) return;                                        // a one- or two-round kernel by construction
        const double fill = (double)wgs / ((double)ceil_div(wgs, n_cu) * n_cu);
        const double waste = ((double)tm_ * tbm / d.M) * ((double)tn_ * tbn / d.N);
        const double score = fill * eff / waste;
        if (score > best && fill / waste >= 0.8) { best = score; *bm = tbm; *bn = tbn; }
    };
    // a convolution to a handful of channels over a large map (the decoder's conv_out, 128 -> 3 channels padded to 8, 65 536
    // pixels): the general tiles would run 128-wide MFMA columns for 3 outputs (38 us on the 64 x 64 kernel); 256 pixels x 16 columns
    if (d.a_mode == PRX_A_CONV3X3 && d.N <= 16 && d.M >= 4096 && prx_gemmfit_eligible(d, 256, 16)) { *bm = 256; *bn = 16; return; }
    // weight-heavy products with FEW output rows (a 16^2 / 8^2 / 4^2 map of the StyleLoss extractor, M = 16 ... 256 by 512 channels
    // over K = 4608): no tile grid fills the chip, but the smallest tile with 8 K groups per workgroup still streams the weight
    // matrix through 16 ... 256 workgroups in ~9 us where a 128 x 64 tile with split-K + reduce takes 22 - 47 (measured:
    // profiles/r04_small_m_streaming_vs_ring.txt) -- the fill rule below does not apply to them
    if ((long long)d.M * d.N <= 256ll * 512 && d.K >= 512 && d.N >= 32) {
        if (prx_gemmfit_eligible(d, 16, 32)) { *bm = 16; *bn = 32; return; }
        // K not a multiple of 512 (8 K groups of 64): 4 K groups on the 16 x 64 tile (the decoder's conv_in, 256 -> 512 channels over
        // K = 2304 at 16^2: 128 workgroups instead of a 64 x 64 split-K launch + reduce, 20 + 8 us)
        if (d.N >= 64 && prx_gemmfit_eligible(d, 16, 64)) { *bm = 16; *bn = 64; return; }
    }
    for (const FitTile& t : kFitTiles)
        if (t.bn >= 32) consider(t.bm, t.bn, t.eff);
}
extern std::atomic<long long> g_prx_gemm8p_spec_launches;      // gemm8p.hip: the 8-phase kernel's specialised instances
extern "C" long long prx_gemm_fit_spec_launches(void) {
    return g_spec_launches.load(std::memory_order_relaxed) + g_prx_gemm8p_spec_launches.load(std::memory_order_relaxed);
}

// The compile-time epilogue (FIT_EPI_*) a descriptor is an instance of, FIT_EPI_GENERIC when none: IEEE-half operands, one
// 16-bit output (plus FC1's 16-bit pre-activation), no fp32 output, no per-row bias, at most one 16-bit row operand.
int prx_gemmfit_epi_kind(const GemmDesc& d) {
    if (!d.h16 || d.f32 || d.a_is_f32 || d.out_f32 || !d.out_bf16 || d.bias_m) return FIT_EPI_GENERIC;
    // 32-bit element offsets from the operand bases
    const unsigned long long lim = 1ull << 31;
    if ((unsigned long long)d.M * d.ldc_bf16 >= lim || (d.resid && (unsigned long long)d.M * d.ldr >= lim) ||
        (d.aux && (unsigned long long)d.M * d.ldaux >= lim) || (unsigned long long)d.M * d.N >= lim) return FIT_EPI_GENERIC;
    const bool stats = d.gn_stats != nullptr, gnb = d.gnb_x != nullptr;
    if (gnb && !stats) return FIT_EPI_GENERIC;
    switch (d.act) {
    case PRX_ACT_NONE:
        if (d.aux) return FIT_EPI_GENERIC;
        if (gnb) return (!d.resid && (d.row16 & 2)) ? FIT_EPI_GNB : FIT_EPI_GENERIC;
        if (d.resid) return (d.row16 & 1) ? (stats ? FIT_EPI_RES16_GN : FIT_EPI_RES16) : FIT_EPI_GENERIC;
        return stats ? FIT_EPI_GN : FIT_EPI_OUT16;
    case PRX_ACT_QUICKGELU:
        return (d.out_bf16_pre && !d.resid && !stats) ? FIT_EPI_GELU : FIT_EPI_GENERIC;
    case PRX_ACT_MUL_DQUICKGELU:
        return (d.aux && !d.resid && !stats) ? FIT_EPI_DGELU : FIT_EPI_GENERIC;
    default:
        return FIT_EPI_GENERIC;
    }
}
#ifdef PRX_FIT_TRACE
// diagnostic build: between prx_fit_trace_begin(buffer) and prx_fit_trace_end every fit launch writes its waves' phase stamps
// to its own slice of the buffer (instead of the caller's workspace); the host keeps one text record per launch
namespace {
unsigned long long* g_trace_buf = nullptr;
size_t g_trace_cap = 0, g_trace_used = 0;
std::string g_trace_log;
}
extern "C" void prx_fit_trace_begin(void* buf, size_t bytes) { g_trace_buf = (unsigned long long*)buf; g_trace_cap = bytes / 8; g_trace_used = 0; g_trace_log.clear(); }
extern "C" long long prx_fit_trace_end(char* out, size_t cap) {
    g_trace_buf = nullptr;
    if (out && cap) { size_t n = std::min(cap - 1, g_trace_log.size()); memcpy(out, g_trace_log.data(), n); out[n] = 0; }
    return (long long)g_trace_log.size();
}
#endif
int prx_gemmfit_launch(const prx_gemm_dev::GemmArgs& a_in, int bm, int bn, dim3 grid, hipStream_t s) {
    const bf16_t* zp = prx_gemm_zero_page();
    PRX_REQUIRE(zp != nullptr, "gemm: could not allocate the zero page");
#ifdef PRX_FIT_TRACE
    prx_gemm_dev::GemmArgs a = a_in;
    if (g_trace_buf) {
        const size_t need = (size_t)grid.x * 8 * 8;
        if (g_trace_used + need <= g_trace_cap) {
            a.ws = reinterpret_cast<float*>(g_trace_buf + g_trace_used);
            char line[256];
            const GemmDesc& d = a.d;
            snprintf(line, sizeof line, "%zu %u %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", g_trace_used, grid.x, bm, bn, d.M, d.N, d.K, d.a_mode, d.up, d.act,
                     (int)(d.out_f32 != nullptr), (int)(d.out_bf16 != nullptr) + (int)(d.out_bf16_pre != nullptr), (int)(d.resid != nullptr) + 2 * (int)(d.aux != nullptr),
                     (int)(d.gn_stats != nullptr) + 2 * (int)(d.gnb_x != nullptr), (int)d.row16, a.fit_flags);
            g_trace_log += line;
            g_trace_used += need;
        } else a.ws = nullptr;
    } else a.ws = nullptr;
    static const int trace_rep = getenv("PRX_FIT_TRACE_REP") ? atoi(getenv("PRX_FIT_TRACE_REP")) : 0;
    if (trace_rep) a.fit_flags |= 128;
    static const int trace_loop = getenv("PRX_FIT_TRACE_LOOP") ? atoi(getenv("PRX_FIT_TRACE_LOOP")) : 0;       // 1: no MFMAs, 2: no DMA in the steady state
    a.fit_flags |= (trace_loop & 3) << 8;
#else
    const prx_gemm_dev::GemmArgs& a = a_in;
#endif
    // the descriptor patterns of the two runners have kernels with a compile-time epilogue (gemmfit_spec_*.hip); bit 6 of the
    // switch word (PRX_FIT_FLAGS / override -8) keeps every launch on the generic kernels (A/B runs, tests of the generic path)
    if (a.d.f32) {
        PRX_REQUIRE(prx_gemmfit_launch_f32(a, bm, bn, grid, s, zp), "gemmfit: no fp32-operand kernel for a %d x %d tile", bm, bn);
        g_spec_launches.fetch_add(1, std::memory_order_relaxed);       // FIT_EPI_F32 is a compile-time epilogue too
        return 0;
    }
    if (!(a.fit_flags & 64)) {
        const int epi = prx_gemmfit_epi_kind(a.d);
        if (epi != FIT_EPI_GENERIC &&
            (prx_gemmfit_launch_spec_tower(a, bm, bn, epi, grid, s, zp) || prx_gemmfit_launch_spec_dec_a(a, bm, bn, epi, grid, s, zp) ||
             prx_gemmfit_launch_spec_dec_b(a, bm, bn, epi, grid, s, zp) || prx_gemmfit_launch_spec_dec_c(a, bm, bn, epi, grid, s, zp))) {
            g_spec_launches.fetch_add(1, std::memory_order_relaxed);
            return 0;
        }
    }
    if (bm == 160 && bn == 256) launch_fit<2, 4, 5, 4, 1, false>(a, grid, s, zp);
    else if (bm == 160 && bn == 192) launch_fit<2, 4, 5, 3, 1, false>(a, grid, s, zp);
    else if (bm == 256 && bn == 128) launch_fit<4, 2, 4, 4, 1>(a, grid, s, zp);
    else if (bm == 160 && bn == 128) launch_fit<2, 4, 5, 2, 1, false>(a, grid, s, zp);
    else if (bm == 128 && bn == 128) launch_fit<2, 4, 4, 2, 1>(a, grid, s, zp);
    else if (bm == 80 && bn == 128) launch_fit<1, 4, 5, 2, 2, false>(a, grid, s, zp);
    else if (bm == 128 && bn == 64) launch_fit<2, 2, 4, 2, 2>(a, grid, s, zp);
    else if (bm == 64 && bn == 64) launch_fit<2, 2, 2, 2, 2>(a, grid, s, zp);
    else if (bm == 32 && bn == 64) launch_fit<1, 2, 2, 2, 4>(a, grid, s, zp);
    else if (bm == 16 && bn == 64) launch_fit<1, 2, 1, 2, 4>(a, grid, s, zp);
    else if (bm == 16 && bn == 32) launch_fit<1, 1, 1, 2, 8>(a, grid, s, zp);
    else if (bm == 256 && bn == 16) launch_fit<8, 1, 2, 1, 1>(a, grid, s, zp);
    else PRX_REQUIRE(false, "gemmfit: no kernel for a %d x %d tile", bm, bn);
    return 0;
}
