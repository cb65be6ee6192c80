// C-ABI core: error reporting, library info, kernel-level GEMM entry point.
#include "common.h"
#include "gemm.h"
#include "../../include/prx.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[1024] = "";

void prx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int prx_xcd_local() {
    static const int v = getenv("PRX_XCD_LOCAL") ? atoi(getenv("PRX_XCD_LOCAL")) : 1;     // immutable after first use
    return v;
}

int prx_grad_target_log2() {
    static const int v = [] {
        const char* e = getenv("PRX_GRAD_TARGET_LOG2");
        int k = e ? atoi(e) : 4;
        return k < -8 ? -8 : (k > 14 ? 14 : k);
    }();
    return v;
}

extern "C" {

const char* prx_last_error(void) { return g_err; }

int prx_abi_version(void) { return PRX_ABI_VERSION; }

int prx_device_info(int* cu_count, char* arch_name, int arch_name_len) {
    int dev = 0;
    PRX_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    PRX_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return 0;
}

static int desc_of(const prx_gemm_args* g, GemmDesc& d) {
    PRX_REQUIRE(g != nullptr, "prx_k_gemm: null args");
    PRX_REQUIRE(prec_valid(g->f32), "prx_k_gemm: unknown operand precision %d", g->f32);
    d.f32 = prec_is_f32(g->f32); d.h16 = prec_is_h16(g->f32);
    d.A = g->A; d.a_is_f32 = g->a_is_f32; d.a_mode = g->a_mode; d.lda = g->lda;
    d.B = g->B; d.ldb = g->ldb;
    d.M = g->M; d.N = g->N; d.K = g->K;
    d.H = g->H; d.W = g->W; d.Cin = g->Cin; d.up = g->up;
    d.alpha = g->alpha;
    d.bias_n = g->bias_n; d.bias_m = g->bias_m;
    d.aux = g->aux; d.ldaux = g->ldaux;
    d.resid = g->resid; d.ldr = g->ldr;
    d.act = g->act;
    d.out_f32 = g->out_f32; d.ldc_f32 = g->ldc_f32;
    d.out_bf16 = g->out_bf16; d.out_bf16_pre = g->out_bf16_pre;
    d.ldc_bf16 = g->ldc_bf16;
    d.row16 = (short)(g->row16 & 3);
    return 0;
}
int prx_k_gemm(const prx_gemm_args* g, void* ws, size_t ws_bytes, prx_stream_t stream_) {
    GemmDesc d;
    int r = desc_of(g, d);
    if (r) return r;
    return prx_gemm_launch(d, (float*)ws, ws_bytes, (hipStream_t)stream_, (GemmCtx*)g->ctx);
}
int prx_k_gemm_gn(const prx_gemm_args* g, double* gn_stats, int gn_gs, const float* gnb_x, const double* gnb_fstats,
                  const float* gnb_gamma, const float* gnb_beta, int gnb_swish, float gnb_eps, void* ws, size_t ws_bytes,
                  prx_stream_t stream_) {
    GemmDesc d;
    int r = desc_of(g, d);
    if (r) return r;
    PRX_REQUIRE(gn_stats != nullptr, "prx_k_gemm_gn: null gn_stats");
    d.gn_stats = gn_stats; d.gn_gs = gn_gs;
    d.gnb_x = gnb_x; d.gnb_fstats = gnb_fstats; d.gnb_gamma = gnb_gamma; d.gnb_beta = gnb_beta; d.gnb_swish = gnb_swish; d.gnb_eps = gnb_eps;
    return prx_gemm_launch(d, (float*)ws, ws_bytes, (hipStream_t)stream_, (GemmCtx*)g->ctx);
}

prx_gemm_ctx* prx_gemm_ctx_create(void) { return (prx_gemm_ctx*)new GemmCtx(); }
void prx_gemm_ctx_destroy(prx_gemm_ctx* c) { delete (GemmCtx*)c; }
void prx_profile_gemm_enable(prx_gemm_ctx* c, int on) { prx_gemm_ctx_profile_enable((GemmCtx*)c, on); }
void prx_gemm_tile_override(prx_gemm_ctx* c, int bm, int bn, int splits) { prx_gemm_ctx_force_tile((GemmCtx*)c, bm, bn, splits); }
int prx_gemm_plan_rows_8phase(prx_gemm_ctx* c, int M, int N, int K) { return prx_gemm_plan_rows_8phase_impl((const GemmCtx*)c, M, N, K); }
void prx_gemm_tile_rule(prx_gemm_ctx* c, int M, int N, int K, int mode, int bm, int bn, int splits) {
    prx_gemm_ctx_tile_rule((GemmCtx*)c, M, N, K, mode, bm, bn, splits);
}
int prx_profile_gemm_collect(prx_gemm_ctx* c, double* total_ms, double* total_flop, long long* launches) {
    return prx_gemm_ctx_profile_collect((GemmCtx*)c, total_ms, total_flop, launches);
}

}  // extern "C"
