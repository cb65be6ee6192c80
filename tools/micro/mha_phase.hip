// Phase timing of the T<=64 attention forward/backward (one wave per head): instrumented copies of the kernels with
// s_memtime stamps, plus end-to-end timing of the real kernels.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pixray_amd/csrc tools/micro/mha_phase.hip -o mha_phase
#include "../../pixray_amd/csrc/attention.hip"
#include <cstdio>
#include <cstdarg>
#include <vector>
void prx_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc(10, stderr); }

__global__ __launch_bounds__(64) void fwd_timed(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int T, int C, float scale,
                                                long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE];
    bf16_t* Qs = smem; bf16_t* Ks = smem + TILE; bf16_t* Vt = smem + 2 * TILE; bf16_t* Ps = smem + 3 * TILE;
    const int lane = threadIdx.x;
    const int h = blockIdx.x, n = blockIdx.y;
    const long long ld = 3LL * C;
    const bf16_t* base = qkv + (long long)n * T * ld + h * 64;
    long long t[8]; int k = 0;
    t[k++] = __builtin_amdgcn_s_memtime();
    load_tile(base, ld, T, Qs, nullptr, lane);
    load_tile(base + C, ld, T, Ks, nullptr, lane);
    t[k++] = __builtin_amdgcn_s_memtime();
    load_tile(base + 2 * C, ld, T, nullptr, Vt, lane);
    __syncthreads();
    t[k++] = __builtin_amdgcn_s_memtime();
    f32x16 s[2][2];
    zero_acc(s);
    mma_64x64x64(Qs, Ks, s, lane);
    t[k++] = __builtin_amdgcn_s_memtime();
    softmax_c_layout(s, scale, T, lane);
    t[k++] = __builtin_amdgcn_s_memtime();
    store_c_tile(s, Ps, nullptr, lane);
    __syncthreads();
    t[k++] = __builtin_amdgcn_s_memtime();
    f32x16 o[2][2];
    zero_acc(o);
    mma_64x64x64(Ps, Vt, o, lane);
    t[k++] = __builtin_amdgcn_s_memtime();
    store_c_global(o, out + (long long)n * T * C + h * 64, C, T, lane);
    t[k++] = __builtin_amdgcn_s_memtime();
    if (lane == 0 && h == 3 && n == 5) for (int i = 0; i < 8; ++i) stamps[i] = t[i];
}

int main() {
    const int N = 64, T = 50, C = 768, heads = 12;
    size_t nq = (size_t)N * T * 3 * C;
    std::vector<bf16_t> hq(nq);
    for (size_t i = 0; i < nq; ++i) hq[i] = (bf16_t)(((int)(i * 2654435761u % 2001) - 1000) / 1000.f);
    bf16_t *qkv, *out, *dout, *dqkv; long long* st;
    hipMalloc(&qkv, nq * 2); hipMalloc(&out, nq * 2 / 3); hipMalloc(&dout, nq * 2 / 3); hipMalloc(&dqkv, nq * 2); hipMalloc(&st, 64);
    hipMemcpy(qkv, hq.data(), nq * 2, hipMemcpyHostToDevice);
    hipMemcpy(dout, hq.data(), nq * 2 / 3, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(fwd_timed, dim3(heads, N), dim3(64), 0, 0, qkv, out, T, C, 0.125f, st);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
        const char* names[7] = {"load Q,K", "load V (transposed LDS image)", "S = QK^T", "softmax", "P -> LDS", "O = PV", "store O"};
        printf("rep %d (s_memtime ticks, 100 MHz => x10 ns):", rep);
        for (int i = 0; i < 7; ++i) printf("  %s %lld", names[i], h[i + 1] - h[i]);
        printf("  total %lld\n", h[7] - h[0]);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 3; ++i) { if (which == 0) prx_mha_fwd(qkv, out, N, T, C, heads, 0); else prx_mha_bwd(qkv, dout, dqkv, N, T, C, heads, 0); }
        hipEventRecord(a, 0);
        for (int i = 0; i < 200; ++i) { if (which == 0) prx_mha_fwd(qkv, out, N, T, C, heads, 0); else prx_mha_bwd(qkv, dout, dqkv, N, T, C, heads, 0); }
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.2f us per launch (back-to-back)\n", which == 0 ? "mha_fwd" : "mha_bwd", ms * 1e3f / 200);
    }
    return 0;
}
