// Stand-alone bring-up + timing harness for the 256x256 8-phase GEMM tile of pixray_amd/csrc/gemm8p.h
// (C[M,N] = A[M,K] * Bt[N,K]^T, bf16 / fp16 operands, fp32 accumulate).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pixray_amd/csrc tools/micro/gemm8p.hip -o tools/micro/gemm8p.bin
// Run (GPU box):  tools/micro/gemm8p.bin [M N K] ...   -> refcheck (asymmetric random operands, fp64 host reference on
// sampled rows / columns incl. the ragged edges) + TFLOP/s over 50 timed launches on uniform [-1,1) data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "common.h"
#include "gemm8p.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

void prx_set_error(const char*, ...) {}

static inline uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <typename T16>
__global__ __launch_bounds__(512, 2) void bench_kernel(const bf16_t* A, int lda, const bf16_t* B, int ldb, float* C, int ldc, int M, int N, int K, int tiles_n, int xcd) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[G8_LDS_ELEMS];
    int bid = blockIdx.x;
    if (xcd) bid = (int)xcd_linear(bid, gridDim.x);
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    f32x16 acc[4][2];
    g8_mainloop<T16>(A, lda, B, ldb, M, N, K / G8_BK, tm, tn, lds, acc);
    // plain C-layout stores (the engine's kernel stages through LDS instead)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    const int row0 = tm * 256 + wr * 128 + 4 * (lane >> 5), col0 = tn * 256 + wc * 64 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + j * 32;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < M) C[(size_t)row * ldc + col] = acc[i][j][r];
            }
        }
}

int main(int argc, char** argv) {
    std::vector<int> shapes;
    for (int i = 1; i + 2 < argc; i += 3) { shapes.push_back(atoi(argv[i])); shapes.push_back(atoi(argv[i + 1])); shapes.push_back(atoi(argv[i + 2])); }
    if (shapes.empty()) shapes = {256, 256, 128,  300, 520, 256,  3200, 3072, 768,  3200, 2304, 768,  3200, 768, 3072,  25216, 768, 768,
                                  25216, 3072, 768,  65792, 1024, 4096,  4096, 4096, 4096,  8192, 8192, 8192};
    for (size_t s = 0; s < shapes.size(); s += 3) {
        const int M = shapes[s], N = shapes[s + 1], K = shapes[s + 2];
        if (K % 128) { printf("skip K=%d (needs K %% 128 == 0)\n", K); continue; }
        std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t st = 12345u + (uint32_t)s;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = f2bf(rnd());
        for (auto& v : hB) v = f2bf(rnd());
        bf16_t *dA, *dB; float* dC;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, tiles = tiles_m * tiles_n;
        auto launch = [&](int xcd) {
            hipLaunchKernelGGL(bench_kernel<bf16_t>, dim3(tiles), dim3(512), 0, 0, dA, K, dB, K, dC, N, M, N, K, tiles_n, xcd);
        };
        launch(0);
        CK(hipDeviceSynchronize());
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        // reference on sampled rows x all columns
        double maxerr = 0, maxref = 0; long bad = 0, checked = 0;
        std::vector<int> rows;
        for (int r = 0; r < M; r += (M > 2048 ? 257 : 37)) rows.push_back(r);
        rows.push_back(M - 1); if (M > 130) { rows.push_back(127); rows.push_back(128); rows.push_back(M - 129 > 0 ? M - 129 : 0); }
        for (int r : rows)
            for (int c = 0; c < N; c += (N > 2048 ? 13 : 1)) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * (double)bf2f(hB[(size_t)c * K + k]);
                const double got = hC[(size_t)r * N + c];
                const double err = fabs(got - ref);
                maxerr = fmax(maxerr, err); maxref = fmax(maxref, fabs(ref));
                if (!(err <= 2e-3 * sqrt((double)K) + 1e-3 * fabs(ref))) { if (bad < 5) printf("  MISMATCH r=%d c=%d got %g ref %g\n", r, c, got, ref); ++bad; }
                ++checked;
            }
        // race screen: repeat and compare bitwise
        long diff = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
            launch(rep & 1);
            CK(hipDeviceSynchronize());
            std::vector<float> h2((size_t)M * N);
            CK(hipMemcpy(h2.data(), dC, h2.size() * 4, hipMemcpyDeviceToHost));
            if (memcmp(h2.data(), hC.data(), h2.size() * 4)) ++diff;
        }
        // timing
        for (int xcd = 0; xcd < 2; ++xcd) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 5; ++i) launch(xcd);
            CK(hipEventRecord(e0, 0));
            const int iters = 50;
            for (int i = 0; i < iters; ++i) launch(xcd);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / iters;
            printf("M=%6d N=%5d K=%5d tiles=%4d xcd=%d: %8.1f us  %7.1f TFLOP/s | refcheck %ld pts maxerr %.3g (max|ref| %.3g) bad %ld, rerun-diffs %ld\n",
                   M, N, K, tiles, xcd, us, 2.0 * M * N * K / us * 1e-6, checked, maxerr, maxref, bad, diff);
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
