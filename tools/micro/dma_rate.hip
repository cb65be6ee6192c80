// Micro-benchmark (run on the GPU box): how fast can ONE CU pull L2-resident data into LDS?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/dma_rate.bin tools/micro/dma_rate.hip && tools/micro/dma_rate.bin
// Variants: direct-to-LDS global_load_lds_dwordx4 (what the fit / 8-phase GEMM rings use) and register-staged
// global_load_dwordx4 + ds_write_b128, for 4 / 8 / 16 waves per CU, rows of 128 bytes (a K tile of 64 16-bit elements)
// gathered from a panel that fits the L2 (2 MB per workgroup set) -- one workgroup per CU, 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

template <int WAVES, bool DMA, int INFLIGHT>
__global__ __launch_bounds__(64 * WAVES) void pull_kernel(const char* __restrict__ src, size_t panel_bytes, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[WAVES * INFLIGHT * 1024 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + ((size_t)blockIdx.x % 8) * 0;      // every workgroup walks the same panel: L2 hits after the first touch
    size_t off = ((size_t)wave * 64 + lane) * 16 + (size_t)(blockIdx.x % 7) * 4096;
    const size_t stride = (size_t)WAVES * 1024;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (DMA) {
#pragma unroll
            for (int q = 0; q < INFLIGHT; ++q) {
                __builtin_amdgcn_global_load_lds((gptr)(base + off), (lptr)(lds + (wave * INFLIGHT + q) * 1024), 16, 0, 0);
                off += stride; if (off >= panel_bytes) off -= panel_bytes;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float4 v[INFLIGHT];
#pragma unroll
            for (int q = 0; q < INFLIGHT; ++q) {
                v[q] = *reinterpret_cast<const float4*>(base + off);
                off += stride; if (off >= panel_bytes) off -= panel_bytes;
            }
#pragma unroll
            for (int q = 0; q < INFLIGHT; ++q) *reinterpret_cast<float4*>(lds + (wave * INFLIGHT + q) * 1024 + lane * 16) = v[q];
        }
    }
    __syncthreads();
    acc += *reinterpret_cast<float*>(lds + threadIdx.x * 4);
    if (acc == 12345.678f) sink[0] = acc;
}

template <int WAVES, bool DMA, int INFLIGHT>
void run(const char* src, size_t panel, float* sink, const char* tag) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pull_kernel<WAVES, DMA, INFLIGHT><<<grid, 64 * WAVES>>>(src, panel, 50, sink);
    hipEventRecord(e0);
    pull_kernel<WAVES, DMA, INFLIGHT><<<grid, 64 * WAVES>>>(src, panel, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * WAVES * INFLIGHT * 1024.0 * iters;
    printf("%-34s waves %2d in flight %d panel %5.1f MB: %7.1f GB/s per CU, %6.2f TB/s chip (%.3f ms)\n", tag, WAVES, INFLIGHT, panel / 1048576.0,
           bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, ms);
}

int main() {
    const size_t big = 512u << 20;
    char* src; float* sink;
    hipMalloc(&src, big); hipMemset(src, 1, big); hipMalloc(&sink, 16);
    for (size_t panel : {(size_t)1 << 20, (size_t)3 << 20, (size_t)24 << 20, (size_t)400 << 20}) {
        run<8, true, 4>(src, panel, sink, "global_load_lds_dwordx4");
        run<8, true, 8>(src, panel, sink, "global_load_lds_dwordx4");
        run<16, true, 4>(src, panel, sink, "global_load_lds_dwordx4");
        run<4, true, 8>(src, panel, sink, "global_load_lds_dwordx4");
        run<8, false, 4>(src, panel, sink, "global_load_dwordx4 + ds_write_b128");
        run<8, false, 8>(src, panel, sink, "global_load_dwordx4 + ds_write_b128");
        run<16, false, 4>(src, panel, sink, "global_load_dwordx4 + ds_write_b128");
    }
    return 0;
}
