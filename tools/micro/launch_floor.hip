// Back-to-back launch floor on one stream: empty kernels of several grid sizes, and a dependent chain of tiny
// read-modify-write kernels (hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel() {}
__global__ void touch_kernel(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.f;
}
template <typename F> float time_us(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main() {
    float* p; hipMalloc(&p, 64 << 20); hipMemset(p, 0, 64 << 20);
    for (int blocks : {1, 32, 256, 1024, 2048, 8192})
        for (int threads : {64, 256, 512})
            printf("empty  grid %5d x %3d: %.2f us/launch\n", blocks, threads,
                   time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), 0, 0); }, 2000));
    for (int n : {1 << 10, 1 << 16, 1 << 20, 1 << 22, 1 << 24})
        printf("touch  n=%8d (%5d blocks): %.2f us/launch\n", n, (n + 255) / 256,
               time_us([&] { hipLaunchKernelGGL(touch_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, p, n); }, 2000));
    return 0;
}
