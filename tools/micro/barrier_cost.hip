// Microbenchmark: cost of an s_barrier-per-iteration loop (4 waves / workgroup, 64 KB LDS -> 2 workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float* out, int iters, int work) {
    __shared__ char smem[65536];
    float acc = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 0.5f;
        if (acc == 12345.f) smem[threadIdx.x] = 1;
    }
    if (acc == 123.f) out[0] = acc + smem[0];
}
int main() {
    float* d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int work : {0, 16, 64, 256}) for (int iters : {0, 48, 480}) {
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, iters, work);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, iters, work);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("work=%3d iters=%3d: %.2f us per launch\n", work, iters, ms / 20 * 1e3);
    }
    return 0;
}
