// Microbenchmark: VALU issue rate of the integer ops a counter-hash RNG is made of (per-CU wave-instructions per clock):
//   0: v_xor/v_lshr mix (full-rate baseline)   1: v_mul_lo_u32   2: v_mul_u32_u24   3: v_exp_f32 (transcendental reference)
// 256 workgroups x 1024 threads (4 waves per SIMD), 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* sink, int iters, unsigned seed) {
    unsigned v[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + threadIdx.x * 9781u + i * 77u; f[i] = (float)(threadIdx.x + i) * 1e-3f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) v[i] = (v[i] >> 7) ^ (v[i] + 0x9E3779B9u);
            else if (MODE == 1) v[i] = v[i] * 0x7feb352du + 1u;
            else if (MODE == 2) v[i] = __umul24(v[i], 0x7feb35u) + 1u;
            else f[i] = __builtin_amdgcn_exp2f(f[i]) * 0.5f;
        }
    }
    unsigned a = 0;
    for (int i = 0; i < 8; ++i) a ^= v[i] ^ __float_as_uint(f[i]);
    if (a == 0x12345u) sink[0] = a;
}
template <int MODE> void run(unsigned* sink, const char* name, int ops_per_iter) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, sink, iters, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, sink, iters, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)iters * 8 * ops_per_iter * 16;          // wave-instructions per CU (16 waves)
    printf("%-16s %.1f us   %.2f wave-instr / (us * SIMD)  (%d VALU ops per element-step)\n", name, ms * 1e3, winstr / 4 / (ms * 1e3), ops_per_iter);
}
int main() {
    unsigned* sink; hipMalloc(&sink, 4);
    run<0>(sink, "xor/shift/add", 3); run<1>(sink, "v_mul_lo_u32+add", 2); run<2>(sink, "v_mul_u32_u24+add", 2); run<3>(sink, "v_exp_f32+mul", 2);
    return 0;
}
