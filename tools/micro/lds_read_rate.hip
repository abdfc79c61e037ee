// Microbenchmark: LDS read throughput per CU for the fragment-read instructions of the GEMM kernels:
//   0: ds_read_b128 (16 B/lane, conflict-free linear)   1: ds_read_b64 (8 B/lane)   2: ds_read_b64_tr_b16 (8 B/lane, transposing)
// 256 workgroups x 512 threads (8 waves per CU), each wave issues ITER x 8 independent reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int base = ((wid * 8 + u + it) & 31) * 2048;
            if (MODE == 0) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (base & 0xffff) + lane * 16);
                acc ^= v.x ^ v.w;
            } else if (MODE == 1) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(smem + (base & 0xffff) + lane * 8);
                acc ^= v.x ^ v.y;
            } else {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + (base & 0xffff) + lane * 8)   /* linear: the four 16-lane row groups 128 B apart (256 B apart would alias the banks 2-way) */);
                acc ^= (unsigned)v.x ^ ((unsigned)v.w << 16);
            }
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int MODE> void run(unsigned* sink, const char* name, int bytes_per_lane) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)iters * 8 * 8 * 64 * bytes_per_lane;       // per CU (8 waves)
    printf("%-22s %.1f us  %.1f B/clk/CU (at 2.4 GHz)  %.1f TB/s chip\n", name, ms * 1e3, bytes_cu / (ms * 1e-3) / 2.4e9, bytes_cu * 256 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned* sink; hipMalloc(&sink, 4);
    run<0>(sink, "ds_read_b128", 16); run<1>(sink, "ds_read_b64", 8); run<2>(sink, "ds_read_b64_tr_b16", 8);
    return 0;
}
