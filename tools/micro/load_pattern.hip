// Microbenchmark: HBM read rate of the A-stationary GEMM's activation loads.  A is [M, K] bf16; each wave owns 32 rows
// and loads them completely (2*KT loads of 16 B per lane, all in flight), 4 waves per workgroup, M/128 workgroups.
//   PAT 0: MFMA fragment order: lane (c = lane&15, g = lane>>4), load (i, kt) = row 16i + c, bytes 64kt + 16g   (64 B per row per instruction)
//   PAT 1: row-major: load t covers 1 KB = rows (1024/RB) ..: lane l, load t -> byte offset t*1024 + 16l of the wave's 32-row panel
//          (whole rows per instruction: 128-byte lines fully used by one instruction)
//   PAT 2: like 0 but lane (c, g) loads bytes 128*(kt>>1) + 32g + 16(kt&1): 4 lanes x 16 B at 32-byte stride
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int KT, int PAT>
__global__ __launch_bounds__(256) void k(const char* A, unsigned* sink, int M) {
    constexpr int RB = KT * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int mw = blockIdx.x * 128 + wid * 32;
    const char* base = A + (size_t)mw * RB;
    u32x4 v[2 * KT];
    if (PAT == 0) {
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) v[i * KT + kt] = *reinterpret_cast<const u32x4*>(base + (size_t)(16 * i + c) * RB + 64 * kt + 16 * g);
    } else if (PAT == 1) {
#pragma unroll
        for (int t = 0; t < 2 * KT; ++t) v[t] = *reinterpret_cast<const u32x4*>(base + t * 1024 + 16 * lane);
    } else {
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) v[i * KT + kt] = *reinterpret_cast<const u32x4*>(base + (size_t)(16 * i + c) * RB + 128 * (kt >> 1) + 32 * g + 16 * (kt & 1));
    }
    unsigned acc = 0;
#pragma unroll
    for (int t = 0; t < 2 * KT; ++t) acc ^= v[t][0] ^ v[t][1] ^ v[t][2] ^ v[t][3];
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int KT, int PAT> void run(const char* d, unsigned* sink, int M) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipLaunchKernelGGL((k<KT, PAT>), dim3(M / 128), dim3(256), 0, 0, d, sink, M);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<KT, PAT>), dim3(M / 128), dim3(256), 0, 0, d, sink, M);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("K=%d pat=%d: %.1f us  %.2f TB/s\n", KT * 32, PAT, ms / 20 * 1e3, (double)M * KT * 64 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const int M = 98304 * 4;      // 200 / 400 MB: larger than the 256 MB of L2 + MALL
    char* d; hipMalloc(&d, (size_t)M * 1024); hipMemset(d, 1, (size_t)M * 1024);
    unsigned* sink; hipMalloc(&sink, 4);
    run<8, 0>(d, sink, M); run<8, 1>(d, sink, M); run<8, 2>(d, sink, M);
    run<16, 0>(d, sink, M); run<16, 1>(d, sink, M); run<16, 2>(d, sink, M);
    const int M1 = 98304;
    run<8, 0>(d, sink, M1); run<8, 1>(d, sink, M1); run<16, 0>(d, sink, M1); run<16, 1>(d, sink, M1);
    return 0;
}
