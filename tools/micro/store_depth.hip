// Microbenchmark: HBM write rate of the A-stationary GEMM's store stream.  Each wave owns 32 rows of a [M, N] bf16
// matrix and walks the N columns 32 (or 64) at a time, 768 workgroups of 4 waves (3 per CU) as in gemm_as.hip.
//   PAT 0: 8 B per lane, rows 16i + (lane&15), columns n0 + 16j + 4*(lane>>4)     (accumulator layout)
//   PAT 1: 16 B per lane, row = lane>>2 (16 rows per instruction), 4 lanes x 16 B = 64 B per row
//   PAT 3: 16 B per lane, row = lane&15, piece = lane>>4 (what the permuted-accumulator epilogue of gemm_as.hip emits)
//   PAT 2: 16 B per lane, row = lane>>3 (8 rows per instruction), 8 lanes x 16 B = 128 B per row (COLS must be 64)
//   SKEW: 0 every wave starts at column 0 (all waves write the same column band at the same time);
//         1 workgroup b starts at column band (b % nsteps) and wraps
//   D: s_waitcnt vmcnt(D) after each step (how many stores a wave keeps in flight)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int D, int COLS, int PAT, int SKEW>
__global__ __launch_bounds__(256) void k(unsigned short* out, int M, int N) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int mw = blockIdx.x * 128 + wid * 32;
    const int nsteps = N / COLS;
    int st = SKEW ? (int)(blockIdx.x % nsteps) : 0;
    for (int s = 0; s < nsteps; ++s) {
        const int n0 = st * COLS;
        if (PAT == 0) {
            const int c = lane & 15, g = lane >> 4;
            unsigned short* cb = out + (size_t)(mw + c) * N + 4 * g;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < COLS / 16; ++j)
                    *reinterpret_cast<uint2*>(cb + (size_t)(16 * i) * N + n0 + 16 * j) = make_uint2(n0 + i, j);
        } else if (PAT == 1) {
            const int r = lane >> 2, p = lane & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < COLS / 32; ++j)
                    *reinterpret_cast<uint4*>(out + (size_t)(mw + 16 * i + r) * N + n0 + 32 * j + 8 * p) = make_uint4(n0, i, j, 3);
        } else if (PAT == 3) {      // 16 B per lane in the MFMA lane arrangement: row = lane&15, 16-B piece = lane>>4
            const int r = lane & 15, p = lane >> 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < COLS / 32; ++j)
                    *reinterpret_cast<uint4*>(out + (size_t)(mw + 16 * i + r) * N + n0 + 32 * j + 8 * p) = make_uint4(n0, i, j, 3);
        } else {
            const int r = lane >> 3, p = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < COLS / 64; ++j)
                    *reinterpret_cast<uint4*>(out + (size_t)(mw + 8 * i + r) * N + n0 + 64 * j + 8 * p) = make_uint4(n0, i, j, 3);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
        st = st + 1 == nsteps ? 0 : st + 1;
    }
}
template <int D, int COLS, int PAT, int SKEW> void run(unsigned short* d, int M, int N) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipLaunchKernelGGL((k<D, COLS, PAT, SKEW>), dim3(M / 128), dim3(256), 0, 0, d, M, N);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<D, COLS, PAT, SKEW>), dim3(M / 128), dim3(256), 0, 0, d, M, N);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("N=%4d cols/step=%d pat=%d skew=%d vmcnt(%2d): %.1f us  %.2f TB/s\n", N, COLS, PAT, SKEW, D, ms / 20 * 1e3, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
    const int M = 98304;
    unsigned short* d; hipMalloc(&d, (size_t)M * 1024 * 2);
    for (int N : {512, 256}) {
        run<0, 32, 0, 0>(d, M, N); run<60, 32, 0, 0>(d, M, N);
        run<8, 32, 0, 0>(d, M, N); run<8, 32, 0, 1>(d, M, N);
        run<8, 64, 0, 0>(d, M, N); run<8, 64, 0, 1>(d, M, N);
        run<8, 32, 1, 0>(d, M, N); run<8, 32, 1, 1>(d, M, N);
        run<8, 64, 1, 0>(d, M, N); run<8, 64, 1, 1>(d, M, N);
        run<8, 64, 2, 0>(d, M, N); run<8, 64, 2, 1>(d, M, N);
        run<32, 64, 2, 1>(d, M, N);
        run<8, 32, 3, 0>(d, M, N); run<0, 32, 3, 0>(d, M, N); run<8, 64, 3, 0>(d, M, N);
    }
    return 0;
}
