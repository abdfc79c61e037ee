// Microbenchmark: HBM write rate of the GEMM-epilogue store pattern vs a linear stream.
//   pattern 0: linear 16-B-per-lane stream
//   pattern 1: each 256-thread block writes a 64-row x SEG-byte tile of a [M, N*2 B] bf16 matrix
//              (8 lanes x 16 B per 128-B piece), tiles enumerated n-fastest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void linear_k(uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_uint4(i, 1, 2, 3);
}
__global__ void tile_k(char* out, int M, int rowbytes, int segbytes) {
    const int nseg = rowbytes / segbytes;
    const int mt = blockIdx.x / nseg, nt = blockIdx.x % nseg;
    const int pieces = segbytes / 16;                 // 16-B pieces per row segment
    for (int c = threadIdx.x; c < 64 * pieces; c += blockDim.x) {
        const int row = c / pieces, p = c % pieces;
        *reinterpret_cast<uint4*>(out + (size_t)(mt * 64 + row) * rowbytes + nt * segbytes + p * 16) = make_uint4(c, 1, 2, 3);
    }
}
int main() {
    const int M = 98304, rowbytes = 1024;             // N=512 bf16
    const size_t bytes = (size_t)M * rowbytes;
    char* d; hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(linear_k, dim3(4096), dim3(256), 0, 0, (uint4*)d, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("linear: %.1f us  %.2f TB/s\n", ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
        for (int seg : {128, 256, 512, 1024}) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tile_k, dim3((M / 64) * (rowbytes / seg)), dim3(256), 0, 0, d, M, rowbytes, seg);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("tile 64 rows x %4d B: %.1f us  %.2f TB/s\n", seg, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
        }
    }
    return 0;
}
