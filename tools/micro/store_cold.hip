// Microbenchmark: HBM write rate of the A-stationary GEMM's store stream into COLD memory.  Inside the model every launch writes an output
// tensor the chip has not touched for a whole step (10 GB of workspace), while a microbenchmark that rewrites one 100 MB buffer keeps it in
// the 256 MB Infinity Cache (tools/gemm_cold.py: the same GEMM launch takes 46 us warm, 60 us cold).  Here NBUF distinct [M, N] bf16 outputs
// are written in rotation (cold) or one of them repeatedly (warm), 768 workgroups x 4 waves x 32 rows as in gemm_as.hip:
//   PAT 3 / COLS 32: 16 B per lane, row = lane&15, piece = lane>>4: 16 rows x 64 B per instruction, the other half of each 128-B line comes
//                    one column step later (what gemm_as.hip emits today)
//   PAT 3 / COLS 64: the same lanes, two instructions back to back complete a row's 128-B line
//   PAT 2 / COLS 64: row = lane>>3, 8 lanes x 16 B = one 128-B line per row per instruction, 8 rows per instruction
//   NT: 0 plain store, 1 __builtin_nontemporal_store
//   D: s_waitcnt vmcnt(D) after each step; WORK: s_sleep between steps (a column step of the GEMM takes ~1-2 us)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
template <int NT> __device__ __forceinline__ void st16(unsigned short* p, u4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p)); else *reinterpret_cast<u4*>(p) = v;
}
template <int D, int COLS, int PAT, int NT, int WORK>
__global__ __launch_bounds__(256) void k(unsigned short* out, int M, int N) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int mw = blockIdx.x * 128 + wid * 32;
    const int nsteps = N / COLS;
    for (int s = 0; s < nsteps; ++s) {
        const int n0 = s * COLS;
        const u4 v = {(unsigned)n0, (unsigned)s, (unsigned)lane, 3u};
        if (PAT == 3) {
            const int r = lane & 15, p = lane >> 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < COLS / 32; ++j) st16<NT>(out + (size_t)(mw + 16 * i + r) * N + n0 + 32 * j + 8 * p, v);
        } else {
            const int r = lane >> 3, p = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < COLS / 64; ++j) st16<NT>(out + (size_t)(mw + 8 * i + r) * N + n0 + 64 * j + 8 * p, v);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        if (WORK) { for (int w = 0; w < WORK; ++w) asm volatile("s_sleep 8" ::: "memory"); }
    }
}
template <int D, int COLS, int PAT, int NT, int WORK> void run(unsigned short** bufs, int nbuf, int M, int N) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int cold = 0; cold < 2; ++cold) {
        for (int i = 0; i < nbuf; ++i) hipLaunchKernelGGL((k<D, COLS, PAT, NT, WORK>), dim3(M / 128), dim3(256), 0, 0, bufs[cold ? i : 0], M, N);
        hipEventRecord(e0);
        for (int i = 0; i < 2 * nbuf; ++i) hipLaunchKernelGGL((k<D, COLS, PAT, NT, WORK>), dim3(M / 128), dim3(256), 0, 0, bufs[cold ? i % nbuf : 0], M, N);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[cold], e0, e1);
        ms[cold] /= 2 * nbuf;
    }
    const double gb = (double)M * N * 2 / 1e12;
    printf("N=%4d cols/step=%2d pat=%d nt=%d vmcnt(%2d) work=%d: warm %.1f us %.2f TB/s | cold %.1f us %.2f TB/s\n", N, COLS, PAT, NT, D, WORK,
           ms[0] * 1e3, gb / (ms[0] * 1e-3), ms[1] * 1e3, gb / (ms[1] * 1e-3));
    fflush(stdout);
}
int main() {
    const int M = 98304, NBUF = 16;
    unsigned short* bufs[NBUF];
    for (int i = 0; i < NBUF; ++i) { hipMalloc(&bufs[i], (size_t)M * 512 * 2); hipMemset(bufs[i], 0, (size_t)M * 512 * 2); }
    for (int N : {512, 256}) {
        run<8, 32, 3, 0, 0>(bufs, NBUF, M, N); run<8, 32, 3, 1, 0>(bufs, NBUF, M, N);
        run<8, 64, 3, 0, 0>(bufs, NBUF, M, N); run<8, 64, 3, 1, 0>(bufs, NBUF, M, N);
        run<8, 64, 2, 0, 0>(bufs, NBUF, M, N); run<8, 64, 2, 1, 0>(bufs, NBUF, M, N);
        run<32, 32, 3, 0, 0>(bufs, NBUF, M, N); run<32, 64, 2, 0, 0>(bufs, NBUF, M, N);
        run<4, 32, 3, 0, 4>(bufs, NBUF, M, N); run<4, 64, 2, 0, 8>(bufs, NBUF, M, N);
        run<2, 32, 3, 0, 4>(bufs, NBUF, M, N); run<16, 32, 3, 0, 4>(bufs, NBUF, M, N);
        // spaced steps (a GEMM column step between the stores): does completing a row's 128-B line by two back-to-back instructions recover the
        // full-line rate?  and the non-temporal hint on each shape
        run<4, 64, 3, 0, 8>(bufs, NBUF, M, N); run<4, 64, 3, 1, 8>(bufs, NBUF, M, N);
        run<4, 64, 2, 1, 8>(bufs, NBUF, M, N); run<4, 32, 3, 1, 4>(bufs, NBUF, M, N);
        run<4, 32, 3, 0, 2>(bufs, NBUF, M, N); run<4, 64, 3, 0, 4>(bufs, NBUF, M, N); run<4, 64, 2, 0, 4>(bufs, NBUF, M, N);
    }
    return 0;
}
