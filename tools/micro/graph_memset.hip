// Does a hipMemsetAsync captured into a hipGraph keep its stream order on this ROCm build?  (hipGraph note: round 1
// replaced the memsets in front of the depthwise-conv statistics atomics by fill kernels after intermittent wrong replays.)
// The captured chain mimics what ishara_forward did then, 12 "blocks" deep:  K_prev writes the neighbour buffers | memset(ssum),
// memset(ssq) (2 KiB each, adjacent) | K_acc: atomicAdd into ssum/ssq from 64 workgroups | K_use: reads the sums, writes the next
// block's input.  After R replays a verify kernel has counted every replay whose sums were wrong.  Variant 1 uses memset NODES,
// variant 0 fill kernels.   build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/graph_memset tools/micro/graph_memset.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)
constexpr int NB = 12, C = 512, WG = 64;
__global__ void k_prev(float* nb, int n, float v) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) nb[i] = v; }
__global__ void k_fill(float* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f; }
__global__ void k_acc(float* ssum, float* ssq) { for (int c = threadIdx.x; c < C; c += blockDim.x) { atomicAdd(&ssum[c], 1.0f); atomicAdd(&ssq[c], 2.0f); } }
__global__ void k_use(const float* ssum, const float* ssq, const float* nb, int n, unsigned* bad) {
    int wrong = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) wrong += (ssum[c] != (float)WG) + (ssq[c] != 2.0f * WG);
    for (int i = threadIdx.x; i < n; i += blockDim.x) wrong += (nb[i] != 3.0f);          // neighbours must keep what k_prev wrote
    if (wrong) atomicAdd(bad, 1u);
}
int main(int argc, char** argv) {
    const int replays = argc > 1 ? atoi(argv[1]) : 3000;
    for (int variant = 1; variant >= 0; --variant) {
        float* ws; unsigned* bad;
        const int nbn = 4096;                                  // neighbour floats in front of and behind the two statistics buffers
        const size_t per = (size_t)(nbn + 2 * C + nbn);
        CHECK(hipMalloc(&ws, per * NB * sizeof(float))); CHECK(hipMalloc(&bad, NB * sizeof(unsigned)));
        CHECK(hipMemset(bad, 0, NB * sizeof(unsigned)));
        hipStream_t s; CHECK(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int b = 0; b < NB; ++b) {
            float* base = ws + per * b; float* ssum = base + nbn; float* ssq = ssum + C; float* after = ssq + C;
            hipLaunchKernelGGL(k_prev, dim3(8), dim3(256), 0, s, base, nbn, 3.0f);
            hipLaunchKernelGGL(k_prev, dim3(8), dim3(256), 0, s, after, nbn, 3.0f);
            if (variant) { CHECK(hipMemsetAsync(ssum, 0, C * 4, s)); CHECK(hipMemsetAsync(ssq, 0, C * 4, s)); }
            else { hipLaunchKernelGGL(k_fill, dim3(2), dim3(256), 0, s, ssum, C); hipLaunchKernelGGL(k_fill, dim3(2), dim3(256), 0, s, ssq, C); }
            hipLaunchKernelGGL(k_acc, dim3(WG), dim3(256), 0, s, ssum, ssq);
            hipLaunchKernelGGL(k_use, dim3(1), dim3(256), 0, s, ssum, ssq, base, nbn, bad + b);
            hipLaunchKernelGGL(k_use, dim3(1), dim3(256), 0, s, ssum, ssq, after, nbn, bad + b);
        }
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        size_t nn = 0; CHECK(hipGraphGetNodes(g, nullptr, &nn));
        for (int r = 0; r < replays; ++r) CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        unsigned h[NB]; CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
        unsigned tot = 0; for (int b = 0; b < NB; ++b) tot += h[b];
        printf("variant=%s nodes=%zu replays=%d wrong_checks=%u (of %d)\n", variant ? "memset-node" : "fill-kernel", nn, replays, tot, replays * NB * 2);
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipFree(ws)); CHECK(hipFree(bad)); CHECK(hipStreamDestroy(s));
    }
    return 0;
}
