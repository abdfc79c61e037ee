// Fused multi-tensor optimizer over the flat fp32 parameter buffer:
// tfa.optimizers.RectifiedAdam(sma_threshold=4) wrapped in tfa.optimizers.Lookahead(
// sync_period=5, slow_step_size=0.5) — conv-hybrid-model.ipynb c7:68-69.  The step-dependent
// scalars (bias corrections, rectification r_t, sync flag) are computed on the host.
#include "kernels.h"

__global__ __launch_bounds__(256) void radam_lookahead_kernel(float* __restrict__ theta, const float* __restrict__ grad,
                                                              float* __restrict__ m, float* __restrict__ v, float* __restrict__ slow,
                                                              int64_t n, RAdamArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float g = grad[i];
        const float mi = a.beta1 * m[i] + (1.f - a.beta1) * g;
        const float vi = a.beta2 * v[i] + (1.f - a.beta2) * g * g;
        m[i] = mi; v[i] = vi;
        const float mh = mi * a.c1;
        float upd = a.rect ? a.r_t * mh / (sqrtf(vi * a.c2) + a.eps) : mh;
        float th = theta[i];
        if (a.wd != 0.f) upd += a.wd * th;
        th -= a.lr * upd;
        if (a.sync) {
            const float sl = slow[i] + a.slow_step * (th - slow[i]);
            slow[i] = sl;
            th = sl;
        }
        theta[i] = th;
    }
}

int launch_radam_lookahead(float* theta, const float* grad, float* m, float* v, float* slow, int64_t n,
                           RAdamArgs a, hipStream_t s) {
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(radam_lookahead_kernel, dim3(grid), dim3(256), 0, s, theta, grad, m, v, slow, n, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

__global__ void scale_kernel(float* __restrict__ x, int64_t n, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= scale;
}
int launch_scale(float* x, int64_t n, float scale, hipStream_t s) {
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, s, x, n, scale);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
