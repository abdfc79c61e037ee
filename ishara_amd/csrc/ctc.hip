// CTC loss + gradient (conv-hybrid-model.ipynb c6:1-13 -> tf.nn.ctc_loss semantics with
// blank = last class, logit_length = T) and the greedy decoder (c8:4-12).
//
// One 512-thread workgroup per sample.  The lattice has S = 2*len+1 <= 2L+1 states;
// phase 1 runs the alpha recursion (threads 0..S-1, one barrier per frame, state vector
// double-buffered in LDS), phase 2 the beta recursion, folding alpha+beta into state
// posteriors in place in the fp32 workspace [B,T,S]; phase 3 scatters posteriors to
// classes in parallel over (t, c):  dlogits = grad_scale * (softmax - posterior).
#include "kernels.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -2)
#define CTC_NEG (-1e30)

// log(e^a + e^b + e^c) with the running state in fp64 and the transcendentals in fp32: the
// lattice values reach ~-1500 at T=384 where an fp32 ulp is 1.2e-4 and the drift over T steps
// reaches 1e-3 relative in the posteriors; fp64 add/max keeps the drift at the 1e-6 level while
// exp/log only ever see small-magnitude differences.
DEVI double lse3(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (m <= -1e29) return CTC_NEG;
    const float sum = __expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m));
    return m + (double)__logf(sum);
}

size_t ctc_workspace_floats(int B, int T, int L) { return 2 * (size_t)B * T * (2 * L + 1); }   // fp64 lattice

__global__ __launch_bounds__(512) void ctc_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  int Tn, int C, int L, int blank, float* __restrict__ nll,
                                                  float* __restrict__ dlogits, float grad_scale, double* __restrict__ ws) {
    extern __shared__ double shd[];
    const int Smax = 2 * L + 1;
    double* buf = shd;                                   // [2][Smax + 2]
    float* lse = reinterpret_cast<float*>(buf + 2 * (Smax + 2));   // [Tn]
    int* ext = reinterpret_cast<int*>(lse + Tn);         // [Smax]
    __shared__ int s_len;
    __shared__ double s_logp;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (size_t)b * Tn * C;
    const int64_t* lab = labels + (size_t)b * L;
    double* wsb = ws + (size_t)b * Tn * Smax;

    if (tid == 0) { int n = 0; for (int i = 0; i < L; ++i) n += (lab[i] != blank) ? 1 : 0; s_len = n; }
    for (int t = tid; t < Tn; t += blockDim.x) {
        float m = -1e30f;
        for (int c = 0; c < C; ++c) m = fmaxf(m, lg[(size_t)t * C + c]);
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += expf(lg[(size_t)t * C + c] - m);
        lse[t] = m + logf(a);
    }
    for (int s = tid; s < Smax; s += blockDim.x) ext[s] = (s & 1) ? (int)lab[s >> 1] : blank;
    __syncthreads();
    const int len = s_len, S = 2 * len + 1;
    const int s = tid;
    const bool act = s < S;
    const int my = act ? ext[s] : blank;
    const bool skip_ok = act && s >= 2 && my != blank && my != ext[s - 2];          // s-2 -> s
    const bool skip_fw = act && s + 2 < S && ext[s + 2] != blank && ext[s + 2] != my;   // s -> s+2 (beta)

    // ---- phase 1: alpha ----
    {
        double* p0 = buf + 2;                // index -2..Smax-1
        double* p1 = buf + (Smax + 2) + 2;
        if (tid < 2) { buf[tid] = CTC_NEG; buf[(Smax + 2) + tid] = CTC_NEG; }
        double a = CTC_NEG;
        if (act && (s == 0 || (s == 1 && len > 0))) a = (double)(lg[my] - lse[0]);
        if (s < Smax) { p0[s] = a; wsb[s] = a; }
        __syncthreads();
        for (int t = 1; t < Tn; ++t) {
            double* prev = (t & 1) ? p0 : p1;
            double* cur = (t & 1) ? p1 : p0;
            double v = CTC_NEG;
            if (act) {
                const double x2 = skip_ok ? prev[s - 2] : CTC_NEG;
                v = lse3(prev[s], prev[s - 1], x2);
                if (v > -1e29) v += (double)(lg[(size_t)t * C + my] - lse[t]);
            }
            if (s < Smax) { cur[s] = v; wsb[(size_t)t * Smax + s] = v; }
            __syncthreads();
        }
        if (tid == 0) {
            double* last = ((Tn - 1) & 1) ? p1 : p0;
            const double aL = last[S - 1], aL1 = (len > 0) ? last[S - 2] : CTC_NEG;
            const double lp = lse3(aL, aL1, CTC_NEG);
            s_logp = lp;
            nll[b] = (float)(-lp);
        }
        __syncthreads();
    }
    const double logp = s_logp;
    // ---- phase 2: beta, posteriors written over alpha ----
    {
        double* p0 = buf;                    // index 0..Smax+1 (two trailing pads)
        double* p1 = buf + (Smax + 2);
        if (tid < 2) { p0[Smax + tid] = CTC_NEG; p1[Smax + tid] = CTC_NEG; }
        __syncthreads();
        double bt = CTC_NEG;
        const int tl = Tn - 1;
        if (act && (s == S - 1 || (s == S - 2 && len > 0))) bt = (double)(lg[(size_t)tl * C + my] - lse[tl]);
        if (s < Smax) {
            p0[s] = bt;
            const double al = wsb[(size_t)tl * Smax + s];
            const double lpy = (double)(lg[(size_t)tl * C + my] - lse[tl]);
            wsb[(size_t)tl * Smax + s] = (act && al > -1e29 && bt > -1e29) ? (double)__expf((float)(al + bt - lpy - logp)) : 0.0;
        }
        __syncthreads();
        for (int t = Tn - 2, it = 1; t >= 0; --t, ++it) {
            double* nxt = (it & 1) ? p0 : p1;
            double* cur = (it & 1) ? p1 : p0;
            double v = CTC_NEG;
            double lpy = 0.0;
            if (act) {
                lpy = (double)(lg[(size_t)t * C + my] - lse[t]);
                const double x2 = skip_fw ? nxt[s + 2] : CTC_NEG;
                const double x1 = (s + 1 < S) ? nxt[s + 1] : CTC_NEG;
                v = lse3(nxt[s], x1, x2);
                if (v > -1e29) v += lpy;
            }
            if (s < Smax) {
                cur[s] = v;
                const double al = wsb[(size_t)t * Smax + s];
                wsb[(size_t)t * Smax + s] = (act && al > -1e29 && v > -1e29) ? (double)__expf((float)(al + v - lpy - logp)) : 0.0;
            }
            __syncthreads();
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- phase 3: class posteriors and gradient, parallel over (t, c) ----
    if (dlogits) {
        float* dl = dlogits + (size_t)b * Tn * C;
        const int cl = tid & 63, ts = tid >> 6;        // 8 frames in flight x 64 class lanes
        for (int t = ts; t < Tn; t += 8) {
            if (cl < C) {
                const double* g = wsb + (size_t)t * Smax;
                double acc = 0.0;
                if (cl == blank) { for (int s2 = 0; s2 < S; s2 += 2) acc += g[s2]; }
                else { for (int s2 = 1; s2 < S; s2 += 2) if (ext[s2] == cl) acc += g[s2]; }
                const float sm = expf(lg[(size_t)t * C + cl] - lse[t]);
                dl[(size_t)t * C + cl] = grad_scale * (sm - (float)acc);
            }
        }
    }
}

int launch_ctc(const float* logits, const int64_t* labels, int B, int T, int C, int L, int blank,
               float* nll, float* dlogits, float grad_scale, float* ws, hipStream_t s) {
    if (2 * L + 1 > 512 || C > 64) { ishara_set_error("ctc: L=%d (max 255) or C=%d (max 64) unsupported", L, C); return -1; }
    const size_t shmem = (size_t)(2 * (2 * L + 3)) * sizeof(double) + (size_t)T * sizeof(float) + (size_t)(2 * L + 1) * sizeof(int);
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(512), shmem, s, logits, labels, T, C, L, blank, nll, dlogits, grad_scale, reinterpret_cast<double*>(ws));
    return LAUNCH_OK();
}

__global__ void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n, float scale) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}
int launch_mean(const float* v, float* out, int n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, v, out, n, scale);
    return LAUNCH_OK();
}

// greedy decode: argmax per frame (first max on ties), keep x[i] (i <= T-2) where
// x[i] != x[i+1], drop blanks (the reference never emits the final run, c8:7-9).
__global__ __launch_bounds__(256) void greedy_decode_kernel(const float* __restrict__ logits, int Tn, int C, int blank,
                                                            int* __restrict__ out_idx, int* __restrict__ out_len) {
    extern __shared__ int shi[];   // am[Tn], flag[Tn]
    int* am = shi;
    int* pos = shi + Tn;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (size_t)b * Tn * C;
    for (int t = tid; t < Tn; t += blockDim.x) {
        float best = lg[(size_t)t * C];
        int bi = 0;
        for (int c = 1; c < C; ++c) { const float v = lg[(size_t)t * C + c]; if (v > best) { best = v; bi = c; } }
        am[t] = bi;
    }
    __syncthreads();
    for (int t = tid; t < Tn; t += blockDim.x) pos[t] = (t + 1 < Tn && am[t] != am[t + 1] && am[t] != blank) ? 1 : 0;
    __syncthreads();
    if (tid == 0) {   // T <= 512: a serial exclusive scan is cheaper than a parallel one here
        int n = 0;
        for (int t = 0; t < Tn; ++t) { const int f = pos[t]; pos[t] = f ? n : -1; n += f; }
        out_len[b] = n;
    }
    __syncthreads();
    for (int t = tid; t < Tn; t += blockDim.x) {
        if (pos[t] >= 0) out_idx[(size_t)b * Tn + pos[t]] = am[t];
    }
}

int launch_greedy_decode(const float* logits, int B, int T, int C, int blank, int* out_idx, int* out_len, hipStream_t s) {
    (void)hipMemsetAsync(out_idx, 0xFF, (size_t)B * T * sizeof(int), s);   // -1 fill
    hipLaunchKernelGGL(greedy_decode_kernel, dim3(B), dim3(256), 2 * T * sizeof(int), s, logits, T, C, blank, out_idx, out_len);
    return LAUNCH_OK();
}
