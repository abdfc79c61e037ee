// CTC loss + gradient (conv-hybrid-model.ipynb c6:1-13 -> tf.nn.ctc_loss semantics with
// blank = last class, logit_length = T) and the greedy decoder (c8:4-12).
//
// One workgroup per sample; the lattice has S = 2*len+1 <= 2L+1 states (see ctc_scaled_kernel).
#include <type_traits>
#include "kernels.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -2)

static inline int ctc_ns(int L) { return (2 * L + 1 + 63) / 64; }            // lattice states per lane of the recursion wave
size_t ctc_workspace_floats(int B, int T, int L) { return 4 * (size_t)B * T * 64 * ctc_ns(L); }   // two fp64 lattices [B][T][64*NS]: alpha, beta sums

#define CTC_NEG (-1e30)
// log(e^a + e^b + e^c) with the running state in fp64 and the transcendentals in fp32: the lattice values reach ~-1700 at
// T=384 where an fp32 ulp is 1.2e-4 and the drift over T frames reaches 1e-3 relative in the posteriors; fp64 add/max keeps
// the drift at the 1e-6 level while exp/log only ever see small-magnitude differences.  (A scaled PROBABILITY-space
// recursion is not an option: the state vector of one frame spans > 250 decades on random logits, tools/ notes in DESIGN.md.)
DEVI double lse3(double a, double b, double c) {
    // branch-free: with all three terms dead (m = CTC_NEG) the exponentials are exp(0) and the result is discarded by the select — an early
    // return was an exec-mask branch on the recursions' dependency chain
    const double m = fmax(a, fmax(b, c));
    const float sum = __expf((float)(a - m)) + __expf((float)(b - m)) + __expf((float)(c - m));
    const double r = m + (double)__logf(sum);
    return m <= -1e29 ? CTC_NEG : r;
}
DEVI double shfl_up_d(double v, int d) { return __shfl_up(v, d, 64); }
DEVI double shfl_down_d(double v, int d) { return __shfl_down(v, d, 64); }

// ---------------------------------------------------------------------------------------------------------------
// One 256-thread workgroup per sample:
//   phase 0 (all threads)  lse[t] = logsumexp(logits[t]); extended label sequence
//   phase 1 (wave 0)       alpha recursion in log space: the S = 2*len+1 states live in REGISTERS, state s = lane + 64 k;
//                          only the ceil(S/64) occupied k are computed; the s-1 / s-2 neighbours come from the previous
//                          lanes by two rotations per frame and k.  No barrier and no LDS on the dependency chain (the previous kernel
//                          spent a workgroup barrier and an L2 round trip per frame: 0.9 ms at T=384); the emission
//                          log-probabilities are gathered 8 frames ahead of the chain.
//   phase 1' (wave 1)      beta recursion the same way, CONCURRENTLY with alpha on another SIMD (the two chains are
//                          independent); it stores bsum_t[s] = log sum of the successors' betas
//   phase 2 (all threads)  state posteriors exp(alpha + bsum - logp) scatter-added to classes in LDS, one frame per wave at
//                          a time; dlogits = grad_scale * (softmax - posterior)
// ---------------------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  int Tn, int C, int L, int blank, float* __restrict__ nll,
                                                  float* __restrict__ dlogits, float grad_scale, double* __restrict__ ws,
                                                  uint32_t* __restrict__ dlb) {
    constexpr int SP = 64 * NS;
    extern __shared__ float shf[];
    float* lse = shf;                                   // [Tn]
    int* ext = reinterpret_cast<int*>(lse + Tn);        // [SP]
    __shared__ int s_len;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* lg = logits + (size_t)b * Tn * C;
    const int64_t* lab = labels + (size_t)b * L;
    double* Gw = ws + (size_t)b * 2 * Tn * SP;          // [Tn][SP] alpha
    double* Hw = Gw + (size_t)Tn * SP;                  // [Tn][SP] bsum
    __shared__ double s_logp;

    if (tid == 0) { int n = 0; for (int i = 0; i < L; ++i) n += (lab[i] != blank) ? 1 : 0; s_len = n; }
    for (int t = tid; t < Tn; t += 256) {
        float m = -1e30f;
        for (int c = 0; c < C; ++c) m = fmaxf(m, lg[(size_t)t * C + c]);
        float a = 0.f;
        for (int c = 0; c < C; ++c) a += expf(lg[(size_t)t * C + c] - m);
        lse[t] = m + logf(a);
    }
    for (int s = tid; s < SP; s += 256) ext[s] = (s < 2 * L + 1 && (s & 1)) ? (int)lab[s >> 1] : blank;
    __syncthreads();
    const int len = s_len, S = 2 * len + 1;

    // The recursions, instantiated for a COMPILE-TIME number NK of occupied state registers per lane (1: labels of up to 31 symbols, 2: up to 63)
    // so that the per-k loops are straight-line code; NK = 0 keeps the run-time count (longer labels).  The block picks its instantiation below.
    auto recursions = [&](auto nkc) {
        constexpr int NK = decltype(nkc)::value;
        constexpr int KM = NK ? NK : NS;
        const int wave = tid >> 6;
        // state s = lane + 64 k (k < NS): a label of up to 31 symbols (S <= 63) occupies k = 0 only, and the recursion skips the
        // other k (wave-uniform nk) -- a third of the work of NS consecutive states per lane; lattice rows are written as
        // contiguous 512-byte pieces.  The s-1 / s-2 neighbours come from lanes l-1 / l-2 by a rotation; lanes 0 (and 1) take
        // them from lanes 63 (and 62) of the previous k.
        const int nk = NK ? NK : ((S + 63) >> 6);
        bool act[NS], skip_bw[NS], skip_fw[NS];
        int my[NS];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int s = lane + 64 * k;
            my[k] = ext[s];
            act[k] = s < S;
            skip_bw[k] = act[k] && s >= 2 && my[k] != blank && my[k] != ext[s - 2];                   // s-2 -> s
            skip_fw[k] = s + 2 < S && ext[s + 2] != blank && ext[s + 2] != my[k];                     // s -> s+2
        }
        float em[8][NS], emn[8][NS];
        // RAW logits of frames t0, t0 + dir, .., t0 + 7 dir (loads only: subtracting lse here made hipcc wait for the loads right
        // after issuing them)
        auto gather = [&](int t0, int dir, float (&dst)[8][NS]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = min(max(t0 + dir * u, 0), Tn - 1);
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (NK != 0 || k < nk) dst[u][k] = lg[(size_t)t * C + my[k]];
            }
        };
        // emission log-probabilities of a gathered group, one group later: the loads have landed by then (the empty asm keeps the
        // subtraction from being scheduled up to the loads)
        auto settle = [&](int t0, int dir, float (&src)[8][NS], float (&dst)[8][NS]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float ls = lse[min(max(t0 + dir * u, 0), Tn - 1)];
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (NK != 0 || k < nk) { asm volatile("" : "+v"(src[u][k])); dst[u][k] = src[u][k] - ls; }
            }
        };
      if (wave == 0) {
        // ---- alpha ----
        double a[NS];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int s = lane + 64 * k;
            a[k] = (act[k] && (s == 0 || (s == 1 && len > 0))) ? (double)(lg[my[k]] - lse[0]) : CTC_NEG;
            if (NK != 0 || k < nk) Gw[s] = a[k];
        }
        gather(1, 1, emn);
        settle(1, 1, emn, em);
        for (int t0 = 1; t0 < Tn; t0 += 8) {
            if (t0 + 8 < Tn) gather(t0 + 8, 1, emn);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u;
                if (t < Tn) {
                    double r1[NS], r2[NS], n[NS];
#pragma unroll
                    for (int k = 0; k < KM; ++k)
                        if (NK != 0 || k < nk) { r1[k] = __shfl(a[k], (lane + 63) & 63, 64); r2[k] = __shfl(a[k], (lane + 62) & 63, 64); }
#pragma unroll
                    for (int k = 0; k < KM; ++k)
                        if (NK != 0 || k < nk) {
                            const double a1 = lane >= 1 ? r1[k] : (k >= 1 ? r1[k >= 1 ? k - 1 : 0] : CTC_NEG);
                            const double a2 = lane >= 2 ? r2[k] : (k >= 1 ? r2[k >= 1 ? k - 1 : 0] : CTC_NEG);
                            double v = act[k] ? lse3(a[k], a1, skip_bw[k] ? a2 : CTC_NEG) : CTC_NEG;
                            if (v > -1e29) v += (double)em[u][k];
                            n[k] = v;
                        }
#pragma unroll
                    for (int k = 0; k < KM; ++k)
                        if (NK != 0 || k < nk) { a[k] = n[k]; Gw[(size_t)t * SP + lane + 64 * k] = n[k]; }
                }
            }
            if (t0 + 8 < Tn) settle(t0 + 8, 1, emn, em);
        }
        // log p(y | x) = logsumexp(alpha[S-1], alpha[S-2]): both live in (at most two) lanes; reduce max then sum over the wave
        double logp;
        {
            double cand = CTC_NEG, cand2 = CTC_NEG;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const int s = lane + 64 * k;
                if (s == S - 1) cand = a[k];
                if (s == S - 2 && len > 0) cand2 = a[k];
            }
            double m = fmax(cand, cand2);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
            float e = 0.f;
            if (m > -1e29) e = ((cand > -1e29) ? __expf((float)(cand - m)) : 0.f) + ((cand2 > -1e29) ? __expf((float)(cand2 - m)) : 0.f);
            e = wave_sum(e);
            logp = (m > -1e29) ? m + (double)__logf(e) : CTC_NEG;
            if (lane == 0) { nll[b] = (float)(-logp); s_logp = logp; }
        }
      } else {
        // ---- beta (with emission): bt = beta_{t+1}; stores bsum_t[s] = logsumexp of the successors' betas ----
        double bt[NS];
#pragma unroll
        for (int k = 0; k < KM; ++k) bt[k] = CTC_NEG;
        gather(Tn - 1, -1, emn);
        settle(Tn - 1, -1, emn, em);
        for (int tb = Tn - 1; tb >= 0; tb -= 8) {
            if (tb - 8 >= 0) gather(tb - 8, -1, emn);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = tb - u;
                if (t >= 0) {
                    double bsum[NS];
                    if (t == Tn - 1) {
#pragma unroll
                        for (int k = 0; k < KM; ++k) {
                            const int s = lane + 64 * k;
                            bsum[k] = (act[k] && (s == S - 1 || (s == S - 2 && len > 0))) ? 0.0 : CTC_NEG;
                        }
                    } else {
                        double d1[NS], d2[NS];
#pragma unroll
                        for (int k = 0; k < KM; ++k)
                            if (NK != 0 || k < nk) { d1[k] = __shfl(bt[k], (lane + 1) & 63, 64); d2[k] = __shfl(bt[k], (lane + 2) & 63, 64); }
#pragma unroll
                        for (int k = 0; k < KM; ++k)
                            if (NK != 0 || k < nk) {
                                // successors s+1 / s+2: lanes 63 (and 62) take them from lanes 0 (and 1) of the next k (absent: no state)
                                const bool nx = k + 1 < KM && k + 1 < nk;
                                const double b1 = lane <= 62 ? d1[k] : (nx ? d1[k + 1 < KM ? k + 1 : k] : CTC_NEG);
                                const double b2 = lane <= 61 ? d2[k] : (nx ? d2[k + 1 < KM ? k + 1 : k] : CTC_NEG);
                                bsum[k] = act[k] ? lse3(bt[k], b1, skip_fw[k] ? b2 : CTC_NEG) : CTC_NEG;
                            }
                    }
#pragma unroll
                    for (int k = 0; k < KM; ++k)
                        if (NK != 0 || k < nk) {
                            Hw[(size_t)t * SP + lane + 64 * k] = bsum[k];
                            bt[k] = bsum[k] > -1e29 ? bsum[k] + (double)em[u][k] : CTC_NEG;
                        }
                }
            }
            if (tb - 8 >= 0) settle(tb - 8, -1, emn, em);
        }
      }
    };
    if (tid < 128) {
        const int nkb = (S + 63) >> 6;
        if (NS >= 2 && nkb == 1) recursions(std::integral_constant<int, 1>{});
        else if (NS >= 3 && nkb == 2) recursions(std::integral_constant<int, 2>{});
        else recursions(std::integral_constant<int, 0>{});
    }
    __threadfence_block();
    __syncthreads();
    // ---- phase 3: state posteriors -> class posteriors by LDS scatter-add (one frame per wave at a time: O(T*S) work;
    // a (frame, class) thread looping over the states was O(T*C*S) and took longer than both recursions), then the gradient
    if (dlogits) {
        // four frames of a wave in flight together: the loop used to wait out one L2 round trip per frame (96 dependent trips per wave)
        constexpr int U = 4;
        __shared__ float cls[4][U][64];
        const double logp = s_logp;
        float* dl = dlogits + (size_t)b * Tn * C;
        const int wv = tid >> 6;
        for (int t0 = wv; t0 < Tn; t0 += 4 * U) {
            double al[U][NS], bs[U][NS];
            float lgv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(t0 + 4 * u, Tn - 1);
                cls[wv][u][lane] = 0.f;
                const double* ga = Gw + (size_t)t * SP;
                const double* gb = Hw + (size_t)t * SP;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int s = lane + 64 * k;              // strided: coalesced 512-byte reads of the lattice rows
                    al[u][k] = s < S ? ga[s] : CTC_NEG; bs[u][k] = s < S ? gb[s] : CTC_NEG;
                }
                lgv[u] = lane < C ? lg[(size_t)t * C + lane] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + 4 * u;
                if (t >= Tn) continue;                        // wave-uniform
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int s = lane + 64 * k;
                    if (s < S && al[u][k] > -1e29 && bs[u][k] > -1e29 && logp > -1e29) atomicAdd(&cls[wv][u][ext[s]], __expf((float)(al[u][k] + bs[u][k] - logp)));
                }
                float gv = 0.f;
                if (lane < C) {
                    const float sm = expf(lgv[u] - lse[t]);
                    gv = grad_scale * (sm - cls[wv][u][lane]);
                    dl[(size_t)t * C + lane] = gv;
                }
                if (dlb) {      // bf16 copy of the row, zero padded to 128 classes (MFMA operand of the classifier's dgrad / wgrad)
                    const float lo = __shfl(gv, (2 * lane) & 63, 64), hi = __shfl(gv, (2 * lane + 1) & 63, 64);
                    typedef __attribute__((ext_vector_type(2))) __bf16 ctc_bf2;
                    ctc_bf2 pk; pk[0] = (__bf16)(lane < 32 ? lo : 0.f); pk[1] = (__bf16)(lane < 32 ? hi : 0.f);
                    dlb[((size_t)b * Tn + t) * 64 + lane] = __builtin_bit_cast(uint32_t, pk);
                }
            }
        }
    }
}

int launch_ctc(const float* logits, const int64_t* labels, int B, int T, int C, int L, int blank,
               float* nll, float* dlogits, float grad_scale, float* ws, hipStream_t s, void* dlb) {
    if (2 * L + 1 > 512 || C > 64) { ishara_set_error("ctc: L=%d (max 255) or C=%d (max 64) unsupported", L, C); return -1; }
    const int ns = ctc_ns(L);
    const size_t shmem = (size_t)T * sizeof(float) + (size_t)64 * ns * sizeof(int);
#define CTC_L(NS) hipLaunchKernelGGL(ctc_kernel<NS>, dim3(B), dim3(256), shmem, s, logits, labels, T, C, L, blank, nll, dlogits, grad_scale, reinterpret_cast<double*>(ws), reinterpret_cast<uint32_t*>(dlb))
    switch (ns) { case 1: CTC_L(1); break; case 2: CTC_L(2); break; case 3: CTC_L(3); break; case 4: CTC_L(4); break;
                  case 5: CTC_L(5); break; case 6: CTC_L(6); break; case 7: CTC_L(7); break; default: CTC_L(8); break; }
#undef CTC_L
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
    extern __shared__ int shi[];   // am[Tn]
    int* am = shi;
    __shared__ int s_wcnt[4], s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* lg = logits + (size_t)b * Tn * C;
    for (int t = tid; t < Tn; t += blockDim.x) {
        float best = lg[(size_t)t * C];
        int bi = 0;
        for (int c0 = 0; c0 < C; c0 += 16) {       // 16 loads in flight, then the (ordered: first max wins) comparisons
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = lg[(size_t)t * C + min(c0 + u, C - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (c0 + u < C && v[u] > best) { best = v[u]; bi = c0 + u; }
        }
        am[t] = bi;
    }
    if (tid == 0) s_base = 0;
    __syncthreads();
    // kept frames go out in order: positions by a ballot prefix, 256 frames per round (a serial scan by one thread was 18 us of the
    // kernel's 25 at T = 384)
    for (int t0 = 0; t0 < Tn; t0 += 256) {
        const int t = t0 + tid;
        const int keep = (t + 1 < Tn && am[t] != am[t + 1] && am[t] != blank) ? 1 : 0;
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[wid] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wid; ++w) off += s_wcnt[w];
        if (keep) out_idx[(size_t)b * Tn + off + before] = am[t];
        __syncthreads();
        if (tid == 0) s_base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int n = s_base;
    if (tid == 0) out_len[b] = n;
    for (int t = n + tid; t < Tn; t += blockDim.x) out_idx[(size_t)b * Tn + t] = -1;      // the -1 padding (was a fill launch of its own)
}

int launch_greedy_decode(const float* logits, int B, int T, int C, int blank, int* out_idx, int* out_len, hipStream_t s) {
    hipLaunchKernelGGL(greedy_decode_kernel, dim3(B), dim3(256), T * sizeof(int), s, logits, T, C, blank, out_idx, out_len);
    return LAUNCH_OK();
}
