// Multi-head self-attention for the Ishara encoder (conv-hybrid-model.ipynb c5:91-118):
// softmax(q.k^T * scale) with inverted dropout on the probabilities, then .v
// Layouts: q,k [B,H,T,dh]; vt [B,H,dh,T] (V transposed by the QKV GEMM epilogue);
// o, dout [B*T, H*dh]; dqkv [B*T, 3*H*dh] packed like the qkv projection output.
//
// impl 0 ("lane-split"): exact-fp32 VALU kernels, one query (or key) per group of 4
// lanes, each lane owning dh/4 of the head dimension; scores never touch HBM
// (online softmax forward, recompute-from-LSE backward).  Works for f32 and bf16 I/O, head dims 8, 16, 24, 32, 48, 64
// (24 and 48 — e.g. the reference's d384 / 8-head sibling model — have no MFMA kernel and run here in bf16 mode too).
#include "kernels.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -2)

DEVI float quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v;
}

#define ATT_KC 32   // keys (or queries) staged per LDS chunk

// stage rows [r0, r0+ATT_KC) of a [T, DH] row-major matrix into LDS as fp32 (zero beyond T)
template <typename T, int DH>
DEVI void stage_rows(const T* __restrict__ src, int ld, int r0, int Tn, float* dst, int tid, int nthreads) {
    for (int i = tid; i < ATT_KC * DH; i += nthreads) {
        const int r = i / DH, c = i - r * DH;
        dst[i] = (r0 + r < Tn) ? to_f(src[(size_t)(r0 + r) * ld + c]) : 0.f;
    }
}
// stage columns [r0, r0+ATT_KC) of vt [DH, T] into LDS as [ATT_KC][DH]
template <typename T, int DH>
DEVI void stage_vt(const T* __restrict__ vt, int r0, int Tn, float* dst, int tid, int nthreads) {
    for (int i = tid; i < ATT_KC * DH; i += nthreads) {
        const int c = i / ATT_KC, r = i - c * ATT_KC;      // consecutive threads -> consecutive keys (coalesced)
        dst[r * DH + c] = (r0 + r < Tn) ? to_f(vt[(size_t)c * Tn + r0 + r]) : 0.f;
    }
}

template <typename T, int DHL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
                                                       T* __restrict__ o, float* __restrict__ lse,
                                                       int B, int H, int Tn, float scale, DropSpec drop) {
    constexpr int DH = DHL * 4;
    __shared__ float Ks[ATT_KC * DH];
    __shared__ float Vs[ATT_KC * DH];
    const int tid = threadIdx.x, sub = tid & 3, ql = tid >> 2;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int t = blockIdx.x * 64 + ql;
    const bool qact = t < Tn;
    const T* qb = q + (size_t)bh * Tn * DH;
    const T* kb = k + (size_t)bh * Tn * DH;
    const T* vb = vt + (size_t)bh * DH * Tn;
    float qr[DHL], acc[DHL];
#pragma unroll
    for (int i = 0; i < DHL; ++i) { qr[i] = qact ? to_f(qb[(size_t)t * DH + sub * DHL + i]) * scale : 0.f; acc[i] = 0.f; }
    float m = -1e30f, l = 0.f;
    const uint32_t rk = rng_row_key(drop.key, (uint32_t)(bh * Tn + t));
    for (int k0 = 0; k0 < Tn; k0 += ATT_KC) {
        __syncthreads();
        stage_rows<T, DH>(kb, DH, k0, Tn, Ks, tid, 256);
        stage_vt<T, DH>(vb, k0, Tn, Vs, tid, 256);
        __syncthreads();
        const int nk = min(ATT_KC, Tn - k0);
#pragma unroll 1
        for (int g0 = 0; g0 < nk; g0 += 8) {
            float s[8];
            float gmax = -1e30f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < DHL; ++i) part += qr[i] * Ks[(g0 + j) * DH + sub * DHL + i];
                s[j] = quad_sum(part);
                if (g0 + j >= nk) s[j] = -1e30f;
                gmax = fmaxf(gmax, s[j]);
            }
            const float mn = fmaxf(m, gmax);
            const float corr = __expf(m - mn);
            l *= corr;
#pragma unroll
            for (int i = 0; i < DHL; ++i) acc[i] *= corr;
            m = mn;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = (g0 + j < nk) ? __expf(s[j] - mn) : 0.f;
                l += p;
                float pd = p;
                if (drop.thr) pd = rng_keep_q(rk, (uint32_t)(k0 + g0 + j), drop.thr) ? p * drop.scale : 0.f;
#pragma unroll
                for (int i = 0; i < DHL; ++i) acc[i] += pd * Vs[(g0 + j) * DH + sub * DHL + i];
            }
        }
    }
    if (qact) {
        const float inv = 1.f / l;
        T* op = o + ((size_t)b * Tn + t) * (H * DH) + h * DH + sub * DHL;
#pragma unroll
        for (int i = 0; i < DHL; ++i) op[i] = from_f<T>(acc[i] * inv);
        if (sub == 0) lse[(size_t)bh * Tn + t] = m + __logf(l);
    }
}

// dq (+ delta).  One query per 4 lanes.
template <typename T, int DHL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
                                                          const T* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, T* __restrict__ dqkv,
                                                          int B, int H, int Tn, float scale, DropSpec drop, int head_major) {
    constexpr int DH = DHL * 4;
    __shared__ float Ks[ATT_KC * DH];
    __shared__ float Vs[ATT_KC * DH];
    const int tid = threadIdx.x, sub = tid & 3, ql = tid >> 2;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int t = blockIdx.x * 64 + ql;
    const bool qact = t < Tn;
    const int d = H * DH;
    const T* qb = q + (size_t)bh * Tn * DH;
    const T* kb = k + (size_t)bh * Tn * DH;
    const T* vb = vt + (size_t)bh * DH * Tn;
    float qr[DHL], dor[DHL], dq[DHL];
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < DHL; ++i) {
        const size_t oo = ((size_t)b * Tn + (qact ? t : 0)) * d + h * DH + sub * DHL + i;
        qr[i] = qact ? to_f(qb[(size_t)t * DH + sub * DHL + i]) : 0.f;
        dor[i] = qact ? to_f(dout[oo]) : 0.f;
        dl += qact ? dor[i] * to_f(o[oo]) : 0.f;
        dq[i] = 0.f;
    }
    dl = quad_sum(dl);
    const float ls = qact ? lse[(size_t)bh * Tn + t] : 0.f;
    if (qact && sub == 0) delta[(size_t)bh * Tn + t] = dl;
    const uint32_t rk = rng_row_key(drop.key, (uint32_t)(bh * Tn + t));
    for (int k0 = 0; k0 < Tn; k0 += ATT_KC) {
        __syncthreads();
        stage_rows<T, DH>(kb, DH, k0, Tn, Ks, tid, 256);
        stage_vt<T, DH>(vb, k0, Tn, Vs, tid, 256);
        __syncthreads();
        const int nk = min(ATT_KC, Tn - k0);
        for (int j = 0; j < nk; ++j) {
            float ps = 0.f, pv = 0.f;
#pragma unroll
            for (int i = 0; i < DHL; ++i) { ps += qr[i] * Ks[j * DH + sub * DHL + i]; pv += dor[i] * Vs[j * DH + sub * DHL + i]; }
            const float s = quad_sum(ps) * scale;
            float dp = quad_sum(pv);
            const float p = __expf(s - ls);
            if (drop.thr) dp = rng_keep_q(rk, (uint32_t)(k0 + j), drop.thr) ? dp * drop.scale : 0.f;
            const float ds = p * (dp - dl) * scale;
#pragma unroll
            for (int i = 0; i < DHL; ++i) dq[i] += ds * Ks[j * DH + sub * DHL + i];
        }
    }
    if (qact) {
        const int col = head_major ? (h * 3 * DH + sub * DHL) : (h * DH + sub * DHL);
        T* dst = dqkv + ((size_t)b * Tn + t) * (3 * d) + col;
#pragma unroll
        for (int i = 0; i < DHL; ++i) dst[i] = from_f<T>(dq[i]);
    }
}

// dk, dv.  One key per 4 lanes; queries staged through LDS.
template <typename T, int DHL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt,
                                                           const T* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, T* __restrict__ dqkv,
                                                           int B, int H, int Tn, float scale, DropSpec drop, int head_major) {
    constexpr int DH = DHL * 4;
    __shared__ float Qs[ATT_KC * DH];
    __shared__ float Ds[ATT_KC * DH];
    __shared__ float Ls[ATT_KC], Dl[ATT_KC];
    const int tid = threadIdx.x, sub = tid & 3, kl = tid >> 2;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int key = blockIdx.x * 64 + kl;
    const bool kact = key < Tn;
    const int d = H * DH;
    const T* qb = q + (size_t)bh * Tn * DH;
    const T* kb = k + (size_t)bh * Tn * DH;
    const T* vb = vt + (size_t)bh * DH * Tn;
    float kr[DHL], vr[DHL], dk[DHL], dv[DHL];
#pragma unroll
    for (int i = 0; i < DHL; ++i) {
        kr[i] = kact ? to_f(kb[(size_t)key * DH + sub * DHL + i]) : 0.f;
        vr[i] = kact ? to_f(vb[(size_t)(sub * DHL + i) * Tn + key]) : 0.f;
        dk[i] = 0.f; dv[i] = 0.f;
    }
    for (int q0 = 0; q0 < Tn; q0 += ATT_KC) {
        __syncthreads();
        stage_rows<T, DH>(qb, DH, q0, Tn, Qs, tid, 256);
        stage_rows<T, DH>(dout + (size_t)b * Tn * d + h * DH, d, q0, Tn, Ds, tid, 256);
        if (tid < ATT_KC) {
            Ls[tid] = (q0 + tid < Tn) ? lse[(size_t)bh * Tn + q0 + tid] : 0.f;
            Dl[tid] = (q0 + tid < Tn) ? delta[(size_t)bh * Tn + q0 + tid] : 0.f;
        }
        __syncthreads();
        const int nq = min(ATT_KC, Tn - q0);
        for (int j = 0; j < nq; ++j) {
            float ps = 0.f, pv = 0.f;
#pragma unroll
            for (int i = 0; i < DHL; ++i) { ps += kr[i] * Qs[j * DH + sub * DHL + i]; pv += vr[i] * Ds[j * DH + sub * DHL + i]; }
            const float s = quad_sum(ps) * scale;
            float dp = quad_sum(pv);
            const float p = __expf(s - Ls[j]);
            float pd = p;
            if (drop.thr) {
                const bool keep = rng_keep_q(rng_row_key(drop.key, (uint32_t)(bh * Tn + q0 + j)), (uint32_t)key, drop.thr);
                pd = keep ? p * drop.scale : 0.f;
                dp = keep ? dp * drop.scale : 0.f;
            }
            const float ds = p * (dp - Dl[j]) * scale;
#pragma unroll
            for (int i = 0; i < DHL; ++i) { dv[i] += pd * Ds[j * DH + sub * DHL + i]; dk[i] += ds * Qs[j * DH + sub * DHL + i]; }
        }
    }
    if (kact) {
        const int ck = head_major ? (h * 3 * DH + DH + sub * DHL) : (d + h * DH + sub * DHL);
        const int cv = head_major ? (h * 3 * DH + 2 * DH + sub * DHL) : (2 * d + h * DH + sub * DHL);
        T* row = dqkv + ((size_t)b * Tn + key) * (3 * d);
#pragma unroll
        for (int i = 0; i < DHL; ++i) { row[ck + i] = from_f<T>(dk[i]); row[cv + i] = from_f<T>(dv[i]); }
    }
}

#define ATT_DISPATCH(KERNEL, TT, ...)                                                                    \
    do {                                                                                                  \
        dim3 grid((T + 63) / 64, B * H);                                                                  \
        switch (dh) {                                                                                     \
            case 8: hipLaunchKernelGGL((KERNEL<TT, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;        \
            case 16: hipLaunchKernelGGL((KERNEL<TT, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;       \
            case 24: hipLaunchKernelGGL((KERNEL<TT, 6>), grid, dim3(256), 0, s, __VA_ARGS__); break;       \
            case 32: hipLaunchKernelGGL((KERNEL<TT, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;       \
            case 48: hipLaunchKernelGGL((KERNEL<TT, 12>), grid, dim3(256), 0, s, __VA_ARGS__); break;      \
            case 64: hipLaunchKernelGGL((KERNEL<TT, 16>), grid, dim3(256), 0, s, __VA_ARGS__); break;      \
            default: ishara_set_error("attention: head dim %d unsupported (8,16,24,32,48,64)", dh); return -1;  \
        }                                                                                                 \
    } while (0)

int launch_attn_bwd_mfma(const void* q, const void* k, const void* vt, const void* o, const void* dout, const float* lse,
                         float* delta, void* dqkv, int B, int H, int T, int dh, float scale, DropSpec drop, uint32_t* maskbits, hipStream_t s);
int launch_attn_fwd_mfma(const void* q, const void* k, const void* vt, void* o, float* lse,
                         int B, int H, int T, int dh, float scale, DropSpec drop, uint32_t* maskbits, hipStream_t s);

int launch_attn_fwd_mfma_f16(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int H, int T, int dh, float scale, hipStream_t s);

int launch_attn_fwd(int dt, const void* q, const void* k, const void* vt, void* o, float* lse,
                    int B, int H, int T, int dh, float scale, DropSpec drop, int impl, uint32_t* maskbits, hipStream_t s) {
    if (impl == 1 && dt == DT_BF16 && (dh == 32 || dh == 64)) return launch_attn_fwd_mfma(q, k, vt, o, lse, B, H, T, dh, scale, drop, maskbits, s);
    if (impl == 1 && dt == DT_F16 && (dh == 32 || dh == 64) && drop.thr == 0 && T % 8 == 0) return launch_attn_fwd_mfma_f16(q, k, vt, o, lse, B, H, T, dh, scale, s);
    if (dt == DT_BF16) { ATT_DISPATCH(attn_fwd_kernel, bf16, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (bf16*)o, lse, B, H, T, scale, drop); }
    else if (dt == DT_F16) { ATT_DISPATCH(attn_fwd_kernel, f16, (const f16*)q, (const f16*)k, (const f16*)vt, (f16*)o, lse, B, H, T, scale, drop); }
    else { ATT_DISPATCH(attn_fwd_kernel, float, (const float*)q, (const float*)k, (const float*)vt, (float*)o, lse, B, H, T, scale, drop); }
    return LAUNCH_OK();
}

int launch_attn_bwd(int dt, const void* q, const void* k, const void* vt, const void* o, const void* dout,
                    const float* lse, float* delta, void* dqkv, int B, int H, int T, int dh, float scale,
                    DropSpec drop, int head_major, int impl, uint32_t* maskbits, hipStream_t s) {
    if (impl == 1 && dt == DT_BF16 && (dh == 32 || dh == 64) && head_major)
        return launch_attn_bwd_mfma(q, k, vt, o, dout, lse, delta, dqkv, B, H, T, dh, scale, drop, maskbits, s);
    if (dt == DT_BF16) {
        ATT_DISPATCH(attn_bwd_dq_kernel, bf16, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (const bf16*)o, (const bf16*)dout, lse, delta, (bf16*)dqkv, B, H, T, scale, drop, head_major);
        ATT_DISPATCH(attn_bwd_dkv_kernel, bf16, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (const bf16*)dout, lse, (const float*)delta, (bf16*)dqkv, B, H, T, scale, drop, head_major);
    } else {
        ATT_DISPATCH(attn_bwd_dq_kernel, float, (const float*)q, (const float*)k, (const float*)vt, (const float*)o, (const float*)dout, lse, delta, (float*)dqkv, B, H, T, scale, drop, head_major);
        ATT_DISPATCH(attn_bwd_dkv_kernel, float, (const float*)q, (const float*)k, (const float*)vt, (const float*)dout, lse, (const float*)delta, (float*)dqkv, B, H, T, scale, drop, head_major);
    }
    return LAUNCH_OK();
}
