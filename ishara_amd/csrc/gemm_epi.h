// Fused GEMM epilogue on 8-wide row chunks (bias, PE table, saved pre-activation, activation, dropout, drop-path row scale, act', residual,
// QKV head split) — shared by the tile kernels of gemm.hip and gemm_big.hip.
#pragma once
#include "kernels.h"

// ---------------------------------------------------------------------------------
// epilogue on one 8-wide row chunk
// ---------------------------------------------------------------------------------
template <typename TC>
DEVI void epilogue_chunk(float (&v)[8], int m, int n, int nv, int N, const EpiArgs& ea, TC* __restrict__ C) {
    if (ea.bias) {
        const float bs = (ea.rowscale && ea.rowscale_bias) ? ea.rowscale[m / ea.T] : 1.f;      // row scale folded into the A operand: bias only
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nv) v[e] += ea.bias[n + e] * bs;
    }
    if (ea.addtab) {
        const float* t = ea.addtab + (size_t)(m % ea.tab_period) * N + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < nv) v[e] += t[e];
    }
    const size_t off = (size_t)m * N + n;
    if (ea.pre_out) store8_n(reinterpret_cast<TC*>(ea.pre_out) + off, v, nv);
    if (ea.act == ACT_SWISH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
    } else if (ea.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (ea.drop.thr) {
        const uint32_t rk = rng_row_key(ea.drop.key, (uint32_t)m);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {       // n is a multiple of 8: one hash per column pair
            const uint32_t h = rng_pair(rk, (uint32_t)(n + e));
            v[e] = (h & 0xffffu) >= ea.drop.thr ? v[e] * ea.drop.scale : 0.f;
            v[e + 1] = (h >> 16) >= ea.drop.thr ? v[e + 1] * ea.drop.scale : 0.f;
        }
    }
    if (ea.rowscale && !ea.rowscale_bias) {
        const float s = ea.rowscale[m / ea.T];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= s;
    }
    if (ea.dact != DACT_NONE) {
        float a[8];
        load8_n(reinterpret_cast<const TC*>(ea.aux) + off, a, nv);
        if (ea.dact == DACT_SWISH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= dswishf_(a[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
        }
    }
    if (ea.resid) {
        float a[8];
        load8_n(reinterpret_cast<const TC*>(ea.resid) + off, a, nv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a[e];
    }
    if (ea.mode == EPI_STD) {
        store8_n(C + off, v, nv);
    } else {   // EPI_QKV: q,k parts only (v handled by the transposed sweep)
        const int d = ea.H * ea.dh;
        int h, part, i;
        if (ea.head_major) { h = n / (3 * ea.dh); const int w = n - h * 3 * ea.dh; part = w / ea.dh; i = w - part * ea.dh; }
        else { part = n / d; const int w = n - part * d; h = w / ea.dh; i = w - h * ea.dh; }
        if (part < 2) {
            const int b = m / ea.T, t = m - b * ea.T;
            TC* dst = reinterpret_cast<TC*>(part == 0 ? ea.q : ea.k) + ((size_t)(b * ea.H + h) * ea.T + t) * ea.dh + i;
            store8_n(dst, v, nv);
        }
    }
}

// ---------------------------------------------------------------------------------
// Batched epilogue for one thread: a fixed 8-column chunk (n) of Q rows (m0r + q*mstep).
// prefetch() issues every global read the epilogue needs (residual, act' operand, drop-path
// scale, bias) as one batch BEFORE the accumulators are staged through LDS, so their latency
// overlaps the staging instead of forming Q serial load->wait->store chains; finish() then
// does the math and the 16-byte stores.  Partial chunks (N % 8) and the PE-table add take the
// generic per-chunk path.
// ---------------------------------------------------------------------------------
template <typename TC, int Q>
struct EpiRows {
    float bias[8], res[Q][8], ax[Q][8], rs[Q];
    int m[Q];
    size_t off[Q];
    int n, nv;
    bool fast;

    DEVI void prefetch(int m0r, int mstep, int n_, int M, int N, const EpiArgs& ea) {
        n = n_;
        nv = min(8, N - n);
        fast = (nv == 8) && (ea.addtab == nullptr);
#pragma unroll
        for (int q = 0; q < Q; ++q) m[q] = m0r + q * mstep;
        if (!fast) return;
#pragma unroll
        for (int e = 0; e < 8; ++e) bias[e] = ea.bias ? ea.bias[n + e] : 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) off[q] = (size_t)min(m[q], M - 1) * N + n;
        if (ea.resid) {
#pragma unroll
            for (int q = 0; q < Q; ++q) load8(reinterpret_cast<const TC*>(ea.resid) + off[q], res[q]);
        }
        if (ea.dact != DACT_NONE) {
#pragma unroll
            for (int q = 0; q < Q; ++q) load8(reinterpret_cast<const TC*>(ea.aux) + off[q], ax[q]);
        }
        if (ea.rowscale) {
#pragma unroll
            for (int q = 0; q < Q; ++q) rs[q] = ea.rowscale[min(m[q], M - 1) / ea.T];
        }
    }

    // stage: pointer to this thread's chunk of row q=0; sstep = floats between successive q rows
    DEVI void finish(const float* stage, int sstep, int M, int N, const EpiArgs& ea, TC* __restrict__ C) {
        if (n >= N) return;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float v[8];
            const float4 x0 = *reinterpret_cast<const float4*>(stage + q * sstep);
            const float4 x1 = *reinterpret_cast<const float4*>(stage + q * sstep + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            if (!fast) {
                if (m[q] < M) epilogue_chunk<TC>(v, m[q], n, nv, N, ea, C);
                continue;
            }
            const bool ok = m[q] < M;
            const float bs = (ea.rowscale && ea.rowscale_bias) ? rs[q] : 1.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[e] * bs;
            if (ea.pre_out && ok) store8(reinterpret_cast<TC*>(ea.pre_out) + off[q], v);
            if (ea.act == ACT_SWISH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
            } else if (ea.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (ea.drop.thr) {
                const uint32_t rk = rng_row_key(ea.drop.key, (uint32_t)m[q]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {       // n is a multiple of 8: one hash per column pair
            const uint32_t h = rng_pair(rk, (uint32_t)(n + e));
            v[e] = (h & 0xffffu) >= ea.drop.thr ? v[e] * ea.drop.scale : 0.f;
            v[e + 1] = (h >> 16) >= ea.drop.thr ? v[e + 1] * ea.drop.scale : 0.f;
        }
            }
            if (ea.rowscale && !ea.rowscale_bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= rs[q];
            }
            if (ea.dact == DACT_SWISH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dswishf_(ax[q][e]);
            } else if (ea.dact == DACT_POS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ax[q][e] > 0.f ? v[e] : 0.f;
            }
            if (ea.resid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += res[q][e];
            }
            if (!ok) continue;
            if (ea.mode == EPI_STD) {
                store8(C + off[q], v);
            } else {
                const int d = ea.H * ea.dh;
                int h, part, i;
                if (ea.head_major) { h = n / (3 * ea.dh); const int w = n - h * 3 * ea.dh; part = w / ea.dh; i = w - part * ea.dh; }
                else { part = n / d; const int w = n - part * d; h = w / ea.dh; i = w - h * ea.dh; }
                if (part < 2) {
                    const int b = m[q] / ea.T, t = m[q] - b * ea.T;
                    store8(reinterpret_cast<TC*>(part == 0 ? ea.q : ea.k) + ((size_t)(b * ea.H + h) * ea.T + t) * ea.dh + i, v);
                }
            }
        }
    }
};

