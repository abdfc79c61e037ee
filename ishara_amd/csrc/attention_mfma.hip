// MFMA flash attention (bf16) for the Ishara encoder — impl 1 of attention.hip's interface.
//   q,k [B,H,T,dh]; vt [B,H,dh,T]; o/dout [B*T, H*dh]; lse [B,H,T]; dh in {32, 64}, T % 8 == 0.
//
// Forward.  A workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries
// (two 16-wide MFMA column tiles).  Keys are swept in chunks of 64 staged through a
// double-buffered LDS image (K rows as stored, V already transposed by the QKV GEMM epilogue).
// The scores are computed TRANSPOSED, S^T = K.Q^T (v_mfma_f32_16x16x32_bf16: A = K tile, B = Q^T
// fragment kept in registers), so a lane holds 4 consecutive keys of ONE query: the online
// softmax needs two cross-lane shuffles per chunk, and the exponentiated tile is already the B
// operand of the second product O^T += V^T.P^T (k-slot order {4g..4g+3} U {16+4g..16+4g+3},
// matched on the V^T fragment loads) — P never touches LDS or HBM.  Dropout on the probabilities
// uses the same counter hash as every other kernel (row = (b*H+h)*T+q, col = key).
#include "kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define AF_KC 64     // keys per LDS chunk
#define AF_QB 128    // queries per workgroup

DEVI uint32_t pk2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 t; t[0] = (bf16)lo; t[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, t);
}

template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ vt,
                                                            bf16* __restrict__ o, float* __restrict__ lse, int H, int Tn, float scale, DropSpec drop) {
    constexpr int KS = DH / 32;      // MFMA k-steps over the head dimension
    constexpr int DT = DH / 16;      // 16-wide output (dv) tiles
    constexpr int NP = DH / 32;      // 16-byte pieces per thread per staged operand (64*DH*2 B / 4 KB)
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][AF_KC * DH];
    __shared__ __attribute__((aligned(16))) bf16 Vs[2][DH * AF_KC];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int qbase = blockIdx.x * AF_QB + wid * 32;
    const bf16* qb = q + (size_t)bh * Tn * DH;
    const bf16* kb = k + (size_t)bh * Tn * DH;
    const bf16* vb = vt + (size_t)bh * DH * Tn;

    bf16x8 qf[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qrow = min(qbase + 16 * t + c, Tn - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[t][s] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qrow * DH + 32 * s + 8 * g);
    }
    f32x4 acc_o[DT][2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc_o[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
    const float cs = scale * 1.4426950408889634f;       // exp(x*scale) = exp2(x*cs)
    const int nch = (Tn + AF_KC - 1) / AF_KC;

    u32x4 rk[NP], rv[NP];
    auto gload = [&](int ch) {
        const int key0 = ch * AF_KC;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int pi = tid + 256 * u;
            {   // K: [64 keys][DH], DH/8 pieces per key
                const int key = pi / (DH / 8), part = pi % (DH / 8);
                rk[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)min(key0 + key, Tn - 1) * DH + part * 8);
            }
            {   // V^T: [DH rows][64 keys], 8 pieces per row; pieces beyond T are zero (T % 8 == 0)
                const int dv = pi >> 3, part = pi & 7;
                const int key = key0 + part * 8;
                rv[u] = key < Tn ? *reinterpret_cast<const u32x4*>(vb + (size_t)dv * Tn + key) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int pi = tid + 256 * u;
            *reinterpret_cast<u32x4*>(&Ks[buf][pi * 8]) = rk[u];      // same linear order as the global chunk
            *reinterpret_cast<u32x4*>(&Vs[buf][pi * 8]) = rv[u];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) gload(ch + 1);
        const bf16* Kc = Ks[ch & 1];
        const bf16* Vc = Vs[ch & 1];
        const int key0 = ch * AF_KC;
        // ---- S^T tiles: 4 key tiles x 2 query tiles
        f32x4 sacc[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bf16x8 kf[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(Kc + (16 * kt + c) * DH + 32 * s + 8 * g);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sacc[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) sacc[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[t][s], sacc[kt][t], 0, 0, 0);
            }
        }
        // ---- online softmax per query tile; lane = (query c, keys 16kt + 4g + r)
        bf16x8 pb[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (key0 + 16 * kt + 4 * g + r >= Tn) sacc[kt][t][r] = -1e30f;
                    mx = fmaxf(mx, sacc[kt][t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m_run[t], mx);
            const float corr = exp2f((m_run[t] - mn) * cs);
            m_run[t] = mn;
            l_run[t] *= corr;
#pragma unroll
            for (int d = 0; d < DT; ++d) acc_o[d][t] *= corr;
            const uint32_t rkey = rng_row_key(drop.key, (uint32_t)(bh * Tn + qbase + 16 * t + c));
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = exp2f((sacc[kt][t][r] - mn) * cs);
                    l_run[t] += pv;
                    float pd = pv;
                    if (drop.thr) pd = rng_keep(rkey, (uint32_t)(key0 + 16 * kt + 4 * g + r), drop.thr) ? pv * drop.scale : 0.f;
                    p[kt][r] = pd;
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w;
                w.x = pk2(p[2 * ks][0], p[2 * ks][1]);         w.y = pk2(p[2 * ks][2], p[2 * ks][3]);
                w.z = pk2(p[2 * ks + 1][0], p[2 * ks + 1][1]); w.w = pk2(p[2 * ks + 1][2], p[2 * ks + 1][3]);
                pb[t][ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const bf16* vrow = Vc + (16 * d + c) * AF_KC + 32 * ks + 4 * g;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 16);
                const bf16x8 vf = __builtin_bit_cast(bf16x8, (u32x4){lo.x, lo.y, hi.x, hi.y});
#pragma unroll
                for (int t = 0; t < 2; ++t) acc_o[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[t][ks], acc_o[d][t], 0, 0, 0);
            }
        if (more) lstore((ch + 1) & 1);
        __syncthreads();
    }
    // ---- normalise and store: lane = (query c, dv 16d + 4g + r)
    const int dmodel = H * DH;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qrow = qbase + 16 * t + c;
        if (qrow < Tn) {
            const float inv = 1.f / l;
            bf16* orow = o + ((size_t)b * Tn + qrow) * dmodel + h * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (bf16)(acc_o[d][t][r] * inv);
                *reinterpret_cast<bf16x4*>(orow + 16 * d + 4 * g) = w;
            }
            if (g == 0) lse[(size_t)bh * Tn + qrow] = m_run[t] * scale + __logf(l);
        }
    }
}

int launch_attn_fwd_mfma(const void* q, const void* k, const void* vt, void* o, float* lse,
                         int B, int H, int T, int dh, float scale, DropSpec drop, hipStream_t s) {
    if (T % 8 != 0) { ishara_set_error("attn_fwd_mfma: T %% 8 != 0"); return -1; }
    dim3 grid((T + AF_QB - 1) / AF_QB, B * H);
    if (dh == 32) hipLaunchKernelGGL(attn_fwd_mfma_kernel<32>, grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (bf16*)o, lse, H, T, scale, drop);
    else if (dh == 64) hipLaunchKernelGGL(attn_fwd_mfma_kernel<64>, grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (bf16*)o, lse, H, T, scale, drop);
    else { ishara_set_error("attn_fwd_mfma: head dim %d unsupported (32, 64)", dh); return -1; }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
