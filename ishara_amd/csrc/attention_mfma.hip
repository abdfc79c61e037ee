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
#define AF_VLD 72    // V^T tile row stride (64 keys + 8 pad): 144-B rows make the 8-byte fragment reads conflict-free
#define AF_PAD 8     // row-major [64][DH] tiles get DH+8 columns (80-B / 144-B rows): conflict-free 16-byte fragment reads

DEVI uint32_t pk2(float lo, float hi) {      // ONE v_cvt_pk_bf16_f32 (element-wise casts compile to two of them and a v_perm_b32)
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}


// keepbits |= the four keep flags at bit positions sh .. sh+3 (sh is a compile-time constant after unrolling)
DEVI void constexpr_shift_or(uint32_t& bits, bool k0, bool k1, bool k2, bool k3, int sh) {
    bits |= (k0 ? (1u << sh) : 0u) | (k1 ? (2u << sh) : 0u) | (k2 ? (4u << sh) : 0u) | (k3 ? (8u << sh) : 0u);
}

// XCD-aware workgroup -> (head bh, block xb) mapping.  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with
// its own L2; the nxb query (key) blocks of one (batch, head) all stream the SAME K/V (Q/dO) rows, so they are given ids
// with the same residue mod 8 and consecutive positions on that XCD: the streamed operand is read from HBM once per head
// instead of once per block.  Grid = nxb * BH workgroups, 1-D.
DEVI void attn_block_of(int nxb, int BH, int& bh, int& xb) {
    const int id = blockIdx.x;
    if ((BH & 7) == 0) { const int xcd = id & 7, j = id >> 3; bh = (j / nxb) * 8 + xcd; xb = j % nxb; }
    else { bh = id / nxb; xb = id % nxb; }
}

// DM: dropout mode, compile-time so that no per-score uniform branch is left: 0 none, 1 counter hash, 2 counter hash in the
// forward + keep bits cached in `maskbits` for the two backward kernels
template <typename E> struct af_vec;
template <> struct af_vec<bf16> { typedef bf16x8 v8; typedef bf16x4 v4; };
template <> struct af_vec<f16> { typedef f16x8 v8; typedef __attribute__((ext_vector_type(4))) _Float16 v4; };
DEVI f32x4 af_mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
DEVI f32x4 af_mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
template <typename E> DEVI uint32_t pk2e(float lo, float hi);
template <> DEVI uint32_t pk2e<bf16>(float lo, float hi) { return pk2(lo, hi); }
template <> DEVI uint32_t pk2e<f16>(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    f16x2 t; t[0] = (f16)lo; t[1] = (f16)hi;
    return __builtin_bit_cast(uint32_t, t);
}

// E: element type of q, k, v^T and o — bf16 (training and inference) or f16 (the ISHARA_F16 inference path, dropout-free)
template <int DH, int DM, typename E = bf16>
__global__ __launch_bounds__(256, DH <= 32 ? 3 : 2) void attn_fwd_mfma_kernel(const E* __restrict__ q, const E* __restrict__ k, const E* __restrict__ vt,
                                                            E* __restrict__ o, float* __restrict__ lse, int H, int Tn, float scale, DropSpec drop, int BH, uint32_t* __restrict__ maskbits) {
    constexpr int KS = DH / 32;      // MFMA k-steps over the head dimension
    constexpr int DT = DH / 16;      // 16-wide output (dv) tiles
    constexpr int NP = DH / 32;      // 16-byte pieces per thread per staged operand (64*DH*2 B / 4 KB)
    constexpr int KLD = DH + AF_PAD;
    __shared__ __attribute__((aligned(16))) E Ks[2][AF_KC * KLD];
    __shared__ __attribute__((aligned(16))) E Vs[2][DH * AF_VLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    int bh, xb;
    const int nqb = (Tn + AF_QB - 1) / AF_QB;
    attn_block_of(nqb, BH, bh, xb);
    const int b = bh / H, h = bh - b * H;
    const int qbase = xb * AF_QB + wid * 32;
    const E* qb = q + (size_t)bh * Tn * DH;
    const E* kb = k + (size_t)bh * Tn * DH;
    const E* vb = vt + (size_t)bh * DH * Tn;

    typename af_vec<E>::v8 qf[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qrow = min(qbase + 16 * t + c, Tn - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[t][s] = *reinterpret_cast<const typename af_vec<E>::v8*>(qb + (size_t)qrow * DH + 32 * s + 8 * g);
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
            *reinterpret_cast<u32x4*>(&Ks[buf][(pi / (DH / 8)) * KLD + (pi % (DH / 8)) * 8]) = rk[u];
            *reinterpret_cast<u32x4*>(&Vs[buf][(pi >> 3) * AF_VLD + (pi & 7) * 8]) = rv[u];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) gload(ch + 1);
        const E* Kc = Ks[ch & 1];
        const E* Vc = Vs[ch & 1];
        const int key0 = ch * AF_KC;
        const bool partial = key0 + AF_KC > Tn;      // only the last chunk needs per-key bounds masks
        // ---- S^T tiles: 4 key tiles x 2 query tiles
        f32x4 sacc[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            typename af_vec<E>::v8 kf[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) kf[s] = *reinterpret_cast<const typename af_vec<E>::v8*>(Kc + (16 * kt + c) * KLD + 32 * s + 8 * g);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sacc[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) sacc[kt][t] = af_mfma(kf[s], qf[t][s], sacc[kt][t]);
            }
        }
        // ---- online softmax per query tile; lane = (query c, keys 16kt + 4g + r)
        typename af_vec<E>::v8 pb[2][2];
        uint32_t keepbits = 0u;          // bit 16t + 4kt + r: dropout keep flag of (query tile t, key 16kt + 4g + r)
        if (partial) {                   // a real (uniform) branch: as a per-score select this cost 32 v_cndmask + 15 v_cmp in EVERY chunk
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + 16 * kt + 4 * g + r >= Tn) sacc[kt][t][r] = -1e30f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kt][t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m_run[t], mx);
            const float corr = __builtin_amdgcn_exp2f((m_run[t] - mn) * cs);      // raw v_exp_f32: exp2f() adds a 6-instruction denormal-range wrapper
            const float mnc = -mn * cs;
            m_run[t] = mn;
            l_run[t] *= corr;
#pragma unroll
            for (int d = 0; d < DT; ++d) acc_o[d][t] *= corr;
            const uint32_t rkey = rng_row_key(drop.key, (uint32_t)(bh * Tn + qbase + 16 * t + c));
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kt][t][r], cs, mnc));
                    l_run[t] += pv;
                    p[kt][r] = pv;
                }
                if constexpr (DM != 0) {      // 4 consecutive keys: ONE hash, a byte per key (common.h rng_quad); the keep bits are remembered for the backward kernels
                    // one compare per score serves both the select and the bit (the 1/P(keep) factor is applied once per output below)
                    const uint32_t h0 = rng_quad(rkey, (uint32_t)(key0 + 16 * kt + 4 * g));
                    const bool k0 = (h0 & 0xffu) >= drop.thr, k1 = ((h0 >> 8) & 0xffu) >= drop.thr, k2 = ((h0 >> 16) & 0xffu) >= drop.thr, k3 = (h0 >> 24) >= drop.thr;
                    p[kt][0] = k0 ? p[kt][0] : 0.f; p[kt][1] = k1 ? p[kt][1] : 0.f; p[kt][2] = k2 ? p[kt][2] : 0.f; p[kt][3] = k3 ? p[kt][3] : 0.f;
                    if constexpr (DM == 2) {
                        constexpr_shift_or(keepbits, k0, k1, k2, k3, 16 * t + 4 * kt);
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 w;
                w.x = pk2e<E>(p[2 * ks][0], p[2 * ks][1]);         w.y = pk2e<E>(p[2 * ks][2], p[2 * ks][3]);
                w.z = pk2e<E>(p[2 * ks + 1][0], p[2 * ks + 1][1]); w.w = pk2e<E>(p[2 * ks + 1][2], p[2 * ks + 1][3]);
                pb[t][ks] = __builtin_bit_cast(typename af_vec<E>::v8, w);
            }
        }
        if constexpr (DM == 2) __builtin_nontemporal_store(keepbits, &maskbits[((size_t)(bh * nqb + xb) * nch + ch) * 256 + tid]);      // read again only by the backward pass
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const E* vrow = Vc + (16 * d + c) * AF_VLD + 32 * ks + 4 * g;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 16);
                const typename af_vec<E>::v8 vf = __builtin_bit_cast(typename af_vec<E>::v8, (u32x4){lo.x, lo.y, hi.x, hi.y});
#pragma unroll
                for (int t = 0; t < 2; ++t) acc_o[d][t] = af_mfma(vf, pb[t][ks], acc_o[d][t]);
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
            const float inv = (DM != 0 ? drop.scale : 1.f) / l;
            E* orow = o + ((size_t)b * Tn + qrow) * dmodel + h * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                typename af_vec<E>::v4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (E)(acc_o[d][t][r] * inv);
                *reinterpret_cast<typename af_vec<E>::v4*>(orow + 16 * d + 4 * g) = w;
            }
            if (g == 0) lse[(size_t)bh * Tn + qrow] = m_run[t] * scale + __logf(l);
        }
    }
}

// =====================================================================================
// Backward: two MFMA kernels, both recomputing P from Q, K and the forward's LSE.
//   dq kernel  (query-stationary, same sweep as the forward): S^T = K.Q^T, dP^T = V.dO^T,
//              dS^T = P^T o (dP^T o D - delta) * scale, dQ^T += K^T.dS^T          (+ writes delta)
//   dkv kernel (key-stationary, sweeps query chunks): S = Q.K^T, dP = dO.V^T,
//              dV^T += dO^T.(P o D), dK^T += Q^T.dS
// Operands that are needed transposed (V rows / K^T in the first, dO^T / Q^T in the second) are
// fetched from the row-major LDS chunk with ds_read_b64_tr_b16; score-shaped accumulators feed
// the next product as its B operand in registers (same k-slot permutation as the forward).
// No atomics: dq is complete in the first kernel, dk/dv in the second.
// =====================================================================================
typedef __attribute__((ext_vector_type(4))) short s16x4_;
typedef __attribute__((ext_vector_type(8))) short s16x8_;

// transposed fragment: 16 columns starting at col0 of rows {r0 + q', r1 + q'} (q' = 0..3) of a row-major bf16 tile
DEVI bf16x8 trfrag(const bf16* tile, int ld, int r0, int r1, int col0, int lane) {
    const int qq = (lane >> 2) & 3, pp = lane & 3;
    const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(tile + (r0 + qq) * ld + col0 + 4 * pp));
    const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(tile + (r1 + qq) * ld + col0 + 4 * pp));
    return __builtin_bit_cast(bf16x8, (s16x8_)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
DEVI bf16x8 pack8(const float (&a)[4], const float (&b)[4]) {
    u32x4 w;
    w.x = pk2(a[0], a[1]); w.y = pk2(a[2], a[3]); w.z = pk2(b[0], b[1]); w.w = pk2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, w);
}

template <int DH, int DM>
__global__ __launch_bounds__(256, DH <= 32 ? 3 : 2) void attn_bwd_dq_mfma_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ vt,
                                                               const bf16* __restrict__ o, const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                               float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                               int H, int Tn, float scale, DropSpec drop, int BH, uint32_t* __restrict__ maskbits) {
    constexpr int KS = DH / 32, DT = DH / 16, NP = DH / 32;
    constexpr int KLD = DH + AF_PAD;
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][AF_KC * KLD];
    __shared__ __attribute__((aligned(16))) bf16 Vs[2][DH * AF_VLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    int bh, xb;
    const int nqb = (Tn + AF_QB - 1) / AF_QB;
    attn_block_of(nqb, BH, bh, xb);
    const int b = bh / H, h = bh - b * H;
    const int qbase = xb * AF_QB + wid * 32;
    const int dmodel = H * DH;
    const bf16* qb = q + (size_t)bh * Tn * DH;
    const bf16* kb = k + (size_t)bh * Tn * DH;
    const bf16* vb = vt + (size_t)bh * DH * Tn;

    bf16x8 qf[2][KS], dof[2][KS];
    float dlt[2], lsl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qrow = min(qbase + 16 * t + c, Tn - 1);
        const size_t orow = ((size_t)b * Tn + qrow) * dmodel + h * DH;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[t][s] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qrow * DH + 32 * s + 8 * g);
            dof[t][s] = *reinterpret_cast<const bf16x8*>(dout + orow + 32 * s + 8 * g);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + orow + 32 * s + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)dof[t][s][e] * (float)of[e];
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        dlt[t] = part;
        lsl[t] = lse[(size_t)bh * Tn + qrow] * 1.4426950408889634f;
        if (g == 0 && qbase + 16 * t + c < Tn) delta[(size_t)bh * Tn + qrow] = part;
    }
    f32x4 acc[DT][2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float cs = scale * 1.4426950408889634f;
    const int nch = (Tn + AF_KC - 1) / AF_KC;

    u32x4 rk[NP], rv[NP];
#define DQ_GLOAD(ch)                                                                                              \
    {                                                                                                             \
        const int key0_ = (ch) * AF_KC;                                                                           \
        _Pragma("unroll") for (int u = 0; u < NP; ++u) {                                                          \
            const int pi = tid + 256 * u;                                                                         \
            const int key = pi / (DH / 8), part = pi % (DH / 8);                                                  \
            rk[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)min(key0_ + key, Tn - 1) * DH + part * 8);       \
            const int dv = pi >> 3, kk = key0_ + (pi & 7) * 8;                                                    \
            rv[u] = kk < Tn ? *reinterpret_cast<const u32x4*>(vb + (size_t)dv * Tn + kk) : u32x4{0u, 0u, 0u, 0u}; \
        }                                                                                                         \
    }
#define DQ_LSTORE(buf)                                                                                            \
    {                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < NP; ++u) {                                                          \
            const int pi = tid + 256 * u;                                                                         \
            *reinterpret_cast<u32x4*>(&Ks[buf][(pi / (DH / 8)) * KLD + (pi % (DH / 8)) * 8]) = rk[u];            \
            *reinterpret_cast<u32x4*>(&Vs[buf][(pi >> 3) * AF_VLD + (pi & 7) * 8]) = rv[u];                       \
        }                                                                                                         \
    }
    DQ_GLOAD(0);
    DQ_LSTORE(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) DQ_GLOAD(ch + 1);
        const bf16* Kc = Ks[ch & 1];
        const bf16* Vc = Vs[ch & 1];
        const int key0 = ch * AF_KC;
        const bool partial = key0 + AF_KC > Tn;      // only the last chunk needs per-key bounds masks
        const uint32_t keepbits = DM == 2 ? __builtin_nontemporal_load(&maskbits[((size_t)(bh * nqb + xb) * nch + ch) * 256 + tid]) : 0u;
        bf16x8 dsb[2][2];
        {
            f32x4 sacc[4][2], dpa[4][2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                bf16x8 kf[KS], vf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    kf[s] = *reinterpret_cast<const bf16x8*>(Kc + (16 * kt + c) * KLD + 32 * s + 8 * g);
                    vf[s] = trfrag(Vc, AF_VLD, 32 * s + 8 * g, 32 * s + 8 * g + 4, 16 * kt, lane);    // V[key c][dv 32s+8g..+7]
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    sacc[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dpa[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        sacc[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[t][s], sacc[kt][t], 0, 0, 0);
                        dpa[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[s], dof[t][s], dpa[kt][t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t rkey = rng_row_key(drop.key, (uint32_t)(bh * Tn + qbase + 16 * t + c));
                float ds[4][4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    float dp[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[r] = dpa[kt][t][r];
                    if constexpr (DM != 0) {
                        // keep bits of the forward pass when it stored them (same lane layout), else the hash again
                        const uint32_t kb4 = DM == 2 ? (keepbits >> (16 * t + 4 * kt)) & 15u : rng_bits4_q(rkey, (uint32_t)(key0 + 16 * kt + 4 * g), drop.thr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) dp[r] = ((kb4 >> r) & 1u) ? dp[r] * drop.scale : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + 16 * kt + 4 * g + r;
                        const bool inb = (!partial) | (key < Tn);                // branchless: exp2(-inf) = 0 for the keys past the end
                        const float xq = fmaf(sacc[kt][t][r], cs, -lsl[t]);
                        const float pv = __builtin_amdgcn_exp2f(inb ? xq : -INFINITY);
                        ds[kt][r] = pv * (dp[r] - dlt[t]);                        // * scale once per output (epilogue)
                    }
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) dsb[t][ks] = pack8(ds[2 * ks], ds[2 * ks + 1]);
            }
        }
        // dQ^T[dh][q] += K^T[dh][key] . dS^T[key][q]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const bf16x8 ktf = trfrag(Kc, KLD, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * d, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsb[t][ks], acc[d][t], 0, 0, 0);
            }
        if (more) DQ_LSTORE((ch + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qrow = qbase + 16 * t + c;
        if (qrow < Tn) {
            bf16* drow = dqkv + ((size_t)b * Tn + qrow) * (3 * dmodel) + h * 3 * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (bf16)(acc[d][t][r] * scale);
                *reinterpret_cast<bf16x4*>(drow + 16 * d + 4 * g) = w;
            }
        }
    }
}

template <int DH, int DM>
__global__ __launch_bounds__(256, DH <= 32 ? 3 : 2) void attn_bwd_dkv_mfma_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ vt,
                                                                const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                                int H, int Tn, float scale, DropSpec drop, int BH, uint32_t* __restrict__ maskbits) {
    constexpr int KS = DH / 32, DT = DH / 16, NP = DH / 32;
    constexpr int KLD = DH + AF_PAD;
    __shared__ __attribute__((aligned(16))) bf16 Qs[2][AF_KC * KLD];
    __shared__ __attribute__((aligned(16))) bf16 Ds[2][AF_KC * KLD];
    __shared__ float Ls[2][AF_KC], Dl[2][AF_KC];
    __shared__ uint32_t Rk[2][AF_KC];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    int bh, xb;
    const int nqb = (Tn + AF_QB - 1) / AF_QB;
    attn_block_of(nqb, BH, bh, xb);
    const int b = bh / H, h = bh - b * H;
    const int kbase = xb * AF_QB + wid * 32;
    const int dmodel = H * DH;
    const bf16* qb = q + (size_t)bh * Tn * DH;
    const bf16* kb = k + (size_t)bh * Tn * DH;
    const bf16* vb = vt + (size_t)bh * DH * Tn;
    const bf16* dob = dout + (size_t)b * Tn * dmodel + h * DH;

    bf16x8 kf[2][KS], vf[2][KS];          // B operands: K[key c][dh 32s+8g..], V[key c][dv 32s+8g..]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int key = min(kbase + 16 * t + c, Tn - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[t][s] = *reinterpret_cast<const bf16x8*>(kb + (size_t)key * DH + 32 * s + 8 * g);
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = vb[(size_t)(32 * s + 8 * g + e) * Tn + key];
            vf[t][s] = v;
        }
    }
    f32x4 adv[DT][2], adk[DT][2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) { adv[d][t] = f32x4{0.f, 0.f, 0.f, 0.f}; adk[d][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float cs = scale * 1.4426950408889634f;
    const int nch = (Tn + AF_KC - 1) / AF_KC;

    u32x4 rq[NP], rd[NP];
    float rl = 0.f;
    uint32_t rr = 0;
#define DKV_GLOAD(ch)                                                                                             \
    {                                                                                                             \
        const int q0_ = (ch) * AF_KC;                                                                             \
        _Pragma("unroll") for (int u = 0; u < NP; ++u) {                                                          \
            const int pi = tid + 256 * u;                                                                         \
            const int row = min(q0_ + pi / (DH / 8), Tn - 1), part = pi % (DH / 8);                               \
            rq[u] = *reinterpret_cast<const u32x4*>(qb + (size_t)row * DH + part * 8);                            \
            rd[u] = *reinterpret_cast<const u32x4*>(dob + (size_t)row * dmodel + part * 8);                       \
        }                                                                                                         \
        const int qr_ = min(q0_ + (tid & 63), Tn - 1);                                                            \
        if (tid < 64) rl = lse[(size_t)bh * Tn + qr_] * 1.4426950408889634f;                                      \
        else if (tid < 128) rl = delta[(size_t)bh * Tn + qr_];                                                    \
        else if (tid < 192) rr = rng_row_key(drop.key, (uint32_t)(bh * Tn + q0_ + (tid & 63)));                  \
    }
#define DKV_LSTORE(buf)                                                                                           \
    {                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < NP; ++u) {                                                          \
            const int pi = tid + 256 * u;                                                                         \
            *reinterpret_cast<u32x4*>(&Qs[buf][(pi / (DH / 8)) * KLD + (pi % (DH / 8)) * 8]) = rq[u];            \
            *reinterpret_cast<u32x4*>(&Ds[buf][(pi / (DH / 8)) * KLD + (pi % (DH / 8)) * 8]) = rd[u];            \
        }                                                                                                         \
        if (tid < 64) Ls[buf][tid] = rl;                                                                          \
        else if (tid < 128) Dl[buf][tid - 64] = rl;                                                               \
        else if (tid < 192) Rk[buf][tid - 128] = rr;                                                              \
    }
    DKV_GLOAD(0);
    DKV_LSTORE(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) DKV_GLOAD(ch + 1);
        const bf16* Qc = Qs[ch & 1];
        const bf16* Dc = Ds[ch & 1];
        const float* Lc = Ls[ch & 1];
        const float* Dlc = Dl[ch & 1];
        const uint32_t* Rc = Rk[ch & 1];
        const int q0 = ch * AF_KC;
        const bool partial = q0 + AF_KC > Tn;
        // 32 queries (ks) at a time: scores / dP for both key tiles, the elementwise pass, then straight into the dV / dK
        // products of those 32 queries -- keeps ~70 fewer registers live than doing all 64 queries at once (3 waves / SIMD)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 pdb[2], dsb[2];
            // keep bits stored by the forward kernel: for key tile t, the 4 queries r of BOTH query tiles hq sit in 4 consecutive
            // words (one 16-byte load): word r holds query 4g + r, bit 16hq + 4kt_f + r_f
            u32x4 mw[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
            if constexpr (DM == 2) {
                const int qq = q0 + 32 * ks;                              // first query of this 32-query half
                const size_t qbw = (size_t)(bh * nqb + qq / AF_QB) * nch;
                const int wave_f = (qq % AF_QB) >> 5;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int key = kbase + 16 * t;                       // key tile of this wave
                    mw[t] = *reinterpret_cast<const u32x4*>(maskbits + (qbw + key / AF_KC) * 256 + wave_f * 64 + (c >> 2) * 16 + 4 * g);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 sacc[2], dpa[2];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    const int qt = 2 * ks + hq;
                    sacc[hq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dpa[hq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const bf16x8 qfr = *reinterpret_cast<const bf16x8*>(Qc + (16 * qt + c) * KLD + 32 * s + 8 * g);
                        const bf16x8 dfr = *reinterpret_cast<const bf16x8*>(Dc + (16 * qt + c) * KLD + 32 * s + 8 * g);
                        sacc[hq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[t][s], sacc[hq], 0, 0, 0);
                        dpa[hq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[t][s], dpa[hq], 0, 0, 0);
                    }
                }
                // element (hq, r): query q0 + 32ks + 16hq + 4g + r, key kbase + 16t + c
                const uint32_t key = (uint32_t)(kbase + 16 * t + c);
                float pd[2][4], ds[2][4];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ql = 32 * ks + 16 * hq + 4 * g + r;
                        const bool inb = (!partial) | (q0 + ql < Tn);            // branchless: exp2(-inf) = 0 for the queries past the end
                        const float xq = fmaf(sacc[hq][r], cs, -Lc[ql]);          // unconditional LDS read + fma: no exec-mask branch per score
                        const float pv = __builtin_amdgcn_exp2f(inb ? xq : -INFINITY);
                        float dp = dpa[hq][r], pdv = pv;
                        if constexpr (DM != 0) {
                            const bool keep = DM == 2 ? ((mw[t][r] >> (16 * hq + 4 * (((kbase + 16 * t) % AF_KC) >> 4) + (c & 3))) & 1u) != 0u
                                                      : rng_keep_q(Rc[ql], key, drop.thr);
                            dp = keep ? dp * drop.scale : 0.f;
                            pdv = keep ? pv : 0.f;                     // * drop.scale once per dV output (epilogue)
                        }
                        pd[hq][r] = pdv;
                        ds[hq][r] = pv * (dp - Dlc[ql]);               // * scale once per dK output (epilogue)
                    }
                pdb[t] = pack8(pd[0], pd[1]);
                dsb[t] = pack8(ds[0], ds[1]);
            }
            // dV^T[dv][key] += dO^T[dv][q].(P o D)[q][key] ; dK^T[dh][key] += Q^T[dh][q].dS[q][key]
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const bf16x8 dtf = trfrag(Dc, KLD, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * d, lane);
                const bf16x8 qtf = trfrag(Qc, KLD, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * d, lane);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    adv[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtf, pdb[t], adv[d][t], 0, 0, 0);
                    adk[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsb[t], adk[d][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) DKV_LSTORE((ch + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int key = kbase + 16 * t + c;
        if (key < Tn) {
            bf16* drow = dqkv + ((size_t)b * Tn + key) * (3 * dmodel) + h * 3 * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                bf16x4 wk, wv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { wk[r] = (bf16)(adk[d][t][r] * scale); wv[r] = (bf16)(adv[d][t][r] * (DM != 0 ? drop.scale : 1.f)); }
                *reinterpret_cast<bf16x4*>(drow + DH + 16 * d + 4 * g) = wk;
                *reinterpret_cast<bf16x4*>(drow + 2 * DH + 16 * d + 4 * g) = wv;
            }
        }
    }
}

// =====================================================================================
// Backward in ONE pass (dh = 32, T <= 384): a workgroup of 8 waves owns one (batch, head) and sweeps the queries once.
// Wave w owns the key tiles {128 j + 16 w .. +15 : j < NT} for the whole kernel: their K / V rows (B operands of S = Q.K^T and
// dP = dO.V^T) and the dK^T / dV^T accumulators of those keys stay in registers, exactly as in the key-stationary kernel above.
// What that kernel cannot produce is dQ (a sum over keys, i.e. over waves).  Here every wave also keeps K^T of its keys as A
// fragments, writes its dS tile (bf16, [own key][32 queries]) to a private LDS tile, reads it back TRANSPOSED
// (ds_read_b64_tr_b16) as the B operand of dQ^T += K^T.dS^T over its own keys, and the 8 partial dQ^T tiles are summed through
// LDS in a fixed order.  S, dP and the whole elementwise pass (exp2, dropout, dS) are computed once instead of twice:
// 5 GEMM units instead of 7 and half the VALU work of the two-kernel backward (the attention backward is VALU bound).
// delta = rowsum(dO o O) is computed while the dO chunk is staged.
// =====================================================================================
#define FB_QLD 40                                   // dS / K tile row stride: 32 columns + 8 pad (80-byte rows)
// LDS: Q / dO chunks (double buffered) + row constants, NW private dS tiles of 32 * NP2 rows, two dQ exchange buffers of NW * 4 KB
constexpr int fb_smem_bytes(int NW, int NT) { return 2 * 2 * 64 * 40 * 2 + 3 * 2 * 64 * 4 + NW * 32 * ((NT + 1) / 2) * FB_QLD * 2 + 2 * NW * 4 * 64 * 16; }
// NW waves (8 or 12: two or three per SIMD), NT key tiles per wave: wave w owns the keys 16 NW j + 16 w .. +15, j < NT (T <= 16 NW NT).
// FULL: T == 16 NW NT (no ragged tile / chunk: the per-tile branches and bounds selects are compiled out)
template <int NW, int NT, int DM, bool FULL>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_bwd_fused_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ vt,
                                                                const bf16* __restrict__ o, const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                bf16* __restrict__ dqkv, int H, int Tn, float scale, DropSpec drop,
                                                                const uint32_t* __restrict__ maskbits) {
    constexpr int DH = 32, DT = 2, KLD = DH + AF_PAD, NP2 = (NT + 1) / 2, KST = 16 * NW, ROWS = 32 * NP2;
    extern __shared__ __attribute__((aligned(16))) char fb_smem[];
    bf16* Qs = reinterpret_cast<bf16*>(fb_smem);                       // [2][64 * KLD]
    bf16* Ds = Qs + 2 * 64 * KLD;                                      // [2][64 * KLD]
    float* Ls = reinterpret_cast<float*>(Ds + 2 * 64 * KLD);           // [2][64]  lse * log2(e)
    float* Dl = Ls + 128;                                              // [2][64]  delta
    uint32_t* Rk = reinterpret_cast<uint32_t*>(Dl + 128);              // [2][64]  dropout row keys
    bf16* dSl = reinterpret_cast<bf16*>(Rk + 128);                     // [NW waves][ROWS own-key rows * FB_QLD]
    float* X = reinterpret_cast<float*>(dSl + NW * ROWS * FB_QLD);     // [2 (half parity)][NW waves][4 tiles][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int dmodel = H * DH;
    const bf16* qb = q + (size_t)bh * Tn * DH;
    const bf16* kb = k + (size_t)bh * Tn * DH;
    const bf16* vb = vt + (size_t)bh * DH * Tn;
    const bf16* dob = dout + (size_t)b * Tn * dmodel + h * DH;
    const bf16* ob = o + (size_t)b * Tn * dmodel + h * DH;
    bf16* dSw = dSl + wid * ROWS * FB_QLD;
    const int nqb = (Tn + AF_QB - 1) / AF_QB, nch = (Tn + AF_KC - 1) / AF_KC;

    // ---- K^T fragments of the own keys: stage the rows (local row 16 j + cc = key 128 j + 16 w + cc, zero beyond) in the
    // private tile and read them transposed; then clear the tile (rows of absent tiles must read as zero dS later).
    // A wave's LDS operations execute in order, so no barrier is needed around its private tile.
#pragma unroll
    for (int it = 0; it < ROWS / 16; ++it) {
        const int i = lane + 64 * it, L = i >> 2, part = i & 3, j = L >> 4, cc = L & 15;
        const int key = KST * j + 16 * wid + cc;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (j < NT && key < Tn) v = *reinterpret_cast<const u32x4*>(kb + (size_t)key * DH + part * 8);
        *reinterpret_cast<u32x4*>(dSw + L * FB_QLD + part * 8) = v;
    }
    asm volatile("" ::: "memory");
    bf16x8 ktf[DT][NP2];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int p = 0; p < NP2; ++p) ktf[d][p] = trfrag(dSw, FB_QLD, 32 * p + 4 * g, 32 * p + 16 + 4 * g, 16 * d, lane);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < ROWS / 16; ++it) {
        const int i = lane + 64 * it;
        *reinterpret_cast<u32x4*>(dSw + (i >> 2) * FB_QLD + (i & 3) * 8) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- B operands of the own keys: K[key c][dh 8g..], V[key c][dv 8g..]
    bf16x8 kf[NT], vf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int key = min(KST * j + 16 * wid + c, Tn - 1);
        kf[j] = *reinterpret_cast<const bf16x8*>(kb + (size_t)key * DH + 8 * g);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = vb[(size_t)(8 * g + e) * Tn + key];
        vf[j] = v;
    }
    f32x4 adv[DT][NT], adk[DT][NT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int j = 0; j < NT; ++j) { adv[d][j] = f32x4{0.f, 0.f, 0.f, 0.f}; adk[d][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float cs = scale * 1.4426950408889634f;

    // ---- chunk staging: threads 0..255 one 16-byte part of a Q row, threads 256..511 one part of a dO row (+ delta)
    // (loads only in FB_GLOAD: everything computed from them waits until FB_LSTORE, one chunk later, so that no wait for the
    // prefetch lands at the top of a chunk)
    u32x4 rq = u32x4{0u, 0u, 0u, 0u}, ro = u32x4{0u, 0u, 0u, 0u};
    float rl = 0.f;
    uint32_t rr = 0;
#define FB_GLOAD(ch)                                                                                              \
    {                                                                                                             \
        const int q0_ = (ch) * AF_KC, pi_ = tid & 255;                                                            \
        const int row_ = min(q0_ + (pi_ >> 2), Tn - 1), part_ = pi_ & 3;                                          \
        if (tid < 256) rq = *reinterpret_cast<const u32x4*>(qb + (size_t)row_ * DH + part_ * 8);                  \
        else if (tid < 512) {                                                                                     \
            rq = *reinterpret_cast<const u32x4*>(dob + (size_t)row_ * dmodel + part_ * 8);                        \
            ro = *reinterpret_cast<const u32x4*>(ob + (size_t)row_ * dmodel + part_ * 8);                         \
        }                                                                                                         \
        const int qr_ = min(q0_ + (tid & 63), Tn - 1);                                                            \
        if (tid < 64) rl = lse[(size_t)bh * Tn + qr_];                                                            \
        else if (tid < 128) rr = rng_row_key(drop.key, (uint32_t)(bh * Tn + q0_ + (tid & 63)));                  \
    }
#define FB_LSTORE(buf)                                                                                            \
    {                                                                                                             \
        const int pi_ = tid & 255;                                                                                \
        bf16* dst_ = (tid < 256 ? Qs : Ds) + (buf) * 64 * KLD + (pi_ >> 2) * KLD + (pi_ & 3) * 8;                 \
        asm volatile("" : "+v"(rq), "+v"(ro), "+v"(rl));    /* nothing computed from the prefetch before this point */   \
        if (tid < 512) *reinterpret_cast<u32x4*>(dst_) = rq;                                                      \
        if (tid >= 256 && tid < 512) {            /* delta = rowsum(dO o O): 8 products per part, the 4 parts of a row are adjacent lanes */ \
            const bf16x8 df_ = __builtin_bit_cast(bf16x8, rq), of_ = __builtin_bit_cast(bf16x8, ro);              \
            float pd_ = 0.f;                                                                                      \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) pd_ += (float)df_[e] * (float)of_[e];                   \
            pd_ += __shfl_xor(pd_, 1, 64);                                                                        \
            pd_ += __shfl_xor(pd_, 2, 64);                                                                        \
            if ((pi_ & 3) == 0) Dl[(buf) * 64 + (pi_ >> 2)] = pd_;                                                \
        }                                                                                                         \
        if (tid < 64) Ls[(buf) * 64 + tid] = rl * 1.4426950408889634f;                                            \
        else if (tid < 128) Rk[(buf) * 64 + tid - 64] = rr;                                                       \
    }
    u32x4 mwn[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        mwn[j] = u32x4{0u, 0u, 0u, 0u};
        if constexpr (DM == 2)
            mwn[j] = *reinterpret_cast<const u32x4*>(maskbits + ((size_t)(bh * nqb) * nch + min(KST * j + 16 * wid, Tn - 8) / AF_KC) * 256 + (c >> 2) * 16 + 4 * g);
    }
    FB_GLOAD(0);
    FB_LSTORE(0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) FB_GLOAD(ch + 1);
        const bf16* Qc = Qs + (ch & 1) * 64 * KLD;
        const bf16* Dc = Ds + (ch & 1) * 64 * KLD;
        const float* Lc = Ls + (ch & 1) * 64;
        const float* Dlc = Dl + (ch & 1) * 64;
        const uint32_t* Rc = Rk + (ch & 1) * 64;
        const int q0 = ch * AF_KC;
#pragma unroll 1
        for (int ks = 0; ks < 2; ++ks) {
            // ---- per half (32 queries): A fragments of S / dP, transposed dO / Q fragments, row constants
            bf16x8 qfr[2], dfr[2], dtf[DT], qtf[DT];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                qfr[hq] = *reinterpret_cast<const bf16x8*>(Qc + (16 * (2 * ks + hq) + c) * KLD + 8 * g);
                dfr[hq] = *reinterpret_cast<const bf16x8*>(Dc + (16 * (2 * ks + hq) + c) * KLD + 8 * g);
            }
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                dtf[d] = trfrag(Dc, KLD, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * d, lane);
                qtf[d] = trfrag(Qc, KLD, 32 * ks + 4 * g, 32 * ks + 16 + 4 * g, 16 * d, lane);
            }
            float nlv[2][4], dlv[2][4];
            uint32_t rkv[2][4];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = 32 * ks + 16 * hq + 4 * g + r;
                    const float lraw = Lc[ql];                               // unconditional LDS read + select (a conditional read is an exec-mask branch per element)
                    nlv[hq][r] = (FULL || q0 + ql < Tn) ? -lraw : -INFINITY; // queries past the end: exp2(-inf) = 0, no per-score select
                    dlv[hq][r] = Dlc[ql];
                    rkv[hq][r] = DM == 1 ? Rc[ql] : 0u;
                }
            const int qq = q0 + 32 * ks;
            u32x4 mwv[NT];                                                  // keep bits of the own tiles for THIS half (loaded one half ahead: an L2 / HBM
#pragma unroll                                                              // round trip is as long as a whole half with only two waves per SIMD)
            for (int j = 0; j < NT; ++j) mwv[j] = mwn[j];
            if constexpr (DM == 2) {          // branch-free (clamped indices): a conditional load made hipcc wait for and copy each one on the spot
                const int qn = min(qq + 32, Tn - 8);                        // next half (the last half re-reads itself)
                const size_t qbn = (size_t)(bh * nqb + qn / AF_QB) * nch;
                const int wfn = (qn % AF_QB) >> 5;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    mwn[j] = *reinterpret_cast<const u32x4*>(maskbits + (qbn + min(KST * j + 16 * wid, Tn - 8) / AF_KC) * 256 + wfn * 64 + (c >> 2) * 16 + 4 * g);
            }
            f32x4 dq[4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) dq[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
            // scores and dP of all own tiles first: their MFMA latency hides behind one another instead of in front of every tile's VALU pass
            f32x4 saccv[NT][2], dpav[NT][2];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    saccv[j][hq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[hq], kf[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    dpav[j][hq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[hq], vf[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int keyt = KST * j + 16 * wid;                        // wave-uniform
                if (FULL || keyt < Tn) {
                    const f32x4 (&sacc)[2] = saccv[j];
                    const f32x4 (&dpa)[2] = dpav[j];
                    const u32x4 mw = mwv[j];
                    const int sh = 4 * ((keyt % AF_KC) >> 4) + (c & 3);
                    const uint32_t key = (uint32_t)(keyt + c);
                    // element (hq, r): query q0 + 32ks + 16hq + 4g + r, key keyt + c
                    float pd[2][4], ds[2][4];
                    // DM == 2: the keep flag of element (hq, r) is bit 16hq + sh of mw[r]: one variable shift per r, then a sign-extending
                    // bit-field extract gives the 0 / ~0 mask that is ANDed onto the two float values (and / compare / two selects before)
                    uint32_t mws[4] = {0u, 0u, 0u, 0u};
                    if constexpr (DM == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) mws[r] = mw[r] >> sh;
                    }
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[hq][r], cs, nlv[hq][r]));
                            float dp = dpa[hq][r], pdv = pv;
                            if constexpr (DM == 2) {
                                const uint32_t km = (uint32_t)__builtin_amdgcn_sbfe((int)mws[r], 16 * hq, 1);
                                dp = __uint_as_float(__float_as_uint(dp * drop.scale) & km);
                                pdv = __uint_as_float(__float_as_uint(pv) & km);      // * drop.scale once per dV output
                            } else if constexpr (DM != 0) {
                                const bool keep = rng_keep_q(rkv[hq][r], key, drop.thr);
                                dp = keep ? dp * drop.scale : 0.f;
                                pdv = keep ? pv : 0.f;
                            }
                            pd[hq][r] = pdv;
                            ds[hq][r] = pv * (dp - dlv[hq][r]);              // * scale once per dK / dQ output
                        }
                    if (!FULL && keyt + 16 > Tn && (int)key >= Tn) {                  // ragged last key tile (T % 16 == 8): its absent keys contribute nothing
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { pd[hq][r] = 0.f; ds[hq][r] = 0.f; }
                    }
                    const bf16x8 pdb = pack8(pd[0], pd[1]);
                    const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        adv[d][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtf[d], pdb, adv[d][j], 0, 0, 0);
                        adk[d][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[d], dsb, adk[d][j], 0, 0, 0);
                    }
                    // dS tile row (own key 16 j + c), queries 16hq + 4g .. +3: the halves of dsb are exactly those two 8-byte pieces
                    const u32x4 dw = __builtin_bit_cast(u32x4, dsb);
                    *reinterpret_cast<u32x2*>(dSw + (16 * j + c) * FB_QLD + 4 * g) = u32x2{dw.x, dw.y};
                    *reinterpret_cast<u32x2*>(dSw + (16 * j + c) * FB_QLD + 16 + 4 * g) = u32x2{dw.z, dw.w};
                }
            }
            // ---- dQ^T (dh x 32 queries) over the own keys, then the fixed-order sum over the NW waves
            asm volatile("" ::: "memory");
#pragma unroll
            for (int p = 0; p < NP2; ++p)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const bf16x8 bfr = trfrag(dSw, FB_QLD, 32 * p + 4 * g, 32 * p + 16 + 4 * g, 16 * qt, lane);
#pragma unroll
                    for (int d = 0; d < DT; ++d) dq[2 * d + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[d][p], bfr, dq[2 * d + qt], 0, 0, 0);
                }
            if constexpr (DM == 2) {          // the next half's keep bits have had this whole half to arrive: wait for them HERE, in front of this
#pragma unroll                                // half's dQ store, or the wait at the top of the next half would also sit out that store's round trip
                for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(mwn[j]));
            }
            float* Xh = X + ks * (NW * 4 * 64 * 4);                          // two halves per chunk: the parity of the half picks the buffer
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) *reinterpret_cast<f32x4*>(Xh + ((wid * 4 + t4) * 64 + lane) * 4) = dq[t4];
            if (ks == 1 && more) FB_LSTORE((ch + 1) & 1);
            __syncthreads();                                                // the ONE barrier of a half: partial dQ tiles (and the next chunk) visible;
            if (tid < 256) {                                                // waves 4..7 run ahead into the next half while waves 0..3 sum this one
                const int t4 = tid >> 6, d = t4 >> 1, qt = t4 & 1;
                f32x4 a = *reinterpret_cast<const f32x4*>(Xh + (t4 * 64 + lane) * 4);
#pragma unroll
                for (int wv = 1; wv < NW; ++wv) a += *reinterpret_cast<const f32x4*>(Xh + ((wv * 4 + t4) * 64 + lane) * 4);
                const int qrow = qq + 16 * qt + c;
                if (qrow < Tn) {
                    bf16x4 w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (bf16)(a[r] * scale);
                    *reinterpret_cast<bf16x4*>(dqkv + ((size_t)b * Tn + qrow) * (3 * dmodel) + h * 3 * DH + 16 * d + 4 * g) = w;
                }
            }
        }
    }
#undef FB_GLOAD
#undef FB_LSTORE
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int key = KST * j + 16 * wid + c;
        if (key < Tn) {
            bf16* drow = dqkv + ((size_t)b * Tn + key) * (3 * dmodel) + h * 3 * DH;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                bf16x4 wk, wv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { wk[r] = (bf16)(adk[d][j][r] * scale); wv[r] = (bf16)(adv[d][j][r] * (DM != 0 ? drop.scale : 1.f)); }
                *reinterpret_cast<bf16x4*>(drow + DH + 16 * d + 4 * g) = wk;
                *reinterpret_cast<bf16x4*>(drow + 2 * DH + 16 * d + 4 * g) = wv;
            }
        }
    }
}

int g_attn_bwd_two_pass = 0;     // tests / tools: 1 forces the two-kernel backward

template <int NW, int NT, int DM, bool FULL>
static int launch_fused_bwd_(const void* q, const void* k, const void* vt, const void* o, const void* dout, const float* lse, void* dqkv,
                             int B, int H, int T, float scale, DropSpec drop, const uint32_t* maskbits, hipStream_t s) {
    constexpr int SM = fb_smem_bytes(NW, NT);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_fused_kernel<NW, NT, DM, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, SM) != hipSuccess) {
            ishara_set_error("attn_bwd_fused: cannot reserve %d bytes of LDS", SM); return -2;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_bwd_fused_kernel<NW, NT, DM, FULL>), dim3(B * H), dim3(NW * 64), SM, s, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (const bf16*)o,
                       (const bf16*)dout, lse, (bf16*)dqkv, H, T, scale, drop, maskbits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
template <int NW, int NT>
static int launch_fused_bwd(int dm, const void* q, const void* k, const void* vt, const void* o, const void* dout, const float* lse, void* dqkv,
                            int B, int H, int T, float scale, DropSpec drop, const uint32_t* maskbits, hipStream_t s) {
#define FB_GO(DMM) (T == 16 * NW * NT ? launch_fused_bwd_<NW, NT, DMM, true>(q, k, vt, o, dout, lse, dqkv, B, H, T, scale, drop, maskbits, s) \
                                      : launch_fused_bwd_<NW, NT, DMM, false>(q, k, vt, o, dout, lse, dqkv, B, H, T, scale, drop, maskbits, s))
    return dm == 0 ? FB_GO(0) : (dm == 1 ? FB_GO(1) : FB_GO(2));
#undef FB_GO
}

int launch_attn_bwd_mfma(const void* q, const void* k, const void* vt, const void* o, const void* dout, const float* lse,
                         float* delta, void* dqkv, int B, int H, int T, int dh, float scale, DropSpec drop, uint32_t* maskbits, hipStream_t s) {
    if (T % 8 != 0) { ishara_set_error("attn_bwd_mfma: T %% 8 != 0"); return -1; }
    dim3 grid(((T + AF_QB - 1) / AF_QB) * B * H);
    const int dm = drop.thr == 0 ? 0 : (maskbits ? 2 : 1);      // measured per layer (B256 H8 T384 dh32): hash fwd 155 + bwd 452 us, cached bits 168 + 361 us
    if (dh == 32 && T <= 384 && !g_attn_bwd_two_pass) {          // one-pass backward: S, dP and the elementwise pass computed once
        // (waves, key tiles per wave): 12 waves = 3 per SIMD wherever T allows it with <= 2 tiles
        if (T <= 128) return launch_fused_bwd<8, 1>(dm, q, k, vt, o, dout, lse, dqkv, B, H, T, scale, drop, maskbits, s);
        if (T <= 192) return launch_fused_bwd<12, 1>(dm, q, k, vt, o, dout, lse, dqkv, B, H, T, scale, drop, maskbits, s);
        if (T <= 256) return launch_fused_bwd<8, 2>(dm, q, k, vt, o, dout, lse, dqkv, B, H, T, scale, drop, maskbits, s);
        return launch_fused_bwd<12, 2>(dm, q, k, vt, o, dout, lse, dqkv, B, H, T, scale, drop, maskbits, s);
    }
#define ATT_BWD(DHH, DMM)                                                                                                                    \
    do {                                                                                                                                     \
        hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<DHH, DMM>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (const bf16*)o, \
                           (const bf16*)dout, lse, delta, (bf16*)dqkv, H, T, scale, drop, B * H, maskbits);                                  \
        hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<DHH, DMM>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)vt,     \
                           (const bf16*)dout, lse, (const float*)delta, (bf16*)dqkv, H, T, scale, drop, B * H, maskbits);                     \
    } while (0)
#define ATT_BWD_DM(DHH) do { if (dm == 0) ATT_BWD(DHH, 0); else if (dm == 1) ATT_BWD(DHH, 1); else ATT_BWD(DHH, 2); } while (0)
    if (dh == 32) ATT_BWD_DM(32);
    else if (dh == 64) ATT_BWD_DM(64);
    else { ishara_set_error("attn_bwd_mfma: head dim %d unsupported (32, 64)", dh); return -1; }
#undef ATT_BWD_DM
#undef ATT_BWD
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t attn_mask_words(int B, int H, int T) { return (size_t)B * H * ((T + AF_QB - 1) / AF_QB) * ((T + AF_KC - 1) / AF_KC) * 256; }

// fp16 operands (inference, no dropout)
int launch_attn_fwd_mfma_f16(const void* q, const void* k, const void* vt, void* o, float* lse, int B, int H, int T, int dh, float scale, hipStream_t s) {
    if (T % 8 != 0) { ishara_set_error("attn_fwd_mfma: T %% 8 != 0"); return -1; }
    dim3 grid(((T + AF_QB - 1) / AF_QB) * B * H);
    const DropSpec nodrop{0u, 0u, 1.f};
    if (dh == 32) hipLaunchKernelGGL((attn_fwd_mfma_kernel<32, 0, f16>), grid, dim3(256), 0, s, (const f16*)q, (const f16*)k, (const f16*)vt, (f16*)o, lse, H, T, scale, nodrop, B * H, (uint32_t*)nullptr);
    else if (dh == 64) hipLaunchKernelGGL((attn_fwd_mfma_kernel<64, 0, f16>), grid, dim3(256), 0, s, (const f16*)q, (const f16*)k, (const f16*)vt, (f16*)o, lse, H, T, scale, nodrop, B * H, (uint32_t*)nullptr);
    else { ishara_set_error("attn_fwd_mfma: head dim %d unsupported (32, 64)", dh); return -1; }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_attn_fwd_mfma(const void* q, const void* k, const void* vt, void* o, float* lse,
                         int B, int H, int T, int dh, float scale, DropSpec drop, uint32_t* maskbits, hipStream_t s) {
    if (T % 8 != 0) { ishara_set_error("attn_fwd_mfma: T %% 8 != 0"); return -1; }
    dim3 grid(((T + AF_QB - 1) / AF_QB) * B * H);
    const int dm = drop.thr == 0 ? 0 : (maskbits ? 2 : 1);      // measured per layer (B256 H8 T384 dh32): hash fwd 155 + bwd 452 us, cached bits 168 + 361 us
#define ATT_FWD(DHH, DMM) hipLaunchKernelGGL((attn_fwd_mfma_kernel<DHH, DMM>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)k, (const bf16*)vt, (bf16*)o, lse, H, T, scale, drop, B * H, maskbits)
#define ATT_FWD_DM(DHH) do { if (dm == 0) ATT_FWD(DHH, 0); else if (dm == 1) ATT_FWD(DHH, 1); else ATT_FWD(DHH, 2); } while (0)
    if (dh == 32) ATT_FWD_DM(32);
    else if (dh == 64) ATT_FWD_DM(64);
    else { ishara_set_error("attn_fwd_mfma: head dim %d unsupported (32, 64)", dh); return -1; }
#undef ATT_FWD_DM
#undef ATT_FWD
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
