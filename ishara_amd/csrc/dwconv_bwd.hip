// dwconv_bwd.hip — fused backward of the depthwise Conv1D over time (gfx950).
//
// Forward (reference notebook c5: CausalDWConv1D / the conformer ConvolutionModule's DepthwiseConv1D):
//     y[b,t,c] = sum_j w[j,c] * f(x)[b, t - padl + j, c]          f = identity | swish | GLU (x = [a | g], f = a * sigmoid(g))
// Backward, with the ONE sliding window D[j] = dy[b, t + padl - j, c] (j = 0..K-1) around the position t:
//     dx[b,t,c]  = f'(x[b,t,c]) * sum_j w[j,c] * D[j]
//     dw[j,c]   += f(x[b,t,c]) * D[j]                             dbias[c] += dy[b,t,c]
// so the data gradient and the weight gradient need the same K rows of dy and one row of x per step: one pass over dy
// and x (and one write of dx) instead of the two passes (4 tensor reads) of the separate kernels.
//
// Thread = 4 channels (8-byte bf16 accesses, 512 contiguous bytes per wave for C = 512) walking a segment of time steps of
// one sample; the window lives in registers as PACKED bf16 rows (2 VGPRs per row; dy is bf16 in memory, so this is
// lossless) in K circular slots, the loop is unrolled by K so every slot index is a compile-time constant; w[K][4] and
// the dw[K][4] accumulators are fp32 registers (float2 pairs -> v_pk_fma_f32).  A workgroup = (256 / (C/4)) item lanes x
// C/4 channel quads looping over (sample, segment) items; at the end the lanes are combined in LDS and the workgroup writes
// ONE partial row part[blockIdx.x][(K+1)*C] (dw then dbias), summed by reduce_slabs: no same-address atomics.
#include "common.h"
#include "kernels.h"

typedef __attribute__((ext_vector_type(2))) float dwf2;
typedef __attribute__((ext_vector_type(2))) uint32_t dwu2;

// a row of 4 channels: packed bf16 (2 dwords) or 4 floats
template <typename T> struct DwRow;
template <> struct DwRow<bf16> {
    dwu2 r;
    DEVI void zero() { r = dwu2{0u, 0u}; }
    DEVI void load(const bf16* p) { r = __builtin_nontemporal_load(reinterpret_cast<const dwu2*>(p)); }      // every operand of the backward conv is at its last use
    DEVI void unpack(dwf2& lo, dwf2& hi) const {
        lo = dwf2{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
        hi = dwf2{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    }
    DEVI void pack(dwf2 lo, dwf2 hi) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 a = {(bf16)lo.x, (bf16)lo.y}, b = {(bf16)hi.x, (bf16)hi.y};
        r = dwu2{__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b)};
    }
};
template <> struct DwRow<float> {
    float4 r;
    DEVI void zero() { r = make_float4(0.f, 0.f, 0.f, 0.f); }
    DEVI void load(const float* p) { r = *reinterpret_cast<const float4*>(p); }
    DEVI void unpack(dwf2& lo, dwf2& hi) const { lo = dwf2{r.x, r.y}; hi = dwf2{r.z, r.w}; }
    DEVI void pack(dwf2 lo, dwf2 hi) { r = make_float4(lo.x, lo.y, hi.x, hi.y); }
};
// BatchNorm backward applied to a dy row as it enters the window (BN = true): row <- dy * k1 + k0 - h * k2, the arithmetic of
// bn_bwd_apply_kernel (elementwise.hip), rounded to the storage type exactly where that kernel's output was
template <typename T> DEVI void dw_bn_row(DwRow<T>& d, const DwRow<T>& h, const dwf2 (&k0)[2], const dwf2 (&k1)[2], const dwf2 (&k2)[2]) {
    dwf2 dl, dh, hl, hh;
    d.unpack(dl, dh); h.unpack(hl, hh);
    d.pack(dl * k1[0] + k0[0] - hl * k2[0], dh * k1[1] + k0[1] - hh * k2[1]);
}
DEVI void dw_store4(bf16* p, dwf2 lo, dwf2 hi) { const float v[4] = {lo.x, lo.y, hi.x, hi.y}; store4(p, v); }
DEVI void dw_store4(float* p, dwf2 lo, dwf2 hi) { *reinterpret_cast<float4*>(p) = make_float4(lo.x, lo.y, hi.x, hi.y); }

template <int K, bool BN = false, int LBX = 0> struct DwCfg {
    static constexpr int LB = LBX ? LBX : (K <= 5 ? K : 4);   // steps whose loads are in flight together (LBX: override; K = 11 runs with 8 — 1.355 -> 1.298 ms of depthwise backward per step at config #2, LB 6: 1.338)
    static constexpr int WPS = K >= 11 ? 2 : (K >= 5 ? 3 : 4);  // waves per SIMD the register budget allows
};

// WU: C/4 is a multiple of 64, so every wave lies inside one item lane: the item index (and with it every time bound and base
// pointer) is wave-uniform and is kept in scalar registers -> scalar loop control instead of exec-mask branches per step
// BN: dy is the gradient of a BatchNorm OUTPUT; the BatchNorm backward (bn.h = the BatchNorm input = this conv's forward output)
// is applied to every dy row on its way into the window, so the [M, C] gradient of the conv output never exists in memory
template <typename T, int K, int INOP, bool WU, bool BN, int LBX = 0>
__global__ __launch_bounds__(256, (DwCfg<K, BN>::WPS)) void dwconv_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ w,
                                                                            T* __restrict__ dx, float* __restrict__ part,
                                                                            int B, int Tn, int C, int padl, int seg_len, DwBnArgs bn) {
    extern __shared__ float red[];             // (K+1)*C block accumulator
    constexpr int LB = DwCfg<K, BN, LBX>::LB;
    const int tid = threadIdx.x;
    const int cg = C >> 2, lanes = 256 / cg;
    const int c4 = tid % cg;
    const int il = WU ? __builtin_amdgcn_readfirstlane(tid / cg) : tid / cg;
    const int ch = c4 * 4;
    const int Cin = (INOP == DWIN_GLU) ? 2 * C : C;
    const int nseg = (Tn + seg_len - 1) / seg_len;
    const int nitems = B * nseg;

    dwf2 wr[K][2], acc[K][2], accb[2] = {dwf2{0.f, 0.f}, dwf2{0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)j * C + ch);
        wr[j][0] = dwf2{wv.x, wv.y}; wr[j][1] = dwf2{wv.z, wv.w};
        acc[j][0] = dwf2{0.f, 0.f}; acc[j][1] = dwf2{0.f, 0.f};
    }
    if (il < lanes) {
        for (int item = blockIdx.x * lanes + il; item < nitems; item += gridDim.x * lanes) {
            const int b = item / nseg, t0 = (item % nseg) * seg_len, tend = min(Tn, t0 + seg_len);
            const T* dyb = dy + (size_t)b * Tn * C + ch;
            const T* xb = x + (size_t)b * Tn * Cin + ch;
            T* dxb = dx + (size_t)b * Tn * Cin + ch;
            const T* hb = BN ? reinterpret_cast<const T*>(bn.h) + (size_t)b * Tn * C + ch : nullptr;
            dwf2 k0[2], k1[2], k2[2];
            if (BN) {
                float mu[4], rs[4], aa[4], fc[4], ee[4], g[4] = {1.f, 1.f, 1.f, 1.f};
                load4(bn.mean + ch, mu); load4(bn.rstd + ch, rs); load4(bn.a + ch, aa); load4(bn.Fc + ch, fc);
                load4(bn.E + (bn.e_per_sample ? (size_t)b * C : 0) + ch, ee);
                if (bn.sg) load4(bn.sg + (size_t)b * C + ch, g);
                float q0[4], q1[4], q2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { q2[e] = aa[e] * rs[e] * fc[e]; q1[e] = aa[e] * g[e]; q0[e] = aa[e] * ee[e] + mu[e] * q2[e]; }
                k0[0] = dwf2{q0[0], q0[1]}; k0[1] = dwf2{q0[2], q0[3]};
                k1[0] = dwf2{q1[0], q1[1]}; k1[1] = dwf2{q1[2], q1[3]};
                k2[0] = dwf2{q2[0], q2[1]}; k2[1] = dwf2{q2[2], q2[3]};
            }
            // slot m (1..K-1) holds dy[t0 + padl - K + m]; slot u receives dy[tb + u + padl] at step u of a K-step group,
            // so that D[j] at step u is slot (u - j) mod K
            DwRow<T> win[K];
            win[0].zero();
#pragma unroll
            for (int m = 1; m < K; ++m) {
                const int tin = t0 + padl - K + m;
                if (tin >= 0 && tin < Tn) {
                    win[m].load(dyb + (size_t)tin * C);
                    if (BN) { DwRow<T> hr; hr.load(hb + (size_t)tin * C); dw_bn_row(win[m], hr, k0, k1, k2); }
                } else win[m].zero();
                // dbias = sum of every dy row exactly once: the rows [0, padl) of a sample as preloaded rows of its first
                // segment, every other row when it enters the window as the newest row (below)
                if (t0 == 0 && tin >= 0 && tin < padl) { dwf2 dl, dh; win[m].unpack(dl, dh); accb[0] += dl; accb[1] += dh; }
            }
            for (int tb = t0; tb < tend; tb += K) {
#pragma unroll
                for (int u0 = 0; u0 < K; u0 += LB) {
                    DwRow<T> nd[LB], xr[LB], gr[LB], hn[BN ? LB : 1];
#pragma unroll
                    for (int uu = 0; uu < LB; ++uu) {
                        const int t = tb + u0 + uu;
                        nd[uu].zero(); xr[uu].zero(); gr[uu].zero();
                        if (u0 + uu < K && t < tend) {
                            if (t + padl < Tn) { nd[uu].load(dyb + (size_t)(t + padl) * C); if (BN) hn[uu].load(hb + (size_t)(t + padl) * C); }
                            xr[uu].load(xb + (size_t)t * Cin);
                            if (INOP == DWIN_GLU) gr[uu].load(xb + (size_t)t * Cin + C);
                        }
                    }
#pragma unroll
                    for (int uu = 0; uu < LB; ++uu) {
                        const int u = u0 + uu;
                        if (u < K) {
                            const int t = tb + u;
                            if (BN && t < tend && t + padl < Tn) dw_bn_row(nd[uu], hn[uu], k0, k1, k2);      // rows out of range stay zero
                            win[u] = nd[uu];
                            { dwf2 dl, dh; nd[uu].unpack(dl, dh); accb[0] += dl; accb[1] += dh; }     // zero row when out of range
                            if (t < tend) {
                                // input transform: f (for dw) and f' (for dx)
                                dwf2 xl, xh, f[2], fp[2], fg[2];
                                xr[uu].unpack(xl, xh);
                                if (INOP == DWIN_SWISH) {
                                    const float xv[4] = {xl.x, xl.y, xh.x, xh.y};
                                    float fv[4], dv[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float sg = sigmoidf_(xv[e]);
                                        fv[e] = xv[e] * sg;
                                        dv[e] = sg * (1.f + xv[e] * (1.f - sg));
                                    }
                                    f[0] = dwf2{fv[0], fv[1]}; f[1] = dwf2{fv[2], fv[3]};
                                    fp[0] = dwf2{dv[0], dv[1]}; fp[1] = dwf2{dv[2], dv[3]};
                                } else if (INOP == DWIN_GLU) {
                                    dwf2 gl, gh;
                                    gr[uu].unpack(gl, gh);
                                    const float av[4] = {xl.x, xl.y, xh.x, xh.y}, gv[4] = {gl.x, gl.y, gh.x, gh.y};
                                    float fv[4], dv[4], dg[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float sg = sigmoidf_(gv[e]);
                                        fv[e] = av[e] * sg;            // f
                                        dv[e] = sg;                    // df/da
                                        dg[e] = av[e] * sg * (1.f - sg);   // df/dg
                                    }
                                    f[0] = dwf2{fv[0], fv[1]}; f[1] = dwf2{fv[2], fv[3]};
                                    fp[0] = dwf2{dv[0], dv[1]}; fp[1] = dwf2{dv[2], dv[3]};
                                    fg[0] = dwf2{dg[0], dg[1]}; fg[1] = dwf2{dg[2], dg[3]};
                                } else {
                                    f[0] = xl; f[1] = xh;
                                    fp[0] = dwf2{1.f, 1.f}; fp[1] = dwf2{1.f, 1.f};
                                }
                                dwf2 pre[2] = {dwf2{0.f, 0.f}, dwf2{0.f, 0.f}};
#pragma unroll
                                for (int j = 0; j < K; ++j) {
                                    dwf2 dl, dh;
                                    win[(u - j + K) % K].unpack(dl, dh);
                                    pre[0] += wr[j][0] * dl; pre[1] += wr[j][1] * dh;
                                    acc[j][0] += f[0] * dl; acc[j][1] += f[1] * dh;
                                }
                                dw_store4(dxb + (size_t)t * Cin, fp[0] * pre[0], fp[1] * pre[1]);
                                if (INOP == DWIN_GLU) dw_store4(dxb + (size_t)t * Cin + C, fg[0] * pre[0], fg[1] * pre[1]);
                            }
                        }
                    }
                }
            }
        }
    }
    // combine the item lanes of the workgroup in LDS, then one coalesced partial row
    const int n = (K + 1) * C;
    for (int q = tid; q < n; q += 256) red[q] = 0.f;
    __syncthreads();
    for (int l = 0; l < lanes; ++l) {
        if (il == l) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                float4* p = reinterpret_cast<float4*>(red + j * C + ch);
                float4 v = *p;
                v.x += acc[j][0].x; v.y += acc[j][0].y; v.z += acc[j][1].x; v.w += acc[j][1].y;
                *p = v;
            }
            float4* p = reinterpret_cast<float4*>(red + K * C + ch);
            float4 v = *p;
            v.x += accb[0].x; v.y += accb[0].y; v.z += accb[1].x; v.w += accb[1].y;
            *p = v;
        }
        __syncthreads();
    }
    float* dst = part + (size_t)blockIdx.x * n;
    for (int q = tid; q < n; q += 256) dst[q] = red[q];
}

bool dwconv_bwd_fused_ok(int dt, int C, int k, int padl) {
    if (!(k == 3 || k == 5 || k == 11 || k == 15)) return false;
    if (dt != DT_BF16 && k > 5) return false;                 // fp32 rows would need K*4 more registers for the window
    if (C % 4 != 0 || C / 4 > 256 || padl < 0 || padl >= k) return false;
    return true;
}

// partial rows: returns the number of rows written to `part` ((k+1)*C floats each), or -1
template <typename T>
static int run_dw_fused(int inop, const T* dy, const T* x, const float* w, T* dx, float* part, int B, int Tn, int C, int k, int padl, int max_rows, hipStream_t s, const DwBnArgs& bn) {
    const int cg = C / 4, lanes = 256 / cg;
    const int seg_len = k >= 11 ? 48 : 32;
    const int nseg = (Tn + seg_len - 1) / seg_len;
    const int nitems = B * nseg;
    int grid = (nitems + lanes - 1) / lanes;
    if (grid > max_rows) grid = max_rows;
    if (grid > 512) grid = 512;
    const size_t sh = (size_t)(k + 1) * C * sizeof(float);
    static const int lbx = getenv("ISHARA_DW_LB") ? atoi(getenv("ISHARA_DW_LB")) : 8;      // rows in flight per group at K >= 11: 8 (A/B: ISHARA_DW_LB=4 / 6)
#define DWF2(KK, OP, WUU, BNN) do { if (KK >= 11 && lbx == 6) hipLaunchKernelGGL((dwconv_bwd_fused_kernel<T, KK, OP, WUU, BNN, (KK >= 11 ? 6 : 0)>), dim3(grid), dim3(256), sh, s, dy, x, w, dx, part, B, Tn, C, padl, seg_len, bn); \
    else if (KK >= 11 && lbx == 8) hipLaunchKernelGGL((dwconv_bwd_fused_kernel<T, KK, OP, WUU, BNN, (KK >= 11 ? 8 : 0)>), dim3(grid), dim3(256), sh, s, dy, x, w, dx, part, B, Tn, C, padl, seg_len, bn); \
    else hipLaunchKernelGGL((dwconv_bwd_fused_kernel<T, KK, OP, WUU, BNN>), dim3(grid), dim3(256), sh, s, dy, x, w, dx, part, B, Tn, C, padl, seg_len, bn); } while (0)
#define DWF(KK, OP) do { if (bn.h) { if (cg % 64 == 0) DWF2(KK, OP, true, true); else DWF2(KK, OP, false, true); } \
                         else { if (cg % 64 == 0) DWF2(KK, OP, true, false); else DWF2(KK, OP, false, false); } } while (0)
#define DWFK(OP) switch (k) { case 3: DWF(3, OP); break; case 5: DWF(5, OP); break; case 11: if constexpr (is_bf16_t<T>::value) { DWF(11, OP); } break; \
                              default: if constexpr (is_bf16_t<T>::value) { DWF(15, OP); } break; }
    if (inop == DWIN_SWISH) { DWFK(DWIN_SWISH) } else if (inop == DWIN_GLU) { DWFK(DWIN_GLU) } else { DWFK(DWIN_NONE) }
#undef DWFK
#undef DWF
#undef DWF2
    return hipGetLastError() == hipSuccess ? grid : -1;
}

int launch_dwconv_bwd_fused(int dt, int inop, const void* dy, const void* x, const float* w, void* dx, float* part,
                            int B, int T, int C, int k, int padl, int max_rows, hipStream_t s, const DwBnArgs& bn) {
    if (dt == DT_BF16) return run_dw_fused<bf16>(inop, (const bf16*)dy, (const bf16*)x, w, (bf16*)dx, part, B, T, C, k, padl, max_rows, s, bn);
    return run_dw_fused<float>(inop, (const float*)dy, (const float*)x, w, (float*)dx, part, B, T, C, k, padl, max_rows, s, bn);
}
