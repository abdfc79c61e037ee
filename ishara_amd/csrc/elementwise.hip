// HBM-bound kernels of the Ishara encoder: LayerNorm, depthwise conv (causal / same,
// fused Swish / GLU input ops, fused BatchNorm + GAP statistics), BatchNorm / ECA /
// Squeeze-Excite small ops and their backward passes.  Activations are [B*T, C]
// row-major (channel fastest), read and written as 16-byte (8 x bf16) or 2 x 16-byte
// (8 x f32) chunks per lane; all arithmetic is fp32.
#include "kernels.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : -2)
void launch_reduce_slabs(const float* slab, float* out, int n, int splits, size_t stride, hipStream_t s);

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// =====================================================================================
// LayerNorm.  A row is owned by a group of G lanes (G = pow2 >= C/8, <= 64), each lane
// holding one 8-channel chunk in registers (C <= 512) -> exact two-pass statistics.
// =====================================================================================
// sum over aligned groups of G lanes.  Within a 16-lane DPP row the butterfly runs on the VALU (quad_perm / row_half_mirror /
// row_mirror: no LDS round trip); only the steps across rows (G = 32, 64) use ds_bpermute.
template <int CTRL> DEVI float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int G> DEVI float group_sum(float v) {
    if constexpr (G >= 2) v = dpp_add<0xB1>(v);          // quad_perm [1,0,3,2]: lane ^ 1
    if constexpr (G >= 4) v = dpp_add<0x4E>(v);          // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (G >= 8) v = dpp_add<0x141>(v);         // row_half_mirror: lane -> 7 - lane within 8 (sums the two quads)
    if constexpr (G >= 16) v = dpp_add<0x140>(v);        // row_mirror: lane -> 15 - lane within 16
#pragma unroll
    for (int o = 16; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M, int C) {
    constexpr int RPB = 256 / G;
    const int tid = threadIdx.x, gl = tid % G, gr = tid / G;
    const int nch = C >> 3;
    const bool act = gl < nch;
    float ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ga[e] = act ? gamma[gl * 8 + e] : 0.f; be[e] = act ? beta[gl * 8 + e] : 0.f; }
    // four rows in flight per lane group
    const int stride = gridDim.x * RPB;
    for (int row0 = blockIdx.x * RPB + gr; row0 < M; row0 += 4 * stride) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + u * stride;
            if (act && row < M) load8(x + (size_t)row * C + gl * 8, v[u]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + u * stride;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[u][e];
            const float mu = group_sum<G>(s) / (float)C;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = act ? v[u][e] - mu : 0.f; q += d * d; }
            const float rs = rsqrtf(group_sum<G>(q) / (float)C + eps);
            if (row < M) {
                if (act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[u][e] = (v[u][e] - mu) * rs * ga[e] + be[e];
                    store8(y + (size_t)row * C + gl * 8, v[u]);
                }
                if (gl == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
            }
        }
    }
}

int launch_layernorm_fwd(int dt, const void* x, const float* gamma, const float* beta, float eps,
                         void* y, float* mean, float* rstd, int M, int C, hipStream_t s) {
    if (C % 8 != 0 || C > 512) { ishara_set_error("layernorm: C=%d unsupported (need C%%8==0, C<=512)", C); return -1; }
    const int G = next_pow2(C / 8);
    const int rpb = 256 / G;
    // persistent-style grid (2-4 workgroups per CU looping over row batches): at M = 98304 one batch per workgroup (3072
    // workgroups) measured 33.9 us, 512-1536 workgroups 25.7-26.7 us
    const int grid = max(1, min((M + 4 * rpb - 1) / (4 * rpb), 768));
#define LN_F(TT, GG) hipLaunchKernelGGL((layernorm_fwd_kernel<TT, GG>), dim3(grid), dim3(256), 0, s, (const TT*)x, gamma, beta, eps, (TT*)y, mean, rstd, M, C)
#define LN_FG(TT) switch (G) { case 1: LN_F(TT, 1); break; case 2: LN_F(TT, 2); break; case 4: LN_F(TT, 4); break; case 8: LN_F(TT, 8); break; \
                               case 16: LN_F(TT, 16); break; case 32: LN_F(TT, 32); break; default: LN_F(TT, 64); break; }
    if (dt == DT_BF16) { LN_FG(bf16) } else if (dt == DT_F16) { LN_FG(f16) } else { LN_FG(float) }
    return LAUNCH_OK();
}

// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)) (+ resid); dgamma += sum_rows dy*xhat; dbeta += sum_rows dy
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const T* __restrict__ resid,
                                                            T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ partial, int M, int C) {
    constexpr int RPB = 256 / G;
    __shared__ float red[2][256][8];
    const int tid = threadIdx.x, gl = tid % G, gr = tid / G;
    const int nch = C >> 3;
    const bool act = gl < nch;
    float ga[8], dg[8], db[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ga[e] = act ? gamma[gl * 8 + e] : 0.f; dg[e] = 0.f; db[e] = 0.f; }
    // two rows in flight per group: 4 x 16-byte loads outstanding per lane
    const int stride = gridDim.x * RPB;
    for (int row0 = blockIdx.x * RPB + gr; row0 < M; row0 += 2 * stride) {
        float d[2][8], xv[2][8], o[2][8];
        float mu[2], rs[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = row0 + u * stride;
            const bool ok = act && row < M;
            if (ok) { load8(dy + (size_t)row * C + gl * 8, d[u]); load8(x + (size_t)row * C + gl * 8, xv[u]); }
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { d[u][e] = 0.f; xv[u][e] = 0.f; }
            }
            if (resid && ok) load8(resid + (size_t)row * C + gl * 8, o[u]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[u][e] = 0.f;
            }
            mu[u] = row < M ? mean[row] : 0.f;
            rs[u] = row < M ? rstd[row] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = row0 + u * stride;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[u][e] = (act && row < M) ? (xv[u][e] - mu[u]) * rs[u] : 0.f;     // xhat
                dg[e] += d[u][e] * xv[u][e];
                db[e] += d[u][e];
                d[u][e] *= ga[e];
                s1 += d[u][e];
                s2 += d[u][e] * xv[u][e];
            }
            s1 = group_sum<G>(s1) / (float)C;
            s2 = group_sum<G>(s2) / (float)C;
            if (act && row < M) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[u][e] += rs[u] * (d[u][e] - s1 - xv[u][e] * s2);
                store8(dx + (size_t)row * C + gl * 8, o[u]);
            }
        }
    }
    // reduce dgamma/dbeta over the RPB row groups of the block, then one atomic per channel
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = dg[e]; red[1][tid][e] = db[e]; }
    __syncthreads();
    if (gr == 0 && act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RPB; ++r) { a += red[0][r * G + gl][e]; b += red[1][r * G + gl][e]; }
            if (partial) {      // per-block partial rows [grid][2][C], summed by reduce_slabs (no same-address atomics)
                partial[((size_t)blockIdx.x * 2) * C + gl * 8 + e] = a;
                partial[((size_t)blockIdx.x * 2 + 1) * C + gl * 8 + e] = b;
            } else {
                atomicAdd(dgamma + gl * 8 + e, a);
                atomicAdd(dbeta + gl * 8 + e, b);
            }
        }
    }
}

void launch_reduce_slabs(const float* slab, float* out, int n, int splits, size_t stride, hipStream_t s);
void launch_reduce_slabs2(const float* slab, float* out0, int n0, float* out1, int n1, int splits, size_t stride, hipStream_t s, int nb = 0, int nbv = 0);
bool dwconv_bwd_fused_ok(int dt, int C, int k, int padl);
int launch_dwconv_bwd_fused(int dt, int inop, const void* dy, const void* x, const float* w, void* dx, float* part,
                            int B, int T, int C, int k, int padl, int max_rows, hipStream_t s, const DwBnArgs& bn);

int launch_layernorm_bwd(int dt, const void* dy, const void* x, const float* mean, const float* rstd,
                         const float* gamma, const void* resid, void* dx, float* dgamma, float* dbeta,
                         float* scratch, int M, int C, hipStream_t s) {
    if (C % 8 != 0 || C > 512) { ishara_set_error("layernorm_bwd: C=%d unsupported", C); return -1; }
    const int G = next_pow2(C / 8);
    const int rpb = 256 / G;
    const int grid = max(1, min((M + 2 * rpb - 1) / (2 * rpb), 1024));     // measured in-model: 512 -> 55 us, 1024 -> 45 us, 2048 -> 49 us
#define LN_B(TT, GG) hipLaunchKernelGGL((layernorm_bwd_kernel<TT, GG>), dim3(grid), dim3(256), 0, s, (const TT*)dy, (const TT*)x, mean, rstd, gamma, (const TT*)resid, (TT*)dx, dgamma, dbeta, scratch, M, C)
#define LN_BG(TT) switch (G) { case 1: LN_B(TT, 1); break; case 2: LN_B(TT, 2); break; case 4: LN_B(TT, 4); break; case 8: LN_B(TT, 8); break; \
                               case 16: LN_B(TT, 16); break; case 32: LN_B(TT, 32); break; default: LN_B(TT, 64); break; }
    if (dt == DT_BF16) { LN_BG(bf16) } else { LN_BG(float) }
    if (scratch) launch_reduce_slabs2(scratch, dgamma, C, dbeta, C, grid, (size_t)2 * C, s);
    return LAUNCH_OK();
}
size_t layernorm_bwd_scratch_floats(int C) { return (size_t)2048 * 2 * C; }

// =====================================================================================
// Depthwise conv over time.  One workgroup = 64 output steps x 128 channels of one sample.
// The input tile (+ k-1 halo) is transformed once (Swish / GLU) and staged in LDS as fp32;
// each thread slides an 8-step register window over its 4 channels.
//   OUT_NONE  : y = conv (+bias), optional per-sample sum / sum-of-squares (BN stats, GAP)
//   OUT_DSWISH: y = conv * swish'(aux)                 (backward through a Swish input op)
//   OUT_DGLU  : y[:, :C] = conv*sig(a2) ; y[:, C:] = conv*a1*sig(a2)*(1-sig(a2))   (aux has 2C)
// `flip` indexes the taps in reverse (backward data pass).
// =====================================================================================
enum : int { OUT_NONE = 0, OUT_DSWISH = 1, OUT_DGLU = 2 };
#define DW_TT 64
#define DW_CT 128
#define DW_MAXK 31
#define DWG_TT 32   // time tile of the weight-grad kernel (two LDS tiles must fit 64 KB)

// KC: compile-time tap count (11: the tap loop is fully unrolled, so the 8-row register window slides by renaming
// instead of 28 v_mov per tap and the tap weights are loaded ahead of use); 0: run-time k
template <typename T, int KC, int KM>      // KC: unrolled tap count (0 = runtime k); KM: largest k this instantiation takes (sizes the LDS tile: 37 / 40 / 48 KB)
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     T* __restrict__ y, const T* __restrict__ aux,
                                                     float* __restrict__ ssum, float* __restrict__ ssq,
                                                     int B, int Tn, int C, int k, int padl, int inop, int outop, int flip, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float tile[(DW_TT + KM - 1) * DW_CT];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * DW_TT, c0 = blockIdx.y * DW_CT, b = blockIdx.z;
    const int Cin = (inop == DWIN_GLU) ? 2 * C : C;
    const int rows = DW_TT + k - 1;
    // ---- stage: thread -> chunk (tid&15) of 8 channels, rows (tid>>4) + 16*it.  All global loads of the
    // tile are issued before the first use (rows <= 94 -> at most 6 row groups per thread).
    {
        const int ch = c0 + (tid & 15) * 8;
        constexpr int NIT = (DW_TT + KM - 1 + 15) / 16;
        float v[NIT][8], gl[NIT][8];
        bool ok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = (tid >> 4) + 16 * it;
            const int tin = t0 - padl + r;
            ok[it] = r < rows && tin >= 0 && tin < Tn && ch < C;
            if (ok[it]) {
                const T* p = x + ((size_t)b * Tn + tin) * Cin + ch;
                load8(p, v[it]);
                if (inop == DWIN_GLU) load8(p + C, gl[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (r < rows) {
                if (ok[it]) {
                    if (inop == DWIN_SWISH) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[it][e] = swishf_(v[it][e]);
                    } else if (inop == DWIN_GLU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[it][e] *= sigmoidf_(gl[it][e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
                }
                float* dst = tile + r * DW_CT + (tid & 15) * 8;
                *reinterpret_cast<float4*>(dst) = make_float4(v[it][0], v[it][1], v[it][2], v[it][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[it][4], v[it][5], v[it][6], v[it][7]);
            }
        }
    }
    __syncthreads();
    // ---- compute: thread -> 4 channels (cl) x 8 consecutive steps (tl)
    const int cl = tid & 31, tl = tid >> 5;
    const int ch = c0 + cl * 4;
    const bool cact = ch < C;
    float acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][e] = 0.f;
    if (cact) {
        float4 win[8];
        const float* tp = tile + (tl * 8) * DW_CT + cl * 4;
#pragma unroll
        for (int r = 0; r < 7; ++r) win[r + 1] = *reinterpret_cast<const float4*>(tp + r * DW_CT);
        // invariant at tap j: win[1..7] = tile rows tl*8 + j + (0..6); each tap shifts and loads one row
        auto tap = [&](int j) {
#pragma unroll
            for (int r = 0; r < 7; ++r) win[r] = win[r + 1];
            win[7] = *reinterpret_cast<const float4*>(tp + (j + 7) * DW_CT);
            const float4 wj = *reinterpret_cast<const float4*>(w + (size_t)(flip ? (k - 1 - j) : j) * C + ch);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc[r][0] += wj.x * win[r].x; acc[r][1] += wj.y * win[r].y;
                acc[r][2] += wj.z * win[r].z; acc[r][3] += wj.w * win[r].w;
            }
        };
        if constexpr (KC > 0) {
#pragma unroll
            for (int j = 0; j < KC; ++j) tap(j);
        } else {
            for (int j = 0; j < k; ++j) tap(j);
        }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (cact) {
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = bias[ch + e];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int t = t0 + tl * 8 + r;
            if (t < Tn) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = acc[r][e] + bv[e]; s1[e] += o[e]; s2[e] += o[e] * o[e]; }
                const size_t row = (size_t)b * Tn + t;
                if (outop == OUT_NONE) {
                    store4(y + row * C + ch, o);
                } else if (outop == OUT_DSWISH) {
                    float a[4];
                    load4g(aux + row * C + ch, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= dswishf_(a[e]);
                    store4(y + row * C + ch, o);
                } else {
                    float a1[4], a2[4], o2[4];
                    load4g(aux + row * 2 * C + ch, a1);
                    load4g(aux + row * 2 * C + C + ch, a2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sg = sigmoidf_(a2[e]);
                        o2[e] = o[e] * a1[e] * sg * (1.f - sg);
                        o[e] *= sg;
                    }
                    store4(y + row * 2 * C + ch, o);
                    store4(y + row * 2 * C + C + ch, o2);
                }
            }
        }
    }
    if (ssum) {   // per-sample channel sums of this tile -> [B,C] (uniform branch)
        __syncthreads();
        float* red = tile;   // [8 tl][128 ch][2]
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(tl * DW_CT + cl * 4 + e) * 2] = s1[e]; red[(tl * DW_CT + cl * 4 + e) * 2 + 1] = s2[e]; }
        __syncthreads();
        if (tid < DW_CT && c0 + tid < C) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { a += red[(r * DW_CT + tid) * 2]; q += red[(r * DW_CT + tid) * 2 + 1]; }
            if (part) {      // deterministic: this time tile's partial row [B][P = gridDim.x][2][C], summed by stats_reduce_kernel
                float* pr = part + (((size_t)b * gridDim.x + blockIdx.x) * 2) * C + c0 + tid;
                pr[0] = a; pr[C] = q;
            } else {
                atomicAdd(ssum + (size_t)b * C + c0 + tid, a);
                if (ssq) atomicAdd(ssq + (size_t)b * C + c0 + tid, q);
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// Register-window variants for the kernel sizes the model uses (K in {3,5,11,15}): a thread
// owns 4 channels x DWR_SEG consecutive time steps of one sample, keeps the last K transformed
// inputs and the K taps in registers (slot indices are compile-time through a K-unrolled body),
// so every input element is loaded and activated once, there is no LDS tile and no barrier.
// -------------------------------------------------------------------------------------
#define DWR_SEG 32
#define DWR_SEG_SMALL 8      // dwconv_reg8_kernel at B <= DW_SMALL_B
#define DW_SMALL_B 8

// raw 4-channel load (value v, GLU gate g) and the input transform, kept separate so that a K-group's loads can all be
// issued before the first one is consumed
template <typename T>
DEVI void dw_raw(const T* __restrict__ x, int b, int tin, int Tn, int C, int Cin, int ch, int inop, float (&v)[4], float (&g)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    g[0] = g[1] = g[2] = g[3] = 0.f;
    if (tin < 0 || tin >= Tn) return;
    const T* p = x + ((size_t)b * Tn + tin) * Cin + ch;
    load4g(p, v);
    if (inop == DWIN_GLU) load4g(p + C, g);
}
DEVI void dw_xform(int inop, float (&v)[4], const float (&g)[4]) {
    if (inop == DWIN_SWISH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = swishf_(v[e]);
    } else if (inop == DWIN_GLU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= sigmoidf_(g[e]);
    }
}
template <typename T>
DEVI void dw_load_in(const T* __restrict__ x, int b, int tin, int Tn, int C, int Cin, int ch, int inop, float (&v)[4]) {
    float g[4];
    dw_raw(x, b, tin, Tn, C, Cin, ch, inop, v, g);
    dw_xform(inop, v, g);
}

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_reg_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, const T* __restrict__ aux,
                                                         float* __restrict__ ssum, float* __restrict__ ssq,
                                                         int Tn, int C, int padl, int inop, int outop, int flip, float* __restrict__ part) {
    // workgroup = 64 channel quads (256 channels: one wave reads 512 contiguous bytes of a row) x 4 consecutive segments of
    // one sample: the per-(sample, channel) statistics of the 4 segments are combined in LDS before the atomics
    __shared__ float sred[4][64][8];
    const int cg = C >> 2;
    const int nseg = (Tn + DWR_SEG - 1) / DWR_SEG;
    const int ncb = (cg + 63) >> 6;
    const int c4 = (blockIdx.x % ncb) * 64 + (threadIdx.x & 63), seg = (blockIdx.x / ncb) * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    const bool live = c4 < cg && seg < nseg;
    const int ch = min(c4, cg - 1) * 4;
    const int Cin = (inop == DWIN_GLU) ? 2 * C : C;
    float wr[K][4], win[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) load4(w + (size_t)(flip ? (K - 1 - j) : j) * C + ch, wr[j]);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) load4(bias + ch, bv);
    const int t0 = seg * DWR_SEG, tend = live ? min(Tn, t0 + DWR_SEG) : t0;
    // slots 0..K-2 hold inputs t0-padl .. t0-padl+K-2
#pragma unroll
    for (int j = 0; j < K - 1; ++j) dw_load_in(x, b, t0 - padl + j, Tn, C, Cin, ch, inop, win[j]);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int LB = K <= 5 ? K : 4;           // input rows whose loads are in flight together (K = 11, 15: batches of 4)
    for (int tb = t0; tb < tend; tb += K) {
#pragma unroll
      for (int u0 = 0; u0 < K; u0 += LB) {
        float nv[LB][4], ng[LB][4];
#pragma unroll
        for (int uu = 0; uu < LB; ++uu) dw_raw(x, b, (u0 + uu < K && tb + u0 + uu < tend) ? tb + u0 + uu - padl + K - 1 : -1, Tn, C, Cin, ch, inop, nv[uu], ng[uu]);
#pragma unroll
        for (int uu = 0; uu < LB; ++uu) {        // output t = tb + u uses slots (u + j) % K ; the new input lands in slot (u + K - 1) % K
            const int u = u0 + uu;
            const int t = tb + u;
            if (u < K && t < tend) {
                dw_xform(inop, nv[uu], ng[uu]);
#pragma unroll
                for (int e = 0; e < 4; ++e) win[(u + K - 1) % K][e] = nv[uu][e];
                float o[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
                for (int j = 0; j < K; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += wr[j][e] * win[(u + j) % K][e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += o[e]; s2[e] += o[e] * o[e]; }
                const size_t row = (size_t)b * Tn + t;
                if (outop == OUT_NONE) {
                    store4(y + row * C + ch, o);
                } else if (outop == OUT_DSWISH) {
                    float a[4];
                    load4g(aux + row * C + ch, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= dswishf_(a[e]);
                    store4(y + row * C + ch, o);
                } else {
                    float a1[4], a2[4], o2[4];
                    load4g(aux + row * 2 * C + ch, a1);
                    load4g(aux + row * 2 * C + C + ch, a2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sg = sigmoidf_(a2[e]);
                        o2[e] = o[e] * a1[e] * sg * (1.f - sg);
                        o[e] *= sg;
                    }
                    store4(y + row * 2 * C + ch, o);
                    store4(y + row * 2 * C + C + ch, o2);
                }
            }
        }
      }
    }
    if (ssum) {
        const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
#pragma unroll
        for (int e = 0; e < 4; ++e) { sred[sl][cl][e] = live ? s1[e] : 0.f; sred[sl][cl][4 + e] = live ? s2[e] : 0.f; }
        __syncthreads();
        if (sl == 0 && c4 < cg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a += sred[r][cl][e]; q += sred[r][cl][4 + e]; }
                if (part) {  // deterministic partial row of this 4-segment group: [B][P = gridDim.x / ncb][2][C]
                    const int P = gridDim.x / ncb, pi = blockIdx.x / ncb;
                    float* pr = part + (((size_t)b * P + pi) * 2) * C + ch + e;
                    pr[0] = a; pr[C] = q;
                } else {
                    atomicAdd(ssum + (size_t)b * C + ch + e, a);
                    if (ssq) atomicAdd(ssq + (size_t)b * C + ch + e, q);
                }
            }
        }
    }
}

// weight gradient: thread = 4 channels; a workgroup = 256/cg (sample, segment) lanes that each loop over many items with
// dw[K][4] (+ dbias) in registers; lanes are combined through LDS and each workgroup writes ONE partial row
// part[blockIdx.x][K+1][C] (summed by reduce_slabs: no same-address atomics).
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_reg_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part,
                                                               int B, int Tn, int C, int padl, int inop) {
    extern __shared__ float red[];           // [(K+1)*C] block accumulator
    const int cg = C >> 2;
    const int lanes = 256 / cg;              // item lanes per workgroup (cg <= 256)
    const int c4 = threadIdx.x % cg, il = threadIdx.x / cg;
    const int ch = c4 * 4;
    const int Cin = (inop == DWIN_GLU) ? 2 * C : C;
    const int nseg = (Tn + DWR_SEG - 1) / DWR_SEG;
    const int nitems = B * nseg;
    for (int q = threadIdx.x; q < (K + 1) * C; q += 256) red[q] = 0.f;
    __syncthreads();
    float acc[K][4], accb[4] = {0.f, 0.f, 0.f, 0.f}, win[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) { acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f; }
    if (il < lanes) {
        for (int item = blockIdx.x * lanes + il; item < nitems; item += gridDim.x * lanes) {
            const int b = item / nseg, t0 = (item % nseg) * DWR_SEG, tend = min(Tn, t0 + DWR_SEG);
#pragma unroll
            for (int j = 0; j < K - 1; ++j) dw_load_in(x, b, t0 - padl + j, Tn, C, Cin, ch, inop, win[j]);
            for (int tb = t0; tb < tend; tb += K) {
                float nv[K][4], ng[K][4], dd[K][4];
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const bool ok = tb + u < tend;
                    dw_raw(x, b, ok ? tb + u - padl + K - 1 : -1, Tn, C, Cin, ch, inop, nv[u], ng[u]);
                    if (ok) load4g(dy + ((size_t)b * Tn + tb + u) * C + ch, dd[u]);
                    else { dd[u][0] = dd[u][1] = dd[u][2] = dd[u][3] = 0.f; }
                }
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const int t = tb + u;
                    if (t < tend) {
                        dw_xform(inop, nv[u], ng[u]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) win[(u + K - 1) % K][e] = nv[u][e];
                        float d[4] = {dd[u][0], dd[u][1], dd[u][2], dd[u][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) accb[e] += d[e];
#pragma unroll
                        for (int j = 0; j < K; ++j) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[j][e] += d[e] * win[(u + j) % K][e];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(red + j * C + ch + e, acc[j][e]);      // LDS atomics across the item lanes
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(red + K * C + ch + e, accb[e]);
    }
    __syncthreads();
    float* dst = part + (size_t)blockIdx.x * (K + 1) * C;
    for (int q = threadIdx.x; q < (K + 1) * C; q += 256) dst[q] = red[q];
}

// weight gradient, LDS tile + per-thread tap accumulators: thread = 4 channels x all K taps x 4 consecutive time steps
// of a 32-step tile; the in(x) window slides through registers, so an item costs 2 LDS reads per K*4 FMAs (the older
// tap-lane mapping below needs 5 reads per 16 FMAs and is LDS-bound).  Accumulators live across the items of the
// workgroup; lanes are combined with LDS atomics and one partial row per workgroup goes to `part`.
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_win_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part,
                                                               int B, int Tn, int C, int padl, int inop) {
    __shared__ __attribute__((aligned(16))) float xt[(DWG_TT + K - 1) * DW_CT];
    __shared__ __attribute__((aligned(16))) float dt_[DWG_TT * DW_CT];
    __shared__ float red[(K + 1) * DW_CT];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * DW_CT;
    const int Cin = (inop == DWIN_GLU) ? 2 * C : C;
    constexpr int rows = DWG_TT + K - 1;
    const int ntt = (Tn + DWG_TT - 1) / DWG_TT;
    const int cl = tid & 31, tl = tid >> 5;          // 32 channel lanes x 8 time lanes (4 steps each)
    for (int q = tid; q < (K + 1) * DW_CT; q += 256) red[q] = 0.f;
    float acc[K][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
    const int ch8 = c0 + (tid & 15) * 8;
    for (int item = blockIdx.y; item < B * ntt; item += gridDim.y) {
        const int b = item / ntt, t0 = (item % ntt) * DWG_TT;
        __syncthreads();
        for (int r = tid >> 4; r < rows; r += 16) {
            const int tin = t0 - padl + r;
            float v[8];
            if (tin >= 0 && tin < Tn && ch8 < C) {
                const T* p = x + ((size_t)b * Tn + tin) * Cin + ch8;
                load8(p, v);
                if (inop == DWIN_SWISH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
                } else if (inop == DWIN_GLU) {
                    float g[8];
                    load8(p + C, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= sigmoidf_(g[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            float* dst = xt + r * DW_CT + (tid & 15) * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        for (int r = tid >> 4; r < DWG_TT; r += 16) {
            const int t = t0 + r;
            float v[8];
            if (t < Tn && ch8 < C) load8(dy + ((size_t)b * Tn + t) * C + ch8, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            float* dst = dt_ + r * DW_CT + (tid & 15) * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        __syncthreads();
        float4 win[K];                                   // in(x) rows tl*4 + (0..K-1)
#pragma unroll
        for (int j = 0; j < K - 1; ++j) win[j + 1] = *reinterpret_cast<const float4*>(xt + (tl * 4 + j) * DW_CT + cl * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < K - 1; ++j) win[j] = win[j + 1];
            win[K - 1] = *reinterpret_cast<const float4*>(xt + (tl * 4 + u + K - 1) * DW_CT + cl * 4);
            const float4 d = *reinterpret_cast<const float4*>(dt_ + (tl * 4 + u) * DW_CT + cl * 4);
            accb[0] += d.x; accb[1] += d.y; accb[2] += d.z; accb[3] += d.w;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                acc[j][0] += d.x * win[j].x; acc[j][1] += d.y * win[j].y; acc[j][2] += d.z * win[j].z; acc[j][3] += d.w * win[j].w;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(red + j * DW_CT + cl * 4 + e, acc[j][e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(red + K * DW_CT + cl * 4 + e, accb[e]);
    __syncthreads();
    // part[(blockIdx.y)][K+1][C]
    float* dst = part + (size_t)blockIdx.y * (K + 1) * C;
    for (int q = tid; q < (K + 1) * DW_CT; q += 256) {
        const int j = q / DW_CT, c = c0 + (q % DW_CT);
        if (c < C) dst[(size_t)j * C + c] = red[q];
    }
}

// ---- forward only, 8 channels per lane (16-byte accesses: a wave reads / writes 1 KB of a row per instruction — the 8-byte kernel
// above tops out near 3.4 TB/s, tools/micro/load_pattern.hip / store_pattern.hip).  K = 3, 5; C % 8 == 0 with C / 8 a divisor of 256.
// Workgroup = C/8 lanes x (256 / (C/8)) consecutive 32-step segments of one sample; statistics as in dwconv_reg_kernel.
template <typename T>
DEVI void dw8_raw(const T* __restrict__ x, int b, int tin, int Tn, int C, int Cin, int ch, int inop, float (&v)[8], float (&g)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = 0.f; g[e] = 0.f; }
    if (tin < 0 || tin >= Tn) return;
    const T* p = x + ((size_t)b * Tn + tin) * Cin + ch;
    load8(p, v);
    if (inop == DWIN_GLU) load8(p + C, g);
}
template <typename T, int K, int INOP>
__global__ __launch_bounds__(256) void dwconv_reg8_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          T* __restrict__ y, int Tn, int C, int padl, float* __restrict__ part, int seglen) {
    constexpr int inop = INOP;
    extern __shared__ float sred8[];            // [segments per workgroup][C/8][16]
    const int cg = C >> 3, spw = 256 / cg;
    const int nseg = (Tn + seglen - 1) / seglen;      // seglen: DWR_SEG, or DWR_SEG_SMALL for a handful of samples (more, shorter chains)
    const int cl = threadIdx.x % cg, sl = threadIdx.x / cg;
    const int seg = blockIdx.x * spw + sl, b = blockIdx.y;
    const bool live = seg < nseg;
    const int ch = cl * 8;
    const int Cin = (inop == DWIN_GLU) ? 2 * C : C;
    float wr[K][8], win[K][8];
#pragma unroll
    for (int j = 0; j < K; ++j) { load4(w + (size_t)j * C + ch, *reinterpret_cast<float(*)[4]>(&wr[j][0])); load4(w + (size_t)j * C + ch + 4, *reinterpret_cast<float(*)[4]>(&wr[j][4])); }
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias) { load4(bias + ch, *reinterpret_cast<float(*)[4]>(&bv[0])); load4(bias + ch + 4, *reinterpret_cast<float(*)[4]>(&bv[4])); }
    const int t0 = seg * seglen, tend = live ? min(Tn, t0 + seglen) : t0;
    auto xform = [&](float (&v)[8], const float (&g)[8]) {
        if (inop == DWIN_SWISH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
        } else if (inop == DWIN_GLU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= sigmoidf_(g[e]);
        }
    };
#pragma unroll
    for (int j = 0; j < K - 1; ++j) { float g8[8]; dw8_raw(x, b, live ? t0 - padl + j : -1, Tn, C, Cin, ch, inop, win[j], g8); xform(win[j], g8); }
    float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Two groups of K input rows in flight (ping-pong register sets of RAW 16-byte rows): with one group the kernel is bound by
    // memory latency x bytes in flight (12 waves per CU x 64 lanes x K x 16 B: 3.7 TB/s by Little's law, as measured)
    typedef __attribute__((ext_vector_type(4))) uint32_t raw4;
    struct Grp { raw4 v[K], g[INOP == DWIN_GLU ? K : 1]; };
    Grp ga, gb;
    auto issue = [&](Grp& G, int tb) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int tin = tb + u - padl + K - 1;
            G.v[u] = raw4{0u, 0u, 0u, 0u};
            if (INOP == DWIN_GLU) G.g[u] = raw4{0u, 0u, 0u, 0u};
            if (tb + u < tend && tin >= 0 && tin < Tn) {
                const T* p = x + ((size_t)b * Tn + tin) * Cin + ch;
                G.v[u] = *reinterpret_cast<const raw4*>(p);
                if (INOP == DWIN_GLU) G.g[u] = *reinterpret_cast<const raw4*>(p + C);
            }
        }
    };
    auto unpack = [&](const raw4& r, float (&v)[8]) {
        T tmp[8];
        *reinterpret_cast<raw4*>(tmp) = r;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = to_f(tmp[e]);
    };
    auto process = [&](const Grp& G, int tb) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int t = tb + u;
            if (t < tend) {
                float nv[8], ng[8];
                unpack(G.v[u], nv);
                if (INOP == DWIN_GLU) unpack(G.g[INOP == DWIN_GLU ? u : 0], ng);
                xform(nv, ng);
#pragma unroll
                for (int e = 0; e < 8; ++e) win[(u + K - 1) % K][e] = nv[e];
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = bv[e];
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += wr[j][e] * win[(u + j) % K][e];
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += o[e]; s2[e] += o[e] * o[e]; }
                store8(y + ((size_t)b * Tn + t) * C + ch, o);
            }
        }
    };
    issue(ga, t0);
    for (int tb = t0; tb < tend; tb += 2 * K) {
        issue(gb, tb + K);
        process(ga, tb);
        issue(ga, tb + 2 * K);
        process(gb, tb + K);
    }
    if (part) {     // deterministic partial row of this workgroup's segments: part[B][P = gridDim.x][2][C]
        float* sr = sred8 + ((size_t)sl * cg + cl) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sr[e] = live ? s1[e] : 0.f; sr[8 + e] = live ? s2[e] : 0.f; }
        __syncthreads();
        if (sl == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = 0.f, q = 0.f;
                for (int r = 0; r < spw; ++r) { a += sred8[((size_t)r * cg + cl) * 16 + e]; q += sred8[((size_t)r * cg + cl) * 16 + 8 + e]; }
                float* pr = part + (((size_t)b * gridDim.x + blockIdx.x) * 2) * C + ch + e;
                pr[0] = a; pr[C] = q;
            }
        }
    }
}

// ---- forward, K = 11 / 15 (the first Conv1DBlock of a group, the transformer blocks' conv modules): a STREAMING LDS kernel.
// The 64 x 128 tile kernel above re-stages a (64 + K - 1)-row tile per workgroup (23 % halo at K = 15), runs load -> barrier -> compute -> store
// strictly in sequence inside a workgroup, and stores 8 bytes per lane: 2.6 TB/s at K = 11.  Here a workgroup owns 128 channels of ONE sample for
// a whole time range and walks it in 32-row chunks through a 64-row LDS ring of TRANSFORMED inputs (fp32, Swish / GLU applied once per element):
//   * no halo re-reads inside a range, the tap weights (K x 4 channels per thread) are loaded once per workgroup;
//   * the next chunk's global loads are in flight while the current chunk is computed (two barriers per chunk);
//   * (channel pairs as float2 / v_pk_fma_f32 were tried: forward depthwise conv 0.897 -> 1.087 ms/step — scalar FMAs stay)
//   * thread = 4 channels x 4 consecutive rows, row-stationary accumulation (every ring row is read once per thread and feeds up to four outputs);
//   * lane pairs swap halves (DPP quad_perm) so that every store is 16 bytes: an even lane writes 8 channels of rows 0 / 2, an odd lane of rows 1 / 3;
//   * BatchNorm / GAP statistics: per-thread sums over the whole range, one partial row per workgroup at the end (deterministic, as before).
// Non-causal convolutions (padl < K - 1) compute output row t when input row t + K - 1 - padl is in the ring: the output window lags the input
// window by `lag` rows and one more (input-free) chunk drains it.
template <typename T, int K, int INOP>
__global__ __launch_bounds__(256, 3) void dwconv_stream_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                             T* __restrict__ y, int Tn, int C, int padl, float* __restrict__ part, int tsplit) {
    constexpr int RB = 64, CH = 32, CT = 128;           // ring rows, chunk rows, channels per workgroup
    __shared__ __attribute__((aligned(16))) float ring[RB * CT];
    __shared__ float sred[8][2][CT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int c0 = blockIdx.x * CT, b = blockIdx.y;
    const int Cin = (INOP == DWIN_GLU) ? 2 * C : C;
    const int lag = K - 1 - padl;
    // time range of this workgroup (tsplit ranges per sample, multiples of CH rows)
    const int per = ((Tn + tsplit - 1) / tsplit + CH - 1) / CH * CH;
    const int r_beg = blockIdx.z * per, r_end = min(Tn, r_beg + per);
    if (r_beg >= Tn) return;
    // ---- compute mapping: 32 channel quads x 8 row groups of 4
    const int cq = tid & 31, rg = tid >> 5;
    const int ch = c0 + cq * 4;
    float wr[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) load4(w + (size_t)j * C + ch, wr[j]);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) load4(bias + ch, bv);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // ---- staging mapping: 16 chunks of 8 channels x 16 rows, two row sets per chunk of 32 rows
    const int sc = tid & 15, sr = tid >> 4;
    typedef __attribute__((ext_vector_type(4))) uint32_t raw4;
    raw4 rv[2], rgl[2];
    auto gload = [&](int t0, bool hi_only = false) {           // input rows t0 .. t0+31 (zero outside [0, Tn); hi_only: rows t0+16 .. only — the K - 1 <= 14 history rows)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + sr + 16 * h;
            rv[h] = raw4{0u, 0u, 0u, 0u}; rgl[h] = raw4{0u, 0u, 0u, 0u};
            if (t >= 0 && t < Tn && (h == 1 || !hi_only)) {
                const T* p = x + ((size_t)b * Tn + t) * Cin + c0 + sc * 8;
                rv[h] = *reinterpret_cast<const raw4*>(p);
                if (INOP == DWIN_GLU) rgl[h] = *reinterpret_cast<const raw4*>(p + C);
            }
        }
    };
    auto lstore = [&](int t0) {          // transform once, fp32 into the ring slot of the row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t0 + sr + 16 * h;
            float v[8];
            { T tmp[8]; *reinterpret_cast<raw4*>(tmp) = rv[h];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = to_f(tmp[e]); }
            if (INOP == DWIN_SWISH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
            } else if (INOP == DWIN_GLU) {
                T tg[8]; *reinterpret_cast<raw4*>(tg) = rgl[h];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= sigmoidf_(to_f(tg[e]));
            }
            if (t < 0 || t >= Tn) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            float* dst = ring + ((t & (RB - 1)) * CT) + sc * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    };
    // prologue: the history rows [in0 - CH, in0) (zeros before the sample / real rows when a range starts inside it) and the first chunk
    const int in0 = (r_beg + lag) / CH * CH;            // output row t is computed by the chunk that brings input row t + lag
    gload(in0 - CH, true); lstore(in0 - CH);
    gload(in0); lstore(in0);
    __syncthreads();
    // chunk i: inputs [ti, ti + CH) are in the ring; outputs [ti - lag, ti + CH - lag) clipped to [r_beg, r_end)
    for (int ti = in0; ti - lag < r_end; ti += CH) {
        const bool more = ti + CH - lag < r_end;
        if (more) gload(ti + CH);
        const int o0 = ti - lag + rg * 4;                 // first output row of this thread
        float acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][e] = 0.f;
        // input rows o0 - padl .. o0 - padl + K + 2
#pragma unroll
        for (int qd = 0; qd < K + 3; ++qd) {
            const int tin = o0 - padl + qd;
            const float4 rw = *reinterpret_cast<const float4*>(ring + ((tin & (RB - 1)) * CT) + cq * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = qd - u;
                if (j >= 0 && j < K) { acc[u][0] += wr[j][0] * rw.x; acc[u][1] += wr[j][1] * rw.y; acc[u][2] += wr[j][2] * rw.z; acc[u][3] += wr[j][3] * rw.w; }
            }
        }
        // pack, swap halves inside lane pairs, 16-byte stores
        uint32_t pk[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = o0 + u >= r_beg && o0 + u < r_end;
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[u][e] += bv[e]; if (live) { s1[e] += acc[u][e]; s2[e] += acc[u][e] * acc[u][e]; } }
            T tmp[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) tmp[e] = from_f<T>(acc[u][e]);
            pk[u][0] = reinterpret_cast<const uint32_t*>(tmp)[0]; pk[u][1] = reinterpret_cast<const uint32_t*>(tmp)[1];
        }
        const bool odd = lane & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // even lane keeps rows 2h (own low half | partner's half), odd lane rows 2h + 1 (partner's half | own high half)
            const int mine = 2 * h + (odd ? 1 : 0), theirs = 2 * h + (odd ? 0 : 1);
            uint32_t give0 = pk[theirs][0], give1 = pk[theirs][1];
            const uint32_t got0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give0, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
            const uint32_t got1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give1, 0xB1, 0xf, 0xf, true);
            const int t = o0 + mine;
            if (t >= r_beg && t < r_end) {
                raw4 o = odd ? raw4{got0, got1, pk[mine][0], pk[mine][1]} : raw4{pk[mine][0], pk[mine][1], got0, got1};
                *reinterpret_cast<raw4*>(y + ((size_t)b * Tn + t) * C + c0 + (cq >> 1) * 8) = o;
            }
        }
        __syncthreads();                 // every thread is done reading the rows the next chunk overwrites
        if (more) lstore(ti + CH);
        __syncthreads();
    }
    if (part) {      // one partial row per (sample, time range): part[B][P = tsplit][2][C]
        sred[rg][0][cq * 4 + 0] = s1[0]; sred[rg][0][cq * 4 + 1] = s1[1]; sred[rg][0][cq * 4 + 2] = s1[2]; sred[rg][0][cq * 4 + 3] = s1[3];
        sred[rg][1][cq * 4 + 0] = s2[0]; sred[rg][1][cq * 4 + 1] = s2[1]; sred[rg][1][cq * 4 + 2] = s2[2]; sred[rg][1][cq * 4 + 3] = s2[3];
        __syncthreads();
        if (tid < CT) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { a += sred[r][0][tid]; q += sred[r][1][tid]; }
            float* pr = part + (((size_t)b * gridDim.z + blockIdx.z) * 2) * C + c0 + tid;
            pr[0] = a; pr[C] = q;
        }
    }
}
static bool dw_stream_ok(int dt, int C, int k, int T) { return dt != DT_F32 && (k == 11 || k == 15) && C % 128 == 0 && T >= 64; }
template <typename T>
static int launch_dw_stream(int k, const T* x, const float* w, const float* bias, T* y, int B, int Tn, int C, int padl, int inop, float* part, hipStream_t s) {
    // enough workgroups for ~3 rounds of the chip's 768-1024 slots, whole 32-row chunks per range
    int tsplit = 1;
    while ((C / 128) * B * tsplit < 2048 && Tn / (tsplit * 2) >= 128) tsplit *= 2;      // every extra range re-reads 16 history rows
    const dim3 grid(C / 128, B, tsplit);
#define DWS(KK, OP) hipLaunchKernelGGL((dwconv_stream_kernel<T, KK, OP>), grid, dim3(256), 0, s, x, w, bias, y, Tn, C, padl, part, tsplit)
#define DWSK(OP) do { if (k == 11) DWS(11, OP); else DWS(15, OP); } while (0)
    if (inop == DWIN_SWISH) DWSK(DWIN_SWISH); else if (inop == DWIN_GLU) DWSK(DWIN_GLU); else DWSK(DWIN_NONE);
#undef DWSK
#undef DWS
    return tsplit;      // partial rows per sample
}

// applicable: 16-bit storage, K in {3, 5}, C/8 in {32, 64, 128, 256}
static bool dw_reg8_ok(int dt, int C, int k) { return dt != DT_F32 && (k == 3 || k == 5) && C % 8 == 0 && C / 8 >= 32 && C / 8 <= 256 && 256 % (C / 8) == 0; }
template <typename T>
static int launch_dw_reg8(int k, const T* x, const float* w, const float* bias, T* y, int B, int Tn, int C, int padl, int inop, float* part, hipStream_t s) {
    // a handful of samples (B = 1 inference): 8-step segments — four times the workgroups, a quarter of the dependent steps per thread
    // (3 workgroups of 32-step chains took 17 us at T = 384)
    const int seglen = B <= DW_SMALL_B ? DWR_SEG_SMALL : DWR_SEG;
    const int cg = C / 8, spw = 256 / cg, nseg = (Tn + seglen - 1) / seglen;
    const dim3 grid((nseg + spw - 1) / spw, B);
    const size_t sh = (size_t)256 * 16 * sizeof(float);
#define DW8(KK, OP) hipLaunchKernelGGL((dwconv_reg8_kernel<T, KK, OP>), grid, dim3(256), sh, s, x, w, bias, y, Tn, C, padl, part, seglen)
#define DW8K(OP) do { if (k == 3) DW8(3, OP); else DW8(5, OP); } while (0)
    if (inop == DWIN_SWISH) DW8K(DWIN_SWISH); else if (inop == DWIN_GLU) DW8K(DWIN_GLU); else DW8K(DWIN_NONE);
#undef DW8K
#undef DW8
    return (int)grid.x;      // partial rows per sample
}

static bool dw_reg_ok(int C, int k) { return (k == 3 || k == 5) && C % 4 == 0   /* K = 11, 15: the register window (201 / 233 VGPRs) measured 94 / 103 us vs 82 / 86 us for the LDS-tiled kernel */ && C / 4 <= 256 && 256 % (C / 4) == 0; }
int g_force_dw_lds = 0;    // tests: force the LDS-tiled kernels

template <typename T>
static void launch_dw_reg(int k, const T* x, const float* w, const float* bias, T* y, const T* aux, float* ssum, float* ssq,
                          int B, int Tn, int C, int padl, int inop, int outop, int flip, hipStream_t s, float* part = nullptr) {
    const int nseg = (Tn + DWR_SEG - 1) / DWR_SEG;
    dim3 grid(((C / 4 + 63) / 64) * ((nseg + 3) / 4), B);
#define DWR(KK) hipLaunchKernelGGL((dwconv_reg_kernel<T, KK>), grid, dim3(256), 0, s, x, w, bias, y, aux, ssum, ssq, Tn, C, padl, inop, outop, flip, part)
    switch (k) { case 3: DWR(3); break; case 5: DWR(5); break; case 11: DWR(11); break; default: DWR(15); break; }
#undef DWR
}
#define DWG_BLOCKS 512
template <typename T>
static void launch_dw_wgrad_reg(int k, const T* dy, const T* x, float* part, int B, int Tn, int C, int padl, int inop, hipStream_t s) {
    const size_t sh = (size_t)(k + 1) * C * sizeof(float);
#define DWG(KK) hipLaunchKernelGGL((dwconv_wgrad_reg_kernel<T, KK>), dim3(DWG_BLOCKS), dim3(256), sh, s, dy, x, part, B, Tn, C, padl, inop)
    switch (k) { case 3: DWG(3); break; case 5: DWG(5); break; case 11: DWG(11); break; default: DWG(15); break; }
#undef DWG
}
size_t dwconv_bwd_scratch_floats(int C, int k) { return (size_t)DWG_BLOCKS * (k + 1) * C; }

static int dwconv_check(int C, int k) {
    if (C % 8 != 0) { ishara_set_error("dwconv: C=%d must be a multiple of 8", C); return -1; }
    if (k < 1 || k > DW_MAXK) { ishara_set_error("dwconv: kernel size %d unsupported (1..%d)", k, DW_MAXK); return -1; }
    return 0;
}

// ssum / ssq [B, C] = per-sample channel sums of the partial rows part[B][P][2][C], in a fixed order (no float atomics:
// the forward pass is bit-reproducible run to run, and no zero-fill launches are needed)
__global__ __launch_bounds__(256) void stats_reduce_kernel(const float* __restrict__ part, int P, float* __restrict__ ssum, float* __restrict__ ssq, int B, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float* p = part + ((size_t)b * P * 2) * C + c;
    float a = 0.f, q = 0.f;
    for (int r = 0; r < P; ++r) { a += p[(size_t)(2 * r) * C]; q += p[(size_t)(2 * r + 1) * C]; }
    ssum[i] = a;
    if (ssq) ssq[i] = q;
}
#define DWK_LAUNCH(TT, ...)                                                                                              \
    do {                                                                                                                \
        if (k == 11) hipLaunchKernelGGL((dwconv_kernel<TT, 11, 11>), grid, dim3(256), 0, s, __VA_ARGS__);   /* 74 -> 65 us */  \
        else if (k <= 15) hipLaunchKernelGGL((dwconv_kernel<TT, 0, 15>), grid, dim3(256), 0, s, __VA_ARGS__);   /* K = 15 unrolled: 86 vs 78 us */ \
        else hipLaunchKernelGGL((dwconv_kernel<TT, 0, DW_MAXK>), grid, dim3(256), 0, s, __VA_ARGS__);            \
    } while (0)
size_t dwconv_fwd_scratch_floats(int B, int T, int C) {
    const size_t big = (size_t)B * ((T + DWR_SEG - 1) / DWR_SEG), small = (size_t)(B < DW_SMALL_B ? B : DW_SMALL_B) * ((T + DWR_SEG_SMALL - 1) / DWR_SEG_SMALL);
    return (big > small ? big : small) * 2 * C;
}   // the most partial rows any forward kernel writes per sample (one per 32-step segment)

// `part`: scratch of dwconv_fwd_scratch_floats(B, T, C) floats for the deterministic statistics, or nullptr (then colsum /
// colsq must be zero-filled by the caller and are accumulated with float atomics)
int launch_dwconv_fwd(int dt, int inop, const void* x, const float* w, const float* bias, void* y,
                      float* colsum, float* colsq, float* part, int B, int T, int C, int k, int padl, hipStream_t s, int* part_rows) {
    if (dwconv_check(C, k)) return -1;
    if (!colsum) part = nullptr;
    int P;
    static const bool no_stream = getenv("ISHARA_NO_DW_STREAM") != nullptr;      // A/B switch: the 64 x 128 tile kernel instead
    if (dw_stream_ok(dt, C, k, T) && B > DW_SMALL_B && !g_force_dw_lds && !no_stream && (part || !colsum)) {
        if (dt == DT_BF16) P = launch_dw_stream<bf16>(k, (const bf16*)x, w, bias, (bf16*)y, B, T, C, padl, inop, colsum ? part : nullptr, s);
        else P = launch_dw_stream<f16>(k, (const f16*)x, w, bias, (f16*)y, B, T, C, padl, inop, colsum ? part : nullptr, s);
    } else if (dw_reg8_ok(dt, C, k) && !g_force_dw_lds && (part || !colsum)) {          // statistics only through the deterministic partial rows
        if (dt == DT_BF16) P = launch_dw_reg8<bf16>(k, (const bf16*)x, w, bias, (bf16*)y, B, T, C, padl, inop, colsum ? part : nullptr, s);
        else P = launch_dw_reg8<f16>(k, (const f16*)x, w, bias, (f16*)y, B, T, C, padl, inop, colsum ? part : nullptr, s);
    } else if (dw_reg_ok(C, k) && !g_force_dw_lds) {
        const int nseg = (T + DWR_SEG - 1) / DWR_SEG;
        P = (nseg + 3) / 4;
        if (dt == DT_BF16) launch_dw_reg<bf16>(k, (const bf16*)x, w, bias, (bf16*)y, (const bf16*)nullptr, colsum, colsq, B, T, C, padl, inop, OUT_NONE, 0, s, part);
        else if (dt == DT_F16) launch_dw_reg<f16>(k, (const f16*)x, w, bias, (f16*)y, (const f16*)nullptr, colsum, colsq, B, T, C, padl, inop, OUT_NONE, 0, s, part);
        else launch_dw_reg<float>(k, (const float*)x, w, bias, (float*)y, (const float*)nullptr, colsum, colsq, B, T, C, padl, inop, OUT_NONE, 0, s, part);
    } else {
        dim3 grid((T + DW_TT - 1) / DW_TT, (C + DW_CT - 1) / DW_CT, B);
        P = grid.x;
        if (dt == DT_BF16) DWK_LAUNCH(bf16, (const bf16*)x, w, bias, (bf16*)y, (const bf16*)nullptr, colsum, colsq, B, T, C, k, padl, inop, (int)OUT_NONE, 0, part);
        else if (dt == DT_F16) DWK_LAUNCH(f16, (const f16*)x, w, bias, (f16*)y, (const f16*)nullptr, colsum, colsq, B, T, C, k, padl, inop, (int)OUT_NONE, 0, part);
        else DWK_LAUNCH(float, (const float*)x, w, bias, (float*)y, (const float*)nullptr, colsum, colsq, B, T, C, k, padl, inop, (int)OUT_NONE, 0, part);
    }
    if (part && part_rows) *part_rows = P;          // the caller sums the partial rows itself (eca_fwd's inference form)
    else if (part) hipLaunchKernelGGL(stats_reduce_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, part, P, colsum, colsq, B, C);
    return LAUNCH_OK();
}

// dw[j,c] += sum_{b,t} dy[b,t,c] * in(x)[b, t-padl+j, c] ; dbias[c] += sum dy.
// grid = (C/128, splits); each workgroup loops over (sample, time-tile) pairs and issues
// one atomic per (tap, channel) at the end.  thread -> 4 channels x taps {tl, tl+8, tl+16, tl+24}.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           float* __restrict__ dw, float* __restrict__ dbias,
                                                           int B, int Tn, int C, int k, int padl, int inop) {
    __shared__ __attribute__((aligned(16))) float xt[(DWG_TT + DW_MAXK - 1) * DW_CT];
    __shared__ __attribute__((aligned(16))) float dt_[DWG_TT * DW_CT];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * DW_CT;
    const int Cin = (inop == DWIN_GLU) ? 2 * C : C;
    const int rows = DWG_TT + k - 1;
    const int ntt = (Tn + DWG_TT - 1) / DWG_TT;
    const int cl = tid & 31, tl = tid >> 5;
    float acc[4][4], accb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;
    // staging registers: x tile rows (tid>>4)+16*it (it < NX), dy tile rows (tid>>4)+16*it (it < 2); the next item's
    // loads are issued before the current item's compute and written to LDS after it (one LDS image, two barriers/item)
    constexpr int NX = (DWG_TT + DW_MAXK - 1 + 15) / 16;
    float xv[NX][8], xg[NX][8], dv[2][8];
    bool xok[NX], dok[2];
    const int ch8 = c0 + (tid & 15) * 8;
    auto gload = [&](int item) {
        const int b = item / ntt, t0 = (item % ntt) * DWG_TT;
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int r = (tid >> 4) + 16 * it;
            const int tin = t0 - padl + r;
            xok[it] = r < rows && tin >= 0 && tin < Tn && ch8 < C;
            if (xok[it]) {
                const T* p = x + ((size_t)b * Tn + tin) * Cin + ch8;
                load8(p, xv[it]);
                if (inop == DWIN_GLU) load8(p + C, xg[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = t0 + (tid >> 4) + 16 * it;
            dok[it] = t < Tn && ch8 < C;
            if (dok[it]) load8(dy + ((size_t)b * Tn + t) * C + ch8, dv[it]);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (r < rows) {
                if (xok[it]) {
                    if (inop == DWIN_SWISH) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xv[it][e] = swishf_(xv[it][e]);
                    } else if (inop == DWIN_GLU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xv[it][e] *= sigmoidf_(xg[it][e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv[it][e] = 0.f;
                }
                float* dst = xt + r * DW_CT + (tid & 15) * 8;
                *reinterpret_cast<float4*>(dst) = make_float4(xv[it][0], xv[it][1], xv[it][2], xv[it][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(xv[it][4], xv[it][5], xv[it][6], xv[it][7]);
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = (tid >> 4) + 16 * it;
            if (!dok[it]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dv[it][e] = 0.f;
            }
            float* dst = dt_ + r * DW_CT + (tid & 15) * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(dv[it][0], dv[it][1], dv[it][2], dv[it][3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(dv[it][4], dv[it][5], dv[it][6], dv[it][7]);
        }
    };
    const int nitems = B * ntt;
    int item = blockIdx.y;
    if (item < nitems) gload(item);
    for (; item < nitems; item += gridDim.y) {
        __syncthreads();                       // previous item's compute is done with the LDS image
        lstore();
        __syncthreads();
        const int nxt = item + gridDim.y;
        if (nxt < nitems) gload(nxt);          // in flight during the compute below
        for (int t = 0; t < DWG_TT; ++t) {
            const float4 d = *reinterpret_cast<const float4*>(dt_ + t * DW_CT + cl * 4);
            if (tl == 0) { accb[0] += d.x; accb[1] += d.y; accb[2] += d.z; accb[3] += d.w; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = tl + 8 * q;
                if (j < k) {
                    const float4 xq = *reinterpret_cast<const float4*>(xt + (t + j) * DW_CT + cl * 4);
                    acc[q][0] += d.x * xq.x; acc[q][1] += d.y * xq.y; acc[q][2] += d.z * xq.z; acc[q][3] += d.w * xq.w;
                }
            }
        }
    }
    const int ch = c0 + cl * 4;
    if (ch < C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = tl + 8 * q;
            if (j < k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dw + (size_t)j * C + ch + e, acc[q][e]);
            }
        }
        if (dbias && tl == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dbias + ch + e, accb[e]);
        }
    }
}

int launch_dwconv_bwd_bn(int dt, int inop, const void* dy, const DwBnArgs& bn, const void* x, const float* w, void* dx,
                         float* dw, float* dbias, float* scratch, int B, int T, int C, int k, int padl, hipStream_t s) {
    if (dwconv_check(C, k)) return -1;
    if (!bn.h || !scratch || g_force_dw_lds || !dwconv_bwd_fused_ok(dt, C, k, padl) || k >= 15) return 0;      // k = 15: the 15-row window + the BatchNorm coefficients spill (26+ VGPRs)
    const int rows = launch_dwconv_bwd_fused(dt, inop, dy, x, w, dx, scratch, B, T, C, k, padl, DWG_BLOCKS, s, bn);
    if (rows < 0) return -2;
    launch_reduce_slabs(scratch, dw, k * C, rows, (size_t)(k + 1) * C, s);
    if (dbias) launch_reduce_slabs(scratch + (size_t)k * C, dbias, C, rows, (size_t)(k + 1) * C, s);
    return hipGetLastError() == hipSuccess ? 1 : -2;
}

int launch_dwconv_bwd(int dt, int inop, const void* dy, const void* x, const float* w, void* dx,
                      float* dw, float* dbias, float* scratch, int B, int T, int C, int k, int padl, hipStream_t s) {
    if (dwconv_check(C, k)) return -1;
    if (scratch && !g_force_dw_lds && dwconv_bwd_fused_ok(dt, C, k, padl)) {
        // one pass: dx and the per-workgroup partial rows of (dw, dbias), then the row sum
        const int rows = launch_dwconv_bwd_fused(dt, inop, dy, x, w, dx, scratch, B, T, C, k, padl, DWG_BLOCKS, s, DwBnArgs());
        if (rows < 0) return -2;
        launch_reduce_slabs(scratch, dw, k * C, rows, (size_t)(k + 1) * C, s);
        if (dbias) launch_reduce_slabs(scratch + (size_t)k * C, dbias, C, rows, (size_t)(k + 1) * C, s);
        return LAUNCH_OK();
    }
    const int outop = inop == DWIN_SWISH ? OUT_DSWISH : (inop == DWIN_GLU ? OUT_DGLU : OUT_NONE);
    const bool reg = dw_reg_ok(C, k) && !g_force_dw_lds;
    // data grad: correlation with flipped taps, left pad k-1-padl; then through the input op
    if (reg) {
        if (dt == DT_BF16) launch_dw_reg<bf16>(k, (const bf16*)dy, w, nullptr, (bf16*)dx, (const bf16*)x, nullptr, nullptr, B, T, C, k - 1 - padl, DWIN_NONE, outop, 1, s);
        else launch_dw_reg<float>(k, (const float*)dy, w, nullptr, (float*)dx, (const float*)x, nullptr, nullptr, B, T, C, k - 1 - padl, DWIN_NONE, outop, 1, s);
    } else {
        dim3 grid((T + DW_TT - 1) / DW_TT, (C + DW_CT - 1) / DW_CT, B);
        if (dt == DT_BF16) DWK_LAUNCH(bf16, (const bf16*)dy, w, (const float*)nullptr, (bf16*)dx, (const bf16*)x, (float*)nullptr, (float*)nullptr, B, T, C, k, k - 1 - padl, (int)DWIN_NONE, outop, 1, (float*)nullptr);
        else DWK_LAUNCH(float, (const float*)dy, w, (const float*)nullptr, (float*)dx, (const float*)x, (float*)nullptr, (float*)nullptr, B, T, C, k, k - 1 - padl, (int)DWIN_NONE, outop, 1, (float*)nullptr);
    }
    const bool winok = scratch && (k == 3 || k == 5 || k == 11 || k == 15) && !g_force_dw_lds;
    if (winok) {
        const int cblocks = (C + DW_CT - 1) / DW_CT;
        const int ntt = (T + DWG_TT - 1) / DWG_TT;
        const int splits = max(1, min(B * ntt, DWG_BLOCKS / cblocks));
        dim3 grid(cblocks, splits);
#define DWW(TT, KK) hipLaunchKernelGGL((dwconv_wgrad_win_kernel<TT, KK>), grid, dim3(256), 0, s, (const TT*)dy, (const TT*)x, scratch, B, T, C, padl, inop)
#define DWWK(TT) switch (k) { case 3: DWW(TT, 3); break; case 5: DWW(TT, 5); break; case 11: DWW(TT, 11); break; default: DWW(TT, 15); break; }
        if (dt == DT_BF16) { DWWK(bf16) } else { DWWK(float) }
        launch_reduce_slabs(scratch, dw, k * C, splits, (size_t)(k + 1) * C, s);
        if (dbias) launch_reduce_slabs(scratch + (size_t)k * C, dbias, C, splits, (size_t)(k + 1) * C, s);
    } else if (reg && scratch) {   // weight / bias grad through per-workgroup partial rows
        if (dt == DT_BF16) launch_dw_wgrad_reg<bf16>(k, (const bf16*)dy, (const bf16*)x, scratch, B, T, C, padl, inop, s);
        else launch_dw_wgrad_reg<float>(k, (const float*)dy, (const float*)x, scratch, B, T, C, padl, inop, s);
        launch_reduce_slabs(scratch, dw, k * C, DWG_BLOCKS, (size_t)(k + 1) * C, s);
        if (dbias) launch_reduce_slabs(scratch + (size_t)k * C, dbias, C, DWG_BLOCKS, (size_t)(k + 1) * C, s);
    } else {
        const int ntt = (T + DWG_TT - 1) / DWG_TT;
        const int cblocks = (C + DW_CT - 1) / DW_CT;
        int splits = max(1, min(B * ntt, 1024 / cblocks));
        dim3 grid(cblocks, splits);
        if (dt == DT_BF16) hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, dw, dbias, B, T, C, k, padl, inop);
        else hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)x, dw, dbias, B, T, C, k, padl, inop);
    }
    return LAUNCH_OK();
}

// =====================================================================================
// per-sample reductions over time:  S1[b,c] += sum_t dy ; S2[b,c] += sum_t dy * o,
// o = other (optionally normalised (other-mean)*rstd).  grid = (B, time splits).
// =====================================================================================
// grid = (B, ceil(C / 128)): a workgroup owns 128 channels of one sample — 16 chunk lanes (8 channels, 16 bytes) x 16 row
// lanes, four rows in flight per lane — and WRITES its sums (no atomics, no zero-fill launches).
template <typename T>
__global__ __launch_bounds__(256) void sample_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ other,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ S1, float* __restrict__ S2, int B, int Tn, int C) {
    __shared__ float red[2][16][16][8];
    const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
    const int b = blockIdx.x;
    const int chunk = blockIdx.y * 16 + cl;
    const bool act = chunk * 8 < C;
    float a[8], q[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; q[e] = 0.f; mu[e] = (act && mean) ? mean[chunk * 8 + e] : 0.f; rs[e] = (act && mean) ? rstd[chunk * 8 + e] : 1.f; }
    if (act) {
        for (int t0 = rl; t0 < Tn; t0 += 64) {
            float d[4][8], o[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(t0 + 16 * u, Tn - 1);
                const size_t off = ((size_t)b * Tn + t) * C + chunk * 8;
                load8(dy + off, d[u]);
                if (other) load8(other + off, o[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t0 + 16 * u < Tn) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a[e] += d[u][e]; if (other) q[e] += d[u][e] * ((o[u][e] - mu[e]) * rs[e]); }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][rl][cl][e] = a[e]; red[1][rl][cl][e] = q[e]; }
    __syncthreads();
    if (tid < 128) {            // thread -> channel tid of the 128
        const int c = blockIdx.y * 128 + tid;
        if (c < C) {
            float sa = 0.f, sq = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa += red[0][r][tid >> 3][tid & 7]; sq += red[1][r][tid >> 3][tid & 7]; }
            S1[(size_t)b * C + c] = sa;
            if (S2) S2[(size_t)b * C + c] = sq;
        }
    }
}

int launch_sample_reduce(int dt, const void* dy, const void* other, const float* mean, const float* rstd,
                         float* S1, float* S2, int B, int T, int C, hipStream_t s) {
    if (C % 8 != 0) { ishara_set_error("sample_reduce: C%%8 != 0"); return -1; }
    dim3 grid(B, (C + 127) / 128);
    if (dt == DT_BF16) hipLaunchKernelGGL(sample_reduce_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)other, mean, rstd, S1, S2, B, T, C);
    else if (dt == DT_F16) hipLaunchKernelGGL(sample_reduce_kernel<f16>, grid, dim3(256), 0, s, (const f16*)dy, (const f16*)other, mean, rstd, S1, S2, B, T, C);
    else hipLaunchKernelGGL(sample_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)other, mean, rstd, S1, S2, B, T, C);
    return LAUNCH_OK();
}

// =====================================================================================
// BatchNorm finalize from per-sample sums [nb, C] (fp64 accumulation across samples)
// =====================================================================================
// block = FIN_CL channels x FIN_BL sample lanes (1024 threads); fp64 accumulation across samples.  16 x 64 instead of
// 64 x 16: C = 512 gives 32 workgroups with 4 samples per thread instead of 8 with 16 (these [B, C] passes are latency
// bound).  A wave holds 4 sample lanes x 16 channels: two shuffles fold the
// lanes, then the 16 waves are summed through LDS in a fixed order.
#define FIN_CL 16
#define FIN_BL 64
#define FIN_NW (FIN_CL * FIN_BL / 64)
DEVI double fin_fold(double v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ ssum, const float* __restrict__ ssq, int nb, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ mmean, float* __restrict__ mvar, int training,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ a, float* __restrict__ bsh, int C, float var_corr, int stride) {
    // stride: floats between consecutive rows of ssum / ssq (C: [nb, C] arrays; 2C: the depthwise conv's partial statistic rows [nb][2][C] read in place)
    __shared__ double rs_[FIN_NW][FIN_CL], rq_[FIN_NW][FIN_CL];
    const int cl = threadIdx.x & (FIN_CL - 1), bl = threadIdx.x / FIN_CL, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * FIN_CL + cl;
    double s = 0.0, q = 0.0;
    if (training && c < C)
        for (int b = bl; b < nb; b += FIN_BL) { s += (double)ssum[(size_t)b * stride + c]; q += (double)ssq[(size_t)b * stride + c]; }
    s = fin_fold(s); q = fin_fold(q);
    if ((threadIdx.x & 63) < FIN_CL) { rs_[wv][cl] = s; rq_[wv][cl] = q; }
    __syncthreads();
    if (bl != 0 || c >= C) return;
    float mu, var;
    if (training) {
        for (int i = 1; i < FIN_NW; ++i) { s += rs_[i][cl]; q += rq_[i][cl]; }
        const double m = s / (double)count;
        double v = q / (double)count - m * m;
        if (v < 0.0) v = 0.0;
        mu = (float)m; var = (float)v;
        mmean[c] = mmean[c] * momentum + mu * (1.f - momentum);
        mvar[c] = mvar[c] * momentum + var * var_corr * (1.f - momentum);
    } else { mu = mmean[c]; var = mvar[c]; }
    const float rs = rsqrtf(var + eps);
    mean[c] = mu; rstd[c] = rs;
    const float aa = gamma[c] * rs;
    a[c] = aa; bsh[c] = beta[c] - mu * aa;
}

int launch_bn_finalize(const float* ssum, const float* ssq, int nb, float count, const float* gamma, const float* beta,
                          float eps, float momentum, float* moving_mean, float* moving_var, int training,
                          float* mean, float* rstd, float* a, float* b, int C, hipStream_t s, float var_corr, int stride) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CL - 1) / FIN_CL), dim3(FIN_CL * FIN_BL), 0, s, ssum, ssq, nb, count, gamma, beta, eps, momentum,
                       moving_mean, moving_var, training, mean, rstd, a, b, C, var_corr, stride ? stride : C);
    return LAUNCH_OK();
}

// =====================================================================================
// ECA gate on [B,C] (one workgroup per sample)
// =====================================================================================
// inf.part != nullptr (inference): the kernel also does what two launches did before it — the sum of the depthwise conv's partial statistic
// rows (the sample's channel sums over time) and the BatchNorm constants from the moving statistics (a = gamma * rsqrt(mv + eps),
// b = beta - mm * a) — so a Conv1DBlock's forward is 4 launches instead of 6 (configs[4]: B = 1, every launch is ~9 us of latency)
// inf.part != nullptr with inf.mm == nullptr (training, round 3): the partial rows are summed here too (the per-sample sums go to inf.gap_out
// for the backward pass — the stats_reduce launch is gone), the BatchNorm constants come from bn_finalize (batch statistics) as before
struct EcaInfer { const float* part = nullptr; int prows = 0; const float* mm = nullptr; const float* mv = nullptr; const float* gamma = nullptr; const float* beta = nullptr; float eps = 0.f;
                  float* gap_out = nullptr; };
__global__ __launch_bounds__(1024) void eca_fwd_kernel(const float* __restrict__ gap, const float* __restrict__ a, const float* __restrict__ bsh,
                                                      const float* __restrict__ w5, float invT, float* __restrict__ gn,
                                                      float* __restrict__ sg, float* __restrict__ P, float* __restrict__ Q, int C, float* __restrict__ rs, DropSpec dp, int dp_fold, EcaInfer inf) {
    extern __shared__ float sh[];   // [C + 4] (+ [C] a, [C] b for the inference form)
    float* al = sh + C + 4;
    float* bl = al + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C + 4; c += blockDim.x) {
        const int cc = c - 2;
        float g = 0.f;
        if (cc >= 0 && cc < C) {
            if (inf.part) {
                const float* p = inf.part + ((size_t)b * inf.prows * 2) * C + cc;
                float s0 = 0.f;
                for (int r = 0; r < inf.prows; ++r) s0 += p[(size_t)(2 * r) * C];
                if (inf.mm) {
                    const float aa = inf.gamma[cc] * rsqrtf(inf.mv[cc] + inf.eps), bb = inf.beta[cc] - inf.mm[cc] * aa;
                    al[cc] = aa; bl[cc] = bb;
                    g = aa * s0 * invT + bb;
                } else {
                    inf.gap_out[(size_t)b * C + cc] = s0;
                    g = a[cc] * s0 * invT + bsh[cc];
                }
            } else g = a[cc] * gap[(size_t)b * C + cc] * invT + bsh[cc];
            gn[(size_t)b * C + cc] = g;
        }
        sh[c] = g;
    }
    __syncthreads();
    if (inf.part && inf.mm) { a = al; bsh = bl; }
    const float w0 = w5[0], w1 = w5[1], w2 = w5[2], w3 = w5[3], w4 = w5[4];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float z = w0 * sh[c] + w1 * sh[c + 1] + w2 * sh[c + 2] + w3 * sh[c + 3] + w4 * sh[c + 4];
        const float sv = sigmoidf_(z);
        sg[(size_t)b * C + c] = sv;
        // drop-path scale of this sample (c5:82-83, noise_shape (None,1,1)): drawn here (dp.thr != 0), published in rs[b] for the GEMM
        // epilogues and the backward pass, and folded into P, Q when `rs` is given
        float r = 1.f;
        if (rs) {
            r = (dp.thr == 0u || rng_keep(rng_row_key(dp.key, (uint32_t)b), 0u, dp.thr)) ? dp.scale : 0.f;
            if (c == 0) rs[b] = r;
            if (!dp_fold) r = 1.f;
        }
        P[(size_t)b * C + c] = a[c] * sv * r;
        Q[(size_t)b * C + c] = bsh[c] * sv * r;
    }
}

int launch_eca_fwd(const float* gap, const float* a, const float* b, const float* w5, float invT,
                   float* gn, float* sgate, float* P, float* Q, int B, int C, hipStream_t s, float* rs, DropSpec dp, int dp_fold) {
    hipLaunchKernelGGL(eca_fwd_kernel, dim3(B), dim3(256), (C + 4) * sizeof(float), s, gap, a, b, w5, invT, gn, sgate, P, Q, C, rs, dp, dp_fold, EcaInfer{});
    return LAUNCH_OK();
}
// training form over the depthwise conv's partial statistic rows: sums them per sample (-> gap_out [B, C]) and gates with bn_finalize's constants
int launch_eca_fwd_part(const float* part, int prows, float* gap_out, const float* a, const float* b, const float* w5, float invT,
                        float* gn, float* sgate, float* P, float* Q, int B, int C, hipStream_t s, float* rs, DropSpec dp, int dp_fold) {
    EcaInfer inf; inf.part = part; inf.prows = prows; inf.gap_out = gap_out;
    hipLaunchKernelGGL(eca_fwd_kernel, dim3(B), dim3(256), (3 * C + 4) * sizeof(float), s, (const float*)nullptr, a, b, w5, invT, gn, sgate, P, Q, C, rs, dp, dp_fold, inf);
    return LAUNCH_OK();
}
int launch_eca_fwd_infer(const float* part, int prows, const float* mm, const float* mv, const float* gamma, const float* beta, float eps, const float* w5, float invT,
                         float* gn, float* sgate, float* P, float* Q, int B, int C, hipStream_t s) {
    EcaInfer inf; inf.part = part; inf.prows = prows; inf.mm = mm; inf.mv = mv; inf.gamma = gamma; inf.beta = beta; inf.eps = eps;
    const int threads = B <= 8 ? (C + 4 > 512 ? 1024 : 512) : 256;      // a clip or a few: a thread per channel (one workgroup per sample is all the parallelism there is)
    hipLaunchKernelGGL(eca_fwd_kernel, dim3(B), dim3(threads), (3 * C + 4) * sizeof(float), s, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, w5, invT, gn, sgate, P, Q, C,
                       (float*)nullptr, DropSpec{0, 0, 1.f}, 0, inf);
    return LAUNCH_OK();
}

// =====================================================================================
// y = x*P[b,c] + Q[b,c] (+resid)  /  y = x*a[c] + b[c]
// =====================================================================================
// grid = (row blocks, samples); a thread keeps one 8-channel chunk, so the per-sample /
// per-channel coefficients are loaded once and the row loop is pure 16-byte streaming.
template <typename T>
__global__ __launch_bounds__(256) void affine_kernel(const T* __restrict__ x, const float* __restrict__ P, const float* __restrict__ Q,
                                                     const T* __restrict__ resid, T* __restrict__ y, int Tn, int C, int per_sample) {
    const int nch = C >> 3;
    const int cpr = min(nch, 256), rpb = 256 / cpr;
    const int cl = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    const int b = blockIdx.y;
    if (rl >= rpb) return;
    for (int chunk = cl; chunk < nch; chunk += cpr) {
        const float* pp = P + (per_sample ? (size_t)b * C : 0) + chunk * 8;
        float p[8], q[8];
        load8(pp, p);
        if (Q) load8(Q + (per_sample ? (size_t)b * C : 0) + chunk * 8, q);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = 0.f;
        }
        for (int t = blockIdx.x * rpb + rl; t < Tn; t += gridDim.x * rpb) {
            const size_t off = ((size_t)b * Tn + t) * C + chunk * 8;
            float v[8];
            load8(x + off, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * p[e] + q[e];
            if (resid) {
                float r[8];
                load8(resid + off, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            store8(y + off, v);
        }
    }
}

static int run_affine(int dt, const void* x, const float* P, const float* Q, const void* resid, void* y, int B, int T, int C, int per_sample, hipStream_t s) {
    if (C % 8 != 0) { ishara_set_error("affine: C%%8 != 0"); return -1; }
    const int cpr = min(C / 8, 256), rpb = 256 / cpr;
    int gx = (T + rpb - 1) / rpb;
    const int cap = max(1, 4096 / max(B, 1));
    if (gx > cap) gx = cap;
    dim3 grid(gx, B);
    if (dt == DT_BF16) hipLaunchKernelGGL(affine_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, P, Q, (const bf16*)resid, (bf16*)y, T, C, per_sample);
    else if (dt == DT_F16) hipLaunchKernelGGL(affine_kernel<f16>, grid, dim3(256), 0, s, (const f16*)x, P, Q, (const f16*)resid, (f16*)y, T, C, per_sample);
    else hipLaunchKernelGGL(affine_kernel<float>, grid, dim3(256), 0, s, (const float*)x, P, Q, (const float*)resid, (float*)y, T, C, per_sample);
    return LAUNCH_OK();
}
int launch_sample_affine(int dt, const void* x, const float* P, const float* Q, const void* resid, void* y, int B, int T, int C, hipStream_t s) {
    return run_affine(dt, x, P, Q, resid, y, B, T, C, 1, s);
}
int launch_col_affine(int dt, const void* x, const float* a, const float* b, void* y, int M, int C, hipStream_t s) {
    // rows are independent: present them as min(M, 256) pseudo-samples for grid parallelism
    int Bp = 1;
    for (int cand = 256; cand >= 1; cand >>= 1) if (M % cand == 0) { Bp = cand; break; }
    return run_affine(dt, x, a, b, nullptr, y, Bp, M / Bp, C, 0, s);
}

// =====================================================================================
// BatchNorm backward pieces
// =====================================================================================
// Conv1DBlock (BN -> ECA): step 1, per sample.  E <- dgn[b,c]; dw5 += sum dz*gn(shifted)
// grid = (B, channel chunks of CC): a workgroup owns CC channels of one sample (the 5-tap channel convolution reaches 2 channels into the
// neighbouring chunks).  ps.G != nullptr (PsaStats, kernels.h): S1, S2 are first computed from what the per-sample-affine weight-gradient
// GEMM emitted — S1[c] = rs * sum_n Wt[n,c] G[n], S2[c] = rstd[c] * (rs * sum_p Rpart[p][c] - mean[c] * S1[c]) — for the chunk and one
// 8-channel group of halo on each side (kept in LDS; only the chunk's own values are written out): thread (cg, w) takes 8 channels (16-byte
// weight loads, channel = fast index) and a quarter of the N weight rows, 32 loads in flight (one thread per channel over all N sat on load
// latency: 37 us as a kernel of its own; one workgroup per sample over all channels: 32 us at C = 1024, B = 64)
__global__ __launch_bounds__(256) void eca_bwd_sample_kernel(float* __restrict__ S1, float* __restrict__ S2,
                                                             const float* __restrict__ gn, const float* __restrict__ sg,
                                                             const float* __restrict__ w5, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ E,
                                                             float* __restrict__ dw5part, int C, int CC, PsaStats ps) {
    extern __shared__ float sh[];
    const int CH = CC + 16;
    float* dz = sh;                 // [CC + 4]: channel c_lo - 2 + i
    float* g = dz + CC + 4;         // [CC + 4]
    float* s1l = g + CC + 4;        // [CH]: channel c_lo - 8 + i
    float* s2l = s1l + CH;
    float* gl = s2l + CH;           // [N]
    __shared__ float wred[5][4];
    const int b = blockIdx.x, c_lo = (int)blockIdx.y * CC, c_hi = min(C, c_lo + CC);
    if (ps.G) {
        const int N = ps.N, cg = threadIdx.x & 63, w = threadIdx.x >> 6;
        float* part = gl + N;       // [4][CH]
        for (int n = threadIdx.x; n < N; n += 256) gl[n] = ps.G[(size_t)b * N + n];
        const bf16* Wt = reinterpret_cast<const bf16*>(ps.Wt);
        const int nq = (N + 3) / 4, nbeg = w * nq, nend = min(N, nbeg + nq);
        __syncthreads();
        for (int i0 = cg * 8; i0 < CH; i0 += 512) {
            const int c0 = c_lo - 8 + i0;
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = 0.f;
            if (c0 >= 0 && c0 < C) {
                for (int n = nbeg; n < nend; n += 32) {
                    bf16x8 wv[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) wv[u] = *reinterpret_cast<const bf16x8*>(Wt + (size_t)min(n + u, nend - 1) * ps.ldt + c0);
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        const float gv = n + u < nend ? gl[n + u] : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] += (float)wv[u][e] * gv;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) part[w * CH + i0 + e] = a[e];
        }
        __syncthreads();
        const float r = ps.rs ? ps.rs[b] : 1.f;
        for (int i = threadIdx.x; i < CH; i += 256) {
            const int c = c_lo - 8 + i;
            float s1 = 0.f, s2 = 0.f;
            if (c >= 0 && c < C) {
                float R = 0.f;
                for (int p = 0; p < ps.nparts; ++p) R += ps.Rpart[((size_t)b * ps.nparts + p) * C + c];
                s1 = r * (part[i] + part[CH + i] + part[2 * CH + i] + part[3 * CH + i]);
                s2 = ps.rstd[c] * (r * R - ps.mean[c] * s1);
                if (c >= c_lo && c < c_hi) { S1[(size_t)b * C + c] = s1; S2[(size_t)b * C + c] = s2; }
            }
            s1l[i] = s1; s2l[i] = s2;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < CC + 4; i += blockDim.x) {
        const int cc = c_lo - 2 + i;
        float z = 0.f, gg = 0.f;
        if (cc >= 0 && cc < C) {
            const size_t idx = (size_t)b * C + cc;
            const float s1 = ps.G ? s1l[i + 6] : S1[idx], s2 = ps.G ? s2l[i + 6] : S2[idx];
            const float ds = gamma[cc] * s2 + beta[cc] * s1;
            const float sv = sg[idx];
            z = ds * sv * (1.f - sv);
            gg = gn[idx];
        }
        dz[i] = z; g[i] = gg;
    }
    __syncthreads();
    float wp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const int li = c - c_lo;
        // dgn[c] = sum_j w5[j] * dz(channel c - j + 2)  -> local index li - j + 4
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) acc += w5[j] * dz[li - j + 4];
        E[(size_t)b * C + c] = acc;
        // dw5[j] += dz(c) * gn(c + j - 2) -> local index li + j
        const float z = dz[li + 2];
#pragma unroll
        for (int j = 0; j < 5; ++j) wp[j] += z * g[li + j];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) { const float v = wave_sum(wp[j]); if (lane == 0) wred[j][wid] = v; }
    __syncthreads();
    if (threadIdx.x < 5) dw5part[((size_t)b * gridDim.y + blockIdx.y) * 8 + threadIdx.x] = wred[threadIdx.x][0] + wred[threadIdx.x][1] + wred[threadIdx.x][2] + wred[threadIdx.x][3];   // summed in order by the channel kernel
}

// step 2, per channel (FIN_CL channels x FIN_BL sample lanes per block): dgamma, dbeta, Fc; E[b,c] <- dgn/T - dbeta/Mtot
__global__ __launch_bounds__(1024) void eca_bn_bwd_channel_kernel(const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ gap,
                                          const float* __restrict__ sg, const float* __restrict__ mean, const float* __restrict__ rstd,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ E, float* __restrict__ Fc,
                                          const float* __restrict__ dw5part, int nparts5, float* __restrict__ dw5, int B, int Tn, int C) {
    __shared__ double rg_[FIN_NW][FIN_CL], rb_[FIN_NW][FIN_CL];
    __shared__ float eb_[FIN_CL];
    if (blockIdx.x == 0 && threadIdx.x < 5 * 64) {      // ECA tap gradient: per-(sample, chunk) partials of step 1, summed in a fixed order (wave j = tap j)
        const int j = threadIdx.x >> 6, l = threadIdx.x & 63;
        float a = 0.f;
        for (int b = l; b < nparts5; b += 64) a += dw5part[(size_t)b * 8 + j];
        a = wave_sum(a);
        if (l == 0) dw5[j] += a;
    }
    const int cl = threadIdx.x & (FIN_CL - 1), bl = threadIdx.x / FIN_CL, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * FIN_CL + cl;
    const bool act = c < C;
    const float invT = 1.f / (float)Tn, mu = act ? mean[c] : 0.f, rs = act ? rstd[c] : 0.f;
    double dg = 0.0, db = 0.0;
    if (act)
        for (int b = bl; b < B; b += FIN_BL) {
            const size_t i = (size_t)b * C + c;
            const float ghat = (gap[i] * invT - mu) * rs;
            dg += (double)(sg[i] * S2[i] + E[i] * ghat);
            db += (double)(sg[i] * S1[i] + E[i]);
        }
    dg = fin_fold(dg); db = fin_fold(db);
    if ((threadIdx.x & 63) < FIN_CL) { rg_[wv][cl] = dg; rb_[wv][cl] = db; }
    __syncthreads();
    const float mtot = (float)B * (float)Tn;
    if (bl == 0 && act) {
        for (int i = 1; i < FIN_NW; ++i) { dg += rg_[i][cl]; db += rb_[i][cl]; }
        dgamma[c] += (float)dg;
        dbeta[c] += (float)db;
        Fc[c] = (float)dg / mtot;
        eb_[cl] = (float)db / mtot;
    }
    __syncthreads();
    if (act) {
        const float eb = eb_[cl];
        for (int b = bl; b < B; b += FIN_BL) { const size_t i = (size_t)b * C + c; E[i] = E[i] * invT - eb; }
    }
}

int launch_eca_bn_bwd_finalize(float* S1, float* S2, const float* gap, const float* gn, const float* sgate,
                               const float* w5, const float* gamma, const float* beta, const float* mean, const float* rstd,
                               float* dgamma, float* dbeta, float* dw5, float* E, float* Fc, float* dw5part, int B, int T, int C, hipStream_t s, const PsaStats* ps) {
    PsaStats p = ps ? *ps : PsaStats{};
    if (p.G && (C % 8 != 0 || p.ldt % 8 != 0 || ((uintptr_t)p.Wt) % 16 != 0)) { ishara_set_error("eca_bn_bwd_finalize: PsaStats needs C %% 8 == 0 and 16-byte aligned weight rows"); return -1; }
    const int CC = (C > 256 && C % 256 == 0 && C / 256 <= ECA_MAX_CHUNKS) ? 256 : C, nchunk = C / CC;      // dw5part: B * nchunk * 8 floats
    const size_t shm = (size_t)(2 * (CC + 4) + 2 * (CC + 16) + (p.G ? p.N + 4 * (CC + 16) : 0)) * sizeof(float);
    hipLaunchKernelGGL(eca_bwd_sample_kernel, dim3(B, nchunk), dim3(256), shm, s, S1, S2, gn, sgate, w5, gamma, beta, E, dw5part, C, CC, p);
    hipLaunchKernelGGL(eca_bn_bwd_channel_kernel, dim3((C + FIN_CL - 1) / FIN_CL), dim3(FIN_CL * FIN_BL), 0, s, S1, S2, gap, sgate, mean, rstd, dgamma, dbeta, E, Fc, dw5part, B * nchunk, dw5, B, T, C);
    return LAUNCH_OK();
}

__global__ __launch_bounds__(1024) void bn_bwd_channel_kernel(const float* __restrict__ S1, const float* __restrict__ S2, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, float* __restrict__ Ecol, float* __restrict__ Fc, int B, int Tn, int C) {
    __shared__ double rg_[FIN_NW][FIN_CL], rb_[FIN_NW][FIN_CL];
    const int cl = threadIdx.x & (FIN_CL - 1), bl = threadIdx.x / FIN_CL, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * FIN_CL + cl;
    double dg = 0.0, db = 0.0;
    if (c < C)
        for (int b = bl; b < B; b += FIN_BL) { dg += (double)S2[(size_t)b * C + c]; db += (double)S1[(size_t)b * C + c]; }
    dg = fin_fold(dg); db = fin_fold(db);
    if ((threadIdx.x & 63) < FIN_CL) { rg_[wv][cl] = dg; rb_[wv][cl] = db; }
    __syncthreads();
    if (bl != 0 || c >= C) return;
    for (int i = 1; i < FIN_NW; ++i) { dg += rg_[i][cl]; db += rb_[i][cl]; }
    const float mtot = (float)B * (float)Tn;
    dgamma[c] += (float)dg;
    dbeta[c] += (float)db;
    Fc[c] = (float)dg / mtot;
    Ecol[c] = -(float)db / mtot;
}

int launch_bn_bwd_finalize(const float* S1, const float* S2, float* dgamma, float* dbeta, float* Ecol, float* Fc,
                           int B, int T, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_channel_kernel, dim3((C + FIN_CL - 1) / FIN_CL), dim3(FIN_CL * FIN_BL), 0, s, S1, S2, dgamma, dbeta, Ecol, Fc, B, T, C);
    return LAUNCH_OK();
}

// dx = a[c] * (dy*sg[b,c] + E - xhat*Fc[c]) = dy*k1 + k0 - x*k2 with per-(sample,channel) constants
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ a, const float* __restrict__ sg,
                                                           const float* __restrict__ E, int e_per_sample, const float* __restrict__ Fc,
                                                           T* __restrict__ dx, int Tn, int C) {
    const int nch = C >> 3;
    const int cpr = min(nch, 256), rpb = 256 / cpr;
    const int cl = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    const int b = blockIdx.y;
    if (rl >= rpb) return;
    for (int chunk = cl; chunk < nch; chunk += cpr) {
        const int ch = chunk * 8;
        float k0[8], k1[8], k2[8];
        {
            float mu[8], rs[8], aa[8], fc[8], ee[8], g[8];
            load8(mean + ch, mu); load8(rstd + ch, rs); load8(a + ch, aa); load8(Fc + ch, fc);
            load8(E + (e_per_sample ? (size_t)b * C : 0) + ch, ee);
            if (sg) load8(sg + (size_t)b * C + ch, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                k2[e] = aa[e] * rs[e] * fc[e];
                k1[e] = aa[e] * (sg ? g[e] : 1.f);
                k0[e] = aa[e] * ee[e] + mu[e] * k2[e];
            }
        }
        for (int t = blockIdx.x * rpb + rl; t < Tn; t += gridDim.x * rpb) {
            const size_t off = ((size_t)b * Tn + t) * C + ch;
            float d[8], xv[8];
            load8(dy + off, d);
            load8(x + off, xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = d[e] * k1[e] + k0[e] - xv[e] * k2[e];
            store8(dx + off, d);
        }
    }
}

int launch_bn_bwd_apply(int dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* a,
                        const float* sg, const float* E, int e_per_sample, const float* Fc, void* dx,
                        int B, int T, int C, hipStream_t s) {
    if (C % 8 != 0) { ishara_set_error("bn_bwd_apply: C%%8 != 0"); return -1; }
    const int cpr = min(C / 8, 256), rpb = 256 / cpr;
    int gx = (T + rpb - 1) / rpb;
    const int cap = max(1, 4096 / max(B, 1));
    if (gx > cap) gx = cap;
    dim3 grid(gx, B);
    if (dt == DT_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, mean, rstd, a, sg, E, e_per_sample, Fc, (bf16*)dx, T, C);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)x, mean, rstd, a, sg, E, e_per_sample, Fc, (float*)dx, T, C);
    return LAUNCH_OK();
}

// =====================================================================================
// Squeeze-Excite MLP (one workgroup per sample; C <= 1024, R <= 128)
// =====================================================================================
// sum_i v[i0 + i * vstep] * w[i * wstride] over n terms, 16 weight loads in flight (these matvecs are chains of L2 latencies, not of bytes)
DEVI float se_dot(const float* __restrict__ w, size_t wstride, const float* v, int vstep, int n) {
    float acc = 0.f;
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = w[(size_t)(i + u) * wstride];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t[u] * v[(i + u) * vstep];
    }
    for (; i < n; ++i) acc += w[(size_t)i * wstride] * v[i * vstep];
    return acc;
}
__global__ __launch_bounds__(256) void se_fwd_kernel(const float* __restrict__ gap, float invT, const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ hid_pre,
                                                     float* __restrict__ se, int C, int R) {
    extern __shared__ float sh[];   // z[C], h[R], part[256]
    float* z = sh;
    float* h = sh + C;
    float* part = h + R;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) z[c] = gap[(size_t)b * C + c] * invT;
    __syncthreads();
    if (R <= 256 && 256 % R == 0 && blockDim.x == 256) {      // the squeeze matvec over all 256 threads: thread (r, p) sums every (256 / R)-th channel
        const int np = 256 / R, r = threadIdx.x % R, p = threadIdx.x / R;      // (R threads over all C channels: a 65 us chain of loads at C = 512)
        part[p * R + r] = se_dot(W1 + (size_t)p * R + r, (size_t)np * R, z + p, np, (C - p + np - 1) / np);
        __syncthreads();
        if (threadIdx.x < R) {
            float t = b1[r];
            for (int q = 0; q < np; ++q) t += part[q * R + r];
            hid_pre[(size_t)b * R + r] = t;
            h[r] = swishf_(t);
        }
    } else {
        for (int r = threadIdx.x; r < R; r += blockDim.x) {
            float acc = b1[r];
            for (int c = 0; c < C; ++c) acc += z[c] * W1[(size_t)c * R + r];
            hid_pre[(size_t)b * R + r] = acc;
            h[r] = swishf_(acc);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) se[(size_t)b * C + c] = sigmoidf_(b2[c] + se_dot(W2 + c, (size_t)C, h, 1, R));
}

int launch_se_fwd(const float* gap, float invT, const float* W1, const float* b1, const float* W2, const float* b2,
                  float* hid_pre, float* se, int B, int C, int R, hipStream_t s) {
    hipLaunchKernelGGL(se_fwd_kernel, dim3(B), dim3(256), (C + R + 256) * sizeof(float), s, gap, invT, W1, b1, W2, b2, hid_pre, se, C, R);
    return LAUNCH_OK();
}

// Squeeze-excite backward, step 1 (one workgroup per sample): dz2 = dse*se*(1-se), dhp = (W2 dz2) * swish'(hid_pre), dgapT = W1 dhp / T;
// dz2, dhp and h = swish(hid_pre) go to scr[b][C + 2R] for the weight-gradient pass
__global__ __launch_bounds__(256) void se_bwd_kernel(const float* __restrict__ dse, const float* __restrict__ gap, float invT,
                                                     const float* __restrict__ W1, const float* __restrict__ W2,
                                                     const float* __restrict__ hid_pre, const float* __restrict__ se,
                                                     float* __restrict__ scr, float* __restrict__ dgapT, int C, int R) {
    extern __shared__ float sh[];   // dp2[C], dhp[R], part[256]
    float* dp2 = sh;
    float* dhp = sh + C;
    float* part = dhp + R;
    const int b = blockIdx.x;
    float* sb = scr + (size_t)b * (C + 2 * R);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const size_t i = (size_t)b * C + c;
        const float sv = se[i];
        const float d = dse[i] * sv * (1.f - sv);
        dp2[c] = d;
        sb[c] = d;
    }
    __syncthreads();
    if (R <= 256 && 256 % R == 0 && blockDim.x == 256) {      // W2[r, :] . dp2 over all 256 threads: thread (p, r) takes the channels c = p mod (256 / R)
        const int np = 256 / R, r = threadIdx.x / np, p = threadIdx.x % np;      // consecutive threads read consecutive channels of one weight row
        part[r * np + p] = se_dot(W2 + (size_t)r * C + p, (size_t)np, dp2 + p, np, (C - p + np - 1) / np);
        __syncthreads();
        if (threadIdx.x < R) {
            const int rr = threadIdx.x;
            float t = 0.f;
            for (int q = 0; q < np; ++q) t += part[rr * np + q];
            const float hp = hid_pre[(size_t)b * R + rr];
            const float d = t * dswishf_(hp);
            dhp[rr] = d;
            sb[C + rr] = d;
            sb[C + R + rr] = swishf_(hp);
        }
    } else {
        for (int r = threadIdx.x; r < R; r += blockDim.x) {
            float acc = 0.f;
            for (int c = 0; c < C; ++c) acc += W2[(size_t)r * C + c] * dp2[c];
            const float hp = hid_pre[(size_t)b * R + r];
            const float d = acc * dswishf_(hp);
            dhp[r] = d;
            sb[C + r] = d;
            sb[C + R + r] = swishf_(hp);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) dgapT[(size_t)b * C + c] = se_dot(W1 + (size_t)c * R, 1, dhp, 1, R) * invT;
}
// step 2: weight gradients as sums over the samples in a fixed order (no float atomics: the gradients repeat bit for bit).
// workgroup = 64 parameters (lane) x 4 sample groups (wave w takes b = w, w+4, ...; 8 independent loads in flight), combined through LDS:
// dW1[c][r] += sum_b z[b,c]*dhp[b,r], db1[r] += sum_b dhp[b,r], dW2[r][c] += sum_b h[b,r]*dz2[b,c], db2[c] += sum_b dz2[b,c]
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ scr, const float* __restrict__ gap, float invT,
                                                       float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                       int B, int C, int R) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int CR = C * R, S = C + 2 * R;
    // parameter i -> (first factor pointer / stride, second factor pointer / stride, output)
    const float* f0 = nullptr; const float* f1 = nullptr; size_t s0 = 0, s1 = S; float sc = 1.f; float* out = nullptr;
    if (i < CR) { const int r = i / C, c = i - r * C; f0 = scr + C + R + r; s0 = S; f1 = scr + c; out = dW2 + i; }                       // dW2[r][c]: c fastest
    else if (i < 2 * CR) { const int k = i - CR, c = k / R, r = k - c * R; f0 = gap + c; s0 = C; sc = invT; f1 = scr + C + r; out = dW1 + k; }   // dW1[c][r]
    else if (i < 2 * CR + C) { f1 = scr + (i - 2 * CR); out = db2 + (i - 2 * CR); }
    else if (i < 2 * CR + C + R) { f1 = scr + C + (i - 2 * CR - C); out = db1 + (i - 2 * CR - C); }
    float acc = 0.f;
    if (out) {
        for (int b0 = w; b0 < B; b0 += 32) {
            float a[8], c8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 4 * u;
                a[u] = (b < B && f0) ? f0[(size_t)b * s0] : 1.f;
                c8[u] = b < B ? f1[(size_t)b * s1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += a[u] * sc * c8[u];
        }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && out) *out += red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// scr: B * (C + 2R) floats of scratch
int launch_se_bwd(const float* dse, const float* gap, float invT, const float* W1, const float* W2,
                  const float* hid_pre, const float* se, float* dW1, float* db1, float* dW2, float* db2,
                  float* dgapT, float* scr, int B, int C, int R, hipStream_t s) {
    hipLaunchKernelGGL(se_bwd_kernel, dim3(B), dim3(256), (C + R + 256) * sizeof(float), s, dse, gap, invT, W1, W2, hid_pre, se, scr, dgapT, C, R);
    hipLaunchKernelGGL(se_wgrad_kernel, dim3((2 * C * R + C + R + 63) / 64), dim3(256), 0, s, scr, gap, invT, dW1, db1, dW2, db2, B, C, R);
    return LAUNCH_OK();
}

// =====================================================================================
// Small streaming passes that materialise an operand transform once so that every GEMM
// (forward, dgrad and wgrad) runs its fast untransformed path:
//   MAP_SWISH    y = swish(x)                  (Squeezeformer conv3 input)
//   MAP_ROWSCALE y = x * rs[row / T]           (drop-path applied to the incoming gradient)
//   MAP_DROPMASK y = x * mask(row, col)        (inverted dropout applied to the incoming gradient)
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256) void map_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int op, const float* __restrict__ rs,
                                                       DropSpec drop, int M, int Tn, int C) {
    const int nch = C >> 3;
    const int cpr = min(nch, 256), rpb = 256 / cpr;
    const int cl = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    if (rl >= rpb) return;
    for (int row = blockIdx.x * rpb + rl; row < M; row += gridDim.x * rpb) {
        float sc = 1.f;
        uint32_t rk = 0;
        if (op == MAP_ROWSCALE) sc = rs[row / Tn];
        else if (op == MAP_DROPMASK) rk = rng_row_key(drop.key, (uint32_t)row);
        for (int chunk = cl; chunk < nch; chunk += cpr) {
            const size_t off = (size_t)row * C + chunk * 8;
            float v[8];
            load8(x + off, v);
            if (op == MAP_SWISH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = swishf_(v[e]);
            } else if (op == MAP_ROWSCALE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= sc;
            } else {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const uint32_t h = rng_pair(rk, (uint32_t)(chunk * 8 + e));
                    v[e] = (h & 0xffffu) >= drop.thr ? v[e] * drop.scale : 0.f;
                    v[e + 1] = (h >> 16) >= drop.thr ? v[e + 1] * drop.scale : 0.f;
                }
            }
            store8(y + off, v);
        }
    }
}

int launch_map_rows(int dt, int op, const void* x, void* y, const float* rs, DropSpec drop, int M, int T, int C, hipStream_t s) {
    if (C % 8 != 0) { ishara_set_error("map_rows: C%%8 != 0"); return -1; }
    const int cpr = min(C / 8, 256), rpb = 256 / cpr;
    const int grid = max(1, min((M + rpb - 1) / rpb, 4096));
    if (dt == DT_BF16) hipLaunchKernelGGL(map_rows_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)x, (bf16*)y, op, rs, drop, M, T, C);
    else if (dt == DT_F16) hipLaunchKernelGGL(map_rows_kernel<f16>, dim3(grid), dim3(256), 0, s, (const f16*)x, (f16*)y, op, rs, drop, M, T, C);
    else hipLaunchKernelGGL(map_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, op, rs, drop, M, T, C);
    return LAUNCH_OK();
}

// ------------------------------------------------------------------ row log-softmax (the torch Squeezeformer's output layer)
// `F.log_softmax(self.fc(encoder_outputs), dim=-1)` — squeezeformer/model.py:448-449.  One wavefront per row of C fp32 logits
// (C is a class count: tens to a few thousand), exact two-pass form in registers/loop: max, then sum of exp, cross-lane by DPP
// butterflies; a workgroup of 4 waves takes 4 rows per iteration of a grid-stride loop.
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int M, int C, int ld) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int m = blockIdx.x * 4 + w; m < M; m += gridDim.x * 4) {
        const float* xr = x + (size_t)m * ld;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xr[c]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += __expf(xr[c] - mx);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o, 64);
        const float lse = mx + __logf(se);
        float* yr = y + (size_t)m * ld;
        for (int c = lane; c < C; c += 64) yr[c] = xr[c] - lse;
        for (int c = C + lane; c < ld; c += 64) yr[c] = 0.f;          // padding columns of the row stride
    }
}
// dx = dy - exp(y) * sum_c dy      (y = the forward's output)
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int M, int C, int ld) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int m = blockIdx.x * 4 + w; m < M; m += gridDim.x * 4) {
        const float* gr = dy + (size_t)m * ld;
        const float* yr = y + (size_t)m * ld;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += gr[c];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        float* dr = dx + (size_t)m * ld;
        for (int c = lane; c < C; c += 64) dr[c] = gr[c] - __expf(yr[c]) * s;
        for (int c = C + lane; c < ld; c += 64) dr[c] = 0.f;
    }
}
int launch_log_softmax_fwd(const float* x, float* y, int M, int C, int ld, hipStream_t s) {
    if (M < 1 || C < 1 || ld < C) { ishara_set_error("log_softmax: M=%d C=%d ld=%d", M, C, ld); return -1; }
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3(max(1, min((M + 3) / 4, 2048))), dim3(256), 0, s, x, y, M, C, ld);
    return LAUNCH_OK();
}
int launch_log_softmax_bwd(const float* dy, const float* y, float* dx, int M, int C, int ld, hipStream_t s) {
    if (M < 1 || C < 1 || ld < C) { ishara_set_error("log_softmax: M=%d C=%d ld=%d", M, C, ld); return -1; }
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(max(1, min((M + 3) / 4, 2048))), dim3(256), 0, s, dy, y, dx, M, C, ld);
    return LAUNCH_OK();
}
