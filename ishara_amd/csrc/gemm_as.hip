// gemm_as.hip — "A-stationary" NT GEMM for the tall-skinny products of the encoder (gfx950).
//
//   C[M,N] = epi( A[M,K] . Bt[N,K]^T ),  bf16 operands, fp32 accumulate, K in {256, 512}, N <= 1024, M ~ 1e5.
//
// Every dense / pointwise-conv product of the model (reference notebook cells c6-c8: Dense / Conv1D(kernel 1) layers)
// has a tiny weight matrix (<= 512 KB, L2 resident) and a huge activation matrix that must be streamed from HBM exactly
// once.  A tile kernel re-stages BOTH operands through LDS for every output tile (~400 MB of L2->LDS DMA per launch at
// M=98304, K=256, N=512, and one barrier per 32-wide K step).  Here instead
//   * each wave keeps its 32 rows x full K of A as MFMA fragments in registers, loaded once straight from global
//     memory with the whole K in flight (no LDS, no barrier on the activation side);
//   * only the weight matrix streams through LDS: a ring of 3 x 16 KB stages filled by LDS-DMA, one stage = NS weight
//     rows (output columns) x full K, shared by the workgroup's 4 waves; one barrier per NS output columns;
//   * accumulators are held transposed (acc = mfma(Bt fragment, A fragment)), so a lane owns 4 consecutive output
//     columns of one row per 16-column MFMA tile; for bf16 C the weight rows fed to a PAIR of tiles are permuted
//     (tile jj of the pair takes staged rows 8*(t>>2) + 4*jj + (t&3), t = 0..15) so that the lane's 4 + 4 values are 8
//     CONSECUTIVE columns: the epilogue runs straight from registers with 16-byte loads / stores, 64 contiguous bytes
//     per row per instruction.  (tools/micro/store_depth.hip: the unpermuted 8-byte stores reach 4.0 TB/s of HBM write,
//     the 16-byte ones 5.9 TB/s.)  The permutation is free: it only changes the LDS address of the fragment read.
// The VM counter retires loads, stores and LDS-DMA in issue order, so the waits on the DMA ring are counted
// (`s_waitcnt vmcnt(N)`); partial row blocks fall back to vmcnt(0).  The residual loads of the fast epilogue are ordinary
// loads issued right after the barrier and BEFORE the step's DMA, consumed after the MFMA phase.  (Issuing them one
// step ahead by inline asm does not survive register allocation: hipcc copies the destination registers at the loop
// back-edge while the loads are still in flight.)
#include <cstdio>
#include <map>
#include <string>
#include <type_traits>
#include "common.h"
#include "kernels.h"

// The 16-bit operand type of this translation unit: bf16 here; gemm_as_f16.hip includes this file again with AS_F16 defined for the
// fp16 inference path (ISHARA_F16).  Only the forward instantiations exist there.
#ifdef AS_F16
typedef f16 as_t;
typedef f16x8 as_v8;
#define AS_MFMA(b, a, acc) __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0)
#define AS_DT DT_F16
#define AS_NAME(x) x##_f16
#else
typedef bf16 as_t;
typedef bf16x8 as_v8;
#define AS_MFMA(b, a, acc) __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0)
#define AS_DT DT_BF16
#define AS_NAME(x) x
#endif

extern int g_force_regstage;
bool gemm_nt_as_applicable(int dtC, int M, int N, int K, int ldb, const EpiArgs& ea);

#define AS_NS 32            // output columns (staged weight rows) per step
#define AS_SMALL_M 1536     // K <= 512: at or below this many rows the workgroups take 64 rows instead of 128 / 192, and the columns are split
                            // down to one 32-column step per workgroup (latency of a B = 1 clip: 22.6 -> ~7 us per K = 512 GEMM)
#define AS_MID_M_K512 32768 // K = 512 only: up to this many rows (<= 512 workgroups of 64 rows = one round at 2 per CU) the 64-row form also wins on
                            // load balance — M = 32768 gives 171 workgroups of 192 rows x 2 column halves for 512 slots (configs[3]: 46.8 -> 44.8 ms/step)
#define AS_MAXN 1024

typedef __attribute__((ext_vector_type(4))) uint32_t as_u32x4;

template <int N> DEVI void as_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// swizzle of a staged weight row: 16-byte chunk q of row r sits at position q ^ as_swz(r).  Uses row bits 0,1,3 so that
// both fragment-row patterns (r = 16j + c and the paired r = 8*(c>>2) + 4*jj + (c&3)) give 8 distinct values over any
// 8 consecutive lanes (conflict-free ds_read_b128).
DEVI int as_swz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int GW, typename T> DEVI void as_st(T* p, const float (&v)[GW]) {
    if constexpr (GW == 8) store8(p, v); else store4(p, v);
}
// the same with the non-temporal hint: tensors that are only read again in the BACKWARD pass (saved pre-activations, the prologues'
// transformed rows) should not push the tensors the next launch reads out of L2 / Infinity Cache
template <int GW, typename T> DEVI void as_st_nt(T* p, const float (&v)[GW], bool nt = true) {
    if constexpr (GW == 8 && is_16b_t<T>::value) {
        as_v8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (as_t)v[i];
        if (nt) __builtin_nontemporal_store(t, reinterpret_cast<as_v8*>(p)); else *reinterpret_cast<as_v8*>(p) = t;
    } else as_st<GW>(p, v);
}
// packed 16-byte store of an already converted group
DEVI void as_st_pk(as_t* p, const as_v8& t, bool nt) {
    if (nt) __builtin_nontemporal_store(t, reinterpret_cast<as_v8*>(p)); else *reinterpret_cast<as_v8*>(p) = t;
}
// 16 bytes of TC -> GW floats (8 bf16 / 4 f32)
template <typename TC, int GW> DEVI void as_unpack(const as_u32x4& r, float (&v)[GW]) {
    if constexpr (GW == 8 && std::is_same<TC, f16>::value) {
        const f16x8 h = __builtin_bit_cast(f16x8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    } else if constexpr (GW == 8) {
#pragma unroll
        for (int h = 0; h < 4; ++h) { v[2 * h] = __uint_as_float(r[h] << 16); v[2 * h + 1] = __uint_as_float(r[h] & 0xffff0000u); }
    } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) v[h] = __uint_as_float(r[h]);
    }
}

// epilogue features, compiled in per kernel instantiation (AS_ALL: every feature behind its run-time flag)
enum { AS_RESID = 1, AS_DACT = 2, AS_ACT = 4, AS_DROP = 8, AS_ROWSCALE = 16, AS_PREOUT = 32, AS_QKV = 64, AS_ADDTAB = 128, AS_ALL = 255 };
template <int MASK, int F> DEVI bool as_on(bool runtime) {
    if constexpr ((MASK & F) == 0) return false;
    else if constexpr (MASK == AS_ALL) return runtime;
    else return true;
}

// One pass of a workgroup over all N columns for the rows [m_base, m_base + 64*RT): wave w owns rows m_base + 16*RT*w ..,
// RT row tiles of 16.  KT = K / 32 (8 or 16); MASK: epilogue features.
// PRO: operand prologue — 0 none, 1 LayerNorm over K (ea.ln_*), 2 per-sample affine (ea.pa_*); both write the transformed rows to
// ea.pro_out when it is set.  The coefficient vectors are staged in the ring slot the DMA fills last (free until step 0's issue).
// CS > 1: the CHUNKED form — a stage holds CS column steps (CS * 32 weight rows), two stages; the DMA wait, the barrier and the next stage's DMA
// issue happen once per CS steps, the steps inside a stage run without any synchronisation (see gemm_nt_as_chunk_kernel).
template <typename TC, int KT, int MASK, int RT, int DBG, int PRO, int NW = 4, int CS = 1>
DEVI void as_pass(const as_t* __restrict__ A, const as_t* __restrict__ Bt, TC* __restrict__ C, int M, int N, int ldb, const EpiArgs& ea,
                  char* smem, const float* bias_s, int m_base) {
    constexpr int RB = KT * 64;                    // bytes of one staged weight row (full K)
    constexpr int NS = AS_NS;                      // output columns per step
    constexpr int SUB = NS * RB;                   // bytes of one column step's weight rows: 16 KB (K=256) / 32 KB (K=512)
    constexpr int STAGE = SUB * CS;                // one DMA stage
    constexpr int R = CS > 1 ? 2 : (KT <= 8 ? 3 : 2);      // ring depth: 48 KB / 64 KB of LDS (chunked: 2 x 64 KB at K = 256, CS = 4)
    constexpr int K = KT * 32;
    constexpr int NI = STAGE / 1024;               // 1 KB DMA instructions per stage
    constexpr int DPW = (NI + NW - 1) / NW;        // ... per wave (NW waves per workgroup); when NW does not divide NI the last waves repeat the first pieces
    constexpr bool PAIR = is_16b_t<TC>::value;     // pair MFMA tiles so that a lane owns 8 consecutive columns (16-byte bf16 accesses)
    constexpr int GW = PAIR ? 8 : 4;               // columns per lane per group
    constexpr int NG = NS / (4 * GW);              // groups per row tile per step: 1 (bf16 C) / 2 (f32 C)
    constexpr int SPS = RT * NG;                   // stores per lane per step
    constexpr bool COUNTED = (MASK & (AS_RESID | AS_DACT | AS_ADDTAB | AS_QKV)) == 0;   // epilogue = a fixed number of stores, no loads
    constexpr int OPS = SPS * ((MASK & AS_PREOUT) ? 2 : 1);
    // HOLD: the 16-byte groups of an even column step wait, packed, in registers and are stored together with the next (odd) step's: the two
    // 64-byte halves of every row's 128-byte line then reach L2 back to back.  One step apart (~2 us) the first half has often left L2 for
    // cold HBM before the second arrives, and such half-line writes run at 3.8 TB/s against 4.6+ for whole lines (tools/micro/store_cold.hip).
    constexpr bool HOLDABLE = PAIR && std::is_same<TC, as_t>::value && MASK != AS_ALL && (MASK & AS_QKV) == 0;
    const bool hold = HOLDABLE && (ea.as_flags & 1) && !ea.n_valid;
    const bool nt_side = (ea.as_flags & 2) != 0;

    const int dbg = DBG ? ea.dbg : 0;          // ablation bits of tools/gemm_ablate.py, compiled out of the production kernels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int mw = m_base + wid * 16 * RT;                   // first row of this wave
    const bool full = m_base + NW * 16 * RT <= M;
    // blockIdx.y splits the N columns between workgroups when M alone gives too few of them (launcher): this one covers
    // the column steps [nbase / NS, nbase / NS + nsteps)
    const int nsteps = N / NS / (int)gridDim.y;
    const int nbase = (int)blockIdx.y * nsteps * NS;
    const int dmodel = ea.H * ea.dh;
    const bool f_resid = as_on<MASK, AS_RESID>(ea.resid != nullptr), f_dact = as_on<MASK, AS_DACT>(ea.dact != DACT_NONE);
    const bool f_act = as_on<MASK, AS_ACT>(ea.act != ACT_NONE), f_drop = as_on<MASK, AS_DROP>(ea.drop.thr != 0);
    const bool f_rowscale = as_on<MASK, AS_ROWSCALE>(ea.rowscale != nullptr), f_preout = as_on<MASK, AS_PREOUT>(ea.pre_out != nullptr);
    const bool f_qkv = as_on<MASK, AS_QKV>(ea.mode == EPI_QKV), f_addtab = as_on<MASK, AS_ADDTAB>(ea.addtab != nullptr);

    // weight DMA: instruction u = wid*DPW + t moves bytes [u*1024, +1024) of the stage; 16-byte chunk p of row r holds
    // source chunk p ^ as_swz(r) (swizzle on the source address: the LDS image of an LDS-DMA is lane-linear)
    const as_t* bsrc[DPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int o = ((wid * DPW + t) % NI) * 1024 + lane * 16;
        const int r = o / RB, p = (o % RB) >> 4;
        bsrc[t] = Bt + (size_t)(nbase + r) * ldb + ((p ^ as_swz(r)) << 3);
    }
    const size_t bstep = (size_t)NS * CS * ldb;
    auto issue = [&](int slot) {
#pragma unroll
        for (int t = 0; t < DPW; ++t) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)bsrc[t],
                                             (__attribute__((address_space(3))) void*)(smem + slot + ((wid * DPW + t) % NI) * 1024), 16, 0, 0);
            bsrc[t] += bstep;
        }
    };
    const int nstages = (nsteps + CS - 1) / CS;
#pragma unroll
    for (int st = 0; st < R - 1; ++st)
        if (st < nstages && !(dbg & 8)) issue(st * STAGE);       // dbg: ablation bits of tools/gemm_ablate.py (1 no epilogue, 2 no MFMA, 4 no LDS reads, 8 no DMA)

    // A fragments: lane (c, g) holds row 16i + c, k = 32kt + 8g .. +7
    as_v8 a[RT][KT];
    int mrow[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        mrow[i] = min(mw + 16 * i + c, M - 1);
        const as_v8* p = reinterpret_cast<const as_v8*>(A + (size_t)mrow[i] * K + g * 8);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) a[i][kt] = p[kt * 4];
    }

    if constexpr (PRO != 0) {
        float* cs = reinterpret_cast<float*>(smem + (R - 1) * STAGE);       // [2][K] coefficients: gamma | beta   or   P[b] | Q[b]
        const int bsm = min(m_base, M - 1) / ea.T;                          // PRO 2: the workgroup's rows lie in ONE sample (launcher)
        const float* c0 = PRO == 1 ? ea.ln_gamma : ea.pa_P + (size_t)bsm * K;
        const float* c1 = PRO == 1 ? ea.ln_beta : ea.pa_Q + (size_t)bsm * K;
        for (int x = tid; x < K; x += NW * 64) { cs[x] = c0[x]; cs[K + x] = c1[x]; }
        __syncthreads();
        // Every pass over the fragments unpacks them again from the packed registers, behind an opaque asm: otherwise hipcc keeps
        // all 8*KT*RT unpacked floats alive across the three passes (statistics, variance, normalise) and spills hundreds of VGPRs.
        auto unpack = [&](const as_v8& f, float (&v)[8]) {
            as_u32x4 t = __builtin_bit_cast(as_u32x4, f);
            asm volatile("" : "+v"(t));
            as_unpack<as_t, 8>(t, v);
        };
        float mean[RT], rstd[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) { mean[i] = 0.f; rstd[i] = 1.f; }
        if constexpr (PRO == 1) {              // exact two-pass statistics over the lane's 8*KT values, then over the row's 4 lanes
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float s = 0.f;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    float v[8];
                    unpack(a[i][kt], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += v[e];
                }
                s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                mean[i] = s * (1.f / K);
                float q = 0.f;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    float v[8];
                    unpack(a[i][kt], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dlt = v[e] - mean[i]; q += dlt * dlt; }
                }
                q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
                rstd[i] = rsqrtf(q * (1.f / K) + ea.ln_eps);
                const int m = mw + 16 * i + c;
                if (g == 0 && m < M && ea.ln_mean && blockIdx.y == 0) { ea.ln_mean[m] = mean[i]; ea.ln_rstd[m] = rstd[i]; }      // column splits redo the prologue; one of them writes
            }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {      // coefficients of this k slice live for the RT row tiles only
            float wv[8], bv[8];
            {
                const float* cp = cs + 32 * kt + 8 * g;
                asm volatile("" : "+v"(cp));
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(cp), w1 = *reinterpret_cast<const f32x4*>(cp + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(cp + K), b1 = *reinterpret_cast<const f32x4*>(cp + K + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wv[e] = w0[e]; wv[4 + e] = w1[e]; bv[e] = b0[e]; bv[4 + e] = b1[e]; }
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                float v[8];
                unpack(a[i][kt], v);
                as_v8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (as_t)(PRO == 1 ? (v[e] - mean[i]) * rstd[i] * wv[e] + bv[e] : v[e] * wv[e] + bv[e]);
                a[i][kt] = t;
            }
        }
        __syncthreads();                     // the coefficient slot is free again before any wave reaches step 0's DMA issue
    }

    // weight fragment reads: MFMA tile j, tile row t = c comes from staged row
    //   PAIR: 8*(c>>2) + (c&3) + 4*(j&1)   (j = 0,1)        else: 16*j + c
    // chunk 4kt + g of that row sits at position (4kt + g) ^ swz; swz does not depend on j.
    const int frow = PAIR ? 8 * (c >> 2) + (c & 3) : c;
    const int sx = g ^ as_swz(frow);
    const int off_e = frow * RB + ((sx & 3) << 4) + ((sx >> 2) << 6);
    const int off_o = frow * RB + ((sx & 3) << 4) + (((sx >> 2) ^ 1) << 6);
    constexpr int JSTRIDE = (PAIR ? 4 : 16) * RB;

    // per-row constants
    float rsc[RT];
    uint32_t rk[RT];
    int bsamp[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int m = mw + 16 * i + c;
        rsc[i] = 1.f; rk[i] = 0u; bsamp[i] = 0;
        if (f_rowscale || f_qkv) bsamp[i] = mrow[i] / ea.T;
        if (f_rowscale) rsc[i] = ea.rowscale[bsamp[i]];
        if (f_drop) rk[i] = rng_row_key(ea.drop.key, (uint32_t)m);
    }

    // per-lane base offsets of the [M, N] operands (rows clamped: out-of-range rows load row M-1 and store nothing)
    size_t eoff[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) eoff[i] = (size_t)mrow[i] * (ea.ldc ? ea.ldc : N) + GW * g;
    const TC* resid = reinterpret_cast<const TC*>(ea.resid);
    const TC* aux = reinterpret_cast<const TC*>(ea.aux);

    // Everything loaded before the loop (A fragments, per-row constants) passes through an empty asm here: hipcc then waits
    // for those loads ONCE, in front of the loop.  Otherwise its wait-counter model still sees them pending at the loop
    // header and puts `s_waitcnt vmcnt(0)` in front of the first MFMA of every step -- behind the DMA of the next stage,
    // which then never overlaps this wave's MFMAs.
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            as_u32x4 t = __builtin_bit_cast(as_u32x4, a[i][kt]);
            asm volatile("" : "+v"(t));
            a[i][kt] = __builtin_bit_cast(as_v8, t);
        }
        asm volatile("" : "+v"(rsc[i]));
    }

    as_v8 held[RT], held_pre[RT];         // HOLD: the even step's packed groups (main output, saved pre-activation)
#pragma unroll
    for (int i = 0; i < RT; ++i) { held[i] = as_v8{}; held_pre[i] = as_v8{}; }
    int slot = 0;
    for (int s = 0; s < nsteps; ++s) {
        // ---- T1: stage s has landed.  Younger operations that may stay in flight (in-order VM counter): the DMAs of
        // stages s+1 .. s+R-2 and, when the epilogue is a fixed number of stores and the row block is full, the stores
        // of the R-1 steps since its issue.
        const int sub = CS > 1 ? (s & (CS - 1)) : 0;          // column step inside the stage (CS is a power of two)
        const bool sync = CS == 1 || sub == 0;                // chunked: wait / barrier / next DMA once per stage
        const int sg = CS > 1 ? s / CS : s;                   // stage index
        if (sync) {
            if constexpr (CS > 1) {
                // two stages: stage sg's DMA was issued at the previous stage boundary, in front of that stage's CS steps of stores
                // (also with residual / act' operand loads in the epilogue: they were consumed, hence complete, before the stage boundary, so the only
                // operations younger than the DMA that can still be pending are the CS steps' stores)
                constexpr bool CCOUNT = MASK != AS_ALL && (MASK & AS_QKV) == 0;
                if (CCOUNT && full && !ea.n_valid && sg >= 1) as_wait_vm<CS * OPS>(); else as_wait_vm<0>();
            } else if constexpr (COUNTED) {
                if (full && !ea.n_valid && s >= R - 1 && s + R - 2 < nsteps) {
                    if (!hold) as_wait_vm<(R - 2) * DPW + (R - 1) * OPS>();
                    // held stores leave in pairs at the odd steps: of the two steps since stage s was requested exactly one was odd (R = 3);
                    // with R = 2 only the step before this one counts — a pair when it was odd, nothing when it was even
                    else if constexpr (R == 3) as_wait_vm<DPW + 2 * OPS>();
                    else { if (s & 1) as_wait_vm<0>(); else as_wait_vm<2 * OPS>(); }
                } else as_wait_vm<0>();
            } else {
                if (s + R - 2 < nsteps) as_wait_vm<(R - 2) * DPW>(); else as_wait_vm<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const int n0 = nbase + s * NS;
        // ---- T3: residual / act' operand loads of this step (consumed after the MFMA phase), then DMA of stage s+R-1
        // into the slot read in step s-1
        as_u32x4 rs[RT][NG], au[RT][NG];
        if (f_resid) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int q = 0; q < NG; ++q) rs[i][q] = *reinterpret_cast<const as_u32x4*>(resid + eoff[i] + n0 + 4 * GW * q);
        }
        if (f_dact) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int q = 0; q < NG; ++q) au[i][q] = __builtin_nontemporal_load(reinterpret_cast<const as_u32x4*>(aux + eoff[i] + n0 + 4 * GW * q));      // last use of a saved pre-activation
        }
        if (sync && sg + R - 1 < nstages && !(dbg & 8)) issue(slot == 0 ? (R - 1) * STAGE : slot - STAGE);
        // ---- MFMA: acc[j][i] = sum_k Bt[row(j), k] * A[16i + .., k]
        f32x4 acc[2][RT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* st = smem + slot + sub * SUB;
        if (!(dbg & 6)) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const as_v8 b = *reinterpret_cast<const as_v8*>(st + ((kt & 1) ? off_o : off_e) + (kt >> 1) * 128 + j * JSTRIDE);
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[j][i] = AS_MFMA(b, a[i][kt], acc[j][i]);
                }
            }
        } else if (!(dbg & 4)) {       // LDS reads only
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(st + ((kt & 1) ? off_o : off_e) + (kt >> 1) * 128 + j * JSTRIDE);
                    acc[j][0] += b;
                }
        } else if (!(dbg & 2)) {       // MFMA only
            const as_v8 b = a[0][0];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[j][i] = AS_MFMA(b, a[i][kt], acc[j][i]);
        } else {
#pragma unroll
            for (int i = 0; i < RT; ++i) acc[0][i][0] = (float)a[i][0][0] + (float)a[i][KT - 1][7];
        }
        if (dbg & 1) { if (acc[0][0][0] == 123.456f) C[0] = from_f<TC>(acc[1][RT - 1][2]); if (CS == 1 || sub == CS - 1) slot = slot == (R - 1) * STAGE ? 0 : slot + STAGE; continue; }
        // ---- epilogue from the accumulators: lane owns row 16i + c and, per group q, columns n0 + 4*GW*q + GW*g .. +GW-1
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int n = n0 + 4 * GW * q + GW * g;
            if (ea.n_valid && n >= ea.n_valid) continue;      // padding columns of a narrow output
            float bias[GW];
#pragma unroll
            for (int h = 0; h < GW / 4; ++h) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[4 * h + e] = bv[e];
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int m = mw + 16 * i + c;
                if (!full && m >= M) continue;
                const size_t off = eoff[i] + n0 + 4 * GW * q;          // = m*N + n
                float v[GW];
                const float bsc = (f_rowscale && ea.rowscale_bias) ? rsc[i] : 1.f;       // drop-path folded into the A operand: only the bias is scaled
                if constexpr (PAIR) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[0][i][e] + bias[e] * bsc; v[4 + e] = acc[1][i][e] + bias[4 + e] * bsc; }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[q][i][e] + bias[e] * bsc;
                }
                if (f_addtab) {
                    const float* tp = ea.addtab + (size_t)(m % ea.tab_period) * N + n;
#pragma unroll
                    for (int h = 0; h < GW / 4; ++h) {
                        float t4[4];
                        load4(tp + 4 * h, t4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * h + e] += t4[e];
                    }
                }
                if (f_preout) {
                    if constexpr (HOLDABLE) {
                        if (hold) {
                            as_v8 t;
#pragma unroll
                            for (int e = 0; e < 8; ++e) t[e] = (as_t)v[e];
                            as_t* pp = reinterpret_cast<as_t*>(ea.pre_out) + off;
                            if (s & 1) { as_st_pk(pp - NS, held_pre[i], nt_side); as_st_pk(pp, t, nt_side); }
                            else if (s == nsteps - 1) as_st_pk(pp, t, nt_side);
                            else held_pre[i] = t;
                        } else as_st_nt<GW>(reinterpret_cast<TC*>(ea.pre_out) + off, v, nt_side);
                    } else as_st_nt<GW>(reinterpret_cast<TC*>(ea.pre_out) + off, v, nt_side);
                }
                if (f_act) {
                    if (ea.act == ACT_SWISH) {
#pragma unroll
                        for (int e = 0; e < GW; ++e) v[e] = swishf_(v[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < GW; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                }
                if (f_drop) {
                    rng_apply<GW>(rk[i], (uint32_t)n, ea.drop.thr, ea.drop.scale, v);        // n is a multiple of GW
                }
                if (f_rowscale && !ea.rowscale_bias) {
#pragma unroll
                    for (int e = 0; e < GW; ++e) v[e] *= rsc[i];
                }
                if (f_dact) {
                    float x[GW];
                    as_unpack<TC, GW>(au[i][q], x);
                    if (ea.dact == DACT_SWISH) {
#pragma unroll
                        for (int e = 0; e < GW; ++e) v[e] *= dswishf_(x[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < GW; ++e) v[e] = x[e] > 0.f ? v[e] : 0.f;
                    }
                }
                if (f_resid) {
                    float x[GW];
                    as_unpack<TC, GW>(rs[i][q], x);
#pragma unroll
                    for (int e = 0; e < GW; ++e) v[e] += x[e];
                }
                if (!f_qkv) {
                    if constexpr (HOLDABLE) {
                        if (hold) {
                            as_v8 t;
#pragma unroll
                            for (int e = 0; e < 8; ++e) t[e] = (as_t)v[e];
                            as_t* cp = reinterpret_cast<as_t*>(C) + off;
                            if (s & 1) { as_st_pk(cp - NS, held[i], false); as_st_pk(cp, t, false); }
                            else if (s == nsteps - 1) as_st_pk(cp, t, false);
                            else held[i] = t;
                        } else as_st<GW>(C + off, v);
                    } else as_st<GW>(C + off, v);
                } else {      // q,k [B,H,T,dh] rows; v transposed to vt [B,H,dh,T]   (GW consecutive columns stay inside one head: dh % 8 == 0)
                    int h, part, ii;
                    if (ea.head_major) { h = n / (3 * ea.dh); const int w = n - h * 3 * ea.dh; part = w / ea.dh; ii = w - part * ea.dh; }
                    else { part = n / dmodel; const int w = n - part * dmodel; h = w / ea.dh; ii = w - h * ea.dh; }
                    const int t = m - bsamp[i] * ea.T;
                    if (part < 2) {
                        as_st<GW>(reinterpret_cast<TC*>(part == 0 ? ea.q : ea.k) + ((size_t)(bsamp[i] * ea.H + h) * ea.T + t) * ea.dh + ii, v);
                    } else {
                        TC* dst = reinterpret_cast<TC*>(ea.vt) + ((size_t)(bsamp[i] * ea.H + h) * ea.dh + ii) * ea.T + t;
#pragma unroll
                        for (int e = 0; e < GW; ++e) dst[(size_t)e * ea.T] = from_f<TC>(v[e]);
                    }
                }
            }
        }
        if (CS == 1 || sub == CS - 1) slot = slot == (R - 1) * STAGE ? 0 : slot + STAGE;
    }
    // the prologue's transformed rows, for the backward pass: stored AFTER the column loop (the fragments are still in registers), so
    // that these stores do not sit in front of the loop's counted vmcnt waits and drain beside other workgroups' loops
    if constexpr (PRO != 0) {
        if (ea.pro_out && blockIdx.y == 0) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int m = mw + 16 * i + c;
                if (m < M) {
                    as_v8* p = reinterpret_cast<as_v8*>(reinterpret_cast<as_t*>(ea.pro_out) + (size_t)m * K + g * 8);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) as_st_pk(reinterpret_cast<as_t*>(&p[kt * 4]), a[i][kt], nt_side);
                }
            }
        }
    }
}

// K = 256: 128-row workgroups, 3 per CU (768 at M = 98304 = one round of the chip).  K = 512: the A fragments take 128
// VGPRs, so only 2 workgroups fit per CU; 192-row workgroups (512 at M = 98304, again exactly one round) run as a
// 128-row pass followed by a 64-row pass.
template <typename TC, int KT, int MASK, int DBG = 0, int PRO = 0>
__global__ __launch_bounds__(KT <= 16 ? 256 : 512, KT <= 8 ? 3 : (KT <= 16 ? 2 : 1)) void gemm_nt_as_kernel(const as_t* __restrict__ A, const as_t* __restrict__ Bt, TC* __restrict__ C,
                                                                         int M, int N, int ldb, EpiArgs ea) {
    constexpr int R = KT <= 8 ? 3 : 2;
    constexpr int STAGE = AS_NS * KT * 64;
    __shared__ __attribute__((aligned(16))) char smem[R * STAGE + AS_MAXN * 4];
    float* bias_s = reinterpret_cast<float*>(smem + R * STAGE);
    // bias -> LDS (visible after the first barrier of the step loop)
    for (int n = threadIdx.x; n < N; n += (int)blockDim.x) bias_s[n] = (ea.bias && (!ea.n_valid || n < ea.n_valid)) ? ea.bias[n] : 0.f;
    if constexpr (KT <= 8) {
        if (M <= AS_SMALL_M) as_pass<TC, KT, MASK, 1, DBG, PRO>(A, Bt, C, M, N, ldb, ea, smem, bias_s, blockIdx.x * 64);      // a clip's worth of rows: 64-row workgroups (see below)
        else as_pass<TC, KT, MASK, 2, DBG, PRO>(A, Bt, C, M, N, ldb, ea, smem, bias_s, blockIdx.x * 128);
    } else if constexpr (KT >= 32) {
        // K = 1024 (config #4's 2d = 1024 operands): 16 rows per wave (the fragments of ONE row tile already take 128 VGPRs), EIGHT
        // waves = 128-row workgroups around a ring of 2 x 64 KB -> one workgroup per CU, two waves per SIMD.  Half the rows per staged
        // weight byte of the K <= 512 variants, so the LDS fragment reads bound it near half the MFMA peak.
        as_pass<TC, KT, MASK, 1, DBG, PRO, 8>(A, Bt, C, M, N, ldb, ea, smem, bias_s, blockIdx.x * 128);
    } else if (M <= AS_MID_M_K512) {
        // few rows (B = 1 inference: M = T; and M up to one round of 64-row workgroups): 64-row workgroups, one 16-row tile per wave — three times the workgroups of the 192-row
        // form and one pass instead of two sequential ones; the chip is empty either way, what counts is the latency of one workgroup
        as_pass<TC, KT, MASK, 1, DBG, PRO>(A, Bt, C, M, N, ldb, ea, smem, bias_s, blockIdx.x * 64);
    } else {
        const int m_base = blockIdx.x * 192;
        as_pass<TC, KT, MASK, 2, DBG, PRO>(A, Bt, C, M, N, ldb, ea, smem, bias_s, m_base);
        if (m_base + 128 < M) {
            __builtin_amdgcn_s_barrier();          // every wave is done reading the ring before the second pass refills it
            as_pass<TC, KT, MASK, 1, DBG, PRO>(A, Bt, C, M, N, ldb, ea, smem, bias_s, m_base + 128);
        }
    }
}

// CHUNKED form for K = 256 at large M (bf16 training shapes): ONE workgroup of 12 waves per CU owns 384 rows (256 workgroups at M = 98304);
// the weight matrix streams through two 64 KB stages of 128 output columns each, so the DMA wait + workgroup barrier + next DMA issue happen
// once per FOUR column steps (4 / 6 per GEMM instead of 16 / 24) and a stage's DMA is requested a whole stage (four steps of MFMA + stores)
// ahead.  Why: every memory instruction of a CU goes through one in-order queue; when the store stream backs it up (the kernel is
// HBM-write bound), the per-step weight DMA of the 16 KB ring arrives late and its wait + barrier stall all four waves of a workgroup every
// step — phase ablation on cold operands (tools/gemm_cold_ablate.py): loads + stores alone 31.6 us, + DMA / LDS / MFMA 40.7 us, i.e. the
// compute side does not hide under the store stream.  With the wait amortised over four steps the waves drift apart and compute overlaps stores.
// K = 512: the A fragments take 128 registers, 8 waves per CU: stages of 64 columns (2 steps), a 256-row pass (32 rows per wave) followed by a
// 128-row pass (16 rows per wave) over the same 384 rows-per-workgroup split.
#define AS_CHUNK_ROWS 384
template <int KT> struct AsChunk { static constexpr int NW = KT <= 8 ? 12 : 8, CS = KT <= 8 ? 4 : 2; };
template <typename TC, int KT, int MASK, int PRO = 0>
__global__ __launch_bounds__(AsChunk<KT>::NW * 64, 1) void gemm_nt_as_chunk_kernel(const as_t* __restrict__ A, const as_t* __restrict__ Bt, TC* __restrict__ C, int M, int N, int ldb, EpiArgs ea) {
    extern __shared__ __attribute__((aligned(16))) char csm[];
    constexpr int NW = AsChunk<KT>::NW, CS = AsChunk<KT>::CS;
    constexpr int STAGE = AS_NS * KT * 64 * CS;
    float* bias_s = reinterpret_cast<float*>(csm + 2 * STAGE);
    for (int n = threadIdx.x; n < N; n += (int)blockDim.x) bias_s[n] = (ea.bias && (!ea.n_valid || n < ea.n_valid)) ? ea.bias[n] : 0.f;
    const int m_base = blockIdx.x * AS_CHUNK_ROWS;
    if constexpr (KT <= 8) {
        as_pass<TC, KT, MASK, 2, 0, PRO, NW, CS>(A, Bt, C, M, N, ldb, ea, csm, bias_s, m_base);
    } else {
        as_pass<TC, KT, MASK, 2, 0, PRO, NW, CS>(A, Bt, C, M, N, ldb, ea, csm, bias_s, m_base);
        if (m_base + 256 < M) {
            __builtin_amdgcn_s_barrier();          // every wave is done reading the stages before the second pass refills them
            as_pass<TC, KT, MASK, 1, 0, PRO, NW, CS>(A, Bt, C, M, N, ldb, ea, csm, bias_s, m_base + 256);
        }
    }
}
template <typename KF> static bool as_chunk_prepare(KF kernel, int bytes) {      // dynamic LDS above 64 KB needs the attribute once per kernel
    static std::map<const void*, bool> done;
    const void* key = reinterpret_cast<const void*>(kernel);
    auto it = done.find(key);
    if (it != done.end()) return it->second;
    const bool ok = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    done[key] = ok;
    return ok;
}

static int as_mask_of(const EpiArgs& ea) {
    return (ea.resid ? AS_RESID : 0) | (ea.dact != DACT_NONE ? AS_DACT : 0) | (ea.act != ACT_NONE ? AS_ACT : 0) | (ea.drop.thr ? AS_DROP : 0) |
           (ea.rowscale ? AS_ROWSCALE : 0) | (ea.pre_out ? AS_PREOUT : 0) | (ea.mode == EPI_QKV ? AS_QKV : 0) | (ea.addtab ? AS_ADDTAB : 0);
}

// the instantiation run_as picks for a feature mask (bf16 C: the listed combinations, else AS_ALL; f32 C: 0 or AS_ALL)
static int as_inst_mask(bool c_bf16, int mask, int K = 256) {
    if (!c_bf16) return mask == 0 ? 0 : AS_ALL;
    if (K == 128) return (mask == AS_DACT || mask == (AS_DACT | AS_DROP)) ? mask : AS_ALL;      // K = 128: the classifier's dgrad only
    switch (mask) {
        case 0: case AS_RESID: case AS_RESID | AS_ROWSCALE: case AS_ROWSCALE: case AS_RESID | AS_DROP: case AS_ACT | AS_PREOUT: case AS_ACT | AS_PREOUT | AS_DROP:
        case AS_ACT | AS_DROP: case AS_ACT: case AS_DACT: case AS_DACT | AS_DROP: case AS_QKV: case AS_ADDTAB: return mask;
        default: return AS_ALL;
    }
}

#define AS_LAUNCH_PLAIN(MASK, PRO) hipLaunchKernelGGL((gemm_nt_as_kernel<TC, KT, MASK, 0, PRO>), grid, block, 0, s, (const as_t*)A, (const as_t*)Bt, (TC*)C, M, N, ldb, ea)
#ifdef AS_F16
#define AS_LAUNCH2(MASK, PRO) AS_LAUNCH_PLAIN(MASK, PRO)
#else
// K = 256, bf16 C, large M, whole stages: the chunked kernel (as_flags bit 4)
#define AS_LAUNCH2(MASK, PRO)                                                                                                         \
    do {                                                                                                                              \
        if constexpr ((KT == 8 || KT == 16) && std::is_same<TC, as_t>::value && (MASK) != AS_ALL) {                                   \
            constexpr int CB = 2 * AS_NS * KT * 64 * AsChunk<KT>::CS + AS_MAXN * 4;                                                   \
            if (chunked && as_chunk_prepare(gemm_nt_as_chunk_kernel<TC, KT, MASK, PRO>, CB)) {                                        \
                hipLaunchKernelGGL((gemm_nt_as_chunk_kernel<TC, KT, MASK, PRO>), dim3((M + AS_CHUNK_ROWS - 1) / AS_CHUNK_ROWS), dim3(AsChunk<KT>::NW * 64), CB, s, \
                                   (const as_t*)A, (const as_t*)Bt, (TC*)C, M, N, ldb, ea);                                           \
                break;                                                                                                                \
            }                                                                                                                         \
        }                                                                                                                             \
        AS_LAUNCH_PLAIN(MASK, PRO);                                                                                                   \
    } while (0)
#endif
#define AS_LAUNCH(MASK) AS_LAUNCH2(MASK, 0)
// library default of EpiArgs.as_flags (bit 0 paired half-line stores, bit 1 non-temporal side outputs); ISHARA_AS_FLAGS overrides (A/B runs)
#ifdef AS_F16
extern int g_as_flags_override;
#else
int g_as_flags_override = -1;       // tests / A-B runs inside one process (ishara_debug_set_as_flags)
#endif
static int as_default_flags() {
    static const int v = getenv("ISHARA_AS_FLAGS") ? atoi(getenv("ISHARA_AS_FLAGS")) : 51;     // 1 paired stores | 2 nt side outputs | 16 chunked K = 256 | 32 chunked K = 512
    return g_as_flags_override >= 0 ? g_as_flags_override : v;
}
static bool as_chunk_applies(int flags, int M, int N, int K, const EpiArgs& ea) {      // (bf16 C checked by the caller); bit 4: K = 256, bit 5: K = 512
    const int cs = K == 256 ? 4 : 2;
    // PSA prologue: a workgroup's rows (both passes at K = 512: 256 + 128) must lie inside one sample
    return (K == 256 ? (flags & 16) : (K == 512 && (flags & 32))) && M >= 128 * AS_CHUNK_ROWS && (N / AS_NS) % cs == 0 && N >= 128 && !ea.n_valid && !ea.dbg &&
           (!ea.pa_P || (ea.T > 0 && ea.T % AS_CHUNK_ROWS == 0));
}
template <typename TC, int KT>
static int run_as(const void* A, const void* Bt, void* C, int M, int N, int ldb, const EpiArgs& ea_in, hipStream_t s) {
    EpiArgs ea = ea_in;
    if (ea.as_flags < 0) ea.as_flags = as_default_flags();
    const int BR = KT <= 8 ? (M <= AS_SMALL_M ? 64 : 128) : (KT <= 16 ? (M <= AS_MID_M_K512 ? 64 : 192) : 128);     // rows per workgroup
    // few rows (config #4 at small batches: M / 192 = 171 workgroups for 512 slots): split the columns 2- or 4-way; every
    // workgroup then loads its A rows again, which is cheap exactly when M is small
    const int gx = (M + BR - 1) / BR, slots = KT <= 8 ? 768 : (KT <= 16 ? 512 : 256);
    int gy = 1;
    // (a clip's worth of rows: up to 8 splits of at least 64 columns — what counts there is the number of sequential column steps of one workgroup)
    const int max_gy = M <= AS_SMALL_M ? 16 : 4, min_cols = M <= AS_SMALL_M ? 32 : 128;
    while (gy < max_gy && gx * gy * 2 <= slots && (N / AS_NS) % (gy * 2) == 0 && N / (gy * 2) >= min_cols) gy *= 2;
    const dim3 grid(gx, gy), block(KT <= 16 ? 256 : 512);
    const int mask = as_mask_of(ea);
    // chunked form (K = 256): whole 128-column stages, rows enough to give every CU a 384-row workgroup, no column split, no narrow output
    const bool chunked = (KT == 8 || KT == 16) && as_chunk_applies(ea.as_flags, M, N, KT * 32, ea) && gy == 1;
    (void)chunked;
    if constexpr (is_16b_t<TC>::value && KT >= 8) {
#define AS_PRO(MASK, PRO) AS_LAUNCH2(MASK, PRO)
        if (ea.ln_gamma) {           // LayerNorm prologue: the GEMMs that consume a LayerNorm output (FFN expand, QKV, conv-module expand)
            switch (mask) {
                case 0: AS_PRO(0, 1); break;
                case AS_ACT | AS_PREOUT: AS_PRO(AS_ACT | AS_PREOUT, 1); break;
                case AS_ACT | AS_PREOUT | AS_DROP: AS_PRO(AS_ACT | AS_PREOUT | AS_DROP, 1); break;
                case AS_QKV: AS_PRO(AS_QKV, 1); break;
                default: ishara_set_error("gemm_nt_as: LayerNorm prologue with epilogue mask %d is not compiled", mask); return -1;
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
        if (ea.pa_P) {               // per-sample affine prologue: the project GEMM of a Conv1DBlock
            switch (mask) {
                case AS_RESID: AS_PRO(AS_RESID, 2); break;
                case AS_RESID | AS_ROWSCALE: AS_PRO(AS_RESID | AS_ROWSCALE, 2); break;
                default: ishara_set_error("gemm_nt_as: affine prologue with epilogue mask %d is not compiled", mask); return -1;
            }
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
#undef AS_PRO
    }
    if constexpr (KT == 4) {
        if (mask == AS_DACT) AS_LAUNCH(AS_DACT); else if (mask == (AS_DACT | AS_DROP)) AS_LAUNCH(AS_DACT | AS_DROP); else AS_LAUNCH(AS_ALL);
    }
#ifdef AS_F16
    else if constexpr (is_16b_t<TC>::value) {      // fp16 = inference: the forward passes' combinations, everything else through AS_ALL
        switch (mask) {
            case 0: AS_LAUNCH(0); break;
            case AS_RESID: AS_LAUNCH(AS_RESID); break;
            case AS_ACT | AS_PREOUT: AS_LAUNCH(AS_ACT | AS_PREOUT); break;
            case AS_ACT: AS_LAUNCH(AS_ACT); break;
            case AS_QKV: AS_LAUNCH(AS_QKV); break;
            case AS_ADDTAB: AS_LAUNCH(AS_ADDTAB); break;
            default: AS_LAUNCH(AS_ALL); break;
        }
    }
#else
    else if constexpr (is_16b_t<TC>::value) {
        // the feature combinations the encoder's forward / backward passes use (model.hip), compiled without the others
        switch (mask) {
            case 0:
                if (ea.dbg) hipLaunchKernelGGL((gemm_nt_as_kernel<TC, KT, 0, 1>), grid, block, 0, s, (const as_t*)A, (const as_t*)Bt, (TC*)C, M, N, ldb, ea);
                else AS_LAUNCH(0);
                break;
            case AS_RESID: AS_LAUNCH(AS_RESID); break;                                           // W2 / Wb / Wp eval, dgrad + skip gradient
            case AS_RESID | AS_ROWSCALE: AS_LAUNCH(AS_RESID | AS_ROWSCALE); break;               // conv block W2 with drop-path
            case AS_ROWSCALE: AS_LAUNCH(AS_ROWSCALE); break;                                     // its dgrad: (g W2^T) * drop-path scale
            case AS_RESID | AS_DROP: AS_LAUNCH(AS_RESID | AS_DROP); break;                       // Wb / Wp with output dropout
            case AS_ACT | AS_PREOUT: AS_LAUNCH(AS_ACT | AS_PREOUT); break;                       // FFN Wa eval
            case AS_ACT | AS_PREOUT | AS_DROP: AS_LAUNCH(AS_ACT | AS_PREOUT | AS_DROP); break;   // FFN Wa training
            case AS_ACT | AS_DROP: AS_LAUNCH(AS_ACT | AS_DROP); break;                           // head
            case AS_ACT: AS_LAUNCH(AS_ACT); break;
            case AS_DACT: AS_LAUNCH(AS_DACT); break;                                             // dgrad through swish / relu
            case AS_DACT | AS_DROP: AS_LAUNCH(AS_DACT | AS_DROP); break;
            case AS_QKV: AS_LAUNCH(AS_QKV); break;
            case AS_ADDTAB: AS_LAUNCH(AS_ADDTAB); break;
            default: AS_LAUNCH(AS_ALL); break;
        }
    }
#endif
    else {
        if (mask == 0) AS_LAUNCH(0); else AS_LAUNCH(AS_ALL);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
#undef AS_LAUNCH

#ifndef AS_F16
bool gemm_nt_as_prologue_ok(int dtA, int dtM, int dtC, int M, int N, int K, int ldb, const EpiArgs& ea) {
    if (!dt_is16(dtA) || dtM != dtA || dtC != dtA || (K != 256 && K != 512 && !(K == 1024 && dtA == DT_BF16)) || !gemm_nt_as_applicable(dtC, M, N, K, ldb, ea)) return false;
    if (ldb % 64 != 0 || g_force_regstage) return false;
    const int BR = K == 256 ? (M <= AS_SMALL_M ? 64 : 128) : (K == 512 ? (M <= AS_MID_M_K512 ? 64 : 192) : 128), gx = (M + BR - 1) / BR, slots = K == 256 ? 768 : (K == 512 ? 512 : 256);
    // (when the launcher splits the columns every split redoes the prologue on its rows and the blockIdx.y == 0 split writes the side outputs)
    (void)gx; (void)slots;
    const int mask = as_mask_of(ea);
    if (ea.ln_gamma) return mask == 0 || mask == (AS_ACT | AS_PREOUT) || mask == (AS_ACT | AS_PREOUT | AS_DROP) || mask == AS_QKV;
    if (ea.pa_P) return ea.T > 0 && ea.T % BR == 0 && (mask == AS_RESID || mask == (AS_RESID | AS_ROWSCALE));   // a workgroup's rows inside one sample
    return false;
}

bool gemm_nt_as_applicable(int dtC, int M, int N, int K, int ldb, const EpiArgs& ea) {
    if (K != 256 && K != 512 && !((K == 128 || K == 1024) && dtC == DT_BF16)) return false;      // fp16 operands (dtC DT_F16 / DT_F32 from gemm_as_f16.hip): K 256 / 512
    if (N % AS_NS != 0 || N > AS_MAXN || ldb < K || ldb % 8 != 0 || M < 1) return false;
    if (ea.mode == EPI_QKV && (ea.dh % 8 != 0 || ea.T % 8 != 0)) return false;
    return true;
}

// profiler key = the rocprof kernel name of the instantiation launch_gemm_nt_as runs
const char* gemm_nt_as_name(int dtC, int K, const EpiArgs& ea, int M, int N) {
    static std::map<int, std::string> names;
    const int inst = as_inst_mask(dtC == DT_BF16, as_mask_of(ea), K);
    const int pro = ea.ln_gamma ? 1 : (ea.pa_P ? 2 : 0);
    const bool chunk = dtC == DT_BF16 && inst != AS_ALL && as_chunk_applies(ea.as_flags < 0 ? as_default_flags() : ea.as_flags, M, N, K, ea);
    const int id = (dtC == DT_BF16 ? 0 : 1 << 20) | (pro << 22) | (chunk ? 1 << 24 : 0) | (K << 8) | inst;
    auto it = names.find(id);
    if (it == names.end()) {
        char buf[96];
        if (chunk) snprintf(buf, sizeof buf, "gemm_nt_as_chunk_kernel<bf16,%d,%d,%d>", K / 32, pro ? as_mask_of(ea) : inst, pro);
        else if (pro) snprintf(buf, sizeof buf, "gemm_nt_as_kernel<%s,%d,%d,0,%d>", dtC == DT_BF16 ? "bf16" : "f32", K / 32, as_mask_of(ea), pro);
        else snprintf(buf, sizeof buf, "gemm_nt_as_kernel<%s,%d,%d,0>", dtC == DT_BF16 ? "bf16" : "f32", K / 32, inst);
        it = names.emplace(id, buf).first;
    }
    return it->second.c_str();
}

#endif   // !AS_F16

// returns 1 when the shape is not one this kernel takes (caller falls through to the tile kernels)
#ifdef AS_F16
int launch_gemm_nt_as_f16(int dtC, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s) {
    if ((K != 256 && K != 512) || !gemm_nt_as_applicable(dtC, M, N, K, ldb, ea)) return 1;
    if (dtC == DT_F16) return K == 256 ? run_as<f16, 8>(A, Bt, C, M, N, ldb, ea, s) : run_as<f16, 16>(A, Bt, C, M, N, ldb, ea, s);
    return K == 256 ? run_as<float, 8>(A, Bt, C, M, N, ldb, ea, s) : run_as<float, 16>(A, Bt, C, M, N, ldb, ea, s);
}
#else
int launch_gemm_nt_as(int dtC, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s) {
    if (!gemm_nt_as_applicable(dtC, M, N, K, ldb, ea)) return 1;
    if (dtC == DT_BF16) return K == 128 ? run_as<bf16, 4>(A, Bt, C, M, N, ldb, ea, s) : (K == 256 ? run_as<bf16, 8>(A, Bt, C, M, N, ldb, ea, s) :
                               (K == 512 ? run_as<bf16, 16>(A, Bt, C, M, N, ldb, ea, s) : run_as<bf16, 32>(A, Bt, C, M, N, ldb, ea, s)));
    return K == 256 ? run_as<float, 8>(A, Bt, C, M, N, ldb, ea, s) : run_as<float, 16>(A, Bt, C, M, N, ldb, ea, s);
}
#endif
