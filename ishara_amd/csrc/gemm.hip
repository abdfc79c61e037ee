// MFMA GEMMs for the dense projections of the Ishara encoder (gfx950 / CDNA4).
//
//   gemm_nt : C[M,N]  = epi( op(A)[M,K] . Bt[N,K]^T )       forward + dgrad
//   gemm_tn : dW[K,N] += opA(A)[M,K]^T . opB(B)[M,N]         wgrad (+ bias grad), split over M
//
// M = batch*frames is huge (98,304 for B256,T384) while N,K <= 768, so both kernels
// stream the activation operand once from HBM and keep the weight tile L2-resident.
// One 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each wave a 64x64
// sub-tile = 4x4 MFMA 16x16 accumulators.  bf16 mode uses v_mfma_f32_16x16x32_bf16,
// f32 mode uses the exact-f32 v_mfma_f32_16x16x4_f32 (same tile structure, K tile is
// 128 bytes per row in both: 64 bf16 / 32 f32).  Operands are register-staged
// (global -> VGPR -> transform -> LDS, issue-early/write-late) into a double-buffered,
// XOR-swizzled LDS image; the accumulators leave through an fp32 LDS stage so that
// the fused epilogue (bias, PE table, activation, dropout, drop-path, act', residual,
// QKV head split with V transposed) runs on whole 8-element row chunks with 16-byte
// coalesced global accesses.
#include <type_traits>
#include "kernels.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

template <typename TM> struct MmaCfg;
template <> struct MmaCfg<bf16>  { static constexpr int EPC = 8; static constexpr int BK = 64; };
template <> struct MmaCfg<f16>   { static constexpr int EPC = 8; static constexpr int BK = 64; };
template <> struct MmaCfg<float> { static constexpr int EPC = 4; static constexpr int BK = 32; };

// ---------------------------------------------------------------------------------
// operand transforms
// ---------------------------------------------------------------------------------
template <int N>
DEVI void apply_op(int op, float (&v)[N], int row, int col, const OpArgs& a) {
    switch (op) {
        case OP_SWISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = swishf_(v[e]);
            break;
        case OP_COLAFFINE:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * a.c1[col + e] + a.c0[col + e];
            break;
        case OP_ROWSCALE: {
            const float s = a.rs[row / a.T];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] *= s;
        } break;
        case OP_DROPMASK:
            if (a.drop.thr) {
                const uint32_t rk = rng_row_key(a.drop.key, (uint32_t)row);
#pragma unroll
                for (int e = 0; e < N; ++e) v[e] = rng_keep(rk, (uint32_t)(col + e), a.drop.thr) ? v[e] * a.drop.scale : 0.f;
            }
            break;
        default: break;
    }
}

// load N (4 or 8) consecutive elements of row `row` starting at column `col`, zero filled
// outside [rows x cols]; vec_ok = row stride and base are 16-byte aligned.
// load N (4 or 8) consecutive elements of row `row` starting at column `col`, zero filled outside
// [rows x cols].  Row strides are multiples of 16 bytes (launchers enforce cols % 4 == 0 for f32,
// cols % 8 == 0 for bf16), so a chunk is made of whole 16-byte pieces: no element-granular tail.
template <typename T, int N>
DEVI void load_row_chunk(const T* __restrict__ base, int ld, int rows, int cols, int row, int col,
                         bool /*vec_ok*/, float (&v)[N]) {
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = 0.f;
    if (row >= rows) return;
    const T* p = base + (size_t)row * ld + col;
    if constexpr (is_16b_t<T>::value) {
        if (col + N <= cols) {
            if constexpr (N == 8) load8(p, v);
            else load4g(p, v);
        }
    } else {
#pragma unroll
        for (int h = 0; h < N / 4; ++h) {
            if (col + 4 * h + 4 <= cols) {
                const float4 x = *reinterpret_cast<const float4*>(p + 4 * h);
                v[4 * h] = x.x; v[4 * h + 1] = x.y; v[4 * h + 2] = x.z; v[4 * h + 3] = x.w;
            }
        }
    }
}

DEVI uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 t; t[0] = (bf16)lo; t[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, t);
}
DEVI uint32_t pack_f16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    f16x2 t; t[0] = (f16)lo; t[1] = (f16)hi;
    return __builtin_bit_cast(uint32_t, t);
}
template <typename TM, int N> DEVI u32x4 pack_chunk(const float (&v)[N]) {
    u32x4 r;
    if constexpr (std::is_same<TM, f16>::value) {
        r.x = pack_f16x2(v[0], v[1]); r.y = pack_f16x2(v[2], v[3]);
        r.z = pack_f16x2(v[4], v[5]); r.w = pack_f16x2(v[6], v[7]);
    } else if constexpr (is_bf16_t<TM>::value) {
        r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]);
        r.z = pack_bf16x2(v[4], v[5]); r.w = pack_bf16x2(v[6], v[7]);
    } else {
        r.x = __float_as_uint(v[0]); r.y = __float_as_uint(v[1]);
        r.z = __float_as_uint(v[2]); r.w = __float_as_uint(v[3]);
    }
    return r;
}

// ---------------------------------------------------------------------------------
// MFMA over one LDS tile pair.  ldsA/ldsB: [128 rows][128 bytes], 16-byte slots XOR
// swizzled by SWZ(row).  acc[i][j] = 16x16 tile (rows wr*64+16i.., cols wc*64+16j..).
// ---------------------------------------------------------------------------------
template <int SW> DEVI int swz(int row) { return SW == 0 ? (row & 7) : ((row ^ (row >> 3)) & 7); }

template <typename TM, int SW>
DEVI void mma_tile(const char* ldsA, const char* ldsB, int wr, int wc, int lane, f32x4 (&acc)[4][4]) {
    const int r = lane & 15, g = lane >> 4;
    if constexpr (std::is_same<TM, f16>::value) {      // same tile layout as bf16, the f16 MFMA
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + 16 * i + r;
                a[i] = *reinterpret_cast<const f16x8*>(ldsA + row * 128 + (((4 * s + g) ^ swz<SW>(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + 16 * j + r;
                b[j] = *reinterpret_cast<const f16x8*>(ldsB + row * 128 + (((4 * s + g) ^ swz<SW>(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    } else if constexpr (is_bf16_t<TM>::value) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + 16 * i + r;
                a[i] = *reinterpret_cast<const bf16x8*>(ldsA + row * 128 + (((4 * s + g) ^ swz<SW>(row)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + 16 * j + r;
                b[j] = *reinterpret_cast<const bf16x8*>(ldsB + row * 128 + (((4 * s + g) ^ swz<SW>(row)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 64 + 16 * i + r;
                a[i] = *reinterpret_cast<const float*>(ldsA + row * 128 + ((s ^ swz<SW>(row)) << 4) + g * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wc * 64 + 16 * j + r;
                b[j] = *reinterpret_cast<const float*>(ldsB + row * 128 + ((s ^ swz<SW>(row)) << 4) + g * 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

#include "gemm_epi.h"

// stage one K tile of A (transformed) and Bt into registers: 4 x 16-byte chunks each per thread
template <typename TA, typename TM, int OP>
DEVI void nt_gload(const TA* __restrict__ A, const TM* __restrict__ Bt, int M, int K, int ldb, bool a_vec_ok,
                   int m0, int n0, int kt, int tid, const OpArgs& oa, u32x4 (&ra)[4], u32x4 (&rb)[4]) {
    constexpr int EPC = MmaCfg<TM>::EPC, BK = MmaCfg<TM>::BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 3, slot = c & 7;
        const int k = kt * BK + slot * EPC;
        float v[EPC];
        load_row_chunk<TA, EPC>(A, K, M, K, m0 + row, k, a_vec_ok, v);
        if (OP != OP_NONE) {
            apply_op<EPC>(OP, v, m0 + row, k, oa);
            if (m0 + row >= M) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = 0.f;
            } else if (k + EPC > K) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) if (k + e >= K) v[e] = 0.f;
            }
        }
        ra[i] = pack_chunk<TM, EPC>(v);
        rb[i] = *reinterpret_cast<const u32x4*>(Bt + (size_t)(n0 + row) * ldb + k);
    }
}
DEVI void nt_lstore(char* sa, int tid, const u32x4 (&ra)[4], const u32x4 (&rb)[4]) {
    char* sb = sa + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 3, slot = c & 7;
        const int off = row * 128 + ((slot ^ (row & 7)) << 4);
        *reinterpret_cast<u32x4*>(sa + off) = ra[i];
        *reinterpret_cast<u32x4*>(sb + off) = rb[i];
    }
}

// ---------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------
template <typename TA, typename TM, typename TC, int OP>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const TA* __restrict__ A, const TM* __restrict__ Bt, TC* __restrict__ C,
                                                      int M, int N, int K, int ldb, int a_vec_ok, OpArgs oa, EpiArgs ea) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    constexpr int EPC = MmaCfg<TM>::EPC, BK = MmaCfg<TM>::BK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int nMt = (M + 127) >> 7, nNt = (N + 127) >> 7;
    int mt, nt;
    {   // XCD-aware mapping: blocks b, b+8 share an XCD (L2); keep one A row-panel's N tiles together
        const int id = blockIdx.x;
        if ((nMt & 7) == 0) { const int xcd = id & 7, local = id >> 3; mt = (local / nNt) * 8 + xcd; nt = local % nNt; }
        else { mt = id / nNt; nt = id % nNt; }
    }
    const int m0 = mt << 7, n0 = nt << 7;
    const int nk = (K + BK - 1) / BK;

    u32x4 ra[4], rb[4];
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    nt_gload<TA, TM, OP>(A, Bt, M, K, ldb, a_vec_ok != 0, m0, n0, 0, tid, oa, ra, rb);
    nt_lstore(smem, tid, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) nt_gload<TA, TM, OP>(A, Bt, M, K, ldb, a_vec_ok != 0, m0, n0, kt + 1, tid, oa, ra, rb);
        const char* sa = smem + (kt & 1) * 32768;
        mma_tile<TM, 0>(sa, sa + 16384, wr, wc, lane, acc);
        if (more) nt_lstore(smem + ((kt + 1) & 1) * 32768, tid, ra, rb);
        __syncthreads();
    }

    // ---- epilogue through an fp32 LDS stage, 64 rows per pass ----
    float* stage = reinterpret_cast<float*>(smem);   // [64][132]
    constexpr int SLD = 132;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        EpiRows<TC, 4> er;       // thread -> columns (tid&15)*8.., rows (tid>>4) + 16q of this 64-row pass
        er.prefetch(m0 + 64 * p + (tid >> 4), 16, n0 + (tid & 15) * 8, M, N, ea);
        if (wr == p) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(16 * i + 4 * (lane >> 4) + r) * SLD + wc * 64 + 16 * j + (lane & 15)] = acc[i][j][r];
        }
        __syncthreads();
        er.finish(stage + (tid >> 4) * SLD + (tid & 15) * 8, 16 * SLD, M, N, ea, C);
        if (ea.mode == EPI_QKV) {   // V columns: write transposed vt[b,h,i,t], 8 consecutive t per store
            const int d = ea.H * ea.dh;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int c = tid + 256 * qq, col = c & 127, rg = c >> 7;
                const int n = n0 + col, mb = m0 + 64 * p + 8 * rg;
                if (n < N && mb < M) {
                    int h, part, i;
                    if (ea.head_major) { h = n / (3 * ea.dh); const int w = n - h * 3 * ea.dh; part = w / ea.dh; i = w - part * ea.dh; }
                    else { part = n / d; const int w = n - part * d; h = w / ea.dh; i = w - h * ea.dh; }
                    if (part == 2) {
                        float v[8];
                        const float bias = ea.bias ? ea.bias[n] : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = stage[(8 * rg + e) * SLD + col] + bias;
                        const int b = mb / ea.T, t = mb - b * ea.T;
                        TC* dst = reinterpret_cast<TC*>(ea.vt) + ((size_t)(b * ea.H + h) * ea.dh + i) * ea.T + t;
                        store8_n(dst, v, min(8, M - mb));
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// NT kernel, LDS-DMA pipelined (the common case: A and Bt both of the MFMA type, no operand
// transform, K a multiple of the K tile).  Tile 64(M) x 128(N); 4 waves (2x2), wave tile
// 32x64 = 2x4 MFMA 16x16.  A 3-deep LDS ring is filled by global_load_lds_dwordx4 — no VGPR
// staging, no ds_write; the XOR swizzle is applied to the per-lane SOURCE address because
// the LDS side of an LDS-DMA is lane-linear — so two K tiles stay in flight while a third is
// consumed: one raw s_barrier and one counted s_waitcnt vmcnt(6) per K tile.  Each wave then
// drains its accumulators through a private fp32 LDS stage (no block barrier) into the fused
// epilogue with 16-byte coalesced stores.
// ---------------------------------------------------------------------------------
#define GL_STAGE 12288          // A 64 rows x 64 B + B 128 rows x 64 B   (K tile = 32 bf16 / 16 f32)
#define GL_NSTAGE 4
DEVI int gl_f(int row) { return ((row >> 3) & 1) << 1; }      // conflict-free slot swizzle for 64-byte rows (brute-forced)

// per-lane state of the LDS-DMA NT kernel: everything the K loop needs is computed once; per K tile the loop only
// bumps three global pointers by one tile and uses compile-time stage offsets (the loop is unrolled over the ring)
template <typename TM>
struct GlState {
    const TM* pa;            // this lane's 16-byte source of the A piece (row clamp + swizzle applied)
    const TM* pb[2];         // ... of the two B pieces
    int offA[2], offB[4];    // byte offsets of the MFMA fragment reads inside a stage (k-step 0)
};

template <typename TM>
DEVI void gl_init(GlState<TM>& g, const TM* __restrict__ A, const TM* __restrict__ Bt, int M, int K, int ldb, int m0, int n0,
                  int wid, int wr, int wc, int lane) {
    constexpr int EPC = MmaCfg<TM>::EPC;
    const int r = lane >> 2, sp = lane & 3;
    const int kcol = (sp ^ gl_f(r)) * EPC;
    g.pa = A + (size_t)min(m0 + 16 * wid + r, M - 1) * K + kcol;
#pragma unroll
    for (int u = 0; u < 2; ++u) g.pb[u] = Bt + (size_t)(n0 + 16 * (wid + 4 * u) + r) * ldb + kcol;
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int row = wr * 32 + 16 * i + fr; g.offA[i] = row * 64 + (is_bf16_t<TM>::value ? ((fg ^ gl_f(row)) << 4) : (gl_f(row) << 4) + fg * 4); }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int row = wc * 64 + 16 * j + fr; g.offB[j] = 4096 + row * 64 + (is_bf16_t<TM>::value ? ((fg ^ gl_f(row)) << 4) : (gl_f(row) << 4) + fg * 4); }
}

template <typename TM>
DEVI void gl_issue(GlState<TM>& g, char* stage, int wid) {
    constexpr int BKH = MmaCfg<TM>::BK / 2;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g.pa,
                                     (__attribute__((address_space(3))) void*)(stage + wid * 1024), 16, 0, 0);
#pragma unroll
    for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g.pb[u],
                                         (__attribute__((address_space(3))) void*)(stage + 4096 + (wid + 4 * u) * 1024), 16, 0, 0);
    g.pa += BKH; g.pb[0] += BKH; g.pb[1] += BKH;
}

template <typename TM>
DEVI void gl_mma(const GlState<TM>& g, const char* st, f32x4 (&acc)[2][4]) {
    if constexpr (is_bf16_t<TM>::value) {
        bf16x8 a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(st + g.offA[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(st + g.offB[j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {          // f32: 4 k-steps of 4 per 64-byte row; slot s -> physical slot s ^ f(row): XOR the precomputed f-slot
            float a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float*>(st + (g.offA[i] ^ (s << 4)));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float*>(st + (g.offB[j] ^ (s << 4)));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

template <typename TM, typename TC>
__global__ __launch_bounds__(256, 3) void gemm_nt_glds_kernel(const TM* __restrict__ A, const TM* __restrict__ Bt, TC* __restrict__ C,
                                                           int M, int N, int K, int ldb, EpiArgs ea) {
    __shared__ __attribute__((aligned(16))) char smem[GL_NSTAGE * GL_STAGE];
    constexpr int BK = MmaCfg<TM>::BK / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nMt = (M + 63) >> 6, nNt = (N + 127) >> 7;
    int mt, nt;
    {
        const int id = blockIdx.x;
        if ((nMt & 7) == 0) { const int xcd = id & 7, local = id >> 3; mt = (local / nNt) * 8 + xcd; nt = local % nNt; }
        else { mt = id / nNt; nt = id % nNt; }
    }
    const int m0 = mt << 6, n0 = nt << 7;
    const int nk = K / BK;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool ld_on = !(ea.dbg & 4), mma_on = !(ea.dbg & 2);
    // epilogue operands (bias, residual, act' input, drop-path scale) are requested first: their latency hides under the K loop
    const int mw = m0 + wr * 32, nw = n0 + wc * 64;
    EpiRows<TC, 4> er;       // thread -> columns (lane&7)*8.., rows (lane>>3) + 8q
    er.prefetch(mw + (lane >> 3), 8, nw + (lane & 7) * 8, M, N, ea);
    GlState<TM> gs;
    gl_init<TM>(gs, A, Bt, M, K, ldb, m0, n0, wid, wr, wc, lane);
#pragma unroll
    for (int st = 0; st < GL_NSTAGE - 1; ++st)
        if (st < nk && ld_on) gl_issue<TM>(gs, smem + st * GL_STAGE, wid);
    for (int kt0 = 0; kt0 < nk; kt0 += GL_NSTAGE) {
#pragma unroll
        for (int u = 0; u < GL_NSTAGE; ++u) {      // stage index u is a compile-time constant: LDS offsets fold into the instructions
            const int kt = kt0 + u;
            if (kt < nk) {
                // 3 DMA per wave per K tile; up to two later tiles stay in flight across the barrier
                const int ahead = min(nk - 1 - kt, GL_NSTAGE - 2);
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();          // tile kt landed for every wave; tile kt-1 fully consumed
                if (kt + GL_NSTAGE - 1 < nk && ld_on) gl_issue<TM>(gs, smem + ((u + GL_NSTAGE - 1) % GL_NSTAGE) * GL_STAGE, wid);
                if (mma_on) gl_mma<TM>(gs, smem + u * GL_STAGE, acc);
            }
        }
    }
    __syncthreads();     // ring is free: reuse it as four wave-private fp32 stages
    if (ea.dbg & 1) { if (acc[0][0][0] == 123.456f) C[0] = from_f<TC>(acc[1][3][2]); return; }

    constexpr int SLD = 68;
    float* stage = reinterpret_cast<float*>(smem) + wid * (32 * SLD);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                stage[(16 * i + 4 * (lane >> 4) + r) * SLD + 16 * j + (lane & 15)] = acc[i][j][r];
    er.finish(stage + (lane >> 3) * SLD + (lane & 7) * 8, 8 * SLD, M, N, ea, C);
    if (ea.mode == EPI_QKV) {
        const int d = ea.H * ea.dh;
        const int n = nw + lane;
        if (n < N) {
            int h, part, i;
            if (ea.head_major) { h = n / (3 * ea.dh); const int w = n - h * 3 * ea.dh; part = w / ea.dh; i = w - part * ea.dh; }
            else { part = n / d; const int w = n - part * d; h = w / ea.dh; i = w - h * ea.dh; }
            if (part == 2) {
                const float bias = ea.bias ? ea.bias[n] : 0.f;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int mb = mw + 8 * rg;
                    if (mb < M) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = stage[(8 * rg + e) * SLD + lane] + bias;
                        const int b = mb / ea.T, t = mb - b * ea.T;
                        TC* dst = reinterpret_cast<TC*>(ea.vt) + ((size_t)(b * ea.H + h) * ea.dh + i) * ea.T + t;
                        store8_n(dst, v, min(8, M - mb));
                    }
                }
            }
        }
    }
}

template <typename TM, typename TC>
static int run_nt_glds(const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s) {
    const int nMt = (M + 63) / 64, nNt = (N + 127) / 128;
    hipLaunchKernelGGL((gemm_nt_glds_kernel<TM, TC>), dim3(nMt * nNt), dim3(256), 0, s, (const TM*)A, (const TM*)Bt, (TC*)C, M, N, K, ldb, ea);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------
// NT kernel v2 ("T"): 128x128 tile, LDS-DMA ring of 3 x 16 KB (K tile 32 bf16 / 16 f32, 64-byte rows), 4 waves (2x2),
// wave tile 64x64 held TRANSPOSED: acc[j][i] = mfma(Bt fragment, A fragment) so that a lane owns, for row
// m = 16i + (lane&15), the 4 CONSECUTIVE output columns n = 16j + 4*(lane>>4) .. +3.  The epilogue therefore runs
// straight from the accumulators — 8-byte (bf16) / 16-byte (f32) loads of residual / act' operands and stores of C per
// lane, no LDS staging, no block barrier — and the weight tile is re-read from L2 half as often as with 64-row tiles.
// ---------------------------------------------------------------------------------
#define GT_STAGE 16384
#define GT_NSTAGE 3

template <typename TM>
struct GtState {
    const TM* pa[2];
    const TM* pb[2];
    int offA[4], offB[4];
};

template <typename T> DEVI void load4t(const T* p, float (&v)[4]) { load4g(p, v); }
template <typename T> DEVI void store4t(T* p, const float (&v)[4]) { store4(p, v); }

// EK: 0 generic run-time-flag epilogue; 1 fast "C = acc + bias"; 2 fast "C = acc + bias + resid" (the epilogue is
// instruction-issue bound: the generic one costs ~800 instructions per wave-tile, the fast ones ~150)
template <typename TM, typename TC, int EK>
__global__ __launch_bounds__(256, 3) void gemm_nt_t_kernel(const TM* __restrict__ A, const TM* __restrict__ Bt, TC* __restrict__ C,
                                                           int M, int N, int K, int ldb, EpiArgs ea) {
    __shared__ __attribute__((aligned(16))) char smem[GT_NSTAGE * GT_STAGE];
    constexpr int EPC = MmaCfg<TM>::EPC, BK = MmaCfg<TM>::BK / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nMt = (M + 127) >> 7, nNt = (N + 127) >> 7;
    const int ntiles = nMt * nNt;
    const int nk = K / BK;
    const int c = lane & 15, g = lane >> 4;
    const int dmodel = ea.H * ea.dh;

    // tile id -> (mt, nt); XCD-aware (blocks b, b+8 share an L2): the N tiles of one A row-panel stay on one XCD
    auto tile_of = [&](int id, int& mt, int& nt) {
        if ((nMt & 7) == 0) { const int xcd = id & 7, local = id >> 3; mt = (local / nNt) * 8 + xcd; nt = local % nNt; }
        else { mt = id / nNt; nt = id % nNt; }
    };
    GtState<TM> gs;
    auto setup = [&](int m0, int n0) {
        const int r = lane >> 2, sp = lane & 3;
        const int kcol = (sp ^ gl_f(r)) * EPC;
#pragma unroll
        for (int u = 0; u < 2; ++u) {     // 8 sixteen-row pieces per operand; wave takes pieces wid, wid+4
            gs.pa[u] = A + (size_t)min(m0 + 16 * (wid + 4 * u) + r, M - 1) * K + kcol;
            gs.pb[u] = Bt + (size_t)(n0 + 16 * (wid + 4 * u) + r) * ldb + kcol;
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + 16 * i + c, rb = wc * 64 + 16 * i + c;
        gs.offA[i] = ra * 64 + (is_bf16_t<TM>::value ? ((g ^ gl_f(ra)) << 4) : (gl_f(ra) << 4) + g * 4);
        gs.offB[i] = 8192 + rb * 64 + (is_bf16_t<TM>::value ? ((g ^ gl_f(rb)) << 4) : (gl_f(rb) << 4) + g * 4);
    }
    auto issue = [&](char* stage) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gs.pa[u],
                                             (__attribute__((address_space(3))) void*)(stage + (wid + 4 * u) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gs.pb[u],
                                             (__attribute__((address_space(3))) void*)(stage + 8192 + (wid + 4 * u) * 1024), 16, 0, 0);
            gs.pa[u] += BK; gs.pb[u] += BK;
        }
    };

    // ---- persistent loop over this workgroup's tiles.  The first two DMA stages of tile t+1 are issued BEFORE the
    // epilogue of tile t, so one workgroup's C stores overlap its own next loads (and workgroups drift out of phase).
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    int mt, nt;
    tile_of(tile, mt, nt);
    setup(mt << 7, nt << 7);
#pragma unroll
    for (int st = 0; st < GT_NSTAGE - 1; ++st)
        if (st < nk) issue(smem + st * GT_STAGE);
    bool first = true;
    while (true) {
        const int m0 = mt << 7, n0 = nt << 7;
        f32x4 acc[4][4];       // acc[j][i]: n tile j, m tile i
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nl = n0 + wc * 64 + 4 * g;          // + 16j
        float bias[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[j][e] = (ea.bias && nl + 16 * j + e < N) ? ea.bias[nl + 16 * j + e] : 0.f;

        for (int kt0 = 0; kt0 < nk; kt0 += GT_NSTAGE) {
#pragma unroll
            for (int u = 0; u < GT_NSTAGE; ++u) {
                const int kt = kt0 + u;
                if (kt < nk) {
                    // 4 DMA per wave per K tile, one later tile in flight.  At the first K tile of a later output tile the
                    // previous epilogue's loads/stores are younger than these DMAs in the in-order counter: drain everything.
                    if (kt + 1 < nk && (kt > 0 || first)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (kt + GT_NSTAGE - 1 < nk) issue(smem + ((u + GT_NSTAGE - 1) % GT_NSTAGE) * GT_STAGE);
                    const char* st = smem + u * GT_STAGE;
                    if constexpr (is_bf16_t<TM>::value) {
                        bf16x8 a[4], b[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { a[i] = *reinterpret_cast<const bf16x8*>(st + gs.offA[i]); b[i] = *reinterpret_cast<const bf16x8*>(st + gs.offB[i]); }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            float a[4], b[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) { a[i] = *reinterpret_cast<const float*>(st + (gs.offA[i] ^ (s << 4))); b[i] = *reinterpret_cast<const float*>(st + (gs.offB[i] ^ (s << 4))); }
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[j][i], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- next tile: once every wave has finished reading the ring, start its first DMA stages, then run the epilogue
        const int ntile = tile + gridDim.x;
        const bool more = ntile < ntiles;
        int mt2 = 0, nt2 = 0;
        if (more) {
            tile_of(ntile, mt2, nt2);
            __builtin_amdgcn_s_barrier();
            setup(mt2 << 7, nt2 << 7);
#pragma unroll
            for (int st = 0; st < GT_NSTAGE - 1; ++st)
                if (st < nk) issue(smem + st * GT_STAGE);
        }
        if constexpr (EK != 0) {
            // ---- fast epilogue: one base pointer per lane, rows 16 apart, column groups 16 apart
            const int mrow = m0 + wr * 64 + c;
            TC* cb = C + (size_t)mrow * N + nl;
            const TC* rb = reinterpret_cast<const TC*>(ea.resid) + (size_t)mrow * N + nl;
            const bool full = (m0 + 128 <= M) && (n0 + 128 <= N);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!full && mrow + 16 * i >= M) continue;
                float ext[4][4];
                if constexpr (EK == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (full || nl + 16 * j < N) load4t(rb + (size_t)(16 * i) * N + 16 * j, ext[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!full && nl + 16 * j >= N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][e] + bias[j][e];
                    if constexpr (EK == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ext[j][e];
                    }
                    store4t(cb + (size_t)(16 * i) * N + 16 * j, v);
                }
            }
        } else if (!(ea.dbg & 1)) {
            // ---- epilogue straight from the accumulators: row m = .. + 16i + c, columns nl + 16j .. +3
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + wr * 64 + 16 * i + c;
                if (m >= M) continue;
                const size_t rowoff = (size_t)m * N;
                float ext[4][4];
                const bool need_res = ea.resid != nullptr, need_aux = ea.dact != DACT_NONE;
                if (need_res || need_aux) {           // batch the residual (or act') loads of the 4 column groups
                    const TC* src = reinterpret_cast<const TC*>(need_res ? ea.resid : ea.aux);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (nl + 16 * j < N) load4t(src + rowoff + nl + 16 * j, ext[j]);
                    }
                }
                const float rsc = ea.rowscale ? ea.rowscale[m / ea.T] : 1.f;
                const uint32_t rk = ea.drop.thr ? rng_row_key(ea.drop.key, (uint32_t)m) : 0u;
                const int bsamp = (ea.mode == EPI_QKV) ? m / ea.T : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = nl + 16 * j;
                    if (n >= N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][e] + bias[j][e] * ((ea.rowscale && ea.rowscale_bias) ? rsc : 1.f);
                    if (ea.addtab) {
                        float t4[4];
                        load4(ea.addtab + (size_t)(m % ea.tab_period) * N + n, t4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += t4[e];
                    }
                    if (ea.pre_out) store4t(reinterpret_cast<TC*>(ea.pre_out) + rowoff + n, v);
                    if (ea.act == ACT_SWISH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = swishf_(v[e]);
                    } else if (ea.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (ea.drop.thr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = rng_keep(rk, (uint32_t)(n + e), ea.drop.thr) ? v[e] * ea.drop.scale : 0.f;
                    }
                    if (ea.rowscale && !ea.rowscale_bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= rsc;
                    }
                    if (need_aux) {
                        float a4[4];
                        if (need_res) load4t(reinterpret_cast<const TC*>(ea.aux) + rowoff + n, a4);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a4[e] = ext[j][e];
                        }
                        if (ea.dact == DACT_SWISH) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= dswishf_(a4[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = a4[e] > 0.f ? v[e] : 0.f;
                        }
                    }
                    if (need_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ext[j][e];
                    }
                    if (ea.mode == EPI_STD) {
                        store4t(C + rowoff + n, v);
                    } else {      // q,k [B,H,T,dh] rows (8-byte stores); v transposed to vt [B,H,dh,T] (2-byte stores, 1/3 of one GEMM in 30)
                        int h, part, ii;
                        if (ea.head_major) { h = n / (3 * ea.dh); const int w = n - h * 3 * ea.dh; part = w / ea.dh; ii = w - part * ea.dh; }
                        else { part = n / dmodel; const int w = n - part * dmodel; h = w / ea.dh; ii = w - h * ea.dh; }
                        const int t = m - bsamp * ea.T;
                        if (part < 2) {
                            store4t(reinterpret_cast<TC*>(part == 0 ? ea.q : ea.k) + ((size_t)(bsamp * ea.H + h) * ea.T + t) * ea.dh + ii, v);
                        } else {
                            TC* dst = reinterpret_cast<TC*>(ea.vt) + ((size_t)(bsamp * ea.H + h) * ea.dh + ii) * ea.T + t;
#pragma unroll
                            for (int e = 0; e < 4; ++e) dst[(size_t)e * ea.T] = from_f<TC>(v[e]);
                        }
                    }
                }
            }
        } else if (acc[0][0][0] == 123.456f) C[0] = from_f<TC>(acc[1][3][2]);
        if (!more) break;
        tile = ntile; mt = mt2; nt = nt2; first = false;
    }
}

template <typename TM, typename TC>
static int run_nt_t(const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s) {
    const int nMt = (M + 127) / 128, nNt = (N + 127) / 128;
    const int ntiles = nMt * nNt;
    const int grid = ntiles < 768 ? ntiles : 768;         // persistent: 3 workgroups per CU, each walks tiles `grid` apart
    const bool fast = ea.addtab == nullptr && ea.pre_out == nullptr && ea.act == ACT_NONE && ea.drop.thr == 0 && ea.rowscale == nullptr &&
                      ea.dact == DACT_NONE && ea.mode == EPI_STD && ea.dbg == 0;
    if (fast && ea.resid) hipLaunchKernelGGL((gemm_nt_t_kernel<TM, TC, 2>), dim3(grid), dim3(256), 0, s, (const TM*)A, (const TM*)Bt, (TC*)C, M, N, K, ldb, ea);
    else if (fast) hipLaunchKernelGGL((gemm_nt_t_kernel<TM, TC, 1>), dim3(grid), dim3(256), 0, s, (const TM*)A, (const TM*)Bt, (TC*)C, M, N, K, ldb, ea);
    else hipLaunchKernelGGL((gemm_nt_t_kernel<TM, TC, 0>), dim3(grid), dim3(256), 0, s, (const TM*)A, (const TM*)Bt, (TC*)C, M, N, K, ldb, ea);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename TA, typename TM, typename TC, int OP>
static int run_nt(const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const OpArgs& oa, const EpiArgs& ea, hipStream_t s) {
    const int nMt = (M + 127) / 128, nNt = (N + 127) / 128;
    const int a_vec_ok = (((size_t)K * sizeof(TA)) % 16 == 0) && (((uintptr_t)A) % 16 == 0);
    hipLaunchKernelGGL((gemm_nt_kernel<TA, TM, TC, OP>), dim3(nMt * nNt), dim3(256), 0, s,
                       (const TA*)A, (const TM*)Bt, (TC*)C, M, N, K, ldb, a_vec_ok, oa, ea);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename TA, typename TM, typename TC>
static int run_nt_op(int op, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const OpArgs& oa, const EpiArgs& ea, hipStream_t s) {
    switch (op) {
        case OP_NONE: return run_nt<TA, TM, TC, OP_NONE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        case OP_SWISH: return run_nt<TA, TM, TC, OP_SWISH>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        case OP_COLAFFINE: return run_nt<TA, TM, TC, OP_COLAFFINE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        case OP_DROPMASK: return run_nt<TA, TM, TC, OP_DROPMASK>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        default: ishara_set_error("gemm_nt: unsupported operand op %d", op); return -1;
    }
}

// name of the kernel launch_gemm_nt / launch_gemm_tn will pick (profiler keys = rocprof kernel names)
const char* gemm_nt_kernel_name(int dtA, int dtM, int dtC, int op, const void* A, int M, int N, int K, int ldb, const EpiArgs& ea);
const char* gemm_nt_as_name(int dtC, int K, const EpiArgs& ea, int M, int N);
int g_force_regstage = 0;   // NT kernel choice: 0 A-stationary kernel (gemm_as.hip) where it applies, else the 128x128 LDS-DMA tile kernel; 3 tile kernel only; 2 LDS-DMA 64x128 kernel; 1 register-staged
bool gemm_nt_as_applicable(int dtC, int M, int N, int K, int ldb, const EpiArgs& ea);
int launch_gemm_nt_as(int dtC, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s);
int launch_gemm_nt_as_f16(int dtC, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s);
int gemm_tn_big_plan(int M, int Ka, int Nb, int* splits_out);
int launch_gemm_tn_big(const void* A, const void* B, float* slab, int want_bias, int M, int Ka, int Nb, int* splits_out, const float* brs, int brsT, hipStream_t s);
bool gemm_nt_big_applicable(int dtA, int dtM, int dtC, int op, const void* A, int M, int N, int K, int ldb, const EpiArgs& ea);
int launch_gemm_nt_big(int dtA, int dtM, int dtC, int op, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s);
int g_dbg_tn = 0;           // ablation bits for the TN kernel: 1 skip MFMA, 2 skip LDS stores, 4 skip global loads

int launch_gemm_nt(int dtA, int dtM, int dtC, int op, const void* A, const void* Bt, void* C,
                   int M, int N, int K, int ldb, const OpArgs& oa, const EpiArgs& ea, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) { ishara_set_error("gemm_nt: bad shape %d %d %d", M, N, K); return -1; }
    if ((dt_is16(dtA) && K % 8 != 0) || (dtA == DT_F32 && K % 4 != 0) || ((uintptr_t)A) % 16 != 0) {
        ishara_set_error("gemm_nt: A rows must be 16-byte aligned (K=%d)", K); return -1;
    }
    if (ea.mode == EPI_QKV && (ea.T % 8 != 0 || N % 8 != 0 || ea.dh % 8 != 0)) {
        ishara_set_error("gemm_nt: QKV split needs T, dh multiples of 8 (T=%d dh=%d)", ea.T, ea.dh); return -1;
    }
    if (dtM == DT_F16) {          // inference-only storage type: the A-stationary kernel (gemm_as_f16.hip) where it applies, else the register-staged 128x128 tile kernel
        if (op != OP_NONE) { ishara_set_error("gemm_nt: f16 operands take no operand transform"); return -1; }
        if (dtA == DT_F16 && g_force_regstage == 0 && K % 32 == 0 && ldb % 64 == 0 && ((uintptr_t)A) % 16 == 0) {
            const int rc = launch_gemm_nt_as_f16(dtC, A, Bt, C, M, N, K, ldb, ea, s);
            if (rc != 1) return rc;
        }
        if (ea.ln_gamma || ea.pa_P) { ishara_set_error("gemm_nt: f16 operand prologue on a shape the A-stationary kernel does not take"); return -1; }
        if (dtA == DT_F16 && dtC == DT_F16) return run_nt<f16, f16, f16, OP_NONE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        if (dtA == DT_F32 && dtC == DT_F16) return run_nt<float, f16, f16, OP_NONE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        if (dtA == DT_F16 && dtC == DT_F32) return run_nt<f16, f16, float, OP_NONE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
        ishara_set_error("gemm_nt: unsupported f16 dtype combination %d/%d/%d", dtA, dtM, dtC); return -1;
    }
    if ((ea.ln_gamma || ea.pa_P) && (op != OP_NONE || !gemm_nt_as_prologue_ok(dtA, dtM, dtC, M, N, K, ldb, ea))) {
        ishara_set_error("gemm_nt: operand prologue requested for a shape the A-stationary kernel does not take (check gemm_nt_as_prologue_ok first)"); return -1;
    }
    if (g_force_regstage == 0) {          // config #4's compute-heavy shapes: the 256 x 256 two-operand tile (gemm_big.hip)
        const int rc = launch_gemm_nt_big(dtA, dtM, dtC, op, A, Bt, C, M, N, K, ldb, ea, s);
        if (rc != 1) return rc;
    }
    const int bk = dtM == DT_BF16 ? 32 : 16;     // K tile of the LDS-DMA kernel
    const bool dma_ok = op == OP_NONE && dtA == dtM && K % bk == 0 && ldb % (2 * bk) == 0 && ((uintptr_t)A) % 16 == 0;
    if (dma_ok && g_force_regstage == 0 && dtM == DT_BF16) {
        const int rc = launch_gemm_nt_as(dtC, A, Bt, C, M, N, K, ldb, ea, s);
        if (rc != 1) return rc;
    }
    if (dma_ok && (g_force_regstage == 0 || g_force_regstage == 3) && N % 4 == 0 && (ea.mode != EPI_QKV || ea.dh % 4 == 0)) {
        if (dtM == DT_F32 && dtC == DT_F32) return run_nt_t<float, float>(A, Bt, C, M, N, K, ldb, ea, s);
        if (dtM == DT_BF16 && dtC == DT_BF16) return run_nt_t<bf16, bf16>(A, Bt, C, M, N, K, ldb, ea, s);
        if (dtM == DT_BF16 && dtC == DT_F32) return run_nt_t<bf16, float>(A, Bt, C, M, N, K, ldb, ea, s);
    }
    if (dma_ok && g_force_regstage != 1) {
        if (dtM == DT_F32 && dtC == DT_F32) return run_nt_glds<float, float>(A, Bt, C, M, N, K, ldb, ea, s);
        if (dtM == DT_BF16 && dtC == DT_BF16) return run_nt_glds<bf16, bf16>(A, Bt, C, M, N, K, ldb, ea, s);
        if (dtM == DT_BF16 && dtC == DT_F32) return run_nt_glds<bf16, float>(A, Bt, C, M, N, K, ldb, ea, s);
    }
    if (dtA == DT_F32 && dtM == DT_F32 && dtC == DT_F32) return run_nt_op<float, float, float>(op, A, Bt, C, M, N, K, ldb, oa, ea, s);
    if (dtA == DT_BF16 && dtM == DT_BF16 && dtC == DT_BF16) return run_nt_op<bf16, bf16, bf16>(op, A, Bt, C, M, N, K, ldb, oa, ea, s);
    if (dtA == DT_F32 && dtM == DT_BF16 && dtC == DT_BF16 && op == OP_NONE) return run_nt<float, bf16, bf16, OP_NONE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
    if (dtA == DT_BF16 && dtM == DT_BF16 && dtC == DT_F32 && op == OP_NONE) return run_nt<bf16, bf16, float, OP_NONE>(A, Bt, C, M, N, K, ldb, oa, ea, s);
    ishara_set_error("gemm_nt: unsupported dtype combination %d/%d/%d op %d", dtA, dtM, dtC, op);
    return -1;
}

// ---------------------------------------------------------------------------------
// TN kernel (wgrad).  grid = (tiles_k * tiles_n, splits).  Both operands are staged
// TRANSPOSED into LDS ([feature row][m contiguous]) so the MFMA fragment reads are the
// same as in the NT kernel; swizzle SW=1 keeps both the 8-byte transposed writes and
// the 16-byte fragment reads at <= 2-way bank conflicts.
// ---------------------------------------------------------------------------------
template <typename T, typename TM>
DEVI void stage_t_load(const T* __restrict__ base, int ld, int m_end, int cols, int mbase, int col0, bool vec_ok,
                       int op, const OpArgs& oa, int tid, u32x4 (&out)[4], float* colsum) {
    // bf16 TM: thread -> kc = tid&15 (8 cols), mg = tid>>4 (4 rows of 64) ; out[e] holds rows for cols 2e,2e+1 (uint2 each)
    // f32  TM: thread -> kc = tid&31 (4 cols), mg = tid>>5 (4 rows of 32) ; out[e] = 4 rows of col e
    constexpr int EPC = MmaCfg<TM>::EPC;
    const int kc = is_bf16_t<TM>::value ? (tid & 15) : (tid & 31);
    const int mg = is_bf16_t<TM>::value ? (tid >> 4) : (tid >> 5);
    float v[4][EPC];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = mbase + 4 * mg + r, col = col0 + kc * EPC;
        load_row_chunk<T, EPC>(base, ld, m_end, cols, m, col, vec_ok, v[r]);
        if (op != OP_NONE) {
            apply_op<EPC>(op, v[r], m, col, oa);
            if (m >= m_end) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[r][e] = 0.f;
            } else if (col + EPC > cols) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) if (col + e >= cols) v[r][e] = 0.f;
            }
        }
    }
    if (colsum) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) colsum[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
    }
    if constexpr (is_bf16_t<TM>::value) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            out[e].x = pack_bf16x2(v[0][2 * e], v[1][2 * e]);         out[e].y = pack_bf16x2(v[2][2 * e], v[3][2 * e]);
            out[e].z = pack_bf16x2(v[0][2 * e + 1], v[1][2 * e + 1]); out[e].w = pack_bf16x2(v[2][2 * e + 1], v[3][2 * e + 1]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            out[e].x = __float_as_uint(v[0][e]); out[e].y = __float_as_uint(v[1][e]);
            out[e].z = __float_as_uint(v[2][e]); out[e].w = __float_as_uint(v[3][e]);
        }
    }
}

template <typename TM>
DEVI void stage_t_store(char* lds, int tid, const u32x4 (&r)[4]) {
    if constexpr (is_bf16_t<TM>::value) {
        const int kc = tid & 15, mg = tid >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = 8 * kc + 2 * e + h;
                const int off = row * 128 + (((mg >> 1) ^ swz<1>(row)) << 4) + ((mg & 1) << 3);
                *reinterpret_cast<u32x2*>(lds + off) = h == 0 ? u32x2{r[e].x, r[e].y} : u32x2{r[e].z, r[e].w};
            }
        }
    } else {
        const int kc = tid & 31, mg = tid >> 5;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * kc + e;
            *reinterpret_cast<u32x4*>(lds + row * 128 + ((mg ^ swz<1>(row)) << 4)) = r[e];
        }
    }
}

template <typename TA, typename TB, typename TM>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TA* __restrict__ A, const TB* __restrict__ B,
                                                      float* __restrict__ slab, float* __restrict__ bias_slab,
                                                      int M, int Ka, int Nb, int rows_per_split, int a_vec_ok, int b_vec_ok,
                                                      int opA, int opB, OpArgs oa, OpArgs ob, int dbg) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    constexpr int EPC = MmaCfg<TM>::EPC, MC = MmaCfg<TM>::BK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int nNt = (Nb + 127) >> 7;
    const int kt = blockIdx.x / nNt, nt = blockIdx.x % nNt;
    const int k0 = kt << 7, n0 = nt << 7;
    const int split = blockIdx.y;
    const int m_beg = split * rows_per_split;
    const int m_end = min(M, m_beg + rows_per_split);
    const int nmc = (m_end - m_beg + MC - 1) / MC;
    const bool want_bias = (bias_slab != nullptr) && (kt == 0);

    u32x4 ra[4], rb[4];
    float csum[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) csum[e] = 0.f;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nmc > 0) {
        stage_t_load<TA, TM>(A, Ka, m_end, Ka, m_beg, k0, a_vec_ok != 0, opA, oa, tid, ra, nullptr);
        stage_t_load<TB, TM>(B, Nb, m_end, Nb, m_beg, n0, b_vec_ok != 0, opB, ob, tid, rb, want_bias ? csum : nullptr);
        stage_t_store<TM>(smem, tid, ra);
        stage_t_store<TM>(smem + 16384, tid, rb);
    }
    __syncthreads();
    for (int mc = 0; mc < nmc; ++mc) {
        const bool more = mc + 1 < nmc;
        if (more && !(dbg & 4)) {
            stage_t_load<TA, TM>(A, Ka, m_end, Ka, m_beg + (mc + 1) * MC, k0, a_vec_ok != 0, opA, oa, tid, ra, nullptr);
            stage_t_load<TB, TM>(B, Nb, m_end, Nb, m_beg + (mc + 1) * MC, n0, b_vec_ok != 0, opB, ob, tid, rb, want_bias ? csum : nullptr);
        }
        const char* sa = smem + (mc & 1) * 32768;
        if (!(dbg & 1)) mma_tile<TM, 1>(sa, sa + 16384, wr, wc, lane, acc);
        if (more && !(dbg & 2)) {
            char* sn = smem + ((mc + 1) & 1) * 32768;
            stage_t_store<TM>(sn, tid, ra);
            stage_t_store<TM>(sn + 16384, tid, rb);
        }
        __syncthreads();
    }

    float* out = slab + (size_t)split * Ka * Nb;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = k0 + wr * 64 + 16 * i + 4 * (lane >> 4) + r;
                const int nn = n0 + wc * 64 + 16 * j + (lane & 15);
                if (kk < Ka && nn < Nb) out[(size_t)kk * Nb + nn] = acc[i][j][r];
            }

    if (want_bias) {   // block-reduce the per-thread column sums (uniform branch: kt is per-block)
        float* red = reinterpret_cast<float*>(smem);       // [groups][128]
        constexpr int GROUPS = is_bf16_t<TM>::value ? 16 : 8;
        const int kc = is_bf16_t<TM>::value ? (tid & 15) : (tid & 31);
        const int mg = is_bf16_t<TM>::value ? (tid >> 4) : (tid >> 5);
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[mg * 128 + kc * EPC + e] = csum[e];
        __syncthreads();
        if (tid < 128) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) sacc += red[g * 128 + tid];
            if (n0 + tid < Nb) bias_slab[(size_t)split * Nb + n0 + tid] = sacc;
        }
    }
}

// ---------------------------------------------------------------------------------
// TN kernel, bf16, LDS-DMA + hardware-transposed fragment reads (the common wgrad case: no
// operand transform, M % 64 == 0, Ka % 128 == 0, Nb % 128 == 0).
// Both operands are [m][feature] row-major in HBM and the MFMA reduction index is m, i.e. the
// fragments are COLUMNS of the stored tiles.  Instead of transposing through registers, the
// 32-row x 128-column tiles (256-byte rows) are copied as they are by global_load_lds_dwordx4
// into a 4-deep LDS ring (three tiles in flight, two workgroups per CU), and each fragment is fetched with two
// ds_read_b64_tr_b16 (a 4-row x 16-column block delivered column-major).  The 32-byte column
// blocks of a row are XOR-swizzled with f(row) = (row&3) | ((row>>3)&1)<<2 — applied on the
// SOURCE address of the DMA and on the read — so the 8 rows a 32-lane half reads land in 8
// different 32-byte bank groups (conflict-free).  The bias gradient (column sums of dY) is
// accumulated from the B fragments already in registers.  M is split over workgroups; each writes
// its 128x128 fp32 partial to a slab with coalesced 16-byte stores (an all-at-once fp32-atomic
// epilogue measured 40 us for 32 MB, the chip-wide atomic rate) and reduce_slabs_kernel sums them.
// ---------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
#define TR_STAGE 16384          // A 32 rows x 256 B + B 32 rows x 256 B
#define TR_ROWS 32
#define TR_AUX 0             // cache policy of the operand DMA: 0 = default: the tiles of one M-split share the operand rows through L2 (non-temporal, 2, measured 2.38 -> 2.87 ms/step of wgrad)
#define TR_NSTAGE 4          // measured: 3 stages (3 workgroups/CU) 116 us, 4 stages (2/CU) 90 us, 5 stages (80 KB, 1-2/CU) 94 us per wgrad

DEVI int tr_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

DEVI void tr_issue(const bf16* __restrict__ A, const bf16* __restrict__ B, int Ka, int Nb, int k0, int n0, int mrow0,
                   char* stage, int wid, int lane) {
    const int r = lane >> 4, sp = lane & 15;            // row within a 4-row piece, physical 16-byte slot
#pragma unroll
    for (int u = 0; u < 2; ++u) {                        // 8 four-row pieces per operand; wave takes wid, wid+4
        const int pc = wid + 4 * u;
        const int row = 4 * pc + r;
        const int col = (((sp >> 1) ^ tr_f(row)) << 4) + ((sp & 1) << 3);     // logical column of physical slot sp
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)(mrow0 + row) * Ka + k0 + col),
                                         (__attribute__((address_space(3))) void*)(stage + pc * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + (size_t)(mrow0 + row) * Nb + n0 + col),
                                         (__attribute__((address_space(3))) void*)(stage + 8192 + pc * 1024), 16, 0, 0);
    }
}

// fragment of column block lb (16 columns), k-step s (32 rows): lane (g,q,p) addresses row 32s+8g+4h+q, 8 bytes at p
DEVI bf16x8 tr_frag(const char* tile, int lb, int s, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 32 * s + 8 * g + q, r1 = r0 + 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + r0 * 256 + ((lb ^ tr_f(r0)) << 5) + (pp << 3)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + r1 * 256 + ((lb ^ tr_f(r1)) << 5) + (pp << 3)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

typedef __attribute__((ext_vector_type(2))) unsigned tn_u32x2;
struct TrFrags { u32x4 a[4], b[4]; };
// fragment reads of the A (offset 0) and B (offset 8192) tiles of ring slot SLOT, as inline asm (see the kernel comment)
template <int SLOT>
DEVI void tr_read_asm(const unsigned (&fa)[4], const unsigned (&fb)[4], TrFrags& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tn_u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(fa[i]), "n"(SLOT * TR_STAGE));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(fa[i]), "n"(SLOT * TR_STAGE + 1024));
        f.a[i] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tn_u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(fb[j]), "n"(SLOT * TR_STAGE + 8192));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(fb[j]), "n"(SLOT * TR_STAGE + 8192 + 1024));
        f.b[j] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
}
// the reads above have landed: the fragments pass through the wait so that their consumers are ordered behind it
DEVI void tr_wait(TrFrags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]));
}

// The slab sums of the PREVIOUS weight-gradient GEMM ride along as extra workgroups of the next one (blockIdx.x >= nmain): the sums are
// ~5 us kernels that leave the chip idle, 59 of them per step; here they run beside the GEMM's workgroups (two slab buffers alternate).
struct TnRed { const float* slab; float* out0; float* out1; int n0, n, splits; size_t stride; int nb, nbv, nmain; int inl; };   // inl: no rider workgroups — every main workgroup sums its share after its tile (PSA: one workgroup per CU, riders could not run beside them)
DEVI void tn_reduce_block(const TnRed& r, int rb, int nrb, float4 (*red)[32]) {
    constexpr int SL = 8, CQ = 32;                       // the layout of reduce_slabs_cols_kernel<8, 32>
    const int tid = threadIdx.x, cq = tid % CQ, sl = tid / CQ;
    for (int grp = rb; grp * (CQ * 4) < r.n; grp += nrb) {
        const int col = grp * (CQ * 4) + cq * 4;
        const int cin = r.nb ? (col < r.n0 ? col % r.nb : col - r.n0) : 0;
        const bool valid = col < r.n && (r.nb == 0 || cin < r.nbv);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            for (int s0 = sl; s0 < r.splits; s0 += SL * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sidx = s0 + SL * u;
                    v[u] = sidx < r.splits ? *reinterpret_cast<const float4*>(r.slab + (size_t)sidx * r.stride + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
        red[sl][cq] = acc;
        __syncthreads();
        if (sl == 0 && valid) {
            float4 t = red[0][cq];
#pragma unroll
            for (int w = 1; w < SL; ++w) { t.x += red[w][cq].x; t.y += red[w][cq].y; t.z += red[w][cq].z; t.w += red[w][cq].w; }
            float* dst = col < r.n0 ? (r.nb ? r.out0 + (size_t)(col / r.nb) * r.nbv + cin : r.out0 + col) : r.out1 + (col - r.n0);
            dst[0] += t.x; dst[1] += t.y; dst[2] += t.z; dst[3] += t.w;
        }
        __syncthreads();
    }
}

template <int CTRL> DEVI float tn_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// One sample of the split is complete (PSA).  Its accumulator is folded into the running total with the sample's affine, and the
// statistics of the gradient of the transformed operand are emitted — but not on the spot: a pause of the whole workgroup (~700 VALU
// instructions per wave) lets the 3-stage operand ring run dry, and refilling it cost more than the arithmetic (first version: 4 us per
// boundary, 77 us per launch against 41 without).  tn_psa_begin parks the finished accumulator (hold) and the sample's column sums; the
// four 16-row chunks are processed by tn_psa_chunk<I> in the next four steps, in the same basic block as their MFMAs (samples are a
// multiple of four stages long, so chunk I always sits in step copy I of the 4x unrolled loop).  Two accumulator sets alternating between
// samples (no parking) made hipcc spill 250 registers per lane to scratch.  The statistics go to LDS (po: this lane's R quad of the
// sample's 384-float record [4 waves x 64 R | 128 G], asm stores: a builtin LDS access would wait for all operand DMA) and to memory
// after the loop — global stores in the loop would sit in vmcnt between the DMAs and be waited for by the steps' vmcnt(8).
DEVI void tn_psa_begin(f32x4 (&acc)[4][4], f32x4 (&hold)[4][4], f32x4 (&gacc)[4], float (&ctot)[4], float (&gbh)[4], unsigned gaddr,
                       bool psa_g, const float* __restrict__ brs, int sample, int lane, u32x4 z8) {
    // column sums of B over the sample's rows: an MFMA with an all-ones A operand per B fragment (every accumulator row holds them — no
    // cross-lane fold; summing the unpacked fragments on the VALU in every wave cost 10 us per launch)
#pragma unroll
    for (int j = 0; j < 4; ++j) { gbh[j] = gacc[j][0]; gacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (psa_g) {
        const float sc = brs ? brs[sample] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ctot[j] += sc * gbh[j];
        if ((lane >> 4) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_write_b32 %0, %1" :: "v"(gaddr + 64u * j), "v"(gbh[j]) : "memory");
        }
    }
    // park the finished accumulator and clear it ON THE MATRIX PIPE (z8 = an all-zero operand the compiler cannot see through):
    // hold = 0 x 0 + acc, acc = 0 x 0 + 0 — 32 MFMAs the pipe has room for, instead of ~200 accumulator-register moves on the VALU
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hold[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, z8), __builtin_bit_cast(bf16x8, z8), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, z8), __builtin_bit_cast(bf16x8, z8), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
}
// The running total lives in LDS (tot_lds: this lane's 16-byte slot of the wave's [16 tiles][64 lanes] f32x4 block): with it in registers
// the kernel needed 530 of the 512 registers a wave can have.
template <int I>
DEVI void tn_psa_chunk(const f32x4 (&hold)[4][4], unsigned tot_lds, const tn_u32x2 (&wv)[4][4], const float (&gbh)[4], unsigned prow, unsigned orow, int lane) {
    f32x4 p4, q4, tot[1][4];
    asm volatile("ds_read_b128 %0, %1" : "=v"(p4) : "v"(prow + 64u * I));
    asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(q4) : "v"(prow + 64u * I));
    const unsigned ta = tot_lds + 4096u * I;
    asm volatile("ds_read_b128 %0, %1" : "=v"(tot[0][0]) : "v"(ta));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(tot[0][1]) : "v"(ta));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(tot[0][2]) : "v"(ta));
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(tot[0][3]) : "v"(ta));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p4), "+v"(q4), "+v"(tot[0][0]), "+v"(tot[0][1]), "+v"(tot[0][2]), "+v"(tot[0][3]));
    f32x4 rr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float w0 = __uint_as_float(wv[I][j].x << 16), w1 = __uint_as_float(wv[I][j].x & 0xffff0000u);
        const float w2 = __uint_as_float(wv[I][j].y << 16), w3 = __uint_as_float(wv[I][j].y & 0xffff0000u);
        const f32x4 a = hold[I][j];
        const f32x4 w4 = {w0, w1, w2, w3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tot[0][j][r] += p4[r] * a[r] + q4[r] * gbh[j];
            rr[r] += w4[r] * a[r];
        }
    }
    asm volatile("ds_write_b128 %0, %1" :: "v"(ta), "v"(tot[0][0]) : "memory");
    asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(ta), "v"(tot[0][1]) : "memory");
    asm volatile("ds_write_b128 %0, %1 offset:2048" :: "v"(ta), "v"(tot[0][2]) : "memory");
    asm volatile("ds_write_b128 %0, %1 offset:3072" :: "v"(ta), "v"(tot[0][3]) : "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) {                         // sum over the 16 column lanes of the row (quad swaps, half mirror, mirror)
        float v = rr[r];
        v = tn_dpp_add<0xB1>(v); v = tn_dpp_add<0x4E>(v); v = tn_dpp_add<0x141>(v); v = tn_dpp_add<0x140>(v);
        rr[r] = v;
    }
    if ((lane & 15) == 0) asm volatile("ds_write_b128 %0, %1" :: "v"(orow + 64u * I), "v"(rr) : "memory");
}

#define TN_PSA_MAXS 8          // samples per M-split the per-sample-affine variant stages coefficients for
// DBG = 1: the ablation bits of tools/gemm_ablate.py are honoured (kept out of the production loop); BRS: weighted bias sum;
// PSA: per-sample affine of the A operand + the BatchNorm / ECA backward statistics (TnPsa, kernels.h): one workgroup per CU (the per-sample
// accumulator, the running total and the weight tile take 192 registers)
template <int DBG, bool BRS = false, bool PSA = false>
__global__ __launch_bounds__(256, PSA ? 1 : 2) void gemm_tn_tr_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                         float* __restrict__ out, float* __restrict__ dbias,
                                                         int M, int Ka, int Nb, int rows_per_split, int tiles, int nsplits, int dbg,
                                                         const float* __restrict__ brs, int brsT, TnRed prev, TnPsa psa) {
    // brs != nullptr: the bias gradient is the column sum of brs[m / brsT] * B[m,:] (drop-path scale of the sample a row belongs to;
    // brsT % 32 == 0, so the 32 rows of a stage share it)
    // slab layout: [split][Ka*Nb weight partial | Nb bias partial] so that ONE reduction launch sums both
    const size_t sstride = (size_t)Ka * Nb + Nb;
    __shared__ __attribute__((aligned(16))) char smem[TR_NSTAGE * TR_STAGE];
    __shared__ __attribute__((aligned(16))) float pq_tab[PSA ? TN_PSA_MAXS * 256 : 4];     // [sample of this split][P of the tile's 128 rows | Q]
    __shared__ __attribute__((aligned(16))) float psa_out[PSA ? TN_PSA_MAXS * 384 : 4];    // [sample of this split][R: 4 waves x 64 rows | G: 128 columns]
    __shared__ __attribute__((aligned(16))) f32x4 tot_tab[PSA ? 4 * 16 * 64 : 1];          // [wave][tile i*4+j][lane]: the running total (sum over finished samples)
    if (prev.nmain > 0 && !prev.inl && (int)blockIdx.x >= prev.nmain) {        // a rider: sums a slice of the previous launch's slabs
        tn_reduce_block(prev, (int)blockIdx.x - prev.nmain, (int)gridDim.x - prev.nmain, reinterpret_cast<float4 (*)[32]>(smem));
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nNt = Nb >> 7;
    // XCD-aware mapping (blocks b, b+8 share an XCD / L2): the `tiles` output tiles of one M-split read the same
    // A and dY rows, so they are dealt to ONE XCD back to back; consecutive splits go to different XCDs.
    int tile, split;
    {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        if ((nsplits & 7) == 0) { split = (j / tiles) * 8 + xcd; tile = j % tiles; }
        else { split = id / tiles; tile = id % tiles; }
    }
    const int kt = tile / nNt, nt = tile % nNt;
    const int k0 = kt << 7, n0 = nt << 7;
    const int m_beg = split * rows_per_split;
    const int m_end = min(M, m_beg + rows_per_split);
    const int nmc = max(0, m_end - m_beg) / TR_ROWS;      // whole 32-row tiles (launcher guarantees)
    const bool want_bias = (dbias != nullptr) && (kt == 0) && (wr == 0);

    f32x4 acc[4][4], hold[4][4];           // hold (PSA): the previous sample's accumulator while its chunks are processed
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; hold[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    // weighted bias sum: csum collects the rows of the current sample; at a sample boundary (every brsT / 32 stages) it is folded into
    // ctot with that sample's scale — one scalar load per sample instead of one per stage (a scalar load in the step loop shares
    // lgkmcnt with the fragment reads and stalls them)
    float ctot[4] = {0.f, 0.f, 0.f, 0.f};
    const int seg_stages = PSA ? psa.T / TR_ROWS : (BRS ? brsT / TR_ROWS : 0);
    const int seg_iters = seg_stages / 4;                    // PSA: samples are a multiple of 4 stages (launcher)
    int seg_left = PSA ? seg_iters : (BRS ? seg_stages - (m_beg % brsT) / TR_ROWS : -1);      // PSA: splits are whole samples (launcher)
    int seg_sample = PSA ? m_beg / psa.T : (BRS ? m_beg / brsT : 0);
    // ---- PSA state: tot = sum over finished samples of P[b] * acc_b + Q[b] x colsum_b ; wv = this wave's 64x64 block of W in the
    // accumulator layout (row 16i + 4(lane>>4) + r, column 16j + (lane&15))
    tn_u32x2 wv[4][4];                                       // bf16 pairs (rows r0|r1, r2|r3)
    float gbh[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 gacc[4];                                           // column sums of the current sample's B rows (ones x B fragments)
#pragma unroll
    for (int j = 0; j < 4; ++j) gacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 ones8 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    bool first = false;                                      // the running iteration is the first of a sample whose predecessor is parked
    u32x4 z8 = {0u, 0u, 0u, 0u};                             // an all-zero MFMA operand, opaque to the compiler (tn_psa_begin)
    asm volatile("" : "+v"(z8));
    unsigned pq_hold = 0, po_hold = 0;
    unsigned pq_row = (unsigned)(uintptr_t)pq_tab + 4u * (wr * 64 + 4 * (lane >> 4));     // LDS byte address of this lane's coefficient quad
    unsigned po_row = (unsigned)(uintptr_t)psa_out + 4u * (wid * 64 + 4 * (lane >> 4));    // ... of its R quad, and of its G column
    unsigned po_g = (unsigned)(uintptr_t)psa_out + 4u * (256 + wc * 64 + (lane & 15));
    const unsigned tot_lds = (unsigned)(uintptr_t)tot_tab + 16u * (wid * 16 * 64 + lane);
    const bool psa_g = PSA && kt == 0 && wr == 0;            // the waves that publish G[b, :] and own the bias sum
    float tv[PSA ? TN_PSA_MAXS : 1];
    if constexpr (PSA) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) tot_tab[(wid * 16 + i * 4 + j) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ns = nmc / seg_stages;
        const float* src = (tid < 128 ? psa.P : psa.Q) + (size_t)seg_sample * Ka + k0 + (tid & 127);
#pragma unroll
        for (int sidx = 0; sidx < TN_PSA_MAXS; ++sidx) tv[sidx] = (sidx < ns && !(psa.dbg & 4)) ? src[(size_t)sidx * Ka] : 0.f;
        const bf16* wsrc = reinterpret_cast<const bf16*>(psa.W) + (size_t)(k0 + wr * 64 + 4 * (lane >> 4)) * psa.ldw + n0 + wc * 64 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned short* wp = reinterpret_cast<const unsigned short*>(wsrc) + (size_t)(16 * i) * psa.ldw + 16 * j;
                if (psa.dbg & 4) { wv[i][j].x = 0u; wv[i][j].y = 0u; continue; }
                wv[i][j].x = (unsigned)wp[0] | ((unsigned)wp[psa.ldw] << 16);
                wv[i][j].y = (unsigned)wp[2 * (size_t)psa.ldw] | ((unsigned)wp[3 * (size_t)psa.ldw] << 16);
            }
    }
    (void)psa_out; (void)tv;

    // Software pipeline over the 32-row stages (ring of 4 x 16 KB, slots addressed statically: the loop is unrolled by 4):
    // at step mc the MFMAs of stage mc run from fragments already in registers while the transposing LDS reads of stage
    // mc+1 are in flight into the OTHER fragment set (ping-pong, no copies), and stages mc+2 .. mc+4 are in flight from
    // L2/HBM (the slot of stage mc is free as soon as every wave has passed this step's barrier, because its fragments
    // were read and waited for during step mc-1).
    // The fragment reads are inline asm: for the ds_read_tr builtin hipcc puts `s_waitcnt vmcnt(0)` in front of the reads
    // (it cannot tell them from the LDS-DMA writes in flight) and `lgkmcnt(0)` in front of the MFMAs, which serialises
    // the DMA of three stages, the LDS reads and the MFMAs of every step.
    // DMA source pointers of this lane for the NEXT stage to issue (advanced by 32 rows per issue: no 64-bit address
    // arithmetic in the loop); stages are always issued in order 0, 1, 2, ...
    const bf16* pa[2];
    const bf16* pb[2];
    {
        const int r = lane >> 4, sp = lane & 15;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pc = wid + 4 * u, row = 4 * pc + r;
            const int col = (((sp >> 1) ^ tr_f(row)) << 4) + ((sp & 1) << 3);
            pa[u] = A + (size_t)(m_beg + row) * Ka + k0 + col;
            pb[u] = B + (size_t)(m_beg + row) * Nb + n0 + col;
        }
    }
    const size_t stepA = (size_t)TR_ROWS * Ka, stepB = (size_t)TR_ROWS * Nb;
    // LDS byte addresses of this lane's fragment reads in slot 0 (lane (g,q,p) addresses row 8g+q, +4 rows at +1024)
    unsigned fa[4], fb[4];
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, r0 = 8 * g + q;
        const unsigned base = (unsigned)(uintptr_t)smem + r0 * 256 + (pp << 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = base + (((wr * 4 + i) ^ tr_f(r0)) << 5);
            fb[i] = base + (((wc * 4 + i) ^ tr_f(r0)) << 5);
        }
    }
    const bool no_dma = DBG != 0 && (dbg & 4) != 0, no_frag = DBG != 0 && (dbg & 2) != 0, no_mma = DBG != 0 && (dbg & 1) != 0;
    const int nmc_run = (DBG != 0 && (dbg & 16) != 0) ? 0 : nmc;
    TrFrags P, Q;
#pragma unroll
    for (int i = 0; i < 4; ++i) { P.a[i] = P.b[i] = Q.a[i] = Q.b[i] = u32x4{0u, 0u, 0u, 0u}; }

#define TN_ISSUE(SLOT, ST)                                                                                               \
    if ((ST) < nmc && !no_dma) {                                                                                         \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                  \
            const int pc = wid + 4 * u;                                                                                  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[u],                       \
                                             (__attribute__((address_space(3))) void*)(smem + (SLOT) * TR_STAGE + pc * 1024), 16, 0, TR_AUX);        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[u],                       \
                                             (__attribute__((address_space(3))) void*)(smem + (SLOT) * TR_STAGE + 8192 + pc * 1024), 16, 0, TR_AUX); \
            pa[u] += stepA; pb[u] += stepB;                                                                              \
        }                                                                                                                \
    }
#define TN_COMPUTE(CUR)                                                                                                  \
    if (PSA) {                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
            gacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones8), __builtin_bit_cast(bf16x8, CUR.b[j]), gacc[j], 0, 0, 0); \
    }                                                                                                                    \
    if (!PSA && want_bias) {                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
            float t = 0.f;                                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                \
                t += __uint_as_float(CUR.b[j][e] << 16) + __uint_as_float(CUR.b[j][e] & 0xffff0000u);                    \
            csum[j] += t;                                                                                                \
        }                                                                                                                \
    }                                                                                                                    \
    if (!no_mma) {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, CUR.a[i]), __builtin_bit_cast(bf16x8, CUR.b[j]), acc[i][j], 0, 0, 0); \
    } else {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][0][0] += __uint_as_float(CUR.a[i][0]) + __uint_as_float(CUR.b[i][0]); \
    }
    // one step: FULL = at least two younger stages are in flight behind stage mc+U+1 (steady state: no branches)
#define TN_STEP(U, CUR, NXT, FULL, CH)                                                                                   \
    if (FULL || mc + (U) < nmc_run) {                                                                                    \
        if (FULL) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                       \
        else {                                                                                                           \
            const int younger = min(nmc - 1, mc + (U) + 3) - (mc + (U) + 1);                                             \
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                           \
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                      \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        }                                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                                    \
        TN_ISSUE(U, mc + (U) + TR_NSTAGE)                                                                                \
        if ((FULL || mc + (U) + 1 < nmc_run) && !no_frag) tr_read_asm<((U) + 1) & 3>(fa, fb, NXT);                       \
        TN_COMPUTE(CUR)                                                                                                  \
        if (PSA && first) tn_psa_chunk<(U)>(hold, tot_lds, wv, gbh, pq_hold, po_hold, lane);     /* behind the MFMAs' issue */   \
        if (!PSA && BRS && want_bias && --seg_left == 0) {                                                               \
            const float sc = brs[seg_sample];                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { ctot[j] += sc * csum[j]; csum[j] = 0.f; }                    \
            seg_left = seg_stages; ++seg_sample;                                                                         \
        }                                                                                                                \
        if (PSA) {                                                                                                       \
            if ((U) == 3) {                                                                                              \
                first = false;                                                                                           \
                if (--seg_left == 0 && !(psa.dbg & 1)) {                                                                 \
                    tn_psa_begin(acc, hold, gacc, ctot, gbh, po_g, psa_g, brs, seg_sample, lane, z8);                    \
                    first = true; pq_hold = pq_row; po_hold = po_row;                                                    \
                    pq_row += 1024u; po_row += 1536u; po_g += 1536u; seg_left = seg_iters; ++seg_sample;                 \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
        tr_wait(NXT);                                                                                                    \
    }

    TN_ISSUE(0, 0) TN_ISSUE(1, 1) TN_ISSUE(2, 2) TN_ISSUE(3, 3)
    if constexpr (PSA) {
        // the coefficient / weight loads were issued before the first four stages' DMA and complete before it (loads return in order); the
        // values are pinned here — passing them through an empty asm keeps hipcc from putting their wait at the first use inside the
        // pipelined loop, where a vmcnt(0) would drain the operand DMA at every sample boundary
#pragma unroll
        for (int sidx = 0; sidx < TN_PSA_MAXS; ++sidx) pq_tab[sidx * 256 + tid] = tv[sidx];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(wv[i][j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int mc = 0;
    if (nmc_run > 0) {
        // stage 0 landed: its 4 DMA are the oldest of up to 16
        if (nmc >= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!no_frag) tr_read_asm<0>(fa, fb, P);
        tr_wait(P);
#define TN_ITER(FULL, CH) TN_STEP(0, P, Q, FULL, CH) TN_STEP(1, Q, P, FULL, CH) TN_STEP(2, P, Q, FULL, CH) TN_STEP(3, Q, P, FULL, CH)
#define TN_CHUNKS tn_psa_chunk<0>(hold, tot_lds, wv, gbh, pq_hold, po_hold, lane); tn_psa_chunk<1>(hold, tot_lds, wv, gbh, pq_hold, po_hold, lane); \
                  tn_psa_chunk<2>(hold, tot_lds, wv, gbh, pq_hold, po_hold, lane); tn_psa_chunk<3>(hold, tot_lds, wv, gbh, pq_hold, po_hold, lane);
        for (; mc + 7 <= nmc_run; mc += 4) {          // steps mc .. mc+3 all have stages mc+U+3 <= nmc-1 behind them
            TN_ITER(true, false)                      // PSA: the four chunks of a parked sample ride in the sample's first four steps (a second
                                                      // copy of the loop body with unconditional chunks made hipcc spill 160 registers)
        }
        for (; mc < nmc_run; mc += 4) {
            TN_ITER(false, false)
        }
    }
#undef TN_ITER
#undef TN_STEP
#undef TN_COMPUTE
#undef TN_ISSUE
    if constexpr (PSA) {
        if (first) { TN_CHUNKS }          // the split's last sample was parked by the final step
    }
#undef TN_CHUNKS

    // ---- write this split's 128x128 partial to its fp32 slab: each wave transposes its 64x64 accumulator block through
    // a private LDS stage (two 32-row passes) so that every store instruction writes 4 rows x 256 contiguous bytes
    __syncthreads();                       // the ring is free
    if constexpr (PSA) {                   // the samples' statistics: LDS -> Rpart / G
        const int ns = nmc / seg_stages, b0 = m_beg / psa.T;
        for (int idx = tid; idx < ns * 384 && !(psa.dbg & 16); idx += 256) {
            const int sl = idx / 384, e = idx - sl * 384;
            const float v = psa_out[idx];
            if (e < 256) {
                const int w_ = e >> 6, row = e & 63;
                psa.Rpart[((size_t)(b0 + sl) * (Nb >> 6) + nt * 2 + (w_ & 1)) * Ka + k0 + (w_ >> 1) * 64 + row] = v;
            } else if (kt == 0) psa.G[(size_t)(b0 + sl) * Nb + n0 + (e - 256)] = v;
        }
    }
    if (DBG != 0 && (dbg & 8) != 0) { if (acc[0][0][0] == 123.f) out[0] = acc[1][1][1]; return; }
    {
        constexpr int SLD = 68;
        float* stage = reinterpret_cast<float*>(smem) + wid * (32 * SLD);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(16 * ii + 4 * (lane >> 4) + r) * SLD + 16 * j + (lane & 15)] = PSA ? tot_tab[(wid * 16 + (2 * p + ii) * 4 + j) * 64 + lane][r] : acc[2 * p + ii][j][r];
            float* orow = out + (size_t)split * sstride + (size_t)(k0 + wr * 64 + 32 * p + (lane >> 4)) * Nb + n0 + wc * 64 + (lane & 15) * 4;
#pragma unroll
            for (int it = 0; it < 8; ++it)       // one instruction = 4 rows x 256 contiguous bytes
                *reinterpret_cast<float4*>(orow + (size_t)(4 * it) * Nb) = *reinterpret_cast<const float4*>(stage + ((lane >> 4) + 4 * it) * SLD + (lane & 15) * 4);
        }
    }
    if (want_bias) {   // lane (g, c) holds the sum over rows 8g..8g+7 (mod 32) of column 16j + c: fold the 4 row groups
        if (PSA) {      // every sample of the split was folded at its boundary; all four row groups hold the full sums
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[j] = (lane >> 4) == 0 ? ctot[j] : 0.f;
        } else if (BRS) {      // the rows of the last, unfinished sample of this split
            const float sc = seg_left != seg_stages && seg_sample * brsT < M ? brs[seg_sample] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) csum[j] = ctot[j] + sc * csum[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = csum[j];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (lane < 16) dbias[(size_t)split * sstride + n0 + wc * 64 + 16 * j + lane] = t;
        }
    }
    if (PSA && prev.nmain > 0 && prev.inl) {        // the previous launch's slab sums, a share per workgroup
        __syncthreads();
        tn_reduce_block(prev, (int)blockIdx.x, (int)gridDim.x, reinterpret_cast<float4 (*)[32]>(smem));
    }
}

// out[i] += sum_s slab[s*stride + i].  grid = (n/4/256, split groups): each block sums its
// subset of splits with 4 independent 16-byte loads in flight and issues one atomic per element.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, int n, int splits, size_t stride) {
    const int i4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    const int per = (splits + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * per, s1 = min(splits, s0 + per);
    if (i4 + 4 <= n && (stride & 3) == 0) {
        float4 acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int sidx = s0;
        for (; sidx + 8 <= s1; sidx += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)(sidx + u) * stride + i4);
                acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            }
        }
        for (; sidx < s1; ++sidx) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)sidx * stride + i4);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
#pragma unroll
        for (int u = 1; u < 8; ++u) { acc[0].x += acc[u].x; acc[0].y += acc[u].y; acc[0].z += acc[u].z; acc[0].w += acc[u].w; }
        if (gridDim.y == 1) {
            float4 o = *reinterpret_cast<const float4*>(out + i4);
            o.x += acc[0].x; o.y += acc[0].y; o.z += acc[0].z; o.w += acc[0].w;
            *reinterpret_cast<float4*>(out + i4) = o;
        } else {
            atomicAdd(out + i4, acc[0].x); atomicAdd(out + i4 + 1, acc[0].y);
            atomicAdd(out + i4 + 2, acc[0].z); atomicAdd(out + i4 + 3, acc[0].w);
        }
    } else {
        for (int e = 0; e < 4 && i4 + e < n; ++e) {
            float acc = 0.f;
            for (int sidx = s0; sidx < s1; ++sidx) acc += slab[(size_t)sidx * stride + i4 + e];
            atomicAdd(out + i4 + e, acc);
        }
    }
}

void launch_reduce_slabs(const float* slab, float* out, int n, int splits, size_t stride, hipStream_t s);
// Slab sums without atomics: a workgroup of 256 threads owns 4*CQ columns; thread (cq = tid % CQ, sl = tid / CQ) sums the
// slabs sl, sl+SL, ... of column quad cq with 8 loads in flight (SL = 8 covers 64 splits in one pass), the SL partial
// rows are combined through LDS in a fixed order, and ONE thread adds the total to out; one launch serves two outputs
// (columns [0, n0) -> out0, [n0, n) -> out1).  Same-address atomics were the whole cost of the first reducer (2-way 80 us,
// 4-way 83 us, 8-way 106 us per wgrad including the GEMM); 128 slab lanes with one load each ran 10.5 us per 32 MB.
template <int SL, int CQ>
__global__ __launch_bounds__(SL * CQ) void reduce_slabs_cols_kernel(const float* __restrict__ slab, float* __restrict__ out0, float* __restrict__ out1,
                                                                    int n0, int n, int splits, size_t stride, int nb, int nbv) {
    __shared__ float4 red[SL][CQ];
    const int tid = threadIdx.x, cq = tid % CQ, sl = tid / CQ;
    const int col = blockIdx.x * (CQ * 4) + cq * 4;
    // nb != 0: the slab rows are nb wide with only the first nbv columns real (zero-padded B operand); out0 rows are nbv wide
    const int cin = nb ? (col < n0 ? col % nb : col - n0) : 0;
    const bool valid = col < n && (nb == 0 || cin < nbv);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        for (int s0 = sl; s0 < splits; s0 += SL * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sidx = s0 + SL * u;
                v[u] = sidx < splits ? *reinterpret_cast<const float4*>(slab + (size_t)sidx * stride + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    red[sl][cq] = acc;
    __syncthreads();
    if (sl == 0 && valid) {
        float4 t = red[0][cq];
#pragma unroll
        for (int w = 1; w < SL; ++w) { t.x += red[w][cq].x; t.y += red[w][cq].y; t.z += red[w][cq].z; t.w += red[w][cq].w; }
        float* dst = col < n0 ? (nb ? out0 + (size_t)(col / nb) * nbv + cin : out0 + col) : out1 + (col - n0);      // parameter blocks of the flat gradient are only 4-byte aligned
        dst[0] += t.x; dst[1] += t.y; dst[2] += t.z; dst[3] += t.w;
    }
}
static void launch_reduce_cols(const float* slab, float* out0, float* out1, int n0, int n, int splits, size_t stride, hipStream_t s, int nb = 0, int nbv = 0) {
    if (n <= 2048) hipLaunchKernelGGL((reduce_slabs_cols_kernel<64, 4>), dim3((n + 15) / 16), dim3(256), 0, s, slab, out0, out1, n0, n, splits, stride, nb, nbv);   // narrow, many partial rows (LayerNorm, dwconv)
    else if (splits <= 64) hipLaunchKernelGGL((reduce_slabs_cols_kernel<8, 32>), dim3((n + 127) / 128), dim3(256), 0, s, slab, out0, out1, n0, n, splits, stride, nb, nbv);
    else hipLaunchKernelGGL((reduce_slabs_cols_kernel<16, 16>), dim3((n + 63) / 64), dim3(256), 0, s, slab, out0, out1, n0, n, splits, stride, nb, nbv);
}

static bool reduce_cols_ok(const float* slab, const float* out0, const float* out1, int n0, int n, size_t stride) {
    (void)out0; (void)out1;
    return n % 4 == 0 && n0 % 4 == 0 && stride % 4 == 0 && ((uintptr_t)slab) % 16 == 0;
}

// ---- the deferred form: every recorded job in one launch.  A workgroup = 16 slab lanes x 16 column quads (64 columns of one job); the jobs'
// first block indices ride in the kernel arguments (scalar loads), the body is reduce_slabs_cols_kernel<16, 16>'s.
RedSink* g_red_sink = nullptr;
struct RedBatch { RedJob job[RED_MAXJOBS]; int njobs; };
__global__ __launch_bounds__(256) void reduce_jobs_kernel(RedBatch b) {
    constexpr int SL = 16, CQ = 16;
    __shared__ float4 red[SL][CQ];
    int j = 0;
    for (int t = 1; t < b.njobs; ++t) if ((int)blockIdx.x >= b.job[t].first_block) j = t;      // uniform: scalar compares
    const RedJob J = b.job[j];
    const int tid = threadIdx.x, cq = tid % CQ, sl = tid / CQ;
    const int col = ((int)blockIdx.x - J.first_block) * (CQ * 4) + cq * 4;
    const int cin = J.nb ? (col < J.n0 ? col % J.nb : col - J.n0) : 0;
    const bool valid = col < J.n && (J.nb == 0 || cin < J.nbv);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        for (int s0 = sl; s0 < J.splits; s0 += SL * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sidx = s0 + SL * u;
                v[u] = sidx < J.splits ? *reinterpret_cast<const float4*>(J.slab + (size_t)sidx * J.stride + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    red[sl][cq] = acc;
    __syncthreads();
    if (sl == 0 && valid) {
        float4 t = red[0][cq];
#pragma unroll
        for (int w = 1; w < SL; ++w) { t.x += red[w][cq].x; t.y += red[w][cq].y; t.z += red[w][cq].z; t.w += red[w][cq].w; }
        float* dst = col < J.n0 ? (J.nb ? J.out0 + (size_t)(col / J.nb) * J.nbv + cin : J.out0 + col) : J.out1 + (col - J.n0);
        dst[0] += t.x; dst[1] += t.y; dst[2] += t.z; dst[3] += t.w;
    }
}
bool reduce_sink_full() { return g_red_sink && g_red_sink->njobs >= RED_MAXJOBS - 2; }
static bool reduce_sink_take(const float* slab, float* out0, float* out1, int n0, int n, int splits, size_t stride, int nb, int nbv) {
    RedSink* k = g_red_sink;
    if (!k || k->njobs >= RED_MAXJOBS || !reduce_cols_ok(slab, out0, out1, n0, n, stride)) return false;
    // two jobs into the same parameter (the Conformer block's shared layer_norm1 is differentiated twice) would add to it from two workgroups
    // of the one launch: the later one is launched on its own right away, the batch follows in stream order (two ordered adds)
    for (int i = 0; i < k->njobs; ++i)
        if (k->job[i].out0 == out0 || (out1 && k->job[i].out1 == out1) || (out1 && k->job[i].out0 == out1) || k->job[i].out1 == out0) return false;
    RedJob& J = k->job[k->njobs++];
    J.slab = slab; J.out0 = out0; J.out1 = out1; J.stride = stride; J.n0 = n0; J.n = n; J.splits = splits; J.nb = nb; J.nbv = nbv;
    J.first_block = k->nblocks;
    k->nblocks += (n + 63) / 64;
    return true;
}
int launch_reduce_flush(RedSink* sink, hipStream_t s) {
    if (!sink || sink->njobs == 0) return 0;
    RedBatch b;
    for (int i = 0; i < sink->njobs; ++i) b.job[i] = sink->job[i];
    b.njobs = sink->njobs;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(sink->nblocks), dim3(256), 0, s, b);
    sink->njobs = 0; sink->nblocks = 0;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// out0[0..n0) += column sums of slab[:, 0..n0), out1[0..n1) += column sums of slab[:, n0..n0+n1)   (slab rows `stride` floats apart)
void launch_reduce_slabs2(const float* slab, float* out0, int n0, float* out1, int n1, int splits, size_t stride, hipStream_t s, int nb, int nbv) {
    if (reduce_sink_take(slab, out0, out1, n0, n0 + n1, splits, stride, nb, nbv)) return;
    if (reduce_cols_ok(slab, out0, out1, n0, n0 + n1, stride)) { launch_reduce_cols(slab, out0, out1, n0, n0 + n1, splits, stride, s, nb, nbv); return; }
    launch_reduce_slabs(slab, out0, n0, splits, stride, s);
    if (out1 && n1 > 0) launch_reduce_slabs(slab + n0, out1, n1, splits, stride, s);
}

void launch_reduce_slabs(const float* slab, float* out, int n, int splits, size_t stride, hipStream_t s) {
    if (reduce_sink_take(slab, out, nullptr, n, n, splits, stride, 0, 0)) return;
    if (reduce_cols_ok(slab, out, nullptr, n, n, stride)) { launch_reduce_cols(slab, out, nullptr, n, n, splits, stride, s); return; }
    const int gx = (n + 1023) / 1024;
    int gy = 1;                                            // split groups: enough workgroups to fill the chip
    while (gx * gy < 256 && gy * 16 <= splits) gy *= 2;     // same-address atomics are expensive: 8-way 106 us, 4-way 83 us per wgrad (incl. GEMM)
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, gy), dim3(256), 0, s, slab, out, n, splits, stride);
}

static void tn_plan(int M, int Ka, int Nb, int dtM, int& splits, int& rows_per_split) {
    const int MC = dtM == DT_BF16 ? 64 : 32;
    const int tiles = ((Ka + 127) / 128) * ((Nb + 127) / 128);
    int want = (768 + tiles - 1) / tiles;                 // ~3 workgroups per CU
    const int maxs = (M + 4 * MC - 1) / (4 * MC);         // at least 4 LDS tiles per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    rows_per_split = ((M + want - 1) / want + MC - 1) / MC * MC;
    splits = (M + rows_per_split - 1) / rows_per_split;
}

size_t gemm_tn_slab_floats(int M, int Ka, int Nb, int dtM) {
    int splits, rps;
    tn_plan(M, Ka, Nb, dtM, splits, rps);
    if (splits < 512) splits = 512;        // the transposed-read kernel plans <= 512 splits
    return (size_t)splits * ((size_t)Ka * Nb + Nb);
}

int g_force_tn_regstage = 0;   // tests: force the register-transposing TN kernel

int g_tn_blocks = 0;
int g_tn_phase = 0;            // 0: GEMM + slab sums; 1: GEMM kernel only; 2: slab sums only (the model profiles the two separately)

// PSA split plan: whole samples per split (the statistics of a sample come from one workgroup per output tile), at most TN_PSA_MAXS of them,
// as close to one workgroup per CU as the batch allows; 0 = no such plan
static int tn_psa_splits(int M, int Ka, int Nb, int T) {
    if (T <= 0 || T % (4 * TR_ROWS) != 0 || M % T != 0) return 0;      // samples = whole iterations of the 4x unrolled step loop
    const int Bn = M / T, tiles = (Ka / 128) * (Nb / 128);
    const int want = max(1, 256 / tiles);
    for (int d = min(want, Bn); d >= 1; --d) {
        if (Bn % d != 0) continue;
        if (d > 8 && (d & 7) != 0) continue;              // keeps the XCD-aware (tile, split) mapping
        const int ns = Bn / d;
        if (ns > TN_PSA_MAXS) return 0;
        if (ns * T < 256) continue;                       // at least 8 stages per split
        return d;
    }
    return 0;
}
bool gemm_tn_psa_ok(int dtA, int dtB, int dtM, int M, int Ka, int Nb, int T) {
    return dtA == DT_BF16 && dtB == DT_BF16 && dtM == DT_BF16 && M % 64 == 0 && Ka % 128 == 0 && Nb % 128 == 0 && M >= 256 && !g_force_tn_regstage &&
           !g_dbg_tn && tn_psa_splits(M, Ka, Nb, T) > 0;
}

static int run_tn_tr(const void* A, const void* B, float* out, float* dbias, float* slab, int M, int Ka, int Nb, hipStream_t s, int ka_valid, int nb_valid,
                     const float* brs, int brsT, TnDefer* defer, const TnPsa* psa) {
    if (!psa && (!brs || (brsT > 0 && brsT % 128 == 0)) && !g_dbg_tn && !g_tn_blocks && ka_valid == Ka && !nb_valid) {
        // config #4's shapes (Ka, Nb >= 512 in whole 256-wide tiles, M >= 32768): the 256 x 256 tile kernel of gemm_big.hip, sums right behind it
        int bsplits = 0;
        if (gemm_tn_big_plan(M, Ka, Nb, &bsplits) > 0) {
            launch_gemm_tn_flush(defer, s);
            if (g_tn_phase != 2) {
                const int rc = launch_gemm_tn_big(A, B, slab, dbias ? 1 : 0, M, Ka, Nb, &bsplits, (brs && dbias) ? brs : nullptr, brsT, s);
                if (rc != 0) return rc == 1 ? -2 : rc;
            }
            if (g_tn_phase != 1) launch_reduce_slabs2(slab, out, Ka * Nb, dbias, dbias ? Nb : 0, bsplits, (size_t)Ka * Nb + Nb, s, 0, 0);
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
    }
    const int tiles = (Ka / 128) * (Nb / 128);
    const TnPsa nopsa = {};
    // workgroups: one per CU for up to 8 tiles (same kernel time as two per CU, half the slab bytes: the slab sums go
    // 9.6 -> 7.3 us), two per CU for 12+ tiles (N = 768: 61 vs 72 us); g_tn_blocks != 0 overrides (tools/tn_ablate.py)
    const int blocks = g_tn_blocks ? g_tn_blocks : (tiles <= 8 ? 256 : 512);
    int want = max(1, blocks / tiles);
    const int maxs = max(1, M / 256);                     // at least 8 tiles per split
    if (want > maxs) want = maxs;
    if (want > 8) want &= ~7;                             // a multiple of 8 keeps the XCD-aware (tile, split) mapping: 12 tiles x 42 splits ran 102 us, x 40: see DESIGN.md
    int rps = ((M + want - 1) / want + TR_ROWS - 1) / TR_ROWS * TR_ROWS;
    int splits = (M + rps - 1) / rps;
    while (want > 8 && (splits & 7) != 0 && rps > TR_ROWS) { rps -= TR_ROWS; splits = (M + rps - 1) / rps; if (splits > 512) break; }
    if (splits > 512) { rps = ((M + want - 1) / want + TR_ROWS - 1) / TR_ROWS * TR_ROWS; splits = (M + rps - 1) / rps; }
    if (psa) {
        splits = tn_psa_splits(M, Ka, Nb, psa->T);
        if (splits <= 0 || g_dbg_tn) { ishara_set_error("gemm_tn: no per-sample-affine plan for M=%d T=%d (gemm_tn_psa_ok)", M, psa->T); return -1; }
        rps = M / splits;
    }
    if (defer && g_tn_phase == 0 && !g_dbg_tn && tiles * splits > 256 && !psa) {
        // a launch that fills the chip twice over (12+ tiles at two workgroups per CU) neither carries riders nor defers its own sums: the riders
        // would queue behind 512 workgroups and its slabs (2 MB x 16 splits for a 1024 x 512 weight) make the next launch's riders the tail
        // (configs[3]: 47.9 -> 51.5 ms/step with the deferral on everywhere)
        launch_gemm_tn_flush(defer, s);
        defer = nullptr;
    }
    if (defer && g_tn_phase == 0 && !g_dbg_tn) {
        // deferred sums: this launch writes the slab buffer whose turn it is, and carries the sums of the previous launch's slabs
        slab = defer->slab[defer->turn];
        float* bias_slab2 = dbias ? slab + (size_t)Ka * Nb : nullptr;
        TnRed prev = {};
        const int nmain = tiles * splits;
        int riders = 0;
        if (defer->pending) {
            prev = TnRed{defer->p_slab, defer->p_out0, defer->p_out1, defer->p_n0, defer->p_n, defer->p_splits, defer->p_stride, defer->p_nb, defer->p_nbv, nmain, psa ? 1 : 0};
            riders = psa ? 0 : min(128, (defer->p_n + 127) / 128);
        }
        const dim3 grid(nmain + riders);
        if (psa) hipLaunchKernelGGL((gemm_tn_tr_kernel<0, false, true>), grid, dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab2, M, Ka, Nb, rps, tiles, splits, 0, brs, brsT, prev, *psa);
        else if (brs && bias_slab2) hipLaunchKernelGGL((gemm_tn_tr_kernel<0, true>), grid, dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab2, M, Ka, Nb, rps, tiles, splits, 0, brs, brsT, prev, nopsa);
        else hipLaunchKernelGGL((gemm_tn_tr_kernel<0, false>), grid, dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab2, M, Ka, Nb, rps, tiles, splits, 0, brs, brsT, prev, nopsa);
        const size_t stride = (size_t)Ka * Nb + Nb;
        const int n0 = ka_valid * Nb, n1 = dbias ? Nb : 0;
        if (!reduce_cols_ok(slab, out, dbias, n0, n0 + n1, stride) || splits > 64) {        // shapes the rider layout does not take: sum now
            launch_reduce_slabs2(slab, out, n0, dbias, n1, splits, stride, s, nb_valid ? Nb : 0, nb_valid);
            defer->pending = false;
        } else {
            defer->pending = true;
            defer->p_slab = slab; defer->p_out0 = out; defer->p_out1 = dbias; defer->p_n0 = n0; defer->p_n = n0 + n1; defer->p_splits = splits; defer->p_stride = stride;
            defer->p_nb = nb_valid ? Nb : 0; defer->p_nbv = nb_valid;
        }
        defer->turn ^= 1;
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    float* bias_slab = dbias ? slab + (size_t)Ka * Nb : nullptr;        // bias partials sit right behind each split's weight partial
    const TnRed noprev = {};
    if (g_tn_phase != 2)
    {
        if (psa) hipLaunchKernelGGL((gemm_tn_tr_kernel<0, false, true>), dim3(tiles * splits), dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab, M, Ka, Nb, rps, tiles, splits, 0, brs, brsT, noprev, *psa);
        else if (g_dbg_tn) hipLaunchKernelGGL((gemm_tn_tr_kernel<1, false>), dim3(tiles * splits), dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab, M, Ka, Nb, rps, tiles, splits, g_dbg_tn, brs, brsT, noprev, nopsa);
        else if (brs && bias_slab) hipLaunchKernelGGL((gemm_tn_tr_kernel<0, true>), dim3(tiles * splits), dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab, M, Ka, Nb, rps, tiles, splits, 0, brs, brsT, noprev, nopsa);
        else hipLaunchKernelGGL((gemm_tn_tr_kernel<0, false>), dim3(tiles * splits), dim3(256), 0, s, (const bf16*)A, (const bf16*)B, slab, bias_slab, M, Ka, Nb, rps, tiles, splits, 0, brs, brsT, noprev, nopsa);
    }
    if (g_tn_phase != 1)
        launch_reduce_slabs2(slab, out, ka_valid * Nb, dbias, dbias ? Nb : 0, splits, (size_t)Ka * Nb + Nb, s, nb_valid ? Nb : 0, nb_valid);   // rows >= ka_valid of A / columns >= nb_valid of B are zero padding
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename TA, typename TB, typename TM>
static int run_tn(int opA, int opB, const void* A, const void* B, float* out, float* dbias, float* slab,
                  int M, int Ka, int Nb, int dtM, const OpArgs& oa, const OpArgs& ob, hipStream_t s) {
    int splits, rps;
    tn_plan(M, Ka, Nb, dtM, splits, rps);
    float* bias_slab = dbias ? slab + (size_t)splits * Ka * Nb : nullptr;
    const int tiles = ((Ka + 127) / 128) * ((Nb + 127) / 128);
    const int a_ok = (((size_t)Ka * sizeof(TA)) % 16 == 0) && (((uintptr_t)A) % 16 == 0);
    const int b_ok = (((size_t)Nb * sizeof(TB)) % 16 == 0) && (((uintptr_t)B) % 16 == 0);
    if (g_tn_phase != 2)
        hipLaunchKernelGGL((gemm_tn_kernel<TA, TB, TM>), dim3(tiles, splits), dim3(256), 0, s,
                           (const TA*)A, (const TB*)B, slab, bias_slab, M, Ka, Nb, rps, a_ok, b_ok, opA, opB, oa, ob, g_dbg_tn);
    if (g_tn_phase != 1) {
        launch_reduce_slabs(slab, out, Ka * Nb, splits, (size_t)Ka * Nb, s);
        if (dbias) launch_reduce_slabs(bias_slab, dbias, Nb, splits, (size_t)Nb, s);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// sums whatever slabs a deferring caller still has pending (end of the backward pass, or before the gradient is read)
int launch_gemm_tn_flush(TnDefer* defer, hipStream_t s) {
    if (defer && defer->pending) {
        launch_reduce_slabs2(defer->p_slab, defer->p_out0, defer->p_n0, defer->p_out1, defer->p_n - defer->p_n0, defer->p_splits, defer->p_stride, s, defer->p_nb, defer->p_nbv);
        defer->pending = false;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

bool gemm_tn_bias_rowscale_ok(int dtA, int dtB, int dtM, int M, int Ka, int Nb, int T) {
    return dtA == DT_BF16 && dtB == DT_BF16 && dtM == DT_BF16 && M % 64 == 0 && Ka % 128 == 0 && Nb % 128 == 0 && M >= 256 && !g_force_tn_regstage && T > 0 && T % 32 == 0;
}

int launch_gemm_tn(int dtA, int dtB, int dtM, int opA, int opB, const void* A, const void* B,
                   float* out, float* dbias, float* slab, int M, int Ka, int Nb,
                   const OpArgs& oa, const OpArgs& ob, hipStream_t s, int ka_valid, int nb_valid, const float* bias_rowscale, int bias_T, TnDefer* defer, const TnPsa* psa) {
    if (M <= 0 || Ka <= 0 || Nb <= 0) { ishara_set_error("gemm_tn: bad shape"); return -1; }
    if (psa && (!gemm_tn_psa_ok(dtA, dtB, dtM, M, Ka, Nb, psa->T) || opA != OP_NONE || opB != OP_NONE || ka_valid > 0 || nb_valid > 0 || !psa->P || !psa->Q || !psa->W || !psa->G || !psa->Rpart ||
                (bias_rowscale && bias_T != psa->T))) {
        ishara_set_error("gemm_tn: per-sample affine needs the transposed-read kernel, whole samples per split and all of P, Q, W, G, Rpart (gemm_tn_psa_ok)"); return -1;
    }
    if (bias_rowscale && !gemm_tn_bias_rowscale_ok(dtA, dtB, dtM, M, Ka, Nb, bias_T)) { ishara_set_error("gemm_tn: bias row scale needs the transposed-read kernel and T %% 32 == 0"); return -1; }
    if (ka_valid <= 0) ka_valid = Ka;
    if (nb_valid >= Nb || nb_valid < 0) nb_valid = 0;
    if (nb_valid % 4 != 0) { ishara_set_error("gemm_tn: nb_valid %d must be a multiple of 4", nb_valid); return -1; }
    if ((dtA == DT_BF16 && Ka % 8 != 0) || (dtA == DT_F32 && Ka % 4 != 0) || (dtB == DT_BF16 && Nb % 8 != 0) || (dtB == DT_F32 && Nb % 4 != 0) ||
        ((uintptr_t)A) % 16 != 0 || ((uintptr_t)B) % 16 != 0) {
        ishara_set_error("gemm_tn: operand rows must be 16-byte aligned (Ka=%d Nb=%d)", Ka, Nb); return -1;
    }
    if (dtA == DT_F32 && dtB == DT_F32 && dtM == DT_F32) return run_tn<float, float, float>(opA, opB, A, B, out, dbias, slab, M, Ka, Nb, dtM, oa, ob, s);
    if (dtA == DT_BF16 && dtB == DT_BF16 && dtM == DT_BF16 && opA == OP_NONE && opB == OP_NONE && M % 64 == 0 && Ka % 128 == 0 && Nb % 128 == 0 &&
        M >= 256 && !g_force_tn_regstage)
    {
        if (ka_valid < Ka && dbias) { ishara_set_error("gemm_tn: padded A columns with a bias gradient"); return -1; }
        return run_tn_tr(A, B, out, dbias, slab, M, Ka, Nb, s, ka_valid, nb_valid, bias_rowscale, bias_T, defer, psa);
    }
    if (ka_valid != Ka || nb_valid) { ishara_set_error("gemm_tn: padded A columns need the bf16 transposed-read kernel (M %% 64, Ka %% 128, Nb %% 128)"); return -1; }
    if (dtA == DT_BF16 && dtB == DT_BF16 && dtM == DT_BF16) return run_tn<bf16, bf16, bf16>(opA, opB, A, B, out, dbias, slab, M, Ka, Nb, dtM, oa, ob, s);
    if (dtA == DT_F32 && dtB == DT_BF16 && dtM == DT_BF16) return run_tn<float, bf16, bf16>(opA, opB, A, B, out, dbias, slab, M, Ka, Nb, dtM, oa, ob, s);
    if (dtA == DT_BF16 && dtB == DT_F32 && dtM == DT_BF16) return run_tn<bf16, float, bf16>(opA, opB, A, B, out, dbias, slab, M, Ka, Nb, dtM, oa, ob, s);
    ishara_set_error("gemm_tn: unsupported dtype combination %d/%d/%d", dtA, dtB, dtM);
    return -1;
}

// ---------------------------------------------------------------------------------
// weight shadows: Wt[n][k] = W[k][n] (ld ldt) and Wn[k][n] = W[k][n] (ld ldn), zero padded
// by a preceding memset of the whole shadow arena.
// ---------------------------------------------------------------------------------
template <typename TM>
__global__ void make_shadow_kernel(const float* __restrict__ W, int K, int N, TM* __restrict__ Wt, int ldt, TM* __restrict__ Wn, int ldn) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        float v = 0.f;
        if (k < K && n < N) { v = W[(size_t)k * N + n]; if (Wn) Wn[(size_t)k * ldn + n] = from_f<TM>(v); }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (Wt) {
        for (int r = ty; r < 32; r += 8) {
            const int n = n0 + r, k = k0 + tx;
            if (k < K && n < N) Wt[(size_t)n * ldt + k] = from_f<TM>(tile[tx][r]);
        }
    }
}

// all weight shadows of the model in ONE launch: block -> (weight, 32x32 tile) through a descriptor table in device memory
template <typename TM>
__global__ void make_shadow_batched_kernel(const ShadowDesc* __restrict__ tab, int ntab) {
    __shared__ float tile[32][33];
    int wi = 0;
    for (int hi = ntab; hi - wi > 1;) {       // the weight this tile belongs to: binary search over the descriptors' first tiles (a linear walk
        const int mid = (wi + hi) >> 1;       // was up to ntab dependent loads per block: 0.7 ms per step for configs[3]'s 88 M parameters)
        if ((int)blockIdx.x >= tab[mid].tile0) wi = mid; else hi = mid;
    }
    const ShadowDesc d = tab[wi];
    const int lt = blockIdx.x - d.tile0;
    const int k0 = (lt / d.tiles_n) * 32, n0 = (lt % d.tiles_n) * 32;
    TM* Wt = reinterpret_cast<TM*>(d.Wt);
    TM* Wn = reinterpret_cast<TM*>(d.Wn);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        float v = 0.f;
        if (k < d.K && n < d.N) { v = d.W[(size_t)k * d.N + n]; if (Wn) Wn[(size_t)k * d.ldn + n] = from_f<TM>(v); }
        tile[r][tx] = v;
    }
    __syncthreads();
    if (Wt) {
        for (int r = ty; r < 32; r += 8) {
            const int n = n0 + r, k = k0 + tx;
            if (k < d.K && n < d.N) Wt[(size_t)n * d.ldt + k] = from_f<TM>(tile[tx][r]);
        }
    }
}
int launch_make_shadow_batched(int dtM, const ShadowDesc* tab, int ntab, int total_tiles, hipStream_t s) {
    if (ntab <= 0 || total_tiles <= 0) return 0;
    if (dtM == DT_BF16) hipLaunchKernelGGL(make_shadow_batched_kernel<bf16>, dim3(total_tiles), dim3(256), 0, s, tab, ntab);
    else if (dtM == DT_F16) hipLaunchKernelGGL(make_shadow_batched_kernel<f16>, dim3(total_tiles), dim3(256), 0, s, tab, ntab);
    else hipLaunchKernelGGL(make_shadow_batched_kernel<float>, dim3(total_tiles), dim3(256), 0, s, tab, ntab);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_make_shadow(int dtM, const float* W, int K, int N, void* Wt, int ldt, void* Wn, int ldn, hipStream_t s) {
    dim3 grid((N + 31) / 32, (K + 31) / 32);
    if (dtM == DT_BF16) hipLaunchKernelGGL(make_shadow_kernel<bf16>, grid, dim3(256), 0, s, W, K, N, (bf16*)Wt, ldt, (bf16*)Wn, ldn);
    else if (dtM == DT_F16) hipLaunchKernelGGL(make_shadow_kernel<f16>, grid, dim3(256), 0, s, W, K, N, (f16*)Wt, ldt, (f16*)Wn, ldn);
    else hipLaunchKernelGGL(make_shadow_kernel<float>, grid, dim3(256), 0, s, W, K, N, (float*)Wt, ldt, (float*)Wn, ldn);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// xb[M, Kp] (bf16) = x[M, F] (f32) zero padded: the stem's input rows as an MFMA operand (the f32-A GEMM kernels round the
// same way while staging; done once here, the stem Dense and its wgrad run on the bf16 fast paths with K = Kp)
template <typename TM>
__global__ __launch_bounds__(256) void pack_rows_bf16_kernel(const float* __restrict__ x, TM* __restrict__ xb, int M, int F, int Kp) {
    const int cpr = Kp >> 3;                                   // 16-byte output chunks per row
    const size_t total = (size_t)M * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / cpr;
        const int c0 = (int)(i - row * cpr) * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* src = x + row * F + c0;
        if (c0 + 4 <= F) { const float4 a = *reinterpret_cast<const float4*>(src); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
        if (c0 + 8 <= F) { const float4 a = *reinterpret_cast<const float4*>(src + 4); v[4] = a.x; v[5] = a.y; v[6] = a.z; v[7] = a.w; }
        *reinterpret_cast<u32x4*>(xb + row * Kp + c0) = pack_chunk<TM, 8>(v);
    }
}
template <typename TM>
__global__ __launch_bounds__(256) void dense_narrow_kernel(const TM* __restrict__ A, const TM* __restrict__ Wt, int ldt, const float* __restrict__ bias,
                                                           float* __restrict__ C, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int m0 = (blockIdx.x * 4 + wid) * 4;                  // this wave's 4 rows
    const int n = min(lane, N - 1);
    const TM* wrow = Wt + (size_t)n * ldt;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int mr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mr[r] = min(m0 + r, M - 1);
    for (int k = 0; k < K; k += 32) {                            // 4 weight chunks and 16 operand chunks (wave-uniform addresses) in flight
        u32x4 w[4], a[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            w[u] = *reinterpret_cast<const u32x4*>(wrow + k + 8 * u);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r][u] = *reinterpret_cast<const u32x4*>(A + (size_t)mr[r] * K + k + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            typedef __attribute__((ext_vector_type(8))) TM v8;
            const v8 wv = __builtin_bit_cast(v8, w[u]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v8 av = __builtin_bit_cast(v8, a[r][u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r] += (float)av[e] * (float)wv[e];
            }
        }
    }
    if (lane < N) {
        const float b = bias ? bias[lane] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (m0 + r < M) C[(size_t)(m0 + r) * N + lane] = acc[r] + b;
    }
}
int launch_dense_narrow(int dt, const void* A, const void* Wt, int ldt, const float* bias, float* C, int M, int N, int K, hipStream_t s) {
    if (!dt_is16(dt) || N < 1 || N > 64 || K % 32 != 0 || ldt % 8 != 0 || ((uintptr_t)A) % 16 != 0 || ((uintptr_t)Wt) % 16 != 0) { ishara_set_error("dense_narrow: N=%d K=%d unsupported", N, K); return -1; }
    const dim3 grid((M + 15) / 16);
    if (dt == DT_F16) hipLaunchKernelGGL(dense_narrow_kernel<f16>, grid, dim3(256), 0, s, (const f16*)A, (const f16*)Wt, ldt, bias, C, M, N, K);
    else hipLaunchKernelGGL(dense_narrow_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)A, (const bf16*)Wt, ldt, bias, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_pack_rows_bf16(const float* x, void* xb, int M, int F, int Kp, hipStream_t s, int dt) {
    if (F % 4 != 0 || Kp % 8 != 0 || Kp < F || ((uintptr_t)x) % 16 != 0) { ishara_set_error("pack_rows_bf16: F=%d Kp=%d unsupported", F, Kp); return -1; }
    const int grid = (int)std::min<size_t>(2048, ((size_t)M * (Kp >> 3) + 255) / 256);
    if (dt == DT_F16) hipLaunchKernelGGL(pack_rows_bf16_kernel<f16>, dim3(grid), dim3(256), 0, s, x, (f16*)xb, M, F, Kp);
    else hipLaunchKernelGGL(pack_rows_bf16_kernel<bf16>, dim3(grid), dim3(256), 0, s, x, (bf16*)xb, M, F, Kp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

const char* gemm_nt_kernel_name(int dtA, int dtM, int dtC, int op, const void* A, int M, int N, int K, int ldb, const EpiArgs& ea) {
    if (dtM == DT_F16) return "gemm_nt_kernel<f16>";
    if (g_force_regstage == 0 && gemm_nt_big_applicable(dtA, dtM, dtC, op, A, M, N, K, ldb, ea)) return "gemm_nt_big_kernel<bf16>";
    const int bk = dtM == DT_BF16 ? 32 : 16;
    const bool dma = op == OP_NONE && dtA == dtM && K % bk == 0 && ldb % (2 * bk) == 0 && ((uintptr_t)A) % 16 == 0;
    if (dma && g_force_regstage == 0 && dtM == DT_BF16 && gemm_nt_as_applicable(dtC, M, N, K, ldb, ea)) return gemm_nt_as_name(dtC, K, ea, M, N);
    if (dma && (g_force_regstage == 0 || g_force_regstage == 3)) return dtM == DT_F32 ? "gemm_nt_t_kernel<f32,f32>" : (dtC == DT_F32 ? "gemm_nt_t_kernel<bf16,f32>" : "gemm_nt_t_kernel<bf16,bf16>");
    if (dma && g_force_regstage != 1) return dtM == DT_F32 ? "gemm_nt_glds_kernel<f32,f32>" : (dtC == DT_F32 ? "gemm_nt_glds_kernel<bf16,f32>" : "gemm_nt_glds_kernel<bf16,bf16>");
    if (dtA == DT_F32 && dtM == DT_BF16) return "gemm_nt_kernel<f32,bf16,bf16>";
    if (dtM == DT_F32) return "gemm_nt_kernel<f32,f32,f32>";
    return dtC == DT_F32 ? "gemm_nt_kernel<bf16,bf16,f32>" : "gemm_nt_kernel<bf16,bf16,bf16>";
}
const char* gemm_tn_kernel_name(int dtA, int dtB, int dtM, int opA, int opB, int M, int Ka, int Nb, bool brs) {
    { int bs = 0; if (dtA == DT_BF16 && dtB == DT_BF16 && dtM == DT_BF16 && opA == OP_NONE && opB == OP_NONE && !g_force_tn_regstage && !g_dbg_tn && !g_tn_blocks && gemm_tn_big_plan(M, Ka, Nb, &bs) > 0) return "gemm_tn_big_kernel"; }
    if (dtA == DT_BF16 && dtB == DT_BF16 && dtM == DT_BF16 && opA == OP_NONE && opB == OP_NONE && M % 64 == 0 && Ka % 128 == 0 && Nb % 128 == 0 &&
        M >= 256 && !g_force_tn_regstage) return "gemm_tn_tr_kernel<0>";
    return dtM == DT_F32 ? "gemm_tn_kernel<f32,f32,f32>" : (dtA == DT_F32 ? "gemm_tn_kernel<f32,bf16,bf16>" : (dtB == DT_F32 ? "gemm_tn_kernel<bf16,f32,bf16>" : "gemm_tn_kernel<bf16,bf16,bf16>"));
}
