// NT GEMM for the compute-heavy shapes of config #4 (d = 512: K, N >= 512 at M = batch * frames >= 32768): C[M,N] = epi(A[M,K] . Bt[N,K]^T), bf16.
//
// The A-stationary kernel (gemm_as.hip) is built for the memory-bound shapes of configs[1] (K, N <= 768 at 170 FLOP per byte): it keeps the
// activation rows in registers and streams the weights.  At K = 512 / 1024 with N = 512 ... 2048 (340 ... 680 FLOP per byte) it runs at 0.6 - 0.7
// PFLOP/s (MFMA busy 0.19 - 0.36: profiles/r3_pmc_busy_cfg4.json) because every weight fragment read from LDS feeds one or two MFMAs only and
// nothing overlaps the epilogue.  This kernel is the classic two-operand tile instead:
//
//   * 256 x 256 output tile per workgroup, K step 64, 8 waves as 2 (rows) x 4 (columns); a wave owns 128 x 64 outputs = four 64 x 32 quadrants
//     (acc: 128 registers), one quadrant = one phase of 16 MFMAs (16x16x32 bf16);
//   * both operands through LDS by LDS-DMA (global_load_lds, 16 bytes per lane): two buffers of 64 KB (A 256 rows x 128 B, B the same), cut in
//     HALF tiles of 128 rows (16 KB = 2 DMA instructions per wave).  A wave's rows are 64 of each A half and its columns 32 of each B half (B halves interleave in 32-column runs: a wave's 64 columns are contiguous), so
//     quadrant (a, b) reads A half a and B half b, and the halves of a buffer fall free one after the other: A0 and B0 after phase 0, B1 after
//     phase 1, A1 after phase 2 (the B0 fragments stay in registers for phase 3).  One half tile is requested per phase, five to six phases
//     before its first read; a counted `s_waitcnt vmcnt(8)` per phase (never 0 in the loop) retires what was requested four phases earlier,
//     and the data is read one phase after that wait + barrier;
//   * the LDS image is lane-linear per DMA instruction (8 rows x 128 B); the swizzle (16-byte chunk c of row r at position c ^ ((r >> 1) & 7))
//     is applied to the SOURCE address.  ds_read_b128 serves 16 lanes per LDS cycle in the fixed groups {0-3, 12-15, 20-27}, ...: with this
//     swizzle each group covers all 16 sixteen-byte slots of the 256-byte bank row (conflict-free);
//   * the two waves of a SIMD (wave w and w + 4) run half a phase apart: while one does its 16 MFMAs the other issues its fragment reads and
//     DMA requests (two barriers per phase; the second half of the workgroup takes one barrier more before the loop, the first one more after it);
//   * epilogue: four passes of 32 rows x 64 columns per wave through a wave-private fp32 LDS stage, whole 128-byte lines per row, the shared fused
//     epilogue arithmetic (gemm_epi.h: bias, activation, dropout, row scale, act', residual, saved pre-activation); all its loads before its first store.
//
// Shapes: M % 256 == 0, N % 256 == 0, K % 128 == 0 (an even number of K steps), EPI_STD or the plain QKV head split, no operand prologue;
// everything else stays on gemm_as.hip / the 128 x 128 tile kernels.
#include <type_traits>
#include <cstdlib>
#include "kernels.h"
#include "gemm_epi.h"

#define BG_BUF 65536            // one buffer: A 32 KB | B 32 KB
#define BG_HALF 16384
#define BG_BOFF 32768

// operand base pointers of this lane: DMA instruction u (0, 1) of a half tile moves rows 8 * (8u + wid) .. + 7 of the half
struct BgSrc {
    const bf16* a[2];
    const bf16* b[2];
};

template <bool IS_B>
DEVI void bg_stage(const BgSrc& g, char* smem, int parity, int half, int tt, int K, int ldb, int wid) {
    const size_t off = (size_t)tt * 64 + (size_t)half * (IS_B ? (size_t)32 * ldb : (size_t)128 * K);        // B halves interleave in 32-column runs
    char* dst = smem + parity * BG_BUF + (IS_B ? BG_BOFF : 0) + half * BG_HALF + wid * 1024;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((IS_B ? g.b[u] : g.a[u]) + off),
                                         (__attribute__((address_space(3))) void*)(dst + u * 8192), 16, 0, 0);
}

template <int N> DEVI void bg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <typename TC>
__global__ __launch_bounds__(512, 1) void gemm_nt_big_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Bt, TC* __restrict__ C,
                                                              int M, int N, int K, int ldb, EpiArgs ea, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x 64 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int nMt = M >> 8, nNt = N >> 8;
    int mt, nt;
    {       // the column tiles of one row tile run back to back on ONE XCD (its L2 serves the A rows to all but the first)
        const int id = blockIdx.x;
        if ((nMt & 7) == 0) { const int xcd = id & 7, local = id >> 3; mt = (local / nNt) * 8 + xcd; nt = local % nNt; }
        else { mt = id / nNt; nt = id % nNt; }
    }
    const int m0 = mt << 8, n0 = nt << 8;
    const int T = K >> 6;               // K steps (even)

    BgSrc g;
    {
        const int f = ((wid & 1) << 2) + (lane >> 4);            // (row >> 1) & 7 of this lane's row in both DMA instructions
        const int chunk = (lane & 7) ^ f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = 8 * (8 * u + wid) + (lane >> 3);
            g.a[u] = A + (size_t)(m0 + r) * K + chunk * 8;
            g.b[u] = Bt + (size_t)(n0 + (r >> 5) * 64 + (r & 31)) * ldb + chunk * 8;      // B half h, LDS row r = column 64 (r >> 5) + 32 h + (r & 31)
        }
    }
    // fragment read addresses: row (lane & 15) of a 16-row block, chunk 4 ks + (lane >> 4) at position chunk ^ ((lane & 15) >> 1)
    int la[2][2], lb[2][2];             // [ks][buffer]: byte offsets into smem
    {
        const int r16 = lane & 15, gq = lane >> 4, f = r16 >> 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int pos = ((4 * ks + gq) ^ f) << 4;
                la[ks][p] = p * BG_BUF + (wr * 64 + r16) * 128 + pos;
                lb[ks][p] = p * BG_BUF + BG_BOFF + (wc * 32 + r16) * 128 + pos;
            }
    }

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bf0[2][2], bf1[2][2];

    // prologue: the whole of K step 0 and the first-needed halves of step 1
    bg_stage<false>(g, smem, 0, 0, 0, K, ldb, wid);
    bg_stage<true>(g, smem, 0, 0, 0, K, ldb, wid);
    bg_stage<true>(g, smem, 0, 1, 0, K, ldb, wid);
    bg_stage<false>(g, smem, 0, 1, 0, K, ldb, wid);
    bg_stage<false>(g, smem, 1, 0, 1, K, ldb, wid);
    bg_stage<true>(g, smem, 1, 0, 1, K, ldb, wid);
    bg_wait_vm<8>();                    // A0(0), B0(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the second half runs one barrier (half a phase) behind the first

#define BG_READ_A(P, HALF)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
            af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + la[ks][P] + (HALF) * BG_HALF + i * 2048);
#define BG_READ_B(DST, P, HALF)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
            DST[j][ks] = *reinterpret_cast<const bf16x8*>(smem + lb[ks][P] + (HALF) * BG_HALF + j * 2048);
#define BG_MMA(QA, QB, BF)                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
                acc[QA][QB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], BF[j][ks], acc[QA][QB][i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
    // one phase: [fragment reads] [one half-tile request] counted wait | barrier | reads complete, 16 MFMAs | barrier
#define BG_PHASE(READS, STAGE, VM, MMA)                                                                                      \
    READS                                                                                                                    \
    STAGE                                                                                                                    \
    bg_wait_vm<VM>();                                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
    MMA                                                                                                                      \
    __builtin_amdgcn_s_barrier();
    // K step t in buffer P: phase 0 requests B1(t+1), phase 1 A1(t+1) (other buffer), phase 2 A0(t+2), phase 3 B0(t+2) (this buffer)
#define BG_TILE(P, t, S0, S1, S2, S3, V0, V1, V2, V3)                                                                        \
    BG_PHASE(BG_READ_A(P, 0) BG_READ_B(bf0, P, 0), if (S0) bg_stage<true>(g, smem, (P) ^ 1, 1, (t) + 1, K, ldb, wid);, V0, BG_MMA(0, 0, bf0)) \
    BG_PHASE(BG_READ_B(bf1, P, 1), if (S1) bg_stage<false>(g, smem, (P) ^ 1, 1, (t) + 1, K, ldb, wid);, V1, BG_MMA(0, 1, bf1))               \
    BG_PHASE(BG_READ_A(P, 1), if (S2) bg_stage<false>(g, smem, P, 0, (t) + 2, K, ldb, wid);, V2, BG_MMA(1, 1, bf1))                          \
    BG_PHASE(, if (S3) bg_stage<true>(g, smem, P, 0, (t) + 2, K, ldb, wid);, V3, BG_MMA(1, 0, bf0))

    int t = 0;
    for (; t + 4 <= T; t += 2) {
        BG_TILE(0, t, true, true, true, true, 8, 8, 8, 8)
        BG_TILE(1, t + 1, true, true, true, true, 8, 8, 8, 8)
    }
    // last two K steps: nothing left to request after A1(T-1); the counted waits shrink with the requests still in flight
    BG_TILE(0, t, true, true, false, false, 8, 8, 6, 4)
    BG_TILE(1, t + 1, false, false, false, false, 2, 0, 0, 0)
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef BG_TILE
#undef BG_PHASE
#undef BG_MMA
#undef BG_READ_A
#undef BG_READ_B
    __builtin_amdgcn_s_barrier();       // every wave is done with the operand buffers

    if (dbg & 1) {          // ablation (tools/nt_big_bench.py): no epilogue
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) sacc += acc[a][b][i][j][0] + acc[a][b][i][j][3];
        if (sacc == 123.456f) C[0] = from_f<TC>(sacc);
        return;
    }
    // Epilogue.  A wave's outputs are rows m0 + 128a + 64wr .. +63 (a = 0, 1) x the 64 CONTIGUOUS columns n0 + 64wc .. (B halves interleave in
    // 32-column runs), so a row leaves as one whole 128-byte line.  Four passes of 32 rows through a wave-private fp32 LDS stage; a lane then owns
    // 8 consecutive columns of rows (lane >> 3) + 8q.  EVERY global read of the epilogue (bias, the residual or act' operand of all four passes:
    // 16 x 16 bytes per lane, the drop-path scale) is issued before the first store: loads and stores retire in order on one counter, and a
    // load issued behind a pass's stores would wait for them (measured with the loads inside the passes: 14 us per tile, as long as the K loop).
    constexpr int SLD = 68;
    float* stage = reinterpret_cast<float*>(smem) + wid * (32 * SLD);
    const int ncol = n0 + 64 * wc + (lane & 7) * 8;
    const int mrow = m0 + 64 * wr + (lane >> 3);                // + 128a + 32h + 8q
    EpiRows<TC, 4> er;
    er.n = ncol; er.nv = 8; er.fast = true;
#pragma unroll
    for (int e = 0; e < 8; ++e) er.bias[e] = ea.bias ? ea.bias[ncol + e] : 0.f;
    const float rs0 = ea.rowscale ? ea.rowscale[m0 / ea.T] : 1.f;          // the tile's 256 rows lie in one sample (launcher: T % 256 == 0)
    // QKV head split: q and k leave through the row chunks above; the v columns are written TRANSPOSED ([B, H, dh, T]) by a sweep over the stage
    // with a lane per column (8 consecutive frames = 16 bytes per store).  Its bias is read here, in front of the first store.
    const int vcol = n0 + 64 * wc + lane;
    int vh = 0, vpart = 0, vi = 0;
    float vbias = 0.f;
    if (ea.mode == EPI_QKV) {
        const int d = ea.H * ea.dh;
        if (ea.head_major) { vh = vcol / (3 * ea.dh); const int w = vcol - vh * 3 * ea.dh; vpart = w / ea.dh; vi = w - vpart * ea.dh; }
        else { vpart = vcol / d; const int w = vcol - vpart * d; vh = w / ea.dh; vi = w - vh * ea.dh; }
        vbias = ea.bias ? ea.bias[vcol] : 0.f;
    }
    const TC* opsrc = reinterpret_cast<const TC*>(ea.resid ? ea.resid : ea.aux);      // at most one of the two (launcher)
    const bool has_op = ea.resid != nullptr || ea.dact != DACT_NONE;
    typedef __attribute__((ext_vector_type(4))) unsigned int bg_u32x4;
    bg_u32x4 op[16];
    if (has_op) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                op[4 * p + q] = *reinterpret_cast<const bg_u32x4*>(opsrc + (size_t)(mrow + 128 * (p >> 1) + 32 * (p & 1) + 8 * q) * N + ncol);
    }
#define BG_EPI(P)                                                                                                            \
    {                                                                                                                        \
        constexpr int QA = (P) >> 1, H = (P) & 1;                                                                            \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                        \
            _Pragma("unroll") for (int bq = 0; bq < 2; ++bq)                                                                 \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                            \
                        stage[(16 * i + 4 * (lane >> 4) + r) * SLD + 32 * bq + 16 * j + (lane & 15)] = acc[QA][bq][2 * H + i][j][r]; \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                      \
            er.m[q] = mrow + 128 * QA + 32 * H + 8 * q;                                                                      \
            er.off[q] = (size_t)er.m[q] * N + ncol;                                                                          \
            er.rs[q] = rs0;                                                                                                  \
            if (has_op) {                                                                                                    \
                float t8[8];                                                                                                 \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
                    t8[2 * e] = __uint_as_float(op[4 * (P) + q][e] << 16); t8[2 * e + 1] = __uint_as_float(op[4 * (P) + q][e] & 0xffff0000u); \
                }                                                                                                            \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) { er.res[q][e] = t8[e]; er.ax[q][e] = t8[e]; }                 \
            }                                                                                                                \
        }                                                                                                                    \
        er.finish(stage + (lane >> 3) * SLD + (lane & 7) * 8, 8 * SLD, M, N, ea, C);                                         \
        if (ea.mode == EPI_QKV && vpart == 2) {                                                                              \
            _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                                               \
                const int mb = m0 + 64 * wr + 128 * QA + 32 * H + 8 * rg;                                                    \
                float v8[8];                                                                                                 \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) v8[e] = stage[(8 * rg + e) * SLD + lane] + vbias;              \
                const int bb = mb / ea.T, tt_ = mb - bb * ea.T;                                                              \
                store8(reinterpret_cast<TC*>(ea.vt) + ((size_t)(bb * ea.H + vh) * ea.dh + vi) * ea.T + tt_, v8);             \
            }                                                                                                                \
        }                                                                                                                    \
    }
    BG_EPI(0) BG_EPI(1) BG_EPI(2) BG_EPI(3)
#undef BG_EPI
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The weight-gradient (TN) form of the same tile: dW[Ka, Nb] (+)= A[M, Ka]^T . B[M, Nb], split over M, one 256 x 256 output tile and one
// M-split per workgroup, the partial tile written to the caller's fp32 slab (summed by the caller: launch_reduce_slabs2).
//
// Why: the 128 x 128 tile kernel (gemm.hip gemm_tn_tr_kernel) has 16 - 32 workgroups per M-split at d = 512 that all request the same operand
// rows; each keeps its own copy in flight in LDS, so of the 96 KB a CU has in flight only a quarter is distinct bytes, and the kernel runs at
// 2.9 TB/s of operand bytes whatever the L2 does (tools/tn_traffic.py: pacing the sibling workgroups so that L2 serves every re-read brought the
// fetched bytes from 1.7x to 1.0x of the operands and left the time unchanged).  With 256 x 256 tiles there are 4 - 16 workgroups per split.
//
// The K steps are 64 ROWS of the operands; a row is contiguous along the OUTPUT dimension, so the fragments (8 consecutive rows of one column per
// lane) come from transposing LDS reads (ds_read_b64_tr_b16), image and swizzle as in gemm_tn_tr_kernel: [32 rows][256 B] blocks per 128
// columns, 32-byte column blocks XOR tr_f(row).  Half tile a of an operand = its columns 128a .. 128a + 127 = [2 k-steps][32 rows][256 B] = 16 KB;
// phases, request schedule, counted waits and the stagger of the two wave groups are those of gemm_nt_big_kernel.
// Bias gradient: column sums of the B fragments by vector adds in the workgroups of output-row tile 0 (waves of the upper row group).
DEVI int bg_tr_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
typedef __attribute__((ext_vector_type(2))) unsigned bg_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned bg_u32x4t;

template <bool IS_B>
DEVI void bg_tn_stage(const BgSrc& g, char* smem, int parity, int half, int tt, int Ka, int Nb, int wid) {
    const size_t off = (size_t)tt * 64 * (IS_B ? Nb : Ka) + (size_t)half * 128;
    char* dst = smem + (IS_B ? 65536 : 0) + parity * 32768 + half * BG_HALF + wid * 1024;          // layout: A buffer 0 | A buffer 1 | B buffer 0 | B buffer 1 (the buffer is an instruction offset of the fragment reads)
#pragma unroll
    for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((IS_B ? g.b[u] : g.a[u]) + off),
                                         (__attribute__((address_space(3))) void*)(dst + u * 8192), 16, 0, 0);
}

__global__ __launch_bounds__(512, 1) void gemm_tn_big_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, float* __restrict__ slab, int want_bias,
                                                              int M, int Ka, int Nb, int rows_per_split, int tiles, int nsplits, const float* __restrict__ brs, int brsT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x 64 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int nNt = Nb >> 8;
    int tile, split;
    {       // the output tiles of one M-split read the same rows: back to back on ONE XCD
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        if ((nsplits & 7) == 0) { split = (j / tiles) * 8 + xcd; tile = j % tiles; }
        else { split = id / tiles; tile = id % tiles; }
    }
    const int kt = tile / nNt, nt = tile % nNt;
    const int k0 = kt << 8, n0 = nt << 8;
    const int m_beg = split * rows_per_split;
    const int T = rows_per_split >> 6;          // K steps of 64 rows (even: launcher)
    const size_t sstride = (size_t)Ka * Nb + Nb;

    BgSrc g;        // DMA instruction u of a half tile moves the four rows 4 (8u + wid) .. + 3 of the K step: 16 slots of 16 bytes per row
    {
        const int r = lane >> 4, sp = lane & 15;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = 4 * (8 * u + wid) + r;
            const int col = (((sp >> 1) ^ bg_tr_f(row)) << 4) + ((sp & 1) << 3);      // logical column of physical slot sp
            g.a[u] = A + (size_t)(m_beg + row) * Ka + k0 + col;
            g.b[u] = B + (size_t)(m_beg + row) * Nb + n0 + col;
        }
    }
    // transposing fragment reads: lane (g, q, p) addresses row 8g + q (+4 at +1024, +32 at +8192), 8 bytes at p of the 32-byte column block
    unsigned fa[4], fb[2];              // [16-column block]: LDS byte addresses in buffer 0
    {
        const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, r0 = 8 * gq + q;
        const unsigned base = (unsigned)(uintptr_t)smem + r0 * 256 + (pp << 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = base + (((wr * 4 + i) ^ bg_tr_f(r0)) << 5);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = base + 65536 + (((wc * 2 + j) ^ bg_tr_f(r0)) << 5);
    }

    f32x4 acc[2][2][4][2];
    float csum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};          // bias gradient: this lane's share (rows 8g .. 8g + 7 of every 32) of the column sums of B columns n0 + 128b + 32wc + 16j + (lane & 15)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const bool bias_wave = want_bias && kt == 0 && wr == 0;
    // brs != nullptr: the bias gradient is the column sum of brs[m / brsT] * B[m, :] (the drop-path scale of the sample a row belongs to;
    // brsT % 128 == 0: the two K steps of a loop iteration lie in one sample) — the scale is a scalar register, reloaded at sample boundaries
    float bsc = 1.f;
    int brow = 0, bidx = 0;
    if (brs) { bidx = m_beg / brsT; brow = m_beg - bidx * brsT; bsc = brs[bidx]; }
    bg_u32x4t af[4][2], bf0[2][2], bf1[2][2];       // [block][k-step of 32 rows]

    bg_tn_stage<false>(g, smem, 0, 0, 0, Ka, Nb, wid);
    bg_tn_stage<true>(g, smem, 0, 0, 0, Ka, Nb, wid);
    bg_tn_stage<true>(g, smem, 0, 1, 0, Ka, Nb, wid);
    bg_tn_stage<false>(g, smem, 0, 1, 0, Ka, Nb, wid);
    bg_tn_stage<false>(g, smem, 1, 0, 1, Ka, Nb, wid);
    bg_tn_stage<true>(g, smem, 1, 0, 1, Ka, Nb, wid);
    bg_wait_vm<8>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

#define BGT_RD(DST, ADDR, OFF)                                                                                               \
    {                                                                                                                        \
        bg_u32x2 lo_, hi_;                                                                                                   \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo_) : "v"(ADDR), "n"(OFF));                               \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi_) : "v"(ADDR), "n"((OFF) + 1024));                      \
        DST = bg_u32x4t{lo_.x, lo_.y, hi_.x, hi_.y};                                                                         \
    }
#define BGT_READ_A(P, HALF)                                                                                                  \
    BGT_RD(af[0][0], fa[0], (P) * 32768 + (HALF) * BG_HALF) BGT_RD(af[0][1], fa[0], (P) * 32768 + (HALF) * BG_HALF + 8192)                         \
    BGT_RD(af[1][0], fa[1], (P) * 32768 + (HALF) * BG_HALF) BGT_RD(af[1][1], fa[1], (P) * 32768 + (HALF) * BG_HALF + 8192)                         \
    BGT_RD(af[2][0], fa[2], (P) * 32768 + (HALF) * BG_HALF) BGT_RD(af[2][1], fa[2], (P) * 32768 + (HALF) * BG_HALF + 8192)                         \
    BGT_RD(af[3][0], fa[3], (P) * 32768 + (HALF) * BG_HALF) BGT_RD(af[3][1], fa[3], (P) * 32768 + (HALF) * BG_HALF + 8192)
#define BGT_READ_B(DST, P, HALF)                                                                                             \
    BGT_RD(DST[0][0], fb[0], (P) * 32768 + (HALF) * BG_HALF) BGT_RD(DST[0][1], fb[0], (P) * 32768 + (HALF) * BG_HALF + 8192)                       \
    BGT_RD(DST[1][0], fb[1], (P) * 32768 + (HALF) * BG_HALF) BGT_RD(DST[1][1], fb[1], (P) * 32768 + (HALF) * BG_HALF + 8192)
    // the reads have landed: every fragment register passes through the wait, so that no consumer is scheduled in front of it
#define BGT_WAIT()                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]), "+v"(af[3][1]) :: "memory"); \
    asm volatile("" : "+v"(bf0[0][0]), "+v"(bf0[0][1]), "+v"(bf0[1][0]), "+v"(bf0[1][1]), "+v"(bf1[0][0]), "+v"(bf1[0][1]), "+v"(bf1[1][0]), "+v"(bf1[1][1]));
#define BGT_MMA(QA, QB, BF, BIAS)                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
                acc[QA][QB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i][ks]), __builtin_bit_cast(bf16x8, BF[j][ks]), acc[QA][QB][i][j], 0, 0, 0); \
    if ((BIAS) && bias_wave) {          /* vector adds behind the MFMAs' issue */                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                      \
            float t_ = 0.f;                                                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                 \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                \
                    t_ += __uint_as_float(BF[j][ks][e] << 16) + __uint_as_float(BF[j][ks][e] & 0xffff0000u);                 \
            csum[QB][j] += bsc * t_;                                                                                         \
        }                                                                                                                    \
    }                                                                                                                        \
    __builtin_amdgcn_s_setprio(0);
#define BGT_PHASE(READS, STAGE, VM, MMA)                                                                                     \
    READS                                                                                                                    \
    STAGE                                                                                                                    \
    bg_wait_vm<VM>();                                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                            \
    BGT_WAIT()                                                                                                               \
    MMA                                                                                                                      \
    __builtin_amdgcn_s_barrier();
#define BGT_TILE(P, t, S0, S1, S2, S3, V0, V1, V2, V3)                                                                       \
    BGT_PHASE(BGT_READ_A(P, 0) BGT_READ_B(bf0, P, 0), if (S0) bg_tn_stage<true>(g, smem, (P) ^ 1, 1, (t) + 1, Ka, Nb, wid);, V0, BGT_MMA(0, 0, bf0, true)) \
    BGT_PHASE(BGT_READ_B(bf1, P, 1), if (S1) bg_tn_stage<false>(g, smem, (P) ^ 1, 1, (t) + 1, Ka, Nb, wid);, V1, BGT_MMA(0, 1, bf1, true))                \
    BGT_PHASE(BGT_READ_A(P, 1), if (S2) bg_tn_stage<false>(g, smem, P, 0, (t) + 2, Ka, Nb, wid);, V2, BGT_MMA(1, 1, bf1, false))                          \
    BGT_PHASE(, if (S3) bg_tn_stage<true>(g, smem, P, 0, (t) + 2, Ka, Nb, wid);, V3, BGT_MMA(1, 0, bf0, false))

#define BGT_NEXT_SAMPLE() if (brs) { brow += 128; if (brow >= brsT) { brow -= brsT; ++bidx; bsc = brs[min(bidx, (M - 1) / brsT)]; } }
    int t = 0;
    for (; t + 4 <= T; t += 2) {
        BGT_TILE(0, t, true, true, true, true, 8, 8, 8, 8)
        BGT_TILE(1, t + 1, true, true, true, true, 8, 8, 8, 8)
        BGT_NEXT_SAMPLE()
    }
    BGT_TILE(0, t, true, true, false, false, 8, 8, 6, 4)
    BGT_TILE(1, t + 1, false, false, false, false, 2, 0, 0, 0)
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef BGT_TILE
#undef BGT_NEXT_SAMPLE
#undef BGT_PHASE
#undef BGT_MMA
#undef BGT_WAIT
#undef BGT_READ_A
#undef BGT_READ_B
#undef BGT_RD
    __builtin_amdgcn_s_barrier();

    // the partial tile -> this split's slab: quadrant (a, b) = output rows k0 + 128a + 64wr .. +63, columns n0 + 128b + 32wc .. +31 (one 128-byte
    // line per row), through a wave-private fp32 stage so that a store instruction writes 8 whole lines
    constexpr int SLD = 36;
    float* stage = reinterpret_cast<float*>(smem) + wid * (64 * SLD);
    float* sl = slab + (size_t)split * sstride;
#define BGT_EPI(QA, QB)                                                                                                      \
    {                                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                    \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                \
                    stage[(16 * i + 4 * (lane >> 4) + r) * SLD + 16 * j + (lane & 15)] = acc[QA][QB][i][j][r];               \
        float* orow = sl + (size_t)(k0 + 128 * (QA) + 64 * wr + (lane >> 3)) * Nb + n0 + 128 * (QB) + 32 * wc + (lane & 7) * 4; \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                                        \
            *reinterpret_cast<float4*>(orow + (size_t)(8 * q) * Nb) = *reinterpret_cast<const float4*>(stage + ((lane >> 3) + 8 * q) * SLD + (lane & 7) * 4); \
    }
    BGT_EPI(0, 0) BGT_EPI(0, 1) BGT_EPI(1, 0) BGT_EPI(1, 1)
#undef BGT_EPI
    if (bias_wave) {          // fold the four row groups of a column
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t_ = csum[b][j];
                t_ += __shfl_xor(t_, 16, 64);
                t_ += __shfl_xor(t_, 32, 64);
                if (lane < 16) sl[(size_t)Ka * Nb + n0 + 128 * b + 32 * wc + 16 * j + lane] = t_;
            }
    }
}

extern int g_nt_big;
int g_tn_big = 1;
// rows per M-split of the big weight-gradient kernel (0: the shape is not one it takes)
int gemm_tn_big_plan(int M, int Ka, int Nb, int* splits_out) {
    if (!g_nt_big || !g_tn_big || M < 32768 || Ka % 256 != 0 || Nb % 256 != 0 || Ka < 512 || Nb < 512) return 0;
    const int tiles = (Ka >> 8) * (Nb >> 8);
    if (tiles > 32) return 0;
    // as close to one workgroup per CU as whole, even numbers of 64-row K steps allow; a multiple of 8 keeps the XCD-aware (tile, split) order
    int splits = 0;
    for (int sp = 256 / tiles; sp >= 8; --sp)
        if (M % (sp * 128) == 0 && (sp & 7) == 0) { splits = sp; break; }
    if (!splits || tiles * splits < 160) return 0;
    *splits_out = splits;
    return M / splits;
}
// returns 1 when the shape is not one this kernel takes; on 0 the caller sums `splits` slabs of stride Ka * Nb + Nb
int launch_gemm_tn_big(const void* A, const void* B, float* slab, int want_bias, int M, int Ka, int Nb, int* splits_out, const float* brs, int brsT, hipStream_t s) {
    int splits = 0;
    const int rps = gemm_tn_big_plan(M, Ka, Nb, &splits);
    if (rps <= 0) return 1;
    static bool prepared = false, ok = false;
    if (!prepared) {
        prepared = true;
        ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BG_BUF) == hipSuccess;
    }
    if (!ok) return 1;
    const int tiles = (Ka >> 8) * (Nb >> 8);
    hipLaunchKernelGGL(gemm_tn_big_kernel, dim3(tiles * splits), dim3(512), 2 * BG_BUF, s, (const bf16*)A, (const bf16*)B, slab, want_bias, M, Ka, Nb, rps, tiles, splits, brs, brsT);
    *splits_out = splits;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern int g_nt_big;
static int big_env() { const char* e = getenv("ISHARA_NT_BIG"); return e ? atoi(e) : 1; }
int g_nt_big = big_env();           // 0: never (A/B runs, ishara_debug_set_nt_big)

bool gemm_nt_big_applicable(int dtA, int dtM, int dtC, int op, const void* A, int M, int N, int K, int ldb, const EpiArgs& ea) {
    return g_nt_big != 0 && dtA == DT_BF16 && dtM == DT_BF16 && dtC == DT_BF16 && op == OP_NONE && M >= 32768 && M % 256 == 0 && N % 256 == 0 && N >= 512 &&
           K >= 512 && K % 128 == 0 && ldb % 8 == 0 && ldb >= K && ((uintptr_t)A) % 16 == 0 && (ea.mode == EPI_STD || (ea.mode == EPI_QKV && ea.dh % 8 == 0 && ea.T % 8 == 0 && !ea.pre_out && !ea.act && !ea.drop.thr && !ea.rowscale && !ea.resid && ea.dact == DACT_NONE)) && !ea.ln_gamma && !ea.pa_P && !ea.ldc && !ea.n_valid && !ea.dbg && !ea.addtab &&
           !(ea.resid && ea.dact != DACT_NONE) && (!ea.rowscale || (ea.T > 0 && ea.T % 256 == 0));
}

template <typename TC>
static int run_big(const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s) {
    static bool prepared = false, ok = false;
    if (!prepared) {
        prepared = true;
        ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_big_kernel<TC>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BG_BUF) == hipSuccess;
    }
    if (!ok) return 1;
    hipLaunchKernelGGL((gemm_nt_big_kernel<TC>), dim3((M >> 8) * (N >> 8)), dim3(512), 2 * BG_BUF, s, (const bf16*)A, (const bf16*)Bt, (TC*)C, M, N, K, ldb, ea, g_nt_big >> 1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// returns 1 when the shape is not one this kernel takes (the caller goes on to the other kernels)
int launch_gemm_nt_big(int dtA, int dtM, int dtC, int op, const void* A, const void* Bt, void* C, int M, int N, int K, int ldb, const EpiArgs& ea, hipStream_t s) {
    if (!gemm_nt_big_applicable(dtA, dtM, dtC, op, A, M, N, K, ldb, ea)) return 1;
    return run_big<bf16>(A, Bt, C, M, N, K, ldb, ea, s);
}
