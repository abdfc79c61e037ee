// Kernel launchers of the ishara_amd HIP library.  Host-callable; every launch is
// asynchronous on the given stream, allocates nothing and never synchronises.
#pragma once
#include "common.h"

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };       // DT_F16: forward (inference) kernels only
static inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }
static inline bool dt_is16(int dt) { return dt != DT_F32; }

// ---- operand transforms applied while staging a GEMM operand -------------------
enum : int { OP_NONE = 0, OP_SWISH = 1, OP_COLAFFINE = 2, OP_ROWSCALE = 3, OP_DROPMASK = 4 };
struct OpArgs {
    const float* c1 = nullptr;   // OP_COLAFFINE: v*c1[col]+c0[col]
    const float* c0 = nullptr;
    const float* rs = nullptr;   // OP_ROWSCALE: v*rs[row / T]
    int T = 1;
    DropSpec drop = {0, 0, 1.f}; // OP_DROPMASK: v*mask(row, col)
};

// ---- GEMM epilogue (NT kernel) ---------------------------------------------------
enum : int { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2 };
enum : int { DACT_NONE = 0, DACT_SWISH = 1, DACT_POS = 2 };
enum : int { EPI_STD = 0, EPI_QKV = 1 };
struct EpiArgs {
    const float* bias = nullptr;      // [N]
    const float* addtab = nullptr;    // += addtab[(m % tab_period)*N + n]
    int tab_period = 1;
    void* pre_out = nullptr;          // save pre-activation (TC, [M,N])
    int act = ACT_NONE;
    DropSpec drop = {0, 0, 1.f};      // elementwise inverted dropout (row=m, col=n)
    const float* rowscale = nullptr;  // *= rowscale[m / T]
    int T = 1;
    int rowscale_bias = 0;            // 1: the row scale multiplies the BIAS only (the A operand already carries it): acc + bias * rowscale[m / T]
    int dact = DACT_NONE;             // *= act'(aux[m,n])
    const void* aux = nullptr;        // TC, [M,N]
    const void* resid = nullptr;      // TC, [M,N]
    int mode = EPI_STD;               // EPI_QKV: scatter to q,k [B,H,T,dh] and vt [B,H,dh,T]
    void* q = nullptr; void* k = nullptr; void* vt = nullptr;
    int H = 1, dh = 1;
    int dbg = 0;                      // ablation bits (tools/gemm_ablate.py): 1 skip epilogue, 2 skip MFMA, 4 skip loads
    // ---- prologues of the A-stationary kernel (gemm_as.hip): the operand rows are transformed in registers after the load, and the
    // transformed rows are written out once (the weight-gradient GEMM of the backward pass needs them in memory)
    const float* ln_gamma = nullptr;  // LayerNorm over K (the wave holds whole rows): A' = (A - mean) * rstd * gamma + beta
    const float* ln_beta = nullptr;
    float ln_eps = 0.f;
    float* ln_mean = nullptr;         // [M] row statistics for the LayerNorm backward (may be null)
    float* ln_rstd = nullptr;
    const float* pa_P = nullptr;      // per-sample affine: A' = A * pa_P[m / T, k] + pa_Q[m / T, k]   (BatchNorm + ECA gate of a Conv1DBlock)
    const float* pa_Q = nullptr;
    void* pro_out = nullptr;          // A' rows [M, K] in the operand type (null: not needed, inference)
    int head_major = 1;               // 1: cols = h*3dh + {q,k,v}*dh + i (TF path); 0: {q,k,v}*d + h*dh + i (torch twin)
    // A-stationary kernel only: C rows are ldc elements apart (0: N) and only the columns below n_valid are stored (0: all) — a 60-column
    // classifier runs as N = 64 over the zero-padded weight shadow (no residual / act' operands with these)
    int ldc = 0, n_valid = 0;
    // A-stationary kernel, 16-bit C: bit 0 — a row's two 64-byte halves of a 128-byte line are stored back to back every second column step
    // instead of one step apart (cold HBM takes half lines that arrive a step apart at 3.8 TB/s, whole lines at 4.6+: tools/micro/store_cold.hip);
    // bit 1 — non-temporal hint on the side outputs (saved pre-activations, prologue rows); bits 4 / 5 — the chunked form (one workgroup per CU, 64 KB
    // weight stages, a wait + barrier per stage instead of per column step) at K = 256 / 512.  -1: the library default (gemm_as.hip, ISHARA_AS_FLAGS)
    int as_flags = -1;
};

// C[M,N] = epi( op(A)[M,K] . Bt[N,K]^T ).  Bt is a padded weight shadow: rows padded to
// a multiple of 128, row stride ldb (elements) a multiple of the K tile (64 bf16 / 32 f32).
int launch_gemm_nt(int dtA, int dtM, int dtC, int op, const void* A, const void* Bt, void* C,
                   int M, int N, int K, int ldb, const OpArgs& oa, const EpiArgs& ea, hipStream_t s);

// dW[Ka,Nb] += opA(A)[M,Ka]^T . opB(B)[M,Nb]   (fp32 accumulate into `out`, row-major [Ka,Nb]);
// dbias[Nb] += colsum(opB(B)) when dbias != nullptr.  slab = fp32 scratch of
// gemm_tn_slab_floats(...) floats.
size_t gemm_tn_slab_floats(int M, int Ka, int Nb, int dtM);
// ---- batched, deferred column sums of partial rows (parameter gradients of LayerNorm / depthwise conv: ~45 launches of ~5 us per step, all off
// the backward pass's dependency chain).  While a RedSink is installed (g_red_sink), launch_reduce_slabs / _slabs2 record the job instead of
// launching; the caller keeps every recorded slab intact (a bump arena) until launch_reduce_flush sums all of them in ONE launch, each output element in a fixed
// order (run-to-run bit-identical gradients; not the same order as the single launches).
#define RED_MAXJOBS 48
struct RedJob { const float* slab; float* out0; float* out1; size_t stride; int n0, n, splits, nb, nbv, first_block; };
struct RedSink { RedJob job[RED_MAXJOBS]; int njobs = 0; int nblocks = 0; };
extern RedSink* g_red_sink;
bool reduce_sink_full();                                   // no room for another job: flush before the next deferring operator
int launch_reduce_flush(RedSink* sink, hipStream_t s);
bool gemm_tn_bias_rowscale_ok(int dtA, int dtB, int dtM, int M, int Ka, int Nb, int T);   // the transposed-read kernel takes the call and T % 32 == 0
// Deferred slab sums of the transposed-read wgrad kernel: with a TnDefer the launch writes slab[turn] (two caller-owned buffers of
// gemm_tn_slab_floats floats each) and the sums of ITS slabs are carried by the next deferring launch as extra workgroups;
// launch_gemm_tn_flush sums what is still pending.  Launches that take another kernel ignore it (their sums run at once on `slab`).
struct TnDefer { float* slab[2] = {nullptr, nullptr}; int turn = 0; bool pending = false;
                 const float* p_slab = nullptr; float* p_out0 = nullptr; float* p_out1 = nullptr; int p_n0 = 0, p_n = 0, p_splits = 0; size_t p_stride = 0; int p_nb = 0, p_nbv = 0; };
int launch_gemm_tn_flush(TnDefer* defer, hipStream_t s);
// Per-sample affine of the A operand inside the transposed-read wgrad kernel (the project conv of a Conv1DBlock, c5:41-89): the operand the
// forward pass multiplied was A'[m,k] = A[m,k] * P[b,k] + Q[b,k] (BatchNorm . ECA gate . drop-path, b = m / T), but only A is in memory.
// The kernel accumulates A_b^T B_b per sample, folds it into the total with P[b,:] (and Q[b,:] x the sample's column sums of B) at every
// sample boundary, and from the same per-sample accumulator emits what the BatchNorm / ECA backward needs from the gradient of A'
// (dA' = rs[b] * B W^T, never read back): Rpart[b][p][k] = sum over the 64-column group p of W[k,n] * (A_b^T B_b)[k,n], i.e. partial
// sums of sum_t dA'[b,t,k] * A[b,t,k] / rs[b], and G[b,n] = sum_t B[b,t,n].
struct TnPsa { const float* P = nullptr; const float* Q = nullptr;     // [B, Ka]
               const void* W = nullptr; int ldw = 0;                     // bf16 [Ka][ldw]: the dgrad's weight shadow
               float* G = nullptr; float* Rpart = nullptr; int T = 0;     // [B, Nb], [B][Nb / 64][Ka]
               int dbg = 0; };                                           // timing ablation (ISHARA_PSA_DBG; results are wrong): 1 no per-sample work, 2 no column sums, 4 no setup loads, 16 no write-out
bool gemm_tn_psa_ok(int dtA, int dtB, int dtM, int M, int Ka, int Nb, int T);
int launch_gemm_tn(int dtA, int dtB, int dtM, int opA, int opB, const void* A, const void* B,
                   float* out, float* dbias, float* slab, int M, int Ka, int Nb,
                   const OpArgs& oa, const OpArgs& ob, hipStream_t s, int ka_valid = 0, int nb_valid = 0,
                   const float* bias_rowscale = nullptr, int bias_T = 0, TnDefer* defer = nullptr, const TnPsa* psa = nullptr);   // bias_rowscale: dbias = sum_m bias_rowscale[m / bias_T] * B[m,:] (gemm_tn_bias_rowscale_ok shapes only); ka_valid < Ka: A columns [ka_valid, Ka) are zero padding, out has ka_valid rows; nb_valid < Nb: same for B / out columns / dbias
// C[M, N] (f32) = A[M, K] (bf16 / fp16) . Wt[N, K]^T + bias for a NARROW output (N <= 64: the classifier) and few rows: a lane per output
// column, four rows per wave, the weight row streamed from L2 — the 64 x 128 tile kernel needs 37 us for M = 384, N = 60, K = 512
// (6 workgroups); this one is for the latency of a B = 1 clip, not for throughput (M <= 4096)
int launch_dense_narrow(int dt, const void* A, const void* Wt, int ldt, const float* bias, float* C, int M, int N, int K, hipStream_t s);
// xb[M, Kp] (bf16) = x[M, F] (f32), zero padded to Kp columns (F % 4 == 0, Kp % 8 == 0)
int launch_pack_rows_bf16(const float* x, void* xb, int M, int F, int Kp, hipStream_t s, int dt = DT_BF16);   // dt: DT_BF16 or DT_F16 (the fp16 inference path)

bool gemm_nt_as_applicable(int dtC, int M, int N, int K, int ldb, const EpiArgs& ea);
// the kernel also takes ea's prologue (ln_* / pa_*): bf16 output, K in {256, 512}, whole 16-row tiles inside one sample for pa_*
bool gemm_nt_as_prologue_ok(int dtA, int dtM, int dtC, int M, int N, int K, int ldb, const EpiArgs& ea);   // gemm_as.hip: the A-stationary kernel takes this shape
const char* gemm_nt_kernel_name(int dtA, int dtM, int dtC, int op, const void* A, int M, int N, int K, int ldb, const EpiArgs& ea);
extern int g_force_tn_regstage;   // tests: 1 forces the register-transposing TN kernel
extern int g_tn_phase;   // 0 GEMM + slab sums, 1 GEMM kernel only, 2 slab sums only
const char* gemm_tn_kernel_name(int dtA, int dtB, int dtM, int opA, int opB, int M, int Ka, int Nb, bool brs = false);

// weight shadows: Wt[Np][Kp] (transposed) and Wn[Kp2][Np2] (as-is, padded) in dtM
int launch_make_shadow(int dtM, const float* W, int K, int N, void* Wt, int ldt, void* Wn, int ldn, hipStream_t s);
struct ShadowDesc { const float* W; void* Wt; void* Wn; int K, N, ldt, ldn, tile0, tiles_n; };   // tile0: first 32x32 tile (block) of this weight
int launch_make_shadow_batched(int dtM, const ShadowDesc* tab, int ntab, int total_tiles, hipStream_t s);

// ---- normalisation / conv / small ops (elementwise.hip) ---------------------------
int launch_log_softmax_fwd(const float* x, float* y, int M, int C, int ld, hipStream_t s);
int launch_log_softmax_bwd(const float* dy, const float* y, float* dx, int M, int C, int ld, hipStream_t s);
int launch_layernorm_fwd(int dt, const void* x, const float* gamma, const float* beta, float eps,
                         void* y, float* mean, float* rstd, int M, int C, hipStream_t s);
// dx = LN'(dy) (+ resid) ; dgamma/dbeta += column sums: through per-block partial rows in `scratch`
// (layernorm_bwd_scratch_floats(C) floats) + a slab reduce, or same-address atomics if scratch == nullptr
size_t layernorm_bwd_scratch_floats(int C);
int launch_layernorm_bwd(int dt, const void* dy, const void* x, const float* mean, const float* rstd,
                         const float* gamma, const void* resid, void* dx, float* dgamma, float* dbeta,
                         float* scratch, int M, int C, hipStream_t s);

enum : int { DWIN_NONE = 0, DWIN_SWISH = 1, DWIN_GLU = 2 };
// y[b,t,c] = bias[c] + sum_j w[j,c] * in(x)[b, t - padl + j, c]; x has Cin = C (or 2C for GLU).
// stats: ssum/ssq [B,C] = per-sample sum_t y, sum_t y^2 (fp32 atomics; zero them first; nullptr = off).
// ssum / ssq: per-sample channel sums of y and y^2 [B, C] (or nullptr).  `part`: scratch of dwconv_fwd_scratch_floats(B, T, C)
// floats -> deterministic sums without atomics or zero fills; nullptr -> the caller zero-fills ssum / ssq, float atomics
int launch_dwconv_fwd(int dt, int inop, const void* x, const float* w, const float* bias, void* y,
                      float* ssum, float* ssq, float* part, int B, int T, int C, int k, int padl, hipStream_t s, int* part_rows = nullptr);
// part_rows != nullptr: the partial statistic rows stay unsummed in `part` ([B][rows][2][C]: sum | sum of squares), *part_rows = rows per sample
size_t dwconv_fwd_scratch_floats(int B, int T, int C);
// dx = d in(x) ; dw [k,C], dbias [C] accumulated (through `scratch` partial rows of
// dwconv_bwd_scratch_floats(C,k) floats when given, else atomically).
size_t dwconv_bwd_scratch_floats(int C, int k);
// BatchNorm backward folded into the depthwise-conv backward: dy is the gradient of BN(y), h = y (the conv's forward output, [B,T,C]);
// the kernel applies dy <- a[c] * (dy*sg[b,c] + E[b or 0,c] - xhat*Fc[c]) (the arithmetic of launch_bn_bwd_apply) to each dy row as it
// loads it.  h == nullptr: plain depthwise-conv backward.
struct DwBnArgs { const void* h = nullptr; const float* mean = nullptr; const float* rstd = nullptr; const float* a = nullptr;
                  const float* sg = nullptr; const float* E = nullptr; const float* Fc = nullptr; int e_per_sample = 0; };
// returns 1 when the fused one-pass kernel took the call (dx, dw, dbias written), 0 when this shape has no fused kernel (the caller
// runs launch_bn_bwd_apply + launch_dwconv_bwd instead), < 0 on error
int launch_dwconv_bwd_bn(int dt, int inop, const void* dy, const DwBnArgs& bn, const void* x, const float* w, void* dx,
                         float* dw, float* dbias, float* scratch, int B, int T, int C, int k, int padl, hipStream_t s);
int launch_dwconv_bwd(int dt, int inop, const void* dy, const void* x, const float* w, void* dx,
                      float* dw, float* dbias, float* scratch, int B, int T, int C, int k, int padl, hipStream_t s);

// BN finalize from per-sample sums ssum/ssq [nb,C] (count = rows they cover): mean, rstd,
// a = gamma*rstd, b = beta - mean*a ; moving statistics update when training
int launch_bn_finalize(const float* ssum, const float* ssq, int nb, float count, const float* gamma, const float* beta,
                       float eps, float momentum, float* moving_mean, float* moving_var, int training,
                       float* mean, float* rstd, float* a, float* b, int C, hipStream_t s, float var_corr = 1.f, int stride = 0);   // var_corr: factor on the batch variance
                                                                                                                       // entering moving_var (torch: n/(n-1)); stride: floats between rows of ssum / ssq (0: C)
// ECA fwd on [B,C]: g = a*gap/T + b ; s = sigmoid(conv5(g)) ; P = a*s ; Q = b*s
int launch_eca_fwd(const float* gap, const float* a, const float* b, const float* w5, float invT,
                   float* gn, float* sgate, float* P, float* Q, int B, int C, hipStream_t s, float* rs = nullptr, DropSpec dp = {0, 0, 1.f}, int dp_fold = 0);
// rs != nullptr: the kernel also draws the per-sample drop-path scale rs[b] (dp) and, with dp_fold, scales P and Q by it
// training form over the depthwise conv's partial statistic rows part[B][prows][2][C] (no stats_reduce launch): per-sample sums -> gap_out [B, C]
int launch_eca_fwd_part(const float* part, int prows, float* gap_out, const float* a, const float* b, const float* w5, float invT,
                        float* gn, float* sgate, float* P, float* Q, int B, int C, hipStream_t s, float* rs = nullptr, DropSpec dp = {0, 0, 1.f}, int dp_fold = 0);
// inference form: the sample's channel sums come from the depthwise conv's partial rows, a / b from the BatchNorm's moving statistics
int launch_eca_fwd_infer(const float* part, int prows, const float* mm, const float* mv, const float* gamma, const float* beta, float eps, const float* w5, float invT,
                         float* gn, float* sgate, float* P, float* Q, int B, int C, hipStream_t s);
// y = x*P[b,c] + Q[b,c] (+ resid)    (P,Q per sample) ; if Q == nullptr -> no offset
int launch_sample_affine(int dt, const void* x, const float* P, const float* Q, const void* resid, void* y,
                         int B, int T, int C, hipStream_t s);
// y = x*a[c] + b[c]
int launch_col_affine(int dt, const void* x, const float* a, const float* b, void* y, int M, int C, hipStream_t s);
enum : int { MAP_SWISH = 0, MAP_ROWSCALE = 1, MAP_DROPMASK = 2 };
// y = swish(x) | x * rs[row / T] | x * dropmask(row, col)   — one streaming pass over [M, C]
int launch_map_rows(int dt, int op, const void* x, void* y, const float* rs, DropSpec drop, int M, int T, int C, hipStream_t s);
// per-sample reductions over t: S1[b,c] = sum_t dy ; S2[b,c] = sum_t dy * other   (other optional,
// normalised as (other-mean[c])*rstd[c] when mean != nullptr)
int launch_sample_reduce(int dt, const void* dy, const void* other, const float* mean, const float* rstd,
                         float* S1, float* S2, int B, int T, int C, hipStream_t s);
// Conv1DBlock BN+ECA backward finalize (one block; small)
// PsaStats: S1, S2 are not inputs but computed first, per sample, from what the per-sample-affine weight-gradient GEMM emitted (TnPsa):
// S1[b,c] = rs[b] * sum_n Wt[n,c] G[b,n] (= sum_t dh4[b,t,c]), S2[b,c] = rstd[c] * (rs[b] * sum_p Rpart[b][p][c] - mean[c] * S1[b,c]); Wt = bf16 [N][ldt]
#define ECA_MAX_CHUNKS 4          // eca_bwd_sample_kernel splits the channels of a sample over up to 4 workgroups: dw5part holds B * 4 * 8 floats
struct PsaStats { const float* G = nullptr; const float* Rpart = nullptr; int nparts = 0; const void* Wt = nullptr; int ldt = 0, N = 0;
                  const float* rs = nullptr; const float* mean = nullptr; const float* rstd = nullptr; };
int launch_eca_bn_bwd_finalize(float* S1, float* S2, const float* gap, const float* gn, const float* sgate,
                               const float* w5, const float* gamma, const float* beta, const float* mean, const float* rstd,
                               float* dgamma, float* dbeta, float* dw5, float* E, float* Fc, float* dw5part /* B * ECA_MAX_CHUNKS * 8 floats */, int B, int T, int C, hipStream_t s,
                               const PsaStats* ps = nullptr);
// plain BN backward finalize from per-sample S1,S2: dgamma, dbeta, E[c] = -dbeta/Mtot, Fc = dgamma/Mtot
int launch_bn_bwd_finalize(const float* S1, const float* S2, float* dgamma, float* dbeta, float* Ecol, float* Fc,
                           int B, int T, int C, hipStream_t s);
// dx = a[c] * (dy*sg[b,c] + E[b or 0,c] - xhat*Fc[c]) ; sg optional ; E per-sample if e_per_sample
int launch_bn_bwd_apply(int dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* a,
                        const float* sg, const float* E, int e_per_sample, const float* Fc, void* dx,
                        int B, int T, int C, hipStream_t s);
// Squeeze-Excite MLP on [B,C]: z = gap/T ; h = swish(z W1 + b1) ; se = sigmoid(h W2 + b2)
int launch_se_fwd(const float* gap, float invT, const float* W1, const float* b1, const float* W2, const float* b2,
                  float* hid_pre, float* se, int B, int C, int R, hipStream_t s);
// given dse[b,c] (= sum_t dOut*u3): grads of W1,b1,W2,b2 (atomic) and dgapT[b,c] = dL/dgap * (1/T)
int launch_se_bwd(const float* dse, const float* gap, float invT, const float* W1, const float* W2,
                  const float* hid_pre, const float* se, float* dW1, float* db1, float* dW2, float* db2,
                  float* dgapT, float* scr /* B*(C+2R) floats */, int B, int C, int R, hipStream_t s);

// ---- attention (attention.hip) -----------------------------------------------------
// q,k [B,H,T,dh], vt [B,H,dh,T]; o [B*T, H*dh]; lse [B,H,T]
// maskbits: attn_mask_words(B, H, T) dwords where the MFMA forward kernel stores the dropout keep flags for the backward
// kernels (nullptr: the backward kernels hash again; the lane-split kernels always hash)
size_t attn_mask_words(int B, int H, int T);
int launch_attn_fwd(int dt, const void* q, const void* k, const void* vt, void* o, float* lse,
                    int B, int H, int T, int dh, float scale, DropSpec drop, int impl, uint32_t* maskbits, hipStream_t s);
// dqkv [B*T, 3*H*dh] packed like the qkv projection output (head_major flag as in EpiArgs)
int launch_attn_bwd(int dt, const void* q, const void* k, const void* vt, const void* o, const void* dout,
                    const float* lse, float* delta, void* dqkv, int B, int H, int T, int dh, float scale,
                    DropSpec drop, int head_major, int impl, uint32_t* maskbits, hipStream_t s);

// ---- CTC / decode (ctc.hip) ----------------------------------------------------------
size_t ctc_workspace_floats(int B, int T, int L);
// logits [B,T,C] f32; labels [B,L] int64 (padded with blank); nll [B]; dlogits = grad_scale * d nll_b / d logits
int launch_ctc(const float* logits, const int64_t* labels, int B, int T, int C, int L, int blank,
               float* nll, float* dlogits, float grad_scale, float* ws, hipStream_t s, void* dlb = nullptr);
int launch_fill_u32(void* p, size_t n_words, uint32_t v, hipStream_t s);   // model.hip
int launch_mean(const float* v, float* out, int n, float scale, hipStream_t s);
int launch_greedy_decode(const float* logits, int B, int T, int C, int blank, int* out_idx, int* out_len, hipStream_t s);

// ---- inference preprocessing (preprocess.hip): raw [max_frames,276] (+ device clip length) -> [T,276]; mean/std [276] in OUTPUT order
int launch_preprocess(const float* raw, const int* n_frames, int max_frames, const float* mean, const float* stdv, float* out, int T, hipStream_t s);

// ---- optimizer (optimizer.hip) -----------------------------------------------------------
struct RAdamArgs { float lr, wd, beta1, beta2, eps, c1, c2, r_t; int rect; int sync; float slow_step; };
int launch_radam_lookahead(float* theta, const float* grad, float* m, float* v, float* slow, int64_t n,
                           RAdamArgs a, hipStream_t s);
int launch_scale(float* x, int64_t n, float scale, hipStream_t s);
