// Shared host-side types of libishara_hip.so: the model handle, its parameter / workspace bookkeeping, the profiled-launch
// macros and the GEMM wrappers.  Included by model.hip (the Keras get_model family) and conformer_r5.hip (the torch
// ConformerEncoder family).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <map>
#include "kernels.h"
#include "../../include/ishara_hip.h"

#define CK(expr) do { int _r = (expr); if (_r != 0) return _r; } while (0)
// profiled launch: key = kernel family, by = algorithmic bytes, fl = flops of this launch
#define CKP(m, key, by, fl, expr)                                                          \
    do {                                                                                   \
        ProfRec* _pr = nullptr;                                                            \
        if ((m)->prof.on) {                                                                \
            (m)->prof.recs.push_back(ProfRec{key, (m)->prof.get(), (m)->prof.get(), (double)(by), (double)(fl)}); \
            _pr = &(m)->prof.recs.back();                                                  \
            (void)hipEventRecord(_pr->e0, (m)->s);                                         \
        }                                                                                  \
        int _r = (expr);                                                                   \
        if (_pr) (void)hipEventRecord(_pr->e1, (m)->s);                                    \
        if (_r != 0) return _r;                                                            \
    } while (0)

static inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------ model description
struct ParamEntry { std::string name; int ndim; int64_t shape[2]; int64_t offset; bool trainable; };

struct DenseW {           // a Dense / 1x1-conv weight [K,N] (+bias) with its MFMA shadows
    int w = -1, b = -1, K = 0, N = 0;
    size_t wt = 0, wn = 0;    // byte offsets in the workspace
    int ldt = 0, ldn = 0;
};
struct Norm { int gamma = -1, beta = -1; };
struct BNp { int gamma = -1, beta = -1, mm = -1, mv = -1; };

struct Buf { size_t off = 0; };   // byte offset in the workspace

struct ConvBlock {
    DenseW W1, W2; int dw = -1, eca = -1; BNp bn; int k = 0; uint32_t site = 0;
    Buf z1, h2, h4, out, ssum, ssq, mean, rstd, a, bsh, gn, sg, P, Q, rs;
    bool folded = false;      // last training forward folded the drop-path scale into h4 (= rs[b] * (h2 P + Q)): see conv_fwd
    bool psa = false;         // last training forward did not write h4: the project conv's weight gradient applies the per-sample affine itself
                              // and emits the BatchNorm / ECA backward statistics (gemm.hip, TnPsa)
};
struct FFN {
    Norm ln; float eps; DenseW Wa, Wb; uint32_t site_in = 0, site_out = 0; bool has_out_drop = false;
    Buf xn, mean, rstd, za, u, out;
};
struct MHSA {
    Norm ln; float eps; DenseW Wqkv, Wp; float rate = 0.f; uint32_t site_attn = 0, site_out = 0; bool has_out_drop = false;
    Buf xn, mean, rstd, q, k, vt, o, lse, out, maskw;
};
struct SqzConv {
    Norm ln; DenseW Wc1, Wc3; int dw = -1, seW1 = -1, seb1 = -1, seW2 = -1, seb2 = -1; int k = 0, R = 0;
    Buf xn, mean, rstd, zc, zd, hd, u3, gap, hid, se, out;
};
struct ConfConv {
    DenseW Wp1, Wp2; int dw = -1, dwb = -1; BNp bn; Norm ln; int k = 0;
    // Keras defaults (c5:249-309); the torch family overrides them (nn.BatchNorm1d: eps 1e-5, momentum 0.1 on the NEW value, unbiased
    // running variance; nn.LayerNorm eps 1e-5)
    float bn_eps = 1e-3f, bn_keep = 0.99f, ln_eps = 1e-3f; int bn_unbiased = 0;
    // squeezeformer/convolution.py:226-238: Swish between the BatchNorm and the second pointwise conv (sw = swish(bnv)), Dropout on its output
    int swish_after_bn = 0; uint32_t site_out = 0; int has_out_drop = 0; Buf sw;
    Buf g, v, bnv, ssum, ssq, mean, rstd, a, bsh, r, lnmean, lnrstd, out;
};
// ---- torch ConformerEncoder family (conformer/conformer.py:6-87): post-LN sub-modules
struct R5FFN { DenseW W1, W2; Norm ln; uint32_t site_in = 0, site_out = 0; Buf za, u, r, mean, rstd, out;
               float factor = 1.f; };      // r = x + factor * ffn(x)   (squeezeformer half_step_residual: 0.5)
struct R5MHSA { DenseW Wqkv, Wp; Norm ln; uint32_t site_attn = 0; Buf q, k, vt, o, lse, maskw, r, mean, rstd, out; };
struct R5Block { R5FFN ffn1; R5MHSA mha; ConfConv conv; R5FFN ffn2; Norm ln; Buf mean, rstd, out; };
struct Layer {            // one entry of the sequential graph
    enum Kind { CONV, SQZ, CONF } kind;
    int idx;
};
struct SqzBlock { FFN ffn1; MHSA mha; SqzConv conv; FFN ffn2; };
struct ConfBlock { FFN ffn1; MHSA mha; ConfConv conv; FFN ffn2; };

// HIP-event profiler: when enabled every kernel launch site records a (start, stop) event pair
// on the launch stream plus the algorithmic bytes / flops of that launch (bench.py roofline).
struct ProfRec { const char* key; hipEvent_t e0, e1; double bytes, flops; };
struct Profiler {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
};

struct ishara_model {
    ishara_config cfg;
    Profiler prof;
    int dt;                       // activation / MFMA dtype
    int d, T, F, C, H, dh, dtop, Bmax, L;
    std::vector<ParamEntry> entries;
    int64_t n_total = 0, n_train = 0;
    // graph
    DenseW stemW; BNp stem_bn;
    // gradient buckets for an overlapped all-reduce: ranges of the flat gradient, in the order the backward pass completes them
    std::vector<int64_t> bucket_lo, bucket_hi; std::vector<int> bucket_after_layer; std::vector<hipEvent_t> bucket_ev;
    std::vector<size_t> layer_entry_end; size_t stem_entry_end = 0;
    int cls_pad = 0; Buf dlb;          // bf16 model: dlogits also as bf16 [M, cls_pad] (zero padded), 0 = f32 operand path
    int stem_kp = 0; Buf stem_xb;      // bf16 model: input rows packed to bf16 [M, stem_kp] (zero padded), 0 = f32-A GEMM path
    Buf stem_h0, stem_out, stem_ssum, stem_ssq, stem_mean, stem_rstd, stem_a, stem_bsh, pe;
    std::vector<ConvBlock> convs;
    std::vector<SqzBlock> sqz;
    std::vector<ConfBlock> conf;
    std::vector<Layer> layers;
    int family = 0;                    // 0: Keras get_model hybrid; 1: torch ConformerEncoder (conformer_r5.hip)
    std::vector<R5Block> r5; Buf r5_x, t4, fac;          // fac: [Bmax] floats, all = the FFN residual factor (epilogue row scale)
    struct R4State* r4 = nullptr;                          // torch Squeezeformer family (squeezeformer_r4.hip)
    DenseW topW, clsW; uint32_t head_site = 0; Buf head_hh;
    uint32_t nsites = 0;
    std::vector<DenseW*> denses;
    // temps
    RedSink red; Buf red_arena; size_t red_cap = 0, red_off = 0; bool red_on = false;      // deferred column sums of LayerNorm / depthwise-conv parameter gradients (model.hip red_scratch)
    TnDefer tn_defer; Buf slab2[2]; bool tn_defer_on = false;      // deferred wgrad slab sums (gemm.hip): two alternating slab buffers
    Buf gA, gB, t1, t2, t3, S1, S2, E, Fc, Ecol, ecap, dse, dgapT, slab, ctcws, dlogits, nllb, delta;
    Buf psaG, psaR; bool psa_on = false;   // TnPsa outputs: G [B, d], Rpart [B][d / 64][2d]
    size_t shadow_begin = 0, shadow_end = 0;
    size_t shadow_tab_off = 0;             // device descriptor table of the batched shadow build, inside the workspace (no
                                           // hipMalloc/hipFree of our own: a hipFree from a garbage-collected model would break a
                                           // stream capture in progress elsewhere in the process)
    std::vector<ShadowDesc> shadow_tab_host;
    int shadow_ntab = 0, shadow_tiles = 0;
    bool shadow_ready = false;
    size_t ws_need = 0;
    // bound
    float* params = nullptr; float* grads = nullptr; float* om = nullptr; float* ov = nullptr; float* oslow = nullptr;
    char* ws = nullptr; int64_t ws_bytes = 0;
    std::vector<float> pe_host;
    // run state
    int lastB = 0; int last_training = 0; uint32_t last_seed = 0; const float* last_x = nullptr;
    int opt_iter = 0;
    hipStream_t s = nullptr;

    // ---- build helpers
    int addp(const std::string& name, int64_t r, int64_t c, bool trainable) {
        ParamEntry e; e.name = name; e.ndim = c > 0 ? 2 : 1; e.shape[0] = r; e.shape[1] = c > 0 ? c : 0; e.offset = -1; e.trainable = trainable;
        entries.push_back(e);
        return (int)entries.size() - 1;
    }
    size_t cur = 0;
    // ISHARA_WS_GUARD=1 (read at ishara_create): every workspace buffer is followed by a 256-byte guard zone that ishara_bind fills
    // with a pattern and ishara_workspace_guard_check verifies — an out-of-bounds write of any kernel into a neighbouring buffer
    // shows up as a named offset instead of as silent corruption (tests/test_tflite_gpu.py, tests/test_model_gpu.py)
    bool guard = false; std::vector<size_t> guard_offs; std::vector<std::pair<size_t, size_t>> allocs;
    Buf alloc(size_t bytes) {
        Buf b; b.off = cur; cur = rup(cur + bytes, 256);
        allocs.push_back({b.off, bytes});
        if (guard) { guard_offs.push_back(cur); cur += 256; }
        return b;
    }
    Buf act(int cols) { return alloc((size_t)Bmax * T * cols * dt_size(dt)); }
    Buf f32(size_t n) { return alloc(n * sizeof(float)); }
    DenseW dense(const std::string& name, int K, int N, bool bias) {
        DenseW w; w.K = K; w.N = N;
        w.w = addp(name + "/kernel", K, N, true);
        if (bias) w.b = addp(name + "/bias", N, 0, true);
        return w;
    }
    Norm norm(const std::string& name, int c) { Norm n; n.gamma = addp(name + "/gamma", c, 0, true); n.beta = addp(name + "/beta", c, 0, true); return n; }
    BNp bnp(const std::string& name, int c) {
        BNp b; b.gamma = addp(name + "/gamma", c, 0, true); b.beta = addp(name + "/beta", c, 0, true);
        b.mm = addp(name + "/moving_mean", c, 0, false); b.mv = addp(name + "/moving_variance", c, 0, false);
        return b;
    }
    float* P(int idx) const { return params + entries[idx].offset; }
    float* G(int idx) const { return grads + entries[idx].offset; }
    template <typename TT = void> TT* W(Buf b) const { return reinterpret_cast<TT*>(ws + b.off); }
    float* Wf(Buf b) const { return reinterpret_cast<float*>(ws + b.off); }
};


// ------------------------------------------------------------------ shared helpers (model.hip)
struct Run { int B, M, training; uint32_t seed; };
static inline DropSpec dspec(const Run& r, uint32_t site, float rate) { return make_drop(r.seed, site, rate, r.training != 0); }
static inline DropSpec dspec_attn(const Run& r, uint32_t site, float rate) { return make_drop_attn(r.seed, site, rate, r.training != 0); }   // attention probabilities: common.h rng_quad
void plan_shadow(ishara_model* m, DenseW& w, int min_ldt = 0, int min_ldn = 0);
// profiled GEMM launches over a planned Dense weight: forward C = epi(A W), dgrad dX = epi(dY W^T), wgrad dW += A^T dY (+ bias grad)
int gemm_fwd(ishara_model* m, const DenseW& w, const void* A, int dtA, void* Cc, int dtC, int M, int aop, const OpArgs& oa, EpiArgs ea);
int gemm_dgrad(ishara_model* m, const DenseW& w, const void* dY, int dtA, void* dX, int M, int aop, const OpArgs& oa, const EpiArgs& ea);
int gemm_wgrad(ishara_model* m, const DenseW& w, const void* A, int dtA, int aop, const OpArgs& oa, const void* dY, int dtB, int bop, const OpArgs& ob, int M, int ka_valid = 0, int nb_valid = 0,
               const float* bias_rowscale = nullptr, int bias_T = 0, const TnPsa* psa = nullptr);
int wgrad_flush(ishara_model* m);
int confconv_fwd(ishara_model* m, ConfConv& c, const Run& r, const void* x);
int confconv_bwd(ishara_model* m, ConfConv& c, const Run& r, const void* x, const void* g, void* gn);
int r5_ffn_fwd(ishara_model* m, R5FFN& f, const Run& r, const void* x);
int r5_ffn_bwd(ishara_model* m, R5FFN& f, const Run& r, const void* x, const void* g, void* gn);
int r5_ln_fwd(ishara_model* m, const Run& r, const void* x, const Norm& n, void* y, Buf mean, Buf rstd);
int r5_ln_bwd(ishara_model* m, const Run& r, const void* g, const void* x, const Norm& n, Buf mean, Buf rstd, void* dx);
int r5_from_f32(int dt, const float* x, void* y, size_t n, hipStream_t s);
int r5_to_f32(int dt, const void* x, float* y, size_t n, hipStream_t s);
// torch ConformerEncoder family (conformer_r5.hip)
int r5_validate(const ishara_config& c);
void r5_build_graph(ishara_model* m);
void r5_plan_workspace(ishara_model* m);
// torch Squeezeformer family (squeezeformer_r4.hip)
int r4_validate(const ishara_config& c);
void r4_build_graph(ishara_model* m);
void r4_plan_workspace(ishara_model* m);
int r4_bind(ishara_model* m);
void r4_destroy(ishara_model* m);
int r4_output_frames(const ishara_model* m);
int r4_forward(ishara_model* m, const float* x, int32_t B, float* y, int32_t training, uint32_t seed, hipStream_t st);
int r4_backward(ishara_model* m, const float* dy, int32_t B, float* dx, hipStream_t st);
