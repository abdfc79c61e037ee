// Host side of libishara_hip.so: the layer graph get_model(...) builds
// (conv-hybrid-model.ipynb c7:12-65), its flat parameter layout, the workspace plan and the
// forward / backward / optimizer orchestration over the kernels in gemm.hip, elementwise.hip,
// attention.hip, ctc.hip and optimizer.hip.  Everything is launched on the caller's stream.
#include "model_types.h"
extern int g_force_regstage;
#include <algorithm>
#include <stdlib.h>

// ------------------------------------------------------------------ error string
static thread_local char g_err[1024] = "";
void ishara_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ishara_last_error(void) { return g_err; }

// ------------------------------------------------------------------ small kernels
// fills inside forward / backward use a kernel rather than hipMemsetAsync, so that a captured forward / training step holds kernel
// nodes only.  (Round 1 blamed hipGraph memset nodes for intermittently wrong replays; round 2 could not reproduce that — 0 of 72 000
// checks wrong with memset nodes, profiles/r2_graph_memset_experiment.txt — and withdrew the attribution: DESIGN.md §4.)
__global__ void fill_u32_kernel(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill_u32(void* p, size_t n_words, uint32_t v, hipStream_t s) {
    if (n_words == 0) return 0;
    const int grid = (int)((n_words + 255) / 256 < 1024 ? (n_words + 255) / 256 : 1024);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(grid), dim3(256), 0, s, (uint32_t*)p, n_words, v);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
__global__ void droppath_kernel(float* rs, int B, DropSpec d) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) rs[b] = (d.thr == 0u || rng_keep(rng_row_key(d.key, (uint32_t)b), 0u, d.thr)) ? d.scale : 0.f;
}
__global__ void dropout_mask_kernel(float* out, int rows, int cols, DropSpec d) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / cols), c = (uint32_t)(i % cols);
        out[i] = (d.thr == 0u || rng_keep(rng_row_key(d.key, r), c, d.thr)) ? d.scale : 0.f;
    }
}
// packed qkv [M,3d] (head-major) -> q,k [B,H,T,dh], vt [B,H,dh,T]   (operator tests only)
template <typename T>
__global__ void qkv_split_kernel(const T* qkv, T* q, T* k, T* vt, int B, int H, int Tn, int dh) {
    const int d = H * dh;
    const size_t n = (size_t)B * Tn * 3 * d;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / (3 * d);
        const int col = (int)(i - m * 3 * d);
        const int h = col / (3 * dh), w = col - h * 3 * dh, part = w / dh, e = w - part * dh;
        const int b = (int)(m / Tn), t = (int)(m - (size_t)b * Tn);
        const T v = qkv[i];
        if (part == 0) q[((size_t)(b * H + h) * Tn + t) * dh + e] = v;
        else if (part == 1) k[((size_t)(b * H + h) * Tn + t) * dh + e] = v;
        else vt[((size_t)(b * H + h) * dh + e) * Tn + t] = v;
    }
}

// ------------------------------------------------------------------ construction
static void build_conv(ishara_model* m, const std::string& name, int k) {
    ConvBlock cb;
    const int d = m->d, c = 2 * d;
    cb.k = k;
    cb.W1 = m->dense(name + "_expand_conv", d, c, true);
    cb.dw = m->addp(name + "_dwconv/depthwise_kernel", k, c, true);
    cb.bn = m->bnp(name + "_bn", c);
    cb.eca = m->addp(name + "_eca/kernel", 5, 0, true);
    cb.W2 = m->dense(name + "_project_conv", c, d, true);
    cb.site = m->nsites++;
    m->convs.push_back(cb);
    m->layers.push_back({Layer::CONV, (int)m->convs.size() - 1});
    m->layer_entry_end.push_back(m->entries.size());
}
static FFN build_ffn(ishara_model* m, Norm ln, float eps, const std::string& n1, const std::string& n2, int e, bool out_drop) {
    FFN f; f.ln = ln; f.eps = eps;
    f.Wa = m->dense(n1, m->d, m->d * e, true);
    f.Wb = m->dense(n2, m->d * e, m->d, true);
    f.site_in = m->nsites++;
    f.has_out_drop = out_drop;
    if (out_drop) f.site_out = m->nsites++;
    return f;
}
static MHSA build_mhsa(ishara_model* m, Norm ln, float eps, const std::string& name, float rate, bool out_drop) {
    MHSA a; a.ln = ln; a.eps = eps; a.rate = rate;
    a.Wqkv = m->dense(name + "/qkv", m->d, 3 * m->d, false);
    a.Wp = m->dense(name + "/proj", m->d, m->d, false);
    a.site_attn = m->nsites++;
    a.has_out_drop = out_drop;
    if (out_drop) a.site_out = m->nsites++;
    return a;
}

static void build_graph(ishara_model* m) {
    const ishara_config& c = m->cfg;
    const int d = m->d;
    m->stemW = m->dense("stem_conv", m->F, d, false);
    m->stem_bn = m->bnp("stem_bn", d);
    m->stem_entry_end = m->entries.size();
    auto conv_blocks = [&](const std::string& tag) {
        for (int j = 0; j < c.num_conv_per_block; ++j) {
            const int k = c.kernel_sizes[j % c.num_kernel_sizes];
            build_conv(m, "conv" + tag + "_" + std::to_string(j + 1), k);
        }
    };
    const int esq = c.squeeze_expansion > 0 ? c.squeeze_expansion : c.expansion_factor;
    const int ecf = c.conformer_expansion > 0 ? c.conformer_expansion : c.expansion_factor;
    const int tk = c.transformer_kernel_size;
    for (int i = 0; i < c.num_conv_squeeze_blocks; ++i) {
        conv_blocks("squeeze_" + std::to_string(i));
        const std::string n = "squeezeformer_" + std::to_string(i);
        SqzBlock sb;
        // parameter order = oracle/ishara_oracle.py::_squeezeformer_specs
        Norm n1 = m->norm(n + "/norm1", d);
        sb.ffn1 = build_ffn(m, n1, 1e-6f, n + "/ffn1_dense1", n + "/ffn1_dense2", esq, true);
        Norm n2 = m->norm(n + "/norm2", d);
        sb.mha = build_mhsa(m, n2, 1e-6f, n + "/mha", c.dropout_rate, true);
        sb.conv.ln = m->norm(n + "/conv/norm", d);
        sb.conv.k = tk;
        sb.conv.Wc1 = m->dense(n + "/conv/conv1", d, d * esq, true);
        sb.conv.dw = m->addp(n + "/conv/conv2/depthwise_kernel", tk, d * esq, true);
        sb.conv.Wc3 = m->dense(n + "/conv/conv3", d * esq, d, true);
        sb.conv.R = d / 8 > 1 ? d / 8 : 1;
        sb.conv.seW1 = m->addp(n + "/conv/se/fc1/kernel", d, sb.conv.R, true);
        sb.conv.seb1 = m->addp(n + "/conv/se/fc1/bias", sb.conv.R, 0, true);
        sb.conv.seW2 = m->addp(n + "/conv/se/fc2/kernel", sb.conv.R, d, true);
        sb.conv.seb2 = m->addp(n + "/conv/se/fc2/bias", d, 0, true);
        Norm n3 = m->norm(n + "/norm3", d);
        sb.ffn2 = build_ffn(m, n3, 1e-6f, n + "/ffn2_dense1", n + "/ffn2_dense2", esq, true);
        m->sqz.push_back(sb);
        m->layers.push_back({Layer::SQZ, (int)m->sqz.size() - 1});
        m->layer_entry_end.push_back(m->entries.size());
    }
    for (int i = 0; i < c.num_conv_conform_blocks; ++i) {
        conv_blocks("conform_" + std::to_string(i));
        const std::string n = "conformer_" + std::to_string(i);
        ConfBlock cb;
        // order = _conformer_specs: ffn1, mha, conv (pw1, dw, pw2, bn, ln), ffn2, layer_norm1, layer_norm2
        Norm dummy;
        cb.ffn1 = build_ffn(m, dummy, 1e-6f, n + "/ffn1/dense1", n + "/ffn1/dense2", ecf, false);
        cb.mha = build_mhsa(m, dummy, 1e-6f, n + "/mha", c.conformer_attn_dropout, false);
        cb.conv.k = tk;
        cb.conv.Wp1 = m->dense(n + "/conv/pointwise_conv1", d, 2 * d, true);
        cb.conv.dw = m->addp(n + "/conv/depthwise_conv/kernel", tk, d, true);
        cb.conv.dwb = m->addp(n + "/conv/depthwise_conv/bias", d, 0, true);
        cb.conv.Wp2 = m->dense(n + "/conv/pointwise_conv2", d, d, true);
        cb.conv.bn = m->bnp(n + "/conv/batch_norm", d);
        cb.conv.ln = m->norm(n + "/conv/layer_norm", d);
        cb.ffn2 = build_ffn(m, dummy, 1e-6f, n + "/ffn2/dense1", n + "/ffn2/dense2", ecf, false);
        Norm l1 = m->norm(n + "/layer_norm1", d);
        Norm l2 = m->norm(n + "/layer_norm2", d);
        cb.ffn1.ln = l1; cb.mha.ln = l1;      // layer_norm1 is applied twice (c5:324,330)
        cb.ffn2.ln = l2;
        // dropout sites were numbered in build order ffn1, mha, ffn2 == forward order
        m->conf.push_back(cb);
        m->layers.push_back({Layer::CONF, (int)m->conf.size() - 1});
        m->layer_entry_end.push_back(m->entries.size());
    }
    m->topW = m->dense("top_conv", d, m->dtop, true);
    m->clsW = m->dense("classifier", m->dtop, m->C, true);
    m->head_site = m->nsites++;

    // physical offsets: trainable first, then BatchNorm moving statistics
    int64_t off = 0;
    for (auto& e : m->entries) if (e.trainable) { e.offset = off; off += e.shape[0] * (e.ndim == 2 ? e.shape[1] : 1); }
    m->n_train = off;
    for (auto& e : m->entries) if (!e.trainable) { e.offset = off; off += e.shape[0] * (e.ndim == 2 ? e.shape[1] : 1); }
    m->n_total = off;

    // ---- gradient buckets.  Trainable parameters sit in creation order (stem, layers, head) and the backward pass runs head ->
    // layers in reverse -> stem, so the gradient of everything above a layer boundary is final once that layer's backward is
    // enqueued: up to 4 ranges of about equal size, cut at layer boundaries, each with an event recorded on the compute stream.
    {
        const int nl = (int)m->layers.size();
        auto first_off = [&](size_t e0) -> int64_t {           // offset of the first trainable entry at index >= e0
            for (size_t i = e0; i < m->entries.size(); ++i) if (m->entries[i].trainable) return m->entries[i].offset;
            return m->n_train;
        };
        std::vector<int64_t> lo(nl);                            // first gradient element of layer li
        for (int li = 0; li < nl; ++li) lo[li] = first_off(li == 0 ? m->stem_entry_end : m->layer_entry_end[li - 1]);
        const int64_t target = m->n_train / 4 + 1;
        int64_t hi = m->n_train;
        m->bucket_after_layer.assign(nl, -1);
        for (int li = nl - 1; li >= 1 && (int)m->bucket_lo.size() < 3; --li) {
            if (hi - lo[li] >= target && lo[li] > 0) {          // cut below layer li: bucket = [lo[li], hi)
                m->bucket_after_layer[li] = (int)m->bucket_lo.size();
                m->bucket_lo.push_back(lo[li]); m->bucket_hi.push_back(hi);
                hi = lo[li];
            }
        }
        m->bucket_lo.push_back(0); m->bucket_hi.push_back(hi);  // the rest (stem included): complete at the end of the backward pass
    }
}

void plan_shadow(ishara_model* m, DenseW& w, int min_ldt, int min_ldn) {
    const int bk = dt_is16(m->dt) ? 64 : 32;
    const size_t es = dt_size(m->dt);
    w.ldt = (int)rup(w.K, bk);
    if (w.ldt < min_ldt) w.ldt = min_ldt;      // zero-padded K (the shadow arena is zero-filled, the builder writes k < K)
    w.wt = m->alloc(rup(w.N, 128) * (size_t)w.ldt * es).off;
    w.ldn = (int)rup(w.N, bk);
    if (w.ldn < min_ldn) w.ldn = min_ldn;
    w.wn = m->alloc(rup(w.K, 128) * (size_t)w.ldn * es).off;
    m->denses.push_back(&w);
}

static void plan_workspace(ishara_model* m) {
    const int d = m->d, B = m->Bmax, T = m->T;
    const size_t Mx = (size_t)B * T;
    m->cur = 0;
    // ---- shadows first (one contiguous arena that sync_weights zero-fills)
    m->shadow_begin = m->cur;
    m->stem_kp = (dt_is16(m->dt) && m->F <= 512) ? (m->F <= 256 ? 256 : 512) : 0;      // fp16 (inference) too: the stem Dense on the A-stationary kernel
    plan_shadow(m, m->stemW, m->stem_kp);
    for (auto& cb : m->convs) { plan_shadow(m, cb.W1); plan_shadow(m, cb.W2); }
    // FFN/MHSA shadows are planned with their activations below; keep the arena contiguous by
    // planning all shadows before any activation:
    std::vector<DenseW*> later;
    for (auto& sb : m->sqz) { later.insert(later.end(), {&sb.ffn1.Wa, &sb.ffn1.Wb, &sb.mha.Wqkv, &sb.mha.Wp, &sb.conv.Wc1, &sb.conv.Wc3, &sb.ffn2.Wa, &sb.ffn2.Wb}); }
    for (auto& cb : m->conf) { later.insert(later.end(), {&cb.ffn1.Wa, &cb.ffn1.Wb, &cb.mha.Wqkv, &cb.mha.Wp, &cb.conv.Wp1, &cb.conv.Wp2, &cb.ffn2.Wa, &cb.ffn2.Wb}); }
    later.push_back(&m->topW); later.push_back(&m->clsW);
    // classifier: its dY operand is the zero-padded bf16 [M, 128] copy of dlogits (bf16 model, <= 64 classes)
    m->cls_pad = (m->dt == DT_BF16 && m->C <= 64 && m->C % 4 == 0) ? 128 : 0;
    for (DenseW* w : later) plan_shadow(m, *w, 0, w == &m->clsW ? m->cls_pad : 0);
    m->shadow_end = m->cur;
    m->shadow_tab_off = m->alloc(m->denses.size() * sizeof(ShadowDesc)).off;
    // ---- stem
    m->pe = m->f32((size_t)T * d);
    m->stem_h0 = m->act(d); m->stem_out = m->act(d);
    if (m->stem_kp) m->stem_xb = m->alloc(Mx * (size_t)m->stem_kp * 2);
    m->stem_ssum = m->f32((size_t)B * d); m->stem_ssq = m->f32((size_t)B * d);
    m->stem_mean = m->f32(d); m->stem_rstd = m->f32(d); m->stem_a = m->f32(d); m->stem_bsh = m->f32(d);
    for (auto& cb : m->convs) {
        const int c = 2 * d;
        cb.z1 = m->act(c); cb.h2 = m->act(c); cb.h4 = m->act(c); cb.out = m->act(d);
        cb.ssum = m->f32((size_t)B * c); cb.ssq = m->f32((size_t)B * c);
        cb.mean = m->f32(c); cb.rstd = m->f32(c); cb.a = m->f32(c); cb.bsh = m->f32(c);
        cb.gn = m->f32((size_t)B * c); cb.sg = m->f32((size_t)B * c); cb.P = m->f32((size_t)B * c); cb.Q = m->f32((size_t)B * c);
        cb.rs = m->f32(B);
    }
    auto plan_ffn_act = [&](FFN& f) {
        f.xn = m->act(d); f.mean = m->f32(Mx); f.rstd = m->f32(Mx);
        f.za = m->act(f.Wa.N); f.u = m->act(f.Wa.N); f.out = m->act(d);
    };
    auto plan_mhsa_act = [&](MHSA& a) {
        a.xn = m->act(d); a.mean = m->f32(Mx); a.rstd = m->f32(Mx);
        a.q = m->act(d); a.k = m->act(d); a.vt = m->act(d); a.o = m->act(d);
        a.lse = m->f32((size_t)B * m->H * T); a.out = m->act(d);
        a.maskw = m->f32(attn_mask_words(B, m->H, T));        // dropout keep bits of the attention probabilities (fwd -> bwd)
    };
    for (auto& sb : m->sqz) {
        plan_ffn_act(sb.ffn1); plan_mhsa_act(sb.mha);
        SqzConv& c = sb.conv;
        const int de = c.Wc1.N;
        c.xn = m->act(d); c.mean = m->f32(Mx); c.rstd = m->f32(Mx);
        c.zc = m->act(de); c.zd = m->act(de); c.hd = m->act(de); c.u3 = m->act(d);
        c.gap = m->f32((size_t)B * d); c.hid = m->f32((size_t)B * c.R); c.se = m->f32((size_t)B * d); c.out = m->act(d);
        plan_ffn_act(sb.ffn2);
    }
    for (auto& cb : m->conf) {
        plan_ffn_act(cb.ffn1); plan_mhsa_act(cb.mha);
        ConfConv& c = cb.conv;
        c.g = m->act(2 * d); c.v = m->act(d); c.bnv = m->act(d);
        c.ssum = m->f32((size_t)B * d); c.ssq = m->f32((size_t)B * d);
        c.mean = m->f32(d); c.rstd = m->f32(d); c.a = m->f32(d); c.bsh = m->f32(d);
        c.r = m->act(d); c.lnmean = m->f32(Mx); c.lnrstd = m->f32(Mx); c.out = m->act(d);
        plan_ffn_act(cb.ffn2);
    }
    m->head_hh = m->act(m->dtop);
    // ---- temporaries
    int maxw = 3 * d;
    if (m->dtop > maxw) maxw = m->dtop;
    for (auto& sb : m->sqz) if (sb.ffn1.Wa.N > maxw) maxw = sb.ffn1.Wa.N;
    for (auto& cb : m->conf) if (cb.ffn1.Wa.N > maxw) maxw = cb.ffn1.Wa.N;
    m->gA = m->act(d); m->gB = m->act(d);
    m->t1 = m->act(maxw); m->t2 = m->act(maxw); m->t3 = m->act(maxw);
    const int maxc = 2 * d > maxw ? 2 * d : maxw;
    m->S1 = m->f32((size_t)B * maxc); m->S2 = m->f32((size_t)B * maxc); m->E = m->f32((size_t)B * maxc);
    m->Fc = m->f32(maxc); m->Ecol = m->f32(maxc); m->ecap = m->f32((size_t)B * 8 * ECA_MAX_CHUNKS);
    m->dse = m->f32((size_t)B * d); m->dgapT = m->f32((size_t)B * d);
    m->psa_on = m->dt == DT_BF16 && !m->convs.empty() && getenv("ISHARA_NO_PSA") == nullptr;
    if (m->psa_on) { m->psaG = m->f32((size_t)B * d); m->psaR = m->f32((size_t)B * (size_t)((d + 63) / 64) * 2 * d); }
    size_t slabf = 0;
    for (DenseW* w : m->denses) { const size_t f = gemm_tn_slab_floats((int)Mx, w->K, w->N, m->dt); if (f > slabf) slabf = f; }
    if (m->stem_kp) { const size_t f = gemm_tn_slab_floats((int)Mx, m->stem_kp, d, m->dt); if (f > slabf) slabf = f; }
    if (layernorm_bwd_scratch_floats(d) > slabf) slabf = layernorm_bwd_scratch_floats(d);
    if (dwconv_bwd_scratch_floats(2 * maxw, 31) > slabf) slabf = dwconv_bwd_scratch_floats(2 * maxw, 31);
    if (dwconv_fwd_scratch_floats(B, T, 2 * maxw) > slabf) slabf = dwconv_fwd_scratch_floats(B, T, 2 * maxw);
    m->slab = m->f32(slabf);
    { size_t wf = 0; for (DenseW* w : m->denses) { const size_t f = gemm_tn_slab_floats((int)Mx, w->K, w->N, m->dt); if (f > wf) wf = f; }
      if (m->stem_kp) { const size_t f = gemm_tn_slab_floats((int)Mx, m->stem_kp, d, m->dt); if (f > wf) wf = f; }
      m->slab2[0] = m->f32(wf); m->slab2[1] = m->f32(wf); m->tn_defer_on = getenv("ISHARA_NO_DEFERRED_SLAB_SUMS") == nullptr; }
    {   // arena of the deferred parameter-gradient sums: every LayerNorm / depthwise-conv backward of one pass (flushed early when it runs full)
        size_t need = 0;
        const size_t lnf = (layernorm_bwd_scratch_floats(d) + 63) & ~(size_t)63, dwf = (dwconv_bwd_scratch_floats(2 * maxw, 31) + 63) & ~(size_t)63;
        need = (size_t)m->layers.size() * (5 * lnf + 2 * dwf);
        const size_t cap = (size_t)256 << 20;                       // floats: 1 GiB
        m->red_cap = need < cap ? need : cap;
        m->red_on = getenv("ISHARA_NO_DEFERRED_REDUCE") == nullptr && m->red_cap > 0;
        if (m->red_on) m->red_arena = m->f32(m->red_cap);
    }
    m->ctcws = m->f32(ctc_workspace_floats(B, T, m->L));
    m->dlogits = m->f32(Mx * m->C);
    if (m->cls_pad) m->dlb = m->alloc(Mx * (size_t)m->cls_pad * 2);
    m->nllb = m->f32(B);
    m->delta = m->f32((size_t)B * m->H * T);
    m->ws_need = m->cur;
}

extern "C" int ishara_create(const ishara_config* cfg, ishara_model** out) {
    if (!cfg || !out) { ishara_set_error("ishara_create: null argument"); return -1; }
    ishara_config c = *cfg;
    if (c.family != ISHARA_FAMILY_KERAS_HYBRID && c.family != ISHARA_FAMILY_TORCH_CONFORMER && c.family != ISHARA_FAMILY_TORCH_SQUEEZEFORMER) { ishara_set_error("family=%d unknown", c.family); return -1; }
    if (c.family == ISHARA_FAMILY_TORCH_SQUEEZEFORMER) {    // squeezeformer/encoder.py: its own front end (conv2d subsampling), no head, no CTC; any frame count
        CK(r4_validate(c));
        if (c.dim <= 0 || c.dim % 8 != 0 || c.dim > 512 || c.num_heads <= 0 || c.dim % c.num_heads != 0) { ishara_set_error("encoder_dim=%d / heads=%d unsupported", c.dim, c.num_heads); return -1; }
        if (c.transformer_kernel_size < 1 || c.transformer_kernel_size > 31 || c.transformer_kernel_size % 2 == 0) { ishara_set_error("conv_kernel_size must be odd, 1..31"); return -1; }
        if (c.max_batch <= 0 || (c.dtype != ISHARA_F32 && c.dtype != ISHARA_BF16)) { ishara_set_error("max_batch / dtype unsupported"); return -1; }
        ishara_model* m = new ishara_model();
        m->cfg = c; m->dt = c.dtype == ISHARA_BF16 ? DT_BF16 : DT_F32;
        m->d = c.dim; m->T = c.frames; m->F = c.features; m->C = 60; m->H = c.num_heads; m->dh = c.dim / c.num_heads;
        m->dtop = 2 * c.dim; m->Bmax = c.max_batch; m->L = 64;
        m->family = c.family;
        { const char* gv = getenv("ISHARA_WS_GUARD"); m->guard = gv && gv[0] == '1'; }
        r4_build_graph(m); r4_plan_workspace(m);
        *out = m;
        return 0;
    }
    if (c.family == ISHARA_FAMILY_TORCH_CONFORMER) {      // conformer/conformer.py: no stem, no head, no CTC — the encoder stack only
        CK(r5_validate(c));
        c.features = c.dim; c.num_conv_per_block = 0; c.num_conv_squeeze_blocks = 0;
        if (c.num_classes <= 0) c.num_classes = 60;
        if (c.num_kernel_sizes <= 0) { c.num_kernel_sizes = 1; c.kernel_sizes[0] = 3; }
    }
    if (c.dim <= 0 || c.dim % 8 != 0 || c.dim > 512) { ishara_set_error("dim=%d unsupported (multiple of 8, <=512)", c.dim); return -1; }
    if (c.num_heads <= 0 || c.dim % c.num_heads != 0) { ishara_set_error("dim %% num_heads != 0"); return -1; }
    const int dh = c.dim / c.num_heads;
    if (dh != 8 && dh != 16 && dh != 24 && dh != 32 && dh != 48 && dh != 64) { ishara_set_error("head dim %d unsupported (8,16,24,32,48,64)", dh); return -1; }
    if (c.frames <= 0 || c.frames % 8 != 0 || c.frames > 512) { ishara_set_error("frames=%d unsupported (multiple of 8, <=512)", c.frames); return -1; }
    if (c.features <= 0 || c.features % 4 != 0) { ishara_set_error("features=%d unsupported (positive multiple of 4: 16-byte input rows)", c.features); return -1; }
    if (c.num_classes < 2 || c.num_classes > 64) { ishara_set_error("num_classes=%d unsupported (2..64)", c.num_classes); return -1; }
    if (c.num_kernel_sizes <= 0 && c.num_conv_per_block > 0) { ishara_set_error("kernel_sizes is empty"); return -1; }
    if (c.num_kernel_sizes > 8) { ishara_set_error("at most 8 kernel sizes"); return -1; }
    for (int i = 0; i < c.num_kernel_sizes; ++i) if (c.kernel_sizes[i] < 1 || c.kernel_sizes[i] > 31) { ishara_set_error("kernel size %d unsupported (1..31)", c.kernel_sizes[i]); return -1; }
    if (c.transformer_kernel_size < 1 || c.transformer_kernel_size > 31 || c.transformer_kernel_size % 2 == 0) { ishara_set_error("transformer_kernel_size must be odd, 1..31"); return -1; }
    if (c.max_batch <= 0) { ishara_set_error("max_batch must be > 0"); return -1; }
    if (c.dtype != ISHARA_F32 && c.dtype != ISHARA_BF16 && c.dtype != ISHARA_F16) { ishara_set_error("dtype must be ISHARA_F32, ISHARA_BF16 or ISHARA_F16"); return -1; }
    if (c.dtype == ISHARA_F16 && c.family != ISHARA_FAMILY_KERAS_HYBRID) { ishara_set_error("ISHARA_F16 (inference-only storage) exists for the Keras hybrid family only"); return -1; }
    if (c.top_dim <= 0) c.top_dim = 2 * c.dim;
    if (c.top_dim % 8 != 0) { ishara_set_error("top_dim must be a multiple of 8"); return -1; }
    if (c.max_label_len <= 0) c.max_label_len = 64;
    if (c.max_label_len > 255) { ishara_set_error("max_label_len > 255"); return -1; }
    ishara_model* m = new ishara_model();
    m->cfg = c; m->dt = c.dtype == ISHARA_BF16 ? DT_BF16 : (c.dtype == ISHARA_F16 ? DT_F16 : DT_F32);
    m->d = c.dim; m->T = c.frames; m->F = c.features; m->C = c.num_classes; m->H = c.num_heads; m->dh = dh;
    m->dtop = c.top_dim; m->Bmax = c.max_batch; m->L = c.max_label_len;
    m->family = c.family;
    { const char* gv = getenv("ISHARA_WS_GUARD"); m->guard = gv && gv[0] == '1'; }
    if (m->family == ISHARA_FAMILY_TORCH_CONFORMER) { r5_build_graph(m); r5_plan_workspace(m); *out = m; return 0; }
    build_graph(m);
    plan_workspace(m);
    // positional encoding table (c5:226-235): [sin | cos] halves, fp32 arithmetic
    m->pe_host.resize((size_t)m->T * m->d);
    const int half = m->d / 2;
    for (int t = 0; t < m->T; ++t)
        for (int i = 0; i < half; ++i) {
            const float depth = (float)i / (float)half;
            const float rate = 1.0f / powf(10000.0f, depth);
            const float ang = (float)t * rate;
            m->pe_host[(size_t)t * m->d + i] = sinf(ang);
            m->pe_host[(size_t)t * m->d + half + i] = cosf(ang);
        }
    *out = m;
    return 0;
}
extern "C" void ishara_destroy(ishara_model* m) {
    if (m) for (auto e : m->bucket_ev) (void)hipEventDestroy(e);
    if (m && m->r4) r4_destroy(m);
    delete m;
}
extern "C" int64_t ishara_param_total(const ishara_model* m) { return m->n_total; }
extern "C" int64_t ishara_param_trainable(const ishara_model* m) { return m->n_train; }
extern "C" int32_t ishara_param_entries(const ishara_model* m) { return (int32_t)m->entries.size(); }
extern "C" int ishara_param_info(const ishara_model* m, int32_t i, const char** name, int32_t* ndim, int64_t shape[2], int64_t* offset, int32_t* trainable) {
    if (i < 0 || i >= (int)m->entries.size()) { ishara_set_error("param index out of range"); return -1; }
    const ParamEntry& e = m->entries[i];
    if (name) *name = e.name.c_str();
    if (ndim) *ndim = e.ndim;
    if (shape) { shape[0] = e.shape[0]; shape[1] = e.shape[1]; }
    if (offset) *offset = e.offset;
    if (trainable) *trainable = e.trainable ? 1 : 0;
    return 0;
}
extern "C" int64_t ishara_workspace_bytes(const ishara_model* m) { return (int64_t)m->ws_need; }
// Host-side audit of the workspace plan: every buffer 256-byte aligned, inside [0, workspace_bytes), no two buffers (or guard
// zones) overlapping.  Returns the number of buffers, <0 on a violation.
extern "C" int32_t ishara_workspace_plan_check(const ishara_model* m) {
    std::vector<std::pair<size_t, size_t>> r = m->allocs;
    for (size_t off : m->guard_offs) r.push_back({off, 256});
    std::sort(r.begin(), r.end());
    size_t end = 0;
    for (auto& a : r) {
        if (a.first % 256 != 0) { ishara_set_error("workspace buffer at %zu is not 256-byte aligned", a.first); return -1; }
        if (a.first < end) { ishara_set_error("workspace buffers overlap at offset %zu (previous buffer ends at %zu)", a.first, end); return -1; }
        end = a.first + a.second;
        if (end > m->ws_need) { ishara_set_error("workspace buffer [%zu, %zu) exceeds the planned size %zu", a.first, end, m->ws_need); return -1; }
    }
    return (int32_t)m->allocs.size();
}
// Guard zones (ISHARA_WS_GUARD=1 at create): synchronises the device, returns 0 when every guard still holds its pattern, else -3
// with the workspace offset of the first damaged guard (and the buffer in front of it) in ishara_last_error().  0 guards: returns 0.
extern "C" int ishara_workspace_guard_check(ishara_model* m) {
    if (!m->guard || !m->ws) return 0;
    HIP_CHECK_RET(hipDeviceSynchronize());
    std::vector<uint32_t> got(64);
    for (size_t gi = 0; gi < m->guard_offs.size(); ++gi) {
        HIP_CHECK_RET(hipMemcpy(got.data(), m->ws + m->guard_offs[gi], 256, hipMemcpyDeviceToHost));
        for (int i = 0; i < 64; ++i)
            if (got[i] != 0xA5C3A5C3u) {
                size_t boff = 0, bsz = 0;
                for (auto& a : m->allocs) if (a.first < m->guard_offs[gi] && a.first >= boff) { boff = a.first; bsz = a.second; }
                ishara_set_error("workspace guard %zu at offset %zu damaged at byte %d (value 0x%08x): the buffer in front of it is [%zu, %zu)",
                                 gi, m->guard_offs[gi], i * 4, got[i], boff, boff + bsz);
                return -3;
            }
    }
    return 0;
}

extern "C" int ishara_bind(ishara_model* m, float* params, float* grads, float* opt_m, float* opt_v, float* opt_slow, void* workspace, int64_t workspace_bytes) {
    if (!params || !workspace) { ishara_set_error("ishara_bind: params and workspace are required"); return -1; }
    if ((size_t)workspace_bytes < m->ws_need) { ishara_set_error("ishara_bind: workspace too small (%lld < %zu)", (long long)workspace_bytes, m->ws_need); return -1; }
    if (((uintptr_t)workspace) % 256 != 0) { ishara_set_error("ishara_bind: workspace must be 256-byte aligned"); return -1; }
    m->params = params; m->grads = grads; m->om = opt_m; m->ov = opt_v; m->oslow = opt_slow;
    m->ws = (char*)workspace; m->ws_bytes = workspace_bytes;
    m->shadow_ready = false;               // new buffers: rebuild the descriptor table and re-zero the padding
    if (m->guard) {
        std::vector<uint32_t> pat(64, 0xA5C3A5C3u);
        for (size_t off : m->guard_offs) HIP_CHECK_RET(hipMemcpy(m->ws + off, pat.data(), 256, hipMemcpyHostToDevice));
    }
    if (m->family == ISHARA_FAMILY_TORCH_SQUEEZEFORMER) CK(r4_bind(m));
    if (m->family == ISHARA_FAMILY_KERAS_HYBRID) HIP_CHECK_RET(hipMemcpy(m->ws + m->pe.off, m->pe_host.data(), m->pe_host.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ishara_sync_weights(ishara_model* m, ishara_stream st) {
    hipStream_t s = (hipStream_t)st;
    if (!m->ws) { ishara_set_error("not bound"); return -1; }
    // the zero padding of the shadow arena is written once; afterwards every step rewrites only the [K, N] interiors, all
    // weights in one launch (63 launches + a fill of the arena were 0.23 ms of a 21 ms step)
    if (!m->shadow_ready) {
        HIP_CHECK_RET(hipMemsetAsync(m->ws + m->shadow_begin, 0, m->shadow_end - m->shadow_begin, s));
        if (m->guard) {                          // the arena fill above also cleared the guard zones between the shadows: re-arm them
            static const std::vector<uint32_t> pat(64, 0xA5C3A5C3u);
            for (size_t off : m->guard_offs)
                if (off >= m->shadow_begin && off < m->shadow_end) HIP_CHECK_RET(hipMemcpyAsync(m->ws + off, pat.data(), 256, hipMemcpyHostToDevice, s));
        }
        std::vector<ShadowDesc> tab;
        int tile0 = 0;
        for (DenseW* w : m->denses) {
            ShadowDesc d;
            d.W = m->P(w->w); d.Wt = m->ws + w->wt; d.Wn = m->ws + w->wn; d.K = w->K; d.N = w->N; d.ldt = w->ldt; d.ldn = w->ldn;
            d.tile0 = tile0; d.tiles_n = (w->N + 31) / 32;
            tile0 += d.tiles_n * ((w->K + 31) / 32);
            tab.push_back(d);
        }
        m->shadow_tiles = tile0;
        m->shadow_ntab = (int)tab.size();
        m->shadow_tab_host = tab;
        HIP_CHECK_RET(hipMemcpyAsync(m->ws + m->shadow_tab_off, m->shadow_tab_host.data(), tab.size() * sizeof(ShadowDesc), hipMemcpyHostToDevice, s));
        m->shadow_ready = true;
    }
    CK(launch_make_shadow_batched(m->dt, reinterpret_cast<const ShadowDesc*>(m->ws + m->shadow_tab_off), m->shadow_ntab, m->shadow_tiles, s));
    return 0;
}

// ------------------------------------------------------------------ GEMM wrappers
int gemm_fwd(ishara_model* m, const DenseW& w, const void* A, int dtA, void* Cc, int dtC, int M, int aop, const OpArgs& oa, EpiArgs ea) {
    if (w.b >= 0) ea.bias = m->P(w.b);
    const double by = (double)M * w.K * dt_size(dtA) + (double)M * w.N * dt_size(dtC) * (1 + (ea.resid ? 1 : 0) + (ea.pre_out ? 1 : 0)) + (double)w.K * w.N * dt_size(m->dt);
    CKP(m, gemm_nt_kernel_name(dtA, m->dt, dtC, aop, A, M, w.N, w.K, w.ldt, ea), by, 2.0 * M * w.N * w.K, launch_gemm_nt(dtA, m->dt, dtC, aop, A, m->ws + w.wt, Cc, M, w.N, w.K, w.ldt, oa, ea, m->s));
    return 0;
}
int gemm_dgrad(ishara_model* m, const DenseW& w, const void* dY, int dtA, void* dX, int M, int aop, const OpArgs& oa, const EpiArgs& ea) {
    const double by = (double)M * w.N * dt_size(dtA) + (double)M * w.K * dt_size(m->dt) * (1 + (ea.resid ? 1 : 0) + (ea.aux ? 1 : 0)) + (double)w.K * w.N * dt_size(m->dt);
    CKP(m, gemm_nt_kernel_name(dtA, m->dt, m->dt, aop, dY, M, w.K, w.N, w.ldn, ea), by, 2.0 * M * w.N * w.K, launch_gemm_nt(dtA, m->dt, m->dt, aop, dY, m->ws + w.wn, dX, M, w.K, w.N, w.ldn, oa, ea, m->s));
    return 0;
}
// ---- deferred parameter-gradient sums (kernels.h RedSink): the LayerNorm / depthwise-conv backward operators leave their partial rows in a
// bump arena instead of the shared slab, and ONE launch at the end of the backward pass (or when the arena / job table is full, or before a
// gradient bucket is declared final) sums them all.  Off: ISHARA_NO_DEFERRED_REDUCE=1.
int red_flush(ishara_model* m) {
    if (m->red.njobs == 0) { m->red_off = 0; return 0; }
    CKP(m, "reduce_jobs(deferred)", 0, 0, launch_reduce_flush(&m->red, m->s));
    m->red_off = 0;
    return 0;
}
static float* red_scratch(ishara_model* m, size_t floats) {
    if (!m->red_on || floats > m->red_cap) return m->Wf(m->slab);
    floats = (floats + 63) & ~(size_t)63;
    if (m->red_off + floats > m->red_cap || reduce_sink_full()) { g_red_sink = nullptr; (void)red_flush(m); }
    g_red_sink = &m->red;
    float* p = m->Wf(m->red_arena) + m->red_off;
    m->red_off += floats;
    return p;
}
struct RedScope {      // the sink is installed by red_scratch (only for operators that got arena scratch) and removed when the launch returns
    explicit RedScope(ishara_model*) {}
    ~RedScope() { g_red_sink = nullptr; }
};
int gemm_wgrad(ishara_model* m, const DenseW& w, const void* A, int dtA, int aop, const OpArgs& oa, const void* dY, int dtB, int bop, const OpArgs& ob, int M, int ka_valid, int nb_valid,
               const float* bias_rowscale, int bias_T, const TnPsa* psa) {
    const double by = (double)M * w.K * dt_size(dtA) + (double)M * w.N * dt_size(dtB) + (double)w.K * w.N * 4;
    // the GEMM kernel and the sums of its split-M slabs are profiled under separate keys (the kernel's key is its rocprof name)
    if (m->tn_defer_on && !m->prof.on) {       // the sums of this GEMM's slabs ride with the next weight-gradient GEMM (gemm.hip, TnDefer)
        m->tn_defer.slab[0] = m->Wf(m->slab2[0]); m->tn_defer.slab[1] = m->Wf(m->slab2[1]);
        return launch_gemm_tn(dtA, dtB, m->dt, aop, bop, A, dY, m->G(w.w), w.b >= 0 ? m->G(w.b) : nullptr, m->Wf(m->slab), M, w.K, w.N, oa, ob, m->s, ka_valid, nb_valid, bias_rowscale, bias_T, &m->tn_defer, psa);
    }
    g_tn_phase = 1;
    // (the per-sample-affine variant is profiled under its own rocprof name: a different instantiation doing the statistics pass's work too)
    CKP(m, psa ? "gemm_tn_tr_kernel<0,false,true>" : gemm_tn_kernel_name(dtA, dtB, m->dt, aop, bop, M, w.K, w.N, bias_rowscale != nullptr), by, 2.0 * M * w.N * w.K, launch_gemm_tn(dtA, dtB, m->dt, aop, bop, A, dY, m->G(w.w), w.b >= 0 ? m->G(w.b) : nullptr, m->Wf(m->slab), M, w.K, w.N, oa, ob, m->s, ka_valid, nb_valid, bias_rowscale, bias_T, nullptr, psa));
    g_tn_phase = 2;
    CKP(m, "reduce_slabs(wgrad)", 0, 0, launch_gemm_tn(dtA, dtB, m->dt, aop, bop, A, dY, m->G(w.w), w.b >= 0 ? m->G(w.b) : nullptr, m->Wf(m->slab), M, w.K, w.N, oa, ob, m->s, ka_valid, nb_valid, bias_rowscale, bias_T, nullptr, psa));
    g_tn_phase = 0;
    return 0;
}


// keep-bit cache of the attention-probability dropout (forward writes, backward reads); ISHARA_NO_ATTN_BITS=1: both passes hash instead (A/B switch)
static uint32_t* attn_maskw(ishara_model* m, const Buf& off) {
    static const bool off_env = getenv("ISHARA_NO_ATTN_BITS") != nullptr;
    return off_env ? nullptr : reinterpret_cast<uint32_t*>(m->W(off));
}
int wgrad_flush(ishara_model* m) { return launch_gemm_tn_flush(&m->tn_defer, m->s); }

// LayerNorm as a prologue of the GEMM that consumes it (gemm_as.hip): the wave holds whole rows of K, so the statistics cost two
// cross-lane adds; the normalised rows go to `xn` (training: the weight-gradient GEMM reads them) and the statistics to mean / rstd.
// Shapes the A-stationary kernel does not take run the separate LayerNorm kernel.  Returns the GEMM's A operand.
static const void* ln_prologue(ishara_model* m, const DenseW& w, const Run& r, const void* x, const Norm& ln, float eps, Buf xn, Buf mean, Buf rstd, EpiArgs& ea, int* rc) {
    *rc = 0;
    EpiArgs probe = ea;
    probe.ln_gamma = m->P(ln.gamma); probe.ln_beta = m->P(ln.beta);
    probe.ln_mean = r.training ? m->Wf(mean) : nullptr; probe.pro_out = r.training ? m->W(xn) : nullptr;     // inference: no side outputs
    if (gemm_nt_as_prologue_ok(m->dt, m->dt, m->dt, r.M, w.N, w.K, w.ldt, probe)) {
        ea.ln_gamma = m->P(ln.gamma); ea.ln_beta = m->P(ln.beta); ea.ln_eps = eps;
        ea.ln_mean = probe.ln_mean; ea.ln_rstd = r.training ? m->Wf(rstd) : nullptr;
        ea.pro_out = probe.pro_out;
        return x;
    }
    *rc = [&]() -> int {
        CKP(m, "layernorm_fwd", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_fwd(m->dt, x, m->P(ln.gamma), m->P(ln.beta), eps, m->W(xn), m->Wf(mean), m->Wf(rstd), r.M, m->d, m->s));
        return 0;
    }();
    return m->W(xn);
}

// ------------------------------------------------------------------ module forward
static int conv_fwd(ishara_model* m, ConvBlock& cb, const Run& r, const void* x) {
    const int d = m->d, c = 2 * d, B = r.B, T = m->T, dt = m->dt;
    OpArgs no; EpiArgs e1;
    CK(gemm_fwd(m, cb.W1, x, dt, m->W(cb.z1), dt, r.M, OP_NONE, no, e1));
    // inference: the partial statistic rows of the depthwise conv are summed, and the BatchNorm constants formed from the moving statistics,
    // inside eca_fwd (4 launches per Conv1DBlock instead of 6: at B = 1 every launch is ~9 us of latency)
    int prows = 0;
    const bool infer_fused = !r.training && getenv("ISHARA_NO_INFER_FUSION") == nullptr;
    static const bool no_train_fusion = getenv("ISHARA_NO_STATS_FUSION") != nullptr;
    const bool train_fused = r.training && !no_train_fusion;
    CKP(m, "dwconv_fwd", 3.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_dwconv_fwd(dt, DWIN_SWISH, m->W(cb.z1), m->P(cb.dw), nullptr, m->W(cb.h2), m->Wf(cb.ssum), m->Wf(cb.ssq), m->Wf(m->slab), B, T, c, cb.k, cb.k - 1, m->s,
                                                                                          (infer_fused || train_fused) ? &prows : nullptr));
    if (train_fused && prows > 0)      // training: batch statistics straight from the partial rows [B * prows][2][C] (no stats_reduce launch; eca_fwd below sums them per sample)
    CKP(m, "bn_finalize", 0, 0, launch_bn_finalize(m->Wf(m->slab), m->Wf(m->slab) + c, B * prows, (float)B * T, m->P(cb.bn.gamma), m->P(cb.bn.beta), 1e-3f, 0.95f,
                          m->P(cb.bn.mm), m->P(cb.bn.mv), r.training, m->Wf(cb.mean), m->Wf(cb.rstd), m->Wf(cb.a), m->Wf(cb.bsh), c, m->s, 1.f, 2 * c));
    else if (!(infer_fused && prows > 0))
    CKP(m, "bn_finalize", 0, 0, launch_bn_finalize(m->Wf(cb.ssum), m->Wf(cb.ssq), B, (float)B * T, m->P(cb.bn.gamma), m->P(cb.bn.beta), 1e-3f, 0.95f,
                          m->P(cb.bn.mm), m->P(cb.bn.mv), r.training, m->Wf(cb.mean), m->Wf(cb.rstd), m->Wf(cb.a), m->Wf(cb.bsh), c, m->s));
    // Drop-path (c5:82-83) on the branch: y = x + rs[b] * (h4 W2 + b2).  Where the fast kernels apply, rs[b] is folded into the per-sample
    // affine that produces h4 (h4 = rs[b] * (h2 P + Q), free), the GEMM adds rs[b] * b2, and the backward pass needs no scaled copy of
    // the incoming gradient: dgrad scales its OUTPUT rows, wgrad multiplies h4^T by the plain gradient and weights the bias sum.
    const DropSpec ds = dspec(r, cb.site, m->cfg.dropout_rate);
    EpiArgs e2; e2.resid = x;
    cb.folded = false; cb.psa = false;
    if (ds.thr) {                                        // rs[b] itself is drawn by eca_fwd below (one launch less per block)
        e2.rowscale = m->Wf(cb.rs); e2.T = T;
        EpiArgs probe = e2; probe.bias = m->P(cb.W2.b);
        cb.folded = dt == DT_BF16 && !g_force_regstage && gemm_nt_as_applicable(dt, r.M, cb.W2.N, cb.W2.K, cb.W2.ldt, probe) && gemm_nt_as_applicable(dt, r.M, cb.W2.K, cb.W2.N, cb.W2.ldn, probe) &&
                    gemm_tn_bias_rowscale_ok(dt, dt, dt, r.M, cb.W2.K, cb.W2.N, T);
        e2.rowscale_bias = cb.folded ? 1 : 0;
    }
    if (infer_fused && prows > 0)
        CKP(m, "eca_fwd", 0, 0, launch_eca_fwd_infer(m->Wf(m->slab), prows, m->P(cb.bn.mm), m->P(cb.bn.mv), m->P(cb.bn.gamma), m->P(cb.bn.beta), 1e-3f, m->P(cb.eca), 1.f / T,
                                                       m->Wf(cb.gn), m->Wf(cb.sg), m->Wf(cb.P), m->Wf(cb.Q), B, c, m->s));
    else if (train_fused && prows > 0)
        CKP(m, "eca_fwd", 0, 0, launch_eca_fwd_part(m->Wf(m->slab), prows, m->Wf(cb.ssum), m->Wf(cb.a), m->Wf(cb.bsh), m->P(cb.eca), 1.f / T, m->Wf(cb.gn), m->Wf(cb.sg), m->Wf(cb.P), m->Wf(cb.Q), B, c, m->s,
                                                      ds.thr ? m->Wf(cb.rs) : nullptr, ds, cb.folded ? 1 : 0));
    else
    CKP(m, "eca_fwd", 0, 0, launch_eca_fwd(m->Wf(cb.ssum), m->Wf(cb.a), m->Wf(cb.bsh), m->P(cb.eca), 1.f / T, m->Wf(cb.gn), m->Wf(cb.sg), m->Wf(cb.P), m->Wf(cb.Q), B, c, m->s, ds.thr ? m->Wf(cb.rs) : nullptr, ds, cb.folded ? 1 : 0));
    // h4 = h2 * P[b] + Q[b] (BatchNorm + ECA gate [+ drop-path]) as a prologue of the project GEMM: h2 is read once, h4 is written from
    // the transformed fragments for the weight-gradient GEMM (training only); other shapes run the separate affine pass
    {
        EpiArgs probe = e2; probe.pa_P = m->Wf(cb.P); probe.pa_Q = m->Wf(cb.Q); probe.T = T; probe.bias = m->P(cb.W2.b);
        probe.pro_out = r.training ? m->W(cb.h4) : nullptr;
        if (gemm_nt_as_prologue_ok(dt, dt, dt, r.M, cb.W2.N, cb.W2.K, cb.W2.ldt, probe)) {
            // training: h4 is written only when the backward pass needs it in memory — not when the project conv's weight-gradient GEMM applies
            // P, Q itself (gemm.hip TnPsa: whole samples per M-split; the drop-path scale, if any, must be the folded one)
            cb.psa = r.training && m->psa_on && (!ds.thr || cb.folded) && gemm_tn_psa_ok(dt, dt, dt, r.M, cb.W2.K, cb.W2.N, T);
            e2.pa_P = m->Wf(cb.P); e2.pa_Q = m->Wf(cb.Q); e2.T = T; e2.pro_out = (r.training && !cb.psa) ? m->W(cb.h4) : nullptr;
            CK(gemm_fwd(m, cb.W2, m->W(cb.h2), dt, m->W(cb.out), dt, r.M, OP_NONE, no, e2));
            return 0;
        }
    }
    CKP(m, "sample_affine", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_affine(dt, m->W(cb.h2), m->Wf(cb.P), m->Wf(cb.Q), nullptr, m->W(cb.h4), B, T, c, m->s));
    CK(gemm_fwd(m, cb.W2, m->W(cb.h4), dt, m->W(cb.out), dt, r.M, OP_NONE, no, e2));
    return 0;
}

static int ffn_fwd(ishara_model* m, FFN& f, const Run& r, const void* x) {
    const int dt = m->dt;
    OpArgs no;
    EpiArgs ea; ea.pre_out = m->W(f.za); ea.act = ACT_SWISH; ea.drop = dspec(r, f.site_in, m->cfg.dropout_rate);
    int rc;
    const void* ain = ln_prologue(m, f.Wa, r, x, f.ln, f.eps, f.xn, f.mean, f.rstd, ea, &rc);
    CK(rc);
    CK(gemm_fwd(m, f.Wa, ain, dt, m->W(f.u), dt, r.M, OP_NONE, no, ea));
    EpiArgs eb; eb.resid = x;
    if (f.has_out_drop) eb.drop = dspec(r, f.site_out, m->cfg.dropout_rate);
    CK(gemm_fwd(m, f.Wb, m->W(f.u), dt, m->W(f.out), dt, r.M, OP_NONE, no, eb));
    return 0;
}

static int mhsa_fwd(ishara_model* m, MHSA& a, const Run& r, const void* x) {
    const int dt = m->dt;
    OpArgs no;
    EpiArgs eq; eq.mode = EPI_QKV; eq.q = m->W(a.q); eq.k = m->W(a.k); eq.vt = m->W(a.vt); eq.H = m->H; eq.dh = m->dh; eq.T = m->T; eq.head_major = 1;
    int rc;
    const void* ain = ln_prologue(m, a.Wqkv, r, x, a.ln, a.eps, a.xn, a.mean, a.rstd, eq, &rc);
    CK(rc);
    CK(gemm_fwd(m, a.Wqkv, ain, dt, nullptr, dt, r.M, OP_NONE, no, eq));
    const float scale = 1.0f / sqrtf((float)m->d);     // self.scale = dim ** -0.5 (c5:95)
    CKP(m, "attn_fwd", 4.0 * r.M * m->d * (double)dt_size(m->dt), 4.0 * r.B * m->H * (double)m->T * m->T * m->dh, launch_attn_fwd(dt, m->W(a.q), m->W(a.k), m->W(a.vt), m->W(a.o), m->Wf(a.lse), r.B, m->H, m->T, m->dh, scale,
                       dspec_attn(r, a.site_attn, a.rate), m->cfg.attn_impl, attn_maskw(m, a.maskw), m->s));
    EpiArgs ep; ep.resid = x;
    if (a.has_out_drop) ep.drop = dspec(r, a.site_out, m->cfg.dropout_rate);
    CK(gemm_fwd(m, a.Wp, m->W(a.o), dt, m->W(a.out), dt, r.M, OP_NONE, no, ep));
    return 0;
}

static int sqzconv_fwd(ishara_model* m, SqzConv& c, const Run& r, const void* x) {
    const int dt = m->dt, d = m->d, de = c.Wc1.N, B = r.B, T = m->T;
    OpArgs no; EpiArgs e0;
    EpiArgs e1;
    int rc;
    const void* ain = ln_prologue(m, c.Wc1, r, x, c.ln, 1e-6f, c.xn, c.mean, c.rstd, e1, &rc);
    CK(rc);
    CK(gemm_fwd(m, c.Wc1, ain, dt, m->W(c.zc), dt, r.M, OP_NONE, no, e1));
    CKP(m, "dwconv_fwd", 3.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_dwconv_fwd(dt, DWIN_SWISH, m->W(c.zc), m->P(c.dw), nullptr, m->W(c.zd), nullptr, nullptr, nullptr, B, T, de, c.k, c.k - 1, m->s));
    CKP(m, "map_rows", 2.0 * r.M * de * (double)dt_size(m->dt), 0, launch_map_rows(dt, MAP_SWISH, m->W(c.zd), m->W(c.hd), nullptr, DropSpec{0, 0, 1.f}, r.M, T, de, m->s));
    CK(gemm_fwd(m, c.Wc3, m->W(c.hd), dt, m->W(c.u3), dt, r.M, OP_NONE, no, e0));
    CKP(m, "sample_reduce", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_reduce(dt, m->W(c.u3), nullptr, nullptr, nullptr, m->Wf(c.gap), nullptr, B, T, d, m->s));
    CKP(m, "se_fwd", 0, 0, launch_se_fwd(m->Wf(c.gap), 1.f / T, m->P(c.seW1), m->P(c.seb1), m->P(c.seW2), m->P(c.seb2), m->Wf(c.hid), m->Wf(c.se), B, d, c.R, m->s));
    CKP(m, "sample_affine", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_affine(dt, m->W(c.u3), m->Wf(c.se), nullptr, x, m->W(c.out), B, T, d, m->s));
    return 0;
}

int confconv_fwd(ishara_model* m, ConfConv& c, const Run& r, const void* x) {
    const int dt = m->dt, d = m->d, B = r.B, T = m->T;
    OpArgs no; EpiArgs e0;
    CK(gemm_fwd(m, c.Wp1, x, dt, m->W(c.g), dt, r.M, OP_NONE, no, e0));
    int prows = 0;
    static const bool no_train_fusion = getenv("ISHARA_NO_STATS_FUSION") != nullptr;
    const bool train_fused = r.training && !no_train_fusion;        // batch statistics straight from the depthwise conv's partial rows: no stats_reduce launch
    CKP(m, "dwconv_fwd", 3.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_dwconv_fwd(dt, DWIN_GLU, m->W(c.g), m->P(c.dw), c.dwb >= 0 ? m->P(c.dwb) : nullptr, m->W(c.v), r.training ? m->Wf(c.ssum) : nullptr, r.training ? m->Wf(c.ssq) : nullptr, m->Wf(m->slab), B, T, d, c.k, (c.k - 1) / 2, m->s,
                                                                                          train_fused ? &prows : nullptr));      // inference: no batch statistics
    const float var_corr = c.bn_unbiased && B * T > 1 ? (float)((double)B * T / ((double)B * T - 1.0)) : 1.f;
    if (train_fused && prows > 0)
    CKP(m, "bn_finalize", 0, 0, launch_bn_finalize(m->Wf(m->slab), m->Wf(m->slab) + d, B * prows, (float)B * T, m->P(c.bn.gamma), m->P(c.bn.beta), c.bn_eps, c.bn_keep,
                          m->P(c.bn.mm), m->P(c.bn.mv), r.training, m->Wf(c.mean), m->Wf(c.rstd), m->Wf(c.a), m->Wf(c.bsh), d, m->s, var_corr, 2 * d));
    else
    CKP(m, "bn_finalize", 0, 0, launch_bn_finalize(m->Wf(c.ssum), m->Wf(c.ssq), B, (float)B * T, m->P(c.bn.gamma), m->P(c.bn.beta), c.bn_eps, c.bn_keep,
                          m->P(c.bn.mm), m->P(c.bn.mv), r.training, m->Wf(c.mean), m->Wf(c.rstd), m->Wf(c.a), m->Wf(c.bsh), d, m->s, var_corr));
    CKP(m, "col_affine", 2.0 * r.M * d * (double)dt_size(m->dt), 0, launch_col_affine(dt, m->W(c.v), m->Wf(c.a), m->Wf(c.bsh), m->W(c.bnv), r.M, d, m->s));
    const void* pin = m->W(c.bnv);
    if (c.swish_after_bn) {
        CKP(m, "map_rows", 2.0 * r.M * d * (double)dt_size(m->dt), 0, launch_map_rows(dt, MAP_SWISH, m->W(c.bnv), m->W(c.sw), nullptr, DropSpec{0, 0, 1.f}, r.M, T, d, m->s));
        pin = m->W(c.sw);
    }
    EpiArgs e2; e2.resid = x;
    if (c.has_out_drop) e2.drop = dspec(r, c.site_out, m->cfg.dropout_rate);
    CK(gemm_fwd(m, c.Wp2, pin, dt, m->W(c.r), dt, r.M, OP_NONE, no, e2));
    CKP(m, "layernorm_fwd", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_fwd(dt, m->W(c.r), m->P(c.ln.gamma), m->P(c.ln.beta), c.ln_eps, m->W(c.out), m->Wf(c.lnmean), m->Wf(c.lnrstd), r.M, d, m->s));
    return 0;
}

extern "C" int ishara_forward(ishara_model* m, const float* x, int32_t B, float* logits, int32_t training, uint32_t seed, ishara_stream st) {
    if (!m->ws) { ishara_set_error("ishara_forward: model is not bound"); return -1; }
    if (m->family != ISHARA_FAMILY_KERAS_HYBRID) { ishara_set_error("ishara_forward: this handle is an encoder-only family; use ishara_encoder_forward"); return -1; }
    if (B <= 0 || B > m->Bmax) { ishara_set_error("ishara_forward: batch %d outside 1..%d", B, m->Bmax); return -1; }
    if (m->dt == DT_F16 && training) {
        ishara_set_error("ishara_forward: ISHARA_F16 is an inference-only storage type (the reference's fp16 is the TFLite export, c14:1-5; it reports NaNs when TRAINING in fp16)");
        return -1;
    }
    m->s = (hipStream_t)st;
    Run r{B, B * m->T, training, seed};
    const int dt = m->dt, d = m->d, T = m->T;
    OpArgs no;
    // ---- stem: Dense(no bias) + PE, BatchNorm(momentum .95)  (c7:13-17)
    EpiArgs es; es.addtab = m->Wf(m->pe); es.tab_period = T;
    if (m->stem_kp) {       // bf16: pack the input rows once, then the Dense (and its wgrad) run on the bf16 fast paths with K = stem_kp
        CKP(m, "pack_rows_bf16", (double)r.M * (m->F * 4.0 + m->stem_kp * 2.0), 0, launch_pack_rows_bf16(x, m->W(m->stem_xb), r.M, m->F, m->stem_kp, m->s, dt));
        DenseW wp = m->stemW; wp.K = m->stem_kp;
        CK(gemm_fwd(m, wp, m->W(m->stem_xb), dt, m->W(m->stem_h0), dt, r.M, OP_NONE, no, es));
    } else
        CK(gemm_fwd(m, m->stemW, x, DT_F32, m->W(m->stem_h0), dt, r.M, OP_NONE, no, es));
    if (training)           // (inference: the BatchNorm uses its moving statistics, nothing reads the batch sums)
    CKP(m, "sample_reduce", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_reduce(dt, m->W(m->stem_h0), m->W(m->stem_h0), nullptr, nullptr, m->Wf(m->stem_ssum), m->Wf(m->stem_ssq), B, T, d, m->s));
    CKP(m, "bn_finalize", 0, 0, launch_bn_finalize(m->Wf(m->stem_ssum), m->Wf(m->stem_ssq), B, (float)B * T, m->P(m->stem_bn.gamma), m->P(m->stem_bn.beta), 1e-3f, 0.95f,
                          m->P(m->stem_bn.mm), m->P(m->stem_bn.mv), training, m->Wf(m->stem_mean), m->Wf(m->stem_rstd), m->Wf(m->stem_a), m->Wf(m->stem_bsh), d, m->s));
    CKP(m, "col_affine", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_col_affine(dt, m->W(m->stem_h0), m->Wf(m->stem_a), m->Wf(m->stem_bsh), m->W(m->stem_out), r.M, d, m->s));
    const void* h = m->W(m->stem_out);
    for (const Layer& L : m->layers) {
        if (L.kind == Layer::CONV) { ConvBlock& cb = m->convs[L.idx]; CK(conv_fwd(m, cb, r, h)); h = m->W(cb.out); }
        else if (L.kind == Layer::SQZ) {
            SqzBlock& sb = m->sqz[L.idx];
            CK(ffn_fwd(m, sb.ffn1, r, h)); h = m->W(sb.ffn1.out);
            CK(mhsa_fwd(m, sb.mha, r, h)); h = m->W(sb.mha.out);
            CK(sqzconv_fwd(m, sb.conv, r, h)); h = m->W(sb.conv.out);
            CK(ffn_fwd(m, sb.ffn2, r, h)); h = m->W(sb.ffn2.out);
        } else {
            ConfBlock& cb = m->conf[L.idx];
            CK(ffn_fwd(m, cb.ffn1, r, h)); h = m->W(cb.ffn1.out);
            CK(mhsa_fwd(m, cb.mha, r, h)); h = m->W(cb.mha.out);
            CK(confconv_fwd(m, cb.conv, r, h)); h = m->W(cb.conv.out);
            CK(ffn_fwd(m, cb.ffn2, r, h)); h = m->W(cb.ffn2.out);
        }
    }
    // ---- head: Dense(relu) -> Dropout(0.4) -> Dense  (c7:61-63)
    EpiArgs et; et.act = ACT_RELU; et.drop = dspec(r, m->head_site, m->cfg.head_dropout);
    CK(gemm_fwd(m, m->topW, h, dt, m->W(m->head_hh), dt, r.M, OP_NONE, no, et));
    EpiArgs ec;
    if (dt_is16(dt) && r.M <= 1536 && m->C <= 64 && m->C % 4 == 0 && (m->clsW.K == 256 || m->clsW.K == 512) && !g_force_regstage && getenv("ISHARA_NO_INFER_FUSION") == nullptr) {
        // a clip's worth of rows: the A-stationary MFMA kernel over the zero-padded weight shadow (N = 64), storing the real columns only
        ec.bias = m->clsW.b >= 0 ? m->P(m->clsW.b) : nullptr; ec.ldc = m->C; ec.n_valid = m->C;
        CKP(m, "classifier(as)", 0, 2.0 * r.M * m->C * m->clsW.K, launch_gemm_nt(dt, dt, DT_F32, OP_NONE, m->W(m->head_hh), m->ws + m->clsW.wt, logits, r.M, 64, m->clsW.K, m->clsW.ldt, no, ec, m->s));
    } else if (dt_is16(dt) && r.M <= 4096 && m->C <= 64 && m->clsW.K % 32 == 0 && getenv("ISHARA_NO_INFER_FUSION") == nullptr)       // few rows: the narrow-output kernel (latency of a clip)
        CKP(m, "dense_narrow", 0, 2.0 * r.M * m->C * m->clsW.K, launch_dense_narrow(dt, m->W(m->head_hh), m->ws + m->clsW.wt, m->clsW.ldt, m->clsW.b >= 0 ? m->P(m->clsW.b) : nullptr, logits, r.M, m->C, m->clsW.K, m->s));
    else
    CK(gemm_fwd(m, m->clsW, m->W(m->head_hh), dt, logits, DT_F32, r.M, OP_NONE, no, ec));
    m->lastB = B; m->last_training = training; m->last_seed = seed; m->last_x = x;
    return 0;
}

// ------------------------------------------------------------------ module backward
// each *_bwd consumes g (grad wrt the module output) and writes gn (grad wrt its input x)
static int conv_bwd(ishara_model* m, ConvBlock& cb, const Run& r, const void* x, const void* g, void* gn) {
    const int d = m->d, c = 2 * d, B = r.B, T = m->T, dt = m->dt;
    OpArgs no;
    const DropSpec ds = dspec(r, cb.site, m->cfg.dropout_rate);
    const void* gs = g;                                  // gradient through the drop-path: dY * rs[b]
    EpiArgs e1;
    if (cb.psa) {                                        // h4 was never written: dW2 = sum_b diag(P_b) h2_b^T g_b + Q_b x colsum(g_b) inside the GEMM, which
        const float* rs = ds.thr ? m->Wf(cb.rs) : nullptr;   // also emits the statistics of dh4 (S1, S2 below) — no pass over dh4 and h2
        if (rs) { e1.rowscale = rs; e1.T = T; }
        CK(gemm_dgrad(m, cb.W2, g, dt, m->W(m->t1), r.M, OP_NONE, no, e1));
        TnPsa ps; ps.P = m->Wf(cb.P); ps.Q = m->Wf(cb.Q); ps.W = m->ws + cb.W2.wn; ps.ldw = cb.W2.ldn; ps.G = m->Wf(m->psaG); ps.Rpart = m->Wf(m->psaR); ps.T = T;
        { static const int psa_dbg = getenv("ISHARA_PSA_DBG") ? atoi(getenv("ISHARA_PSA_DBG")) : 0; ps.dbg = psa_dbg; }
        CK(gemm_wgrad(m, cb.W2, m->W(cb.h2), dt, OP_NONE, no, g, dt, OP_NONE, no, r.M, 0, 0, rs, rs ? T : 0, &ps));
    } else if (ds.thr && cb.folded) {                    // h4 carries rs[b] (conv_fwd): dh4 = (g W2^T) * rs[b]; dW2 = h4^T g; db2 = sum_m rs[b(m)] g[m]
        e1.rowscale = m->Wf(cb.rs); e1.T = T;
        CK(gemm_dgrad(m, cb.W2, g, dt, m->W(m->t1), r.M, OP_NONE, no, e1));
        CK(gemm_wgrad(m, cb.W2, m->W(cb.h4), dt, OP_NONE, no, g, dt, OP_NONE, no, r.M, 0, 0, m->Wf(cb.rs), T));
    } else {
        if (ds.thr) {
            CKP(m, "map_rows", 2.0 * r.M * d * (double)dt_size(m->dt), 0, launch_map_rows(dt, MAP_ROWSCALE, g, m->W(m->t3), m->Wf(cb.rs), ds, r.M, T, d, m->s));
            gs = m->W(m->t3);
        }
        CK(gemm_dgrad(m, cb.W2, gs, dt, m->W(m->t1), r.M, OP_NONE, no, e1));                       // dh4
        CK(gemm_wgrad(m, cb.W2, m->W(cb.h4), dt, OP_NONE, no, gs, dt, OP_NONE, no, r.M));
    }
    if (!cb.psa) CKP(m, "sample_reduce", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_reduce(dt, m->W(m->t1), m->W(cb.h2), m->Wf(cb.mean), m->Wf(cb.rstd), m->Wf(m->S1), m->Wf(m->S2), B, T, c, m->s));
    PsaStats pst;                                        // cb.psa: S1, S2 come out of the finalize kernel itself (from G and Rpart)
    if (cb.psa) { pst.G = m->Wf(m->psaG); pst.Rpart = m->Wf(m->psaR); pst.nparts = cb.W2.N / 64; pst.Wt = m->ws + cb.W2.wt; pst.ldt = cb.W2.ldt; pst.N = cb.W2.N;
                  pst.rs = ds.thr ? m->Wf(cb.rs) : nullptr; pst.mean = m->Wf(cb.mean); pst.rstd = m->Wf(cb.rstd); }
    CKP(m, "eca_bn_bwd_finalize", 0, 0, launch_eca_bn_bwd_finalize(m->Wf(m->S1), m->Wf(m->S2), m->Wf(cb.ssum), m->Wf(cb.gn), m->Wf(cb.sg), m->P(cb.eca), m->P(cb.bn.gamma), m->P(cb.bn.beta),
                                  m->Wf(cb.mean), m->Wf(cb.rstd), m->G(cb.bn.gamma), m->G(cb.bn.beta), m->G(cb.eca), m->Wf(m->E), m->Wf(m->Fc), m->Wf(m->ecap), B, T, c, m->s, cb.psa ? &pst : nullptr));
    // BatchNorm backward applied inside the depthwise-conv backward (one pass over dh4, h2 and z1); shapes without the fused kernel
    // take the two-kernel path
    DwBnArgs bn; bn.h = m->W(cb.h2); bn.mean = m->Wf(cb.mean); bn.rstd = m->Wf(cb.rstd); bn.a = m->Wf(cb.a); bn.sg = m->Wf(cb.sg); bn.E = m->Wf(m->E); bn.Fc = m->Wf(m->Fc); bn.e_per_sample = 1;
    int fused = 0;
    { RedScope red_scope_(m); CKP(m, "dwconv_bwd", 10.0 * r.M * m->d * (double)dt_size(m->dt), 0, (fused = launch_dwconv_bwd_bn(dt, DWIN_SWISH, m->W(m->t1), bn, m->W(cb.z1), m->P(cb.dw), m->W(m->t2), m->G(cb.dw), nullptr, red_scratch(m, dwconv_bwd_scratch_floats(c, cb.k)), B, T, c, cb.k, cb.k - 1, m->s)) < 0 ? fused : 0); }
    if (!fused) {
        CKP(m, "bn_bwd_apply", 6.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_bn_bwd_apply(dt, m->W(m->t1), m->W(cb.h2), m->Wf(cb.mean), m->Wf(cb.rstd), m->Wf(cb.a), m->Wf(cb.sg), m->Wf(m->E), 1, m->Wf(m->Fc), m->W(m->t1), B, T, c, m->s));
        { RedScope red_scope_(m); CKP(m, "dwconv_bwd", 8.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_dwconv_bwd(dt, DWIN_SWISH, m->W(m->t1), m->W(cb.z1), m->P(cb.dw), m->W(m->t2), m->G(cb.dw), nullptr, red_scratch(m, dwconv_bwd_scratch_floats(c, cb.k)), B, T, c, cb.k, cb.k - 1, m->s)); }
    }
    EpiArgs e2; e2.resid = g;
    CK(gemm_dgrad(m, cb.W1, m->W(m->t2), dt, gn, r.M, OP_NONE, no, e2));
    CK(gemm_wgrad(m, cb.W1, x, dt, OP_NONE, no, m->W(m->t2), dt, OP_NONE, no, r.M));
    return 0;
}

static int ffn_bwd(ishara_model* m, FFN& f, const Run& r, const void* x, const void* g, void* gn) {
    const int dt = m->dt;
    OpArgs no;
    const void* gs = g;                                  // gradient through the outer dropout
    if (f.has_out_drop) {
        const DropSpec od = dspec(r, f.site_out, m->cfg.dropout_rate);
        if (od.thr) {
            CKP(m, "map_rows", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_map_rows(dt, MAP_DROPMASK, g, m->W(m->t3), nullptr, od, r.M, m->T, m->d, m->s));
            gs = m->W(m->t3);
        }
    }
    EpiArgs e1; e1.drop = dspec(r, f.site_in, m->cfg.dropout_rate); e1.dact = DACT_SWISH; e1.aux = m->W(f.za);
    CK(gemm_dgrad(m, f.Wb, gs, dt, m->W(m->t1), r.M, OP_NONE, no, e1));                         // dza
    CK(gemm_wgrad(m, f.Wb, m->W(f.u), dt, OP_NONE, no, gs, dt, OP_NONE, no, r.M));
    EpiArgs e0;
    CK(gemm_dgrad(m, f.Wa, m->W(m->t1), dt, m->W(m->t2), r.M, OP_NONE, no, e0));              // dxn
    CK(gemm_wgrad(m, f.Wa, m->W(f.xn), dt, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M));
    { RedScope red_scope_(m); CKP(m, "layernorm_bwd", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_bwd(dt, m->W(m->t2), x, m->Wf(f.mean), m->Wf(f.rstd), m->P(f.ln.gamma), g, gn, m->G(f.ln.gamma), m->G(f.ln.beta), red_scratch(m, layernorm_bwd_scratch_floats(m->d)), r.M, m->d, m->s)); }
    return 0;
}

static int mhsa_bwd(ishara_model* m, MHSA& a, const Run& r, const void* x, const void* g, void* gn) {
    const int dt = m->dt;
    OpArgs no; EpiArgs e0;
    const void* gs = g;
    if (a.has_out_drop) {
        const DropSpec od = dspec(r, a.site_out, m->cfg.dropout_rate);
        if (od.thr) {
            CKP(m, "map_rows", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_map_rows(dt, MAP_DROPMASK, g, m->W(m->t3), nullptr, od, r.M, m->T, m->d, m->s));
            gs = m->W(m->t3);
        }
    }
    CK(gemm_dgrad(m, a.Wp, gs, dt, m->W(m->t1), r.M, OP_NONE, no, e0));                         // do
    CK(gemm_wgrad(m, a.Wp, m->W(a.o), dt, OP_NONE, no, gs, dt, OP_NONE, no, r.M));
    const float scale = 1.0f / sqrtf((float)m->d);
    CKP(m, "attn_bwd", 8.0 * r.M * m->d * (double)dt_size(m->dt), 10.0 * r.B * m->H * (double)m->T * m->T * m->dh, launch_attn_bwd(dt, m->W(a.q), m->W(a.k), m->W(a.vt), m->W(a.o), m->W(m->t1), m->Wf(a.lse), m->Wf(m->delta), m->W(m->t2),
                       r.B, m->H, m->T, m->dh, scale, dspec_attn(r, a.site_attn, a.rate), 1, m->cfg.attn_impl, attn_maskw(m, a.maskw), m->s));
    CK(gemm_dgrad(m, a.Wqkv, m->W(m->t2), dt, m->W(m->t1), r.M, OP_NONE, no, e0));            // dxn
    CK(gemm_wgrad(m, a.Wqkv, m->W(a.xn), dt, OP_NONE, no, m->W(m->t2), dt, OP_NONE, no, r.M));
    { RedScope red_scope_(m); CKP(m, "layernorm_bwd", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_bwd(dt, m->W(m->t1), x, m->Wf(a.mean), m->Wf(a.rstd), m->P(a.ln.gamma), g, gn, m->G(a.ln.gamma), m->G(a.ln.beta), red_scratch(m, layernorm_bwd_scratch_floats(m->d)), r.M, m->d, m->s)); }
    return 0;
}

static int sqzconv_bwd(ishara_model* m, SqzConv& c, const Run& r, const void* x, const void* g, void* gn) {
    const int dt = m->dt, d = m->d, de = c.Wc1.N, B = r.B, T = m->T;
    OpArgs no; EpiArgs e0;
    CKP(m, "sample_reduce", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_reduce(dt, g, m->W(c.u3), nullptr, nullptr, m->Wf(m->S1), m->Wf(m->dse), B, T, d, m->s));   // dse = sum_t g*u3
    CKP(m, "se_bwd", 0, 0, launch_se_bwd(m->Wf(m->dse), m->Wf(c.gap), 1.f / T, m->P(c.seW1), m->P(c.seW2), m->Wf(c.hid), m->Wf(c.se),
                     m->G(c.seW1), m->G(c.seb1), m->G(c.seW2), m->G(c.seb2), m->Wf(m->dgapT), m->Wf(m->E), B, d, c.R, m->s));
    CKP(m, "sample_affine", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_affine(dt, g, m->Wf(c.se), m->Wf(m->dgapT), nullptr, m->W(m->t1), B, T, d, m->s));           // du3
    EpiArgs e1; e1.dact = DACT_SWISH; e1.aux = m->W(c.zd);
    CK(gemm_dgrad(m, c.Wc3, m->W(m->t1), dt, m->W(m->t2), r.M, OP_NONE, no, e1));                                // dzd
    CK(gemm_wgrad(m, c.Wc3, m->W(c.hd), dt, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M));
    { RedScope red_scope_(m); CKP(m, "dwconv_bwd", 8.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_dwconv_bwd(dt, DWIN_SWISH, m->W(m->t2), m->W(c.zc), m->P(c.dw), m->W(m->t1), m->G(c.dw), nullptr, red_scratch(m, dwconv_bwd_scratch_floats(de, c.k)), B, T, de, c.k, c.k - 1, m->s)); }   // dzc
    CK(gemm_dgrad(m, c.Wc1, m->W(m->t1), dt, m->W(m->t2), r.M, OP_NONE, no, e0));                                // dxn
    CK(gemm_wgrad(m, c.Wc1, m->W(c.xn), dt, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M));
    { RedScope red_scope_(m); CKP(m, "layernorm_bwd", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_bwd(dt, m->W(m->t2), x, m->Wf(c.mean), m->Wf(c.rstd), m->P(c.ln.gamma), g, gn, m->G(c.ln.gamma), m->G(c.ln.beta), red_scratch(m, layernorm_bwd_scratch_floats(m->d)), r.M, d, m->s)); }
    return 0;
}

int confconv_bwd(ishara_model* m, ConfConv& c, const Run& r, const void* x, const void* g, void* gn) {
    const int dt = m->dt, d = m->d, B = r.B, T = m->T;
    OpArgs no;
    { RedScope red_scope_(m); CKP(m, "layernorm_bwd", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_bwd(dt, g, m->W(c.r), m->Wf(c.lnmean), m->Wf(c.lnrstd), m->P(c.ln.gamma), nullptr, m->W(m->t1), m->G(c.ln.gamma), m->G(c.ln.beta), red_scratch(m, layernorm_bwd_scratch_floats(m->d)), r.M, d, m->s)); }   // dr
    const void* gs = m->W(m->t1);                          // gradient through the module's output dropout
    if (c.has_out_drop) {
        const DropSpec od = dspec(r, c.site_out, m->cfg.dropout_rate);
        if (od.thr) {
            CKP(m, "map_rows", 2.0 * r.M * d * (double)dt_size(m->dt), 0, launch_map_rows(dt, MAP_DROPMASK, m->W(m->t1), m->W(m->t3), nullptr, od, r.M, T, d, m->s));
            gs = m->W(m->t3);
        }
    }
    EpiArgs es; if (c.swish_after_bn) { es.dact = DACT_SWISH; es.aux = m->W(c.bnv); }
    CK(gemm_dgrad(m, c.Wp2, gs, dt, m->W(m->t2), r.M, OP_NONE, no, es));                                          // d bn(v)
    CK(gemm_wgrad(m, c.Wp2, c.swish_after_bn ? m->W(c.sw) : m->W(c.bnv), dt, OP_NONE, no, gs, dt, OP_NONE, no, r.M));
    CKP(m, "sample_reduce", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_reduce(dt, m->W(m->t2), m->W(c.v), m->Wf(c.mean), m->Wf(c.rstd), m->Wf(m->S1), m->Wf(m->S2), B, T, d, m->s));
    CKP(m, "bn_bwd_finalize", 0, 0, launch_bn_bwd_finalize(m->Wf(m->S1), m->Wf(m->S2), m->G(c.bn.gamma), m->G(c.bn.beta), m->Wf(m->Ecol), m->Wf(m->Fc), B, T, d, m->s));
    DwBnArgs bn; bn.h = m->W(c.v); bn.mean = m->Wf(c.mean); bn.rstd = m->Wf(c.rstd); bn.a = m->Wf(c.a); bn.E = m->Wf(m->Ecol); bn.Fc = m->Wf(m->Fc);
    int fused = 0;
    { RedScope red_scope_(m); CKP(m, "dwconv_bwd", 6.0 * r.M * m->d * (double)dt_size(m->dt), 0, (fused = launch_dwconv_bwd_bn(dt, DWIN_GLU, m->W(m->t2), bn, m->W(c.g), m->P(c.dw), m->W(m->t3), m->G(c.dw), c.dwb >= 0 ? m->G(c.dwb) : nullptr, red_scratch(m, dwconv_bwd_scratch_floats(d, c.k)), B, T, d, c.k, (c.k - 1) / 2, m->s)) < 0 ? fused : 0); }   // dg [M,2d]
    if (!fused) {
        CKP(m, "bn_bwd_apply", 6.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_bn_bwd_apply(dt, m->W(m->t2), m->W(c.v), m->Wf(c.mean), m->Wf(c.rstd), m->Wf(c.a), nullptr, m->Wf(m->Ecol), 0, m->Wf(m->Fc), m->W(m->t2), B, T, d, m->s));   // dv
        { RedScope red_scope_(m); CKP(m, "dwconv_bwd", 8.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_dwconv_bwd(dt, DWIN_GLU, m->W(m->t2), m->W(c.g), m->P(c.dw), m->W(m->t3), m->G(c.dw), c.dwb >= 0 ? m->G(c.dwb) : nullptr, red_scratch(m, dwconv_bwd_scratch_floats(d, c.k)), B, T, d, c.k, (c.k - 1) / 2, m->s)); }   // dg [M,2d]
    }
    EpiArgs e2; e2.resid = m->W(m->t1);
    CK(gemm_dgrad(m, c.Wp1, m->W(m->t3), dt, gn, r.M, OP_NONE, no, e2));
    CK(gemm_wgrad(m, c.Wp1, x, dt, OP_NONE, no, m->W(m->t3), dt, OP_NONE, no, r.M));
    return 0;
}

extern "C" int ishara_loss_backward(ishara_model* m, const float* logits, const int64_t* labels, int32_t B, float* loss, float* nll, float loss_scale, ishara_stream st) {
    if (!m->ws || !m->grads) { ishara_set_error("ishara_loss_backward: model is not bound (grads required)"); return -1; }
    if (m->family != ISHARA_FAMILY_KERAS_HYBRID) { ishara_set_error("ishara_loss_backward: this handle is an encoder-only family; use ishara_encoder_backward"); return -1; }
    if (B != m->lastB || !m->last_training) { ishara_set_error("ishara_loss_backward: call ishara_forward(training=1) with the same batch first"); return -1; }
    m->s = (hipStream_t)st;
    m->red.njobs = 0; m->red.nblocks = 0; m->red_off = 0; g_red_sink = nullptr;       // ... nor recorded column sums
    m->tn_defer.pending = false;        // a previous backward pass that returned early (error path) must not leave slab sums behind for this one to add
    Run r{B, B * m->T, 1, m->last_seed};
    const int dt = m->dt, d = m->d, T = m->T;
    OpArgs no; EpiArgs e0;
    float* nl = nll ? nll : m->Wf(m->nllb);
    CK(launch_fill_u32(m->grads, (size_t)m->n_train, 0u, m->s));      // a kernel, not a memset node: the whole step stays capturable (hipGraph note)
    CKP(m, "ctc", 2.0 * r.M * m->C * 4, 0, launch_ctc(logits, labels, B, T, m->C, m->L, m->C - 1, nl, m->Wf(m->dlogits), loss_scale / (float)B, m->Wf(m->ctcws), m->s, m->cls_pad ? m->W(m->dlb) : nullptr));
    if (loss) CKP(m, "mean", 0, 0, launch_mean(nl, loss, B, 1.f / (float)B, m->s));
    // ---- head
    // input of the head = output of the last layer
    const void* hin = m->W(m->stem_out);
    if (!m->layers.empty()) {
        const Layer& L = m->layers.back();
        hin = L.kind == Layer::CONV ? m->W(m->convs[L.idx].out) : (L.kind == Layer::SQZ ? m->W(m->sqz[L.idx].ffn2.out) : m->W(m->conf[L.idx].ffn2.out));
    }
    EpiArgs eh; eh.drop = dspec(r, m->head_site, m->cfg.head_dropout); eh.dact = DACT_POS; eh.aux = m->W(m->head_hh);
    if (m->cls_pad && r.M % 64 == 0 && r.M >= 256 && m->clsW.K % 128 == 0 && !g_force_tn_regstage) {
        DenseW wp = m->clsW; wp.N = m->cls_pad;          // reduction / output width of the padded operand; the real classes are the first m->C
        CK(gemm_dgrad(m, wp, m->W(m->dlb), dt, m->W(m->t1), r.M, OP_NONE, no, eh));
        CK(gemm_wgrad(m, wp, m->W(m->head_hh), dt, OP_NONE, no, m->W(m->dlb), dt, OP_NONE, no, r.M, 0, m->C));
    } else {
        CK(gemm_dgrad(m, m->clsW, m->Wf(m->dlogits), DT_F32, m->W(m->t1), r.M, OP_NONE, no, eh));
        CK(gemm_wgrad(m, m->clsW, m->W(m->head_hh), dt, OP_NONE, no, m->Wf(m->dlogits), DT_F32, OP_NONE, no, r.M));
    }
    void* g = m->W(m->gA); void* gn = m->W(m->gB);
    CK(gemm_dgrad(m, m->topW, m->W(m->t1), dt, g, r.M, OP_NONE, no, e0));
    CK(gemm_wgrad(m, m->topW, hin, dt, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M));
    // ---- layers in reverse; `in_of` = input activation of each module
    for (int li = (int)m->layers.size() - 1; li >= 0; --li) {
        const Layer& L = m->layers[li];
        const void* lin = m->W(m->stem_out);
        if (li > 0) {
            const Layer& Pv = m->layers[li - 1];
            lin = Pv.kind == Layer::CONV ? m->W(m->convs[Pv.idx].out) : (Pv.kind == Layer::SQZ ? m->W(m->sqz[Pv.idx].ffn2.out) : m->W(m->conf[Pv.idx].ffn2.out));
        }
#define STEP(call) do { CK(call); void* _t = g; g = gn; gn = _t; } while (0)
        if (L.kind == Layer::CONV) { STEP(conv_bwd(m, m->convs[L.idx], r, lin, g, gn)); }
        else if (L.kind == Layer::SQZ) {
            SqzBlock& sb = m->sqz[L.idx];
            STEP(ffn_bwd(m, sb.ffn2, r, m->W(sb.conv.out), g, gn));
            STEP(sqzconv_bwd(m, sb.conv, r, m->W(sb.mha.out), g, gn));
            STEP(mhsa_bwd(m, sb.mha, r, m->W(sb.ffn1.out), g, gn));
            STEP(ffn_bwd(m, sb.ffn1, r, lin, g, gn));
        } else {
            ConfBlock& cb = m->conf[L.idx];
            STEP(ffn_bwd(m, cb.ffn2, r, m->W(cb.conv.out), g, gn));
            STEP(confconv_bwd(m, cb.conv, r, m->W(cb.mha.out), g, gn));
            STEP(mhsa_bwd(m, cb.mha, r, m->W(cb.ffn1.out), g, gn));
            STEP(ffn_bwd(m, cb.ffn1, r, lin, g, gn));
        }
#undef STEP
        if (!m->bucket_ev.empty() && m->bucket_after_layer[li] >= 0) { CK(wgrad_flush(m)); CK(red_flush(m)); HIP_CHECK_RET(hipEventRecord(m->bucket_ev[m->bucket_after_layer[li]], m->s)); }
    }
    // ---- stem
    CKP(m, "sample_reduce", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_sample_reduce(dt, g, m->W(m->stem_h0), m->Wf(m->stem_mean), m->Wf(m->stem_rstd), m->Wf(m->S1), m->Wf(m->S2), B, T, d, m->s));
    CKP(m, "bn_bwd_finalize", 0, 0, launch_bn_bwd_finalize(m->Wf(m->S1), m->Wf(m->S2), m->G(m->stem_bn.gamma), m->G(m->stem_bn.beta), m->Wf(m->Ecol), m->Wf(m->Fc), B, T, d, m->s));
    CKP(m, "bn_bwd_apply", 6.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_bn_bwd_apply(dt, g, m->W(m->stem_h0), m->Wf(m->stem_mean), m->Wf(m->stem_rstd), m->Wf(m->stem_a), nullptr, m->Wf(m->Ecol), 0, m->Wf(m->Fc), m->W(m->t1), B, T, d, m->s));
    if (m->stem_kp && r.M % 64 == 0 && r.M >= 256 && d % 128 == 0 && !g_force_tn_regstage) {
        DenseW wp = m->stemW; wp.K = m->stem_kp;           // packed rows of the forward pass; only the first F rows of dW exist
        CK(gemm_wgrad(m, wp, m->W(m->stem_xb), dt, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M, m->F));
    } else
        CK(gemm_wgrad(m, m->stemW, m->last_x, DT_F32, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M));
    CK(wgrad_flush(m));
    CK(red_flush(m));
    if (!m->bucket_ev.empty()) HIP_CHECK_RET(hipEventRecord(m->bucket_ev.back(), m->s));
    return 0;
}

// ---- gradient buckets (overlapping the data-parallel all-reduce with the backward pass)
extern "C" int32_t ishara_grad_buckets(const ishara_model* m) { return (int32_t)m->bucket_lo.size(); }
extern "C" int ishara_grad_bucket(const ishara_model* m, int32_t i, int64_t* offset, int64_t* count) {
    if (i < 0 || i >= (int32_t)m->bucket_lo.size()) { ishara_set_error("ishara_grad_bucket: index %d outside 0..%d", i, (int)m->bucket_lo.size() - 1); return -1; }
    *offset = m->bucket_lo[i]; *count = m->bucket_hi[i] - m->bucket_lo[i];
    return 0;
}
// makes `side` wait until bucket i of the LAST ishara_loss_backward is final (events are created on first use)
extern "C" int ishara_grad_bucket_wait(ishara_model* m, int32_t i, ishara_stream side) {
    if (i < 0 || i >= (int32_t)m->bucket_lo.size()) { ishara_set_error("ishara_grad_bucket_wait: index %d outside 0..%d", i, (int)m->bucket_lo.size() - 1); return -1; }
    if (m->bucket_ev.empty()) { ishara_set_error("ishara_grad_bucket_wait: call ishara_grad_buckets_enable before the backward pass"); return -1; }
    HIP_CHECK_RET(hipStreamWaitEvent((hipStream_t)side, m->bucket_ev[i], 0));
    return 0;
}
extern "C" int ishara_grad_buckets_enable(ishara_model* m) {
    if (!m->bucket_ev.empty()) return 0;
    m->bucket_ev.resize(m->bucket_lo.size());
    for (auto& e : m->bucket_ev) HIP_CHECK_RET(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}

// ------------------------------------------------------------------ optimizer
extern "C" int ishara_optimizer_step(ishara_model* m, float lr, float weight_decay, ishara_stream st) {
    if (!m->om || !m->ov || !m->oslow || !m->grads) { ishara_set_error("ishara_optimizer_step: optimizer slots are not bound"); return -1; }
    hipStream_t s = (hipStream_t)st;
    m->opt_iter += 1;
    const double t = (double)m->opt_iter, b1 = 0.9, b2 = 0.999;
    const double b1p = pow(b1, t), b2p = pow(b2, t);
    const double sma_inf = 2.0 / (1.0 - b2) - 1.0;
    const double sma_t = sma_inf - 2.0 * t * b2p / (1.0 - b2p);
    RAdamArgs a;
    a.lr = lr; a.wd = weight_decay; a.beta1 = (float)b1; a.beta2 = (float)b2; a.eps = 1e-7f;
    a.c1 = (float)(1.0 / (1.0 - b1p)); a.c2 = (float)(1.0 / (1.0 - b2p));
    a.rect = sma_t >= 4.0 ? 1 : 0;       // sma_threshold = 4 (c7:68)
    a.r_t = a.rect ? (float)sqrt(fmax((sma_t - 4.0) / (sma_inf - 4.0) * (sma_t - 2.0) / (sma_inf - 2.0) * sma_inf / sma_t, 0.0)) : 0.f;
    a.sync = (m->opt_iter % 5 == 0) ? 1 : 0;   // Lookahead(sync_period=5, slow_step_size=0.5) (c7:69)
    a.slow_step = 0.5f;
    m->s = s;
    CKP(m, "radam_lookahead", 28.0 * m->n_train, 0, launch_radam_lookahead(m->params, m->grads, m->om, m->ov, m->oslow, m->n_train, a, s));
    CKP(m, "weight_shadows", 0, 0, ishara_sync_weights(m, st));
    return 0;
}
extern "C" int32_t ishara_optimizer_iterations(const ishara_model* m) { return m->opt_iter; }
extern "C" int ishara_optimizer_set_iterations(ishara_model* m, int32_t it) { m->opt_iter = it; return 0; }

// ------------------------------------------------------------------ profiler API
extern "C" int ishara_profile_enable(ishara_model* m, int32_t on) {
    m->prof.on = on != 0;
    m->prof.recs.clear();
    m->prof.used = 0;
    return 0;
}
// one text line per kernel family: "key count total_ms bytes flops\n"; returns bytes written (or <0)
extern "C" int ishara_profile_report(ishara_model* m, char* buf, int32_t cap) {
    struct Agg { int n = 0; double ms = 0, by = 0, fl = 0; };
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    for (auto& r : m->prof.recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) { ishara_set_error("profile: event sync failed"); return -2; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (!agg.count(r.key)) order.push_back(r.key);
        Agg& a = agg[r.key];
        a.n++; a.ms += ms; a.by += r.bytes; a.fl += r.flops;
    }
    int pos = 0;
    for (auto& k : order) {
        const Agg& a = agg[k];
        const int w = snprintf(buf + pos, cap > pos ? cap - pos : 0, "%s %d %.6f %.0f %.0f\n", k.c_str(), a.n, a.ms, a.by, a.fl);
        if (w < 0 || pos + w >= cap) { ishara_set_error("profile: buffer too small"); return -1; }
        pos += w;
    }
    m->prof.recs.clear();
    m->prof.used = 0;
    return pos;
}

// ------------------------------------------------------------------ stand-alone entry points
extern "C" int ishara_greedy_decode(const float* logits, int32_t B, int32_t T, int32_t C, int32_t blank, int32_t* out_idx, int32_t* out_len, ishara_stream s) {
    if (T > 4096) { ishara_set_error("greedy_decode: T too large"); return -1; }
    return launch_greedy_decode(logits, B, T, C, blank, out_idx, out_len, (hipStream_t)s);
}
extern "C" int64_t ishara_ctc_workspace_bytes(int32_t B, int32_t T, int32_t L) { return (int64_t)(ctc_workspace_floats(B, T, L) * sizeof(float)); }
extern "C" int ishara_ctc_loss(const float* logits, const int64_t* labels, int32_t B, int32_t T, int32_t C, int32_t L, int32_t blank,
                               float* nll, float* dlogits, float grad_scale, void* ws, ishara_stream s) {
    return launch_ctc(logits, labels, B, T, C, L, blank, nll, dlogits, grad_scale, (float*)ws, (hipStream_t)s);
}
extern "C" int ishara_dropout_mask(uint32_t seed, uint32_t site, int32_t rows, int32_t cols, float rate, float* out, ishara_stream s) {
    const DropSpec d = make_drop(seed, site, rate, true);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(1024), dim3(256), 0, (hipStream_t)s, out, rows, cols, d);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern int g_force_regstage, g_dbg_tn, g_force_tn_regstage, g_force_dw_lds, g_tn_blocks, g_attn_bwd_two_pass;
static int g_dbg_epi = 0;
// bit 0: 1 register-staged NT kernel / 0 LDS-DMA NT kernel; bit 1: 1 register-transposing TN kernel; bit 2: 1 LDS-tiled dwconv; bit 3: 1 LDS-DMA 64x128 NT kernel; bits 4-7: NT ablation; bits 8-12: TN ablation; bit 13: 1 tile NT kernel instead of the A-stationary one; bits 14-15: wgrad workgroups auto / 256 / 512 / 768; bit 16: 1 two-kernel attention backward instead of the one-pass kernel
extern int g_as_flags_override;
// A-stationary GEMM switches (gemm_as.hip as_default_flags: 1 paired half-line stores, 2 non-temporal side outputs, 16 chunked K = 256 form); -1: library default
extern "C" int ishara_debug_set_as_flags(int32_t flags) { g_as_flags_override = flags; return 0; }
extern int g_nt_big;
extern "C" int ishara_debug_set_nt_big(int32_t on) { g_nt_big = on; return 0; }
extern "C" int ishara_debug_force_regstage(int32_t on) { g_force_regstage = (on & 1) ? 1 : ((on >> 3) & 1 ? 2 : ((on >> 13) & 1 ? 3 : 0)); g_dbg_epi = (on >> 4) & 15; g_dbg_tn = (on >> 8) & 31; g_force_tn_regstage = (on >> 1) & 1; g_force_dw_lds = (on >> 2) & 1; { const int tb = (on >> 14) & 3; g_tn_blocks = tb == 1 ? 256 : (tb == 2 ? 512 : (tb == 3 ? 768 : 0)); } g_attn_bwd_two_pass = (on >> 16) & 1; return 0; }

extern "C" int ishara_preprocess(const float* raw, const int32_t* n_frames, int32_t max_frames, const float* mean, const float* stdv,
                                 float* out, int32_t T, ishara_stream s) {
    if (max_frames <= 0 || max_frames > 8192) { ishara_set_error("ishara_preprocess: max_frames %d unsupported (1..8192)", max_frames); return -1; }
    return launch_preprocess(raw, n_frames, max_frames, mean, stdv, out, T, (hipStream_t)s);
}

// ---- operator tests: dense
static void op_shadow_layout(int dt, int K, int N, size_t& wt, int& ldt, size_t& wn, int& ldn, size_t& slab, size_t& total, int M) {
    const int bk = dt == DT_BF16 ? 64 : 32;
    const size_t es = dt_size(dt);
    ldt = (int)rup(K, bk); ldn = (int)rup(N, bk);
    wt = 0;
    wn = rup(rup(N, 128) * (size_t)ldt * es, 256);
    slab = wn + rup(rup(K, 128) * (size_t)ldn * es, 256);
    total = slab + gemm_tn_slab_floats(M, K, N, dt) * sizeof(float);
}
extern "C" int64_t ishara_op_scratch_bytes(int32_t M, int32_t K, int32_t N) {
    size_t wt, wn, slab, total; int ldt, ldn;
    op_shadow_layout(DT_F32, K, N, wt, ldt, wn, ldn, slab, total, M);
    return (int64_t)total;
}
extern "C" int ishara_op_dense_fwd(int32_t dt, const void* x, const float* Wm, const float* bias, void* y, int32_t M, int32_t K, int32_t N, int32_t act, void* scratch, ishara_stream st) {
    hipStream_t s = (hipStream_t)st;
    size_t wt, wn, slab, total; int ldt, ldn;
    op_shadow_layout(dt, K, N, wt, ldt, wn, ldn, slab, total, M);
    char* sc = (char*)scratch;
    HIP_CHECK_RET(hipMemsetAsync(sc, 0, slab, s));
    CK(launch_make_shadow(dt, Wm, K, N, sc + wt, ldt, sc + wn, ldn, s));
    OpArgs no; EpiArgs ea; ea.bias = bias; ea.act = act; ea.dbg = g_dbg_epi;
    return launch_gemm_nt(dt, dt, dt, OP_NONE, x, sc + wt, y, M, N, K, ldt, no, ea, s);
}
// y = act(x @ W + b) + resid
extern "C" int ishara_op_dense_fwd_ex(int32_t dt, const void* x, const float* Wm, const float* bias, const void* resid, void* y,
                                      int32_t M, int32_t K, int32_t N, int32_t act, void* scratch, ishara_stream st) {
    hipStream_t s = (hipStream_t)st;
    size_t wt, wn, slab, total; int ldt, ldn;
    op_shadow_layout(dt, K, N, wt, ldt, wn, ldn, slab, total, M);
    char* sc = (char*)scratch;
    HIP_CHECK_RET(hipMemsetAsync(sc, 0, slab, s));
    CK(launch_make_shadow(dt, Wm, K, N, sc + wt, ldt, sc + wn, ldn, s));
    OpArgs no; EpiArgs ea; ea.bias = bias; ea.act = act; ea.resid = resid; ea.dbg = g_dbg_epi;
    return launch_gemm_nt(dt, dt, dt, OP_NONE, x, sc + wt, y, M, N, K, ldt, no, ea, s);
}
extern "C" int ishara_op_dense_bwd(int32_t dt, const void* x, const float* Wm, const void* dy, void* dx, float* dW, float* db,
                                   int32_t M, int32_t K, int32_t N, void* scratch, ishara_stream st) {
    hipStream_t s = (hipStream_t)st;
    size_t wt, wn, slab, total; int ldt, ldn;
    op_shadow_layout(dt, K, N, wt, ldt, wn, ldn, slab, total, M);
    char* sc = (char*)scratch;
    HIP_CHECK_RET(hipMemsetAsync(sc, 0, slab, s));
    CK(launch_make_shadow(dt, Wm, K, N, sc + wt, ldt, sc + wn, ldn, s));
    OpArgs no; EpiArgs ea;
    if (dx) CK(launch_gemm_nt(dt, dt, dt, OP_NONE, dy, sc + wn, dx, M, K, N, ldn, no, ea, s));
    return launch_gemm_tn(dt, dt, dt, OP_NONE, OP_NONE, x, dy, dW, db, (float*)(sc + slab), M, K, N, no, no, s);
}
// row log-softmax over fp32 logits and its backward: the output layer of the torch Squeezeformer (squeezeformer/model.py:448-449)
extern "C" int ishara_op_log_softmax_fwd(const float* x, float* y, int32_t M, int32_t C, int32_t ld, ishara_stream s) { return launch_log_softmax_fwd(x, y, M, C, ld, (hipStream_t)s); }
extern "C" int ishara_op_log_softmax_bwd(const float* dy, const float* y, float* dx, int32_t M, int32_t C, int32_t ld, ishara_stream s) { return launch_log_softmax_bwd(dy, y, dx, M, C, ld, (hipStream_t)s); }
extern "C" int ishara_op_layernorm_fwd(int32_t dt, const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd, int32_t M, int32_t C, ishara_stream s) {
    return launch_layernorm_fwd(dt, x, gamma, beta, eps, y, mean, rstd, M, C, (hipStream_t)s);
}
extern "C" int ishara_op_layernorm_bwd(int32_t dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx, float* dgamma, float* dbeta, int32_t M, int32_t C, ishara_stream s) {
    return launch_layernorm_bwd(dt, dy, x, mean, rstd, gamma, nullptr, dx, dgamma, dbeta, nullptr, M, C, (hipStream_t)s);
}
extern "C" int ishara_op_dwconv_fwd(int32_t dt, int32_t inop, const void* x, const float* w, const float* bias, void* y, float* ssum, float* ssq,
                                    int32_t B, int32_t T, int32_t C, int32_t k, int32_t padl, ishara_stream s) {
    return launch_dwconv_fwd(dt, inop, x, w, bias, y, ssum, ssq, nullptr, B, T, C, k, padl, (hipStream_t)s);
}
// the same with caller scratch for the deterministic statistics (partial rows summed in a fixed order): the path the model takes, and the
// only one that reaches the streaming K = 11 / 15 kernel at B > 8
extern "C" int64_t ishara_op_dwconv_fwd_scratch_bytes(int32_t B, int32_t T, int32_t C) { return (int64_t)(dwconv_fwd_scratch_floats(B, T, C) * sizeof(float)); }
extern "C" int ishara_op_dwconv_fwd_ex(int32_t dt, int32_t inop, const void* x, const float* w, const float* bias, void* y, float* ssum, float* ssq,
                                       void* scratch, int32_t B, int32_t T, int32_t C, int32_t k, int32_t padl, ishara_stream s) {
    return launch_dwconv_fwd(dt, inop, x, w, bias, y, ssum, ssq, (float*)scratch, B, T, C, k, padl, (hipStream_t)s);
}
extern "C" int64_t ishara_op_dwconv_scratch_bytes(int32_t C, int32_t k) { return (int64_t)(dwconv_bwd_scratch_floats(C, k) * sizeof(float)); }
extern "C" int ishara_op_dwconv_bwd(int32_t dt, int32_t inop, const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias,
                                    void* scratch, int32_t B, int32_t T, int32_t C, int32_t k, int32_t padl, ishara_stream s) {
    return launch_dwconv_bwd(dt, inop, dy, x, w, dx, dw, dbias, (float*)scratch, B, T, C, k, padl, (hipStream_t)s);
}
// scratch layout: q | k | vt | lse | delta
extern "C" int64_t ishara_op_attn_scratch_bytes(int32_t B, int32_t H, int32_t T, int32_t dh) {
    const size_t n = (size_t)B * H * T * dh;
    return (int64_t)(3 * rup(n * 4, 256) + 2 * rup((size_t)B * H * T * 4, 256) + rup(attn_mask_words(B, H, T) * 4, 256));
}
static void attn_scratch(char* sc, int dt, int B, int H, int T, int dh, void*& q, void*& k, void*& vt, float*& lse, float*& delta, uint32_t*& maskw) {
    const size_t n = (size_t)B * H * T * dh;
    const size_t seg = rup(n * 4, 256);
    (void)dt;
    q = sc; k = sc + seg; vt = sc + 2 * seg;
    lse = (float*)(sc + 3 * seg);
    delta = (float*)(sc + 3 * seg + rup((size_t)B * H * T * 4, 256));
    maskw = (uint32_t*)(sc + 3 * seg + 2 * rup((size_t)B * H * T * 4, 256));
}
extern "C" int ishara_op_attn_fwd(int32_t dt, const void* qkv, void* o, int32_t B, int32_t H, int32_t T, int32_t dh, float scale,
                                  uint32_t seed, uint32_t site, float rate, int32_t impl, void* scratch, ishara_stream st) {
    hipStream_t s = (hipStream_t)st;
    void *q, *k, *vt; float *lse, *delta; uint32_t* maskw;
    attn_scratch((char*)scratch, dt, B, H, T, dh, q, k, vt, lse, delta, maskw);
    if (dt == DT_BF16) hipLaunchKernelGGL(qkv_split_kernel<bf16>, dim3(1024), dim3(256), 0, s, (const bf16*)qkv, (bf16*)q, (bf16*)k, (bf16*)vt, B, H, T, dh);
    else hipLaunchKernelGGL(qkv_split_kernel<float>, dim3(1024), dim3(256), 0, s, (const float*)qkv, (float*)q, (float*)k, (float*)vt, B, H, T, dh);
    return launch_attn_fwd(dt, q, k, vt, o, lse, B, H, T, dh, scale, make_drop_attn(seed, site, rate, true), impl, maskw, s);
}
extern "C" int ishara_op_attn_bwd(int32_t dt, const void* o, const void* dout, void* dqkv, int32_t B, int32_t H, int32_t T, int32_t dh, float scale,
                                  uint32_t seed, uint32_t site, float rate, int32_t impl, void* scratch, ishara_stream st) {
    hipStream_t s = (hipStream_t)st;
    void *q, *k, *vt; float *lse, *delta; uint32_t* maskw;
    attn_scratch((char*)scratch, dt, B, H, T, dh, q, k, vt, lse, delta, maskw);
    return launch_attn_bwd(dt, q, k, vt, o, dout, lse, delta, dqkv, B, H, T, dh, scale, make_drop_attn(seed, site, rate, true), 1, impl, maskw, s);
}
