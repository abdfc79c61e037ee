// The torch ConformerEncoder family of the reference (conformer/conformer.py:6-87) on the same kernels as the
// Keras hybrid: post-LN sub-modules, each returning LayerNorm(sub(x) + x):
//   FeedForwardModule (:6-22)        Linear -> SiLU -> Dropout -> Linear -> Dropout -> +x -> LayerNorm
//   MultiHeadSelfAttention (:24-35)  nn.MultiheadAttention (packed in_proj with bias, scale dh^-0.5, dropout on the probabilities,
//                                    out_proj with bias) -> +x -> LayerNorm
//   ConvolutionModule (:37-57)       Conv1d(d,2d,1) -> GLU(dim=1) -> depthwise Conv1d(k, pad k//2, bias) -> BatchNorm1d -> Conv1d(d,d,1)
//                                    -> +x -> LayerNorm        (no activation after BN, its Dropout is never applied)
//   ConformerBlock (:59-73)          ffn1 -> attention -> conv -> ffn2 -> LayerNorm
// Parameter entries carry the reference's state_dict keys, in state_dict order; the arrays are stored in the kernels' layout
// (Linear / pointwise weights [in,out]; in_proj columns head-major h*3dh + {q,k,v}*dh + i; depthwise weight [k,d]) — the host
// binding (ishara_amd/conformer.py) converts to and from the torch layouts.
#include "model_types.h"

// ------------------------------------------------------------------ f32 <-> storage dtype
template <typename T>
__global__ void r5_from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float v[8];
        load8(x + i * 8, v);
        store8(y + i * 8, v);
    }
}
template <typename T>
__global__ void r5_to_f32_kernel(const T* __restrict__ x, float* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float v[8];
        load8(x + i * 8, v);
        store8(y + i * 8, v);
    }
}
static int grid_for(size_t n8) { const size_t g = (n8 + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }
int r5_from_f32(int dt, const float* x, void* y, size_t n, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL(r5_from_f32_kernel<bf16>, dim3(grid_for(n / 8)), dim3(256), 0, s, x, (bf16*)y, n / 8);
    else hipLaunchKernelGGL(r5_from_f32_kernel<float>, dim3(grid_for(n / 8)), dim3(256), 0, s, x, (float*)y, n / 8);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int r5_to_f32(int dt, const void* x, float* y, size_t n, hipStream_t s) {
    if (dt == DT_BF16) hipLaunchKernelGGL(r5_to_f32_kernel<bf16>, dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16*)x, y, n / 8);
    else hipLaunchKernelGGL(r5_to_f32_kernel<float>, dim3(grid_for(n / 8)), dim3(256), 0, s, (const float*)x, y, n / 8);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

static const float R5_EPS = 1e-5f;      // nn.LayerNorm / nn.BatchNorm1d default eps

// ------------------------------------------------------------------ construction
int r5_validate(const ishara_config& c) {
    if (c.num_conv_conform_blocks <= 0) { ishara_set_error("ConformerEncoder: num_layers (num_conv_conform_blocks) must be > 0"); return -1; }
    if (c.expansion_factor <= 0) { ishara_set_error("ConformerEncoder: expansion_factor must be > 0"); return -1; }
    return 0;
}

static DenseW dense_named(ishara_model* m, const std::string& wname, const std::string& bname, int K, int N) {
    DenseW w; w.K = K; w.N = N;
    w.w = m->addp(wname, K, N, true);
    w.b = m->addp(bname, N, 0, true);
    return w;
}
static Norm norm_named(ishara_model* m, const std::string& p, int c) {
    Norm n; n.gamma = m->addp(p + ".weight", c, 0, true); n.beta = m->addp(p + ".bias", c, 0, true);
    return n;
}
static R5FFN r5_build_ffn(ishara_model* m, const std::string& p) {
    R5FFN f;
    const int d = m->d, e = m->cfg.expansion_factor;
    f.W1 = dense_named(m, p + ".linear1.weight", p + ".linear1.bias", d, d * e);
    f.W2 = dense_named(m, p + ".linear2.weight", p + ".linear2.bias", d * e, d);
    f.ln = norm_named(m, p + ".layer_norm", d);
    f.site_in = m->nsites++; f.site_out = m->nsites++;
    return f;
}

void r5_build_graph(ishara_model* m) {
    const int d = m->d, k = m->cfg.transformer_kernel_size;
    for (int i = 0; i < m->cfg.num_conv_conform_blocks; ++i) {
        const std::string p = "layers." + std::to_string(i);
        R5Block b;
        b.ffn1 = r5_build_ffn(m, p + ".ffn1");
        b.mha.Wqkv = dense_named(m, p + ".attention.attention.in_proj_weight", p + ".attention.attention.in_proj_bias", d, 3 * d);
        b.mha.Wp = dense_named(m, p + ".attention.attention.out_proj.weight", p + ".attention.attention.out_proj.bias", d, d);
        b.mha.ln = norm_named(m, p + ".attention.layer_norm", d);
        b.mha.site_attn = m->nsites++;
        ConfConv& c = b.conv;
        c.k = k; c.bn_eps = R5_EPS; c.ln_eps = R5_EPS; c.bn_keep = 0.9f; c.bn_unbiased = 1;      // nn.BatchNorm1d(momentum=0.1): new = 0.9 old + 0.1 batch
        c.Wp1 = dense_named(m, p + ".conv.pointwise_conv1.weight", p + ".conv.pointwise_conv1.bias", d, 2 * d);
        c.dw = m->addp(p + ".conv.depthwise_conv.weight", k, d, true);
        c.dwb = m->addp(p + ".conv.depthwise_conv.bias", d, 0, true);
        c.bn.gamma = m->addp(p + ".conv.batch_norm.weight", d, 0, true);
        c.bn.beta = m->addp(p + ".conv.batch_norm.bias", d, 0, true);
        c.bn.mm = m->addp(p + ".conv.batch_norm.running_mean", d, 0, false);
        c.bn.mv = m->addp(p + ".conv.batch_norm.running_var", d, 0, false);
        c.Wp2 = dense_named(m, p + ".conv.pointwise_conv2.weight", p + ".conv.pointwise_conv2.bias", d, d);
        c.ln = norm_named(m, p + ".conv.layer_norm", d);
        b.ffn2 = r5_build_ffn(m, p + ".ffn2");
        b.ln = norm_named(m, p + ".layer_norm", d);
        m->r5.push_back(b);
        m->layer_entry_end.push_back(m->entries.size());
    }
    int64_t off = 0;
    for (auto& e : m->entries) if (e.trainable) { e.offset = off; off += e.shape[0] * (e.ndim == 2 ? e.shape[1] : 1); }
    m->n_train = off;
    for (auto& e : m->entries) if (!e.trainable) { e.offset = off; off += e.shape[0] * (e.ndim == 2 ? e.shape[1] : 1); }
    m->n_total = off;
    m->bucket_lo.push_back(0); m->bucket_hi.push_back(m->n_train);      // one gradient range, final when the backward pass ends
    m->bucket_after_layer.assign(m->r5.size(), -1);
}

void r5_plan_workspace(ishara_model* m) {
    const int d = m->d, B = m->Bmax, T = m->T, de = d * m->cfg.expansion_factor;
    const size_t Mx = (size_t)B * T;
    m->cur = 0;
    m->shadow_begin = m->cur;
    for (auto& b : m->r5)
        for (DenseW* w : {&b.ffn1.W1, &b.ffn1.W2, &b.mha.Wqkv, &b.mha.Wp, &b.conv.Wp1, &b.conv.Wp2, &b.ffn2.W1, &b.ffn2.W2}) plan_shadow(m, *w);
    m->shadow_end = m->cur;
    m->shadow_tab_off = m->alloc(m->denses.size() * sizeof(ShadowDesc)).off;
    m->r5_x = m->act(d);
    auto plan_ffn = [&](R5FFN& f) {
        f.za = m->act(de); f.u = m->act(de); f.r = m->act(d); f.mean = m->f32(Mx); f.rstd = m->f32(Mx); f.out = m->act(d);
    };
    for (auto& b : m->r5) {
        plan_ffn(b.ffn1);
        R5MHSA& a = b.mha;
        a.q = m->act(d); a.k = m->act(d); a.vt = m->act(d); a.o = m->act(d);
        a.lse = m->f32((size_t)B * m->H * T); a.maskw = m->f32(attn_mask_words(B, m->H, T));
        a.r = m->act(d); a.mean = m->f32(Mx); a.rstd = m->f32(Mx); a.out = m->act(d);
        ConfConv& c = b.conv;
        c.g = m->act(2 * d); c.v = m->act(d); c.bnv = m->act(d);
        c.ssum = m->f32((size_t)B * d); c.ssq = m->f32((size_t)B * d);
        c.mean = m->f32(d); c.rstd = m->f32(d); c.a = m->f32(d); c.bsh = m->f32(d);
        c.r = m->act(d); c.lnmean = m->f32(Mx); c.lnrstd = m->f32(Mx); c.out = m->act(d);
        plan_ffn(b.ffn2);
        b.mean = m->f32(Mx); b.rstd = m->f32(Mx); b.out = m->act(d);
    }
    const int maxw = de > 3 * d ? de : 3 * d;
    m->gA = m->act(d); m->gB = m->act(d); m->t4 = m->act(d);
    m->t1 = m->act(maxw); m->t2 = m->act(maxw); m->t3 = m->act(maxw);
    m->S1 = m->f32((size_t)B * maxw); m->S2 = m->f32((size_t)B * maxw); m->E = m->f32((size_t)B * maxw);
    m->Fc = m->f32(maxw); m->Ecol = m->f32(maxw);
    size_t slabf = 0;
    for (DenseW* w : m->denses) { const size_t f = gemm_tn_slab_floats((int)Mx, w->K, w->N, m->dt); if (f > slabf) slabf = f; }
    if (layernorm_bwd_scratch_floats(d) > slabf) slabf = layernorm_bwd_scratch_floats(d);
    if (dwconv_bwd_scratch_floats(2 * maxw, 31) > slabf) slabf = dwconv_bwd_scratch_floats(2 * maxw, 31);
    if (dwconv_fwd_scratch_floats(B, T, 2 * maxw) > slabf) slabf = dwconv_fwd_scratch_floats(B, T, 2 * maxw);
    m->slab = m->f32(slabf);
    m->delta = m->f32((size_t)B * m->H * T);
    m->ws_need = m->cur;
}

// ------------------------------------------------------------------ forward
int r5_ln_fwd(ishara_model* m, const Run& r, const void* x, const Norm& n, void* y, Buf mean, Buf rstd) {
    CKP(m, "layernorm_fwd", 2.0 * r.M * m->d * (double)dt_size(m->dt), 0, launch_layernorm_fwd(m->dt, x, m->P(n.gamma), m->P(n.beta), R5_EPS, y, m->Wf(mean), m->Wf(rstd), r.M, m->d, m->s));
    return 0;
}
int r5_ffn_fwd(ishara_model* m, R5FFN& f, const Run& r, const void* x) {
    const int dt = m->dt;
    OpArgs no;
    EpiArgs ea; ea.pre_out = m->W(f.za); ea.act = ACT_SWISH; ea.drop = dspec(r, f.site_in, m->cfg.dropout_rate);
    CK(gemm_fwd(m, f.W1, x, dt, m->W(f.u), dt, r.M, OP_NONE, no, ea));
    EpiArgs eb; eb.resid = x; eb.drop = dspec(r, f.site_out, m->cfg.dropout_rate);
    if (f.factor != 1.f) { eb.rowscale = m->Wf(m->fac); eb.T = m->T; }        // r = x + factor * drop(linear2(...)): fac[b] = factor for every sample
    CK(gemm_fwd(m, f.W2, m->W(f.u), dt, m->W(f.r), dt, r.M, OP_NONE, no, eb));
    return r5_ln_fwd(m, r, m->W(f.r), f.ln, m->W(f.out), f.mean, f.rstd);
}
static int r5_mhsa_fwd(ishara_model* m, R5MHSA& a, const Run& r, const void* x) {
    const int dt = m->dt;
    OpArgs no;
    EpiArgs eq; eq.mode = EPI_QKV; eq.q = m->W(a.q); eq.k = m->W(a.k); eq.vt = m->W(a.vt); eq.H = m->H; eq.dh = m->dh; eq.T = m->T; eq.head_major = 1;
    CK(gemm_fwd(m, a.Wqkv, x, dt, nullptr, dt, r.M, OP_NONE, no, eq));
    const float scale = 1.0f / sqrtf((float)m->dh);      // nn.MultiheadAttention: q * head_dim ** -0.5
    CKP(m, "attn_fwd", 4.0 * r.M * m->d * (double)dt_size(dt), 4.0 * r.B * m->H * (double)m->T * m->T * m->dh,
        launch_attn_fwd(dt, m->W(a.q), m->W(a.k), m->W(a.vt), m->W(a.o), m->Wf(a.lse), r.B, m->H, m->T, m->dh, scale, dspec_attn(r, a.site_attn, m->cfg.dropout_rate),
                        m->cfg.attn_impl, reinterpret_cast<uint32_t*>(m->W(a.maskw)), m->s));
    EpiArgs ep; ep.resid = x;
    CK(gemm_fwd(m, a.Wp, m->W(a.o), dt, m->W(a.r), dt, r.M, OP_NONE, no, ep));
    return r5_ln_fwd(m, r, m->W(a.r), a.ln, m->W(a.out), a.mean, a.rstd);
}

extern "C" int ishara_encoder_forward(ishara_model* m, const float* x, int32_t B, float* y, int32_t training, uint32_t seed, ishara_stream st) {
    if (!m->ws) { ishara_set_error("ishara_encoder_forward: model is not bound"); return -1; }
    if (B <= 0 || B > m->Bmax) { ishara_set_error("ishara_encoder_forward: batch %d outside 1..%d", B, m->Bmax); return -1; }
    if (m->family == ISHARA_FAMILY_TORCH_SQUEEZEFORMER) return r4_forward(m, x, B, y, training, seed, (hipStream_t)st);
    if (m->family != ISHARA_FAMILY_TORCH_CONFORMER) { ishara_set_error("ishara_encoder_forward: handle is not an encoder-only family"); return -1; }
    m->s = (hipStream_t)st;
    Run r{B, B * m->T, training, seed};
    const void* h = x;
    if (m->dt != DT_F32) { CKP(m, "cast", 6.0 * r.M * m->d, 0, r5_from_f32(m->dt, x, m->W(m->r5_x), (size_t)r.M * m->d, m->s)); h = m->W(m->r5_x); }
    for (R5Block& b : m->r5) {
        CK(r5_ffn_fwd(m, b.ffn1, r, h));
        CK(r5_mhsa_fwd(m, b.mha, r, m->W(b.ffn1.out)));
        CK(confconv_fwd(m, b.conv, r, m->W(b.mha.out)));
        CK(r5_ffn_fwd(m, b.ffn2, r, m->W(b.conv.out)));
        CK(r5_ln_fwd(m, r, m->W(b.ffn2.out), b.ln, m->W(b.out), b.mean, b.rstd));
        h = m->W(b.out);
    }
    CKP(m, "cast", 6.0 * r.M * m->d, 0, r5_to_f32(m->dt, h, y, (size_t)r.M * m->d, m->s));
    m->lastB = B; m->last_training = training; m->last_seed = seed; m->last_x = x;
    return 0;
}

// ------------------------------------------------------------------ backward
// each *_bwd consumes g (gradient of the module output) and writes gn (gradient of its input x)
int r5_ln_bwd(ishara_model* m, const Run& r, const void* g, const void* x, const Norm& n, Buf mean, Buf rstd, void* dx) {
    CKP(m, "layernorm_bwd", 4.0 * r.M * m->d * (double)dt_size(m->dt), 0,
        launch_layernorm_bwd(m->dt, g, x, m->Wf(mean), m->Wf(rstd), m->P(n.gamma), nullptr, dx, m->G(n.gamma), m->G(n.beta), m->Wf(m->slab), r.M, m->d, m->s));
    return 0;
}
int r5_ffn_bwd(ishara_model* m, R5FFN& f, const Run& r, const void* x, const void* g, void* gn) {
    const int dt = m->dt;
    OpArgs no;
    void* dr = m->W(m->t4);                                         // gradient of r = x + drop(linear2(...)): also the residual branch
    CK(r5_ln_bwd(m, r, g, m->W(f.r), f.ln, f.mean, f.rstd, dr));
    const void* gs = dr;
    const DropSpec od = dspec(r, f.site_out, m->cfg.dropout_rate);
    if (od.thr) {
        CKP(m, "map_rows", 2.0 * r.M * m->d * (double)dt_size(dt), 0, launch_map_rows(dt, MAP_DROPMASK, dr, m->W(m->t3), nullptr, od, r.M, m->T, m->d, m->s));
        gs = m->W(m->t3);
    }
    if (f.factor != 1.f) {                                        // gradient of the branch = factor * dr
        CKP(m, "map_rows", 2.0 * r.M * m->d * (double)dt_size(dt), 0, launch_map_rows(dt, MAP_ROWSCALE, gs, m->W(m->t3), m->Wf(m->fac), DropSpec{0, 0, 1.f}, r.M, m->T, m->d, m->s));
        gs = m->W(m->t3);
    }
    EpiArgs e1; e1.drop = dspec(r, f.site_in, m->cfg.dropout_rate); e1.dact = DACT_SWISH; e1.aux = m->W(f.za);
    CK(gemm_dgrad(m, f.W2, gs, dt, m->W(m->t1), r.M, OP_NONE, no, e1));                         // d za
    CK(gemm_wgrad(m, f.W2, m->W(f.u), dt, OP_NONE, no, gs, dt, OP_NONE, no, r.M));
    EpiArgs e0; e0.resid = dr;
    CK(gemm_dgrad(m, f.W1, m->W(m->t1), dt, gn, r.M, OP_NONE, no, e0));
    CK(gemm_wgrad(m, f.W1, x, dt, OP_NONE, no, m->W(m->t1), dt, OP_NONE, no, r.M));
    return 0;
}
static int r5_mhsa_bwd(ishara_model* m, R5MHSA& a, const Run& r, const void* x, const void* g, void* gn) {
    const int dt = m->dt;
    OpArgs no; EpiArgs e0;
    void* dr = m->W(m->t4);
    CK(r5_ln_bwd(m, r, g, m->W(a.r), a.ln, a.mean, a.rstd, dr));
    CK(gemm_dgrad(m, a.Wp, dr, dt, m->W(m->t1), r.M, OP_NONE, no, e0));                          // d o
    CK(gemm_wgrad(m, a.Wp, m->W(a.o), dt, OP_NONE, no, dr, dt, OP_NONE, no, r.M));
    const float scale = 1.0f / sqrtf((float)m->dh);
    CKP(m, "attn_bwd", 8.0 * r.M * m->d * (double)dt_size(dt), 10.0 * r.B * m->H * (double)m->T * m->T * m->dh,
        launch_attn_bwd(dt, m->W(a.q), m->W(a.k), m->W(a.vt), m->W(a.o), m->W(m->t1), m->Wf(a.lse), m->Wf(m->delta), m->W(m->t2), r.B, m->H, m->T, m->dh, scale,
                        dspec_attn(r, a.site_attn, m->cfg.dropout_rate), 1, m->cfg.attn_impl, reinterpret_cast<uint32_t*>(m->W(a.maskw)), m->s));
    EpiArgs e1; e1.resid = dr;
    CK(gemm_dgrad(m, a.Wqkv, m->W(m->t2), dt, gn, r.M, OP_NONE, no, e1));
    CK(gemm_wgrad(m, a.Wqkv, x, dt, OP_NONE, no, m->W(m->t2), dt, OP_NONE, no, r.M));
    return 0;
}

extern "C" int ishara_encoder_backward(ishara_model* m, const float* dy, int32_t B, float* dx, ishara_stream st) {
    if (!m->ws || !m->grads) { ishara_set_error("ishara_encoder_backward: model is not bound (grads required)"); return -1; }
    if (B != m->lastB || !m->last_training) { ishara_set_error("ishara_encoder_backward: call ishara_encoder_forward(training=1) with the same batch first"); return -1; }
    if (m->family == ISHARA_FAMILY_TORCH_SQUEEZEFORMER) return r4_backward(m, dy, B, dx, (hipStream_t)st);
    if (m->family != ISHARA_FAMILY_TORCH_CONFORMER) { ishara_set_error("ishara_encoder_backward: handle is not an encoder-only family"); return -1; }
    m->s = (hipStream_t)st;
    Run r{B, B * m->T, 1, m->last_seed};
    CK(launch_fill_u32(m->grads, (size_t)m->n_train, 0u, m->s));
    void* g = m->W(m->gA); void* gn = m->W(m->gB);
    const void* g0 = dy;
    if (m->dt != DT_F32) { CKP(m, "cast", 6.0 * r.M * m->d, 0, r5_from_f32(m->dt, dy, g, (size_t)r.M * m->d, m->s)); g0 = g; }
    bool first = true;
    for (int li = (int)m->r5.size() - 1; li >= 0; --li) {
        R5Block& b = m->r5[li];
        const void* lin = li > 0 ? m->W(m->r5[li - 1].out) : (m->dt != DT_F32 ? (const void*)m->W(m->r5_x) : (const void*)m->last_x);
        const void* gin = first ? g0 : g;
        if (first && m->dt == DT_F32) { /* dy is read in place; the first module writes gn */ }
        CK(r5_ln_bwd(m, r, gin, m->W(b.ffn2.out), b.ln, b.mean, b.rstd, gn));
        first = false;
#define STEP(call) do { void* _t = g; g = gn; gn = _t; CK(call); } while (0)
        STEP(r5_ffn_bwd(m, b.ffn2, r, m->W(b.conv.out), g, gn));
        STEP(confconv_bwd(m, b.conv, r, m->W(b.mha.out), g, gn));
        STEP(r5_mhsa_bwd(m, b.mha, r, m->W(b.ffn1.out), g, gn));
        STEP(r5_ffn_bwd(m, b.ffn1, r, lin, g, gn));
#undef STEP
        { void* _t = g; g = gn; gn = _t; }           // g = gradient of this block's input
    }
    if (dx) CKP(m, "cast", 6.0 * r.M * m->d, 0, r5_to_f32(m->dt, g, dx, (size_t)r.M * m->d, m->s));
    if (!m->bucket_ev.empty()) HIP_CHECK_RET(hipEventRecord(m->bucket_ev.back(), m->s));
    return 0;
}

// frames per clip of the encoder output ([B, frames, dim]): T for the ConformerEncoder, the subsampled / reduced / recovered length
// (convolution.py:68-69, :266-267, encoder.py:162) for the SqueezeformerEncoder
extern "C" int32_t ishara_encoder_output_frames(const ishara_model* m) {
    if (m->family == ISHARA_FAMILY_TORCH_SQUEEZEFORMER) return r4_output_frames(m);
    return m->T;
}
