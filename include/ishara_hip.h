/* ishara_hip.h — C ABI of libishara_hip.so: the MI355X (gfx950) hot path of the Ishara
 * ASL-fingerspelling recogniser (Conv1D -> Squeezeformer -> Conformer CTC encoder).
 *
 * The reference (tanmayrainanda/ishara) has no native code and no FFI: its hot path is the
 * Keras object built by get_model(...) in `Test Notebooks/conv-hybrid-model.ipynb` and
 * driven by model.fit.  Each entry point below names the reference construct it replaces.
 *
 * Conventions: every function returns 0 on success, <0 on error (ishara_last_error() gives
 * a thread-local message).  The CALLER owns every buffer (parameters, gradients, optimizer
 * slots, workspace, inputs, outputs; torch-ROCm tensors in the Python host).  The library
 * allocates no device memory, never synchronises, and launches everything on the hipStream_t
 * it is given.  A handle is bound to the device current at ishara_bind() and is not
 * thread-safe: one handle per rank.
 */
#ifndef ISHARA_HIP_H
#define ISHARA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ishara_model ishara_model;
typedef void* ishara_stream;              /* hipStream_t */

/* activation storage + MFMA input type.  ISHARA_F16: inference only (the fp16 TFLite export of c14:1-5; ishara_forward(training=1)
 * is refused) — weights and activations in fp16, fp16 MFMA with fp32 accumulation, fp32 statistics / softmax / logits. */
enum { ISHARA_F32 = 0, ISHARA_BF16 = 1, ISHARA_F16 = 2 };
/* model families behind one handle type: the Keras hybrid of get_model (conv-hybrid-model.ipynb c7:1-72), the torch
 * ConformerEncoder of conformer/conformer.py:76-87 (post-LN blocks, encoder stack only) and the torch SqueezeformerEncoder of
 * squeezeformer/encoder.py:26-166 (conv2d subsampling, relative-position MHSA, time reduction / recovery) */
enum { ISHARA_FAMILY_KERAS_HYBRID = 0, ISHARA_FAMILY_TORCH_CONFORMER = 1, ISHARA_FAMILY_TORCH_SQUEEZEFORMER = 2 };

/* get_model(...) kwargs — conv-hybrid-model.ipynb c7:1-11 — plus the notebook globals the
 * function closes over (INPUT_SHAPE c3:119, len(char_to_num) c1:7) and the variant knobs
 * of the sibling notebooks (top_conv width, per-family expansion). 0 / negative = default. */
typedef struct ishara_config {
    int32_t dim;
    int32_t num_conv_squeeze_blocks;
    int32_t num_conv_conform_blocks;
    int32_t num_kernel_sizes;
    int32_t kernel_sizes[8];
    int32_t num_conv_per_block;
    float   dropout_rate;
    int32_t num_heads;
    int32_t expansion_factor;
    int32_t transformer_kernel_size;
    int32_t frames;                 /* T  = INPUT_SHAPE[0] */
    int32_t features;               /* F  = INPUT_SHAPE[1] */
    int32_t num_classes;            /* 60; blank = num_classes-1 */
    int32_t top_dim;                /* 0 -> 2*dim (c7:61) */
    int32_t squeeze_expansion;      /* 0 -> expansion_factor */
    int32_t conformer_expansion;    /* 0 -> expansion_factor */
    float   head_dropout;           /* c7:62: 0.4 */
    float   conformer_attn_dropout; /* c5:312 default 0.1 */
    int32_t dtype;                  /* ISHARA_F32 | ISHARA_BF16 | ISHARA_F16 (inference only): activation storage + MFMA input type */
    int32_t max_batch;              /* workspace is planned for this many clips */
    int32_t max_label_len;          /* MAX_PHRASE_LENGTH = 64 (c1:28) */
    int32_t attn_impl;              /* 0 lane-split VALU, 1 MFMA (bf16 only) */
    int32_t family;                 /* ISHARA_FAMILY_*.  TORCH_CONFORMER reads: dim, num_conv_conform_blocks (= num_layers), num_heads,
                                     * expansion_factor, transformer_kernel_size (= kernel_size, odd), dropout_rate, frames, dtype, max_batch */
    /* ISHARA_FAMILY_TORCH_SQUEEZEFORMER (encoder.py:54-69) reads: features (= input_dim), dim (= encoder_dim), num_conv_conform_blocks
     * (= num_layers), num_heads, expansion_factor (= feed_forward_expansion_factor), transformer_kernel_size (= conv_kernel_size),
     * dropout_rate (all five dropout probabilities), frames (input frames, any count >= 7), dtype, max_batch and: */
    int32_t reduce_layer_index;     /* >= num_layers: no time reduction */
    int32_t recover_layer_index;    /* >= num_layers: no recovery */
    int32_t half_step_residual;     /* 0 | 1 */
} ishara_config;

const char* ishara_last_error(void);

/* tf.keras.Model construction (c7:12-65).  Host only: touches no GPU. */
int  ishara_create(const ishara_config* cfg, ishara_model** out);
void ishara_destroy(ishara_model* m);

/* model.summary() / model.weights (c7:83): flat fp32 parameter buffer layout.  Trainable
 * entries come first (offsets [0, trainable)), BatchNorm moving statistics after them, so
 * the gradient buffer [0, trainable) is one contiguous all-reduce bucket. */
int64_t ishara_param_total(const ishara_model* m);
int64_t ishara_param_trainable(const ishara_model* m);
int32_t ishara_param_entries(const ishara_model* m);
int  ishara_param_info(const ishara_model* m, int32_t i, const char** name, int32_t* ndim,
                       int64_t shape[2], int64_t* offset, int32_t* trainable);
int64_t ishara_workspace_bytes(const ishara_model* m);

/* Debug aids (no reference counterpart).  plan_check: host-side audit of the workspace plan (alignment, bounds, no overlap);
 * returns the buffer count.  guard_check: with ISHARA_WS_GUARD=1 in the environment at ishara_create every workspace buffer is
 * followed by a 256-byte guard zone armed by ishara_bind; returns 0 if all guards are intact (synchronises the device). */
int32_t ishara_workspace_plan_check(const ishara_model* m);
int ishara_workspace_guard_check(ishara_model* m);

/* Bind caller-owned device buffers.  params/grads: ishara_param_total floats (grads only
 * uses the trainable prefix); opt_m/opt_v/opt_slow: ishara_param_trainable floats each
 * (may be NULL for inference); workspace: ishara_workspace_bytes bytes, 256-B aligned. */
int ishara_bind(ishara_model* m, float* params, float* grads, float* opt_m, float* opt_v,
                float* opt_slow, void* workspace, int64_t workspace_bytes);
/* Re-derive the MFMA-typed weight shadows after the caller wrote `params` (load_weights). */
int ishara_sync_weights(ishara_model* m, ishara_stream s);

/* model(x, training=...) — c7:82, c9:15, c13:17.  x [B,T,F] f32, logits [B,T,C] f32. */
int ishara_forward(ishara_model* m, const float* x, int32_t B, float* logits, int32_t training,
                   uint32_t seed, ishara_stream s);
/* CTCLoss (c6:1-13) + tape.gradient of Keras train_step (c12).  Uses the activations saved
 * by the last ishara_forward(training=1).  labels [B,L] int64 padded with blank.
 * loss (device scalar) = mean_b nll_b; gradients of loss*loss_scale fill grads[0,trainable). */
int ishara_loss_backward(ishara_model* m, const float* logits, const int64_t* labels, int32_t B,
                         float* loss, float* nll, float loss_scale, ishara_stream s);
/* ConformerEncoder.forward (conformer/conformer.py:84-87) for an ISHARA_FAMILY_TORCH_CONFORMER handle: x, y [B,T,dim] f32.
 * training=1: Dropout active (seed), BatchNorm1d uses batch statistics and updates running_mean / running_var, the
 * activations the backward pass needs stay in the workspace. */
int ishara_encoder_forward(ishara_model* m, const float* x, int32_t B, float* y, int32_t training, uint32_t seed, ishara_stream s);
/* frames per clip of y: `frames` for the ConformerEncoder; ((T-3)/2+1-3)/2+1, then (.-3)/2+1 after the reduction layer and twice
 * that after the recovery layer for the SqueezeformerEncoder (x is [B, frames, features] there) */
int32_t ishara_encoder_output_frames(const ishara_model* m);
/* loss.backward() through the encoder (conformer.py:99-103): dy [B,T,dim] f32 = dLoss/dy of the last ishara_encoder_forward(training=1);
 * parameter gradients fill grads[0,trainable) (overwritten); dx [B,T,dim] f32 may be NULL. */
int ishara_encoder_backward(ishara_model* m, const float* dy, int32_t B, float* dx, ishara_stream s);
/* Gradient buckets for data parallelism (replaces what tf.distribute / nn.DataParallel do inside the reference's
 * train step: nb4 c1:63-75, integration.py:1058-1060).  The backward pass completes the flat gradient from its end
 * (head) towards its start (stem); ishara_grad_bucket(i) gives range i in completion order and
 * ishara_grad_bucket_wait(i, side) makes the caller's side stream wait for it, so that the all-reduce of a finished
 * range overlaps the rest of the backward pass.  Call ishara_grad_buckets_enable once before the backward passes
 * whose ranges are waited for. */
int32_t ishara_grad_buckets(const ishara_model* m);
int ishara_grad_bucket(const ishara_model* m, int32_t i, int64_t* offset, int64_t* count);
int ishara_grad_buckets_enable(ishara_model* m);
int ishara_grad_bucket_wait(ishara_model* m, int32_t i, ishara_stream side);
/* Lookahead(RectifiedAdam(sma_threshold=4), sync_period=5) apply_gradients — c7:68-69. */
int ishara_optimizer_step(ishara_model* m, float lr, float weight_decay, ishara_stream s);
int32_t ishara_optimizer_iterations(const ishara_model* m);
int ishara_optimizer_set_iterations(ishara_model* m, int32_t it);

/* HIP-event profiler (no reference counterpart: the reference profiles with %%timeit / Keras
 * progress bars).  When enabled, every kernel launch of forward / loss_backward /
 * optimizer_step is bracketed by events on the launch stream; the report is text, one line per
 * kernel family: "name launches total_ms algorithmic_bytes flops".  Returns bytes written. */
int ishara_profile_enable(ishara_model* m, int32_t on);
int ishara_profile_report(ishara_model* m, char* buf, int32_t cap);

/* decode_phrase (c8:4-12) for a batch: out_idx [B,T] int32 (-1 padded), out_len [B]. */
int ishara_greedy_decode(const float* logits, int32_t B, int32_t T, int32_t C, int32_t blank,
                         int32_t* out_idx, int32_t* out_len, ishara_stream s);
/* pre_process1(*pre_process00(x)) of the TFLite wrapper (c3:61-115, c13:9-15): raw [max_frames,276] landmarks with NaNs
 * (SEL_COLS order, c1:22-26), clip length read from device memory (*n_frames), mean/std [276] in output order
 * -> out [T,276].  One kernel, graph-capturable. */
int ishara_preprocess(const float* raw, const int32_t* n_frames, int32_t max_frames, const float* mean, const float* stdv,
                      float* out, int32_t T, ishara_stream s);
/* tf.nn.ctc_loss alone: nll [B]; dlogits [B,T,C] may be NULL; ws = ishara_ctc_workspace_bytes. */
int64_t ishara_ctc_workspace_bytes(int32_t B, int32_t T, int32_t L);
int ishara_ctc_loss(const float* logits, const int64_t* labels, int32_t B, int32_t T, int32_t C,
                    int32_t L, int32_t blank, float* nll, float* dlogits, float grad_scale,
                    void* ws, ishara_stream s);
/* The dropout mask the kernels draw for (seed, site): out [rows, cols] f32 (0 or 1/(1-rate)). */
int ishara_dropout_mask(uint32_t seed, uint32_t site, int32_t rows, int32_t cols, float rate,
                        float* out, ishara_stream s);

/* tests/ablation: bit0 register-staged NT GEMM, bit1 register-transposing TN GEMM, bit2 LDS-tiled depthwise conv
 * (defaults: LDS-DMA NT, transposed-read TN, register-window depthwise conv); bits 4-11 ablation switches */
/* switches of the A-stationary GEMM (1 paired half-line stores, 2 non-temporal side outputs, 16 chunked K = 256 form); -1 = library default */
int ishara_debug_set_as_flags(int32_t flags);
/* 0: never use the 256 x 256 two-operand tile GEMM (gemm_big.hip) — A/B runs against the A-stationary kernel inside one process; 1: library default */
int ishara_debug_set_nt_big(int32_t on);
int ishara_debug_force_regstage(int32_t on);

/* ---- single-operator entry points (parity tests of the individual kernels) ------------ */
/* y = act(x @ W + b): x [M,K] (dtype dt), W [K,N] f32, y [M,N] (dt); scratch >= ishara_op_scratch_bytes */
int64_t ishara_op_scratch_bytes(int32_t M, int32_t K, int32_t N);
int ishara_op_dense_fwd(int32_t dt, const void* x, const float* W, const float* bias, void* y,
                        int32_t M, int32_t K, int32_t N, int32_t act, void* scratch, ishara_stream s);
/* y = act(x @ W + b) + resid   (resid may be NULL) */
int ishara_op_dense_fwd_ex(int32_t dt, const void* x, const float* W, const float* bias, const void* resid, void* y,
                           int32_t M, int32_t K, int32_t N, int32_t act, void* scratch, ishara_stream s);
/* dx = dy @ W^T ; dW += x^T dy ; db += colsum(dy) */
int ishara_op_dense_bwd(int32_t dt, const void* x, const float* W, const void* dy, void* dx,
                        float* dW, float* db, int32_t M, int32_t K, int32_t N, void* scratch, ishara_stream s);
/* y[m,:C] = x[m,:C] - logsumexp(x[m,:C]) over C fp32 logits in rows of stride ld >= C floats (columns C..ld-1 of the outputs are zeroed: a class
 * count padded to the GEMMs' 16-byte row alignment); dx = dy - exp(y) * rowsum(dy).  Replaces F.log_softmax(self.fc(...), dim=-1) of the
 * reference's torch Squeezeformer (squeezeformer/model.py:448-449). */
int ishara_op_log_softmax_fwd(const float* x, float* y, int32_t M, int32_t C, int32_t ld, ishara_stream s);
int ishara_op_log_softmax_bwd(const float* dy, const float* y, float* dx, int32_t M, int32_t C, int32_t ld, ishara_stream s);
int ishara_op_layernorm_fwd(int32_t dt, const void* x, const float* gamma, const float* beta, float eps,
                            void* y, float* mean, float* rstd, int32_t M, int32_t C, ishara_stream s);
int ishara_op_layernorm_bwd(int32_t dt, const void* dy, const void* x, const float* mean, const float* rstd,
                            const float* gamma, void* dx, float* dgamma, float* dbeta, int32_t M, int32_t C,
                            ishara_stream s);
/* inop: 0 none, 1 swish, 2 GLU (x has 2C channels) */
int ishara_op_dwconv_fwd(int32_t dt, int32_t inop, const void* x, const float* w, const float* bias, void* y,
                         float* ssum, float* ssq, int32_t B, int32_t T, int32_t C, int32_t k, int32_t padl,
                         ishara_stream s);
/* the same with caller scratch (ishara_op_dwconv_fwd_scratch_bytes) for the statistics: per-workgroup partial rows summed in a fixed order, no
 * float atomics — the path the model itself takes */
int64_t ishara_op_dwconv_fwd_scratch_bytes(int32_t B, int32_t T, int32_t C);
int ishara_op_dwconv_fwd_ex(int32_t dt, int32_t inop, const void* x, const float* w, const float* bias, void* y,
                            float* ssum, float* ssq, void* scratch, int32_t B, int32_t T, int32_t C, int32_t k,
                            int32_t padl, ishara_stream s);
int64_t ishara_op_dwconv_scratch_bytes(int32_t C, int32_t k);
/* scratch: ishara_op_dwconv_scratch_bytes(C,k) bytes, or NULL (atomic weight-grad path) */
int ishara_op_dwconv_bwd(int32_t dt, int32_t inop, const void* dy, const void* x, const float* w, void* dx,
                         float* dw, float* dbias, void* scratch, int32_t B, int32_t T, int32_t C, int32_t k,
                         int32_t padl, ishara_stream s);
/* attention on packed qkv [B*T, 3*H*dh] (head-major packing): o [B*T, H*dh]; scratch holds q,k,vt,lse,delta */
int64_t ishara_op_attn_scratch_bytes(int32_t B, int32_t H, int32_t T, int32_t dh);
int ishara_op_attn_fwd(int32_t dt, const void* qkv, void* o, int32_t B, int32_t H, int32_t T, int32_t dh,
                       float scale, uint32_t seed, uint32_t site, float rate, int32_t impl,
                       void* scratch, ishara_stream s);
int ishara_op_attn_bwd(int32_t dt, const void* o, const void* dout, void* dqkv, int32_t B, int32_t H,
                       int32_t T, int32_t dh, float scale, uint32_t seed, uint32_t site, float rate,
                       int32_t impl, void* scratch, ishara_stream s);

#ifdef __cplusplus
}
#endif
#endif
