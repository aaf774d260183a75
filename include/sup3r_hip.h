/* sup3r_hip.h — C-ABI of libsup3r_hip.so, the MI355X (gfx950) compute core
 * for NREL/sup3r's Sup3rGan hot path.
 *
 * The reference has NO native/FFI seam: its hot path is a Python loop over
 * keras layer objects inside TensorFlow.  This header defines the boundary a
 * sup3r maintainer binds with ctypes.  Each entry point names
 * the reference interface it replaces (paths relative to the sup3r repo).
 *
 * Conventions
 *  - plain C: opaque handles, raw device/host pointers, sizes; no torch types.
 *  - every call returns 0 on success or a negative S3_E* code; the message is
 *    retrievable with s3_last_error().  No exceptions cross the ABI.
 *  - tensors are channels-last fp32, always described 5-D (N, s1, s2, t, C);
 *    4-D (spatial) nets use t = 1.  Device pointers are caller-owned unless
 *    stated otherwise; all work is enqueued on the context's HIP stream.
 *  - a context is used by one host thread at a time (one process per GPU).
 */
#ifndef SUP3R_HIP_H
#define SUP3R_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S3_OK 0
#define S3_EINVAL (-1)   /* bad argument / unsupported op description */
#define S3_EHIP (-2)     /* HIP runtime error */
#define S3_ENOMEM (-3)
#define S3_ERCCL (-4)
#define S3_ESTATE (-5)   /* call sequence error (e.g. backward w/o forward) */
#define S3_ETIMEOUT (-6) /* s3_comm_wait: the device work (collectives) did not finish in time */

/* op kinds of the fused plan (sup3r_amd/spec.py lowers the reference's
 * hidden_layers JSON — sup3r/configs/<family>/<model>.json — to these) */
enum {
  S3_OP_CONV = 1,     /* [virtual pad] + conv + bias + act + residual + d2s  */
  S3_OP_REPEAT_T = 2, /* SpatioTemporalExpansion temporal nearest            */
  S3_OP_D2S = 3,      /* depth_to_space (DCR) on (s1, s2)                    */
  S3_OP_ACT = 4,
  S3_OP_ADD = 5,      /* SkipConnection end / Sup3rAdder                     */
  S3_OP_CONCAT = 6,   /* Sup3rConcat                                         */
  S3_OP_DENSE = 7,
  S3_OP_PAD = 8,      /* FlexiblePadding not consumed by a conv              */
  S3_OP_CROP = 9,
  S3_OP_VIEW = 10,    /* Flatten / depth_to_time reshape (alias, no copy)    */
  S3_OP_ROLL_T = 11,  /* tf.roll along t (depth_to_time t_roll)              */
  S3_OP_DILATE = 12   /* zero insertion by stride[] (strided ConvNDTranspose) */
};
enum { S3_ACT_NONE = 0, S3_ACT_RELU = 1, S3_ACT_LEAKY = 2 };
enum { S3_PAD_ZERO = 0, S3_PAD_REFLECT = 1 };
/* arithmetic mode of the MFMA convolution kernels */
enum {
  S3_PREC_F32 = 0,   /* v_mfma_f32_16x16x4_f32: exact fp32 (parity mode)     */
  S3_PREC_BF16 = 1,  /* v_mfma_f32_16x16x32_bf16, fp32 accumulate            */
  S3_PREC_BF16X3 = 2 /* 3-term bf16 split (hi*hi + hi*lo + lo*hi), ~fp32     */
};
enum { S3_LOSS_MAE = 0, S3_LOSS_MSE = 1, S3_LOSS_EXP = 2 };
enum { S3_BUF_W = 0, S3_BUF_G = 1, S3_BUF_M = 2, S3_BUF_V = 3 };

typedef struct s3_ctx s3_ctx;
typedef struct s3_params s3_params;
typedef struct s3_plan s3_plan;

typedef struct {
  int64_t dims[5]; /* N, s1, s2, t, C */
} s3_tensor_desc;

typedef struct {
  int32_t kind;
  int32_t in0, in1, res, out; /* tensor ids, -1 = none                      */
  int32_t w, b;               /* parameter ids, -1 = none                   */
  int32_t k[3], stride[3], lo[3], hi[3]; /* conv / pad / crop geometry      */
  int32_t pad_mode;
  int32_t act;
  float alpha;
  int32_t d2s; /* depth-to-space block fused at the conv store (1 = none)  */
  int32_t rep; /* REPEAT_T factor / ROLL_T shift                           */
  int32_t bcast_c; /* ADD: in1 has one channel, broadcast over C           */
  int32_t reserved[4];
} s3_op_desc;

/* ---- context ----------------------------------------------------------
 * replaces: tf.device(default_device) placement in
 * sup3r/models/abstract.py:41-42,1230 and base.py:104-106.
 * stream: the hipStream_t all work is enqueued on — one the caller already
 * owns (e.g. torch.cuda.current_stream().cuda_stream; NULL is the device's
 * default stream).  create_stream != 0 ignores `stream` and creates a private
 * non-blocking stream instead. */
int s3_ctx_create(int device_id, void* stream, int create_stream, s3_ctx** out);
void s3_ctx_destroy(s3_ctx* ctx);
const char* s3_last_error(const s3_ctx* ctx);
int s3_ctx_sync(s3_ctx* ctx);
void* s3_ctx_stream(s3_ctx* ctx);
/* launch counters of run-time kernel choices the plan cannot report up front
 * (tests assert the path they mean to exercise was taken); -1 for an unknown
 * counter */
enum { S3_STAT_PERSIST_DGRAD = 0, /* trunk data gradients on the persistent kernel */
       S3_STAT_GCONV_SPLITK = 1,  /* gather-MFMA launches with a split contraction  */
       S3_STAT_BUCKET_ELEMS = 2,  /* gradient elements all-reduced bucket by bucket  */
       S3_STAT_DGRAD_C2_SLIDE = 3, /* first-layer data gradients on the sliding kernel */
       S3_STAT_BUCKETS = 4,       /* bucket collectives issued under backward passes */
       S3_STAT_ALLREDUCES = 5,    /* whole-buffer / scalar all-reduces issued        */
       S3_STAT_COUNT = 6 };
int64_t s3_ctx_stat(const s3_ctx* ctx, int which);

/* ---- parameter store ---------------------------------------------------
 * replaces: phygnn.CustomNetwork.weights (list of tf.Variable: kernel, bias
 * per layer in layer order — sup3r/models/abstract.py:312-319,
 * base.py:226-235,388-392), the gradient list of tape.gradient
 * (abstract.py:1237) and the keras Adam slot variables
 * (abstract.py:566-587).  One store per network; four flat fp32 device
 * buffers (weights, grads, Adam m, Adam v) so that the optimizer step and the
 * gradient all-reduce are ONE launch / ONE collective.
 * Kernels are stored in canonical [taps..., C_in, C_out] layout (== keras
 * Conv layout; Conv2DTranspose is flipped/transposed by the host shim). */
int s3_params_create(s3_ctx* ctx, int n, const int64_t* sizes, s3_params** out);
void s3_params_destroy(s3_params* p);
int64_t s3_params_total(const s3_params* p);
int s3_params_set(s3_params* p, int which, int idx, const float* host);
int s3_params_get(s3_params* p, int which, int idx, float* host);
void* s3_params_dptr(s3_params* p, int which, int idx);
int s3_params_zero_grad(s3_params* p);
/* counter that changes whenever the weights change (set, Adam step,
 * broadcast): lets a caller reuse a forward result — Sup3rGan._train_batch
 * runs D(hi_res_true) in the generator step and again, with the SAME
 * discriminator weights, in the discriminator step (base.py:1001-1025) */
uint64_t s3_params_version(const s3_params* p);
/* mean(|x|) of one Adam slot / weight tensor (history columns
 * OptmGen/Adam/m/..., abstract.py:582-586) */
int s3_params_mean_abs(s3_params* p, int which, int idx, float* host_out);

/* keras-2.15 Adam.update_step on the whole store (abstract.py:899,912):
 * alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g*g-v)(1-b2);
 * w -= m*alpha/(sqrt(v)+eps).  t is the 1-based step count. */
int s3_adam_step(s3_params* p, float lr, float beta1, float beta2, float eps,
                 int64_t t);
/* The other keras optimizers ``init_optimizer`` may be handed by name
 * (abstract.py:321-350, models/utilities.py:150-158), keras-2.15
 * ``update_step`` each, one fused pass over the store (slots: BUF_M / BUF_V).
 * hp[] (doubles, cast like keras casts its Python scalars: 1 - beta in double,
 * then fp32): Adam {lr, beta_1, beta_2, epsilon}; SGD {lr, momentum, nesterov};
 * RMSprop {lr, rho, momentum, epsilon} (centered=False); Adagrad {lr,
 * epsilon, initial_accumulator_value}; Adamax {lr, beta_1, beta_2, epsilon};
 * AdamW {lr, beta_1, beta_2, epsilon, weight_decay}. */
typedef enum {
  S3_OPT_ADAM = 0, S3_OPT_SGD = 1, S3_OPT_RMSPROP = 2, S3_OPT_ADAGRAD = 3,
  S3_OPT_ADAMAX = 4, S3_OPT_ADAMW = 5
} s3_optimizer_kind;
int s3_optimizer_step(s3_params* p, int kind, const double* hp, int n_hp,
                      int64_t t);
/* The same step in two halves, for a captured graph (below): `stage` writes
 * the scalars of step t (alpha(t) of Adam, ...) to the device with a 1-thread
 * launch and is called OUTSIDE the graph before every replay; `step_staged`
 * is the update launch, reading them from there — identical every step, so it
 * can be recorded.  (Adagrad's step 1 fills the accumulator: run it with
 * s3_optimizer_step.)  `touch` tells the host side that the weights changed
 * behind its back (after a replay): packed filter images are stale. */
int s3_optimizer_stage(s3_params* p, int kind, const double* hp, int n_hp,
                       int64_t t);
int s3_optimizer_step_staged(s3_params* p, int kind);
int s3_params_touch(s3_params* p);

/* ---- captured steps ----------------------------------------------------------
 * A launch-bound training step — the reference's own CPU-runnable case
 * (tests/training/test_train_gan.py:45-114, batch 15 of 5 x 5 -> 10 x 10) is
 * ~650 launches of a few microseconds — recorded once as a hipGraph and
 * replayed with ONE launch per Sup3rGan._train_batch (base.py:944-1031).
 * Between begin and end every launch of the context is recorded instead of
 * executed.  The caller guarantees what a replay needs: every buffer the step
 * touches stays alive and in place (inputs are copied INTO the recorded input
 * buffers), no host read-back / collective / plan creation inside, the
 * optimizer steps staged.  A call that cannot be recorded fails capture_end
 * (S3_EHIP); s3_capture_abort drops a capture after an error. */
typedef struct s3_graph s3_graph;
int s3_capture_begin(s3_ctx* ctx);
int s3_capture_end(s3_ctx* ctx, s3_graph** out);
int s3_capture_abort(s3_ctx* ctx);
int s3_graph_launch(s3_graph* g);
int64_t s3_graph_nodes(const s3_graph* g);
void s3_graph_destroy(s3_graph* g);

/* ---- options ---------------------------------------------------------------
 * Kernel-selection switches (the A/B comparisons of the parity tests, the
 * profiling ablations) are options of a context and of the plans created from
 * it, not process environment: `NO_PERSIST`, `NO_WGRAD_BF16`, `MFMA_TILE`,
 * `HALO32_MIN_TILES`, ... (s3_option_name_at enumerates them; DESIGN.md §5.4
 * says what each one does).  An option is unset (the default behaviour) or
 * carries an int32.  s3_ctx_create reads the variables SUP3R_AMD_<NAME> ONCE
 * as the initial defaults of that context; nothing reads the environment
 * afterwards.  A plan snapshots the context's options when it is created,
 * overridden by the `options` of s3_plan_create_opt; the snapshot governs
 * that plan's forward / backward launches. */
#define S3_OPTION_UNSET INT32_MIN
typedef struct {
  int32_t n;                  /* number of (name, value) pairs */
  const char* const* names;   /* "NO_PERSIST" (or "SUP3R_AMD_NO_PERSIST") */
  const int32_t* values;      /* S3_OPTION_UNSET removes the option */
} s3_plan_options;
int s3_ctx_set_option(s3_ctx* ctx, const char* name, int32_t value);
/* returns 1 if the option is set (value written), 0 if unset, < 0 unknown */
int s3_ctx_get_option(const s3_ctx* ctx, const char* name, int32_t* value);
/* option names, index 0 .. until NULL */
const char* s3_option_name_at(int index);

/* ---- plan (shape-specialised executor) ---------------------------------
 * replaces: the eager/graph layer loops AbstractSingleModel._tf_generate
 * (abstract.py:1131-1173) and Sup3rGan._tf_discriminate (base.py:283-313).
 * inputs[]: tensor ids fed by the caller at forward time (low_res first, then
 * hi-res exo tensors in Sup3rConcat/Sup3rAdder layer order).
 * training != 0 keeps every activation for s3_plan_backward. */
int s3_plan_create(s3_ctx* ctx, s3_params* params, const s3_tensor_desc* tensors,
                   int n_tensors, const s3_op_desc* ops, int n_ops,
                   const int32_t* inputs, int n_inputs, int32_t output,
                   int precision, int training, s3_plan** out);
/* the same with per-plan options on top of the context's (NULL = none) */
int s3_plan_create_opt(s3_ctx* ctx, s3_params* params, const s3_tensor_desc* tensors,
                       int n_tensors, const s3_op_desc* ops, int n_ops,
                       const int32_t* inputs, int n_inputs, int32_t output,
                       int precision, int training, const s3_plan_options* options,
                       s3_plan** out);
void s3_plan_destroy(s3_plan* plan);
/* inputs: device pointers (fp32, NDHWC) in the order given at creation;
 * output: device pointer receiving the result, or NULL to leave it in the
 * plan's own buffer (s3_plan_tensor). */
int s3_plan_forward(s3_plan* plan, const void* const* inputs, void* output);
/* The chunk executor's forward (ForwardPass._run_generator + the hr_crop_slice
 * of forward_pass.py:384-425 + un_norm_output, abstract.py:243-275) in one
 * pass: the plan's LAST convolution computes only the window [lo, lo + n) of
 * its output positions per axis (the chunk without its halo), applies
 * y * scale[c] + shift[c] (affine_dev: device pointer to scale[n_c] then
 * shift[n_c], two roundings like numpy; NULL = none) and writes the dense
 * (N, n0, n1, n2, C) window to `output`.  No full-size model output exists and
 * the halo positions of the last conv are never computed.  Supported
 * (s3_plan_supports_window == 1) for inference plans whose last op is the
 * bf16-input MFMA tail conv; S3_EINVAL otherwise — the caller then runs
 * s3_plan_forward + s3_chunk_epilogue. */
int s3_plan_supports_window(const s3_plan* plan);
int s3_plan_forward_window(s3_plan* plan, const void* const* inputs, void* output,
                           const int64_t* lo3, const int64_t* n3,
                           const float* affine_dev, int n_c);
/* reverse-mode pass of the last forward (tf.GradientTape().gradient,
 * abstract.py:1230-1237).  d_output: dL/d(output); d_input: nullable, receives
 * dL/d(inputs[0]).  accumulate_wgrad != 0 adds into the grad buffer (the
 * discriminator sees true and generated batches).  need_wgrad == 0 skips
 * weight gradients (generator step: only dgrad flows through the disc). */
int s3_plan_backward(s3_plan* plan, const void* d_output, void* d_input,
                     int need_wgrad, int accumulate_wgrad);
void* s3_plan_tensor(s3_plan* plan, int32_t tensor_id);
int64_t s3_plan_workspace_bytes(const s3_plan* plan);
/* per-op kernel timing with HIP events on the ctx stream (bench.py roofline
 * object).  After s3_plan_profile_begin(plan, max_forwards) every
 * s3_plan_forward records one event between consecutive ops (up to
 * max_forwards forwards); s3_plan_profile_end synchronises, writes the MEAN
 * duration in ms of each op over the recorded forwards into ms_per_op[0..cap)
 * and returns the number of forwards averaged (or a negative error). */
int s3_plan_profile_begin(s3_plan* plan, int max_forwards);
int s3_plan_profile_end(s3_plan* plan, float* ms_per_op, int cap);
/* 0: op i is not on MFMA; 1: MFMA halo-tile kernel (one tile per workgroup);
 * 2: its persistent variant (all-bf16 64 -> 64 trunk convs, >= 1 tile per CU) */
int s3_plan_op_is_mfma(const s3_plan* plan, int op_index);
/* introspection of the kernel selection (what SUP3R_AMD_TRACE prints), used by
 * the parity tests to assert which kernels a configuration runs on and to
 * build the bf16-emulating oracle (which convs round their operands / store
 * bf16).  Fills out[0..min(cap, S3_OPINFO_COUNT)) and returns S3_OPINFO_COUNT.
 * No reference counterpart: keras picks its conv algorithm inside TF. */
enum {
  S3_OPINFO_KIND = 0,          /* S3_OP_*                                      */
  S3_OPINFO_FWD = 1,           /* S3_FWD_* (convs)                             */
  S3_OPINFO_IN16 = 2,          /* input / output / residual stored as bf16     */
  S3_OPINFO_OUT16 = 3,
  S3_OPINFO_RES16 = 4,
  S3_OPINFO_FWD_BF16_OPS = 5,  /* forward kernel rounds x and w to bf16        */
  S3_OPINFO_WGRAD = 6,         /* S3_WGRAD_* (training plans)                  */
  S3_OPINFO_DGRAD = 7,         /* S3_DGRAD_*                                   */
  S3_OPINFO_MASK_FUSED_FROM = 8, /* op whose activation adjoint this conv's
                                  dgrad store applies, or -1                  */
  S3_OPINFO_IN_REP = 9,        /* conv reads its input through a fused temporal
                                  repeat of this factor; a repeat op / a concat op
                                  (Sup3rConcat): 1 = absorbed by its consumer
                                  conv (no launch)                              */
  S3_OPINFO_RES_REP = 10,      /* ... and its residual operand                  */
  S3_OPINFO_DGRAD_FRAME16 = 11, /* the padded-frame data gradient is stored as
                                  bf16 between the conv kernel and its fold    */
  S3_OPINFO_FEWPOS_MFMA = 12,  /* the few-positions launches of this conv are the
                                  one-launch fp32-MFMA kernels (forward, data
                                  gradient from the untransposed filter, weight
                                  + bias gradient)                              */
  S3_OPINFO_COUNT = 13
};
enum {
  S3_FWD_DIRECT = 0, S3_FWD_MFMA_TILE = 1, S3_FWD_MFMA_PERSIST = 2, S3_FWD_GCONV = 3,
  S3_FWD_GCONV_FEWCH = 4, S3_FWD_HALO32 = 5, S3_FWD_FEWPOS = 6, S3_FWD_TAIL_MFMA = 7,
  S3_FWD_SMALL = 8,
  S3_FWD_FUSED2D = 9, /* the whole op list in one launch (small 2-D stacks) */
  S3_FWD_HALO_S2 = 10, /* C_in = 32 stride-2 valid conv on an LDS halo         */
  S3_FWD_MFMA_GEN = 11, /* halo-tile MFMA over logical axes: 2-D nets, few time steps,
                           any C_in <= 256 / C_out (kernels_conv_mfma_gen.hip)     */
  S3_FWD_CONV2D_WS = 12, /* weights-stationary persistent Conv2D, all-bf16 64 -> 64 k
                           trunks of the 2-D generators (kernels_conv2d_ws.hip)    */
  S3_FWD_CONV2D_HEAD = 13 /* the few-feature head conv of those generators: C_in 1 / 2
                           -> 64, fp32 field in, bf16 cells out, filter in registers */
};
enum {
  S3_WGRAD_DIRECT = 0, S3_WGRAD_F32_TRUNK = 1, S3_WGRAD_BF16_TRUNK = 2, S3_WGRAD_F32_GEN = 3,
  S3_WGRAD_BF16_GEN = 4, S3_WGRAD_BF16_2D = 5, S3_WGRAD_C2 = 6, S3_WGRAD_TAIL = 7,
  S3_WGRAD_FEWPOS = 8
};
enum {
  S3_DGRAD_DIRECT = 0, S3_DGRAD_MFMA_FRAME = 1, S3_DGRAD_MFMA_VALID = 2,
  S3_DGRAD_MFMA_CHUNKED = 3, S3_DGRAD_FEWCH_FRAME = 4, S3_DGRAD_S2 = 5, S3_DGRAD_C2 = 6,
  S3_DGRAD_GCONV = 7, S3_DGRAD_FEWPOS = 8
};
int s3_plan_op_info(const s3_plan* plan, int op_index, int32_t* out, int cap);
/* 0 = fp32, 1 = bf16 storage of a plan tensor (bf16 plans keep the trunk in bf16) */
int s3_plan_tensor_dtype(const s3_plan* plan, int32_t tensor_id);
/* raw bytes of a plan tensor (in its storage dtype) to host memory after a
 * stream sync; returns the byte count or a negative error.  Training plans
 * keep every activation: the tests read the LeakyReLU masks the device used. */
int64_t s3_plan_tensor_read(s3_plan* plan, int32_t tensor_id, void* host, size_t cap_bytes);

/* ---- losses ------------------------------------------------------------
 * content loss: keras MeanAbsoluteError / MeanSquaredError as used by
 * Sup3rGan.calc_loss_gen_content (base.py:478-503): mean over the first
 * c_used channels of a (c_a channels, generated) vs b (c_b channels, truth
 * incl. trailing exo channels).  loss_out: device float (accumulated with
 * weight); d_a (nullable): d(weight*loss)/da written (accumulate != 0: added). */
int s3_loss_content(s3_ctx* ctx, int kind, const float* a, int c_a,
                    const float* b, int c_b, int c_used, int64_t n_pos,
                    float weight, float* loss_out, float* d_a, int accumulate);
/* masked variant used by Sup3rCondMom.calc_loss_cond_mom
 * (sup3r/models/conditional.py:221-241): loss(a * mask, b * mask). */
int s3_loss_content_masked(s3_ctx* ctx, int kind, const float* a, int c_a,
                           const float* b, int c_b, const float* mask, int c_m,
                           int c_used, int64_t n_pos, float weight,
                           float* loss_out, float* d_a, int accumulate);
/* relativistic BCE of Sup3rGan.calc_loss_disc (base.py:505-549).
 * loss_out: device float; d_true / d_gen nullable (n floats each), scaled by
 * `scale` (weight_gen_advers for the adversarial term). */
int s3_loss_rel_bce(s3_ctx* ctx, const float* disc_true, const float* disc_gen,
                    int n, float scale, float* loss_out, float* d_true,
                    float* d_gen);

/* ---- structured content losses -----------------------------
 * sup3r/utilities/loss_metrics.py: every loss there is M(F(gen), F(true)) with
 * M = MeanAbsoluteError / MeanSquaredError (s3_loss_content) and F a feature
 * map.  s3_lossmap_fwd writes F(x) for x = (n, s1, s2, t, c) fp32 (t = 1 for
 * 4-D), first c_used channels; s3_lossmap_bwd ADDS F'(x)^T g_out into
 * d_x[..., :c_used].  Shapes of F(x):
 *   S3_LMAP_DERIV_S  (n, s1, s2, t, c_used)   d/ds1 + d/ds2, np.gradient scheme
 *                                             (SpatialDerivativeLoss :228-260)
 *   S3_LMAP_DERIV_T  (n, s1, s2, t, c_used)   d/dt (TemporalDerivativeLoss :263-294)
 *   S3_LMAP_MATERIAL (n, s1, s2, t, c_used/2) du/dt + u du/ds1 + v du/ds2 of each
 *                                             (u, v) pair (MaterialDerivativeLoss :150-225)
 *   S3_LMAP_MEAN_S   (n, t, c_used)           mean over (s1, s2) (CoarseMseLoss :297-322)
 *   S3_LMAP_EXT_S    2 x (n, t, c_used)       [min | max] over (s1, s2) (:325-357)
 *   S3_LMAP_EXT_T    2 x (n, s1, s2, c_used)  [min | max] over t (:360-392)
 *   S3_LMAP_COARSEN  backward only (forward = s3_coarsen): p0 = s_enhance, p1 =
 *                    t_enhance, p2 = S3_TC_AVERAGE | S3_TC_SUBSAMPLE; g_out has
 *                    c channels (LowResLoss :488-638)
 * work: spatial reductions n * 64 * t * c_used floats; extremes adjoint
 * additionally 2 * (size of one extremum) in front.  fx (bwd, extremes only) =
 * the forward map of the same x.  Ties share the gradient equally, as
 * tf.reduce_min / reduce_max do. */
enum { S3_LMAP_DERIV_S = 0, S3_LMAP_DERIV_T = 1, S3_LMAP_MATERIAL = 2, S3_LMAP_MEAN_S = 3,
       S3_LMAP_EXT_S = 4, S3_LMAP_EXT_T = 5, S3_LMAP_COARSEN = 6 };
int s3_lossmap_fwd(s3_ctx* ctx, int kind, const float* x, int n, int s1, int s2,
                   int t, int c, int c_used, int p0, int p1, int p2, float* out,
                   float* work);
int s3_lossmap_bwd(s3_ctx* ctx, int kind, const float* x, const float* fx,
                   const float* g_out, int n, int s1, int s2, int t, int c,
                   int c_used, int p0, int p1, int p2, float* d_x, float* work);
/* MmdLoss (loss_metrics.py:62-147): gaussian-kernel maximum mean discrepancy
 * over all pairs of the n observations at every one of the n_pos positions;
 * a, b = (n, n_pos, c_*).  loss_out = unweighted value; d_a (nullable) +=
 * weight * d loss / d a. */
int s3_loss_mmd(s3_ctx* ctx, const float* a, int c_a, const float* b, int c_b, int n,
                int64_t n_pos, int c_used, float sigma, float weight, float* loss_out,
                float* d_a);

/* SlicedWassersteinLoss (loss_metrics.py:724-789): n_proj (<= 4096) random unit
 * directions over the n_pos positions, the n * c_used (observation, feature)
 * columns of a, b = (n, n_pos, c_*) projected, each column's projections
 * sorted, mean squared difference of the sorted values.  The directions are a
 * counter-based draw from `seed` (Philox4x32-10 + Box-Muller), regenerated in
 * the backward pass instead of stored; the reference draws new directions per
 * call (tf.random.normal, :777), the caller does the same by passing a new
 * seed.  loss_out = unweighted value; d_a (nullable) += weight * d loss / d a.
 * s3_sw_directions writes the raw (un-normalised) direction matrix
 * [n_proj][n_pos] of a seed — how the parity tests feed the oracle. */
int s3_loss_sliced_wasserstein(s3_ctx* ctx, const float* a, int c_a, const float* b, int c_b,
                               int n, int64_t n_pos, int c_used, int n_proj, uint64_t seed,
                               float weight, float* loss_out, float* d_a);
int s3_sw_directions(s3_ctx* ctx, uint64_t seed, int n_proj, int64_t n_pos, float* out);

/* Time windows of a (outer = n * s1 * s2, t, c) field, as SolarCC.calc_loss
 * slices the hi-res tensors (sup3r/models/solar_cc.py:155-232).
 * s3_time_window: adjoint == 0 copies full[:, t0:t0+len, :] into the contiguous
 * `window`; adjoint != 0 adds scale * window back into that slice of `full`.
 * s3_time_mean: adjoint == 0 writes mean[o][c] = tf.reduce_mean(full[:, t0:t0+len,
 * :], axis=time); adjoint != 0 adds (scale / len) * mean[o][c] to every step of
 * the slice. */
int s3_time_window(s3_ctx* ctx, float* full, int64_t outer, int t, int c, int t0, int len,
                   float* window, int adjoint, float scale);
int s3_time_mean(s3_ctx* ctx, float* full, int64_t outer, int t, int c, int t0, int len,
                 float* mean, int adjoint, float scale);

/* SpatialFftLoss / SpatiotemporalFftLoss (loss_metrics.py:395-485): separable
 * direct DFT, one call per axis over a contiguous (outer, L, inner) view,
 * unnormalised; sign < 0 = forward (tf.signal.fft2d / fft3d), > 0 = adjoint;
 * in_im may be NULL (real input).  s3_specmap: backward == 0 writes out0 =
 * log(1 + w |X|) with w = k1^2 k2^2 (kt^2 if mode3d) over (n, s1, s2, t, c);
 * backward != 0 writes (out0, out1) = g_y * w / (1 + w |X|) * X / |X|. */
int s3_dft_axis(s3_ctx* ctx, const float* in_re, const float* in_im, float* out_re,
                float* out_im, int64_t outer, int L, int64_t inner, int sign);
int s3_specmap(s3_ctx* ctx, int backward, const float* re, const float* im,
               const float* g_y, int n, int s1, int s2, int t, int c, int mode3d,
               float* out0, float* out1);

/* ---- small tensor utilities on the ctx stream --------------------------
 * channel slice/concat used by _combine_loss_input / get_hr_exo_input
 * (abstract.py:415-459) and per-feature affine of norm_input /
 * un_norm_output (abstract.py:197-275). */
int s3_copy_channels(s3_ctx* ctx, const float* src, int c_src, int c0_src,
                     float* dst, int c_dst, int c0_dst, int nc, int64_t n_pos,
                     int accumulate);
int s3_affine_channels(s3_ctx* ctx, const float* src, float* dst, int c,
                       int64_t n_pos, const float* scale_host,
                       const float* shift_host);
int s3_fill(s3_ctx* ctx, float* dst, int64_t n, float value);
/* device -> device copy of a (d0, d1, row_elems) fp32 block between two
 * strided layouts (strides in elements; rows are contiguous): the lo-res chunk
 * window `data[lr_pad_slice]` (strategy.py:474-518) cut out of the resident
 * domain, and the halo crop `hi_res[0][hr_crop_slices]` of the generated chunk
 * (forward_pass.py:272). */
int s3_copy_block(s3_ctx* ctx, const float* src, float* dst, int64_t d0, int64_t d1,
                  int64_t row_elems, int64_t src_stride0, int64_t src_stride1,
                  int64_t dst_stride0, int64_t dst_stride1);
/* device half of ForwardPass._output_check (sup3r/pipeline/forward_pass.py:
 * 384-425: NaNs or a constant output channel mean the chunk failed): for x =
 * (n_chunks, pos_per_chunk, c) writes partial[n_chunks][64][c][3] = (min, max,
 * NaN count) per slab of positions; the caller folds the 64 slabs. */
int s3_chunk_stats(s3_ctx* ctx, const float* x, int n_chunks,
                   int64_t pos_per_chunk, int c, float* partial);
/* everything between the generator's last layer and the delivery of a batch of
 * chunks in one pass: un-normalisation (x * scale + shift per channel, the two
 * roundings of un_norm_output, sup3r/models/abstract.py:240-275; NULL scale /
 * shift: none), the halo crop hi_res[0][hr_crop_slices]
 * (sup3r/pipeline/forward_pass.py:272) and the statistics of _output_check
 * (:384-425; partial as s3_chunk_stats).  y = (n_chunks, dims[0..2], c) fp32,
 * yc = (n_chunks, crop_n[0..2], c).  c must divide 1024 and rows must be
 * 16-byte aligned ((crop_lo[2] c), (crop_n[2] c), (dims[2] c) multiples of 4):
 * S3_EINVAL otherwise, and the caller runs s3_affine_channels / s3_copy_block /
 * s3_chunk_stats instead (bit-identical results). */
int s3_chunk_epilogue(s3_ctx* ctx, const float* y, int n_chunks, const int64_t* dims,
                      const int64_t* crop_lo, const int64_t* crop_n, int c,
                      const float* scale_host, const float* shift_host, float* yc,
                      float* partial);
/* the way INTO a 2-D (spatial) model (ForwardPass._reshape_data_chunk,
 * sup3r/pipeline/forward_pass.py:274-337: np.transpose(data_chunk, (2, 0, 1,
 * 3))) fused with Sup3rGan.norm_input (sup3r/models/abstract.py:197-238): x =
 * (n_chunks, hwt[0], hwt[1], hwt[2], c) fp32 raw chunks, out = (n_chunks *
 * hwt[2], hwt[0], hwt[1], c) = (x - mean) / std per channel with numpy's
 * arithmetic — fp32 when the statistics are fp32 arrays (stats_fp32 = 1), fp64
 * rounded to fp32 otherwise; NULL mean / std: transpose only. */
int s3_chunk_time_first(s3_ctx* ctx, const float* x, int n_chunks, const int64_t* hwt, int c,
                        const double* mean_host, const double* std_host, int stats_fp32,
                        float* out);
/* a time-invariant exo field laid over a chunk's time steps on the device:
 * dst[o][a][r][b] = src[o][a][b] for r < reps (outer x a x b floats in, outer x
 * a x reps x b out).  ForwardPass.pad_source_data (sup3r/pipeline/
 * forward_pass.py:160-186) repeats 3-D exo fields along time on the host
 * (np.repeat) before every chunk; here the chunk carries a zero-stride view and
 * the executor uploads the field once. */
int s3_broadcast_axis(s3_ctx* ctx, const float* src, int64_t outer, int64_t a, int64_t b, int64_t reps,
                      float* dst);
/* between two steps of a MultiStepGan chain (MultiStepGan.generate,
 * sup3r/models/multi_step.py:233-259) on the device, position by position: y =
 * (n_pos, c_src) the NORMALISED output of step i's generator; x = (n_pos, c_sel
 * + n_exo) the normalised input of step i + 1: channel k < c_sel = norm(
 * un_norm(y[map[k]])) — un_norm_output of step i (y * scale[c] + shift[c] per
 * SOURCE channel, abstract.py:240-275), _match_model_input's selection
 * (multi_step.py:148-194), norm_input of step i + 1 ((v - mean[k]) / std[k] per
 * DESTINATION channel, abstract.py:197-238); channel c_sel + j = norm(exo[j]),
 * exo = (n_pos, n_exo) the raw 'input' exo fields _combine_fwp_input appends
 * (interface.py:259-356).  numpy's fp32 arithmetic (one rounding per
 * operation).  NULL scale / shift: no un-normalisation; NULL mean / std: none. */
int s3_step_handover(s3_ctx* ctx, const float* y, int64_t n_pos, int c_src, const int* map_host,
                     int c_sel, const float* scale_host, const float* shift_host, const float* exo,
                     int n_exo, const float* mean_host, const float* std_host, float* x);
/* the same hand-over for a 2-D (spatial) model, whose batch axis is the chunk's
 * time axis (ForwardPass._reshape_data_chunk, sup3r/pipeline/forward_pass.py:
 * 274-337: np.transpose(data_chunk, (2, 0, 1, 3)) in, np.transpose(hi_res, (1,
 * 2, 0, 3)) out, then hi_res[0][hr_crop_slices], :272): y = (n_chunks * thw[0],
 * thw[1], thw[2], c) fp32 as generated, yc = (n_chunks, crop_n[0..2], c) in the
 * chunk's (s1, s2, t) order, crop_lo / crop_n over (s1, s2, t); un-normalised
 * with the two roundings of un_norm_output (sup3r/models/abstract.py:240-275;
 * NULL scale / shift: none).  s3_chunk_stats on yc completes _output_check. */
int s3_chunk_time_last(s3_ctx* ctx, const float* y, int n_chunks, const int64_t* thw,
                       const int64_t* crop_lo, const int64_t* crop_n, int c,
                       const float* scale_host, const float* shift_host, float* yc);
/* placement of a cropped hi-res chunk straight into the caller's host array
 * (the `out[hr_slice] = chunk` of the forward pass, sup3r/pipeline/
 * forward_pass.py:582-673 + the writers' window placement): one pitched
 * device -> host DMA of a contiguous (d0, d1, row_elems) device block into the
 * window dst_host[i * dst_stride0 + j * dst_stride1 + 0..row_elems) (strides
 * in elements).  The host array must be registered once (s3_host_register)
 * for the copy to be asynchronous; `stream` = a hipStream_t or NULL for the
 * context stream. */
int s3_host_register(s3_ctx* ctx, void* ptr, size_t bytes);
int s3_host_unregister(s3_ctx* ctx, void* ptr);
int s3_d2h_window(s3_ctx* ctx, const float* src, float* dst_host, int64_t d0,
                  int64_t d1, int64_t row_elems, int64_t dst_stride0,
                  int64_t dst_stride1, void* stream);
/* the delivery of a batch of cropped hi-res chunks to the host (the
 * `.numpy()` at the end of generate, sup3r/models/abstract.py:1100, as the
 * chunk executor of sup3r/pipeline/forward_pass.py:451-500 needs it: under the
 * NEXT batch's forward): a contiguous device buffer -> pinned (mapped) host
 * memory by a kernel of `blocks` workgroups (<= 0: 16) on `stream` (NULL: the
 * context stream) that writes through PCIe directly — throttled on purpose so
 * that it shares the chip with the compute stream instead of occupying every
 * wave slot like the runtime's full-grid blit copy.  16-byte aligned pointers,
 * bytes % 16 == 0. */
int s3_d2h_stream(s3_ctx* ctx, const void* src, void* dst_host, size_t bytes,
                  void* stream, int blocks);
/* delivery buffers of the chunk executor (what the reference's workers return
 * through the process pool, sup3r/pipeline/forward_pass.py:526-580): pinned,
 * device-mapped host memory.  noncoherent != 0: coarse-grained (the GPU may
 * cache its writes in L2 until the kernel ends; the host must only read after
 * the stream / event that follows the copy has completed — which is how the
 * executor uses it).  s3_d2h_async: plain hipMemcpyAsync(DeviceToHost). */
int s3_host_alloc(s3_ctx* ctx, size_t bytes, int noncoherent, void** out);
int s3_host_free(s3_ctx* ctx, void* ptr);
int s3_d2h_async(s3_ctx* ctx, const void* src, void* dst_host, size_t bytes,
                 void* stream);
/* the same delivery on an SDMA engine (ROCr hsa_amd_memory_async_copy): the
 * shader-side copies above stall the chip's other memory traffic while PCIe
 * drains them, the DMA engines do not.  Ordering is the caller's: call _begin
 * only after the work that produced `src` has COMPLETED (host-synchronised
 * event), read `dst_host` (an s3_host_alloc buffer) only after s3_dma_wait
 * returned S3_OK.  *ticket identifies the copy; s3_dma_wait consumes it
 * (timeout_ms <= 0: no deadline). */
int s3_dma_d2h_begin(s3_ctx* ctx, const void* src, void* dst_host, size_t bytes,
                     uint64_t* ticket);
int s3_dma_wait(s3_ctx* ctx, uint64_t ticket, int timeout_ms);

/* ---- batch transform on the device ----------------------
 * replaces the host numpy of SingleBatchQueue.transform
 * (sup3r/preprocessing/batch_queues/base.py:32-87):
 *   s3_coarsen         = spatial_coarsening (sup3r/utilities/utilities.py:
 *                        406-523, s x s block mean) fused with
 *                        temporal_coarsening (:345-403; t_method one of
 *                        S3_TC_*; t_enhance <= 1 = spatial only).  hr is
 *                        (n, s1, s2, t, c) fp32 (4-D batches: t = 1), lr is
 *                        (n, s1/s, s2/s, t/t_enhance, c).
 *   s3_gaussian_smooth = smooth_data (batch_queues/utilities.py:57-103):
 *                        scipy gaussian_filter(mode='nearest') over the two
 *                        spatial axes of every (obs, t, feature) slice of
 *                        channels whose bit is set in channel_mask; weights =
 *                        the 2*radius+1 normalised taps (host), tmp = scratch
 *                        of the size of x. */
#define S3_TC_SUBSAMPLE 0
#define S3_TC_AVERAGE 1
#define S3_TC_TOTAL 2
#define S3_TC_MAX 3
#define S3_TC_MIN 4
int s3_coarsen(s3_ctx* ctx, const float* hr, int n, int s1, int s2, int t, int c,
               int s_enhance, int t_enhance, int t_method, float* lr);
int s3_gaussian_smooth(s3_ctx* ctx, const float* x, int n, int s1, int s2, int t,
                       int c, const float* weights_host, int radius,
                       unsigned channel_mask, float* tmp, float* y);

/* ---- output epilogue on the device ----------------------
 * replaces the host numpy of OutputHandler._transform_output
 * (sup3r/writers/base.py:304-345) on a hi-res chunk (s1*s2, t, c) fp32:
 *   s3_invert_uv     = invert_uv_single_pair (writers/base.py:283-302) /
 *                      invert_uv (preprocessing/derivers/utilities.py:204-258):
 *                      rotate (u, v) by the grid angle theta(s1, s2) — its
 *                      cos / sin are device tables of s1*s2 floats — and
 *                      overwrite channel u_idx with the windspeed and v_idx
 *                      with the direction in degrees [0, 360).
 *   s3_clip_channels = enforce_limits(nn_fill=False) (sup3r/utilities/
 *                      utilities.py:155-220): per-channel clipping to
 *                      [min, max] (host arrays; +-inf = no limit). */
int s3_invert_uv(s3_ctx* ctx, float* data, int64_t n_sp, int64_t t, int c,
                 int u_idx, int v_idx, const float* cos_theta,
                 const float* sin_theta);
int s3_clip_channels(s3_ctx* ctx, float* data, int c, int64_t n_pos,
                     const float* min_host, const float* max_host);
/*   s3_range_mask / s3_fill_indexed = enforce_limits(nn_fill=True) (same lines)
 *                      + nn_fill_array (utilities.py:55-75): channel ch of the
 *                      (n_pos, c) chunk — mask[p] = 1 where the value is
 *                      outside [lo, hi] or NaN (the reference writes NaN there);
 *                      then data[p, ch] = data[src[p], ch] on the masked
 *                      positions.  src is the flattened index map of
 *                      scipy.ndimage.distance_transform_edt(mask,
 *                      return_indices=True) — the reference's own call, made by
 *                      the host on the boolean mask only; the field stays on
 *                      the device. */
int s3_range_mask(s3_ctx* ctx, const float* data, int c, int ch, int64_t n_pos,
                  float lo, float hi, unsigned char* mask);
int s3_fill_indexed(s3_ctx* ctx, float* data, int c, int ch, int64_t n_pos,
                    const unsigned char* mask, const int* src);

/* ---- data-parallel gradient sync (RCCL over xGMI) -----------------------
 * replaces: the host-side python sum of per-GPU gradient lists,
 * AbstractSingleModel._sum_parallel_grad (abstract.py:785-805): elementwise
 * SUM over ranks of the flat gradient buffer, one ncclAllReduce. */
int s3_comm_unique_id(void* out128); /* rank 0; 128 bytes                  */
int s3_comm_init(s3_ctx* ctx, int rank, int nranks, const void* unique_id128);
int s3_params_allreduce_grads(s3_params* p);
int s3_allreduce_sum(s3_ctx* ctx, float* buf, int64_t n);
/* the communicator as RCCL reports it (ncclCommCount / ncclCommUserRank; 1 / 0
 * without one): what a multi-GPU bench line prints so that the first run on
 * real xGMI verifies itself. */
int s3_comm_info(s3_ctx* ctx, int* n_ranks, int* rank);
/* Overlap with the backward pass: arm the store BEFORE the s3_plan_backward
 * call that finalises its gradients (need_wgrad; the last one when several
 * accumulate).  That call then hands the finished tail of the gradient buffer
 * to RCCL bucket by bucket (>= bucket_bytes each; 0: one bucket) on a second
 * stream behind an event, so the xGMI traffic runs under the remaining
 * gradient kernels; the following s3_params_allreduce_grads only joins the two
 * streams.  Every rank arms the same store with the same bucket size.
 * The backward pass CONSUMES the arming (also when it fails), bucket_bytes < 0
 * disarms, and a context without a communicator is never armed: no later
 * backward pass on the store can issue a collective the other ranks do not.
 * A parameter layout whose offsets do not grow with the op order gets one
 * reduction after the last op instead of buckets. */
int s3_params_arm_allreduce(s3_params* p, int64_t bucket_bytes);
/* replicas must start from identical weights (the reference's towers read ONE
 * set of tf.Variables, abstract.py:827-841): ncclBroadcast of a store buffer
 * (S3_BUF_*) / any fp32 buffer from rank `root` */
int s3_params_broadcast(s3_params* p, int which, int root);
int s3_broadcast(s3_ctx* ctx, float* buf, int64_t n, int root);
void s3_comm_destroy(s3_ctx* ctx);
/* Watchdog of the collectives: bounded host wait for everything enqueued on
 * the context's streams so far (the data-parallel step reads its loss scalars
 * back once per batch, abstract.py:843-914 — this is that wait, with a
 * deadline).  While waiting it polls ncclCommGetAsyncError.  On a timeout or
 * an asynchronous RCCL error the communicator is aborted (ncclCommAbort: a
 * rank stuck in a collective whose peer died never returns otherwise), the
 * context falls back to single-rank state and S3_ETIMEOUT / S3_ERCCL comes
 * back with s3_last_error naming what was pending.  timeout_ms < 0: no
 * deadline (plain s3_ctx_sync + the error poll). */
int s3_comm_wait(s3_ctx* ctx, int64_t timeout_ms);

/* library / build information */
const char* s3_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SUP3R_HIP_H */
