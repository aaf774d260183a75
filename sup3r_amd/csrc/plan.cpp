// Context, parameter store and the shape-specialised plan executor of
// libsup3r_hip.so (host side, C++).  The executor replaces the eager keras
// layer loops of sup3r (abstract.py:1131-1173, base.py:283-313) and
// tf.GradientTape (abstract.py:1230-1237): a fused op list runs on one HIP
// stream out of a statically planned activation arena; the backward pass walks
// the same list in reverse.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "common.h"

struct Param {
  int64_t offset, size;
};

struct s3_params {
  s3_ctx* ctx = nullptr;
  std::vector<Param> p;
  int64_t total = 0;
  float* buf[4] = {nullptr, nullptr, nullptr, nullptr};  // W, G, M, V
  uint64_t version = 1;  // bumped whenever W changes (re-pack trigger)
  // bucketed gradient all-reduce under the backward pass (s3_params_arm_allreduce):
  // [0, reduce_end) of the gradient buffer is not yet handed to RCCL
  bool armed = false;     // the next backward pass that writes G reduces it as it goes
  bool reduced = false;   // ... and has done so: s3_params_allreduce_grads only joins
  int64_t reduce_end = 0, bucket_elems = 0;
  int buckets_issued = 0;
  float* hyper_dev = nullptr;   // the optimizer step's scalars, staged (s3_optimizer_stage)
};

struct TensorRec {
  int64_t dims[5];
  int64_t numel = 0;
  int buffer = -1;      // arena buffer id (-1: external input)
  int alias_root = -1;  // tensor id this one aliases (VIEW)
  float* ptr = nullptr;
  float* gptr = nullptr;  // gradient buffer (training plans)
  bool is_input = false;
  int dtype = 0;        // 0 = fp32, 1 = bf16 (inference plans, bf16 mode)
  size_t bytes() const { return (size_t)numel * (dtype ? 2 : 4); }
};

struct OpRec {
  s3_op_desc d;
  ConvGeom cg;
  GatherGeom gg;
  bool mfma = false;
  ConvIO io;
  void* packed = nullptr;
  uint64_t packed_version = 0;
  // MFMA backward (training plans)
  bool wgrad_mfma = false, dgrad_mfma = false, wgrad_bf16 = false, wgrad_c2 = false, wgrad_bf16_gen = false, wgrad_bf16_2d = false, wgrad_tail = false;
  bool dgrad_valid = false;    // dgrad_mfma of a valid-padded conv: no frame / fold
  bool dgrad_frame16 = false;  // the persistent kernel writes the padded frame as bf16
  bool use16 = false;          // data gradient stages the bf16 copy of dPre its mask pass leaves behind
  bool dgrad_gen = false;      // dgrad_mfma on the logical-axes tile kernel (2-D nets, few time steps, any channels)
  bool dgrad_c2 = false;       // few-channel hi-res conv: LDS-halo dgrad
  bool dgrad_s2 = false;       // stride-2 valid conv, C_out = 32: residue classes on an LDS halo
  int mask_prod = -1;          // producer conv of in0 whose activation adjoint is fused into this conv's dgrad store / fold
  int in_prod = -1;            // producer conv of in0 (any number of consumers), -1: not a conv
  bool dgrad_fewch = false;    // C_out <= 4 'same' conv: dgrad = few-channel forward conv over the frame
  bool dgrad_chunked = false;  // 64 -> C_out > 64 'same' conv: 64-channel slices of dPre through the tile kernel
  void* dgc_wbf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool halo32 = false;         // C_in = 32 stride-1 conv: LDS-halo forward
  bool halo_s2 = false;        // C_in = 32 stride-2 valid conv: LDS-halo forward (bf16 cells in)
  bool tail_x3 = false;        // BF16X3 plans: banded split-bf16 MFMA tail (8 -> 2, fp32 in / out)
  void* h32_w = nullptr;
  uint64_t h32_version = 0;
  void* dc2_w = nullptr;
  int64_t dc2_version = -1;
  ConvGeom dg;                 // geometry of the dgrad-as-conv launch
  int rep_src = -1;            // conv: tensor read through a fused temporal repeat (cg.in_rep)
  int res_src = -1;            // ... and the residual (cg.res_rep)
  int exo_src = -1;            // conv behind a fused-away Sup3rConcat: the exogenous field (cg.w_cin)
  int res2_src = -1;           // conv that absorbed the skip add behind it: the add's other operand (cg.res2)
  void* sign_bytes = nullptr;  // training: activation sign bytes next to the output (conv_dgrad_s2's mask)
  bool fused_away = false;     // repeat op absorbed by its consumer conv: no launch
  float* dg_w32 = nullptr;     // flipped / transposed fp32 filter
  void* dg_wbf = nullptr;      // its bf16 slabs (bf16 mode)
  uint64_t dg_version = 0;
  // general gather-MFMA conv (strided / valid-padded, C_in % 32 == 0)
  bool gconv = false, gconv_dgrad = false, wgrad_gen = false;
  void* gc_w = nullptr;        // bf16 [tap][co][ci]
  void* gc_wt = nullptr;       // bf16 [tap][ci][co] (data gradient)
  uint64_t gc_version = 0, gct_version = 0;
  // few-positions GEMM path
  bool fewpos = false;
  bool fewpos_wgrad = false;   // few positions, small filter: only the weight gradient takes the fewpos kernel
  bool fp_mfma = false;        // the fewpos launches of this conv are the one-launch fp32-MFMA kernels
  bool fp_wg_mfma = false;     // ... its weight gradient at least (fewpos_wgrad ops)
  float* fp_wt = nullptr;      // [tap][co][ci] transposed filter (dgrad)
  uint64_t fp_version = 0;
};

struct s3_plan {
  s3_ctx* ctx = nullptr;
  S3Options opt;              // snapshot of the options this plan was created with
  s3_params* params = nullptr;
  std::vector<TensorRec> t;
  std::vector<OpRec> ops;
  std::vector<int32_t> inputs;
  int32_t output = -1;
  int precision = S3_PREC_F32;
  int training = 0;
  // s3_plan_forward_window: op index whose conv runs over win_geom (-1: none) + its affine
  int win_op = -1;
  ConvGeom win_geom;
  const float* win_aff = nullptr;
  std::vector<float*> buffers;
  std::vector<size_t> buffer_bytes;
  std::vector<void*> owned;  // every hipMalloc of this plan
  float* dpre = nullptr;      // conv/dense epilogue-adjoint workspace
  void* dpre16 = nullptr;     // its bf16 copy (mask pass of a conv with use16)
  size_t dpre16_bytes = 0;
  int dpre16_for = -1;        // tensor root whose finished gradient = dPre of its producer is in dpre16, -1: none
  bool dpre16_only = false;   // ... and ONLY there (bf16-only frame fold); false: the fp32 tensor is valid too
  float* gtmp = nullptr;      // gradient staging when a tensor has >1 consumer
  float* wg_partial = nullptr;
  size_t wg_partial_bytes = 0;
  float* dxp = nullptr;       // padded-frame data gradient of the MFMA dgrad
  float* fp_partial = nullptr;   // per-tap partials of the few-positions path
  size_t fp_partial_bytes = 0;
  size_t total_bytes = 0;
  bool forward_done = false;
  std::vector<hipEvent_t> prof_ev;  // prof_cap * (n_ops + 1)
  int prof_cap = 0, prof_n = 0;
  std::vector<char> gwritten;          // 0 none, 1 in gptr, 2 = one contribution, aliased (gsrc)
  std::vector<const float*> gsrc;      // the aliased first contribution (a finished gradient buffer)
  float* bsum = nullptr;               // channel sums left by a frame fold (bias gradient of the producer)
  int bsum_for = -1, bsum_nblk = 0;    // tensor root they belong to (-1: none), slabs
  float* bsum2 = nullptr;              // channel sums left by a conv's own mask pass (consumed at once)
  std::vector<char> premasked;   // tensor gradient already carries its producer's activation adjoint
  // hipGraph replay of the forward op list (inference plans): inputs are
  // copied into plan-owned staging buffers so every pointer inside the
  // captured graph is fixed; re-captured when the weights change
  std::vector<float*> in_stage;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t cap_stream = nullptr;
  uint64_t graph_version = 0;
  int eager_forwards = 0;
  bool graph_off = false;
  Fused2dPlan* fused2d = nullptr;   // whole-network kernel (small 2-D inference plans)
  // batched filter re-pack (bf16 plans): job tables on the device, built once
  S3PackJob* pack_fwd = nullptr;
  S3PackJob* pack_bwd = nullptr;
  std::vector<int> pack_fwd_ops, pack_bwd_ops;
  int pack_fwd_ct = 1, pack_bwd_ct = 1;
  bool pack_built = false;
};

static int plan_alloc(s3_plan* pl, void** out, size_t bytes) {
  s3_ctx* ctx = pl->ctx;
  if (bytes == 0) bytes = 16;
  S3_HIP(ctx, hipMalloc(out, bytes));
  // every plan buffer starts zeroed (hipMalloc hands back stale bytes of freed
  // buffers: padding rows / halo borders that no kernel writes must not depend
  // on what ran before).  SUP3R_AMD_POISON_ALLOC=1 fills all-ones bytes instead
  // (NaN as fp32 and as bf16): a debugging aid that makes any read of a plan
  // buffer before its first write show up in the results.
  S3_HIP(ctx, hipMemsetAsync(*out, s3_opt_has(S3O_POISON_ALLOC) ? 0xFF : 0, bytes, ctx->stream));
  pl->owned.push_back(*out);
  pl->total_bytes += bytes;
  return S3_OK;
}

// ------------------------------------------------------------------ options
thread_local const S3Options* s3_active_options = nullptr;

static const char* const kOptionNames[S3O_COUNT] = {
#define X(n) #n,
    S3_OPTION_LIST(X)
#undef X
};

const char* s3_option_name(int id) { return (id >= 0 && id < S3O_COUNT) ? kOptionNames[id] : nullptr; }

int s3_option_id(const char* name) {
  if (!name) return -1;
  if (!strncmp(name, "SUP3R_AMD_", 10)) name += 10;
  for (int i = 0; i < S3O_COUNT; ++i)
    if (!strcmp(name, kOptionNames[i])) return i;
  return -1;
}

// initial defaults of a context: the SUP3R_AMD_<NAME> variables as they are
// when the context is created (never read again afterwards)
static void options_from_env(S3Options& o) {
  for (int i = 0; i < S3O_COUNT; ++i) {
    const std::string var = std::string("SUP3R_AMD_") + kOptionNames[i];
    const char* v = getenv(var.c_str());
    if (v) { o.has[i] = true; o.v[i] = (int32_t)atoll(v); }
  }
}

static int apply_options(s3_ctx* ctx, S3Options& o, const s3_plan_options* opt) {
  if (!opt) return S3_OK;
  for (int i = 0; i < opt->n; ++i) {
    const int id = s3_option_id(opt->names ? opt->names[i] : nullptr);
    if (id < 0) S3_FAIL(ctx, S3_EINVAL, std::string("unknown option \"") + (opt->names && opt->names[i] ? opt->names[i] : "(null)") + "\"");
    if (opt->values[i] == S3_OPTION_UNSET) { o.has[id] = false; o.v[id] = 0; }
    else { o.has[id] = true; o.v[id] = opt->values[i]; }
  }
  return S3_OK;
}

extern "C" int s3_ctx_set_option(s3_ctx* ctx, const char* name, int32_t value) {
  if (!ctx) return S3_EINVAL;
  const char* names[1] = {name};
  const int32_t values[1] = {value};
  s3_plan_options o = {1, names, values};
  return apply_options(ctx, ctx->opt, &o);
}

extern "C" int s3_ctx_get_option(const s3_ctx* ctx, const char* name, int32_t* value) {
  if (!ctx) return S3_EINVAL;
  const int id = s3_option_id(name);
  if (id < 0) return S3_EINVAL;
  if (value) *value = ctx->opt.v[id];
  return ctx->opt.has[id] ? 1 : 0;
}

extern "C" const char* s3_option_name_at(int index) { return s3_option_name(index); }

// ------------------------------------------------------------------ context
extern "C" int s3_ctx_create(int device_id, void* stream, int create_stream,
                             s3_ctx** out) {
  if (!out) return S3_EINVAL;
  s3_ctx* ctx = new s3_ctx();
  ctx->device = device_id;
  options_from_env(ctx->opt);
  hipError_t e = hipSetDevice(device_id);
  if (e != hipSuccess) {
    // keep the object so the caller can read the message
    ctx->err = std::string("hipSetDevice: ") + hipGetErrorString(e);
    *out = ctx;
    return S3_EHIP;
  }
  if (!create_stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      ctx->err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
      *out = ctx;
      return S3_EHIP;
    }
    ctx->own_stream = true;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    ctx->num_cu = prop.multiProcessorCount;
    // The library is compiled for gfx950 only and its persistent kernels are sized
    // for that part's 160 KB of LDS per workgroup (up to 163,072 B): say so here,
    // once, instead of failing at some kernel's first launch on anything else.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      ctx->err = std::string("sup3r_amd is built for gfx950 (MI355X) only; device ") + std::to_string(device_id) +
                 " is " + prop.gcnArchName;
      *out = ctx;
      return S3_ESTATE;
    }
  }
  *out = ctx;
  return S3_OK;
}

extern "C" void s3_ctx_destroy(s3_ctx* ctx) {
  if (!ctx) return;
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  for (void* p : ctx->retired) (void)hipFree(p);   // scratch blocks outgrown while a graph held them
  ctx->retired.clear();
  if (ctx->capturing) (void)s3_capture_abort(ctx);
  if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
  if (ctx->wg_stream) (void)hipStreamDestroy(ctx->wg_stream);
  for (int k = 0; k < 2; ++k)
    if (ctx->wg_ev[k]) (void)hipEventDestroy(ctx->wg_ev[k]);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int64_t s3_ctx_stat(const s3_ctx* ctx, int which) {
  if (!ctx || which < 0 || which >= S3_STAT_COUNT) return -1;
  return ctx->stat[which];
}

extern "C" const char* s3_last_error(const s3_ctx* ctx) {
  return ctx ? ctx->err.c_str() : "null context";
}

extern "C" int s3_ctx_sync(s3_ctx* ctx) {
  if (!ctx) return S3_EINVAL;
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return S3_OK;
}

extern "C" void* s3_ctx_stream(s3_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" const char* s3_version(void) { return "sup3r_hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------- params
extern "C" int s3_params_create(s3_ctx* ctx, int n, const int64_t* sizes,
                                s3_params** out) {
  if (!ctx || !out || n < 0) return S3_EINVAL;
  s3_params* p = new s3_params();
  p->ctx = ctx;
  int64_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (sizes[i] <= 0) { delete p; S3_FAIL(ctx, S3_EINVAL, "params_create: non-positive size"); }
    p->p.push_back({off, sizes[i]});
    off += (sizes[i] + 3) / 4 * 4;  // keep every tensor 16-B aligned
  }
  p->total = off;
  size_t bytes = (size_t)(off > 0 ? off : 4) * sizeof(float);
  for (int k = 0; k < 4; ++k) {
    hipError_t e = hipMalloc((void**)&p->buf[k], bytes);
    if (e != hipSuccess) {
      ctx->err = std::string("params hipMalloc: ") + hipGetErrorString(e);
      for (int q = 0; q < k; ++q) (void)hipFree(p->buf[q]);
      delete p;
      return S3_ENOMEM;
    }
    S3_HIP(ctx, hipMemsetAsync(p->buf[k], 0, bytes, ctx->stream));
  }
  *out = p;
  return S3_OK;
}

extern "C" void s3_params_destroy(s3_params* p) {
  if (!p) return;
  (void)hipStreamSynchronize(p->ctx->stream);
  for (int k = 0; k < 4; ++k)
    if (p->buf[k]) (void)hipFree(p->buf[k]);
  if (p->hyper_dev) (void)hipFree(p->hyper_dev);
  delete p;
}

extern "C" int64_t s3_params_total(const s3_params* p) { return p ? p->total : 0; }

static int params_check(s3_params* p, int which, int idx) {
  if (!p) return S3_EINVAL;
  if (which < 0 || which > 3 || idx < 0 || idx >= (int)p->p.size())
    S3_FAIL(p->ctx, S3_EINVAL, "params: bad buffer / index");
  return S3_OK;
}

extern "C" int s3_params_set(s3_params* p, int which, int idx, const float* host) {
  int rc = params_check(p, which, idx);
  if (rc) return rc;
  s3_ctx* ctx = p->ctx;
  S3_HIP(ctx, hipMemcpyAsync(p->buf[which] + p->p[idx].offset, host,
                             p->p[idx].size * sizeof(float),
                             hipMemcpyHostToDevice, ctx->stream));
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (which == S3_BUF_W) p->version++;
  return S3_OK;
}

extern "C" int s3_params_get(s3_params* p, int which, int idx, float* host) {
  int rc = params_check(p, which, idx);
  if (rc) return rc;
  s3_ctx* ctx = p->ctx;
  S3_HIP(ctx, hipMemcpyAsync(host, p->buf[which] + p->p[idx].offset,
                             p->p[idx].size * sizeof(float),
                             hipMemcpyDeviceToHost, ctx->stream));
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return S3_OK;
}

extern "C" void* s3_params_dptr(s3_params* p, int which, int idx) {
  if (!p || which < 0 || which > 3) return nullptr;
  if (idx < 0) return p->buf[which];
  if (idx >= (int)p->p.size()) return nullptr;
  return p->buf[which] + p->p[idx].offset;
}


extern "C" uint64_t s3_params_version(const s3_params* p) { return p ? p->version : 0; }

extern "C" int s3_params_zero_grad(s3_params* p) {
  if (!p) return S3_EINVAL;
  s3_ctx* ctx = p->ctx;
  S3_HIP(ctx, hipMemsetAsync(p->buf[S3_BUF_G], 0, (size_t)p->total * sizeof(float), ctx->stream));
  return S3_OK;
}

extern "C" int s3_params_mean_abs(s3_params* p, int which, int idx, float* host_out) {
  int rc = params_check(p, which, idx);
  if (rc) return rc;
  s3_ctx* ctx = p->ctx;
  rc = ensure_scratch(ctx, 1 << 20);
  if (rc) return rc;
  // result lands in the last float of the 1 MiB minimum scratch
  float* out_dev = ctx->scratch + (ctx->scratch_bytes / sizeof(float)) - 1;
  rc = launch_mean_abs(ctx, p->buf[which] + p->p[idx].offset, p->p[idx].size, out_dev);
  if (rc) return rc;
  S3_HIP(ctx, hipMemcpyAsync(host_out, out_dev, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return S3_OK;
}

extern "C" int s3_adam_step(s3_params* p, float lr, float beta1, float beta2,
                            float eps, int64_t t) {
  const double hp[4] = {lr, beta1, beta2, eps};
  return s3_optimizer_step(p, S3_OPT_ADAM, hp, 4, t);
}

extern "C" int s3_params_arm_allreduce(s3_params* p, int64_t bucket_bytes) {
  if (!p) return S3_EINVAL;
  // bucket_bytes < 0 disarms (the caller's try / finally around the backward
  // pass it armed for); without a communicator there is nothing to overlap —
  // an armed store would only leave a flag behind for a later backward pass
  p->reduced = false;
  if (bucket_bytes < 0 || !p->ctx || !p->ctx->comm) {
    p->armed = false;
    p->reduce_end = 0;
    p->buckets_issued = 0;
    return S3_OK;
  }
  p->armed = true;
  p->reduce_end = p->total;
  p->bucket_elems = bucket_bytes > 0 ? bucket_bytes / (int64_t)sizeof(float) : p->total;
  p->buckets_issued = 0;
  return S3_OK;
}

// called by s3_params_allreduce_grads (comm.cpp): 1 = the armed, bucketed
// reduction covered the whole buffer (the caller only joins the streams),
// 0 = nothing was armed (reduce the whole buffer now), -1 = armed but the
// backward pass did not reach the start of the buffer
extern "C" S3_INTERNAL int s3_params_take_armed(s3_params* p, int* n_buckets) {
  if (!p) return 0;
  if (n_buckets) *n_buckets = p->buckets_issued;
  if (p->reduced) {        // an armed backward pass covered the buffer
    p->reduced = false;
    return 1;
  }
  if (!p->armed) return 0;
  p->armed = false;        // armed, but no backward pass wrote the gradients
  return -1;
}

// Hyper-parameters arrive as doubles (they are Python floats in the keras
// configs) and are cast the way keras casts them: `1 - beta` is evaluated in
// double and THEN rounded to fp32 (keras multiplies the fp32 tensor by the
// Python scalar 1 - beta), the powers beta^t in fp32 (tf.pow of the cast
// beta).  fp32(1) - fp32(0.999) would be off by 4.7e-5 of itself.
// the step's scalars h[0..4] as the kernels take them
static int optimizer_scalars(s3_ctx* ctx, int kind, const double* hp, int n_hp, int64_t t, float* h) {
  for (int q = 0; q < 5; ++q) h[q] = 0.f;
  auto need = [&](int n) { return n_hp >= n; };
  switch (kind) {
    case S3_OPT_ADAM: {
      if (!need(4)) S3_FAIL(ctx, S3_EINVAL, "optimizer_step(Adam): {lr, beta_1, beta_2, epsilon}");
      const float b1p = powf((float)hp[1], (float)t), b2p = powf((float)hp[2], (float)t);
      h[0] = (float)hp[0] * sqrtf(1.f - b2p) / (1.f - b1p);
      h[1] = (float)(1.0 - hp[1]); h[2] = (float)(1.0 - hp[2]); h[3] = (float)hp[3];
      break;
    }
    case S3_OPT_SGD:
      if (!need(3)) S3_FAIL(ctx, S3_EINVAL, "optimizer_step(SGD): {lr, momentum, nesterov}");
      h[0] = (float)hp[0]; h[1] = (float)hp[1]; h[2] = (float)hp[2];
      break;
    case S3_OPT_RMSPROP:
      if (!need(4)) S3_FAIL(ctx, S3_EINVAL, "optimizer_step(RMSprop): {lr, rho, momentum, epsilon}");
      h[0] = (float)hp[0]; h[1] = (float)hp[1]; h[2] = (float)hp[2]; h[3] = (float)hp[3];
      h[4] = (float)(1.0 - hp[1]);
      break;
    case S3_OPT_ADAGRAD:
      if (!need(3)) S3_FAIL(ctx, S3_EINVAL, "optimizer_step(Adagrad): {lr, epsilon, initial_accumulator_value}");
      h[0] = (float)hp[0]; h[1] = (float)hp[1];
      break;
    case S3_OPT_ADAMAX: {
      if (!need(4)) S3_FAIL(ctx, S3_EINVAL, "optimizer_step(Adamax): {lr, beta_1, beta_2, epsilon}");
      const float b1p = powf((float)hp[1], (float)t);
      h[0] = (float)hp[0] / (1.f - b1p); h[1] = (float)(1.0 - hp[1]); h[2] = (float)hp[2]; h[3] = (float)hp[3];
      break;
    }
    case S3_OPT_ADAMW: {
      if (!need(5)) S3_FAIL(ctx, S3_EINVAL, "optimizer_step(AdamW): {lr, beta_1, beta_2, epsilon, weight_decay}");
      const float b1p = powf((float)hp[1], (float)t), b2p = powf((float)hp[2], (float)t);
      h[0] = (float)hp[0] * sqrtf(1.f - b2p) / (1.f - b1p);
      h[1] = (float)(1.0 - hp[1]); h[2] = (float)(1.0 - hp[2]); h[3] = (float)hp[3];
      h[4] = (float)hp[4] * (float)hp[0];
      break;
    }
    default: S3_FAIL(ctx, S3_EINVAL, "optimizer_step: unknown optimizer kind");
  }
  return S3_OK;
}

static int optimizer_launch(s3_params* p, int kind, const float* h, const float* h_dev) {
  s3_ctx* ctx = p->ctx;
  int rc;
  if (kind == S3_OPT_ADAM)
    rc = launch_adam(ctx, p->buf[S3_BUF_W], p->buf[S3_BUF_G], p->buf[S3_BUF_M], p->buf[S3_BUF_V], p->total,
                     h[0], h[1], h[2], h[3], h_dev);
  else
    rc = launch_optimizer(ctx, kind, p->buf[S3_BUF_W], p->buf[S3_BUF_G], p->buf[S3_BUF_M], p->buf[S3_BUF_V],
                          p->total, h, h_dev);
  if (rc) return rc;
  p->version++;
  return S3_OK;
}

extern "C" int s3_optimizer_step(s3_params* p, int kind, const double* hp, int n_hp, int64_t t) {
  if (!p || !hp || t < 1) return S3_EINVAL;
  s3_ctx* ctx = p->ctx;
  float h[5];
  int rc = optimizer_scalars(ctx, kind, hp, n_hp, t, h);
  if (rc) return rc;
  if (kind == S3_OPT_ADAGRAD && t == 1) {   // keras creates the accumulator filled with its initial value
    rc = launch_fill(ctx, p->buf[S3_BUF_V], p->total, (float)hp[2]);
    if (rc) return rc;
  }
  return optimizer_launch(p, kind, h, nullptr);
}

// The same step in two halves, for a captured graph: the scalars of step t are
// written to the device by a 1-thread launch OUTSIDE the graph (kernel
// arguments: no host buffer has to outlive the call), the update launch inside
// it reads them from there and is identical every step.
extern "C" int s3_optimizer_stage(s3_params* p, int kind, const double* hp, int n_hp, int64_t t) {
  if (!p || !hp || t < 1) return S3_EINVAL;
  s3_ctx* ctx = p->ctx;
  if (kind == S3_OPT_ADAGRAD && t == 1)
    S3_FAIL(ctx, S3_EINVAL, "optimizer_stage(Adagrad): the first step creates the accumulator, run it with s3_optimizer_step");
  float h[5];
  int rc = optimizer_scalars(ctx, kind, hp, n_hp, t, h);
  if (rc) return rc;
  if (!p->hyper_dev) S3_HIP(ctx, hipMalloc((void**)&p->hyper_dev, 8 * sizeof(float)));
  return launch_stage_hyper(ctx, p->hyper_dev, h);
}

extern "C" int s3_optimizer_step_staged(s3_params* p, int kind) {
  if (!p) return S3_EINVAL;
  s3_ctx* ctx = p->ctx;
  // (recorded before the first stage: the replay stages before it launches)
  if (!p->hyper_dev) S3_HIP(ctx, hipMalloc((void**)&p->hyper_dev, 8 * sizeof(float)));
  if (kind < S3_OPT_ADAM || kind > S3_OPT_ADAMW) S3_FAIL(ctx, S3_EINVAL, "optimizer_step: unknown optimizer kind");
  const float h[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  return optimizer_launch(p, kind, h, p->hyper_dev);
}

// the weights changed behind the host's back (a replayed graph stepped the
// optimizer): packed filter images of every plan are stale
extern "C" int s3_params_touch(s3_params* p) {
  if (!p) return S3_EINVAL;
  p->version++;
  return S3_OK;
}

// ------------------------------------------------------------ stream capture
// A launch-bound step (the C1 training step is ~650 launches of a few
// microseconds each) recorded once and replayed as ONE hipGraphLaunch.  Between
// begin and end every launch of this context goes to a non-blocking side
// stream that records instead of executing; what is recorded must be the same
// every step: static pointers (the caller keeps every buffer of the step
// alive), no host read-back, no collective, step-dependent scalars staged
// (s3_optimizer_stage).  A call that cannot be captured fails the capture; the
// caller then runs eagerly.
struct s3_graph {
  s3_ctx* ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  size_t n_nodes = 0;
};

extern "C" int s3_capture_begin(s3_ctx* ctx) {
  if (!ctx) return S3_EINVAL;
  if (ctx->capturing) S3_FAIL(ctx, S3_EINVAL, "capture_begin: already capturing");
  if (ctx->comm) S3_FAIL(ctx, S3_EINVAL, "capture_begin: not with a communicator (collectives are not captured)");
  if (!ctx->cap_stream) S3_HIP(ctx, hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
  // what was enqueued so far runs before anything the capture stream does later
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipError_t be = hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeRelaxed);
  if (be != hipSuccess) {
    // a capture stream left in a broken state by an earlier, failed recording
    // must not poison every later one: drop it and try once on a fresh stream
    (void)hipGetLastError();
    (void)hipStreamDestroy(ctx->cap_stream);
    ctx->cap_stream = nullptr;
    S3_HIP(ctx, hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
    S3_HIP(ctx, hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeRelaxed));
  }
  ctx->saved_stream = ctx->stream;
  ctx->stream = ctx->cap_stream;
  ctx->capturing = true;
  return S3_OK;
}

static int capture_stop(s3_ctx* ctx, hipGraph_t* g) {
  hipError_t e = hipStreamEndCapture(ctx->cap_stream, g);
  ctx->stream = ctx->saved_stream;
  ctx->capturing = false;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ctx->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e);
    // (the next recording starts on a fresh stream)
    if (ctx->cap_stream) { (void)hipStreamDestroy(ctx->cap_stream); ctx->cap_stream = nullptr; }
    (void)hipGetLastError();
    return S3_EHIP;
  }
  return S3_OK;
}

extern "C" int s3_capture_abort(s3_ctx* ctx) {
  if (!ctx) return S3_EINVAL;
  if (!ctx->capturing) return S3_OK;
  hipGraph_t g = nullptr;
  const std::string keep = ctx->err;
  (void)capture_stop(ctx, &g);
  if (g) (void)hipGraphDestroy(g);
  ctx->err = keep;
  return S3_OK;
}

extern "C" int s3_capture_end(s3_ctx* ctx, s3_graph** out) {
  if (!ctx || !out) return S3_EINVAL;
  if (!ctx->capturing) S3_FAIL(ctx, S3_EINVAL, "capture_end: not capturing");
  hipGraph_t g = nullptr;
  int rc = capture_stop(ctx, &g);
  if (rc) return rc;
  if (!g) S3_FAIL(ctx, S3_EHIP, "capture_end: empty graph");
  s3_graph* G = new s3_graph();
  G->ctx = ctx;
  G->graph = g;
  hipError_t e = hipGraphInstantiate(&G->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    delete G;
    ctx->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
    return S3_EHIP;
  }
  (void)hipGraphGetNodes(g, nullptr, &G->n_nodes);
  ctx->graphs_made = true;
  *out = G;
  return S3_OK;
}

extern "C" int s3_graph_launch(s3_graph* g) {
  if (!g || !g->exec) return S3_EINVAL;
  s3_ctx* ctx = g->ctx;
  if (ctx->capturing) S3_FAIL(ctx, S3_EINVAL, "graph_launch: inside a capture");
  S3_HIP(ctx, hipGraphLaunch(g->exec, ctx->stream));
  return S3_OK;
}

extern "C" int64_t s3_graph_nodes(const s3_graph* g) { return g ? (int64_t)g->n_nodes : -1; }

extern "C" void s3_graph_destroy(s3_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}

// --------------------------------------------------------------------- plan
static int64_t numel5(const int64_t* d) { return d[0] * d[1] * d[2] * d[3] * d[4]; }

static int root_of(const s3_plan* pl, int t) {
  while (pl->t[t].alias_root >= 0) t = pl->t[t].alias_root;
  return t;
}

static void fill_conv_geom(const s3_plan* pl, const s3_op_desc& d, ConvGeom& g) {
  const TensorRec& in = pl->t[d.in0];
  const TensorRec& out = pl->t[d.out];
  g.N = (int)in.dims[0];
  for (int i = 0; i < 3; ++i) {
    g.D[i] = (int)in.dims[1 + i];
    g.k[i] = d.k[i]; g.s[i] = d.stride[i]; g.lo[i] = d.lo[i];
  }
  const int b = d.d2s < 1 ? 1 : d.d2s;
  g.O[0] = (int)out.dims[1] / b; g.O[1] = (int)out.dims[2] / b; g.O[2] = (int)out.dims[3];
  g.Cin = (int)in.dims[4];
  g.Cout = (int)out.dims[4] * b * b;
  g.pad_mode = d.pad_mode; g.act = d.act; g.alpha = d.alpha; g.d2s = b;
}

static void fill_gather_geom(const s3_plan* pl, const s3_op_desc& d, GatherGeom& g) {
  const TensorRec& in = pl->t[d.in0];
  const TensorRec& out = pl->t[d.out];
  g.kind = d.kind; g.N = (int)in.dims[0];
  for (int i = 0; i < 3; ++i) {
    g.Di[i] = (int)in.dims[1 + i]; g.Do[i] = (int)out.dims[1 + i]; g.lo[i] = d.lo[i];
  }
  g.Ci = (int)in.dims[4]; g.Co = (int)out.dims[4];
  g.pad_mode = d.pad_mode; g.rep = d.rep; g.d2s = d.d2s; g.c_off = 0;
}

extern "C" int s3_plan_create(s3_ctx* ctx, s3_params* params,
                              const s3_tensor_desc* tensors, int n_tensors,
                              const s3_op_desc* ops, int n_ops,
                              const int32_t* inputs, int n_inputs,
                              int32_t output, int precision, int training,
                              s3_plan** out) {
  return s3_plan_create_opt(ctx, params, tensors, n_tensors, ops, n_ops, inputs, n_inputs, output,
                            precision, training, nullptr, out);
}

extern "C" int s3_plan_create_opt(s3_ctx* ctx, s3_params* params,
                                  const s3_tensor_desc* tensors, int n_tensors,
                                  const s3_op_desc* ops, int n_ops,
                                  const int32_t* inputs, int n_inputs,
                                  int32_t output, int precision, int training,
                                  const s3_plan_options* options, s3_plan** out) {
  if (!ctx || !params || !tensors || !ops || !out) return S3_EINVAL;
  if (output < 0 || output >= n_tensors) S3_FAIL(ctx, S3_EINVAL, "plan: bad output id");
  S3Options plan_opt = ctx->opt;
  if (int orc = apply_options(ctx, plan_opt, options)) return orc;
  s3_plan* pl = new s3_plan();
  pl->opt = plan_opt;
  S3OptScope opt_scope(&pl->opt);
  pl->ctx = ctx; pl->params = params; pl->precision = precision;
  pl->training = training; pl->output = output;
  pl->t.resize(n_tensors);
  for (int i = 0; i < n_tensors; ++i) {
    memcpy(pl->t[i].dims, tensors[i].dims, sizeof(int64_t) * 5);
    pl->t[i].numel = numel5(tensors[i].dims);
    if (pl->t[i].numel <= 0) { delete pl; S3_FAIL(ctx, S3_EINVAL, "plan: empty tensor"); }
  }
  for (int i = 0; i < n_inputs; ++i) {
    if (inputs[i] < 0 || inputs[i] >= n_tensors) { delete pl; S3_FAIL(ctx, S3_EINVAL, "plan: bad input id"); }
    pl->inputs.push_back(inputs[i]);
    pl->t[inputs[i]].is_input = true;
  }
  auto bad = [&](const char* m) { ctx->err = m; delete pl; return S3_EINVAL; };
  // a launch-bound plan: no tensor has more than 4 096 positions (the C1 / toy
  // training shapes) — every step of it is a chain of ~5 us launches, and the
  // few-channel head / tail convs go to the one-launch fewpos kernels as well
  bool plan_tiny = true;
  for (int i = 0; i < n_tensors; ++i)
    if (pl->t[i].numel / std::max<int64_t>(1, pl->t[i].dims[4]) > 4096) plan_tiny = false;
  const int np = (int)params->p.size();
  pl->ops.resize(n_ops);
  size_t max_dpre = 0, max_partial = 0, max_t = 0, max_dxp = 0, max_fp = 0;
  for (int i = 0; i < n_ops; ++i) {
    OpRec& o = pl->ops[i];
    o.d = ops[i];
    const s3_op_desc& d = o.d;
    auto tid_ok = [&](int id, bool opt) { return (opt && id < 0) || (id >= 0 && id < n_tensors); };
    if (!tid_ok(d.in0, false) || !tid_ok(d.out, false) || !tid_ok(d.in1, true) || !tid_ok(d.res, true))
      return bad("plan: op references a bad tensor id");
    if (d.w >= np || d.b >= np) return bad("plan: op references a bad parameter id");
    switch (d.kind) {
      case S3_OP_CONV: {
        if (d.w < 0) return bad("plan: conv without weights");
        fill_conv_geom(pl, d, o.cg);
        const ConvGeom& g = o.cg;
        int64_t wsz = (int64_t)g.k[0] * g.k[1] * g.k[2] * g.Cin * g.Cout;
        if (params->p[d.w].size != wsz) return bad("plan: conv weight size mismatch");
        if (d.b >= 0 && params->p[d.b].size != g.Cout) return bad("plan: conv bias size mismatch");
        const TensorRec& ot = pl->t[d.out];
        if (ot.dims[0] != g.N || ot.dims[4] * g.d2s * g.d2s != g.Cout)
          return bad("plan: conv output shape mismatch");
        for (int q = 0; q < 3; ++q) {
          if (g.s[q] < 1 || g.k[q] < 1) return bad("plan: bad conv geometry");
          int mx = (g.O[q] - 1) * g.s[q] + g.k[q] - 1 - g.lo[q];
          if (g.pad_mode == S3_PAD_REFLECT &&
              (g.lo[q] > g.D[q] - 1 || mx > 2 * (g.D[q] - 1)))
            return bad("plan: reflect padding exceeds the tensor extent");
        }
        if (d.res >= 0 && pl->t[d.res].numel != ot.numel) return bad("plan: residual shape mismatch");
        o.mfma = conv_mfma_supported(g, precision);
        // Round 6: a trunk-geometry conv (64 -> 64 k, 3 x 3 x 3) over <= 1 024 positions in
        // a TRAINING plan — the lo-res stack of the reference's test shapes, BASELINE
        // configs 4 / 5: lr (N, 4, 4, 4, 2) — leaves the halo-tile family: two or four
        // of its tiles keep two or four CUs busy (17 us forward, 17 us data gradient,
        // 40 + 5 us for the weight gradient whose every workgroup holds the whole 27 x
        // 64 x 64 tile), where the one-launch fewpos kernels split the work over (tap,
        // channel block) and take ~8 us for forward and ~8 us for both gradients.
        // (Training plans only: an inference plan's kernels do not depend on the batch.)
        const int64_t P_out = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
        // (the 3-D trunk geometry only: the logical-axes kernel of 2-D nets / few time
        // steps keeps its layers — its tests pin the selection at such sizes)
        const bool small_trunk = training && o.mfma && !conv_mfma_is_gen(g, precision) && !s3_opt_has(S3O_NO_FEWPOS) &&
                                 !s3_opt_has(S3O_NO_FEWPOS_TRUNK) && precision != S3_PREC_BF16X3 &&
                                 P_out <= 1024 && conv_fewpos_supported(g) && conv_fewpos_mfma_ok(g);
        if (small_trunk) o.mfma = false;
        o.fewpos = !o.mfma && !s3_opt_has(S3O_NO_FEWPOS) && conv_fewpos_supported(g);
        // bf16 plans: the weight-streaming fp32 path only for really few
        // positions; mid-size layers go to the gather-MFMA kernels
        // (... unless the one-launch fp32-MFMA kernels take the layer while the
        // chip is mostly idle: no per-step filter pack, no split-K epilogue)
        // (BF16X3 plans keep their split-bf16 gather-MFMA kernels)
        // (training plans only: an inference plan's kernels must not change
        // with the batch size — chunk-by-chunk and batched runs agree bit for bit)
        const bool fp_small = training && plan_tiny && !o.mfma && !s3_opt_has(S3O_NO_FEWPOS) && precision != S3_PREC_BF16X3 &&
                              conv_fewpos_mfma_small_ok(ctx, g);
        if (o.fewpos && (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] >= 256 &&
            conv_gconv_supported(g, precision) && conv_gconv_dgrad_supported(g, precision) &&
            conv_wgrad_gen_supported(g) && !((fp_small || small_trunk) && conv_fewpos_mfma_ok(g)))
          o.fewpos = false;
        // the few-channel head / tail convs and small filters on those kernels too
        bool fp_small_taken = false;
        if (!o.fewpos && fp_small) { o.fewpos = true; fp_small_taken = true; }
        o.gconv = !o.mfma && !o.fewpos && conv_gconv_supported(g, precision);
        o.halo32 = o.gconv && d.res < 0 && conv_halo32_supported(ctx, g, precision);
        o.halo_s2 = o.gconv && d.res < 0 && conv_halo_s2_supported(ctx, g, precision);
        o.tail_x3 = !o.mfma && !o.fewpos && d.res < 0 && conv_tail_x3_supported(g, precision);
        if (o.fewpos) {
          o.fp_mfma = fp_small_taken || conv_fewpos_mfma_ok(g);
          max_fp = std::max(max_fp, conv_fewpos_partial_bytes(g));
          if (training) {
            max_partial = std::max(max_partial, conv_fewpos_wgrad_partial_bytes(g));
            if (g.pad_mode == S3_PAD_REFLECT) {   // frame of the reflect dgrad
              size_t fb = (size_t)g.N * g.Cin * sizeof(float);
              for (int q = 0; q < 3; ++q) fb *= (size_t)(g.D[q] + 2 * g.lo[q]);
              max_dxp = std::max(max_dxp, fb);
            }
          }
        }
        size_t ysz = (size_t)g.N * g.O[0] * g.O[1] * g.O[2] * g.Cout * sizeof(float);
        max_dpre = std::max(max_dpre, ysz);
        max_partial = std::max(max_partial, conv_generic_wgrad_partial_bytes(g));
        if (training && !s3_opt_has(S3O_NO_MFMA_BWD)) {
          o.wgrad_mfma = !small_trunk && conv_wgrad_mfma_supported(g);
          o.wgrad_bf16 = o.wgrad_mfma && conv_wgrad_bf16_supported(g, precision);
          if (o.wgrad_bf16)
            max_partial = std::max(max_partial, conv_wgrad_bf16_partial_bytes(ctx, g));
          else if (o.wgrad_mfma)
            max_partial = std::max(max_partial, conv_wgrad_mfma_partial_bytes(ctx, g));
          o.wgrad_tail = !o.fewpos && conv_wgrad_tail_supported(g, precision);
          if (o.wgrad_tail)
            max_partial = std::max(max_partial, conv_wgrad_tail_partial_bytes(ctx, g));
          o.wgrad_c2 = !o.fewpos && !o.wgrad_tail && conv_wgrad_c2_supported(g, precision);
          if (o.wgrad_c2)
            max_partial = std::max(max_partial, conv_wgrad_c2_partial_bytes(ctx, g));
          o.wgrad_bf16_gen = !o.wgrad_mfma && !o.fewpos && !o.wgrad_c2 && !o.wgrad_tail &&
                             conv_wgrad_bf16_gen_supported(g, precision);
          if (o.wgrad_bf16_gen)
            max_partial = std::max(max_partial, conv_wgrad_bf16_gen_partial_bytes(ctx, g));
          o.wgrad_bf16_2d = !o.wgrad_mfma && !o.fewpos && !o.wgrad_c2 && !o.wgrad_tail && !o.wgrad_bf16_gen &&
                            conv_wgrad_bf16_2d_supported(g, precision);
          if (o.wgrad_bf16_2d)
            max_partial = std::max(max_partial, conv_wgrad_bf16_2d_partial_bytes(ctx, g));
          if (!o.wgrad_mfma && !o.fewpos && !o.wgrad_c2 && !o.wgrad_tail && !o.wgrad_bf16_gen && !o.wgrad_bf16_2d &&
              conv_wgrad_gen_supported(g)) {
            o.wgrad_gen = true;
            max_partial = std::max(max_partial, conv_wgrad_gen_partial_bytes(ctx, g));
          }
          // what is left would take the generic kernel (one thread per filter
          // element walking every position: 204 us for the 1 500 positions of
          // the C1 discriminator's first layers); with few positions the slab
          // kernel of the fewpos family does any C_in / C_out
          if (!o.wgrad_mfma && !o.fewpos && !o.wgrad_c2 && !o.wgrad_tail && !o.wgrad_bf16_gen && !o.wgrad_bf16_2d &&
              !o.wgrad_gen && conv_fewpos_wgrad_ok(g) && !s3_opt_has(S3O_NO_FEWPOS)) {
            o.fewpos_wgrad = true;
            o.fp_wg_mfma = conv_fewpos_wgrad_mfma_ok(g);
            max_partial = std::max(max_partial, conv_fewpos_wgrad_partial_bytes(g));
          }
          o.dgrad_mfma = !small_trunk && conv_dgrad_mfma_supported(g, precision);
          // (BF16X3: the split-bf16 tile kernel takes the same geometry — the
          // discriminator's valid 32 -> 64 conv left the gather-MFMA adjoint,
          // 2.25 -> 0.5 ms at C2 batch 8)
          if (!o.dgrad_mfma && !o.fewpos &&
              (precision == S3_PREC_BF16 || (precision == S3_PREC_BF16X3 && !s3_opt_has(S3O_NO_DGRAD_X3))) &&
              conv_dgrad_mfma_valid_supported(g, precision)) {
            o.dgrad_mfma = o.dgrad_valid = true;
          }
          if (!o.dgrad_mfma && !o.fewpos && (precision == S3_PREC_BF16 || precision == S3_PREC_BF16X3) &&
              !s3_opt_has(S3O_NO_DGRAD_FEWCH) &&
              (g.Cout == 2 || g.Cout == 4) && g.Cin % 4 == 0 && g.d2s == 1 &&
              (int64_t)g.N * g.D[0] * g.D[1] * g.D[2] >= 4096) {
            // hi-res tail conv (8 -> 2): its data gradient is a conv with 2 input
            // channels — the taps-in-K few-channel kernel over the padded frame
            bool same = true;
            for (int q = 0; q < 3; ++q)
              same = same && g.k[q] == 3 && g.s[q] == 1 && g.lo[q] == 1 && g.O[q] == g.D[q];
            if (same && conv_gconv_supported(conv_dgrad_geom(g), precision))
              o.dgrad_mfma = o.dgrad_fewch = true;
          }
          if (!o.dgrad_mfma && !o.fewpos && conv_dgrad_chunked_supported(g, precision)) {
            o.dgrad_mfma = o.dgrad_chunked = true;
            o.dgrad_valid = g.lo[0] == 0;
          }
          // (BF16X3 plans: the split-bf16 forms of the same two kernels, round 4)
          o.dgrad_c2 = !o.dgrad_mfma && !o.fewpos &&
                       (conv_dgrad_c2_supported(g, precision) || conv_dgrad_c2_x3_supported(g, precision));
          o.dgrad_s2 = !o.dgrad_mfma && !o.dgrad_c2 && !o.fewpos &&
                       (conv_dgrad_s2_supported(ctx, g, precision) || conv_dgrad_s2_x3_supported(ctx, g, precision));
          // what is left behind a REFLECT pad (2-D nets, few time steps, odd channel
          // counts): the padded-frame correlation on the logical-axes tile kernel
          if (!o.dgrad_mfma && !o.dgrad_c2 && !o.dgrad_s2 && !o.fewpos && conv_dgrad_gen_supported(g, precision))
            o.dgrad_mfma = o.dgrad_gen = true;
          o.gconv_dgrad = !o.dgrad_mfma && !o.dgrad_c2 && !o.dgrad_s2 && !o.fewpos && conv_gconv_dgrad_supported(g, precision);
          if (o.gconv_dgrad && g.pad_mode == S3_PAD_REFLECT)
            max_dxp = std::max(max_dxp, (size_t)g.N * (g.D[0] + 2 * g.lo[0]) * (g.D[1] + 2 * g.lo[1]) *
                                            (g.D[2] + 2 * g.lo[2]) * g.Cin * sizeof(float));
          if (o.dgrad_mfma) {
            o.dg = o.dgrad_gen ? conv_dgrad_gen_geom(g)
                   : o.dgrad_chunked ? conv_dgrad_chunk_geom(g, 0)
                                     : (o.dgrad_valid ? conv_dgrad_valid_geom(g) : conv_dgrad_geom(g));
            max_dxp = std::max(max_dxp, (size_t)o.dg.N * o.dg.O[0] * o.dg.O[1] * o.dg.O[2] * o.dg.Cout * sizeof(float));
          }
        }
      } break;
      case S3_OP_DENSE: {
        if (d.w < 0) return bad("plan: dense without weights");
        const TensorRec& it = pl->t[d.in0];
        const TensorRec& ot = pl->t[d.out];
        if (params->p[d.w].size != it.dims[4] * ot.dims[4]) return bad("plan: dense weight size mismatch");
        max_dpre = std::max(max_dpre, (size_t)ot.numel * sizeof(float));
      } break;
      case S3_OP_REPEAT_T: case S3_OP_D2S: case S3_OP_PAD: case S3_OP_CROP:
      case S3_OP_ROLL_T: case S3_OP_DILATE:
        fill_gather_geom(pl, d, o.gg);
        if (d.kind == S3_OP_DILATE)
          for (int q = 0; q < 3; ++q) o.gg.lo[q] = d.stride[q] < 1 ? 1 : d.stride[q];
        break;
      case S3_OP_CONCAT:
        if (d.in1 < 0) return bad("plan: concat without second input");
        break;
      case S3_OP_ADD:
        if (d.in1 < 0) return bad("plan: add without second input");
        break;
      case S3_OP_ACT: break;
      case S3_OP_VIEW:
        if (pl->t[d.in0].numel != pl->t[d.out].numel) return bad("plan: view changes element count");
        pl->t[d.out].alias_root = d.in0;
        break;
      default:
        return bad("plan: unknown op kind");
    }
  }
  // ---- inference plans: Sup3rConcat of a 64-channel tensor and ONE exogenous
  // channel in front of a 3 x 3 Conv2D (sup3rcc/gen_wind_5x_1x_6f at hi-res: a
  // 65-channel fp32 tensor written, read back by a two-pass fp32 conv: 28 of
  // 126 ms at 96 x 750 x 750).  The conv is linear in its input channels: it
  // runs as the 64 -> C_out conv over the bf16 tensor on the weights-stationary
  // kernel, which adds the exogenous channel's nine taps per output from the
  // fp32 field itself (ConvGeom::w_cin / exo); the concat never runs.  Checked
  // again once the dtypes are known; if the kernel cannot take the conv after
  // all the plan is built again without the split.
  static thread_local int tl_no_ws_exo = 0;
  if (!training && precision == S3_PREC_BF16 && !s3_opt_has(S3O_NO_WS_EXO) && !s3_opt_has(S3O_FP32_ACT) &&
      !tl_no_ws_exo) {
    for (int i = 0; i < n_ops; ++i) {
      OpRec& c = pl->ops[i];
      if (c.d.kind != S3_OP_CONCAT) continue;
      const TensorRec& xt = pl->t[c.d.in0];
      const TensorRec& et = pl->t[c.d.in1];
      if (xt.dims[4] != 64 || et.dims[4] != 1 || !pl->t[root_of(pl, c.d.in1)].is_input) continue;
      const int ct_ = root_of(pl, c.d.out);
      if (ct_ == root_of(pl, output)) continue;
      int user = -1, n_use = 0;
      for (int k = 0; k < n_ops; ++k) {
        const s3_op_desc& u = pl->ops[k].d;
        for (int id : {u.in0, u.in1, u.res})
          if (id >= 0 && root_of(pl, id) == ct_) { ++n_use; user = k; }
      }
      if (n_use != 1 || user <= i) continue;
      OpRec& v = pl->ops[user];
      if (v.d.kind != S3_OP_CONV || root_of(pl, v.d.in0) != ct_ || v.cg.Cin != 65) continue;
      ConvGeom gs = v.cg;
      gs.Cin = 64; gs.w_cin = 65;
      if (!conv2d_ws_geom_ok(gs) || conv2d_ws_tail_geom_ok(gs) || !conv_mfma_supported(gs, precision)) continue;
      v.cg = gs;
      v.d.in0 = c.d.in0;
      v.exo_src = c.d.in1;
      v.mfma = true;
      v.fewpos = v.gconv = v.halo32 = v.halo_s2 = v.tail_x3 = false;
      c.fused_away = true;
    }
  }

  // ---- inference plans: a skip add right behind a 2-D 64 -> 64 k conv that
  // already carries a residual (the last block's sum + the big skip of
  // sup3rcc/gen_*_5x_1x_* at hi-res) is absorbed into that conv's store on the
  // weights-stationary kernel (ConvGeom::res2): conv.out := add.out, the add
  // never runs.  Verified with the dtypes below, like the concat split.
  if (!training && precision == S3_PREC_BF16 && !s3_opt_has(S3O_NO_WS_RES2) && !s3_opt_has(S3O_FP32_ACT) &&
      !s3_opt_has(S3O_NO_ADD16) && !tl_no_ws_exo) {
    for (int i = 0; i < n_ops; ++i) {
      OpRec& a = pl->ops[i];
      if (a.d.kind != S3_OP_ADD || a.d.bcast_c || a.fused_away) continue;
      for (int side = 0; side < 2; ++side) {
        const int t_conv = side ? a.d.in1 : a.d.in0, t_other = side ? a.d.in0 : a.d.in1;
        const int rt = root_of(pl, t_conv);
        if (rt == root_of(pl, output) || rt == root_of(pl, t_other)) continue;
        int prod = -1, n_use = 0;
        for (int k = 0; k < n_ops; ++k) {
          const s3_op_desc& u = pl->ops[k].d;
          if (u.kind != S3_OP_VIEW && root_of(pl, u.out) == rt) prod = k;
          for (int id : {u.in0, u.in1, u.res})
            if (id >= 0 && root_of(pl, id) == rt) ++n_use;
        }
        if (prod < 0 || prod >= i || n_use != 1) continue;
        OpRec& c = pl->ops[prod];
        if (c.d.kind != S3_OP_CONV || c.res2_src >= 0 || c.d.res < 0 || c.cg.act != S3_ACT_NONE || c.cg.d2s != 1 ||
            c.cg.w_cin || !c.mfma || !conv2d_ws_geom_ok(c.cg) || conv2d_ws_tail_geom_ok(c.cg))
          continue;
        // (the other operand must exist before the conv runs)
        int prod_other = -1;
        for (int k = 0; k < n_ops; ++k)
          if (pl->ops[k].d.kind != S3_OP_VIEW && root_of(pl, pl->ops[k].d.out) == root_of(pl, t_other)) prod_other = k;
        if (prod_other >= prod) continue;
        c.res2_src = t_other;
        c.d.out = a.d.out;
        a.fused_away = true;
        break;
      }
    }
  }

  // ---- activation dtypes.  Inference plans in bf16 mode keep a tensor in
  // bf16 when its producer can write it (MFMA conv, direct conv, index op) and
  // EVERY consumer can read it (MFMA conv input / residual, index op);
  // everything else — plan inputs/outputs, training plans, f32 mode — is fp32.
  // Training plans (SUP3R_AMD_BF16_TRAIN_ACT=0 opts out): the same, but a saved
  // activation is also read by the backward pass — a tensor stays bf16 only if
  // every conv that consumes it takes its weight gradient through the
  // transpose-read bf16 kernel (which stages bf16 directly); the LeakyReLU
  // mask pass reads the sign of a bf16 output; gradients stay fp32.  The
  // forward is then bit-identical to the bf16 inference plan of the trunk.
  const bool train16 = training && precision == S3_PREC_BF16 && !s3_opt_has(S3O_FP32_ACT) &&
                       !(s3_opt_has(S3O_BF16_TRAIN_ACT) && s3_opt_int(S3O_BF16_TRAIN_ACT, 0) == 0);
  if ((!training || train16) && precision == S3_PREC_BF16 && !s3_opt_has(S3O_FP32_ACT)) {
    std::vector<int> dt(n_tensors, 1);
    auto demote = [&](int id, bool& changed) {
      if (id < 0) return;
      int r = root_of(pl, id);
      if (dt[r]) { dt[r] = 0; changed = true; }
    };
    bool changed = true;
    {
      bool c0 = false;
      for (int id : pl->inputs) demote(id, c0);
      demote(output, c0);
      // tensors nobody produces (defensive) stay fp32
      std::vector<char> produced(n_tensors, 0);
      for (auto& o : pl->ops) produced[root_of(pl, o.d.out)] = 1;
      for (int i = 0; i < n_tensors; ++i)
        if (!produced[root_of(pl, i)]) demote(i, c0);
    }
    // outputs of the few-channel gather conv: bf16 only towards a conv INPUT
    // (the consumer rounds to bf16 when it stages anyway, so nothing changes
    // numerically); as a residual the fp32 value is kept
    std::vector<char> fewch_out(n_tensors, 0);
    for (auto& o : pl->ops)
      if (o.d.kind == S3_OP_CONV && !o.mfma && o.gconv && o.cg.Cin <= 4) fewch_out[root_of(pl, o.d.out)] = 1;
    while (changed) {
      changed = false;
      for (auto& o : pl->ops) {
        const s3_op_desc& d = o.d;
        if ((d.kind == S3_OP_CONCAT || d.kind == S3_OP_ADD) && o.fused_away) continue;   // (never runs: its operands go to the conv)
        switch (d.kind) {
          case S3_OP_CONV:
            if (training && d.res >= 0 && fewch_out[root_of(pl, d.res)]) demote(d.res, changed);
            if (o.mfma) {
              if (!conv_mfma_bf16_out_ok(o.cg)) demote(d.out, changed);
              if (o.cg.Cin % 8 != 0) demote(d.in0, changed);   // (logical-axes kernel: 16-B bf16 chunks)
              // (a saved bf16 input is re-read by the weight gradient: the
              // transpose-read kernels stage bf16 directly)
              // (round 5: ... and so does the 2-D weight gradient, so that the
              // 64 -> 64 layers of a 2-D training plan keep bf16 cells and run
              // forward on the weights-stationary kernel)
              if (training && !o.wgrad_bf16 && !(o.wgrad_bf16_gen && !s3_opt_has(S3O_NO_DISC_BF16)) &&
                  !(o.wgrad_bf16_2d && o.cg.Cin % 8 == 0 && !s3_opt_has(S3O_NO_TRAIN2D_BF16)))
                demote(d.in0, changed);
            } else if (training) {
              // every other conv reads / writes fp32 in training plans — except
              // the hi-res tail conv, whose MFMA forward takes bf16 cells and
              // whose weight gradient (conv_wgrad_tail_kernel) stages them as is
              const bool tail16 = d.res < 0 && !o.fewpos && o.wgrad_tail && conv_tail_mfma_supported(o.cg);
              // ... and the hi-res discriminator pair: the few-channel conv
              // (C_in <= 4) may write bf16 when its consumer is a gather-MFMA
              // conv (bf16 cells in, C_in % 8 == 0) whose weight gradient is
              // the general transpose-read kernel (bf16 staging)
              // ... and so on down the stack: every gather-MFMA / LDS-halo conv with
              // C_in % 8 == 0 takes and writes bf16 cells (SUP3R_AMD_DISC_BF16=1
              // keeps it to the first pair, SUP3R_AMD_NO_DISC_BF16 turns it off)
              const bool disc16 = !s3_opt_has(S3O_NO_DISC_BF16);
              const bool deep16 = disc16 && !(s3_opt_has(S3O_DISC_BF16) && s3_opt_int(S3O_DISC_BF16, 0) == 1);
              const bool gc_in16 = disc16 && o.gconv && (!o.halo32 || deep16) && d.res < 0 && o.wgrad_bf16_gen &&
                                   o.cg.Cin % 8 == 0;
              const bool gc_out16 = disc16 && o.gconv && d.res < 0 && o.cg.Cout % 8 == 0 &&
                                    (o.cg.Cin == 2 || o.cg.Cin == 4 || (deep16 && o.cg.Cin % 8 == 0));
              if (!tail16 && !gc_in16) demote(d.in0, changed);
              demote(d.res, changed);
              if (!gc_out16) demote(d.out, changed);
            } else {
              // the direct kernels read fp32, except the small-channel tail
              // convs (MFMA C_in = 8 / sliding window) which take bf16 cells
              const bool small = d.res < 0 && !o.fewpos &&
                                 (conv_small_supported(o.cg, 1) || conv_tail_mfma_supported(o.cg));
              if (!small) demote(d.in0, changed);
              demote(d.res, changed);
              if (small || o.fewpos) demote(d.out, changed);
            }
            break;
          case S3_OP_REPEAT_T: case S3_OP_D2S: case S3_OP_PAD: case S3_OP_CROP:
          case S3_OP_ROLL_T: case S3_OP_DILATE:
            if (dt[root_of(pl, d.in0)] != dt[root_of(pl, d.out)]) {
              demote(d.in0, changed);
              demote(d.out, changed);
            }
            break;
          case S3_OP_VIEW:
            break;  // alias: one root, one dtype (element count is preserved)
          case S3_OP_ADD:
            // inference plans: a skip add stays in bf16 when both operands and
            // the sum are bf16 tensors (add16_kernel)
            if (training || d.bcast_c || (pl->t[d.out].numel & 7) || s3_opt_has(S3O_NO_ADD16) ||
                !(dt[root_of(pl, d.in0)] && dt[root_of(pl, d.in1)] && dt[root_of(pl, d.out)])) {
              demote(d.in0, changed); demote(d.in1, changed); demote(d.out, changed);
            }
            break;
          default:
            demote(d.in0, changed); demote(d.in1, changed);
            demote(d.res, changed); demote(d.out, changed);
            break;
        }
      }
    }
    for (int i = 0; i < n_tensors; ++i) pl->t[i].dtype = dt[root_of(pl, i)];
  }
  for (auto& o : pl->ops) {
    if (o.d.kind != S3_OP_CONV) continue;
    o.io.in_bf16 = pl->t[root_of(pl, o.d.in0)].dtype;
    o.io.out_bf16 = pl->t[root_of(pl, o.d.out)].dtype;
    o.io.res_bf16 = o.d.res >= 0 ? pl->t[root_of(pl, o.d.res)].dtype : 0;
    if (s3_opt_has(S3O_TRACE))
      fprintf(stderr, "[plan] conv %d->%d %s: mfma %d fewpos %d gconv %d halo32 %d | in16 %d out16 %d res16 %d | "
              "wgrad bf16 %d gen %d 2d %d c2 %d tail %d mfma %d | dgrad mfma %d c2 %d s2 %d gconv %d\n",
              o.cg.Cin, o.cg.Cout, training ? "train" : "infer", (int)o.mfma, (int)o.fewpos, (int)o.gconv,
              (int)o.halo32, o.io.in_bf16, o.io.out_bf16, o.io.res_bf16, (int)o.wgrad_bf16, (int)o.wgrad_bf16_gen,
              (int)o.wgrad_bf16_2d, (int)o.wgrad_c2, (int)o.wgrad_tail, (int)o.wgrad_mfma, (int)o.dgrad_mfma,
              (int)o.dgrad_c2, (int)o.dgrad_s2, (int)o.gconv_dgrad);
  }

  for (auto& o : pl->ops)
    if (o.d.kind == S3_OP_CONV && o.mfma && conv_mfma_is_gen(o.cg, precision) && o.cg.in_rep <= 1 &&
        conv2d_ws_supported(o.cg, precision, o.io, o.d.res >= 0))
      o.cg.ws_only = 1;
  for (auto& o : pl->ops) {
    if (o.d.kind != S3_OP_CONV || (o.exo_src < 0 && o.res2_src < 0)) continue;
    if (conv2d_ws_supported(o.cg, precision, o.io, o.d.res >= 0) &&
        (o.res2_src < 0 || pl->t[root_of(pl, o.res2_src)].dtype == 1))
      continue;
    // (a consumer of the 64-channel tensor that needs fp32 cells, ...): build
    // the plan again with the concat as it is written
    delete pl;
    ++tl_no_ws_exo;
    const int rc = s3_plan_create_opt(ctx, params, tensors, n_tensors, ops, n_ops, inputs, n_inputs, output,
                                      precision, training, options, out);
    --tl_no_ws_exo;
    return rc;
  }

  // ---- activation-adjoint fusion (training): a conv whose data gradient runs
  // on conv_dgrad_s2_kernel and whose input is the fp32 output of an activated
  // conv with no other consumer applies that conv's mask in its own store
  if (training) {
    std::vector<int> ncons(n_tensors, 0), prod(n_tensors, -1);
    for (int i = 0; i < n_ops; ++i) {
      const s3_op_desc& d = pl->ops[i].d;
      for (int id : {d.in0, d.in1, d.res})
        if (id >= 0) ++ncons[root_of(pl, id)];
      if (d.kind != S3_OP_VIEW) prod[root_of(pl, d.out)] = i;
    }
    for (auto& o : pl->ops) {
      if (o.d.kind == S3_OP_CONV) {
        const int pr = prod[root_of(pl, o.d.in0)];
        if (pr >= 0 && pl->ops[pr].d.kind == S3_OP_CONV) o.in_prod = pr;
      }
      // (the stride-2 dgrad kernel masks from an fp32 y; the frame fold of the
      // halo-tile dgrad from fp32 or bf16)
      const bool fold = (o.dgrad_mfma && !o.dgrad_valid) ||
                        (o.fewpos && o.fp_mfma && o.cg.pad_mode == S3_PAD_REFLECT && !o.dgrad_chunked &&
                         !o.dgrad_s2 && !o.dgrad_c2 && !o.gconv_dgrad);   // (the one-launch fewpos dgrad folds its frame too)
      if (o.d.kind != S3_OP_CONV || !(o.dgrad_s2 || fold)) continue;
      const int r = root_of(pl, o.d.in0);
      const int pi = prod[r];
      if (pi < 0 || ncons[r] != 1 || r == root_of(pl, output) || pl->t[r].is_input) continue;

      if (fold && (o.cg.Cin & 3)) continue;
      const OpRec& po = pl->ops[pi];
      if (po.d.kind == S3_OP_CONV && po.cg.act != S3_ACT_NONE && po.cg.d2s == 1 && po.d.res < 0)
        o.mask_prod = pi;
    }
  }

  // ---- inference plans: a temporal repeat whose only consumer is a conv on
  // the persistent trunk kernel is read through that kernel's halo index
  // (cell t of the repeated tensor = cell t / rep of the source) instead of
  // being written out and read back (at C2 batch 32 the 96 -> 288
  // repeat alone is a 400 MB store + load per forward)
  if (!training && precision == S3_PREC_BF16 && !s3_opt_has(S3O_NO_REPEAT_FUSE)) {
    for (int i = 0; i < n_ops; ++i) {
      OpRec& r = pl->ops[i];
      if (r.d.kind != S3_OP_REPEAT_T || !conv_mfma_persist_rep_ok(r.d.rep)) continue;
      const int rt = root_of(pl, r.d.out);
      if (rt == root_of(pl, output) || pl->t[root_of(pl, r.d.in0)].dtype != pl->t[rt].dtype) continue;
      // every consumer is a plain 64 -> 64 trunk conv on the persistent kernel
      // ('same' extents, one padding for all axes, no depth-to-space store)
      // taking it as the input or as the residual — SkipConnection sources sit
      // right behind the last temporal expansion in the reference's generators
      // — and carries no other repeat factor yet
      bool ok = true;
      int n_use = 0;
      for (int k = 0; k < n_ops && ok; ++k) {
        const OpRec& c = pl->ops[k];
        const bool as_in = c.d.in0 >= 0 && root_of(pl, c.d.in0) == rt;
        const bool as_in1 = c.d.in1 >= 0 && root_of(pl, c.d.in1) == rt;
        const bool as_res = c.d.res >= 0 && root_of(pl, c.d.res) == rt;
        if (!as_in && !as_in1 && !as_res) continue;
        ++n_use;
        ok = k > i && !as_in1 && c.d.kind == S3_OP_CONV && c.mfma &&
             conv_mfma_persist_supported(ctx, c.cg, c.io, c.d.res >= 0) && c.cg.d2s == 1 && c.cg.Cout == 64 &&
             c.cg.O[2] < 32768 && c.cg.O[2] % r.d.rep == 0;
        for (int q = 0; q < 3; ++q) ok = ok && c.cg.lo[q] == c.cg.lo[0] && c.cg.O[q] == c.cg.D[q];
        const int have = c.cg.in_rep > 1 ? c.cg.in_rep : c.cg.res_rep;
        if (have > 1 && have != r.d.rep) ok = false;
      }
      if (!ok || !n_use) continue;
      for (int k = i + 1; k < n_ops; ++k) {
        OpRec& c = pl->ops[k];
        if (c.d.kind != S3_OP_CONV) continue;
        if (root_of(pl, c.d.in0) == rt) { c.cg.in_rep = r.d.rep; c.rep_src = r.d.in0; }
        if (c.d.res >= 0 && root_of(pl, c.d.res) == rt) { c.cg.res_rep = r.d.rep; c.res_src = r.d.in0; }
      }
      r.fused_away = true;
    }
  }

  // ---- static arena planning.  Training keeps every tensor; inference
  // reuses buffers by liveness (greedy best-fit).
  std::vector<int> last_use(n_tensors, -1);
  for (int i = 0; i < n_ops; ++i) {
    const s3_op_desc& d = pl->ops[i].d;
    int ids[8] = {d.in0, d.in1, d.res, d.out, pl->ops[i].rep_src, pl->ops[i].res_src, pl->ops[i].exo_src,
                  pl->ops[i].res2_src};
    for (int q = 0; q < 8; ++q)
      if (ids[q] >= 0) last_use[root_of(pl, ids[q])] = i;
  }
  last_use[root_of(pl, output)] = n_ops + 1;
  std::vector<size_t> bsize;
  std::vector<int> bfree_at;  // op index after which the buffer is free
  for (int i = 0; i < n_ops; ++i) {
    const s3_op_desc& d = pl->ops[i].d;
    if (d.kind == S3_OP_VIEW) continue;
    if ((d.kind == S3_OP_CONCAT || d.kind == S3_OP_ADD) && pl->ops[i].fused_away) continue;   // (never materialised / written by the conv)
    TensorRec& ot = pl->t[d.out];
    size_t need = ot.bytes();
    int pick = -1;
    // (option KEEP_ACTIVATIONS: an inference plan with a buffer per tensor, so that a
    // test can read every op's output — s3_plan_tensor_read — of the kernels and
    // fusions only inference plans select)
    if (!training && !s3_opt_has(S3O_KEEP_ACTIVATIONS)) {
      for (int b = 0; b < (int)bsize.size(); ++b)
        if (bfree_at[b] < i && bsize[b] >= need &&
            (pick < 0 || bsize[b] < bsize[pick]))
          pick = b;
    }
    if (pick < 0) { bsize.push_back(need); bfree_at.push_back(0); pick = (int)bsize.size() - 1; }
    ot.buffer = pick;
    bfree_at[pick] = last_use[d.out];
    max_t = std::max(max_t, need);
  }
  for (int i = 0; i < n_tensors; ++i) max_t = std::max(max_t, (size_t)pl->t[i].numel * sizeof(float));
  pl->buffers.resize(bsize.size());
  pl->buffer_bytes = bsize;
  for (size_t b = 0; b < bsize.size(); ++b) {
    int rc = plan_alloc(pl, (void**)&pl->buffers[b], bsize[b]);
    if (rc) { s3_plan_destroy(pl); return rc; }
  }
  for (int i = 0; i < n_tensors; ++i)
    if (pl->t[i].buffer >= 0) pl->t[i].ptr = pl->buffers[pl->t[i].buffer];
  if (training) {
    for (int i = 0; i < n_tensors; ++i) {
      if (pl->t[i].alias_root >= 0) continue;
      int rc = plan_alloc(pl, (void**)&pl->t[i].gptr, (size_t)pl->t[i].numel * sizeof(float));
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
    int rc = plan_alloc(pl, (void**)&pl->dpre, max_dpre);
    // bf16 copy of dPre for the MFMA gradient kernels (they round their
    // operand to bf16 anyway; a bf16 source halves the bytes they stage, and
    // the persistent data gradient / the wave-specialised weight gradient take
    // nothing else).  Whichever pass finishes a conv's dPre leaves it: the mask
    // pass (d2s walk included), the frame folds (compile-time variants: a
    // run-time side store cost them 42 us per 75 MB), the stride-2 data
    // gradient.  One buffer, handed from producer to consumer (dpre16_for).
    if (precision == S3_PREC_BF16 && !s3_opt_has(S3O_NO_DPRE16)) {
      size_t max16 = 0;
      for (auto& o : pl->ops) {
        if (rc || o.d.kind != S3_OP_CONV) continue;
        // (... and the gather-MFMA adjoint of the strided / valid discriminator
        // convs: a lane's 8 channels of a dPre cell are one 16-B load)
        const bool gadj = o.gconv_dgrad && (o.cg.Cout & 7) == 0 && o.cg.pad_mode != S3_PAD_REFLECT &&
                          !s3_opt_has(S3O_NO_GCONV_DY16);
        if (!gadj && (!o.dgrad_mfma || o.dgrad_fewch || (o.cg.Cout & 3))) continue;
        if (o.dgrad_gen && (o.cg.Cout & 7)) continue;   // (16-B bf16 chunks of a dPre cell)
        if (o.dgrad_chunked && ((o.cg.Cout & 7) || s3_opt_has(S3O_NO_CHUNKED_DY16))) continue;
        o.use16 = true;
        max16 = std::max(max16, (size_t)pl->t[root_of(pl, o.d.out)].numel * 2);
        // the reflect-padded 64 -> 64 trunk conv on the persistent kernel:
        // its padded frame is written — and folded from — as bf16 (round 4)
        o.dgrad_frame16 = training && o.dgrad_mfma && !o.dgrad_valid && !o.dgrad_chunked && !o.dgrad_fewch &&
                          o.dg.Cout == 64 && (o.cg.Cin & 3) == 0 && o.cg.pad_mode == S3_PAD_REFLECT &&
                          conv_mfma_persist_dgrad_supported(ctx, o.dg) && !s3_opt_has(S3O_NO_FRAME16);
        // ... and so is the frame of a 2-D 64 -> 64 k conv's data gradient on the
        // weights-stationary kernel (round 5)
        if (training && o.dgrad_gen && !o.dgrad_valid && conv2d_ws_frame_geom_ok(o.dg) &&
            !s3_opt_has(S3O_NO_FRAME16) && !s3_opt_on(S3O_NO_CONV2D_WS))
          o.dgrad_frame16 = true;
      }
      // ... and for the stride-2 data gradient that stores dPre of the
      // few-channel conv below it as bf16 only (see the dgrad_s2 branch)
      for (auto& o : pl->ops)
        if (o.d.kind == S3_OP_CONV && o.dgrad_s2 && o.mask_prod >= 0 && pl->ops[o.mask_prod].wgrad_c2 &&
            conv_dgrad_s2_out16_ok(o.cg))
          max16 = std::max(max16, (size_t)pl->t[root_of(pl, o.d.in0)].numel * 2);
      if (!rc && max16) rc = plan_alloc(pl, &pl->dpre16, max16);
      pl->dpre16_bytes = max16;
    }
    if (!rc) rc = plan_alloc(pl, (void**)&pl->gtmp, max_t);
    if (!rc) rc = plan_alloc(pl, (void**)&pl->bsum, (size_t)4096 * 256 * sizeof(float));
    if (!rc) rc = plan_alloc(pl, (void**)&pl->bsum2, (size_t)4096 * 256 * sizeof(float));
    if (!rc && max_partial) {
      rc = plan_alloc(pl, (void**)&pl->wg_partial, max_partial);
      pl->wg_partial_bytes = max_partial;
    }
    if (!rc && max_dxp) rc = plan_alloc(pl, (void**)&pl->dxp, max_dxp);
    for (auto& o : pl->ops) {
      if (rc || o.d.kind != S3_OP_CONV || !o.dgrad_mfma) continue;
      rc = plan_alloc(pl, (void**)&o.dg_w32, (size_t)27 * o.cg.Cin * o.cg.Cout * sizeof(float));
      if (!rc && o.dgrad_chunked) {
        for (int k = 0; !rc && k < (o.cg.Cout + 63) / 64; ++k)
          rc = plan_alloc(pl, &o.dgc_wbf[k], conv_mfma_packed_bytes(o.dg, precision));
        continue;
      }
      if (!rc && precision != S3_PREC_F32)
        rc = plan_alloc(pl, &o.dg_wbf, o.dgrad_fewch ? conv_gconv_packed_bytes(o.dg, 0, precision == S3_PREC_BF16X3)
                                                     : conv_mfma_packed_bytes(o.dg, precision));
    }
    if (rc) { s3_plan_destroy(pl); return rc; }
  }
  if (max_fp) {
    int rc = plan_alloc(pl, (void**)&pl->fp_partial, max_fp);
    if (rc) { s3_plan_destroy(pl); return rc; }
    pl->fp_partial_bytes = max_fp;
  }
  if (training) {
    for (auto& o : pl->ops) {
      if (o.d.kind != S3_OP_CONV || !o.fewpos) continue;
      int rc = plan_alloc(pl, (void**)&o.fp_wt, (size_t)o.cg.k[0] * o.cg.k[1] * o.cg.k[2] * o.cg.Cin * o.cg.Cout * sizeof(float));
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
  }
  // packed weights of the MFMA convs
  for (auto& o : pl->ops) {
    if (o.d.kind == S3_OP_CONV && o.mfma) {
      int rc = plan_alloc(pl, &o.packed, conv_mfma_packed_bytes(o.cg, precision));
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
  }
  for (auto& o : pl->ops) {
    if (o.d.kind != S3_OP_CONV) continue;
    if (o.gconv) {
      int rc = plan_alloc(pl, &o.gc_w, conv_gconv_packed_bytes(o.cg, 0, precision == S3_PREC_BF16X3));
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
    if (o.halo32 || o.halo_s2) {
      int rc = plan_alloc(pl, &o.h32_w, o.halo32 ? conv_halo32_packed_bytes(o.cg) : conv_halo_s2_packed_bytes(o.cg));
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
    if (o.dgrad_s2) {
      int rc = plan_alloc(pl, &o.dc2_w, precision == S3_PREC_BF16X3 ? conv_dgrad_s2_x3_packed_bytes(o.cg)
                                                                    : conv_dgrad_s2_packed_bytes(o.cg));
      if (rc) { s3_plan_destroy(pl); return rc; }
      // its fused activation mask as sign bytes written by the producer's
      // forward kernel (4 B instead of 64 B per position read back)
      if (o.mask_prod >= 0 && precision == S3_PREC_BF16 && !s3_opt_has(S3O_NO_SIGN_BYTES)) {
        OpRec& po = pl->ops[o.mask_prod];
        if (po.gconv && !po.sign_bytes && conv_gconv_writes_sign_bytes(ctx, po.cg, po.io.out_bf16)) {
          const size_t npos = (size_t)po.cg.N * po.cg.O[0] * po.cg.O[1] * po.cg.O[2];
          rc = plan_alloc(pl, &po.sign_bytes, npos * 4);
          if (rc) { s3_plan_destroy(pl); return rc; }
        }
      }
    }
    if (o.dgrad_c2) {
      int rc = plan_alloc(pl, &o.dc2_w, precision == S3_PREC_BF16X3 ? conv_dgrad_c2_x3_packed_bytes()
                                                                    : conv_dgrad_c2_packed_bytes());
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
    if (o.gconv_dgrad) {
      int rc = plan_alloc(pl, &o.gc_wt, conv_gconv_packed_bytes(o.cg, 1, precision == S3_PREC_BF16X3));
      if (rc) { s3_plan_destroy(pl); return rc; }
    }
  }
  // staging copies of the graph inputs: fixed pointers for the hipGraph replay
  if (!training) {
    for (size_t i = 0; i < pl->inputs.size(); ++i) {
      void* st = nullptr;
      int rc = plan_alloc(pl, &st, (size_t)pl->t[pl->inputs[i]].numel * sizeof(float));
      if (rc) { s3_plan_destroy(pl); return rc; }
      pl->in_stage.push_back((float*)st);
    }
  }
  pl->gwritten.assign(n_tensors, 0);
  // small 2-D conv stacks (the spatial generators at test / C1 sizes): one launch
  // for the whole op list, activations in LDS
  if (!training && precision == S3_PREC_BF16 && n_inputs == 1) {
    std::vector<Fused2dLayer> fl;
    bool ok = true;
    for (auto& o : pl->ops) {
      if (o.d.kind != S3_OP_CONV) { ok = false; break; }
      Fused2dLayer f;
      f.g = o.cg;
      f.in_t = root_of(pl, o.d.in0);
      f.out_t = root_of(pl, o.d.out);
      f.res_t = o.d.res >= 0 ? root_of(pl, o.d.res) : -1;
      f.w_off = params->p[o.d.w].offset;
      f.b_off = o.d.b >= 0 ? params->p[o.d.b].offset : -1;
      fl.push_back(f);
    }
    if (ok) pl->fused2d = fused2d_build(ctx, fl, n_tensors, root_of(pl, pl->inputs[0]), root_of(pl, output));
    if (s3_opt_has(S3O_TRACE))
      fprintf(stderr, "[plan] fused 2-D whole-network kernel: %s\n", pl->fused2d ? "yes" : "no");
  }
  *out = pl;
  return S3_OK;
}

static void graph_drop(s3_plan* pl) {
  if (pl->graph_exec) (void)hipGraphExecDestroy(pl->graph_exec);
  if (pl->graph) (void)hipGraphDestroy(pl->graph);
  pl->graph_exec = nullptr;
  pl->graph = nullptr;
}

extern "C" void s3_plan_destroy(s3_plan* pl) {
  if (!pl) return;
  (void)hipStreamSynchronize(pl->ctx->stream);
  graph_drop(pl);
  if (pl->cap_stream) (void)hipStreamDestroy(pl->cap_stream);
  for (auto& e : pl->prof_ev) (void)hipEventDestroy(e);
  for (void* p : pl->owned) (void)hipFree(p);
  fused2d_free(pl->fused2d);
  delete pl;
}

extern "C" void* s3_plan_tensor(s3_plan* pl, int32_t id) {
  if (!pl || id < 0 || id >= (int)pl->t.size()) return nullptr;
  return pl->t[root_of(pl, id)].ptr;
}

extern "C" int64_t s3_plan_workspace_bytes(const s3_plan* pl) {
  return pl ? (int64_t)pl->total_bytes : 0;
}

static float* tptr(s3_plan* pl, int id) { return pl->t[root_of(pl, id)].ptr; }
static int tdtype(s3_plan* pl, int id) { return pl->t[root_of(pl, id)].dtype; }
static float* gptr(s3_plan* pl, int id) { return pl->t[root_of(pl, id)].gptr; }

// ---- batched filter re-pack.  After an optimizer step every bf16 conv of the
// plan needs its images again; instead of 1 - 3 launches of ~5 us per conv and
// direction (lazily, in front of each conv) one launch per direction walks a
// device table of jobs.  Convs outside the table (other precisions, chunked /
// few-channel data gradients, gather-MFMA convs) keep their lazy packs.
static int pack_tables_build(s3_plan* pl) {
  s3_ctx* ctx = pl->ctx;
  pl->pack_built = true;
  if (pl->precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_BATCHED_PACK)) return S3_OK;
  s3_params* P = pl->params;
  float* W = P->buf[S3_BUF_W];
  std::vector<S3PackJob> fwd, bwd;
  for (int i = 0; i < (int)pl->ops.size(); ++i) {
    OpRec& o = pl->ops[i];
    if (o.d.kind != S3_OP_CONV) continue;
    const ConvGeom& g = o.cg;
    const bool k3 = g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3;
    if (o.mfma && o.packed && g.Cin == 64 && k3 && !conv_mfma_is_gen(g, pl->precision)) {
      S3PackJob j;
      j.w = W + P->p[o.d.w].offset;
      j.cout = g.Cout; j.n_ct = (g.Cout + 63) / 64; j.dgrad = 0;
      j.tile = (unsigned short*)o.packed;
      j.persist = conv_mfma_persist_geom_ok(g) ? j.tile + (size_t)j.n_ct * 27 * 64 * 64 : nullptr;
      fwd.push_back(j); pl->pack_fwd_ops.push_back(i);
      pl->pack_fwd_ct = std::max(pl->pack_fwd_ct, j.n_ct);
    }
    if (pl->training && o.dgrad_mfma && !o.dgrad_gen && !o.dgrad_fewch && !o.dgrad_chunked && o.dg_wbf && g.Cout == 64 && k3 &&
        o.dg.Cin == 64) {
      S3PackJob j;
      j.w = W + P->p[o.d.w].offset;
      j.cout = g.Cin; j.n_ct = (g.Cin + 63) / 64; j.dgrad = 1;
      j.tile = (unsigned short*)o.dg_wbf;
      j.persist = conv_mfma_persist_dgrad_geom_ok(o.dg) ? j.tile + (size_t)j.n_ct * 27 * 64 * 64 : nullptr;
      bwd.push_back(j); pl->pack_bwd_ops.push_back(i);
      pl->pack_bwd_ct = std::max(pl->pack_bwd_ct, j.n_ct);
    }
  }
  if (fwd.size() >= 2) {
    int rc = plan_alloc(pl, (void**)&pl->pack_fwd, fwd.size() * sizeof(S3PackJob));
    if (rc) return rc;
    S3_HIP(ctx, hipMemcpyAsync(pl->pack_fwd, fwd.data(), fwd.size() * sizeof(S3PackJob), hipMemcpyHostToDevice, ctx->stream));
    S3_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the host vector goes away)
  } else {
    pl->pack_fwd_ops.clear();
  }
  if (bwd.size() >= 2) {
    int rc = plan_alloc(pl, (void**)&pl->pack_bwd, bwd.size() * sizeof(S3PackJob));
    if (rc) return rc;
    S3_HIP(ctx, hipMemcpyAsync(pl->pack_bwd, bwd.data(), bwd.size() * sizeof(S3PackJob), hipMemcpyHostToDevice, ctx->stream));
    S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    pl->pack_bwd_ops.clear();
  }
  return S3_OK;
}

// re-pack every listed conv whose images are stale (all or none: the weights
// of a net change together)
static int pack_stale(s3_plan* pl, bool bwd) {
  if (!pl->pack_built) {
    int rc = pack_tables_build(pl);
    if (rc) return rc;
  }
  const std::vector<int>& ops = bwd ? pl->pack_bwd_ops : pl->pack_fwd_ops;
  if (ops.empty()) return S3_OK;
  const uint64_t ver = pl->params->version;
  bool stale = false;
  for (int i : ops) stale = stale || (bwd ? pl->ops[i].dg_version : pl->ops[i].packed_version) != ver;
  if (!stale) return S3_OK;
  int rc = launch_pack_jobs(pl->ctx, bwd ? pl->pack_bwd : pl->pack_fwd, (int)ops.size(),
                            bwd ? pl->pack_bwd_ct : pl->pack_fwd_ct);
  if (rc) return rc;
  for (int i : ops) (bwd ? pl->ops[i].dg_version : pl->ops[i].packed_version) = ver;
  return S3_OK;
}

static int run_op_forward(s3_plan* pl, OpRec& o) {
  s3_ctx* ctx = pl->ctx;
  s3_params* P = pl->params;
  const s3_op_desc& d = o.d;
  float* W = P->buf[S3_BUF_W];
  const TensorRec& ot = pl->t[d.out];
  switch (d.kind) {
    case S3_OP_CONV: {
      const float* w = W + P->p[d.w].offset;
      const float* b = d.b >= 0 ? W + P->p[d.b].offset : nullptr;
      const float* res = d.res >= 0 ? tptr(pl, o.res_src >= 0 ? o.res_src : d.res) : nullptr;
      if (o.mfma) {
        if (o.packed_version != P->version) {
          int rc = launch_conv_mfma_pack(ctx, o.cg, pl->precision, w, o.packed);
          if (rc) return rc;
          o.packed_version = P->version;
        }
        const void* wp = pl->precision != S3_PREC_F32 ? (const void*)o.packed : (const void*)w;
        if (o.exo_src >= 0 || o.res2_src >= 0) {
          ConvGeom ge = o.cg;
          if (o.exo_src >= 0) ge.exo = (const float*)tptr(pl, o.exo_src);
          if (o.res2_src >= 0) ge.res2 = tptr(pl, o.res2_src);
          return launch_conv_mfma_fwd(ctx, ge, pl->precision, tptr(pl, d.in0), wp, b, res, tptr(pl, d.out), o.io);
        }
        return launch_conv_mfma_fwd(ctx, o.cg, pl->precision, tptr(pl, o.rep_src >= 0 ? o.rep_src : d.in0), wp, b,
                                    res, tptr(pl, d.out), o.io);
      }
      if (o.halo32 && !res) {
        if (o.h32_version != P->version) {
          int rc = launch_conv_halo32_pack(ctx, o.cg, w, o.h32_w);
          if (rc) return rc;
          o.h32_version = P->version;
        }
        return launch_conv_halo32_fwd(ctx, o.cg, tptr(pl, d.in0), o.h32_w, b, tptr(pl, d.out), o.io.in_bf16,
                                      o.io.out_bf16);
      }
      if (o.halo_s2 && !res && o.io.in_bf16) {
        if (o.h32_version != P->version) {
          int rc = launch_conv_halo_s2_pack(ctx, o.cg, w, o.h32_w);
          if (rc) return rc;
          o.h32_version = P->version;
        }
        return launch_conv_halo_s2_fwd(ctx, o.cg, tptr(pl, d.in0), o.h32_w, b, tptr(pl, d.out), o.io.out_bf16);
      }
      if (pl->win_op >= 0 && &o == &pl->ops[pl->win_op])   // s3_plan_forward_window: checked there
        return launch_conv_tail_mfma(ctx, pl->win_geom, tptr(pl, d.in0), w, b, (float*)tptr(pl, d.out), pl->win_aff);
      if (o.tail_x3 && !res && !o.io.in_bf16 && !o.io.out_bf16)
        return launch_conv_tail_x3(ctx, o.cg, (const float*)tptr(pl, d.in0), w, b, (float*)tptr(pl, d.out));
      if (o.gconv && (!o.io.in_bf16 || o.cg.Cin % 8 == 0) && !o.io.res_bf16 &&
          (!o.io.out_bf16 || o.cg.Cout % 4 == 0)) {
        if (o.gc_version != P->version) {
          int rc = launch_gconv_pack(ctx, o.cg, w, o.gc_w, 0, pl->precision == S3_PREC_BF16X3);
          if (rc) return rc;
          o.gc_version = P->version;
        }
        return launch_gconv_fwd(ctx, o.cg, (const float*)tptr(pl, d.in0), o.gc_w, b, res, tptr(pl, d.out), o.io.out_bf16, o.io.in_bf16,
                                pl->precision == S3_PREC_BF16X3, o.sign_bytes);
      }
      if (o.fewpos && o.fp_mfma && !o.io.in_bf16 && !o.io.out_bf16)
        return launch_conv_fewpos_mfma(ctx, o.cg, 0, tptr(pl, d.in0), w, b, res, tptr(pl, d.out));
      if (o.fewpos && !o.io.in_bf16 && !o.io.out_bf16)
        return launch_conv_fewpos_fwd(ctx, o.cg, tptr(pl, d.in0), w, b, res, tptr(pl, d.out), pl->fp_partial, pl->fp_partial_bytes);
      return launch_conv_generic_fwd(ctx, o.cg, tptr(pl, d.in0), w, b, res, tptr(pl, d.out), o.io.out_bf16, o.io.in_bf16);
    }
    case S3_OP_DENSE: {
      const TensorRec& it = pl->t[d.in0];
      const float* w = W + P->p[d.w].offset;
      const float* b = d.b >= 0 ? W + P->p[d.b].offset : nullptr;
      int rows = (int)(it.numel / it.dims[4]);
      return launch_dense_fwd(ctx, tptr(pl, d.in0), w, b, tptr(pl, d.out), rows, (int)it.dims[4], (int)ot.dims[4], d.act, d.alpha);
    }
    case S3_OP_REPEAT_T: case S3_OP_D2S: case S3_OP_PAD: case S3_OP_CROP:
    case S3_OP_ROLL_T: case S3_OP_DILATE:
      if (o.fused_away) return S3_OK;        // read through its consumer's halo index
      return launch_gather(ctx, o.gg, tptr(pl, d.in0), tptr(pl, d.out), tdtype(pl, d.out) ? 2 : 4);
    case S3_OP_CONCAT: {
      if (o.fused_away) return S3_OK;   // its consumer conv reads both operands (OpRec::exo_src)
      // two channel-range copies: x -> out[..., :Cx], exo -> out[..., Cx:]
      const TensorRec& a = pl->t[d.in0];
      const TensorRec& b = pl->t[d.in1];
      int64_t npos = ot.numel / ot.dims[4];
      int rc = s3_copy_channels(ctx, tptr(pl, d.in0), (int)a.dims[4], 0, tptr(pl, d.out), (int)ot.dims[4], 0, (int)a.dims[4], npos, 0);
      if (rc) return rc;
      return s3_copy_channels(ctx, tptr(pl, d.in1), (int)b.dims[4], 0, tptr(pl, d.out), (int)ot.dims[4], (int)a.dims[4], (int)b.dims[4], npos, 0);
    }
    case S3_OP_ADD:
      if (o.fused_away) return S3_OK;   // absorbed by the conv in front of it (OpRec::res2_src)
      if (ot.dtype) return launch_add16(ctx, tptr(pl, d.in0), tptr(pl, d.in1), tptr(pl, d.out), ot.numel);
      return launch_add(ctx, tptr(pl, d.in0), tptr(pl, d.in1), tptr(pl, d.out), ot.numel, (int)ot.dims[4], d.bcast_c);
    case S3_OP_ACT:
      return launch_act(ctx, tptr(pl, d.in0), tptr(pl, d.out), ot.numel, d.act, d.alpha);
    case S3_OP_VIEW:
      return S3_OK;
  }
  S3_FAIL(ctx, S3_EINVAL, "forward: unknown op");
}

static int bind_inputs(s3_plan* pl, const void* const* inputs) {
  for (size_t i = 0; i < pl->inputs.size(); ++i) {
    if (!inputs || !inputs[i]) S3_FAIL(pl->ctx, S3_EINVAL, "forward: null input pointer");
    pl->t[pl->inputs[i]].ptr = (float*)inputs[i];
  }
  return S3_OK;
}

// the op list of one forward on ctx->stream (with optional per-op events)
static int forward_ops(s3_plan* pl, hipEvent_t* ev) {
  s3_ctx* ctx = pl->ctx;
  const int n_ops = (int)pl->ops.size();
  {
    int prc = pack_stale(pl, false);
    if (prc) return prc;
  }
  if (ev) S3_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  for (int i = 0; i < n_ops; ++i) {
    int rc = run_op_forward(pl, pl->ops[i]);
    if (rc) {
      // (which launch: a failure inside a stream capture is otherwise anonymous)
      char where[96];
      snprintf(where, sizeof(where), " [forward op %d of %d, kind %d%s]", i, n_ops, pl->ops[i].d.kind,
               ctx->capturing ? ", capturing" : "");
      ctx->err += where;
      return rc;
    }
    if (ev) S3_HIP(ctx, hipEventRecord(ev[i + 1], ctx->stream));
  }
  return S3_OK;
}

// Optional (SUP3R_AMD_GRAPH=1): replay the forward as ONE hipGraph.  The first forwards run eagerly
// (they set kernel attributes, pack filters and size the scratch); the next
// one is captured on a private stream — the context stream may be the legacy
// null stream, which cannot capture — and replayed from then on.
static bool graph_wanted(const s3_plan* pl) {
  if (pl->training || pl->graph_off || pl->in_stage.empty()) return false;
  // opt-in: measured on MI355X / ROCm 7.2 the replay is bit-identical but not
  // faster (C1: 0.524 ms eager vs 0.535 ms replayed — the 36 dependent
  // micro-kernels cost ~14 us each on the GPU side either way)
  return s3_opt_on(S3O_GRAPH);
}

static int forward_graph(s3_plan* pl) {
  s3_ctx* ctx = pl->ctx;
  const uint64_t ver = pl->params->version;
  if (pl->graph_exec && pl->graph_version != ver) {
    graph_drop(pl);               // weights changed: repack eagerly, re-capture
    pl->eager_forwards = 0;
  }
  if (!pl->graph_exec) {
    if (pl->eager_forwards < 1) {
      pl->eager_forwards++;
      return forward_ops(pl, nullptr);
    }
    if (!pl->cap_stream &&
        hipStreamCreateWithFlags(&pl->cap_stream, hipStreamNonBlocking) != hipSuccess) {
      pl->graph_off = true;
      return forward_ops(pl, nullptr);
    }
    // everything queued so far must be visible to the replay
    S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipStream_t user = ctx->stream;
    ctx->stream = pl->cap_stream;
    hipError_t e = hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal);
    int rc = S3_OK;
    if (e == hipSuccess) {
      rc = forward_ops(pl, nullptr);
      e = hipStreamEndCapture(pl->cap_stream, &pl->graph);
    }
    ctx->stream = user;
    if (e == hipSuccess && rc == S3_OK)
      e = hipGraphInstantiate(&pl->graph_exec, pl->graph, nullptr, nullptr, 0);
    if (s3_opt_has(S3O_TRACE))
      fprintf(stderr, "[graph] capture of %d ops: %s\n", (int)pl->ops.size(),
              (e == hipSuccess && rc == S3_OK) ? "ok" : hipGetErrorString(e));
    if (e != hipSuccess || rc != S3_OK) {
      (void)hipGetLastError();
      graph_drop(pl);
      pl->graph_off = true;       // this plan stays on the eager path
      return forward_ops(pl, nullptr);
    }
    pl->graph_version = ver;
  }
  S3_HIP(ctx, hipGraphLaunch(pl->graph_exec, ctx->stream));
  return S3_OK;
}

extern "C" int s3_plan_forward(s3_plan* pl, const void* const* inputs, void* output) {
  if (!pl) return S3_EINVAL;
  s3_ctx* ctx = pl->ctx;
  S3OptScope opt_scope(&pl->opt);
  const int n_ops = (int)pl->ops.size();
  hipEvent_t* ev = nullptr;
  if (pl->prof_cap > 0 && pl->prof_n < pl->prof_cap)
    ev = pl->prof_ev.data() + (size_t)pl->prof_n * (n_ops + 1);
  int rc;
  if (!ev && pl->fused2d && !s3_opt_has(S3O_NO_FUSED2D)) {
    rc = bind_inputs(pl, inputs);
    if (rc) return rc;
    float* dst = output ? (float*)output : tptr(pl, pl->output);
    rc = fused2d_run(ctx, pl->fused2d, pl->params->buf[S3_BUF_W], pl->params->version,
                     (const float*)inputs[0], dst);
    if (rc) {
      ctx->err += ctx->capturing ? " [fused2d forward, capturing]" : " [fused2d forward]";
      return rc;
    }
    pl->forward_done = true;
    return S3_OK;
  }
  if (!ev && graph_wanted(pl)) {
    for (size_t i = 0; i < pl->inputs.size(); ++i) {
      if (!inputs || !inputs[i]) S3_FAIL(ctx, S3_EINVAL, "forward: null input pointer");
      S3_HIP(ctx, hipMemcpyAsync(pl->in_stage[i], inputs[i],
                                 (size_t)pl->t[pl->inputs[i]].numel * sizeof(float),
                                 hipMemcpyDeviceToDevice, ctx->stream));
      pl->t[pl->inputs[i]].ptr = pl->in_stage[i];
    }
    rc = forward_graph(pl);
  } else {
    rc = bind_inputs(pl, inputs);
    if (rc) return rc;
    // Inference plans write the caller's buffer directly: the output tensor is
    // the last thing written and nothing of the plan reads it afterwards, so
    // the device-to-device copy below (472 MB per C2 forward of 32 chunks,
    // 966 MB per C3 batch of 16: ~1 % of the step) is not needed.  Training
    // plans keep their own copy (the backward pass reads it).
    // (the output may be a view — a reshape — of the tensor the last op writes)
    TensorRec& ot = pl->t[root_of(pl, pl->output)];
    const bool direct = output && !pl->training && ot.buffer >= 0 && ot.dtype == 0 && !ot.is_input &&
                        ot.numel == pl->t[pl->output].numel && !s3_opt_has(S3O_NO_DIRECT_OUTPUT);
    if (pl->win_op >= 0 && !direct) S3_FAIL(ctx, S3_ESTATE, "forward_window: the output cannot be written in place");
    if (direct) ot.ptr = (float*)output;
    rc = forward_ops(pl, ev);
    if (direct) {
      ot.ptr = (float*)pl->buffers[ot.buffer];
      if (rc) return rc;
      if (ev) pl->prof_n++;
      pl->forward_done = true;
      return S3_OK;
    }
  }
  if (rc) return rc;
  if (ev) pl->prof_n++;
  if (output) {
    S3_HIP(ctx, hipMemcpyAsync(output, tptr(pl, pl->output),
                               (size_t)pl->t[pl->output].numel * sizeof(float),
                               hipMemcpyDeviceToDevice, ctx->stream));
  }
  pl->forward_done = true;
  return S3_OK;
}

static void prof_free(s3_plan* pl) {
  for (auto& e : pl->prof_ev) (void)hipEventDestroy(e);
  pl->prof_ev.clear();
  pl->prof_cap = 0;
  pl->prof_n = 0;
}

extern "C" int s3_plan_profile_begin(s3_plan* pl, int max_forwards) {
  if (!pl || max_forwards < 1) return S3_EINVAL;
  s3_ctx* ctx = pl->ctx;
  prof_free(pl);
  const size_t n = (size_t)max_forwards * (pl->ops.size() + 1);
  pl->prof_ev.resize(n);
  for (auto& e : pl->prof_ev) S3_HIP(ctx, hipEventCreate(&e));
  pl->prof_cap = max_forwards;
  return S3_OK;
}

extern "C" int s3_plan_profile_end(s3_plan* pl, float* ms_per_op, int cap) {
  if (!pl || !ms_per_op) return S3_EINVAL;
  s3_ctx* ctx = pl->ctx;
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n_ops = (int)pl->ops.size();
  const int nf = pl->prof_n;
  for (int i = 0; i < n_ops && i < cap; ++i) {
    double acc = 0.0;
    for (int f = 0; f < nf; ++f) {
      hipEvent_t* ev = pl->prof_ev.data() + (size_t)f * (n_ops + 1);
      float ms = 0.f;
      S3_HIP(ctx, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
      acc += ms;
    }
    ms_per_op[i] = nf ? (float)(acc / nf) : 0.f;
  }
  prof_free(pl);
  return nf;
}

extern "C" int s3_plan_op_is_mfma(const s3_plan* pl, int i) {
  if (!pl || i < 0 || i >= (int)pl->ops.size()) return 0;
  S3OptScope opt_scope(&pl->opt);
  const auto& o = pl->ops[i];
  if (o.d.kind != S3_OP_CONV || !o.mfma) return 0;
  if (pl->precision == S3_PREC_BF16 &&
      conv_mfma_persist_supported(pl->ctx, o.cg, o.io, o.d.res >= 0))
    return 2;
  return 1;
}

extern "C" int s3_plan_tensor_dtype(const s3_plan* pl, int32_t id) {
  if (!pl || id < 0 || id >= (int)pl->t.size()) return S3_EINVAL;
  int r = id;
  while (pl->t[r].alias_root >= 0) r = pl->t[r].alias_root;
  // the whole-network kernel keeps every intermediate tensor in LDS as bf16
  if (pl->fused2d && !s3_opt_has(S3O_NO_FUSED2D)) {
    int out_r = pl->output;
    while (pl->t[out_r].alias_root >= 0) out_r = pl->t[out_r].alias_root;
    return (pl->t[r].is_input || r == out_r) ? 0 : 1;
  }
  return pl->t[r].dtype;
}

extern "C" int64_t s3_plan_tensor_read(s3_plan* pl, int32_t id, void* host, size_t cap) {
  if (!pl || !host || id < 0 || id >= (int)pl->t.size()) return S3_EINVAL;
  s3_ctx* ctx = pl->ctx;
  const TensorRec& t = pl->t[root_of(pl, id)];
  if (!t.ptr) S3_FAIL(ctx, S3_ESTATE, "tensor_read: tensor has no buffer yet");
  const size_t bytes = (size_t)pl->t[id].numel * (t.dtype ? 2 : 4);
  if (bytes > cap) S3_FAIL(ctx, S3_EINVAL, "tensor_read: host buffer too small");
  S3_HIP(ctx, hipStreamSynchronize(ctx->stream));
  S3_HIP(ctx, hipMemcpy(host, t.ptr, bytes, hipMemcpyDeviceToHost));
  return (int64_t)bytes;
}

// which forward kernel run_op_forward / launch_conv_generic_fwd picks for a conv
static int conv_fwd_kind(const s3_plan* pl, const OpRec& o) {
  const bool res = o.d.res >= 0;
  const bool bfp = pl->precision == S3_PREC_BF16;
  int fwd = S3_FWD_DIRECT;
  // mirrors run_op_forward / launch_conv_generic_fwd
  if (o.mfma) {
    fwd = conv_mfma_is_gen(o.cg, pl->precision)
              ? (conv2d_ws_supported(o.cg, pl->precision, o.io, res) ||
                 conv2d_ws_x3_supported(o.cg, pl->precision, o.io, res) ||
                 conv2d_out_supported(o.cg, pl->precision, o.io, res)    ? S3_FWD_CONV2D_WS
                 : conv2d_head_supported(o.cg, pl->precision, o.io, res) ? S3_FWD_CONV2D_HEAD
                                                                         : S3_FWD_MFMA_GEN)
          : bfp && conv_mfma_persist_supported(pl->ctx, o.cg, o.io, res) ? S3_FWD_MFMA_PERSIST : S3_FWD_MFMA_TILE;
  } else if (o.halo32 && !res) {
    fwd = S3_FWD_HALO32;
  } else if (o.halo_s2 && !res && o.io.in_bf16) {
    fwd = S3_FWD_HALO_S2;
  } else if (o.tail_x3 && !res && !o.io.in_bf16 && !o.io.out_bf16) {
    fwd = S3_FWD_TAIL_MFMA;
  } else if (o.gconv && (!o.io.in_bf16 || o.cg.Cin % 8 == 0) && !o.io.res_bf16 &&
             (!o.io.out_bf16 || o.cg.Cout % 4 == 0)) {
    fwd = o.cg.Cin <= 4 ? S3_FWD_GCONV_FEWCH : S3_FWD_GCONV;
  } else if (o.fewpos && !o.io.in_bf16 && !o.io.out_bf16) {
    fwd = S3_FWD_FEWPOS;
  } else if (o.io.in_bf16 && !o.io.out_bf16 && !res && conv_tail_mfma_supported(o.cg) &&
             !s3_opt_has(S3O_NO_TAIL_MFMA)) {
    fwd = S3_FWD_TAIL_MFMA;
  } else if (!o.io.out_bf16 && !res && conv_small_supported(o.cg, o.io.in_bf16)) {
    fwd = S3_FWD_SMALL;
  }
  return fwd;
}

extern "C" int s3_plan_op_info(const s3_plan* pl, int i, int32_t* out, int cap) {
  if (!pl || !out || i < 0 || i >= (int)pl->ops.size()) return S3_EINVAL;
  S3OptScope opt_scope(&pl->opt);   // the launch-time kernel switches are the PLAN's options
  const OpRec& o = pl->ops[i];
  int32_t v[S3_OPINFO_COUNT] = {0};
  v[S3_OPINFO_KIND] = o.d.kind;
  if (o.d.kind == S3_OP_CONV) {
    int fwd = conv_fwd_kind(pl, o);
    const bool fused = pl->fused2d && !pl->training && !s3_opt_has(S3O_NO_FUSED2D);
    if (fused) fwd = S3_FWD_FUSED2D;
    v[S3_OPINFO_FWD] = fwd;
    v[S3_OPINFO_IN16] = o.io.in_bf16; v[S3_OPINFO_OUT16] = o.io.out_bf16; v[S3_OPINFO_RES16] = o.io.res_bf16;
    v[S3_OPINFO_IN_REP] = o.cg.in_rep;
    v[S3_OPINFO_RES_REP] = o.cg.res_rep;
    // operands rounded to bf16 by the forward kernel
    v[S3_OPINFO_FWD_BF16_OPS] = (pl->precision == S3_PREC_BF16 &&
                                 (fwd == S3_FWD_FUSED2D || fwd == S3_FWD_MFMA_TILE || fwd == S3_FWD_MFMA_GEN || fwd == S3_FWD_CONV2D_WS || fwd == S3_FWD_CONV2D_HEAD || fwd == S3_FWD_MFMA_PERSIST || fwd == S3_FWD_HALO32 || fwd == S3_FWD_HALO_S2 ||
                                  fwd == S3_FWD_GCONV || fwd == S3_FWD_GCONV_FEWCH || fwd == S3_FWD_TAIL_MFMA)) ? 1 : 0;
    v[S3_OPINFO_FEWPOS_MFMA] = (o.fp_mfma || o.fp_wg_mfma) ? 1 : 0;
    if (pl->training) {
      int wg = S3_WGRAD_DIRECT;
      if (o.fewpos || (o.fewpos_wgrad && !o.io.in_bf16)) wg = S3_WGRAD_FEWPOS;
      else if (o.wgrad_tail) wg = S3_WGRAD_TAIL;
      else if (o.wgrad_c2) wg = S3_WGRAD_C2;
      else if (o.wgrad_bf16_2d) wg = S3_WGRAD_BF16_2D;
      else if (o.wgrad_bf16_gen) wg = S3_WGRAD_BF16_GEN;
      else if (o.wgrad_gen) wg = S3_WGRAD_F32_GEN;
      else if (o.wgrad_bf16) wg = S3_WGRAD_BF16_TRUNK;
      else if (o.wgrad_mfma) wg = S3_WGRAD_F32_TRUNK;
      v[S3_OPINFO_WGRAD] = wg;
      int dg = S3_DGRAD_DIRECT;
      if (o.dgrad_chunked) dg = S3_DGRAD_MFMA_CHUNKED;
      else if (o.dgrad_fewch) dg = S3_DGRAD_FEWCH_FRAME;
      else if (o.dgrad_valid) dg = S3_DGRAD_MFMA_VALID;
      else if (o.dgrad_mfma) dg = S3_DGRAD_MFMA_FRAME;
      else if (o.dgrad_s2) dg = S3_DGRAD_S2;
      else if (o.dgrad_c2) dg = S3_DGRAD_C2;
      else if (o.gconv_dgrad) dg = S3_DGRAD_GCONV;
      else if (o.fewpos && o.fp_wt) dg = S3_DGRAD_FEWPOS;
      v[S3_OPINFO_DGRAD] = dg;
      v[S3_OPINFO_DGRAD_FRAME16] = o.dgrad_frame16 ? 1 : 0;
      v[S3_OPINFO_MASK_FUSED_FROM] = o.mask_prod;
    }
  }
  if (o.d.kind == S3_OP_REPEAT_T || o.d.kind == S3_OP_CONCAT || o.d.kind == S3_OP_ADD)
    v[S3_OPINFO_IN_REP] = o.fused_away ? 1 : 0;
  for (int q = 0; q < cap && q < S3_OPINFO_COUNT; ++q) out[q] = v[q];
  return S3_OPINFO_COUNT;
}

// ---- windowed forward: the C3 executor's halo crop + un-normalisation inside
// the tail conv.  The last conv of the plan computes only the window
// [lo, lo + n) of its output positions — the chunk without its halo — applies
// y * scale + shift and writes the (N, n0, n1, n2, C) result densely into the
// caller's buffer: no full-size model output, no epilogue pass over it, and the
// tail conv skips the halo positions (24 % of them at 110 x 110 x 624 ->
// 100 x 100 x 576).  Only for plans whose last op is the bf16-input MFMA tail.
static int window_op(const s3_plan* pl) {
  if (pl->training || pl->ops.empty()) return -1;
  if (pl->fused2d && !s3_opt_has(S3O_NO_FUSED2D)) return -1;
  if (s3_opt_on(S3O_GRAPH) || s3_opt_has(S3O_NO_DIRECT_OUTPUT) || s3_opt_has(S3O_NO_TAIL_WINDOW)) return -1;
  int i = (int)pl->ops.size() - 1;
  while (i >= 0 && pl->ops[i].d.kind == S3_OP_VIEW) --i;
  if (i < 0) return -1;
  const OpRec& o = pl->ops[i];
  if (o.d.kind != S3_OP_CONV || o.d.res >= 0 || o.cg.d2s != 1 || !o.io.in_bf16 || o.io.out_bf16) return -1;
  if (conv_fwd_kind(pl, o) != S3_FWD_TAIL_MFMA) return -1;
  const int ro = root_of(pl, o.d.out);
  if (ro != root_of(pl, pl->output)) return -1;
  const TensorRec& ot = pl->t[ro];
  if (ot.buffer < 0 || ot.dtype != 0 || ot.is_input || ot.numel != pl->t[pl->output].numel) return -1;
  // nobody else writes or reads the output tensor
  for (int k = 0; k < (int)pl->ops.size(); ++k) {
    if (k == i) continue;
    const s3_op_desc& d = pl->ops[k].d;
    if (d.kind == S3_OP_VIEW) continue;
    for (int id : {d.in0, d.in1, d.res, d.out})
      if (id >= 0 && root_of(pl, id) == ro) return -1;
  }
  return i;
}

extern "C" int s3_plan_supports_window(const s3_plan* pl) {
  if (!pl) return 0;
  S3OptScope opt_scope(&pl->opt);
  return window_op(pl) >= 0 ? 1 : 0;
}

extern "C" int s3_plan_forward_window(s3_plan* pl, const void* const* inputs, void* output, const int64_t* lo3,
                                      const int64_t* n3, const float* affine_dev, int n_c) {
  if (!pl || !output || !lo3 || !n3) return S3_EINVAL;
  s3_ctx* ctx = pl->ctx;
  int wi;
  {
    S3OptScope opt_scope(&pl->opt);
    wi = window_op(pl);
  }
  if (wi < 0) S3_FAIL(ctx, S3_EINVAL, "forward_window: the plan's last op is not the MFMA tail conv of an inference plan");
  const OpRec& o = pl->ops[wi];
  if (affine_dev && n_c != o.cg.Cout) S3_FAIL(ctx, S3_EINVAL, "forward_window: affine channel count");
  ConvGeom g = o.cg;
  for (int d = 0; d < 3; ++d) {
    if (lo3[d] < 0 || n3[d] < 1 || lo3[d] + n3[d] > o.cg.O[d]) S3_FAIL(ctx, S3_EINVAL, "forward_window: window outside the output");
    g.O[d] = (int)n3[d];
    g.lo[d] = o.cg.lo[d] - (int)lo3[d] * o.cg.s[d];
  }
  pl->win_op = wi;
  pl->win_geom = g;
  pl->win_aff = affine_dev;
  const int rc = s3_plan_forward(pl, inputs, output);
  pl->win_op = -1;
  pl->win_aff = nullptr;
  return rc;
}

// deliver a gradient contribution `src` (numel floats) to tensor `id`.
// The first contribution that lives in another finished buffer (the gradient
// of a consumer's output: skip adds, residuals, views) is not copied: the
// tensor's gradient aliases it (state 2) until a second contribution arrives,
// which then lands as one add / in-place accumulate instead of copy + axpy.
static int grad_deliver(s3_plan* pl, int id, const float* src) {
  int r = root_of(pl, id);
  TensorRec& t = pl->t[r];
  s3_ctx* ctx = pl->ctx;
  if (!pl->gwritten[r]) {
    if (src == t.gptr) { pl->gwritten[r] = 1; return S3_OK; }
    pl->gsrc[r] = src;
    pl->gwritten[r] = 2;
    return S3_OK;
  }
  if (pl->gwritten[r] && pl->dpre16_for == r && src != t.gptr) {
    // the tensor changes: its bf16 copy is stale (a bf16-only tensor has no fp32 to add to)
    if (pl->dpre16_only) S3_FAIL(ctx, S3_ESTATE, "backward: second contribution to a bf16-only gradient");
    pl->dpre16_for = -1;
  }
  if (pl->gwritten[r] == 2) {
    if (pl->bsum_for == r) pl->bsum_for = -1;   // the tensor changes: its channel sums are stale
    const float* first = pl->gsrc[r];
    pl->gsrc[r] = nullptr;
    pl->gwritten[r] = 1;
    if (src == t.gptr) return launch_axpy(ctx, first, t.gptr, t.numel);
    return launch_add(ctx, first, src, t.gptr, t.numel, 1, 0);
  }
  if (src == t.gptr) return S3_OK;  // accumulated in place by the producer
  if (pl->bsum_for == r) pl->bsum_for = -1;
  return launch_axpy(ctx, src, t.gptr, t.numel);
}

// destination a backward kernel should write dL/d(tensor id) into
static float* grad_dest(s3_plan* pl, int id) {
  int r = root_of(pl, id);
  return pl->gwritten[r] == 1 ? pl->gtmp : pl->t[r].gptr;
}

// the finished gradient of tensor root r
static const float* grad_of(s3_plan* pl, int r) {
  return pl->gwritten[r] == 2 ? pl->gsrc[r] : pl->t[r].gptr;
}

static int plan_backward_impl(s3_plan* pl, const void* d_output, void* d_input, int need_wgrad,
                              int accumulate_wgrad);

// Option WGRAD_SIDE_STREAM.  A launch-bound backward pass is a chain of
// dependent launches (>= 4.6 us each on this part); the weight gradient of a
// conv is not on that chain — nothing in the pass reads it — so it can go to a
// side stream that forks off the compute stream where its operands are final
// and joins before s3_plan_backward returns (inside a stream capture: a
// parallel branch of the graph).  Measured on C1 (48 forks per mini-batch):
// 2.37 -> 2.90 ms eager, 2.38 -> 2.89 ms as a recorded graph — a cross-stream
// dependency costs more than the 6 us kernel it takes off the chain — so it is
// off unless asked for (profiles/r04/README.md).
static int wg_fork(s3_ctx* ctx, hipStream_t* side) {
  if (!ctx->wg_stream) {
    S3_HIP(ctx, hipStreamCreateWithFlags(&ctx->wg_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) S3_HIP(ctx, hipEventCreateWithFlags(&ctx->wg_ev[k], hipEventDisableTiming));
  }
  S3_HIP(ctx, hipEventRecord(ctx->wg_ev[0], ctx->stream));
  S3_HIP(ctx, hipStreamWaitEvent(ctx->wg_stream, ctx->wg_ev[0], 0));
  ctx->wg_forked = true;
  *side = ctx->wg_stream;
  return S3_OK;
}
static int wg_join(s3_ctx* ctx) {
  if (!ctx->wg_forked) return S3_OK;
  ctx->wg_forked = false;
  S3_HIP(ctx, hipEventRecord(ctx->wg_ev[1], ctx->wg_stream));
  S3_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->wg_ev[1], 0));
  return S3_OK;
}

extern "C" int s3_plan_backward(s3_plan* pl, const void* d_output, void* d_input,
                                int need_wgrad, int accumulate_wgrad) {
  if (!pl || !d_output) return S3_EINVAL;
  int rc = plan_backward_impl(pl, d_output, d_input, need_wgrad, accumulate_wgrad);
  const int jrc = wg_join(pl->ctx);      // (also on a failed pass: a capture must not end forked)
  if (rc == S3_OK) rc = jrc;
  if (rc != S3_OK && pl->params && (pl->params->armed || pl->params->reduced)) {
    pl->params->reduced = false;
    // an armed store must not outlive the backward pass it was armed for: the
    // next one on this store (a validation step, a non-sharded step) would
    // enqueue collectives the other ranks never issue
    pl->params->armed = false;
    pl->params->reduce_end = 0;
    pl->params->buckets_issued = 0;
  }
  return rc;
}

static int plan_backward_impl(s3_plan* pl, const void* d_output, void* d_input, int need_wgrad,
                              int accumulate_wgrad) {
  s3_ctx* ctx = pl->ctx;
  S3OptScope opt_scope(&pl->opt);
  if (!pl->training) S3_FAIL(ctx, S3_ESTATE, "backward on an inference plan");
  if (!pl->forward_done) S3_FAIL(ctx, S3_ESTATE, "backward before forward");
  s3_params* P = pl->params;
  float* W = P->buf[S3_BUF_W];
  float* G = P->buf[S3_BUF_G];
  // (gradients about to be rewritten: a reduction nobody joined is moot)
  if (need_wgrad) P->reduced = false;
  if (need_wgrad && P->armed) {
    // the bucketed reduction hands over "everything at or above this op's
    // lowest offset" as the walk passes an op: true only if the parameter
    // offsets grow with the op order and no parameter is shared between ops.
    // Checked here, once per armed pass; a store laid out any other way gets
    // ONE reduction of the whole buffer after the last op instead.
    int64_t prev_end = 0;
    bool monotone = true;
    for (size_t i = 0; i < pl->ops.size() && monotone; ++i) {
      const s3_op_desc& d = pl->ops[i].d;
      int64_t lo = INT64_MAX, hi = -1;
      for (int id : {d.w, d.b}) {
        if (id < 0) continue;
        lo = std::min(lo, P->p[id].offset);
        hi = std::max(hi, P->p[id].offset + P->p[id].size);
      }
      if (hi < 0) continue;
      if (lo < prev_end) monotone = false;
      prev_end = hi;
    }
    if (!monotone) P->bucket_elems = P->total + 1;
  }
  std::fill(pl->gwritten.begin(), pl->gwritten.end(), 0);
  pl->premasked.assign(pl->gwritten.size(), 0);
  pl->gsrc.assign(pl->gwritten.size(), nullptr);
  pl->bsum_for = -1;
  pl->dpre16_for = -1;
  {
    // the caller's buffer is read-only for the duration of the call: alias it
    int r = root_of(pl, pl->output);
    pl->gsrc[r] = (const float*)d_output;
    pl->gwritten[r] = 2;
  }
  {
    int prc = pack_stale(pl, true);
    if (prc) return prc;
  }
  const int x_id = pl->inputs.empty() ? -1 : root_of(pl, pl->inputs[0]);
  // conv `prod` is processed right after conv `cons` in this reverse walk
  // (nothing but views in between): a bf16-ONLY dPre handed from one to the
  // other, with its channel sums in pl->bsum, cannot be clobbered on the way
  auto back_to_back = [&](int prod, int cons) {
    if (prod < 0 || prod >= cons) return false;
    for (int k = prod + 1; k < cons; ++k)
      if (pl->ops[k].d.kind != S3_OP_VIEW) return false;
    return true;
  };
  auto wants_grad = [&](int id) {
    int r = root_of(pl, id);
    if (!pl->t[r].is_input) return true;
    return r == x_id && d_input != nullptr;
  };
  for (int i = (int)pl->ops.size() - 1; i >= 0; --i) {
    OpRec& o = pl->ops[i];
    const s3_op_desc& d = o.d;
    if (d.kind == S3_OP_VIEW) continue;
    const int ro = root_of(pl, d.out);
    if (!pl->gwritten[ro]) continue;  // nothing flows through this op
    const float* dy = grad_of(pl, ro);
    const TensorRec& ot = pl->t[d.out];
    int rc = S3_OK;
    switch (d.kind) {
      case S3_OP_CONV: {
        const ConvGeom& g = o.cg;
        if (d.res >= 0 && wants_grad(d.res)) {
          rc = grad_deliver(pl, d.res, dy);
          if (rc) return rc;
        }
        const float* dpre = dy;
        const void* dpre16 = nullptr;     // bf16 copy of dpre, if one was left behind
        bool mask_sums = false;           // pl->bsum2 holds the channel sums of dpre
        // dPre written as bf16 ONLY by the consumer's frame fold (see fold_frame):
        // the fp32 buffer behind `dpre` holds nothing — every reader below takes
        // dpre16 (the fold made sure they all can)
        const bool only16 = pl->dpre16_for == ro && pl->dpre16_only && pl->premasked[ro];
        if (only16) {
          dpre16 = pl->dpre16;
          pl->dpre16_for = -1;
          const bool trunk16 = o.use16 && (o.wgrad_bf16 || (o.wgrad_bf16_2d && !o.fewpos && o.io.in_bf16 &&
                                                            g.s[0] == 1 && (g.Cout & 3) == 0));
          const bool fewch16 = o.wgrad_c2 && (o.dgrad_c2 || !wants_grad(d.in0));
          if ((!trunk16 && !fewch16) || d.res >= 0)
            S3_FAIL(ctx, S3_ESTATE, "backward: bf16-only dPre reached a conv that needs fp32");
        } else if (pl->dpre16_for == ro && !pl->dpre16_only) {
          // fp32 tensor + bf16 copy (fold + earlier contribution of a skip tensor):
          // dPre = dy for a conv without activation
          if (o.use16 && g.act == S3_ACT_NONE && g.d2s <= 1 && dy == pl->t[ro].gptr) dpre16 = pl->dpre16;
          pl->dpre16_for = -1;
        }
        // one-launch fewpos kernels: the activation's adjoint is applied to dy as
        // the weight / data gradient kernels read it (from y, like the mask pass)
        const float* fp_mask_y = nullptr;
        float fp_slope = 0.f;
        // (both readers of dPre must be the one-launch kernels: the data gradient
        // of a fewpos conv may still run on another family)
        const bool fp_dg = o.fewpos && o.fp_mfma && !o.dgrad_chunked && !o.dgrad_mfma && !o.dgrad_s2 &&
                           !o.dgrad_c2 && !o.gconv_dgrad;
        if (o.fewpos && o.fp_mfma && (fp_dg || !wants_grad(d.in0)) &&
            g.d2s <= 1 && !o.io.out_bf16 && pl->t[ro].dtype == 0 &&
            (g.act == S3_ACT_LEAKY || g.act == S3_ACT_RELU) && !pl->premasked[ro] && !only16 &&
            !s3_opt_has(S3O_NO_MASK_FUSE)) {
          fp_mask_y = (const float*)tptr(pl, d.out);
          fp_slope = g.act == S3_ACT_LEAKY ? g.alpha : 0.f;
        }
        if ((g.act != S3_ACT_NONE || g.d2s > 1) && !pl->premasked[ro] && !fp_mask_y) {
          // (never over a pending bf16-only dPre of another tensor)
          void* side = (o.use16 && pl->dpre16 && pl->dpre16_for < 0 && conv_epilogue_bwd_d16_ok(g) &&
                        (g.d2s <= 1 || o.io.out_bf16)) ? pl->dpre16 : nullptr;
          // the bias gradient = channel sums of dpre: they ride along this pass
          mask_sums = need_wgrad && d.b >= 0 && pl->bsum2 && conv_epilogue_bwd_bsum_ok(g) &&
                      (g.d2s <= 1 || (side && o.io.out_bf16)) && !s3_opt_has(S3O_NO_BIAS_FUSE);
          // Every reader of this dPre takes the bf16 copy — transpose-read /
          // wave-specialised weight gradient, MFMA data gradient over the
          // frame, bias gradient from the channel sums riding along: the
          // fp32 dPre (151 MB per trunk conv at C2 batch 8) is not written.
          const int64_t n_el = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2] * g.Cout;
          const bool wg16 = o.wgrad_bf16 && !o.fewpos && !o.wgrad_tail && !o.wgrad_c2 && !o.wgrad_bf16_2d &&
                            !o.wgrad_bf16_gen && !o.wgrad_gen && o.io.in_bf16 && (g.Cout & 3) == 0;
          const bool dg16 = o.dgrad_mfma && !o.dgrad_chunked && !o.dgrad_fewch && o.use16;
          // (64 -> C_out > 64 + depth-to-space: the slices of the chunked data
          // gradient read the bf16 copy when they run on the persistent kernel)
          const int nk16 = (g.Cout + 63) / 64;
          const bool dgc16 = o.dgrad_chunked && o.use16 && side && (g.Cout & 7) == 0 &&
                             conv_mfma_persist_dgrad_supported(ctx, conv_dgrad_chunk_geom(g, 0)) &&
                             conv_mfma_persist_dgrad_geom_ok(conv_dgrad_chunk_geom(g, nk16 - 1));
          const bool skip32 = side && pl->precision == S3_PREC_BF16 &&
                              (g.d2s <= 1 ? ((n_el & 3) == 0 && (g.act == S3_ACT_LEAKY || g.act == S3_ACT_RELU))
                                          : o.io.out_bf16) &&
                              (!need_wgrad || (wg16 && (d.b < 0 || mask_sums))) &&
                              (!wants_grad(d.in0) || dg16 || dgc16) && !s3_opt_has(S3O_NO_DPRE16_ONLY_MASK);
          rc = launch_conv_epilogue_bwd(ctx, g, tptr(pl, d.out), dy, skip32 ? nullptr : pl->dpre, o.io.out_bf16,
                                        side, mask_sums ? pl->bsum2 : nullptr);
          if (rc) return rc;
          dpre = skip32 ? nullptr : pl->dpre;
          dpre16 = side;
        }
        const int64_t npos = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
        // few positions: the one-launch weight gradient leaves the bias gradient too
        const bool fp_wg = dpre != nullptr &&
                           ((o.fewpos && o.fp_mfma) || (o.fp_wg_mfma && !o.fewpos && !o.wgrad_tail && !o.wgrad_c2 && !o.wgrad_bf16_2d && !o.wgrad_bf16_gen &&
                                         !o.wgrad_gen && !o.wgrad_bf16 && !o.wgrad_mfma && o.fewpos_wgrad && !o.io.in_bf16));
        // ... and when its data gradient is the one-launch kernel too, both go
        // out as ONE launch (at the data gradient's place below)
        const ConvGeom fp_gd = g.pad_mode == S3_PAD_REFLECT ? conv_fewpos_frame_geom(g) : g;
        const bool fp_both = need_wgrad && fp_wg && fp_dg && wants_grad(d.in0) &&
                             !s3_opt_has(S3O_WGRAD_SIDE_STREAM) && !s3_opt_has(S3O_NO_FEWPOS_BWD_FUSE) &&
                             conv_fewpos_bwd_mfma_ok(ctx, g, fp_gd);
        if (need_wgrad && fp_wg && !fp_both) {
          // beside the data-gradient chain when dPre is a tensor's own gradient
          // buffer (final by now; the shared scratch buffers are rewritten by
          // the ops that follow) and no collective reads G under this pass
          const bool side = dpre != pl->dpre && dpre != pl->gtmp && dpre != pl->dxp && !ctx->comm &&
                            !(P->armed || P->reduced) && s3_opt_has(S3O_WGRAD_SIDE_STREAM);
          hipStream_t main_stream = ctx->stream, ws = nullptr;
          if (side) {
            rc = wg_fork(ctx, &ws);
            if (rc) return rc;
            ctx->stream = ws;
          }
          rc = launch_conv_fewpos_wgrad_mfma(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset,
                                             d.b >= 0 ? G + P->p[d.b].offset : nullptr, accumulate_wgrad,
                                             fp_mask_y, fp_slope);
          ctx->stream = main_stream;
          if (rc) return rc;
        } else if (need_wgrad && !fp_both) {
          if (d.b >= 0) {
            // (its launch rides along the reduction of a bf16-family weight gradient)
            const bool ride = !o.fewpos && !o.wgrad_tail && !o.wgrad_c2 &&
                              (o.wgrad_bf16_2d || o.wgrad_bf16_gen || (!o.wgrad_gen && o.wgrad_bf16));
            if (mask_sums)
              rc = launch_bias_grad_from_partial(ctx, pl->bsum2, conv_epilogue_bwd_blocks(ctx, g, true), g.Cout,
                                                 G + P->p[d.b].offset, accumulate_wgrad, ride);
            else if (pl->bsum_for == ro && dpre == pl->t[ro].gptr && pl->gwritten[ro] == 1)
              rc = launch_bias_grad_from_partial(ctx, pl->bsum, pl->bsum_nblk, g.Cout, G + P->p[d.b].offset,
                                                 accumulate_wgrad, ride);
            else if (only16)
              S3_FAIL(ctx, S3_ESTATE, "backward: bf16-only dPre without its channel sums");
            else
              rc = launch_bias_grad(ctx, dpre, npos, g.Cout, G + P->p[d.b].offset, accumulate_wgrad);
            if (rc) return rc;
          }
          if (o.fewpos)
            rc = launch_conv_fewpos_wgrad(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad);
          else if (o.wgrad_tail)
            rc = launch_conv_wgrad_tail(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad, o.io.in_bf16);
          else if (o.wgrad_c2)
            rc = launch_conv_wgrad_c2(ctx, g, tptr(pl, d.in0), only16 ? (const float*)dpre16 : dpre, G + P->p[d.w].offset,
                                      pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad, only16 ? 1 : 0,
                                      pl->precision == S3_PREC_BF16X3);
          else if (o.wgrad_bf16_2d)
          {
            // (bf16-only dPre out of the consumer's masked fold, or a bf16 copy next to the fp32 one)
            const bool dy16 = only16 || (dpre16 && o.io.in_bf16 && g.s[0] == 1 && (g.Cout & 3) == 0 && (g.Cin & 7) == 0);
            rc = launch_conv_wgrad_bf16_2d(ctx, g, tptr(pl, d.in0), dy16 ? (const float*)dpre16 : dpre, G + P->p[d.w].offset,
                                           pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad, o.io.in_bf16, dy16 ? 1 : 0);
          }
          else if (o.wgrad_bf16_gen)
            rc = launch_conv_wgrad_bf16_gen(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad, o.io.in_bf16,
                                            pl->precision == S3_PREC_BF16X3);
          else if (o.wgrad_gen)
            rc = launch_conv_wgrad_gen(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad);
          else if (o.wgrad_bf16)
          {
            const bool dy16 = only16 || (dpre16 && o.io.in_bf16 && (g.Cout & 3) == 0);
            rc = launch_conv_wgrad_bf16(ctx, g, tptr(pl, d.in0), dy16 ? (const float*)dpre16 : dpre, G + P->p[d.w].offset,
                                        pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad, o.io.in_bf16, dy16 ? 1 : 0,
                                        pl->precision == S3_PREC_BF16X3 ? 1 : 0);
          }
          else if (o.wgrad_mfma)
            rc = launch_conv_wgrad_mfma(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad);
          else if (o.fewpos_wgrad && !o.io.in_bf16)
            rc = launch_conv_fewpos_wgrad(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad);
          else
            rc = launch_conv_generic_wgrad(ctx, g, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, pl->wg_partial, pl->wg_partial_bytes, accumulate_wgrad);
          if (rc) return rc;
          rc = s3_flush_pending_bias(ctx);     // (nothing took it along)
          if (rc) return rc;
        }
        if (wants_grad(d.in0)) {
          float* dst = grad_dest(pl, d.in0);
          // fold of the padded-frame data gradient; when this conv is the only
          // consumer of an activated conv's output the fold applies that
          // activation's adjoint (the producer then skips its mask pass)
          auto fold_frame = [&](const GatherGeom& fg, float* out, int frame16 = 0) -> int {
            const int rin = root_of(pl, d.in0);
            const bool fuse = o.mask_prod >= 0 && !pl->gwritten[rin] && gather_bwd_mask_ok(fg) &&
                              !s3_opt_has(S3O_NO_MASK_FUSE);
            // the stored tensor is (so far) the whole gradient of d.in0: its
            // channel sums = the bias gradient of the conv that produced it
            // ride along (grad_deliver drops them if the tensor changes later)
            float* bs = nullptr;
            if (need_wgrad && pl->bsum && out == pl->t[rin].gptr && gather_bwd_bsum_ok(fg) &&
                gather_bwd_bsum_blocks(ctx, fg) <= 4096 && !s3_opt_has(S3O_NO_BIAS_FUSE)) {
              bs = pl->bsum;
              pl->bsum_for = rin;
              pl->bsum_nblk = gather_bwd_bsum_blocks(ctx, fg);
            }
            if (!fuse && pl->gwritten[rin] == 2 && out == pl->t[rin].gptr && gather_bwd_mask_ok(fg)) {
              // second contribution to a skip tensor: fold + the aliased first
              // one in a single store (no staging buffer, no axpy)
              const float* first = pl->gsrc[rin];
              pl->gsrc[rin] = nullptr;
              pl->gwritten[rin] = 1;
              // the producer of this (now finished) skip tensor is a conv without
              // activation whose gradient kernels stage bf16: leave a bf16 copy
              void* side = nullptr;
              if (o.in_prod >= 0 && pl->dpre16 && pl->dpre16_for < 0 && !s3_opt_has(S3O_NO_FOLD16)) {
                const OpRec& po = pl->ops[o.in_prod];
                // (the size test was missing here until the end of round 6: with a dPre16 buffer sized by a
                // smaller tensor — sup3rcc/gen_solar_1x_8x_1f at 8 or 16 samples of (54, 54, 3), whose trunk's
                // data gradient then runs on the persistent kernel — the side copy wrote past its end:
                // non-finite gradients at batch 8, a memory access fault at batch 16)
                if (po.use16 && po.cg.act == S3_ACT_NONE && po.cg.d2s <= 1 && (po.cg.Cout & 3) == 0 &&
                    pl->dpre16_bytes >= (size_t)pl->t[rin].numel * 2)
                  side = pl->dpre16;
              }
              int arc = launch_gather_bwd_add(ctx, fg, pl->dxp, out, first, bs, side, frame16);
              if (!arc && side) { pl->dpre16_for = rin; pl->dpre16_only = false; }
              return arc;
            }
            if (!fuse) {
              if (bs) pl->bsum_for = -1;   // plain fold: no channel sums
              // (so far) the whole gradient of a tensor whose producer is a conv
              // without activation that stages bf16: leave it a bf16 copy
              // (grad_deliver drops the copy if a second contribution arrives)
              void* side = nullptr;
              if (o.in_prod >= 0 && out == pl->t[rin].gptr && !pl->gwritten[rin] && pl->dpre16 &&
                  pl->dpre16_for < 0 && gather_bwd_mask_ok(fg) && !s3_opt_has(S3O_NO_FOLD16) &&
                  !s3_opt_has(S3O_NO_PLAIN_FOLD16)) {
                const OpRec& po = pl->ops[o.in_prod];
                if (po.use16 && po.cg.act == S3_ACT_NONE && po.cg.d2s <= 1 && (po.cg.Cout & 3) == 0 &&
                    pl->dpre16_bytes >= (size_t)pl->t[rin].numel * 2)
                  side = pl->dpre16;
              }
              int prc = launch_gather_bwd(ctx, fg, pl->dxp, out, side, frame16);
              if (!prc && side) { pl->dpre16_for = rin; pl->dpre16_only = false; }
              return prc;
            }
            const OpRec& po = pl->ops[o.mask_prod];
            const ConvGeom& pg = po.cg;
            // The folded tensor is dPre of the producer conv and nothing else
            // (single consumer, mask applied here).  When every reader of it
            // takes bf16 — halo-tile / persistent data gradient, transpose-read
            // weight gradient, bias gradient from the channel sums riding
            // along — it is stored as bf16 ONLY: the fold writes 75 instead of
            // 151 MB and the readers stage half the bytes; they would round to
            // bf16 (the same round-to-nearest-even) anyway.
            const bool to16 = out == pl->t[rin].gptr && back_to_back(o.mask_prod, i) && po.use16 &&
                              (po.wgrad_bf16 || (po.wgrad_bf16_2d && !po.fewpos && !po.wgrad_tail && !po.wgrad_c2 &&
                                                 po.cg.s[0] == 1 && !s3_opt_has(S3O_NO_TRAIN2D_BF16))) &&
                              po.io.in_bf16 && po.d.res < 0 &&
                              (po.cg.Cout & 3) == 0 && pl->dpre16 && pl->dpre16_for < 0 &&
                              (!need_wgrad || po.d.b < 0 || bs != nullptr) && pl->precision == S3_PREC_BF16;
            int frc = launch_gather_bwd_masked(ctx, fg, pl->dxp, to16 ? (float*)pl->dpre16 : out, tptr(pl, d.in0),
                                               pl->t[rin].dtype, pg.act == S3_ACT_LEAKY ? pg.alpha : 0.f, bs,
                                               to16 ? 1 : 0, frame16);
            if (!frc) pl->premasked[rin] = 1;
            if (!frc && to16) { pl->dpre16_for = rin; pl->dpre16_only = true; }
            return frc;
          };
          if (o.dgrad_chunked) {
            // 64-channel slices of dPre through the 64 -> 64 halo-tile kernel,
            // accumulated in place over the padded frame, then the fold
            const int nk = (g.Cout + 63) / 64;
            if (o.dg_version != P->version) {
              for (int k = 0; k < nk; ++k) {
                rc = launch_conv_dgrad_chunk_pack(ctx, g, W + P->p[d.w].offset, o.dg_w32, k);
                if (!rc) rc = launch_conv_mfma_pack(ctx, conv_dgrad_chunk_geom(g, k), pl->precision, o.dg_w32, o.dgc_wbf[k]);
                if (rc) return rc;
              }
              o.dg_version = P->version;
            }
            float* acc_to = o.dgrad_valid ? dst : pl->dxp;   // valid padding: x's own grid
            // with the bf16 copy of dPre: the slices go through the persistent
            // kernel (stacked frames, the later slices add in its store)
            const bool p16 = o.use16 && dpre16 && conv_mfma_persist_dgrad_supported(ctx, conv_dgrad_chunk_geom(g, 0)) &&
                             conv_mfma_persist_dgrad_geom_ok(conv_dgrad_chunk_geom(g, nk - 1));
            for (int k = 0; k < nk; ++k) {
              const ConvGeom cgk = conv_dgrad_chunk_geom(g, k);
              if (p16)
                rc = launch_conv_mfma_persist_dgrad(ctx, cgk, (const unsigned short*)dpre16 + 64 * k,
                                                    (const char*)o.dgc_wbf[k] + (size_t)27 * 64 * 64 * 2, acc_to, k ? 1 : 0);
              else
                rc = launch_conv_mfma_fwd(ctx, cgk, pl->precision, dpre + 64 * k, o.dgc_wbf[k],
                                          nullptr, k ? acc_to : nullptr, acc_to, ConvIO());
              if (rc) return rc;
            }
            if (o.dgrad_valid) {
              rc = grad_deliver(pl, d.in0, dst);
              if (rc) return rc;
              break;
            }
            GatherGeom fg;
            fg.kind = S3_OP_PAD; fg.N = g.N;
            for (int q = 0; q < 3; ++q) { fg.Di[q] = g.D[q]; fg.Do[q] = g.D[q] + 2; fg.lo[q] = 1; }
            fg.Ci = g.Cin; fg.Co = g.Cin; fg.pad_mode = g.pad_mode;
            fg.rep = 1; fg.d2s = 1; fg.c_off = 0;
            rc = fold_frame(fg, dst);
          } else if (o.dgrad_mfma) {
            // dXpad = conv_zero(dPre, flip(W)^T) over the padded frame, then
            // the adjoint of the virtual padding folds the border back
            if (o.dg_version != P->version) {
              rc = launch_conv_dgrad_pack(ctx, g, W + P->p[d.w].offset, o.dg_w32);
              if (!rc && o.dgrad_fewch)
                rc = launch_gconv_pack(ctx, o.dg, o.dg_w32, o.dg_wbf, 0, pl->precision == S3_PREC_BF16X3);
              else if (!rc && pl->precision != S3_PREC_F32)
                rc = launch_conv_mfma_pack(ctx, o.dg, pl->precision, o.dg_w32, o.dg_wbf);
              if (rc) return rc;
              o.dg_version = P->version;
            }
            const void* wp = pl->precision != S3_PREC_F32 ? (const void*)o.dg_wbf : (const void*)o.dg_w32;
            int frame16 = 0;
            if (o.dgrad_fewch)
              rc = launch_gconv_fwd(ctx, o.dg, dpre, o.dg_wbf, nullptr, nullptr, pl->dxp, 0, 0, pl->precision == S3_PREC_BF16X3);
            else if (o.use16 && dpre16 && pl->precision == S3_PREC_BF16 &&
                     conv_mfma_persist_dgrad_supported(ctx, o.dg)) {
              // the persistent trunk kernel over the stacked frames (a valid conv's
              // full correlation lands on x's own grid: straight into dst)
              const size_t tile_img = (size_t)((o.dg.Cout + 63) / 64) * 27 * 64 * 64 * 2;
              frame16 = o.dgrad_frame16;
              rc = launch_conv_mfma_persist_dgrad(ctx, o.dg, dpre16, (const char*)o.dg_wbf + tile_img,
                                                  o.dgrad_valid ? dst : pl->dxp, 0, frame16);
            } else {
              ConvIO dio;
              dio.in_bf16 = (o.use16 && dpre16) ? 1 : 0;
              // 2-D 64 -> 64 k convs: the frame form of the weights-stationary
              // kernel, bf16 dPre in, bf16 frame out (folded from bf16)
              if (dio.in_bf16 && o.dgrad_gen && o.dgrad_frame16 && !o.dgrad_valid && pl->precision == S3_PREC_BF16) {
                ConvIO wio = dio;
                wio.out_bf16 = 1;
                if (conv2d_ws_supported(o.dg, pl->precision, wio, false)) { dio = wio; frame16 = 1; }
              }
              rc = launch_conv_mfma_fwd(ctx, o.dg, pl->precision, dio.in_bf16 ? dpre16 : (const void*)dpre, wp, nullptr,
                                        nullptr, o.dgrad_valid ? dst : pl->dxp, dio);
            }
            if (rc) return rc;
            if (o.dgrad_valid) {
              rc = grad_deliver(pl, d.in0, dst);
              if (rc) return rc;
              break;
            }
            GatherGeom fg;
            fg.kind = S3_OP_PAD; fg.N = g.N;
            for (int q = 0; q < 3; ++q) {
              const int pq = g.k[q] == 3 ? 1 : 0;      // (k = 1 axes of a 2-D conv carry no frame)
              fg.Di[q] = g.D[q]; fg.Do[q] = g.D[q] + 2 * pq; fg.lo[q] = pq;
            }
            fg.Ci = g.Cin; fg.Co = g.Cin; fg.pad_mode = g.pad_mode;
            fg.rep = 1; fg.d2s = 1; fg.c_off = 0;
            rc = fold_frame(fg, dst, frame16);
          } else if (o.dgrad_s2 && pl->precision == S3_PREC_BF16X3) {
            if (o.dc2_version != (int64_t)P->version) {
              rc = launch_conv_dgrad_s2_x3_pack(ctx, g, W + P->p[d.w].offset, o.dc2_w);
              if (rc) return rc;
              o.dc2_version = (int64_t)P->version;
            }
            const int rin = root_of(pl, d.in0);
            const bool fuse = o.mask_prod >= 0 && !pl->gwritten[rin] && !s3_opt_has(S3O_NO_MASK_FUSE) &&
                              pl->t[rin].dtype == 0;
            const ConvGeom& pg = pl->ops[fuse ? o.mask_prod : i].cg;
            rc = launch_conv_dgrad_s2_x3(ctx, g, dpre, o.dc2_w, dst, fuse ? (const float*)tptr(pl, d.in0) : nullptr,
                                         pg.act == S3_ACT_LEAKY ? pg.alpha : 0.f);
            if (!rc && fuse) pl->premasked[rin] = 1;
          } else if (o.dgrad_s2) {
            if (o.dc2_version != (int64_t)P->version) {
              rc = launch_conv_dgrad_s2_pack(ctx, g, W + P->p[d.w].offset, o.dc2_w);
              if (rc) return rc;
              o.dc2_version = (int64_t)P->version;
            }
            // single consumer of an activated conv output: its LeakyReLU / ReLU
            // adjoint is applied in the store (the producer then skips its mask pass)
            const int rin = root_of(pl, d.in0);
            const bool fuse = o.mask_prod >= 0 && !pl->gwritten[rin] && !s3_opt_has(S3O_NO_MASK_FUSE);
            const OpRec& po = pl->ops[fuse ? o.mask_prod : i];
            const ConvGeom& pg = po.cg;
            // dx is dPre of the few-channel conv below (mask fused, single
            // consumer).  Its weight gradient (conv_wgrad_c2_kernel), its data
            // gradient (conv_dgrad_c2_kernel, generator step only) and its bias
            // gradient (channel sums riding along here) all take bf16: store it
            // as bf16 ONLY — 0.89 instead of 1.78 GB written here and read there,
            // and no separate bias pass over it.
            const int nblk = conv_dgrad_s2_blocks(g);
            const bool sums = need_wgrad && po.d.b >= 0;
            const bool to16 = fuse && back_to_back(o.mask_prod, i) && dst == pl->t[rin].gptr &&
                              pl->precision == S3_PREC_BF16 && po.wgrad_c2 &&
                              po.cg.Cin == 2 && po.cg.Cout == 32 && po.d.res < 0 &&
                              (po.dgrad_c2 || !wants_grad(po.d.in0)) && conv_dgrad_s2_out16_ok(g) && pl->dpre16 &&
                              pl->dpre16_bytes >= (size_t)pl->t[rin].numel * 2 && pl->dpre16_for < 0 &&
                              (!sums || (pl->bsum && nblk <= 4096 && !s3_opt_has(S3O_NO_BIAS_FUSE))) &&
                              !s3_opt_has(S3O_NO_DPRE16);
            rc = launch_conv_dgrad_s2(ctx, g, dpre, o.dc2_w, to16 ? (float*)pl->dpre16 : dst,
                                      fuse ? tptr(pl, d.in0) : nullptr, pg.act == S3_ACT_LEAKY ? pg.alpha : 0.f,
                                      o.io.in_bf16, to16 ? 1 : 0, (to16 && sums) ? pl->bsum : nullptr,
                                      (to16 && fuse && o.io.in_bf16) ? po.sign_bytes : nullptr);
            if (!rc && fuse) pl->premasked[rin] = 1;
            if (!rc && to16) {
              pl->dpre16_for = rin; pl->dpre16_only = true;
              if (sums) { pl->bsum_for = rin; pl->bsum_nblk = nblk; }
            }
          } else if (o.dgrad_c2 && pl->precision == S3_PREC_BF16X3) {
            if (o.dc2_version != (int64_t)P->version) {
              rc = launch_conv_dgrad_c2_x3_pack(ctx, g, W + P->p[d.w].offset, o.dc2_w);
              if (rc) return rc;
              o.dc2_version = (int64_t)P->version;
            }
            rc = launch_conv_dgrad_c2_x3(ctx, g, dpre, o.dc2_w, dst);
          } else if (o.dgrad_c2) {
            if (o.dc2_version != (int64_t)P->version) {
              rc = launch_conv_dgrad_c2_pack(ctx, g, W + P->p[d.w].offset, o.dc2_w);
              if (rc) return rc;
              o.dc2_version = (int64_t)P->version;
            }
            rc = launch_conv_dgrad_c2(ctx, g, only16 ? (const float*)dpre16 : dpre, o.dc2_w, dst, only16 ? 1 : 0);
          } else if (o.gconv_dgrad) {
            if (o.gct_version != P->version) {
              rc = launch_gconv_pack(ctx, g, W + P->p[d.w].offset, o.gc_wt, 1, pl->precision == S3_PREC_BF16X3);
              if (rc) return rc;
              o.gct_version = P->version;
            }
            if (g.pad_mode == S3_PAD_REFLECT) {
              // dXpad over the reflect-padded frame, then fold the border back
              rc = launch_gconv_dgrad(ctx, g, dpre, o.gc_wt, pl->dxp, 0, 1, 0, pl->precision == S3_PREC_BF16X3);
              if (rc) return rc;
              GatherGeom fg;
              fg.kind = S3_OP_PAD; fg.N = g.N;
              for (int q = 0; q < 3; ++q) { fg.Di[q] = g.D[q]; fg.Do[q] = g.D[q] + 2 * g.lo[q]; fg.lo[q] = g.lo[q]; }
              fg.Ci = g.Cin; fg.Co = g.Cin; fg.pad_mode = g.pad_mode;
              fg.rep = 1; fg.d2s = 1; fg.c_off = 0;
              rc = launch_gather_bwd(ctx, fg, pl->dxp, dst);
            } else {
              const bool dy16 = o.use16 && dpre16 != nullptr;
              rc = launch_gconv_dgrad(ctx, g, dy16 ? (const float*)dpre16 : dpre, o.gc_wt, dst, 0, 0, dy16 ? 1 : 0,
                                      pl->precision == S3_PREC_BF16X3);
            }
          } else if (o.fewpos && o.fp_mfma) {
            // (reads the [tap][ci][co] filter along co: no transposed copy)
            const float* wf = W + P->p[d.w].offset;
            const bool reflect = g.pad_mode == S3_PAD_REFLECT;
            if (fp_both) {
              rc = launch_conv_fewpos_bwd_mfma(ctx, g, fp_gd, tptr(pl, d.in0), dpre, wf, reflect ? pl->dxp : dst,
                                               G + P->p[d.w].offset, d.b >= 0 ? G + P->p[d.b].offset : nullptr,
                                               accumulate_wgrad, fp_mask_y, fp_slope);
              if (rc) return rc;
            }
            if (reflect) {
              if (!fp_both)
                rc = launch_conv_fewpos_mfma(ctx, fp_gd, 1, dpre, wf, nullptr, nullptr, pl->dxp, fp_mask_y, fp_slope);
              if (rc) return rc;
              GatherGeom fg;
              fg.kind = S3_OP_PAD; fg.N = g.N;
              for (int q = 0; q < 3; ++q) { fg.Di[q] = g.D[q]; fg.Do[q] = g.D[q] + 2 * g.lo[q]; fg.lo[q] = g.lo[q]; }
              fg.Ci = g.Cin; fg.Co = g.Cin; fg.pad_mode = g.pad_mode;
              fg.rep = 1; fg.d2s = 1; fg.c_off = 0;
              // (with the producer's activation adjoint, or the first contribution
              // of a skip tensor, in the same store: no mask pass, no axpy)
              rc = fold_frame(fg, dst);
            } else if (!fp_both) {
              rc = launch_conv_fewpos_mfma(ctx, g, 1, dpre, wf, nullptr, nullptr, dst, fp_mask_y, fp_slope);
            }
          } else if (o.fewpos && o.fp_wt) {
            if (o.fp_version != P->version) {
              rc = launch_conv_fewpos_transpose(ctx, g, W + P->p[d.w].offset, o.fp_wt);
              if (rc) return rc;
              o.fp_version = P->version;
            }
            if (g.pad_mode == S3_PAD_REFLECT) {
              // dXpad over the padded frame (zero boundary), then fold the border back
              rc = launch_conv_fewpos_dgrad(ctx, conv_fewpos_frame_geom(g), dpre, o.fp_wt, pl->dxp, pl->fp_partial, pl->fp_partial_bytes);
              if (rc) return rc;
              GatherGeom fg;
              fg.kind = S3_OP_PAD; fg.N = g.N;
              for (int q = 0; q < 3; ++q) { fg.Di[q] = g.D[q]; fg.Do[q] = g.D[q] + 2 * g.lo[q]; fg.lo[q] = g.lo[q]; }
              fg.Ci = g.Cin; fg.Co = g.Cin; fg.pad_mode = g.pad_mode;
              fg.rep = 1; fg.d2s = 1; fg.c_off = 0;
              rc = launch_gather_bwd(ctx, fg, pl->dxp, dst);
            } else {
              rc = launch_conv_fewpos_dgrad(ctx, g, dpre, o.fp_wt, dst, pl->fp_partial, pl->fp_partial_bytes);
            }
          } else {
            rc = launch_conv_generic_dgrad(ctx, g, dpre, W + P->p[d.w].offset, dst);
          }
          if (rc) return rc;
          rc = grad_deliver(pl, d.in0, dst);
        }
      } break;
      case S3_OP_DENSE: {
        const TensorRec& it = pl->t[d.in0];
        const int rows = (int)(it.numel / it.dims[4]);
        const int cin = (int)it.dims[4], cout = (int)ot.dims[4];
        const float* dpre = dy;
        if (d.act != S3_ACT_NONE) {
          rc = launch_act_bwd(ctx, tptr(pl, d.out), dy, pl->dpre, ot.numel, d.act, d.alpha);
          if (rc) return rc;
          dpre = pl->dpre;
        }
        if (need_wgrad) {
          if (d.b >= 0) {
            rc = launch_bias_grad(ctx, dpre, rows, cout, G + P->p[d.b].offset, accumulate_wgrad);
            if (rc) return rc;
          }
          rc = launch_dense_wgrad(ctx, tptr(pl, d.in0), dpre, G + P->p[d.w].offset, rows, cin, cout, accumulate_wgrad);
          if (rc) return rc;
        }
        if (wants_grad(d.in0)) {
          float* dst = grad_dest(pl, d.in0);
          rc = launch_dense_dgrad(ctx, dpre, W + P->p[d.w].offset, dst, rows, cin, cout);
          if (rc) return rc;
          rc = grad_deliver(pl, d.in0, dst);
        }
      } break;
      case S3_OP_REPEAT_T: case S3_OP_D2S: case S3_OP_PAD: case S3_OP_CROP:
      case S3_OP_ROLL_T: case S3_OP_DILATE:
        if (wants_grad(d.in0)) {
          float* dst = grad_dest(pl, d.in0);
          rc = launch_gather_bwd(ctx, o.gg, dy, dst);
          if (rc) return rc;
          rc = grad_deliver(pl, d.in0, dst);
        }
        break;
      case S3_OP_CONCAT:
        if (wants_grad(d.in0)) {
          const TensorRec& a = pl->t[d.in0];
          float* dst = grad_dest(pl, d.in0);
          rc = s3_copy_channels(ctx, dy, (int)ot.dims[4], 0, dst, (int)a.dims[4], 0, (int)a.dims[4], a.numel / a.dims[4], 0);
          if (rc) return rc;
          rc = grad_deliver(pl, d.in0, dst);
        }
        break;
      case S3_OP_ADD:
        if (wants_grad(d.in0)) rc = grad_deliver(pl, d.in0, dy);
        if (!rc && !d.bcast_c && wants_grad(d.in1)) rc = grad_deliver(pl, d.in1, dy);
        break;
      case S3_OP_ACT:
        if (wants_grad(d.in0)) {
          float* dst = grad_dest(pl, d.in0);
          rc = launch_act_bwd(ctx, tptr(pl, d.out), dy, dst, ot.numel, d.act, d.alpha);
          if (rc) return rc;
          rc = grad_deliver(pl, d.in0, dst);
        }
        break;
      default: break;
    }
    if (rc) return rc;
    // bucketed all-reduce under the backward pass: the gradients of every
    // parameter at or above this op's are final in stream order
    if (need_wgrad && P->armed && (d.w >= 0 || d.b >= 0)) {
      int64_t lowest = P->reduce_end;
      if (d.w >= 0 && P->p[d.w].offset < lowest) lowest = P->p[d.w].offset;
      if (d.b >= 0 && P->p[d.b].offset < lowest) lowest = P->p[d.b].offset;
      if (P->reduce_end - lowest >= P->bucket_elems) {
        rc = s3_comm_reduce_range(ctx, G + lowest, P->reduce_end - lowest);
        if (rc) return rc;
        P->reduce_end = lowest;
        P->buckets_issued++;
      }
    }
  }
  if (need_wgrad && P->armed) {
    if (P->reduce_end > 0) {
      const int rc = s3_comm_reduce_range(ctx, G, P->reduce_end);
      if (rc) return rc;
      P->reduce_end = 0;
      P->buckets_issued++;
    }
    // consumed: a later backward pass on this store issues no collective
    // unless it is armed again
    P->armed = false;
    P->reduced = true;
  }
  if (d_input) {
    if (x_id < 0 || !pl->gwritten[x_id]) S3_FAIL(ctx, S3_ESTATE, "backward: no gradient reached the input");
    S3_HIP(ctx, hipMemcpyAsync(d_input, grad_of(pl, x_id), (size_t)pl->t[x_id].numel * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  }
  return S3_OK;
}
