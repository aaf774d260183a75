// Internal declarations shared by the translation units of libsup3r_hip.so.
// gfx950 (MI355X, CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sup3r_hip.h"


// ---- plan / context options -------------------------------------------------
// Kernel-selection switches (A/B comparisons in the tests, profiling
// ablations) are OPTIONS of a context and of the plans created from it — set
// through s3_ctx_set_option / s3_plan_create_opt (include/sup3r_hip.h), not
// read from the process environment while running.  The environment is read
// ONCE, by s3_ctx_create, as the initial defaults of that context (variable
// SUP3R_AMD_<NAME>; developer convenience for tools/ab.sh and the profilers).
// A plan snapshots its options when it is created; its entry points make them
// the calling thread's active set for the duration of the call.
#define S3_OPTION_LIST(X) \
  X(BF16_TRAIN_ACT) \
  X(DENSE_WGS) \
  X(DGRAD_S2_MIN_TILES) \
  X(DISC_BF16) \
  X(FEWCH_HALO_MIN_TILES) \
  X(FP32_ACT) \
  X(GCONV_MF2) \
  X(GCONV_MF4) \
  X(GRAPH) \
  X(HALO32_MIN_TILES) \
  X(HALO_S2_MIN_TILES) \
  X(MFMA_DBG) \
  X(MFMA_TILE) \
  X(NO_BATCHED_PACK) \
  X(NO_BIAS_FUSE) \
  X(NO_CHUNKED_DY16) \
  X(NO_DGRAD_C2) \
  X(NO_DGRAD_C2_SLIDE) \
  X(DGRAD_C2_SLIDE_MIN_UNITS) \
  X(NO_DGRAD_CHUNKED) \
  X(NO_DGRAD_FEWCH) \
  X(NO_DGRAD_S2) \
  X(NO_DGRAD_X3) \
  X(NO_DIRECT_OUTPUT) \
  X(NO_DISC_BF16) \
  X(NO_DPRE16) \
  X(NO_DPRE16_ONLY_MASK) \
  X(NO_FEWCH) \
  X(NO_FEWCH_HALO) \
  X(NO_FEWPOS) \
  X(NO_FEWPOS_MFMA) \
  X(NO_FEWPOS_BWD_FUSE) \
  X(NO_FEWPOS_SMALL) \
  X(NO_FEWPOS_TRUNK) \
  X(NO_FOLD16) \
  X(NO_FRAME16) \
  X(NO_FUSED2D) \
  X(NO_GCONV) \
  X(NO_GCONV_DY16) \
  X(NO_GCONV_SPLITK) \
  X(NO_HALO32) \
  X(NO_HALO_S2) \
  X(NO_MASK_FUSE) \
  X(NO_MFMA_GEN) \
  X(NO_DGRAD_GEN) \
  X(NO_CONV2D_WS) \
  X(NO_TRAIN2D_BF16) \
  X(NO_ADD16) \
  X(NO_WS_EXO) \
  X(NO_WS_RES2) \
  X(NO_WS_PP) \
  X(NO_WS_X3) \
  X(WS_X3_MIN_POS) \
  X(NO_CONV2D_OUT) \
  X(NO_CONV2D_HEAD) \
  X(KEEP_ACTIVATIONS) \
  X(NO_MFMA_BWD) \
  X(NO_PERSIST) \
  X(NO_PERSIST_DGRAD) \
  X(NO_PERSIST_STRIP) \
  X(PERSIST2) \
  X(NO_REPEAT_FUSE) \
  X(NO_WGRAD_X3) \
  X(NO_GCONV_X3) \
  X(NO_SIGN_BYTES) \
  X(NO_HALO_S2_K64) \
  X(NO_PLAIN_FOLD16) \
  X(NO_SEG_REDUCE) \
  X(NO_TAIL_BAND) \
  X(NO_TAIL_MFMA) \
  X(NO_TAIL_SLIDE) \
  X(NO_TAIL_SWEEP) \
  X(NO_TAIL_WINDOW) \
  X(NO_TAIL_X3) \
  X(NO_TILE66) \
  X(NO_TILE_NF2) \
  X(NO_WGRAD_BF16) \
  X(NO_WGRAD_C2) \
  X(NO_WGRAD_GEN_PF) \
  X(NO_WGRAD_TAIL) \
  X(NO_WGRAD_TAIL_SWEEP) \
  X(NO_WGRAD_WS) \
  X(PERSIST_DGRAD_MIN_TILES) \
  X(PERSIST_MIN_TILES) \
  X(POISON_ALLOC) \
  X(TAIL_SWEEP_SHAPE) \
  X(TRACE) \
  X(WGRAD_DBG) \
  X(WGRAD_SIDE_STREAM)
enum S3OptId {
#define X(n) S3O_##n,
  S3_OPTION_LIST(X)
#undef X
  S3O_COUNT
};
struct S3Options {
  int32_t v[S3O_COUNT];
  bool has[S3O_COUNT];
  S3Options() { for (int i = 0; i < S3O_COUNT; ++i) { v[i] = 0; has[i] = false; } }
};
extern thread_local const S3Options* s3_active_options;
const char* s3_option_name(int id);
int s3_option_id(const char* name);      // accepts "NO_PERSIST" or "SUP3R_AMD_NO_PERSIST"; -1 if unknown
inline bool s3_opt_has(int id) { return s3_active_options && s3_active_options->has[id]; }
inline long long s3_opt_int(int id, long long dflt) { return s3_opt_has(id) ? s3_active_options->v[id] : dflt; }
inline bool s3_opt_on(int id) { return s3_opt_int(id, 0) != 0; }
struct S3OptScope {
  const S3Options* prev;
  explicit S3OptScope(const S3Options* o) : prev(s3_active_options) { s3_active_options = o; }
  ~S3OptScope() { s3_active_options = prev; }
};

struct s3_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  void* comm = nullptr;  // ncclComm_t
  int rank = 0, nranks = 1;
  float* scratch = nullptr;  // small device scratch (reductions)
  size_t scratch_bytes = 0;
  int num_cu = 256;
  int64_t stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // S3_STAT_* launch counters
  S3Options opt;                     // defaults of the plans created from this context
  hipStream_t comm_stream = nullptr; // bucketed gradient all-reduce under the backward pass
  hipEvent_t comm_ev[2] = {nullptr, nullptr};   // [0] compute -> comm, [1] comm -> compute
  hipEvent_t wd_ev[2] = {nullptr, nullptr};     // s3_comm_wait: tail of the compute / comm stream
  int64_t comm_issued = 0;           // collectives enqueued since the last completed s3_comm_wait
  // option WGRAD_SIDE_STREAM (an experiment that lost, kept for the A/B): weight
  // gradients of launch-bound backward passes beside the data-gradient chain
  // (plan.cpp: wg_fork / wg_join; joins every s3_plan_backward)
  hipStream_t wg_stream = nullptr;
  hipEvent_t wg_ev[2] = {nullptr, nullptr};    // [0] compute -> side, [1] side -> compute
  bool wg_forked = false;
  // stream capture (s3_capture_begin .. s3_capture_end): launches go to a
  // non-blocking side stream while it records; `stream` is restored afterwards
  hipStream_t cap_stream = nullptr, saved_stream = nullptr;
  bool capturing = false;
  // recorded graphs hold the scratch pointer of their time: once one exists a
  // scratch block that is outgrown is retired (freed with the context), not freed
  bool graphs_made = false;
  std::vector<void*> retired;
  // a bias gradient's second stage (channel sums of per-workgroup partials)
  // waiting to ride along the next weight-gradient reduction launch
  // (launch_bias_grad_from_partial(.., defer) / s3_take_pending_bias)
  struct PendingBias { const float* partial = nullptr; int nblk = 0, c = 0; float* db = nullptr; int accumulate = 0; } pend_bias;
};

// One-time per-DEVICE setup of a launch function (hipFuncSetAttribute's dynamic-LDS
// limit is a property of the function ON A DEVICE): `static S3DeviceOnce o; if
// (!o.done(ctx->device)) { std::lock_guard<std::mutex> lk(o.m); ...; o.mark(ctx->device); }`
// — lock-free once set; two threads racing on the first call both run the (idempotent)
// setup, one after the other.
struct S3DeviceOnce {
  std::atomic<uint64_t> mask{0};
  std::mutex m;
  bool done(int dev) const { return (mask.load(std::memory_order_acquire) >> (dev & 63)) & 1u; }
  void mark(int dev) { mask.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

#define S3_HIP(ctx, call)                                                    \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);        \
      return S3_EHIP;                                                        \
    }                                                                        \
  } while (0)

#define S3_FAIL(ctx, code, msg) \
  do {                          \
    (ctx)->err = (msg);         \
    return (code);              \
  } while (0)

// geometry of one fused convolution, passed by value to kernels
struct ConvGeom {
  int N;
  int D[3];   // input spatial dims (s1, s2, t)
  int O[3];   // conv output dims BEFORE the depth-to-space store
  int Cin, Cout;
  int k[3], s[3], lo[3];
  int pad_mode;
  int act;
  float alpha;
  int d2s;    // block size (1 = none)
  // halo-tile kernel only: read a 64-channel slice of a wider fp32 tensor — cells
  // are in_cstride floats apart and only the first in_cvalid channels of the
  // slice exist (0 = the tensor has exactly C_in channels).  Used by the
  // chunked data gradient of convs with C_out > 64.
  int in_cstride = 0, in_cvalid = 0;
  // weights-stationary 2-D kernel, inference plans: the conv behind a
  // Sup3rConcat of a 64-channel tensor and ONE exogenous channel (topography),
  // 65 -> C_out, runs as the 64-channel conv over the bf16 tensor plus the
  // exogenous channel's nine taps per output added from the fp32 field in the
  // same launch — the 65-channel tensor is never written.  w_cin = the C_in
  // axis of the canonical weights (65; 0 = C_in), exo = the field (N, s1, s2, 1)
  // (set at launch time)
  int w_cin = 0;
  const float* exo = nullptr;
  // ... and a SECOND skip operand (bf16 cells, the conv's output shape): the
  // SkipConnection add right behind a conv that already carries one
  // (sup3rcc/gen_*_5x_1x_*: the last residual block's sum + the big skip) is
  // absorbed into the conv's store instead of running as a pass of its own
  // (set at launch time)
  const void* res2 = nullptr;
  // set by the plan once the tensor dtypes are known: this conv's forward IS the
  // weights-stationary kernel — the logical-axes tile image is not packed (a
  // training step re-packs every filter: one 5 us launch less per conv and step)
  int ws_only = 0;
  // persistent trunk kernel, inference plans: the input is the temporal repeat
  // (SpatioTemporalExpansion temporal_mult, out[.., j, :] = in[.., j / rep, :]) of
  // a tensor with D[2] / in_rep time steps, read through the halo index instead
  // of being materialised; 0 / 1 = plain input
  int in_rep = 0;
  // ... and the same for the residual operand (d2s == 1): res[.., j, :] is
  // cell j / res_rep of a tensor with O[2] / res_rep steps.  A conv has ONE
  // repeat factor (in_rep == res_rep when both are set), 2, 3 or 4: the
  // kernel variants carry it as a compile-time constant
  int res_rep = 0;
  // halo-tile kernel, GEN instantiations (kernels_conv_mfma_gen.hip): N, D, O,
  // k, lo above are LOGICAL axes (a0, a1, a2) — a permutation of the tensor's
  // (n, s1, s2, t) chosen so that a2 is a long axis (the 16-position run) and
  // a k = 1 axis / the batch is a0.  xs / xn: input cell strides of the three
  // axes / of the logical batch; ys / yn: the same for the FINAL output tensor
  // (depth-to-space applied: a blocked axis' stride carries the factor b);
  // yb: cell offsets of a unit step inside the b x b block (block row, column)
  int gen = 0;
  int64_t xn = 0, yn = 0;
  int64_t xs[3] = {0, 0, 0}, ys[3] = {0, 0, 0}, yb[2] = {0, 0};
};

// generic gather op (pad / crop / repeat / roll / d2s / concat): out <- in
struct GatherGeom {
  int kind;
  int N;
  int Di[3], Ci;  // input dims
  int Do[3], Co;  // output dims
  int lo[3];
  int pad_mode;
  int rep;
  int d2s;
  int c_off;      // concat: channel offset of in within out
};

__host__ __device__ inline int s3_reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i > n - 1) i = 2 * (n - 1) - i;
  return i;
}

// XCD-aware tile order for one-tile-per-workgroup kernels: the dispatcher puts
// block b on XCD b % 8 and every XCD has a private L2, so hand each XCD a
// contiguous run of tiles (bijective on [0, nblk)): neighbouring tiles share
// halo cells, which then hit in that XCD's L2 instead of going to memory twice.
__device__ inline int s3_xcd_tile(int b, int nblk) {
  const int q = nblk / 8, r = nblk % 8, xcd = b % 8, k = b / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// The same for PERSISTENT workgroups that walk a list of n items (tiles,
// k-steps): the XCD of this block owns the contiguous share [lo, hi) — sized by
// its number of blocks — and block k of its nk blocks takes lo + k, lo + k +
// nk, ...  Neighbouring items (halo overlap, the 3 x 3 window rows of a
// weight gradient) are then fetched into ONE L2 instead of up to eight.
__device__ inline void s3_xcd_share(int64_t n, int64_t& lo, int64_t& hi, int& k, int& nk) {
  const int G = gridDim.x, b = blockIdx.x, xcd = b % 8;
  int before = 0;
  for (int q = 0; q < xcd; ++q) before += (G - q + 7) / 8;
  nk = (G - xcd + 7) / 8;
  k = b / 8;
  lo = n * before / G;
  hi = n * (before + nk) / G;
}

// ---- launchers (defined in the .hip files) ------------------------------
// element types of a fused conv's activations (0 = fp32, 1 = bf16)
struct ConvIO {
  int in_bf16 = 0, out_bf16 = 0, res_bf16 = 0;
};
int launch_conv_generic_fwd(s3_ctx* ctx, const ConvGeom& g, const void* x,
                            const float* w, const float* bias,
                            const float* res, void* y, int out_bf16,
                            int in_bf16);
bool conv_small_supported(const ConvGeom& g, int in_bf16);
// hi-res tail conv C_in = 8 (bf16 cells) -> C_out <= 16 (fp32) on MFMA
bool conv_tail_mfma_supported(const ConvGeom& g);
// split-bf16 form for BF16X3 plans (fp32 in / out, C_out == 2)
bool conv_tail_x3_supported(const ConvGeom& g, int precision);
int launch_conv_tail_x3(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* w,
                        const float* bias, float* y);
// aff (device, scale[C_out] then shift[C_out], or null): y * scale + shift on the way out
bool conv_tail_sweep_supported(const ConvGeom& g);
int launch_conv_tail_sweep(s3_ctx* ctx, const ConvGeom& g, const void* x, const float* w,
                           const float* bias, float* y, const float* aff);
int launch_conv_tail_mfma(s3_ctx* ctx, const ConvGeom& g, const void* x,
                          const float* w, const float* bias, float* y, const float* aff = nullptr);
int launch_conv_generic_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy,
                              const float* w, float* dx);
int launch_conv_generic_wgrad(s3_ctx* ctx, const ConvGeom& g, const float* x,
                              const float* dy, float* dw, float* partial,
                              size_t partial_bytes, int accumulate);
size_t conv_generic_wgrad_partial_bytes(const ConvGeom& g);

// MFMA halo-tile conv (3x3x3 / 3x3x1, stride 1, C_in == 64)
bool conv_mfma_supported(const ConvGeom& g, int precision);
size_t conv_mfma_packed_bytes(const ConvGeom& g, int precision);
int launch_conv_mfma_pack(s3_ctx* ctx, const ConvGeom& g, int precision,
                          const float* w, void* packed);
int launch_conv_mfma_fwd(s3_ctx* ctx, const ConvGeom& g, int precision,
                         const void* x, const void* packed, const float* bias,
                         const void* res, void* y, ConvIO io);
bool conv_mfma_bf16_out_ok(const ConvGeom& g);
// logical-axes instantiations (kernels_conv_mfma_gen.hip): 2-D nets, few time
// steps, any C_in <= 256 / C_out; bf16 and BF16X3 plans, forward only
bool conv_mfma_gen_supported(const ConvGeom& g, int precision);
bool conv_mfma_is_gen(const ConvGeom& g, int precision);   // supported AND not a trunk geometry
// ... and the data gradient of such a conv on the same kernel (padded frame + fold)
ConvGeom conv_dgrad_gen_geom(const ConvGeom& g);
bool conv_dgrad_gen_supported(const ConvGeom& g, int precision);
size_t conv_mfma_gen_packed_bytes(const ConvGeom& g, int precision);
int launch_conv_mfma_gen_pack(s3_ctx* ctx, const ConvGeom& g, int precision, const float* w, void* packed);
int launch_conv_mfma_gen_fwd(s3_ctx* ctx, const ConvGeom& g, int precision, const void* x, const void* packed,
                             const float* bias, const void* res, void* y, ConvIO io);
// weights-stationary persistent 2-D conv for the all-bf16 64 -> 64 k trunks of
// the spatial generators (kernels_conv2d_ws.hip); physical 2-D geometry
bool conv2d_ws_geom_ok(const ConvGeom& g);
bool conv2d_ws_frame_geom_ok(const ConvGeom& g);  // its data gradient over the zero-padded frame (bf16 in / out)
bool conv2d_ws_tail_geom_ok(const ConvGeom& g);   // 64 -> C_out <= 16 output conv, fp32 out
bool conv2d_ws_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res);
size_t conv2d_ws_image_bytes(const ConvGeom& g);
int launch_conv2d_ws_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image);
// the few-feature 2-D output conv with the taps as matrix columns (kernels_conv2d_out.hip):
// 64 -> C_out <= 7, bf16 cells (bf16 plans) or fp32 cells (BF16X3 plans) in, fp32 out
bool conv2d_out_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res);
size_t conv2d_out_image_bytes(const ConvGeom& g);
int launch_conv2d_out_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image);
int launch_conv2d_out(s3_ctx* ctx, const ConvGeom& g, int precision, const void* x, const void* image,
                      const float* bias, void* y);
// ... and in the BF16X3 mode (kernels_conv2d_ws_x3.hip): fp32 cells in / out, the contraction split in
// two K passes of 32 channels whose [hi | lo] operands have the bf16 kernel's LDS shapes
bool conv2d_ws_x3_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res);
size_t conv2d_ws_x3_image_bytes(const ConvGeom& g);
int launch_conv2d_ws_x3_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image);
int launch_conv2d_ws_x3(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image, const float* bias,
                        const void* res, void* y);
// ... and the few-feature head conv of those generators (C_in 1 / 2 -> 64, fp32 field in, bf16 out)
bool conv2d_head_geom_ok(const ConvGeom& g);
bool conv2d_head_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res);
size_t conv2d_head_image_bytes(const ConvGeom& g);
// (exact: BF16X3 plans — unrounded fp32 operands, fp32 cells out)
int launch_conv2d_head_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image, int exact);
int launch_conv2d_head(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image, const float* bias, void* y,
                       int exact);
int launch_conv2d_ws(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image, const float* bias,
                     const void* res, void* y);
// persistent wave-specialised variant for the all-bf16 64 -> 64 trunk; its
// filter image (LDS layout) is appended to the packed buffer of the conv
bool conv_mfma_persist_geom_ok(const ConvGeom& g);
bool conv_mfma_persist_rep_ok(int rep);
bool conv_mfma_persist_supported(const s3_ctx* ctx, const ConvGeom& g, ConvIO io,
                                 bool has_res);
size_t conv_mfma_persist_image_bytes(const ConvGeom& g);
// one job of the batched filter re-pack (pack_jobs_kernel): a 64-input-channel
// k3 conv's bf16 images from its fp32 filter
struct S3PackJob {
  const float* w;              // forward filter [27][cin_f][cout_f] in the parameter store
  unsigned short* tile;        // halo-tile image [n_ct][27][64][64]
  unsigned short* persist;     // persistent-kernel image, or nullptr
  int cout;                    // output channels of the PACKED conv (forward: cout_f; dgrad: cin_f)
  int n_ct;                    // ceil(cout / 64)
  int dgrad;                   // 1: packed conv = data gradient (cin' = cout_f = 64), flip + transpose
};
int launch_pack_jobs(s3_ctx* ctx, const S3PackJob* jobs_dev, int n_jobs, int max_ct);
// the trunk's data gradient on the persistent kernel (bf16 dPre in, fp32 frame out)
bool conv_mfma_persist_dgrad_geom_ok(const ConvGeom& g);
bool conv_mfma_persist_dgrad_supported(const s3_ctx* ctx, const ConvGeom& g);
// frame16: the padded frame dxp is written as bf16 (no accumulate then)
int launch_conv_mfma_persist_dgrad(s3_ctx* ctx, const ConvGeom& g, const void* dpre16, const void* image,
                                   float* dxp, int accumulate = 0, int frame16 = 0);
int launch_conv_mfma_persist_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image);
// split-bf16 (BF16X3) LDS-halo data gradients of the hi-res discriminator layers
bool conv_dgrad_c2_x3_supported(const ConvGeom& g, int precision);
size_t conv_dgrad_c2_x3_packed_bytes();
int launch_conv_dgrad_c2_x3_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img);
int launch_conv_dgrad_c2_x3(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img, float* dx);
bool conv_dgrad_s2_x3_supported(const s3_ctx* ctx, const ConvGeom& g, int precision);
size_t conv_dgrad_s2_x3_packed_bytes(const ConvGeom& g);
int launch_conv_dgrad_s2_x3_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img);
int launch_conv_dgrad_s2_x3(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img, float* dx,
                            const float* mask_y, float mask_slope);
// round-4 experiment: 128-position consumer waves, filter fragments from L1 / L2
bool conv_mfma_persist2_supported(const s3_ctx* ctx, const ConvGeom& g, ConvIO io, bool has_res);
int launch_conv_mfma_persist2(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image,
                              const float* bias, const void* res, void* y);
int launch_conv_mfma_persist(s3_ctx* ctx, const ConvGeom& g, const void* x,
                             const void* image, const float* bias,
                             const void* res, void* y);

// general fp32-MFMA weight gradient (C_in <= 64, stride 1 / 2, any padding)
bool conv_wgrad_gen_supported(const ConvGeom& g);
size_t conv_wgrad_gen_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_gen(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                          float* dw, float* partial, size_t partial_bytes, int accumulate);

// general gather-MFMA conv (any stride / padding, C_in % 32 == 0, fp32 I/O,
// bf16 operands): forward and data gradient
bool conv_gconv_supported(const ConvGeom& g, int precision);
bool conv_gconv_dgrad_supported(const ConvGeom& g, int precision);
size_t conv_gconv_packed_bytes(const ConvGeom& g, int dgrad, int x3 = 0);
int launch_gconv_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* packed, int dgrad, int x3 = 0);
int launch_gconv_fwd(s3_ctx* ctx, const ConvGeom& g, const float* x, const void* packed,
                     const float* bias, const float* res, void* y, int out_bf16, int in_bf16 = 0, int x3 = 0,
                     void* sign_bytes = nullptr);
bool conv_gconv_writes_sign_bytes(const s3_ctx* ctx, const ConvGeom& g, int out_bf16);
int launch_gconv_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* packed_t,
                       float* dx, int accumulate, int frame, int dy_bf16 = 0, int x3 = 0);

// MFMA backward of the 3x3x3 stride-1 trunk convs.
bool conv_wgrad_bf16_gen_supported(const ConvGeom& g, int precision);
size_t conv_wgrad_bf16_gen_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_bf16_gen(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                               float* dw, float* partial, size_t partial_bytes, int accumulate,
                               int x_bf16 = 0, int x3 = 0);
// stride-2 valid conv with C_in = 32 on an LDS halo (kernels_conv_halo_s2.hip)
bool conv_halo_s2_supported(const s3_ctx* ctx, const ConvGeom& g, int precision);
size_t conv_halo_s2_packed_bytes(const ConvGeom& g);
int launch_conv_halo_s2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img);
int launch_conv_halo_s2_fwd(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* img,
                            const float* bias, void* y, int out_bf16);
// forward conv with C_in = 32 on an LDS halo (kernels_conv_halo32.hip)
bool conv_halo32_supported(const s3_ctx* ctx, const ConvGeom& g, int precision);
size_t conv_halo32_packed_bytes(const ConvGeom& g);
int launch_conv_halo32_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img);
int launch_conv_halo32_fwd(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* img,
                           const float* bias, void* y, int in_bf16 = 0, int out_bf16 = 0);
// dgrad of a stride-2 valid conv with 32 output channels, per residue class on an
// LDS halo (kernels_conv_dgrad_s2.hip)
bool conv_dgrad_s2_supported(const s3_ctx* ctx, const ConvGeom& g, int precision);
size_t conv_dgrad_s2_packed_bytes(const ConvGeom& g);
int launch_conv_dgrad_s2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img);
int launch_conv_dgrad_s2(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img, float* dx,
                         const void* mask_y, float mask_slope, int mask_bf16 = 0, int out_bf16 = 0,
                         float* bsum = nullptr, const void* mask_bits = nullptr);
bool conv_dgrad_s2_out16_ok(const ConvGeom& g);
int conv_dgrad_s2_blocks(const ConvGeom& g);
// dgrad of the few-channel hi-res conv with an LDS halo (kernels_conv_dgrad_fewch.hip)
bool conv_dgrad_c2_supported(const ConvGeom& g, int precision);
size_t conv_dgrad_c2_packed_bytes();
int launch_conv_dgrad_c2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img);
int launch_conv_dgrad_c2(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img,
                         float* dx, int dy_bf16 = 0);
bool conv_wgrad_bf16_2d_supported(const ConvGeom& g, int precision);
size_t conv_wgrad_bf16_2d_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_bf16_2d(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                              float* dw, float* partial, size_t partial_bytes, int accumulate, int x_bf16 = 0,
                              int dy_bf16 = 0);
// wgrad of the 2-channel hi-res conv, LDS-free bf16 MFMA (kernels_conv_wgrad_fewch.hip)
bool conv_wgrad_c2_supported(const ConvGeom& g, int precision);
size_t conv_wgrad_c2_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_c2(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                         float* dw, float* partial, size_t partial_bytes, int accumulate, int dy_bf16 = 0, int x3 = 0);
// LDS-halo wgrad of the hi-res tail conv (C_in = 8), kernels_conv_wgrad_fewch.hip
bool conv_wgrad_tail_supported(const ConvGeom& g, int precision);
size_t conv_wgrad_tail_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_tail(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                           float* dw, float* partial, size_t partial_bytes, int accumulate,
                           int x_bf16);
// trunk wgrad on bf16 MFMA with LDS transpose reads (kernels_conv_wgrad_bf16.hip)
bool conv_wgrad_bf16_supported(const ConvGeom& g, int precision);
size_t conv_wgrad_bf16_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_bf16(s3_ctx* ctx, const ConvGeom& g, const float* x,
                           const float* dy, float* dw, float* partial,
                           size_t partial_bytes, int accumulate, int x_bf16, int dy_bf16 = 0, int x3 = 0);
// wgrad: persistent-workgroup kernel (kernels_conv_wgrad_mfma.hip)
bool conv_wgrad_mfma_supported(const ConvGeom& g);
size_t conv_wgrad_mfma_partial_bytes(const s3_ctx* ctx, const ConvGeom& g);
int launch_conv_wgrad_mfma(s3_ctx* ctx, const ConvGeom& g, const float* x,
                           const float* dy, float* dw, float* partial,
                           size_t partial_bytes, int accumulate);
// dgrad: the forward halo kernel run on dPre with the flipped / transposed
// filter over the (D+2)^3 padded frame (zero boundary), followed by the
// adjoint of the virtual padding (fold) — kernels_conv_mfma.hip
bool conv_dgrad_mfma_supported(const ConvGeom& g, int precision);
ConvGeom conv_dgrad_valid_geom(const ConvGeom& g);
ConvGeom conv_dgrad_chunk_geom(const ConvGeom& g, int k);
bool conv_dgrad_chunked_supported(const ConvGeom& g, int precision);
int launch_conv_dgrad_chunk_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, float* wt, int k);
bool conv_dgrad_mfma_valid_supported(const ConvGeom& g, int precision);
ConvGeom conv_dgrad_geom(const ConvGeom& g);
int launch_conv_dgrad_pack(s3_ctx* ctx, const ConvGeom& g, const float* w,
                           float* wt);

// few-positions / many-channels convs as weight-streaming GEMMs
// (kernels_conv_fewpos.hip)
bool conv_fewpos_supported(const ConvGeom& g);
size_t conv_fewpos_partial_bytes(const ConvGeom& g);
int launch_conv_fewpos_fwd(s3_ctx* ctx, const ConvGeom& g, const float* x,
                           const float* w, const float* bias, const float* res,
                           float* y, float* partial, size_t partial_bytes);
int launch_conv_fewpos_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy,
                             const float* wt, float* dx, float* partial,
                             size_t partial_bytes);
size_t conv_fewpos_wgrad_partial_bytes(const ConvGeom& g);
bool conv_fewpos_wgrad_ok(const ConvGeom& g);
ConvGeom conv_fewpos_frame_geom(const ConvGeom& g);
int launch_conv_fewpos_wgrad(s3_ctx* ctx, const ConvGeom& g, const float* x,
                             const float* dy, float* dw, float* partial, size_t partial_bytes,
                             int accumulate);
int launch_conv_fewpos_transpose(s3_ctx* ctx, const ConvGeom& g, const float* w,
                                 float* wt);
// ... the same three on the fp32 matrix instruction, one launch each, no
// partial buffers and no filter transpose (kernels_conv_fewpos_mfma.hip)
bool conv_fewpos_mfma_ok(const ConvGeom& g);
bool conv_fewpos_wgrad_mfma_ok(const ConvGeom& g);
bool conv_fewpos_mfma_small_ok(const s3_ctx* ctx, const ConvGeom& g);
bool conv_fewpos_bwd_mfma_ok(const s3_ctx* ctx, const ConvGeom& g, const ConvGeom& gd);
int launch_conv_fewpos_bwd_mfma(s3_ctx* ctx, const ConvGeom& g, const ConvGeom& gd, const float* x,
                                const float* dy, const float* w, float* dx, float* dw, float* db,
                                int accumulate, const float* mask_y, float slope);
int launch_conv_fewpos_mfma(s3_ctx* ctx, const ConvGeom& g, int mode, const float* src,
                            const float* w, const float* bias, const float* res, float* y,
                            const float* mask_y = nullptr, float slope = 0.f);
int launch_conv_fewpos_wgrad_mfma(s3_ctx* ctx, const ConvGeom& g, const float* x, const float* dy,
                                  float* dw, float* db, int accumulate,
                                  const float* mask_y = nullptr, float slope = 0.f);

int launch_gather(s3_ctx* ctx, const GatherGeom& g, const void* in, void* out,
                  int esize);
// (frame16: dout is a bf16 frame — float4-fold geometries only)
int launch_gather_bwd(s3_ctx* ctx, const GatherGeom& g, const float* dout,
                      float* din, void* side16 = nullptr, int frame16 = 0);
bool gather_bwd_mask_ok(const GatherGeom& g);
int launch_gather_bwd_masked(s3_ctx* ctx, const GatherGeom& g, const float* dout, float* din,
                             const void* mask_y, int y_bf16, float slope, float* bsum = nullptr,
                             int out_bf16 = 0, int frame16 = 0);
int launch_act(s3_ctx* ctx, const float* x, float* y, int64_t n, int act,
               float alpha);
// dx = dy * act'(y)  (y is the activation OUTPUT; sign-preserving acts only)
int launch_act_bwd(s3_ctx* ctx, const float* y, const float* dy, float* dx,
                   int64_t n, int act, float alpha);
// dpre[pos][c] = dy[d2s-permuted] * act'(y[d2s-permuted]) (conv epilogue adj.)
int launch_conv_epilogue_bwd(s3_ctx* ctx, const ConvGeom& g, const float* y,
                             const float* dy, float* dpre, int y_bf16, void* d16 = nullptr,
                             float* bsum = nullptr);
bool conv_epilogue_bwd_d16_ok(const ConvGeom& g);
bool conv_epilogue_bwd_bsum_ok(const ConvGeom& g);
int conv_epilogue_bwd_blocks(const s3_ctx* ctx, const ConvGeom& g, bool with_bsum);
int launch_add16(s3_ctx* ctx, const void* a, const void* b, void* y, int64_t n);
int launch_add(s3_ctx* ctx, const float* a, const float* b, float* y, int64_t n,
               int c, int bcast_c);
int launch_axpy(s3_ctx* ctx, const float* x, float* y, int64_t n);  // y += x
int launch_gather_bwd_add(s3_ctx* ctx, const GatherGeom& g, const float* dout, float* din, const float* add,
                          float* bsum = nullptr, void* side16 = nullptr, int frame16 = 0);
bool gather_bwd_bsum_ok(const GatherGeom& g);
int gather_bwd_bsum_blocks(const s3_ctx* ctx, const GatherGeom& g);
int launch_bias_grad_from_partial(s3_ctx* ctx, const float* partial, int nblk, int c, float* db, int accumulate,
                                  bool defer = false);
// the deferred job, if any, launched on its own (nothing took it along)
int s3_flush_pending_bias(s3_ctx* ctx);
// workgroup `ch` of a 256-thread launch: db[ch] (+)= sum over the nblk rows of partial[.][c]
__device__ inline void s3_bias_stage2_body(const float* __restrict__ partial, int nblk, int c, int ch,
                                           float* __restrict__ db, int accumulate, float* sm /* [256] */) {
  float t = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 256) t += partial[(int64_t)b * c + ch];
  sm[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[ch] = accumulate ? db[ch] + sm[0] : sm[0];
}
int launch_bias_grad(s3_ctx* ctx, const float* dy, int64_t n_pos, int c,
                     float* db, int accumulate);
int launch_dense_fwd(s3_ctx* ctx, const float* x, const float* w,
                     const float* bias, float* y, int n, int cin, int cout,
                     int act, float alpha);
int launch_dense_dgrad(s3_ctx* ctx, const float* dy, const float* w, float* dx,
                       int n, int cin, int cout);
int launch_dense_wgrad(s3_ctx* ctx, const float* x, const float* dy, float* dw,
                       int n, int cin, int cout, int accumulate);
int launch_adam(s3_ctx* ctx, float* w, const float* g, float* m, float* v,
                int64_t n, float alpha, float one_minus_b1, float one_minus_b2, float eps,
                const float* h_dev = nullptr);
int launch_optimizer(s3_ctx* ctx, int kind, float* w, const float* g, float* m, float* v,
                     int64_t n, const float* h, const float* h_dev = nullptr);
int launch_stage_hyper(s3_ctx* ctx, float* dst, const float* h);
int launch_fill(s3_ctx* ctx, float* p, int64_t n, float v);
// (cross-file helpers of the library, not part of its ABI: hidden, so that the
// .so exports exactly what include/sup3r_hip.h declares — tests/test_abi.py)
#define S3_INTERNAL __attribute__((visibility("hidden")))
// comm.cpp: SUM over the ranks of buf[0 .. n) on the context's comm stream,
// after everything enqueued so far on its compute stream
extern "C" S3_INTERNAL int s3_comm_reduce_range(s3_ctx* ctx, float* buf, int64_t n);
extern "C" S3_INTERNAL int s3_params_take_armed(s3_params* p, int* n_buckets);
int launch_mean_abs(s3_ctx* ctx, const float* p, int64_t n, float* out_dev);
int ensure_scratch(s3_ctx* ctx, size_t bytes);

// whole-network kernel for small 2-D conv stacks (kernels_fused2d.hip)
struct Fused2dLayer {
  ConvGeom g;
  int in_t, out_t, res_t;   // tensor ids (res_t < 0: none)
  int64_t w_off, b_off;     // float offsets into the weight buffer (b_off < 0: no bias)
};
struct Fused2dPlan;
Fused2dPlan* fused2d_build(s3_ctx* ctx, const std::vector<Fused2dLayer>& layers, int n_tensors,
                           int in_tensor, int out_tensor);
int fused2d_run(s3_ctx* ctx, Fused2dPlan* p, const float* W, uint64_t wversion, const float* x,
                float* y);
void fused2d_free(Fused2dPlan* p);
