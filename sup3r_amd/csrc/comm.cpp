// Data-parallel gradient sync over RCCL (xGMI).  Replaces the host-side python
// sum of per-GPU gradient lists, AbstractSingleModel._sum_parallel_grad
// (sup3r/models/abstract.py:785-805): ONE ncclAllReduce(SUM) over the flat
// fp32 gradient buffer of the network being trained (one process per GPU).
//
// librccl.so is resolved lazily with dlopen so that libsup3r_hip.so loads (and
// exports every symbol) on a box without RCCL / without a GPU.
#include <dlfcn.h>
#include <cstring>

#include "common.h"

struct s3_params_view {  // layout prefix of s3_params (plan.cpp)
  s3_ctx* ctx;
};

namespace {
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef int (*fn_get_uid)(ncclUniqueId_t*);
typedef int (*fn_init_rank)(void**, int, ncclUniqueId_t, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_bcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_destroy)(void*);

struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_bcast bcast = nullptr;
  fn_errstr errstr = nullptr;
  fn_destroy destroy = nullptr;
  bool load(std::string& err) {
    if (h) return true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) { err = std::string("dlopen librccl.so failed: ") + dlerror(); return false; }
    get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
    bcast = (fn_bcast)dlsym(h, "ncclBroadcast");
    errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    if (!get_uid || !init_rank || !allreduce) { err = "librccl.so lacks nccl symbols"; return false; }
    return true;
  }
};
Rccl g_rccl;
constexpr int kNcclFloat32 = 7;  // ncclFloat32
constexpr int kNcclSum = 0;      // ncclSum
}  // namespace

extern "C" int s3_comm_unique_id(void* out128) {
  std::string err;
  if (!out128 || !g_rccl.load(err)) return S3_ERCCL;
  ncclUniqueId_t id;
  if (g_rccl.get_uid(&id) != 0) return S3_ERCCL;
  memcpy(out128, &id, sizeof(id));
  return S3_OK;
}

extern "C" int s3_comm_init(s3_ctx* ctx, int rank, int nranks, const void* unique_id128) {
  if (!ctx || !unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) return S3_EINVAL;
  if (!g_rccl.load(ctx->err)) return S3_ERCCL;
  S3_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId_t id;
  memcpy(&id, unique_id128, sizeof(id));
  int rc = g_rccl.init_rank(&ctx->comm, nranks, id, rank);
  if (rc != 0) {
    ctx->err = std::string("ncclCommInitRank: ") + (g_rccl.errstr ? g_rccl.errstr(rc) : "error");
    return S3_ERCCL;
  }
  ctx->rank = rank;
  ctx->nranks = nranks;
  return S3_OK;
}

extern "C" int s3_allreduce_sum(s3_ctx* ctx, float* buf, int64_t n) {
  if (!ctx || !buf || n < 0) return S3_EINVAL;
  if (ctx->nranks <= 1 && !ctx->comm) return S3_OK;  // single rank, no comm
  if (!ctx->comm) S3_FAIL(ctx, S3_ESTATE, "allreduce before s3_comm_init");
  int rc = g_rccl.allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, ctx->comm, ctx->stream);
  if (rc != 0) {
    ctx->err = std::string("ncclAllReduce: ") + (g_rccl.errstr ? g_rccl.errstr(rc) : "error");
    return S3_ERCCL;
  }
  return S3_OK;
}

// ---- bucketed all-reduce overlapped with the backward pass ----------------
// s3_plan_backward hands over the finished tail of the gradient buffer bucket
// by bucket (plan.cpp); each bucket is reduced on a second stream behind an
// event on the compute stream, so RCCL's xGMI traffic runs under the
// remaining weight / data gradient kernels.  Collectives are issued in the
// same order on every rank (the op order of the plan).
static int comm_stream_ready(s3_ctx* ctx) {
  if (ctx->comm_stream) return S3_OK;
  S3_HIP(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i)
    S3_HIP(ctx, hipEventCreateWithFlags(&ctx->comm_ev[i], hipEventDisableTiming));
  return S3_OK;
}

extern "C" int s3_comm_reduce_range(s3_ctx* ctx, float* buf, int64_t n) {
  if (!ctx || !buf || n <= 0) return S3_EINVAL;
  if (!ctx->comm) return S3_OK;               // single rank: nothing to sum
  int rc = comm_stream_ready(ctx);
  if (rc) return rc;
  S3_HIP(ctx, hipEventRecord(ctx->comm_ev[0], ctx->stream));
  S3_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->comm_ev[0], 0));
  const int nrc = g_rccl.allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, ctx->comm,
                                   ctx->comm_stream);
  if (nrc != 0) {
    ctx->err = std::string("ncclAllReduce (bucket): ") + (g_rccl.errstr ? g_rccl.errstr(nrc) : "error");
    return S3_ERCCL;
  }
  ctx->stat[S3_STAT_BUCKET_ELEMS] += n;
  return S3_OK;
}

// the compute stream continues only after every bucket has been reduced
static int comm_join(s3_ctx* ctx) {
  if (!ctx->comm || !ctx->comm_stream) return S3_OK;
  S3_HIP(ctx, hipEventRecord(ctx->comm_ev[1], ctx->comm_stream));
  S3_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm_ev[1], 0));
  return S3_OK;
}

extern "C" int s3_broadcast(s3_ctx* ctx, float* buf, int64_t n, int root) {
  if (!ctx || !buf || n < 0) return S3_EINVAL;
  if (ctx->nranks <= 1 && !ctx->comm) return S3_OK;
  if (!ctx->comm) S3_FAIL(ctx, S3_ESTATE, "broadcast before s3_comm_init");
  if (!g_rccl.bcast) S3_FAIL(ctx, S3_ERCCL, "librccl.so lacks ncclBroadcast");
  if (root < 0 || root >= ctx->nranks) return S3_EINVAL;
  int rc = g_rccl.bcast(buf, buf, (size_t)n, kNcclFloat32, root, ctx->comm, ctx->stream);
  if (rc != 0) {
    ctx->err = std::string("ncclBroadcast: ") + (g_rccl.errstr ? g_rccl.errstr(rc) : "error");
    return S3_ERCCL;
  }
  return S3_OK;
}

extern "C" void s3_comm_destroy(s3_ctx* ctx) {
  if (!ctx || !ctx->comm) return;
  (void)hipStreamSynchronize(ctx->stream);
  if (g_rccl.destroy) g_rccl.destroy(ctx->comm);
  ctx->comm = nullptr;
  ctx->rank = 0;
  ctx->nranks = 1;
}

extern "C" int s3_params_broadcast(s3_params* p, int which, int root) {
  if (!p || which < 0 || which > 3) return S3_EINVAL;
  s3_ctx* ctx = reinterpret_cast<s3_params_view*>(p)->ctx;
  int rc = s3_broadcast(ctx, (float*)s3_params_dptr(p, which, -1), s3_params_total(p), root);
  if (rc == S3_OK && which == S3_BUF_W) s3_params_touch(p);
  return rc;
}

extern "C" int s3_params_allreduce_grads(s3_params* p) {
  if (!p) return S3_EINVAL;
  s3_ctx* ctx = reinterpret_cast<s3_params_view*>(p)->ctx;
  int n_buckets = 0;
  const int armed = s3_params_take_armed(p, &n_buckets);
  if (armed == 1) return comm_join(ctx);       // reduced bucket by bucket under the backward pass
  if (armed < 0) S3_FAIL(ctx, S3_ESTATE, "allreduce: the armed backward pass did not cover the gradient buffer");
  return s3_allreduce_sum(ctx, (float*)s3_params_dptr(p, S3_BUF_G, -1), s3_params_total(p));
}
