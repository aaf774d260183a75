// Data-parallel gradient sync over RCCL (xGMI).  Replaces the host-side python
// sum of per-GPU gradient lists, AbstractSingleModel._sum_parallel_grad
// (sup3r/models/abstract.py:785-805): ONE ncclAllReduce(SUM) over the flat
// fp32 gradient buffer of the network being trained (one process per GPU).
//
// librccl.so is resolved lazily with dlopen so that libsup3r_hip.so loads (and
// exports every symbol) on a box without RCCL / without a GPU.
#include <dlfcn.h>
#include <chrono>
#include <cstring>
#include <thread>

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
typedef int (*fn_abort)(void*);
typedef int (*fn_async_err)(void*, int*);
typedef int (*fn_comm_int)(void*, int*);

struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_bcast bcast = nullptr;
  fn_errstr errstr = nullptr;
  fn_destroy destroy = nullptr;
  fn_abort abort = nullptr;
  fn_async_err async_err = nullptr;
  fn_comm_int comm_count = nullptr, comm_rank = nullptr;
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
    abort = (fn_abort)dlsym(h, "ncclCommAbort");
    async_err = (fn_async_err)dlsym(h, "ncclCommGetAsyncError");
    comm_count = (fn_comm_int)dlsym(h, "ncclCommCount");
    comm_rank = (fn_comm_int)dlsym(h, "ncclCommUserRank");
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
  ++ctx->stat[S3_STAT_ALLREDUCES];
  ++ctx->comm_issued;
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

extern "C" S3_INTERNAL int s3_comm_reduce_range(s3_ctx* ctx, float* buf, int64_t n) {
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
  ++ctx->stat[S3_STAT_BUCKETS];
  ++ctx->comm_issued;
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
  ++ctx->comm_issued;
  return S3_OK;
}

// ---- watchdog ----------------------------------------------------------------
// A collective whose peer never arrives does not return an error: the kernel
// spins on the device and every later host sync blocks for ever.  The training
// host reads its loss scalars back once per batch; s3_comm_wait is that wait
// with a deadline and an exit.
static void comm_abort(s3_ctx* ctx) {
  if (!ctx->comm) return;
  if (g_rccl.abort) (void)g_rccl.abort(ctx->comm);   // frees the communicator, kills its kernels
  ctx->comm = nullptr;
  ctx->rank = 0;
  ctx->nranks = 1;
}

extern "C" int s3_comm_wait(s3_ctx* ctx, int64_t timeout_ms) {
  if (!ctx) return S3_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (!ctx->wd_ev[i]) S3_HIP(ctx, hipEventCreateWithFlags(&ctx->wd_ev[i], hipEventDisableTiming));
  S3_HIP(ctx, hipEventRecord(ctx->wd_ev[0], ctx->stream));
  const bool two = ctx->comm_stream != nullptr;
  if (two) S3_HIP(ctx, hipEventRecord(ctx->wd_ev[1], ctx->comm_stream));
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  for (;;) {
    hipError_t e0 = hipEventQuery(ctx->wd_ev[0]);
    hipError_t e1 = two ? hipEventQuery(ctx->wd_ev[1]) : hipSuccess;
    if (e0 == hipSuccess && e1 == hipSuccess) break;
    for (hipError_t e : {e0, e1})
      if (e != hipSuccess && e != hipErrorNotReady) {
        ctx->err = std::string("s3_comm_wait: ") + hipGetErrorString(e);
        return S3_EHIP;
      }
    if (ctx->comm && g_rccl.async_err) {
      int aerr = 0;
      if (g_rccl.async_err(ctx->comm, &aerr) == 0 && aerr != 0) {
        ctx->err = std::string("s3_comm_wait: asynchronous RCCL error: ") +
                   (g_rccl.errstr ? g_rccl.errstr(aerr) : "error") + "; communicator aborted";
        comm_abort(ctx);
        return S3_ERCCL;
      }
    }
    const int64_t ms = std::chrono::duration_cast<std::chrono::milliseconds>(
                           std::chrono::steady_clock::now() - t0).count();
    if (timeout_ms >= 0 && ms >= timeout_ms) {
      ctx->err = "s3_comm_wait: device work not finished after " + std::to_string(ms) + " ms (" +
                 std::to_string(ctx->comm_issued) + " collective(s) enqueued since the last completed wait, rank " +
                 std::to_string(ctx->rank) + " of " + std::to_string(ctx->nranks) + ")" +
                 (ctx->comm ? "; communicator aborted" : "");
      comm_abort(ctx);
      return S3_ETIMEOUT;
    }
    // a short busy phase for the common case (the step is about done), then back off
    if (++spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spins < 2000 ? 50 : 1000));
  }
  ctx->comm_issued = 0;
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

// what RCCL itself says about the communicator (ncclCommCount /
// ncclCommUserRank) — not what the caller passed to s3_comm_init
extern "C" int s3_comm_info(s3_ctx* ctx, int* n_ranks, int* rank) {
  if (!ctx) return S3_EINVAL;
  int n = 1, r = 0;
  if (ctx->comm) {
    if (!g_rccl.comm_count || !g_rccl.comm_rank) S3_FAIL(ctx, S3_ERCCL, "librccl.so lacks ncclCommCount");
    if (g_rccl.comm_count(ctx->comm, &n) != 0 || g_rccl.comm_rank(ctx->comm, &r) != 0)
      S3_FAIL(ctx, S3_ERCCL, "ncclCommCount / ncclCommUserRank failed");
  }
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  return S3_OK;
}

extern "C" int s3_params_broadcast(s3_params* p, int which, int root) {
  if (!p || which < 0 || which > 3) return S3_EINVAL;
  s3_ctx* ctx = reinterpret_cast<s3_params_view*>(p)->ctx;
  int rc = s3_broadcast(ctx, (float*)s3_params_dptr(p, which, -1), s3_params_total(p), root);
  if (rc == S3_OK && which == S3_BUF_W) (void)s3_params_touch(p);
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
