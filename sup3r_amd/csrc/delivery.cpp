// Device -> host delivery of the chunk executor's hi-res batches by the SDMA
// engines, driven through ROCr directly.
//
// Why not hipMemcpyAsync: in this runtime a device -> pinned-host copy on a
// stream runs as a shader copy (__amd_rocclr_copyBuffer, seen in every
// rocprofv3 kernel trace of the executor), and a shader that writes PCIe-bound
// host memory stalls the rest of the chip's memory traffic while it runs: the
// next batch's forward — which the copy is supposed to hide under — ran its
// first kernels 10 - 100 x slower beside it (4 -> 64 head conv 23 us -> 1.7 -
// 3.4 ms; profiles/r04/README.md), with 2 workgroups as with a full grid.  The
// SDMA engines move the same bytes at the same 53 - 57 GB/s without touching
// the shader memory path.
//
// hsa_amd_memory_async_copy takes HSA signals, not HIP events, as
// dependencies, so the ordering with the compute stream is the caller's: begin
// a copy only after the producing work has completed (the executor's delivery
// thread waits on the batch's event first), and read the host buffer only
// after s3_dma_wait returned.
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <mutex>

#include "common.h"

namespace {

std::mutex g_mu;
int g_hsa_refs = 0;

const char* hsa_err(hsa_status_t st) {
  const char* s = nullptr;
  if (hsa_status_string(st, &s) == HSA_STATUS_SUCCESS && s) return s;
  return "unknown HSA status";
}

#define S3_HSA(ctx, call)                                                 \
  do {                                                                    \
    hsa_status_t st_ = (call);                                            \
    if (st_ != HSA_STATUS_SUCCESS) {                                      \
      (ctx)->err = std::string(#call) + ": " + hsa_err(st_);              \
      return S3_EHIP;                                                     \
    }                                                                     \
  } while (0)

}  // namespace

extern "C" int s3_dma_d2h_begin(s3_ctx* ctx, const void* src, void* dst_host, size_t bytes,
                                uint64_t* ticket) {
  if (!ctx || !src || !dst_host || !ticket || bytes == 0) return S3_EINVAL;
  *ticket = 0;
  {
    // HIP initialised ROCr long ago; our own reference keeps the calls below
    // legal whatever HIP does at teardown
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_hsa_refs == 0) S3_HSA(ctx, hsa_init());
    g_hsa_refs = 1;
  }
  hsa_amd_pointer_info_t si, di;
  si.size = sizeof(si);
  di.size = sizeof(di);
  S3_HSA(ctx, hsa_amd_pointer_info(src, &si, nullptr, nullptr, nullptr));
  S3_HSA(ctx, hsa_amd_pointer_info(dst_host, &di, nullptr, nullptr, nullptr));
  if (si.type == HSA_EXT_POINTER_TYPE_UNKNOWN || di.type == HSA_EXT_POINTER_TYPE_UNKNOWN)
    S3_FAIL(ctx, S3_EINVAL, "dma_d2h: both buffers must be ROCr allocations (device memory / s3_host_alloc)");
  hsa_signal_t sig;
  S3_HSA(ctx, hsa_signal_create(1, 0, nullptr, &sig));
  const hsa_status_t st = hsa_amd_memory_async_copy(dst_host, di.agentOwner, src, si.agentOwner, bytes, 0,
                                                    nullptr, sig);
  if (st != HSA_STATUS_SUCCESS) {
    hsa_signal_destroy(sig);
    ctx->err = std::string("hsa_amd_memory_async_copy: ") + hsa_err(st);
    return S3_EHIP;
  }
  *ticket = sig.handle;
  return S3_OK;
}

extern "C" int s3_dma_wait(s3_ctx* ctx, uint64_t ticket, int timeout_ms) {
  if (!ctx || !ticket) return S3_EINVAL;
  hsa_signal_t sig;
  sig.handle = ticket;
  // The wait hint of hsa_signal_wait is in ticks of the HSA system timestamp:
  // its frequency is queried, not assumed; the deadline itself is wall time.
  uint64_t freq = 0;
  if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq) != HSA_STATUS_SUCCESS || freq == 0)
    freq = 100000000ull;
  const uint64_t slice = freq / 10 ? freq / 10 : 1;   // blocked waits of ~100 ms
  const auto t0 = std::chrono::steady_clock::now();
  hsa_signal_value_t v = 1;
  while (true) {
    v = hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, slice, HSA_WAIT_STATE_BLOCKED);
    if (v < 1) break;
    if (timeout_ms > 0) {
      const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(
          std::chrono::steady_clock::now() - t0).count();
      // the copy may still land: the signal is left alive (its destination must
      // not be reused — the caller retires that buffer), the ticket stays valid
      if (ms >= timeout_ms) S3_FAIL(ctx, S3_ESTATE, "dma_wait: deadline passed, the copy has not completed");
    }
  }
  hsa_signal_destroy(sig);
  if (v < 0) S3_FAIL(ctx, S3_EHIP, "dma_wait: the copy engine reported an error");
  return S3_OK;
}
