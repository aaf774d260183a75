// How much does a device -> host delivery running on a second stream slow the
// kernels of the compute stream, by delivery mechanism and host-buffer kind?
// (C3 executor: the head conv of the next batch took 1.7 - 2.7 ms instead of
// 23 us beside the delivery of the previous batch's 368 MB.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/d2h_interference.hip -o tools/probes/d2h_interference
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void victim_stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void victim_alu(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
  if (a == 123.f) out[0] = a + b;
}
__global__ __launch_bounds__(256) void d2h_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    u32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i + q * 256 < n16) v[q] = __builtin_nontemporal_load(src + i + q * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i + q * 256 < n16) dst[i + q * 256] = v[q];
  }
}

static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const size_t D2H = 368ull << 20, VB = 26ull << 20;
  hipStream_t s_comp, s_copy;
  CK(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
  void *dsrc, *va, *vb; float* dout;
  CK(hipMalloc(&dsrc, D2H)); CK(hipMalloc(&va, VB)); CK(hipMalloc(&vb, VB)); CK(hipMalloc(&dout, 64));
  CK(hipMemset(dsrc, 1, D2H)); CK(hipMemset(va, 2, VB));
  // host buffers
  struct HB { const char* name; void* host; void* dev; };
  std::vector<HB> hbs;
  {
    void* p; CK(hipHostMalloc(&p, D2H, hipHostMallocMapped | hipHostMallocPortable));
    hbs.push_back({"hipHostMalloc(default)", p, nullptr});
    CK(hipHostMalloc(&p, D2H, hipHostMallocMapped | hipHostMallocNonCoherent));
    hbs.push_back({"hipHostMalloc(noncoherent)", p, nullptr});
    if (hipHostMalloc(&p, D2H, hipHostMallocMapped | hipHostMallocWriteCombined) == hipSuccess)
      hbs.push_back({"hipHostMalloc(writecombined)", p, nullptr});
    else (void)hipGetLastError();
    // transparent huge pages + register
    void* m = mmap(nullptr, D2H + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m != MAP_FAILED) {
      char* al = (char*)(((uintptr_t)m + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
      int rc = madvise(al, D2H, MADV_HUGEPAGE);
      memset(al, 0, D2H);
      hipError_t e = hipHostRegister(al, D2H, hipHostRegisterMapped | hipHostRegisterPortable);
      printf("THP buffer: madvise rc %d, hipHostRegister %s\n", rc, hipGetErrorString(e));
      if (e == hipSuccess) hbs.push_back({"mmap+THP+hipHostRegister", al, nullptr}); else (void)hipGetLastError();
    }
    void* m2 = mmap(nullptr, D2H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_HUGETLB, -1, 0);
    if (m2 != MAP_FAILED) {
      memset(m2, 0, D2H);
      hipError_t e = hipHostRegister(m2, D2H, hipHostRegisterMapped | hipHostRegisterPortable);
      printf("MAP_HUGETLB buffer: hipHostRegister %s\n", hipGetErrorString(e));
      if (e == hipSuccess) hbs.push_back({"mmap(HUGETLB)+hipHostRegister", m2, nullptr}); else (void)hipGetLastError();
    } else printf("MAP_HUGETLB: not available\n");
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (f) { char buf[128] = {0}; if (fgets(buf, 127, f)) printf("THP enabled: %s", buf); fclose(f); }
  }
  for (auto& h : hbs) CK(hipHostGetDevicePointer(&h.dev, h.host, 0));
  hipEvent_t e0, e1, c0, c1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
  auto victim = [&](int kind) {
    if (kind == 0) hipLaunchKernelGGL(victim_stream, dim3(2048), dim3(256), 0, s_comp, (const u32x4*)va, (u32x4*)vb, VB / 16);
    else hipLaunchKernelGGL(victim_alu, dim3(1024), dim3(256), 0, s_comp, dout, 20000);
  };
  auto time_victim = [&](int kind) {
    CK(hipEventRecord(e0, s_comp)); victim(kind); CK(hipEventRecord(e1, s_comp)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
  };
  for (int kind = 0; kind < 2; ++kind) {
    for (int i = 0; i < 3; ++i) time_victim(kind);
    std::vector<float> t; for (int i = 0; i < 9; ++i) t.push_back(time_victim(kind));
    printf("victim %s alone: %.3f ms\n", kind ? "ALU-only" : "26 MB stream copy", median(t));
  }
  // ---- what precedes the big hipMemcpyAsync on the copy stream (the executor
  // waits on an event of the compute stream and copies 6 KB of statistics first)
  {
    void* small_h; CK(hipHostMalloc(&small_h, 8192, hipHostMallocMapped));
    hipEvent_t ready; CK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    const char* names[] = {"nothing", "tiny kernel", "6 KB D2H memcpy", "wait on compute-stream event",
                           "event wait + 6 KB memcpy", "6 KB memcpy AFTER the big one"};
    for (int pre = 0; pre < 6; ++pre) {
      std::vector<float> tv, tc;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        victim(0); CK(hipEventRecord(ready, s_comp));
        CK(hipEventRecord(c0, s_copy));
        if (pre == 1) hipLaunchKernelGGL(victim_alu, dim3(1), dim3(64), 0, s_copy, dout, 10);
        if (pre == 3 || pre == 4) CK(hipStreamWaitEvent(s_copy, ready, 0));
        if (pre == 2 || pre == 4) CK(hipMemcpyAsync(small_h, dsrc, 6144, hipMemcpyDeviceToHost, s_copy));
        CK(hipMemcpyAsync(hbs[0].host, dsrc, D2H, hipMemcpyDeviceToHost, s_copy));
        if (pre == 5) CK(hipMemcpyAsync(small_h, dsrc, 6144, hipMemcpyDeviceToHost, s_copy));
        CK(hipEventRecord(c1, s_copy));
        std::vector<float> inner;
        for (int i = 0; i < 400 && hipEventQuery(c1) == hipErrorNotReady; ++i) inner.push_back(time_victim(0));
        (void)hipGetLastError();
        CK(hipEventSynchronize(c1));
        float ms; CK(hipEventElapsedTime(&ms, c0, c1)); tc.push_back(ms);
        if (inner.size() > 2) { inner.pop_back(); tv.push_back(median(inner)); }
      }
      printf("big memcpyAsync preceded by %-32s: victim beside %8.3f ms, delivery %7.2f ms\n", names[pre],
             tv.empty() ? -1.f : median(tv), median(tc));
    }
  }
  for (auto& h : hbs) {
    if (&h != &hbs[0]) break;
    for (int mech = 0; mech < 4; ++mech) {          // 0 memcpyAsync, 1..3 kernel with 2 / 16 / 128 blocks
      const int blocks = mech == 1 ? 2 : (mech == 2 ? 16 : 128);
      for (int kind = 0; kind < 2; ++kind) {
        std::vector<float> tv, tc;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(c0, s_copy));
          if (mech == 0) CK(hipMemcpyAsync(h.host, dsrc, D2H, hipMemcpyDeviceToHost, s_copy));
          else hipLaunchKernelGGL(d2h_kernel, dim3(blocks), dim3(256), 0, s_copy, (const u32x4*)dsrc, (u32x4*)h.dev, D2H / 16);
          CK(hipEventRecord(c1, s_copy));
          // victims launched back to back while the delivery runs
          std::vector<float> inner;
          for (int i = 0; i < 200 && hipEventQuery(c1) == hipErrorNotReady; ++i) inner.push_back(time_victim(kind));
          (void)hipGetLastError();
          CK(hipEventSynchronize(c1));
          float ms; CK(hipEventElapsedTime(&ms, c0, c1)); tc.push_back(ms);
          if (inner.size() > 2) { inner.pop_back(); tv.push_back(median(inner)); }
        }
        printf("%-30s %-14s victim %-8s beside: %8.3f ms (median of medians), delivery %7.2f ms = %5.1f GB/s\n",
               h.name, mech == 0 ? "memcpyAsync" : (mech == 1 ? "kernel x2" : (mech == 2 ? "kernel x16" : "kernel x128")),
               kind ? "ALU" : "stream", tv.empty() ? -1.f : median(tv), median(tc), D2H / median(tc) / 1e6);
      }
    }
  }
  return 0;
}
