// Streaming-bandwidth probe: float4 copy / 2-in-1-out kernels on mid-size
// (30 MB) and large (1 GB) buffers, several grid shapes, HIP-event timed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void copy4(const float4* __restrict__ x, float4* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) y[i] = x[i];
}
__global__ void mul4(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 u = a[i], v = b[i];
    u.x *= v.x > 0 ? 1.f : 0.2f; u.y *= v.y > 0 ? 1.f : 0.2f; u.z *= v.z > 0 ? 1.f : 0.2f; u.w *= v.w > 0 ? 1.f : 0.2f;
    y[i] = u;
  }
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long mb : {30L, 120L, 1024L}) {
    long n4 = mb * 1024 * 1024 / 16;
    float4 *a, *b, *c;
    hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&c, n4 * 16);
    hipMemsetAsync(a, 0, n4 * 16, st); hipMemsetAsync(b, 0, n4 * 16, st);
    for (int wgs : {2048, 8192, 0}) {
      int grid = wgs ? wgs : (int)((n4 + 255) / 256);
      for (int kind = 0; kind < 3; ++kind) {
        const int reps = 20;
        for (int w = 0; w < 3; ++w) copy4<<<grid, 256, 0, st>>>(a, c, n4);
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) {
          if (kind == 0) copy4<<<grid, 256, 0, st>>>(a, c, n4);
          else if (kind == 1) mul4<<<grid, 256, 0, st>>>(a, b, c, n4);
          else hipMemcpyAsync(c, a, n4 * 16, hipMemcpyDeviceToDevice, st);
        }
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (kind == 1 ? 3.0 : 2.0) * n4 * 16;
        printf("%5ld MB grid %8d %-7s %8.1f us/launch %7.2f TB/s\n", mb, grid,
               kind == 0 ? "copy4" : kind == 1 ? "mul4" : "memcpy", ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e12);
      }
    }
    hipFree(a); hipFree(b); hipFree(c);
  }
  {
    // rotating set of distinct 30 MB buffers (1.8 GB: beyond the MALL): cold reads
    const int NB = 60; long n4 = 30L * 1024 * 1024 / 16;
    std::vector<float4*> bufs(NB);
    for (auto& p : bufs) { hipMalloc(&p, n4 * 16); hipMemsetAsync(p, 0, n4 * 16, st); }
    for (int pass = 0; pass < 2; ++pass) {
      hipEventRecord(e0, st);
      for (int r = 0; r + 2 < NB; r += 3) mul4<<<2048, 256, 0, st>>>(bufs[r], bufs[r + 1], bufs[r + 2], n4);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("rotating 30 MB x %d: mul4 %8.1f us/launch %7.2f TB/s\n", NB, ms * 1e3 / (NB / 3), 3.0 * n4 * 16 * (NB / 3) / (ms * 1e-3) / 1e12);
    }
    // one big slab carved into 30 MB pieces (single allocation)
    float4* slab; hipMalloc(&slab, (size_t)NB * n4 * 16); hipMemsetAsync(slab, 0, (size_t)NB * n4 * 16, st);
    for (int pass = 0; pass < 2; ++pass) {
      hipEventRecord(e0, st);
      for (int r = 0; r + 2 < NB; r += 3) mul4<<<2048, 256, 0, st>>>(slab + r * n4, slab + (r + 1) * n4, slab + (r + 2) * n4, n4);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("slab     30 MB x %d: mul4 %8.1f us/launch %7.2f TB/s\n", NB, ms * 1e3 / (NB / 3), 3.0 * n4 * 16 * (NB / 3) / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
