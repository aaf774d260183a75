// Sustained bf16 MFMA rate and power by instruction shape and operand source:
//   v_mfma_f32_16x16x32_bf16  vs  v_mfma_f32_32x32x16_bf16
// on random operands (the chip is power-limited under dense MFMA: the question
// is joules per FLOP, MI355X_MICROARCH.md DVFS note).  Each wave holds a 64 x 64
// fp32 accumulator tile (the persistent conv's wave tile) and walks K; operands
// come (a) from registers only, rotated so every MFMA sees new bits, or (b)
// from LDS by ds_read_b128 like the conv (a 64 KB image per workgroup, same
// reads per FLOP for both shapes: (64 + 64) x K elements per 64 x 64 x K MACs).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LDS_BYTES = 64 * 1024;

template <int SHAPE, bool FROM_LDS>
__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ src, float* __restrict__ out,
                                             int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < LDS_BYTES / 16; i += blockDim.x)
    reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x * (LDS_BYTES / 16) + i) & 0xFFFFF];
  __syncthreads();
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(smem + ((tid * 4 + i) * 16) % LDS_BYTES);
    b[i] = *reinterpret_cast<const bf16x8*>(smem + ((tid * 4 + i + 2048) * 16) % LDS_BYTES);
  }
  unsigned addr = (unsigned)((lane * 16 + (tid >> 6) * 4096) % LDS_BYTES);
  if (SHAPE == 16) {
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      if (FROM_LDS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(smem + ((addr + i * 1024) & (LDS_BYTES - 1)));
          b[i] = *reinterpret_cast<const bf16x8*>(smem + ((addr + 32768 + i * 1024) & (LDS_BYTES - 1)));
        }
        addr += 4096;
      } else {
        // rotate: every MFMA sees operand bits it has not just seen
        const bf16x8 t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = b[0];
        b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * blockDim.x + tid] = s;
  } else {
    // 64 x 64 tile = 2 x 2 tiles of 32 x 32; K = 32 per iteration = 2 k-steps
    // of 16: operands a[m + 2 ks], b[n + 2 ks]
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (FROM_LDS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const bf16x8*>(smem + ((addr + i * 1024) & (LDS_BYTES - 1)));
          b[i] = *reinterpret_cast<const bf16x8*>(smem + ((addr + 32768 + i * 1024) & (LDS_BYTES - 1)));
        }
        addr += 4096;
      } else {
        const bf16x8 t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = b[0];
        b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m + 2 * ks], b[n + 2 * ks],
                                                                acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][15];
    out[blockIdx.x * blockDim.x + tid] = s;
  }
}

static void smi(const char* tag) {
  char cmd[512];
  snprintf(cmd, sizeof cmd,
           "rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Average Graphics Package Power|Current Socket' | tr '\\n' ' ' | sed 's/^/%s: /'; echo",
           tag);
  if (system(cmd)) {}
}

int main(int argc, char** argv) {
  const int zero = argc > 1 && atoi(argv[1]) == 0 ? 1 : 0;   // argv[1] = 0: all-zero operands
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n16 = 1 << 20;
  std::vector<unsigned> h(n16 * 4);
  unsigned x = 12345u;
  for (auto& v : h) {
    x = x * 1664525u + 1013904223u;
    // two bf16 in [-2, 2): random sign, exponent 0x3F / 0x3E..., random mantissa
    const unsigned lo = (x >> 1) & 0x807F, hi = (x >> 17) & 0x807F;
    v = zero ? 0u : ((0x3F00u | lo) | ((0x3F00u | hi) << 16));
  }
  uint4* src; hipMalloc(&src, n16 * 16);
  hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, 256 * 2 * 512 * 4);
  hipFuncSetAttribute((const void*)probe<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)probe<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)probe<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipFuncSetAttribute((const void*)probe<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const int grid = 256, blk = 512;             // 8 waves per CU = 2 per SIMD
  const int iters = 200000;                    // x 262144 FLOP per wave-iteration
  const double flop = (double)grid * (blk / 64) * (double)iters * 64.0 * 64.0 * 32.0 * 2.0;
  for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 4; ++v) {
      const char* name = v == 0 ? "16x16x32 regs" : v == 1 ? "32x32x16 regs" : v == 2 ? "16x16x32 lds " : "32x32x16 lds ";
      hipEventRecord(e0, st);
      if (v == 0) probe<16, false><<<grid, blk, LDS_BYTES, st>>>(src, out, iters);
      if (v == 1) probe<32, false><<<grid, blk, LDS_BYTES, st>>>(src, out, iters);
      if (v == 2) probe<16, true><<<grid, blk, LDS_BYTES, st>>>(src, out, iters);
      if (v == 3) probe<32, true><<<grid, blk, LDS_BYTES, st>>>(src, out, iters);
      hipEventRecord(e1, st);
      // sample clock / power in the middle of the run
      for (int k = 0; k < 3; ++k) { if (hipEventQuery(e1) == hipSuccess) break; smi(name); }
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s %s: %8.2f ms  %8.1f TFLOP/s\n", zero ? "zeros " : "random", name, ms, flop / (ms * 1e-3) / 1e12);
      fflush(stdout);
    }
  return 0;
}
