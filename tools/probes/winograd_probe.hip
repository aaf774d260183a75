// Cost probe for a Winograd-domain BF16X3 trunk conv (Next #2):
// F(2x2, 3x3) over (s1, s2), the three t taps kept direct.  What is measured:
// the INPUT TRANSFORM + hi/lo split a fused kernel would have to do per tile,
// on the only tile whose working set fits the 160 KB LDS at all —
//   raw halo  6 x 10 (s1, s2) x 18 t x 32 ch fp32 = 138,240 B   (2 x 4 patches
//   of 2 x 2 outputs x 16 t = the 512 output positions of today's X3 tile)
//   + one V_xi buffer 8 patches x 18 t x [hi x 32 | lo x 32] = 18,432 B
// — with the 16 transform positions xi produced one after the other (all 16 at
// once would be 295 KB).  Every lane owns (patch, t, 4 channels): it reads its
// 4 x 4 input patch from LDS (16 x ds_read_b128), forms V = B^T d B (fp32
// adds), splits each of the 16 values into bf16 hi / lo and writes the cell
// halves of ONE xi per pass (the other 15 are recomputed in the other passes
// when REUSE = 0, or kept in 64 registers and written pass by pass when
// REUSE = 1).  Prints cycles per OUTPUT POSITION per CU, to be compared with
// the MFMA time of the Winograd-domain contraction: 16 xi x 3 t-taps x 64 ci x
// 64 co x 3 split products / 4 outputs = 147,456 MAC per position = 72 cycles
// per CU at the dense bf16 rate (2048 MAC / clk / CU), 144 at the 50 % the
// direct kernels sustain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));
__device__ inline unsigned pk(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float hi_f(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

constexpr int P0 = 2, P1 = 4, TT = 18, CH = 32;
constexpr int R0 = 2 * P0 + 2, R1 = 2 * P1 + 2;          // 6 x 10 raw halo
constexpr int RAW_BYTES = R0 * R1 * TT * CH * 4;         // 138,240
constexpr int V_BYTES = P0 * P1 * TT * 128;              // 18,432
constexpr int NT = 512;

template <int REUSE>
__global__ __launch_bounds__(NT) void wino_in_kernel(const float* __restrict__ x, unsigned* __restrict__ sink,
                                                     int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* raw = reinterpret_cast<float*>(smem);
  char* vbuf = smem + RAW_BYTES;
  const int tid = threadIdx.x;
  for (int i = tid; i < RAW_BYTES / 16; i += NT)
    reinterpret_cast<uint4*>(raw)[i] = reinterpret_cast<const uint4*>(x)[i + (size_t)blockIdx.x * 64];
  __syncthreads();
  unsigned acc = 0;
  // items: 8 patches x 18 t x 8 channel groups of 4 = 1152 -> 2.25 per lane
  for (int rep = 0; rep < reps; ++rep) {
    for (int item = tid; item < P0 * P1 * TT * (CH / 4); item += NT) {
      const int cg = item % (CH / 4), t = (item / (CH / 4)) % TT, p = item / ((CH / 4) * TT);
      const int p0 = p / P1, p1 = p % P1;
      float4 d[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          d[a][b] = *reinterpret_cast<const float4*>(
              raw + ((((2 * p0 + a) * R1 + 2 * p1 + b) * TT + t) * CH + cg * 4));
      // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
      float4 u[4][4], v[4][4];
#define F4OP(r, x_, op, y_) r = make_float4(x_.x op y_.x, x_.y op y_.y, x_.z op y_.z, x_.w op y_.w)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        F4OP(u[0][b], d[0][b], -, d[2][b]); F4OP(u[1][b], d[1][b], +, d[2][b]);
        F4OP(u[2][b], d[2][b], -, d[1][b]); F4OP(u[3][b], d[1][b], -, d[3][b]);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        F4OP(v[a][0], u[a][0], -, u[a][2]); F4OP(v[a][1], u[a][1], +, u[a][2]);
        F4OP(v[a][2], u[a][2], -, u[a][1]); F4OP(v[a][3], u[a][1], -, u[a][3]);
      }
      const int cell = (p * TT + t) * 128;
      const int npass = REUSE ? 16 : 1;
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) {
        if (!REUSE && xi != (rep & 15)) continue;    // one xi per pass, the rest recomputed next pass
        const float4 w = v[xi >> 2][xi & 3];
        const unsigned h0 = pk(w.x, w.y), h1 = pk(w.z, w.w);
        const unsigned l0 = pk(w.x - lo_f(h0), w.y - hi_f(h0)), l1 = pk(w.z - lo_f(h1), w.w - hi_f(h1));
        *reinterpret_cast<uint2*>(vbuf + cell + cg * 8) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(vbuf + cell + 64 + cg * 8) = make_uint2(l0, l1);
        acc ^= h0 ^ l1;
        (void)npass;
      }
    }
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc + reinterpret_cast<unsigned*>(vbuf)[tid];
}

int main() {
  int dev = 0;
  hipSetDevice(dev);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int ncu = prop.multiProcessorCount;
  const size_t n = (size_t)RAW_BYTES / 4 + (size_t)ncu * 256 + 1024;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  float* x; unsigned* sink;
  hipMalloc(&x, n * 4); hipMalloc(&sink, ncu * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  const int lds = RAW_BYTES + V_BYTES;
  hipFuncSetAttribute(reinterpret_cast<const void*>(wino_in_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(wino_in_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double clk_ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, %.2f GHz, LDS per workgroup %d B (raw halo %d + one V_xi %d)\n", prop.name, ncu, clk_ghz, lds,
         RAW_BYTES, V_BYTES);
  for (int reuse = 0; reuse < 2; ++reuse) {
    const int reps = reuse ? 64 : 1024;   // REUSE = 0: 16 passes = one full set of xi
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      if (reuse) hipLaunchKernelGGL(wino_in_kernel<1>, dim3(ncu), dim3(NT), lds, 0, x, sink, reps);
      else hipLaunchKernelGGL(wino_in_kernel<0>, dim3(ncu), dim3(NT), lds, 0, x, sink, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // a full set of 16 xi covers the tile's 512 output positions (x 18/16 t halo) for 32 of 64 channels
      const double sets = reuse ? reps : reps / 16.0;
      const double us_per_set = ms * 1e3 / sets;
      const double clk_per_pos = us_per_set * 1e3 * clk_ghz * 2 /* both channel halves */ / 512.0;
      if (it == 2)
        printf("%s: %.2f us per full xi set of one 32-channel half -> %.1f clk per output position per CU (64 ch)\n",
               reuse ? "all 16 xi from one patch read (64 live registers)" : "one xi per pass, patch re-read and re-transformed 16 x",
               us_per_set, clk_per_pos);
    }
  }
  printf("Winograd-domain MFMA time: 72 clk per output position per CU at the dense bf16 rate, 144 at 50 %%\n");
  return 0;
}
