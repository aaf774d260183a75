// Data gradient of the 2-channel hi-res conv that opens the discriminator
// (forward 2 -> 32 over N x 78 x 78 x 286 positions; the gradient reaches the
// generator through it) on bf16 MFMA with an LDS halo — S3_PREC_BF16 plans.
//
//   dx[i][ci] = sum_{tap, co} W[tap][ci][co] * dPre[i + lo - tap][co]
//
// A 32 -> 2 "full correlation": 864 MACs per output value, and each dPre cell
// (32 channels) feeds 27 taps.  The gather kernel re-reads a cell 27 times
// through L1 (6.2 ms at C2 batch 8); here a workgroup stages the
// (4+2) x (8+2) x (16+2) dPre halo of its 4 x 8 x 16 output tile ONCE into LDS
// (fp32 -> bf16, 64-B cells, 16-B chunks XOR-swizzled by (t >> 1) & 3:
// conflict-free for the four 16-lane groups of ds_read_b128 at all three tap
// shifts, brute-forced) and every tap reads its shifted window as the MFMA B
// operand (K = 32 output channels of the forward conv = one k-step).  The A
// operand is the flipped filter: rows = the C_in <= 4 input channels (the
// other rows are zero), all 27 fragments live in registers.  Lanes 0..15 own
// the C_in values of 16 consecutive t: one contiguous store per fragment.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int DT0 = 4, DT1 = 8, DT2 = 16;
constexpr int DH0 = DT0 + 2, DH1 = DT1 + 2, DH2 = DT2 + 2;
constexpr int DHP = DH0 * DH1 * DH2;         // 1080 halo cells
constexpr int DNW = 4;                       // waves; 8 (s1, s2) rows each
constexpr int DNT = DNW * 64;
constexpr int DLDS = DHP * 64;               // 69,120 B

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

// fp32 w[tap][cin][32] -> bf16 img[tap'][16 rows][32], tap' = 26 - tap, rows >= cin zero
__global__ void dgrad_c2_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                     int cin) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 27 * 16 * 32;
       idx += gridDim.x * blockDim.x) {
    const int co = idx & 31, row = (idx >> 5) & 15, tp = idx >> 9;
    const float v = row < cin ? w[((size_t)(26 - tp) * cin + row) * 32 + co] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

__global__ __launch_bounds__(DNT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dgrad_c2_kernel(
    const float* __restrict__ dy, const unsigned short* __restrict__ img,
    float* __restrict__ dx, ConvGeom g, int tiles0, int tiles1, int tiles2, int dy16) {
  extern __shared__ __attribute__((aligned(16))) char halo[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  int tr = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int org0 = t0i * DT0, org1 = t1i * DT1, org2 = t2i * DT2;
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2];

  // ---- stage the dPre halo: cell (c0, c1, c2) = dPre[org + c + lo - 2], zero outside
  // (dy16: dPre is a bf16 tensor — the stride-2 layer above stored it that way —
  // one 16-B chunk per item, no convert)
  if (dy16) {
    const unsigned short* d16 = reinterpret_cast<const unsigned short*>(dy);
    for (int base = tid; base < DHP * 4; base += DNT * 3) {
      uint4 v[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int item = base + u * DNT;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (item < DHP * 4) {
          const int hp = item >> 2, ch = item & 3;
          int h = hp;
          const int c2 = h % DH2; h /= DH2;
          const int c1 = h % DH1; h /= DH1;
          const int c0 = h;
          const int i0 = org0 + c0 + g.lo[0] - 2, i1 = org1 + c1 + g.lo[1] - 2,
                    i2 = org2 + c2 + g.lo[2] - 2;
          if (i0 >= 0 && i0 < O0 && i1 >= 0 && i1 < O1 && i2 >= 0 && i2 < O2)
            v[u] = *reinterpret_cast<const uint4*>(d16 + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int item = base + u * DNT;
        if (item < DHP * 4) {
          const int hp = item >> 2, ch = item & 3;
          const int key = ((hp % DH2) >> 1) & 3;
          *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) = v[u];
        }
      }
    }
  } else
  for (int base = tid; base < DHP * 4; base += DNT * 3) {
    float4 va[3], vb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * DNT;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      if (item < DHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        int h = hp;
        const int c2 = h % DH2; h /= DH2;
        const int c1 = h % DH1; h /= DH1;
        const int c0 = h;
        const int i0 = org0 + c0 + g.lo[0] - 2, i1 = org1 + c1 + g.lo[1] - 2,
                  i2 = org2 + c2 + g.lo[2] - 2;
        if (i0 >= 0 && i0 < O0 && i1 >= 0 && i1 < O1 && i2 >= 0 && i2 < O2) {
          const float* src = dy + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + ch * 8;
          va[u] = *reinterpret_cast<const float4*>(src);
          vb[u] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * DNT;
      if (item < DHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        const int key = ((hp % DH2) >> 1) & 3;
        *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) =
            make_uint4(pk2(va[u].x, va[u].y), pk2(va[u].z, va[u].w), pk2(vb[u].x, vb[u].y),
                       pk2(vb[u].z, vb[u].w));
      }
    }
  }
  // ---- filter fragments: lane (row = j, kg) holds co 8 kg .. 8 kg + 7 of every tap
  // (loaded after the staging so that its registers are free during it)
  bf16x8 afr[27];
#pragma unroll
  for (int tp = 0; tp < 27; ++tp)
    afr[tp] = *reinterpret_cast<const bf16x8*>(img + (tp * 16 + j) * 32 + kg * 8);
  __syncthreads();

  // ---- 8 rows per wave x 27 taps
  int off_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) off_c[c] = (j + c) * 64 + ((kg ^ (((j + c) >> 1) & 3)) << 4);
  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int tp = (a * 3 + b) * 3 + c;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          // row r = wave * 8 + m: r0 = r / 8 = wave, r1 = m
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(
              halo + (((wave + a) * DH1 + (m + b)) * DH2) * 64 + off_c[c]);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[tp], bfr, acc[m], 0, 0, 0);
        }
      }

  // ---- C/D: col = lane & 15 (t), row = 4 kg + r (ci): lanes kg == 0 own ci 0..3
  if (kg == 0) {
    const int cin = g.Cin;
    const int o0 = org0 + wave, o2 = org2 + j;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int o1 = org1 + m;
      if (o0 < g.D[0] && o1 < g.D[1] && o2 < g.D[2]) {
        float* dst = dx + ((((size_t)n * g.D[0] + o0) * g.D[1] + o1) * g.D[2] + o2) * cin;
        if (cin == 2) {
          *reinterpret_cast<float2*>(dst) = make_float2(acc[m][0], acc[m][1]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < cin) dst[r] = acc[m][r];
        }
      }
    }
  }
}

}  // namespace

bool conv_dgrad_c2_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_DGRAD_C2)) return false;
  if (g.Cin < 1 || g.Cin > 4 || g.Cout != 32 || g.d2s != 1) return false;
  if (g.pad_mode == S3_PAD_REFLECT) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] < 0 || g.lo[d] > 2) return false;
    if (g.O[d] != g.D[d] + 2 * g.lo[d] - 2) return false;
  }
  // enough output tiles to fill the chip
  return (int64_t)g.N * ((g.D[0] + DT0 - 1) / DT0) * ((g.D[1] + DT1 - 1) / DT1) *
             ((g.D[2] + DT2 - 1) / DT2) >= 64;
}

size_t conv_dgrad_c2_packed_bytes() { return (size_t)27 * 16 * 32 * 2; }

int launch_conv_dgrad_c2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  hipLaunchKernelGGL(dgrad_c2_pack_kernel, dim3(54), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)img, g.Cin);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_dgrad_c2(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img,
                         float* dx, int dy_bf16) {
  static bool attr_set = false;
  if (!attr_set) {
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_c2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, DLDS));
    attr_set = true;
  }
  const int tiles0 = (g.D[0] + DT0 - 1) / DT0, tiles1 = (g.D[1] + DT1 - 1) / DT1,
            tiles2 = (g.D[2] + DT2 - 1) / DT2;
  hipLaunchKernelGGL(conv_dgrad_c2_kernel, dim3((unsigned)(g.N * tiles0 * tiles1 * tiles2)),
                     dim3(DNT), DLDS, ctx->stream, dy, (const unsigned short*)img, dx, g, tiles0,
                     tiles1, tiles2, dy_bf16);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
