// General implicit-GEMM Conv on the matrix cores for everything the halo-tile
// kernels do not cover: any stride, any (virtual) padding, C_in a multiple of
// 32, fp32 activations — the strided / valid-padded discriminator convs (K3 of
// 32->32 s2, 32->64, 64->64 s2, 64->128 ...), forward, data
// gradient and weight gradient.  bf16 operands, fp32 accumulate
// (v_mfma_f32_16x16x32_bf16); used by S3_PREC_BF16 plans only.
//
// No LDS halo: with strides and ragged valid extents a halo tile is mostly
// padding, so the position operand is GATHERED — lane (position p, k-group kq)
// of a fragment loads the 8 consecutive channels kq*8.. of its own input cell
// (two float4, L1/L2-resident: a cell is re-read by up to 27 taps) and packs
// them to bf16 in registers.  The filter operand comes from a packed bf16
// image [tap][row][K] whose rows are 16-B-chunk contiguous for a lane.
//
//   forward : D[co][pos] += W[tap][co][ci] * X[pos*s + tap - lo][ci]
//   dgrad   : D[ci][ipos] += Wt[tap][ci][co] * dY[(ipos + lo - tap) / s][co]
//             (same kernel: the "input" is dY, the tap map is the adjoint one;
//             a tap contributes only where the division is exact)
//   wgrad   : see gconv_wgrad_kernel below (contraction over positions).
//
// Operands are swapped (A = filter rows, B = positions) so that lane
// (position, kq) owns 4 consecutive output channels per N fragment: float4
// stores, 64 contiguous bytes per position across the 4 k-groups.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int GT_N = 64;       // output-channel tile (4 fragments)
constexpr int GT_MF = 2;       // position fragments per wave
constexpr int GT_WAVES = 4;
constexpr int GT_POS = GT_WAVES * GT_MF * 16;   // 128 positions per workgroup

__device__ inline unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline bf16x8 pack8(const float4& a, const float4& b) {
  uint4 u = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
  return __builtin_bit_cast(bf16x8, u);
}

// fp32 [taps][K][R] (canonical [tap][ci][co], R = C_out) or its transpose
// -> bf16 [taps][R_pad][K]; transpose_flip = 0: rows = co, K = ci (forward);
// 1: rows = ci, K = co (data gradient; tap order is handled by the kernel)
__global__ void gconv_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                  int taps, int cin, int cout, int rows_pad, int mode) {
  const int R = mode == 0 ? cout : cin, Kx = mode == 0 ? cin : cout;
  const int K = (Kx + 7) / 8 * 8;              // rows padded to whole 16-B chunks
  const int64_t total = (int64_t)taps * rows_pad * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int k = (int)(r % K); r /= K;
    const int row = (int)(r % rows_pad); r /= rows_pad;
    const int tap = (int)r;
    float v = 0.f;
    if (row < R && k < Kx) {
      const int ci = mode == 0 ? k : row, co = mode == 0 ? row : k;
      v = w[((int64_t)tap * cin + ci) * cout + co];
    }
    out[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

// ADJ = false: forward gather  i = o*s + tap - lo   (reflect / zero boundary)
// ADJ = true : adjoint gather  o = (i + lo - tap)/s  where exact and in range
template <bool ADJ>
__global__ __launch_bounds__(GT_WAVES * 64) void gconv_mfma_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpk,
    const float* __restrict__ bias, const float* __restrict__ res,
    void* __restrict__ yv, ConvGeom g, int64_t P, int rows_pad, int accumulate, int frame,
    int out_bf16) {
  float* __restrict__ y = reinterpret_cast<float*>(yv);
  // in ADJ mode: g is the FORWARD conv's geometry; positions run over its
  // input grid D, the gathered tensor x is dY on its output grid O, K = C_out
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const int K = ADJ ? g.Cout : g.Cin;          // contraction channels per tap (a cell)
  const int Kp = (K + 7) / 8 * 8;              // row length of the packed filter image
  const int R = ADJ ? g.Cin : g.Cout;          // output channels
  // frame (ADJ only): positions run over the virtually padded input frame
  // (D + 2 lo per axis); the caller folds the border back (reflect adjoint)
  const int f0 = (ADJ && frame) ? g.lo[0] : 0, f1 = (ADJ && frame) ? g.lo[1] : 0,
            f2 = (ADJ && frame) ? g.lo[2] : 0;
  const int G0 = ADJ ? g.D[0] + 2 * f0 : g.O[0], G1 = ADJ ? g.D[1] + 2 * f1 : g.O[1],
            G2 = ADJ ? g.D[2] + 2 * f2 : g.O[2];
  const int S0 = ADJ ? g.O[0] : g.D[0], S1 = ADJ ? g.O[1] : g.D[1], S2 = ADJ ? g.O[2] : g.D[2];
  const int ct = blockIdx.y;
  const int64_t pbase = (int64_t)blockIdx.x * GT_POS + wave * (GT_MF * 16);

  int pn[GT_MF], c0[GT_MF], c1[GT_MF], c2[GT_MF];
  bool pok[GT_MF];
#pragma unroll
  for (int m = 0; m < GT_MF; ++m) {
    int64_t p = pbase + m * 16 + p16;
    pok[m] = p < P;
    if (!pok[m]) p = P - 1;
    c2[m] = (int)(p % G2); p /= G2;
    c1[m] = (int)(p % G1); p /= G1;
    c0[m] = (int)(p % G0); p /= G0;
    pn[m] = (int)p;
  }
  f32x4 acc[GT_MF][4];
#pragma unroll
  for (int m = 0; m < GT_MF; ++m)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nfv = (R - ct * GT_N + 15) / 16 < 4 ? (R - ct * GT_N + 15) / 16 : 4;
  const int k0n = g.k[0], k1n = g.k[1], k2n = g.k[2];
  const int kchunks = (K + 31) / 32;      // K % 8 == 0: a lane's 8-channel group is whole or absent
  for (int ta = 0; ta < k0n; ++ta)
    for (int tb = 0; tb < k1n; ++tb)
      for (int tc = 0; tc < k2n; ++tc) {
        const int tap = (ta * k1n + tb) * k2n + tc;
        // source cell of each position under this tap
        const float* src[GT_MF];
        bool sok[GT_MF];
#pragma unroll
        for (int m = 0; m < GT_MF; ++m) {
          int i0, i1, i2;
          bool ok = true;
          if (!ADJ) {
            i0 = c0[m] * g.s[0] + ta - g.lo[0];
            i1 = c1[m] * g.s[1] + tb - g.lo[1];
            i2 = c2[m] * g.s[2] + tc - g.lo[2];
            if (g.pad_mode == S3_PAD_REFLECT) {
              i0 = s3_reflect(i0, S0); i1 = s3_reflect(i1, S1); i2 = s3_reflect(i2, S2);
            }
          } else {
            const int n0 = c0[m] - f0 + g.lo[0] - ta, n1 = c1[m] - f1 + g.lo[1] - tb,
                      n2 = c2[m] - f2 + g.lo[2] - tc;
            ok = n0 >= 0 && n1 >= 0 && n2 >= 0 && n0 % g.s[0] == 0 && n1 % g.s[1] == 0 &&
                 n2 % g.s[2] == 0;
            i0 = n0 / g.s[0]; i1 = n1 / g.s[1]; i2 = n2 / g.s[2];
          }
          ok = ok && i0 >= 0 && i0 < S0 && i1 >= 0 && i1 < S1 && i2 >= 0 && i2 < S2;
          i0 = i0 < 0 ? 0 : (i0 > S0 - 1 ? S0 - 1 : i0);
          i1 = i1 < 0 ? 0 : (i1 > S1 - 1 ? S1 - 1 : i1);
          i2 = i2 < 0 ? 0 : (i2 > S2 - 1 ? S2 - 1 : i2);
          sok[m] = ok;
          src[m] = x + ((((int64_t)pn[m] * S0 + i0) * S1 + i1) * S2 + i2) * K + kq * 8;
        }
        const unsigned short* wt = wpk + ((int64_t)tap * rows_pad + ct * GT_N + p16) * Kp + kq * 8;
        for (int kc = 0; kc < kchunks; ++kc) {
          bf16x8 wf[4], xf[GT_MF];
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            if (nf < nfv) wf[nf] = *reinterpret_cast<const bf16x8*>(wt + (int64_t)nf * 16 * Kp + kc * 32);
#pragma unroll
          for (int m = 0; m < GT_MF; ++m) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (sok[m] && kc * 32 + kq * 8 < K) {
              if (K >= 4) {
                a = *reinterpret_cast<const float4*>(src[m] + kc * 32);
                if (kc * 32 + kq * 8 + 4 < K) b = *reinterpret_cast<const float4*>(src[m] + kc * 32 + 4);
              } else {               // 2-channel cells (hi-res fields into the discriminator)
                const float2 t = *reinterpret_cast<const float2*>(src[m]);
                a.x = t.x; a.y = t.y;
              }
            }
            xf[m] = pack8(a, b);
          }
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            if (nf < nfv) {       // wave-uniform: fragments past the last channel are skipped
#pragma unroll
              for (int m = 0; m < GT_MF; ++m)
                acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[m], acc[m][nf], 0, 0, 0);
            }
        }
      }

  // epilogue: lane (position, kq) owns channels ct*64 + nf*16 + kq*4 .. +3
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
#pragma unroll
  for (int m = 0; m < GT_MF; ++m) {
    if (!pok[m]) continue;
    const int64_t p = pbase + m * 16 + p16;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int ch = ct * GT_N + nf * 16 + kq * 4;
      if (ch >= R) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[m][nf][r];
        if (!ADJ) {
          if (bias) v[r] += (ch + r < R) ? bias[ch + r] : 0.f;
          v[r] = v[r] > 0.f ? v[r] : slope * v[r];
        }
      }
      if (out_bf16) {            // (forward only, R % 4 == 0, no accumulate)
        if (!ADJ && res) {
          const float4 rr = *reinterpret_cast<const float4*>(res + p * R + ch);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yv) + p * R + ch) =
            make_uint2(pk2(v[0], v[1]), pk2(v[2], v[3]));
        continue;
      }
      float* yp = y + p * R + ch;
      if ((R & 3) == 0) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (!ADJ && res) {
          const float4 rr = *reinterpret_cast<const float4*>(res + p * R + ch);
          o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        if (accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(yp);
          o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *reinterpret_cast<float4*>(yp) = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch + r < R) {
            float o = v[r];
            if (!ADJ && res) o += res[p * R + ch + r];
            yp[r] = accumulate ? yp[r] + o : o;
          }
      }
    }
  }
}

}  // namespace

bool conv_gconv_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16) return false;
  if (getenv("SUP3R_AMD_NO_GCONV")) return false;
  if (g.d2s != 1) return false;
  // C_in = 4: the generator's first conv (a cell is one float4)
  // C_in = 2: hi-res fields into the discriminator (a cell is one float2)
  if (!(g.Cin == 2 || g.Cin == 4 || (g.Cin % 8 == 0 && g.Cin >= 32))) return false;
  return true;
}

// data gradient through the same kernel: contraction over C_out
bool conv_gconv_dgrad_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16) return false;
  if (getenv("SUP3R_AMD_NO_GCONV")) return false;
  // (a depth-to-space store is undone by the epilogue adjoint: dPre arrives in
  // the conv's own output layout)
  // reflect padding: stride-1 'same' frame + fold only
  if (g.pad_mode == S3_PAD_REFLECT)
    for (int d = 0; d < 3; ++d)
      if (g.s[d] != 1) return false;
  return g.Cout % 8 == 0 && g.Cout >= 32;
}

static int rows_padded(int r) { return (r + GT_N - 1) / GT_N * GT_N; }

size_t conv_gconv_packed_bytes(const ConvGeom& g, int dgrad) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int R = dgrad ? g.Cin : g.Cout, K = dgrad ? g.Cout : g.Cin;
  return (size_t)taps * rows_padded(R) * ((K + 7) / 8 * 8) * 2 + 64;   // + over-read of a masked K tail
}

int launch_gconv_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* packed, int dgrad) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int R = dgrad ? g.Cin : g.Cout, K = dgrad ? g.Cout : g.Cin;
  const int64_t total = (int64_t)taps * rows_padded(R) * ((K + 7) / 8 * 8);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gconv_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)packed, taps, g.Cin, g.Cout, rows_padded(R), dgrad);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_gconv_fwd(s3_ctx* ctx, const ConvGeom& g, const float* x, const void* packed,
                     const float* bias, const float* res, void* y, int out_bf16) {
  if (out_bf16 && g.Cout % 4 != 0) S3_FAIL(ctx, S3_EINVAL, "gconv: bf16 output needs C_out % 4 == 0");
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  dim3 grid((unsigned)((P + GT_POS - 1) / GT_POS), (unsigned)((g.Cout + GT_N - 1) / GT_N));
  hipLaunchKernelGGL(gconv_mfma_kernel<false>, grid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                     (const unsigned short*)packed, bias, res, y, g, P, rows_padded(g.Cout), 0, 0, out_bf16);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_gconv_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* packed_t,
                       float* dx, int accumulate, int frame) {
  const int64_t P = frame ? (int64_t)g.N * (g.D[0] + 2 * g.lo[0]) * (g.D[1] + 2 * g.lo[1]) *
                                (g.D[2] + 2 * g.lo[2])
                          : (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  dim3 grid((unsigned)((P + GT_POS - 1) / GT_POS), (unsigned)((g.Cin + GT_N - 1) / GT_N));
  hipLaunchKernelGGL(gconv_mfma_kernel<true>, grid, dim3(GT_WAVES * 64), 0, ctx->stream, dy,
                     (const unsigned short*)packed_t, nullptr, nullptr, dx, g, P,
                     rows_padded(g.Cin), accumulate, frame, 0);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
