// General implicit-GEMM Conv on the matrix cores for everything the halo-tile
// kernels do not cover: any stride, any (virtual) padding, C_in a multiple of
// 32, fp32 activations — the strided / valid-padded discriminator convs (K3 of
// 32->32 s2, 32->64, 64->64 s2, 64->128 ...), forward, data
// gradient and weight gradient.  bf16 operands, fp32 accumulate
// (v_mfma_f32_16x16x32_bf16).  S3_PREC_BF16 plans; S3_PREC_BF16X3 plans run
// the X3 variant of the same kernel: the gathered fp32 cell and the filter are
// split into bf16 pairs (hi = bf16(v), lo = bf16(v - hi); the filter's lo
// image sits behind its hi image) and every product is lo*hi + hi*lo + hi*hi
// — fp32-class forward and data gradients of the discriminator convs at three
// MFMAs per product instead of the direct fp32 kernels.
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
constexpr int GT_WAVES = 4;

__device__ inline unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline bf16x8 pack8(const float4& a, const float4& b) {
  uint4 u = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
  return __builtin_bit_cast(bf16x8, u);
}
// the same with the rounding residue: hi = bf16(v), lo = bf16(v - hi)
__device__ inline void split8(const float4& a, const float4& b, bf16x8& hi, bf16x8& lo) {
  const uint4 h = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
  auto lo_f = [](unsigned u) { return __uint_as_float(u << 16); };
  auto hi_f = [](unsigned u) { return __uint_as_float(u & 0xFFFF0000u); };
  const uint4 l = make_uint4(pk2(a.x - lo_f(h.x), a.y - hi_f(h.x)), pk2(a.z - lo_f(h.y), a.w - hi_f(h.y)),
                             pk2(b.x - lo_f(h.z), b.y - hi_f(h.z)), pk2(b.z - lo_f(h.w), b.w - hi_f(h.w)));
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// fp32 [taps][K][R] (canonical [tap][ci][co], R = C_out) or its transpose
// -> bf16 [taps][R_pad][K]; transpose_flip = 0: rows = co, K = ci (forward);
// 1: rows = ci, K = co (data gradient; tap order is handled by the kernel)
// lo != nullptr (BF16X3): also the image of the rounding residues bf16(v - bf16(v))
__global__ void gconv_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                  int taps, int cin, int cout, int rows_pad, int mode,
                                  unsigned short* __restrict__ lo = nullptr) {
  const int R = mode == 0 ? cout : cin, Kx = mode == 0 ? cin : cout;
  const int K = (Kx + 7) / 8 * 8;              // rows padded to whole 16-B chunks
  const int64_t total = (int64_t)taps * rows_pad * K;
  // The kernels walk K in steps of 32 channels: with K < 32 (C_in = 2 / 4 off the
  // taps-in-K path) the last row's step reads up to 48 B past the image — times
  // zero-padded activations.  Those 64 B are part of the image and written HERE,
  // so the product is 0 x 0 whatever the buffer held before (round 5: this used
  // to rest on plan buffers being zero-filled at allocation).
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total + 32;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx >= total) {
      out[idx] = 0;
      if (lo) lo[idx] = 0;
      continue;
    }
    int64_t r = idx;
    const int k = (int)(r % K); r /= K;
    const int row = (int)(r % rows_pad); r /= rows_pad;
    const int tap = (int)r;
    float v = 0.f;
    if (row < R && k < Kx) {
      const int ci = mode == 0 ? k : row, co = mode == 0 ? row : k;
      v = w[((int64_t)tap * cin + ci) * cout + co];
    }
    const unsigned h = pk2(v, 0.f) & 0xFFFFu;
    out[idx] = (unsigned short)h;
    if (lo) lo[idx] = (unsigned short)(pk2(v - __uint_as_float(h << 16), 0.f) & 0xFFFFu);
  }
}

// ADJ = false: forward gather  i = o*s + tap - lo   (reflect / zero boundary)
// ADJ = true : adjoint gather  o = (i + lo - tap)/s  where exact and in range
// MF: position fragments per wave.  The four filter fragments of a (tap, k-chunk)
// come from L1 / L2 for every wave: with MF = 2 they are two thirds of a
// wave's vector-memory bytes (4 KB of filter + 2 KB of gathered cells per 8
// MFMAs), with MF = 4 half (4 + 4 KB per 16 MFMAs) — used whenever the grid
// still fills the chip.
template <bool ADJ, int MF, bool X3 = false>
__global__ __launch_bounds__(GT_WAVES * 64) void gconv_mfma_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpk,
    const float* __restrict__ bias, const float* __restrict__ res,
    void* __restrict__ yv, ConvGeom g, int64_t P, int rows_pad, int accumulate, int frame,
    int out_bf16, int x16, int nsplit, float* __restrict__ part, int64_t lo_off = 0) {
  float* __restrict__ y = reinterpret_cast<float*>(yv);
  // nsplit > 1 (few positions, long contraction): blockIdx.z also enumerates
  // slices of the (tap, k-chunk) sequence; a slice leaves its raw fp32 sums in
  // part[slice][position][channel] and gconv_splitk_epilogue finishes the job
  const int split = nsplit > 1 ? (int)(blockIdx.z % nsplit) : 0;
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
  const int64_t pbase = (int64_t)blockIdx.x * (GT_WAVES * MF * 16) + wave * (MF * 16);

  // strided data gradient: positions are enumerated per residue class
  // (c mod s per axis, blockIdx.z) — all lanes of a workgroup then share the
  // taps that reach them (ta = (r + lo) mod s, step s) instead of masking
  // 1 - 1/(s0 s1 s2) of the 27 taps
  const bool strided = ADJ && (g.s[0] > 1 || g.s[1] > 1 || g.s[2] > 1);
  int r0 = 0, r1 = 0, r2 = 0, E0 = G0, E1 = G1, E2 = G2;
  int64_t Pc = P;
  if (strided) {
    const int cls = nsplit > 1 ? (int)(blockIdx.z / nsplit) : (int)blockIdx.z;
    r2 = cls % g.s[2]; r1 = (cls / g.s[2]) % g.s[1]; r0 = cls / (g.s[2] * g.s[1]);
    E0 = (G0 - r0 + g.s[0] - 1) / g.s[0]; E1 = (G1 - r1 + g.s[1] - 1) / g.s[1];
    E2 = (G2 - r2 + g.s[2] - 1) / g.s[2];
    Pc = (int64_t)g.N * E0 * E1 * E2;
    if ((int64_t)blockIdx.x * (GT_WAVES * MF * 16) >= Pc) return;
  }
  const int st0 = strided ? g.s[0] : 1, st1 = strided ? g.s[1] : 1, st2 = strided ? g.s[2] : 1;
  const int ta0 = strided ? (r0 + g.lo[0]) % g.s[0] : 0, tb0 = strided ? (r1 + g.lo[1]) % g.s[1] : 0,
            tc0 = strided ? (r2 + g.lo[2]) % g.s[2] : 0;

  int pn[MF], c0[MF], c1[MF], c2[MF];
  int64_t plin[MF];
  bool pok[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    int64_t p = pbase + m * 16 + p16;
    pok[m] = p < Pc;
    if (!pok[m]) p = Pc - 1;
    plin[m] = p;
    if (Pc <= 0x7fffffffLL) {
      // (32-bit: three 64-bit divisions per fragment and lane were as many
      // VALU instructions as the 27 taps' gathers)
      unsigned r = (unsigned)p, q;
      q = r / (unsigned)E2; c2[m] = (int)(r - q * (unsigned)E2); r = q;
      q = r / (unsigned)E1; c1[m] = (int)(r - q * (unsigned)E1); r = q;
      q = r / (unsigned)E0; c0[m] = (int)(r - q * (unsigned)E0);
      pn[m] = (int)q;
    } else {
      c2[m] = (int)(p % E2); p /= E2;
      c1[m] = (int)(p % E1); p /= E1;
      c0[m] = (int)(p % E0); p /= E0;
      pn[m] = (int)p;
    }
    if (strided) {
      c0[m] = c0[m] * g.s[0] + r0; c1[m] = c1[m] * g.s[1] + r1; c2[m] = c2[m] * g.s[2] + r2;
      plin[m] = (((int64_t)pn[m] * G0 + c0[m]) * G1 + c1[m]) * G2 + c2[m];
    }
  }
  f32x4 acc[MF][4];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // forward, no padding: the gather address is base(position) + offset(tap)
  bool inrange = !ADJ && g.pad_mode != S3_PAD_REFLECT;
  int64_t fbase[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) fbase[m] = 0;
  if (!ADJ) {
    for (int d = 0; d < 3; ++d)
      inrange = inrange && g.lo[d] == 0 && (g.O[d] - 1) * g.s[d] + g.k[d] <= g.D[d];
    if (inrange) {
#pragma unroll
      for (int m = 0; m < MF; ++m)
        fbase[m] = ((((int64_t)pn[m] * S0 + c0[m] * g.s[0]) * S1 + c1[m] * g.s[1]) * S2 +
                    c2[m] * g.s[2]) * K;
    }
  }

  const int nfv = (R - ct * GT_N + 15) / 16 < 4 ? (R - ct * GT_N + 15) / 16 : 4;
  const int k0n = g.k[0], k1n = g.k[1], k2n = g.k[2];
  const int kchunks = (K + 31) / 32;      // K % 8 == 0: a lane's 8-channel group is whole or absent
  // this slice's range of the flattened (visited tap, k-chunk) sequence
  int it_lo = 0, it_hi = 0x7fffffff;
  if (nsplit > 1) {
    const int ntap = ((k0n - ta0 + st0 - 1) / st0) * ((k1n - tb0 + st1 - 1) / st1) * ((k2n - tc0 + st2 - 1) / st2);
    const int iters = ntap * kchunks;
    it_lo = (int)((int64_t)iters * split / nsplit);
    it_hi = (int)((int64_t)iters * (split + 1) / nsplit);
  }
  int it_tap = 0;                         // first flattened index of the current tap
  for (int ta = ta0; ta < k0n; ta += st0)
    for (int tb = tb0; tb < k1n; tb += st1)
      for (int tc = tc0; tc < k2n; tc += st2, it_tap += kchunks) {
        if (it_tap + kchunks <= it_lo || it_tap >= it_hi) continue;
        const int tap = (ta * k1n + tb) * k2n + tc;
        // source cell of each position under this tap
        const float* src[MF];
        bool sok[MF];
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          int i0, i1, i2;
          bool ok = true;
          if (!ADJ && inrange) {
            // valid padding: every tap of every output position is inside the input
            sok[m] = true;
            src[m] = x + fbase[m] + (((int64_t)ta * S1 + tb) * S2 + tc) * K + kq * 8;
            continue;
          }
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
          if (it_tap + kc < it_lo || it_tap + kc >= it_hi) continue;
          bf16x8 wf[4], xf[MF];
          bf16x8 wl[X3 ? 4 : 1], xl[X3 ? MF : 1];     // BF16X3: the residue halves
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            if (nf < nfv) {
              wf[nf] = *reinterpret_cast<const bf16x8*>(wt + (int64_t)nf * 16 * Kp + kc * 32);
              if constexpr (X3)
                wl[nf] = *reinterpret_cast<const bf16x8*>(wt + lo_off + (int64_t)nf * 16 * Kp + kc * 32);
            }
#pragma unroll
          for (int m = 0; m < MF; ++m) {
            if (!X3 && x16) {
              // bf16 cells (bf16 saved activations, K % 8 == 0): the lane's 8
              // channels are one 16-B load; src[] was computed in fp32 elements
              uint4 u = make_uint4(0, 0, 0, 0);
              if (sok[m] && kc * 32 + kq * 8 < K)
                u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(x) +
                                                    (src[m] - x) + kc * 32);
              xf[m] = __builtin_bit_cast(bf16x8, u);
              continue;
            }
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
            if constexpr (X3) split8(a, b, xf[m], xl[m]);
            else xf[m] = pack8(a, b);
          }
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            if (nf < nfv) {       // wave-uniform: fragments past the last channel are skipped
#pragma unroll
              for (int m = 0; m < MF; ++m) {
                if constexpr (X3) {   // small terms first
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nf], xf[m], acc[m][nf], 0, 0, 0);
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xl[m], acc[m][nf], 0, 0, 0);
                }
                acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[m], acc[m][nf], 0, 0, 0);
              }
            }
        }
      }

  // epilogue: lane (position, kq) owns channels ct*64 + nf*16 + kq*4 .. +3
  if (nsplit > 1) {
    const int64_t Pall = ADJ ? (int64_t)g.N * G0 * G1 * G2 : P;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      if (!pok[m]) continue;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int ch = ct * GT_N + nf * 16 + kq * 4;
        if (ch >= R) continue;
        float* pp = part + ((int64_t)split * Pall + plin[m]) * R + ch;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch + r < R) pp[r] = acc[m][nf][r];
      }
    }
    return;
  }
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    if (!pok[m]) continue;
    const int64_t p = plin[m];
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


// sums the slices of a split contraction in fixed order, then the forward
// epilogue (bias, activation, residual, fp32 or bf16 store) or the data
// gradient's plain / accumulating store
__global__ void gconv_splitk_epilogue(const float* __restrict__ part, int nsplit, int64_t P, int R,
                                      const float* __restrict__ bias, const float* __restrict__ res,
                                      void* __restrict__ yv, int fwd, float slope, int out_bf16,
                                      int accumulate) {
  const int64_t total = P * R;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < nsplit; ++s) v += part[(int64_t)s * total + i];
    if (fwd) {
      if (bias) v += bias[i % R];
      v = v > 0.f ? v : slope * v;
      if (res) v += res[i];
    }
    if (out_bf16) {
      reinterpret_cast<unsigned short*>(yv)[i] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
    } else {
      float* y = reinterpret_cast<float*>(yv);
      y[i] = accumulate ? y[i] + v : v;
    }
  }
}

// Slices of the contraction for a launch of `wgs` workgroups walking `iters`
// (tap, k-chunk) steps each: only when the grid leaves most CUs idle and the
// walk is long (the 128 / 256-channel discriminator layers on a few thousand
// positions: 16 workgroups x 216 dependent steps = 0.34 ms for 1.7 GFLOP)
int gconv_splits(const s3_ctx* ctx, int64_t wgs, int iters) {
  if (s3_opt_has(S3O_NO_GCONV_SPLITK) || wgs >= ctx->num_cu || iters < 32) return 1;
  int64_t n = (2 * (int64_t)ctx->num_cu + wgs - 1) / wgs;
  if (n > iters / 6) n = iters / 6;
  if (n > 32) n = 32;
  return n < 2 ? 1 : (int)n;
}

// ---------------------------------------------------------------------------
// Few input channels (C_in = 2: hi-res fields into the discriminator, C_in = 4:
// the generator's first conv): the taps are packed INTO the contraction index,
// k = tap * C_in + ci (27 C_in = 54 / 108 -> 2 / 4 chunks of 32), instead of
// one mostly-empty K = 32 MFMA per tap.  Lane (position, kq) of a chunk gathers
// its 8 / C_in own taps (one float2 / float4 each); the per-lane tap offsets
// are computed once, the filter fragments live in registers for the whole
// workgroup, and each wave walks FC_MF position fragments.  These layers are
// bound by the store of the C_out-wide output.
constexpr int FC_MF = 4;                       // position fragments per wave
constexpr int FC_POS = GT_WAVES * FC_MF * 16;  // 256 positions per workgroup

// X3 (BF16X3 plans): gathered taps and filter split into bf16 pairs, three
// MFMAs per product; the filter's residue image sits lo_off elements behind
template <int CIN, bool X3 = false>
__global__ __launch_bounds__(GT_WAVES * 64) void gconv_fewch_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpk,
    const float* __restrict__ bias, const float* __restrict__ res,
    void* __restrict__ yv, ConvGeom g, int64_t P, int out_bf16, int lo_off = 0) {
  constexpr int TPL = 8 / CIN;                 // taps per lane per chunk
  constexpr int KC = (27 * CIN + 31) / 32;     // chunks of 32
  constexpr int KP = KC * 32;
  float* __restrict__ y = reinterpret_cast<float*>(yv);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const int R = g.Cout, ct = blockIdx.y;
  const int S0 = g.D[0], S1 = g.D[1], S2 = g.D[2];
  const int nfv = (R - ct * GT_N + 15) / 16 < 4 ? (R - ct * GT_N + 15) / 16 : 4;
  bool inrange = g.pad_mode != S3_PAD_REFLECT;
  for (int d = 0; d < 3; ++d)
    inrange = inrange && g.lo[d] == 0 && (g.O[d] - 1) * g.s[d] + g.k[d] <= g.D[d];

  // this lane's taps: chunk kc, slot j -> tap = (kc*32 + kq*8) / CIN + j
  int toff[KC][TPL], tdel[KC][TPL];            // element offset; packed (ta, tb, tc), -1 = none
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int j = 0; j < TPL; ++j) {
      const int tap = (kc * 32 + kq * 8) / CIN + j;
      const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
      toff[kc][j] = tap < 27 ? ((ta * S1 + tb) * S2 + tc) * CIN : 0;
      tdel[kc][j] = tap < 27 ? (ta | (tb << 8) | (tc << 16)) : -1;
    }
  // filter fragments (A operand: rows = output channels)
  bf16x8 wf[KC][4];
  bf16x8 wl[X3 ? KC : 1][4];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
      if (nf < nfv) {
        wf[kc][nf] = *reinterpret_cast<const bf16x8*>(
            wpk + ((int64_t)ct * GT_N + nf * 16 + p16) * KP + kc * 32 + kq * 8);
        if constexpr (X3)
          wl[kc][nf] = *reinterpret_cast<const bf16x8*>(
              wpk + lo_off + ((int64_t)ct * GT_N + nf * 16 + p16) * KP + kc * 32 + kq * 8);
      }
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  float bv[4][4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = ct * GT_N + nf * 16 + kq * 4 + r;
      bv[nf][r] = (bias && ch < R) ? bias[ch] : 0.f;
    }

  const int64_t pbase = (int64_t)blockIdx.x * FC_POS + wave * (FC_MF * 16);
#pragma unroll 1
  for (int m = 0; m < FC_MF; ++m) {
    int64_t p = pbase + m * 16 + p16;
    const bool pok = p < P;
    if (!pok) p = P - 1;
    const int64_t pp = p;
    const int c2 = (int)(p % g.O[2]); p /= g.O[2];
    const int c1 = (int)(p % g.O[1]); p /= g.O[1];
    const int c0 = (int)(p % g.O[0]); p /= g.O[0];
    const int pn = (int)p;
    const int b0 = c0 * g.s[0] - g.lo[0], b1 = c1 * g.s[1] - g.lo[1], b2 = c2 * g.s[2] - g.lo[2];
    const float* xb = x + ((((int64_t)pn * S0 + b0) * S1 + b1) * S2 + b2) * CIN;
    f32x4 acc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      float v[8];
#pragma unroll
      for (int j = 0; j < TPL; ++j) {
        const float* src = xb + toff[kc][j];
        bool ok = tdel[kc][j] >= 0;
        if (!inrange) {
          const int ta = tdel[kc][j] & 255, tb = (tdel[kc][j] >> 8) & 255, tc = (tdel[kc][j] >> 16) & 255;
          int i0 = b0 + ta, i1 = b1 + tb, i2 = b2 + tc;
          if (g.pad_mode == S3_PAD_REFLECT) {
            i0 = s3_reflect(i0, S0); i1 = s3_reflect(i1, S1); i2 = s3_reflect(i2, S2);
          }
          ok = ok && i0 >= 0 && i0 < S0 && i1 >= 0 && i1 < S1 && i2 >= 0 && i2 < S2;
          i0 = i0 < 0 ? 0 : (i0 > S0 - 1 ? S0 - 1 : i0);
          i1 = i1 < 0 ? 0 : (i1 > S1 - 1 ? S1 - 1 : i1);
          i2 = i2 < 0 ? 0 : (i2 > S2 - 1 ? S2 - 1 : i2);
          src = x + ((((int64_t)pn * S0 + i0) * S1 + i1) * S2 + i2) * CIN;
        }
        if (CIN == 2) {
          float2 t = make_float2(0.f, 0.f);
          if (ok) t = *reinterpret_cast<const float2*>(src);
          v[2 * j] = t.x; v[2 * j + 1] = t.y;
        } else {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok) t = *reinterpret_cast<const float4*>(src);
          v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        }
      }
      bf16x8 xf, xl;
      if constexpr (X3) {
        split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), xf, xl);
      } else {
        const uint4 u = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
        xf = __builtin_bit_cast(bf16x8, u);
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        if (nf < nfv) {
          if constexpr (X3) {
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[kc][nf], xf, acc[nf], 0, 0, 0);
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc][nf], xl, acc[nf], 0, 0, 0);
          }
          acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc][nf], xf, acc[nf], 0, 0, 0);
        }
    }
    if (!pok) continue;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int ch = ct * GT_N + nf * 16 + kq * 4;
      if (nf >= nfv || ch >= R) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = acc[nf][r] + bv[nf][r];
        o[r] = o[r] > 0.f ? o[r] : slope * o[r];
      }
      if ((R & 3) == 0) {
        if (res) {
          const float4 rr = *reinterpret_cast<const float4*>(res + pp * R + ch);
          o[0] += rr.x; o[1] += rr.y; o[2] += rr.z; o[3] += rr.w;
        }
        if (out_bf16)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yv) + pp * R + ch) =
              make_uint2(pk2(o[0], o[1]), pk2(o[2], o[3]));
        else
          *reinterpret_cast<float4*>(y + pp * R + ch) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch + r < R) y[pp * R + ch + r] = o[r] + (res ? res[pp * R + ch + r] : 0.f);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// LDS-halo variant for stride-1 few-channel convs with many positions (the
// 2 -> 32 discriminator layer over 13.9 M positions, the 8 -> 2 tail conv's
// data gradient as a 2-channel conv over the padded frame): the input halo of
// a 4 x 8 x 32 output tile is tiny (2040 cells x 2 C_in bytes as bf16), so it
// is staged once with the padding rule applied (reflect or zero), and the B
// operand of a position fragment is 8 / C_in LDS reads at per-lane tap offsets
// — no global gather, no per-fragment index math.  The layer is then bound by
// its output stores.
constexpr int FH0 = 4, FH1 = 8, FH2 = 32;
constexpr int FG0 = FH0 + 2, FG1 = FH1 + 2, FG2 = FH2 + 2;
constexpr int FHP = FG0 * FG1 * FG2;          // 2040 halo cells
constexpr int FHW = 4;                        // waves; 16 fragments each

// bf16 output in whole 32-channel halves with a [0, 1] activation slope: the
// permuted-row / lean-walk variant of gconv_fewch_halo_kernel (PERM)
bool fewch_halo_perm(const ConvGeom& g, int out_bf16) {
  const int r = g.Cout > GT_N ? GT_N : g.Cout;   // (every cout tile must qualify)
  return out_bf16 && (g.Cout == 32 || (g.Cout & 63) == 0) && (r & 31) == 0 &&
         !(g.act == S3_ACT_LEAKY && !(g.alpha >= 0.f && g.alpha <= 1.f));
}

// MODE 0: generic walk (any C_out, fp32 or bf16 out).  MODE 1: lean walk,
// permuted filter rows, bf16 out (C_out = 32 or a multiple of 64).  MODE 2:
// lean walk, natural rows, fp32 out (C_out % 4 == 0; the 2 -> 8 data gradient of
// the hi-res tail conv over its padded frame).  NFP: N fragments of the lean walks.
// X3 (MODE 2 only: BF16X3 plans, fp32 in / out): the halo holds a second image
// of the rounding residues lo = bf16(v - bf16(v)), the filter its lo fragments
// (wpk + lo_off), and every product is hi * hi + hi * lo + lo * hi on the matrix
// cores — the first discriminator layer of a BF16X3 training step left the
// gather walk of gconv_fewch_kernel<.., true> (1.18 ms at C2 batch 8).
template <int CIN, int MODE, int NFP = 4, bool X3 = false>
__global__ __launch_bounds__(FHW * 64) void gconv_fewch_halo_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wpk,
    const float* __restrict__ bias, void* __restrict__ yv, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int out_bf16, unsigned char* __restrict__ sign, int lo_off = 0) {
  static_assert(!X3 || MODE == 2, "split-bf16: the lean fp32-out walk");
  constexpr int TPL = 8 / CIN;                 // taps per lane per chunk
  constexpr int KC = (27 * CIN + 31) / 32;     // chunks of 32
  constexpr int KP = KC * 32;
  constexpr int CELLB = 2 * CIN;               // bytes per halo cell (bf16)
  constexpr int LO_IMG = (FHP + 1) * CELLB;    // byte offset of the lo image (X3)
  __shared__ __attribute__((aligned(16))) char halo[(FHP + 1) * CELLB * (X3 ? 2 : 1)];   // + one zero cell
  float* __restrict__ y = reinterpret_cast<float*>(yv);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p16 = lane & 15, kq = lane >> 4;
  const int R = g.Cout, ct = blockIdx.y;
  int tr = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int org0 = t0i * FH0, org1 = t1i * FH1, org2 = t2i * FH2;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  constexpr bool PERM = MODE == 1, LEAN = MODE != 0;
  const int nfv = LEAN ? NFP : ((R - ct * GT_N + 15) / 16 < 4 ? (R - ct * GT_N + 15) / 16 : 4);
  // bf16 output with whole 32-channel halves: filter rows are taken in the
  // order (half h, kq, nf & 1, r) so that a lane's C/D values of a fragment
  // pair are 8 CONSECUTIVE channels h*32 + kq*8 .. +7 — one 16-B store per
  // lane, a whole 64-B row per position across the four k-groups (with the
  // natural order a lane stored 8 B and a wave store was sixteen 32-B pieces:
  // the 0.89 GB first discriminator activation went out at 2 TB/s)
  constexpr bool perm = PERM;

  // ---- stage the halo: cell (c0, c1, c2) = x[org + c - lo] under the padding rule
  for (int hp = tid; hp < FHP + 1; hp += FHW * 64) {
    int h = hp;
    const int c2 = h % FG2; h /= FG2;
    const int c1 = h % FG1; h /= FG1;
    const int c0 = h;
    int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
    if (g.pad_mode == S3_PAD_REFLECT) {
      i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
    }
    const bool ok = hp < FHP && i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2;
    const float* src = x + ((((int64_t)n * D0 + i0) * D1 + i1) * D2 + i2) * CIN;
    auto lo_f = [](unsigned u) { return __uint_as_float(u << 16); };
    auto hi_f = [](unsigned u) { return __uint_as_float(u & 0xFFFF0000u); };
    if (CIN == 2) {
      float2 t = make_float2(0.f, 0.f);
      if (ok) t = *reinterpret_cast<const float2*>(src);
      const unsigned h = pk2(t.x, t.y);
      *reinterpret_cast<unsigned*>(halo + hp * CELLB) = h;
      if constexpr (X3)
        *reinterpret_cast<unsigned*>(halo + LO_IMG + hp * CELLB) = pk2(t.x - lo_f(h), t.y - hi_f(h));
    } else {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) t = *reinterpret_cast<const float4*>(src);
      const unsigned h0 = pk2(t.x, t.y), h1 = pk2(t.z, t.w);
      *reinterpret_cast<uint2*>(halo + hp * CELLB) = make_uint2(h0, h1);
      if constexpr (X3)
        *reinterpret_cast<uint2*>(halo + LO_IMG + hp * CELLB) =
            make_uint2(pk2(t.x - lo_f(h0), t.y - hi_f(h0)), pk2(t.z - lo_f(h1), t.w - hi_f(h1)));
    }
  }
  // this lane's taps: chunk kc, slot q -> tap = (kc*32 + kq*8) / CIN + q; byte offset
  // of its cell relative to the output position's halo origin (zero cell if tap >= 27)
  int toff[KC][TPL];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
      const int tap = (kc * 32 + kq * 8) / CIN + q;
      const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
      toff[kc][q] = tap < 27 ? ((ta * FG1 + tb) * FG2 + tc) * CELLB : -1;
    }
  constexpr int NFA = LEAN ? NFP : 4;
  bf16x8 wf[KC][NFA], wl[X3 ? KC : 1][X3 ? NFA : 1];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int nf = 0; nf < NFA; ++nf)
      if (nf < nfv) {
        const int row = perm ? (nf >> 1) * 32 + (p16 >> 2) * 8 + (nf & 1) * 4 + (p16 & 3) : nf * 16 + p16;
        wf[kc][nf] = *reinterpret_cast<const bf16x8*>(
            wpk + ((int64_t)ct * GT_N + row) * KP + kc * 32 + kq * 8);
        if constexpr (X3)
          wl[kc][nf] = *reinterpret_cast<const bf16x8*>(
              wpk + lo_off + ((int64_t)ct * GT_N + row) * KP + kc * 32 + kq * 8);
      }
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  float bv[NFA][4];
#pragma unroll
  for (int nf = 0; nf < NFA; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = ct * GT_N + (perm ? (nf >> 1) * 32 + kq * 8 + (nf & 1) * 4 : nf * 16 + kq * 4) + r;
      bv[nf][r] = (bias && ch < R) ? bias[ch] : 0.f;
    }
  __syncthreads();

  if constexpr (LEAN) {
    // ---- lean walk (MODE 1: the 13.9 M-position first
    // discriminator layer: the generic loop below spent 80 % of the SIMD
    // cycles on VALU index math, 211 instructions per fragment).  A wave owns
    // the tile row r0 = wave; fragment f = (r1 = f >> 1, half = f & 1): every
    // LDS address is (per-lane register) + (immediate), the accumulators start
    // from the bias, the output offset is 32-bit relative to the wave's row.
    // K-padding slots (taps 27 ..) read tap 26's cell: finite data x zero weight.
    unsigned la[KC][TPL];
    const int pos0 = ((wave * FG1) * FG2 + p16) * CELLB;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int q = 0; q < TPL; ++q)
        la[kc][q] = (unsigned)(pos0 + (toff[kc][q] >= 0 ? toff[kc][q] : ((2 * FG1 + 2) * FG2 + 2) * CELLB));
    const int o0 = org0 + wave;
    if (o0 >= g.O[0]) return;                       // (no barrier below)
    // (element offsets; the element is 2 B in MODE 1, 4 B in MODE 2)
    const int64_t row_el = ((((int64_t)n * g.O[0] + o0) * g.O[1] + org1) * g.O[2] + org2) * R + ct * GT_N;
    unsigned short* yrow = reinterpret_cast<unsigned short*>(yv) + row_el;
    float* yrow32 = reinterpret_cast<float*>(yv) + row_el;
    const unsigned lane_off = (unsigned)(p16 * R + (PERM ? kq * 8 : kq * 4));
    const bool ok_half[2] = {org2 + p16 < g.O[2], org2 + 16 + p16 < g.O[2]};
    const int rows_ok = g.O[1] - org1;               // rows r1 < rows_ok exist
#pragma unroll 1
    for (int r1 = 0; r1 < FH1; ++r1) {
      if (r1 >= rows_ok) break;                      // wave-uniform
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int foff = half * 16 * CELLB;           // immediate
        f32x4 acc[NFP];
#pragma unroll
        for (int nf = 0; nf < NFP; ++nf)
          acc[nf] = (f32x4){bv[nf][0], bv[nf][1], bv[nf][2], bv[nf][3]};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          unsigned u[4];
#pragma unroll
          for (int q = 0; q < TPL; ++q) {
            if (CIN == 2) {
              u[q] = *reinterpret_cast<const unsigned*>(halo + la[kc][q] + foff);
            } else {
              const uint2 t = *reinterpret_cast<const uint2*>(halo + la[kc][q] + foff);
              u[2 * q] = t.x; u[2 * q + 1] = t.y;
            }
          }
          const bf16x8 xf = __builtin_bit_cast(bf16x8, make_uint4(u[0], u[1], u[2], u[3]));
          if constexpr (X3) {
            unsigned ul[4];
#pragma unroll
            for (int q = 0; q < TPL; ++q) {
              if (CIN == 2) {
                ul[q] = *reinterpret_cast<const unsigned*>(halo + LO_IMG + la[kc][q] + foff);
              } else {
                const uint2 t = *reinterpret_cast<const uint2*>(halo + LO_IMG + la[kc][q] + foff);
                ul[2 * q] = t.x; ul[2 * q + 1] = t.y;
              }
            }
            const bf16x8 xl = __builtin_bit_cast(bf16x8, make_uint4(ul[0], ul[1], ul[2], ul[3]));
#pragma unroll
            for (int nf = 0; nf < NFP; ++nf) {
              acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[kc][nf], xf, acc[nf], 0, 0, 0);
              acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc][nf], xl, acc[nf], 0, 0, 0);
            }
          }
#pragma unroll
          for (int nf = 0; nf < NFP; ++nf)
            acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc][nf], xf, acc[nf], 0, 0, 0);
        }
        if (!ok_half[half]) continue;
        const unsigned off = lane_off + (unsigned)((r1 * g.O[2] + half * 16) * R);
        if constexpr (!PERM) {
#pragma unroll
          for (int nf = 0; nf < NFP; ++nf) {
            if (ct * GT_N + nf * 16 + kq * 4 >= R) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = acc[nf][r];
              o[r] = a > 0.f ? a : slope * a;
            }
            *reinterpret_cast<float4*>(yrow32 + off + nf * 16) = make_float4(o[0], o[1], o[2], o[3]);
          }
          continue;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (2 * h >= NFP) continue;
          float o[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float a = acc[(2 * h + (q >> 2)) % NFP][q & 3];
            // slope in [0, 1]: identity / ReLU / LeakyReLU.  (v_max_f32 by hand:
            // fmaxf() first canonicalises an accumulator it cannot prove quiet —
            // a second v_max per value in a loop that is 73 % VALU-busy)
            const float sa = slope * a;
            asm("v_max_f32 %0, %1, %2" : "=v"(o[q]) : "v"(a), "v"(sa));
          }
          const uint4 w4 = make_uint4(pk2(o[0], o[1]), pk2(o[2], o[3]), pk2(o[4], o[5]), pk2(o[6], o[7]));
          *reinterpret_cast<uint4*>(yrow + off + h * 32) = w4;
          if (sign) {
            // (C_out = 32, one cout tile) bit q of byte [position][kq]: channel
            // 8 kq + q of the STORED bf16 value is > 0 — the activation mask
            // conv_dgrad_s2_kernel<.., MB> applies, at 4 B per position
            // (taken from the fp32 value: it and its bf16 rounding have the same
            // sign and are zero together down to the smallest bf16 denormal;
            // one compare + one shift-or per channel instead of the 45
            // instructions the test on the packed halves compiled to)
            unsigned bits = 0u;
#pragma unroll
            for (int q = 7; q >= 0; --q) bits = (bits << 1) | (o[q] > 0.f ? 1u : 0u);
            sign[(row_el + off) / 8] = (unsigned char)bits;    // element (pos * 32 + 8 kq) / 8 = pos * 4 + kq
          }
        }
      }
      // next tile row: every halo address moves one row of cells
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int q = 0; q < TPL; ++q) la[kc][q] += FG2 * CELLB;
    }
  } else {
  // ---- 64 fragments per tile (32 rows x 2 halves of t), 16 per wave
#pragma unroll 2
  for (int f = 0; f < 16; ++f) {
    const int fr = wave * 16 + f;
    const int row = fr >> 1, half = fr & 1;
    const int r0 = row / FH1, r1 = row % FH1;
    const int pos = ((r0 * FG1 + r1) * FG2 + half * 16 + p16) * CELLB;   // halo origin of this lane's position
    f32x4 acc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      unsigned u[4];
#pragma unroll
      for (int q = 0; q < TPL; ++q) {
        const int a = toff[kc][q] >= 0 ? pos + toff[kc][q] : FHP * CELLB;
        if (CIN == 2) {
          u[q] = *reinterpret_cast<const unsigned*>(halo + a);
        } else {
          const uint2 t = *reinterpret_cast<const uint2*>(halo + a);
          u[2 * q] = t.x; u[2 * q + 1] = t.y;
        }
      }
      const uint4 uu = make_uint4(u[0], u[1], u[2], u[3]);
      const bf16x8 xf = __builtin_bit_cast(bf16x8, uu);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
        if (nf < nfv)
          acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc][nf], xf, acc[nf], 0, 0, 0);
    }
    const int o0 = org0 + r0, o1 = org1 + r1, o2 = org2 + half * 16 + p16;
    if (o0 >= g.O[0] || o1 >= g.O[1] || o2 >= g.O[2]) continue;
    const int64_t pp = (((int64_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2;
    if (perm) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (2 * h >= nfv) continue;
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          o[q] = acc[2 * h + (q >> 2)][q & 3] + bv[2 * h + (q >> 2)][q & 3];
          o[q] = o[q] > 0.f ? o[q] : slope * o[q];
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(yv) + pp * R + ct * GT_N + h * 32 + kq * 8) =
            make_uint4(pk2(o[0], o[1]), pk2(o[2], o[3]), pk2(o[4], o[5]), pk2(o[6], o[7]));
      }
      continue;
    }
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int ch = ct * GT_N + nf * 16 + kq * 4;
      if (nf >= nfv || ch >= R) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = acc[nf][r] + bv[nf][r];
        o[r] = o[r] > 0.f ? o[r] : slope * o[r];
      }
      if ((R & 3) == 0) {
        if (out_bf16)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yv) + pp * R + ch) =
              make_uint2(pk2(o[0], o[1]), pk2(o[2], o[3]));
        else
          *reinterpret_cast<float4*>(y + pp * R + ch) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch + r < R) y[pp * R + ch + r] = o[r];
      }
    }
  }
  }
}

// the halo variant needs stride 1, no residual and enough tiles to fill the chip
bool fewch_halo_ok(const s3_ctx* ctx, const ConvGeom& g, const float* res, int out_bf16) {
  if (res || s3_opt_has(S3O_NO_FEWCH_HALO)) return false;
  if (out_bf16 && (g.Cout & 3)) return false;
  for (int d = 0; d < 3; ++d)
    if (g.s[d] != 1 || g.lo[d] < 0 || g.lo[d] > 2) return false;
  const int64_t tiles = (int64_t)g.N * ((g.O[0] + FH0 - 1) / FH0) * ((g.O[1] + FH1 - 1) / FH1) *
                        ((g.O[2] + FH2 - 1) / FH2);
  const int64_t min_tiles = s3_opt_has(S3O_FEWCH_HALO_MIN_TILES)
                                ? s3_opt_int(S3O_FEWCH_HALO_MIN_TILES, 0) : ctx->num_cu;
  // (one tile per CU already beats the gather variant: the generator's 4 -> 64
  // head conv at C2 batch 32 is 256 tiles — 59 us on the gather walk)
  return tiles >= min_tiles;
}

// fp32 [27][cin][cout] -> bf16 [rows_pad][KP], k = tap * cin + ci
__global__ void gconv_fewch_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                        int cin, int cout, int rows_pad, int kp,
                                        unsigned short* __restrict__ lo = nullptr) {
  const int total = rows_pad * kp;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx % kp, row = idx / kp;
    float v = 0.f;
    if (row < cout && k < 27 * cin) v = w[(int64_t)k * cout + row];
    const unsigned h = pk2(v, 0.f) & 0xFFFFu;
    out[idx] = (unsigned short)h;
    if (lo) lo[idx] = (unsigned short)(pk2(v - __uint_as_float(h << 16), 0.f) & 0xFFFFu);
  }
}

bool fewch_geom(const ConvGeom& g) {
  return (g.Cin == 2 || g.Cin == 4) && g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3 &&
         !s3_opt_has(S3O_NO_FEWCH);
}

}  // namespace

// launch_gconv_fwd writes sign bytes (4 B per position: bit q of byte
// [position][kq] = channel 8 kq + q of the stored output is > 0) for this
// geometry when it is handed a buffer: the permuted lean walk, C_out = 32
bool conv_gconv_writes_sign_bytes(const s3_ctx* ctx, const ConvGeom& g, int out_bf16) {
  return fewch_geom(g) && g.Cout == 32 && fewch_halo_ok(ctx, g, nullptr, out_bf16) && fewch_halo_perm(g, out_bf16);
}

bool conv_gconv_supported(const ConvGeom& g, int precision) {
  if (precision == S3_PREC_BF16X3 ? s3_opt_has(S3O_NO_GCONV_X3) : precision != S3_PREC_BF16) return false;
  if (s3_opt_has(S3O_NO_GCONV)) return false;
  if (g.d2s != 1) return false;
  // C_in = 4: the generator's first conv (a cell is one float4)
  // C_in = 2: hi-res fields into the discriminator (a cell is one float2)
  if (!(g.Cin == 2 || g.Cin == 4 || (g.Cin % 8 == 0 && g.Cin >= 32))) return false;
  return true;
}

// data gradient through the same kernel: contraction over C_out
bool conv_gconv_dgrad_supported(const ConvGeom& g, int precision) {
  if (precision == S3_PREC_BF16X3 ? s3_opt_has(S3O_NO_GCONV_X3) : precision != S3_PREC_BF16) return false;
  if (s3_opt_has(S3O_NO_GCONV)) return false;
  // (a depth-to-space store is undone by the epilogue adjoint: dPre arrives in
  // the conv's own output layout)
  // reflect padding: stride-1 'same' frame + fold only
  if (g.pad_mode == S3_PAD_REFLECT)
    for (int d = 0; d < 3; ++d)
      if (g.s[d] != 1) return false;
  return g.Cout % 8 == 0 && g.Cout >= 32;
}

static int rows_padded(int r) { return (r + GT_N - 1) / GT_N * GT_N; }

// element offset of the residue (lo) image of a BF16X3 plan behind the hi image
static int64_t x3_lo_offset(const ConvGeom& g, int dgrad) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int R = dgrad ? g.Cin : g.Cout, K = dgrad ? g.Cout : g.Cin;
  return (int64_t)taps * rows_padded(R) * ((K + 7) / 8 * 8) + 32;
}

static int x3_fewch_lo_offset(const ConvGeom& g) { return rows_padded(g.Cout) * ((27 * g.Cin + 31) / 32 * 32); }

size_t conv_gconv_packed_bytes(const ConvGeom& g, int dgrad, int x3) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int R = dgrad ? g.Cin : g.Cout, K = dgrad ? g.Cout : g.Cin;
  // (few-channel geometry: the taps-in-K images only, hi then lo)
  if (x3 && !dgrad && fewch_geom(g)) return 2 * (size_t)x3_fewch_lo_offset(g) * 2 + 64;
  if (x3) return 2 * ((size_t)x3_lo_offset(g, dgrad) * 2);
  size_t fewch = 0;
  if (!dgrad && fewch_geom(g)) fewch = (size_t)rows_padded(g.Cout) * ((27 * g.Cin + 31) / 32 * 32) * 2;
  return (size_t)taps * rows_padded(R) * ((K + 7) / 8 * 8) * 2 + 64 + fewch;   // + over-read of a masked K tail
}

// the taps-in-K image of the few-channel kernel sits behind the per-tap image
static size_t fewch_image_offset(const ConvGeom& g) {
  const int taps = g.k[0] * g.k[1] * g.k[2];
  return (size_t)taps * rows_padded(g.Cout) * ((g.Cin + 7) / 8 * 8) * 2 + 64;
}

int launch_gconv_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* packed, int dgrad, int x3) {
  if (x3 && !dgrad && fewch_geom(g)) {
    const int kp = (27 * g.Cin + 31) / 32 * 32;
    hipLaunchKernelGGL(gconv_fewch_pack_kernel, dim3((rows_padded(g.Cout) * kp + 255) / 256), dim3(256), 0,
                       ctx->stream, w, (unsigned short*)packed, g.Cin, g.Cout, rows_padded(g.Cout), kp,
                       (unsigned short*)packed + x3_fewch_lo_offset(g));
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int R = dgrad ? g.Cin : g.Cout, K = dgrad ? g.Cout : g.Cin;
  const int64_t total = (int64_t)taps * rows_padded(R) * ((K + 7) / 8 * 8);
  int grid = (int)((total + 32 + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gconv_pack_kernel, dim3(grid), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)packed, taps, g.Cin, g.Cout, rows_padded(R), dgrad,
                     x3 ? (unsigned short*)packed + x3_lo_offset(g, dgrad) : (unsigned short*)nullptr);
  S3_HIP(ctx, hipGetLastError());
  if (!dgrad && fewch_geom(g) && !x3) {
    const int kp = (27 * g.Cin + 31) / 32 * 32;
    hipLaunchKernelGGL(gconv_fewch_pack_kernel, dim3((rows_padded(g.Cout) * kp + 255) / 256), dim3(256), 0,
                       ctx->stream, w, (unsigned short*)((char*)packed + fewch_image_offset(g)), g.Cin,
                       g.Cout, rows_padded(g.Cout), kp);
    S3_HIP(ctx, hipGetLastError());
  }
  return S3_OK;
}

int launch_gconv_fwd(s3_ctx* ctx, const ConvGeom& g, const float* x, const void* packed,
                     const float* bias, const float* res, void* y, int out_bf16, int in_bf16, int x3,
                     void* sign_bytes) {
  if (out_bf16 && g.Cout % 4 != 0) S3_FAIL(ctx, S3_EINVAL, "gconv: bf16 output needs C_out % 4 == 0");
  if (in_bf16 && (g.Cin % 8 != 0 || fewch_geom(g))) S3_FAIL(ctx, S3_EINVAL, "gconv: bf16 input needs C_in % 8 == 0");
  if (x3 && (out_bf16 || in_bf16)) S3_FAIL(ctx, S3_EINVAL, "gconv: the split-bf16 variant takes and writes fp32");
  const int64_t P = (int64_t)g.N * g.O[0] * g.O[1] * g.O[2];
  if (fewch_geom(g) && x3 && !res && !out_bf16 && (g.Cout & 3) == 0 && g.Cout <= GT_N &&
      fewch_halo_ok(ctx, g, res, 0) && !s3_opt_has(S3O_NO_GCONV_X3)) {
    // the LDS-halo lean walk with split operands (natural rows, fp32 out)
    const int t0 = (g.O[0] + FH0 - 1) / FH0, t1 = (g.O[1] + FH1 - 1) / FH1, t2 = (g.O[2] + FH2 - 1) / FH2;
    dim3 hgrid((unsigned)(g.N * t0 * t1 * t2), 1);
    const int nf = (g.Cout + 15) / 16;      // (3 fragments run as 4: rows past C_out are zero)
    auto kern = g.Cin == 2 ? (nf == 1 ? gconv_fewch_halo_kernel<2, 2, 1, true>
                                      : (nf == 2 ? gconv_fewch_halo_kernel<2, 2, 2, true>
                                                 : gconv_fewch_halo_kernel<2, 2, 4, true>))
                           : (nf == 1 ? gconv_fewch_halo_kernel<4, 2, 1, true>
                                      : (nf == 2 ? gconv_fewch_halo_kernel<4, 2, 2, true>
                                                 : gconv_fewch_halo_kernel<4, 2, 4, true>));
    hipLaunchKernelGGL(kern, hgrid, dim3(FHW * 64), 0, ctx->stream, x, (const unsigned short*)packed, bias, y, g,
                       t0, t1, t2, 0, (unsigned char*)nullptr, x3_fewch_lo_offset(g));
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (fewch_geom(g) && x3) {
    dim3 fgrid((unsigned)((P + FC_POS - 1) / FC_POS), (unsigned)((g.Cout + GT_N - 1) / GT_N));
    if (g.Cin == 2)
      hipLaunchKernelGGL((gconv_fewch_kernel<2, true>), fgrid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                         (const unsigned short*)packed, bias, res, y, g, P, 0, x3_fewch_lo_offset(g));
    else
      hipLaunchKernelGGL((gconv_fewch_kernel<4, true>), fgrid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                         (const unsigned short*)packed, bias, res, y, g, P, 0, x3_fewch_lo_offset(g));
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (fewch_geom(g) && !x3) {
    dim3 fgrid((unsigned)((P + FC_POS - 1) / FC_POS), (unsigned)((g.Cout + GT_N - 1) / GT_N));
    const unsigned short* img = (const unsigned short*)((const char*)packed + fewch_image_offset(g));
    if (fewch_halo_ok(ctx, g, res, out_bf16)) {
      const int t0 = (g.O[0] + FH0 - 1) / FH0, t1 = (g.O[1] + FH1 - 1) / FH1, t2 = (g.O[2] + FH2 - 1) / FH2;
      dim3 hgrid((unsigned)(g.N * t0 * t1 * t2), (unsigned)((g.Cout + GT_N - 1) / GT_N));
      const bool pm = fewch_halo_perm(g, out_bf16);
      const bool two = g.Cout == 32;
      // fp32 out, C_out % 4 == 0 and one cout tile: the lean natural-row walk
      const int nf32 = (!out_bf16 && (g.Cout & 3) == 0 && g.Cout <= GT_N) ? (g.Cout + 15) / 16 : 0;
      auto kern = g.Cin == 2 ? (pm ? (two ? gconv_fewch_halo_kernel<2, 1, 2> : gconv_fewch_halo_kernel<2, 1, 4>)
                                   : gconv_fewch_halo_kernel<2, 0>)
                             : (pm ? (two ? gconv_fewch_halo_kernel<4, 1, 2> : gconv_fewch_halo_kernel<4, 1, 4>)
                                   : gconv_fewch_halo_kernel<4, 0>);
      if (!pm && nf32 == 1) kern = g.Cin == 2 ? gconv_fewch_halo_kernel<2, 2, 1> : gconv_fewch_halo_kernel<4, 2, 1>;
      else if (!pm && nf32 == 2) kern = g.Cin == 2 ? gconv_fewch_halo_kernel<2, 2, 2> : gconv_fewch_halo_kernel<4, 2, 2>;
      else if (!pm && nf32 >= 3) kern = g.Cin == 2 ? gconv_fewch_halo_kernel<2, 2, 4> : gconv_fewch_halo_kernel<4, 2, 4>;
      // sign bytes next to y: the permuted lean walk with C_out = 32 only
      unsigned char* sb = (pm && two && g.Cout == 32) ? (unsigned char*)sign_bytes : nullptr;
      hipLaunchKernelGGL(kern, hgrid, dim3(FHW * 64), 0, ctx->stream, x, img, bias, y, g, t0, t1, t2, out_bf16, sb, 0);
      S3_HIP(ctx, hipGetLastError());
      return S3_OK;
    }
    if (g.Cin == 2)
      hipLaunchKernelGGL(gconv_fewch_kernel<2>, fgrid, dim3(GT_WAVES * 64), 0, ctx->stream, x, img,
                         bias, res, y, g, P, out_bf16);
    else
      hipLaunchKernelGGL(gconv_fewch_kernel<4>, fgrid, dim3(GT_WAVES * 64), 0, ctx->stream, x, img,
                         bias, res, y, g, P, out_bf16);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  const int n_ct = (g.Cout + GT_N - 1) / GT_N;
  // four position fragments per wave when that still gives >= 2 workgroups per CU
  const bool wide = (P / (GT_WAVES * 4 * 16)) * n_ct >= 2 * (int64_t)ctx->num_cu && !s3_opt_has(S3O_GCONV_MF2);
  const int pos = GT_WAVES * (wide ? 4 : 2) * 16;
  const int nblk = (int)((P + pos - 1) / pos);
  const int iters = g.k[0] * g.k[1] * g.k[2] * ((g.Cin + 31) / 32);
  int ns = gconv_splits(ctx, (int64_t)nblk * n_ct, iters);
  if (ns > 1 && ensure_scratch(ctx, (size_t)ns * P * g.Cout * sizeof(float)) != S3_OK) ns = 1;
  dim3 grid((unsigned)nblk, (unsigned)n_ct, (unsigned)ns);
  if (x3 && wide)
    hipLaunchKernelGGL((gconv_mfma_kernel<false, 4, true>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                       (const unsigned short*)packed, bias, res, y, g, P, rows_padded(g.Cout), 0, 0, 0, 0,
                       ns, ctx->scratch, x3_lo_offset(g, 0));
  else if (x3)
    hipLaunchKernelGGL((gconv_mfma_kernel<false, 2, true>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                       (const unsigned short*)packed, bias, res, y, g, P, rows_padded(g.Cout), 0, 0, 0, 0,
                       ns, ctx->scratch, x3_lo_offset(g, 0));
  else if (wide)
    hipLaunchKernelGGL((gconv_mfma_kernel<false, 4>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                       (const unsigned short*)packed, bias, res, y, g, P, rows_padded(g.Cout), 0, 0, out_bf16, in_bf16,
                       ns, ctx->scratch);
  else
    hipLaunchKernelGGL((gconv_mfma_kernel<false, 2>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, x,
                       (const unsigned short*)packed, bias, res, y, g, P, rows_padded(g.Cout), 0, 0, out_bf16, in_bf16,
                       ns, ctx->scratch);
  S3_HIP(ctx, hipGetLastError());
  if (ns > 1) {
    ++ctx->stat[S3_STAT_GCONV_SPLITK];
    const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
    const int64_t total = P * g.Cout;
    int eg = (int)((total + 255) / 256);
    if (eg > 2048) eg = 2048;
    hipLaunchKernelGGL(gconv_splitk_epilogue, dim3(eg), dim3(256), 0, ctx->stream, (const float*)ctx->scratch, ns,
                       P, g.Cout, bias, res, y, 1, slope, out_bf16, 0);
    S3_HIP(ctx, hipGetLastError());
  }
  return S3_OK;
}

int launch_gconv_dgrad(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* packed_t,
                       float* dx, int accumulate, int frame, int dy_bf16, int x3) {
  if (x3 && dy_bf16) S3_FAIL(ctx, S3_EINVAL, "gconv: the split-bf16 variant takes fp32 dPre");
  // dy_bf16: dy is the bf16 copy of dPre (C_out % 8 == 0)
  const int64_t P = frame ? (int64_t)g.N * (g.D[0] + 2 * g.lo[0]) * (g.D[1] + 2 * g.lo[1]) *
                                (g.D[2] + 2 * g.lo[2])
                          : (int64_t)g.N * g.D[0] * g.D[1] * g.D[2];
  const int n_ct = (g.Cin + GT_N - 1) / GT_N;
  int64_t pw = P;        // positions one grid slice walks
  int nz = 1;
  if (g.s[0] > 1 || g.s[1] > 1 || g.s[2] > 1) {
    // one grid slice per residue class, sized for the largest class (residue 0)
    pw = g.N;
    for (int d = 0; d < 3; ++d) pw *= (g.D[d] + g.s[d] - 1) / g.s[d];
    nz = g.s[0] * g.s[1] * g.s[2];
  }
  // (measured per layer at C2 batch 8: four fragments per wave pay off in the
  // forward gathers only — the 64 -> 64 stride-2 data gradient ran 312 vs
  // 288 us — so the adjoint takes them on request, SUP3R_AMD_GCONV_MF4=1)
  const bool wide = (pw / (GT_WAVES * 4 * 16)) * n_ct * nz >= 2 * (int64_t)ctx->num_cu && !s3_opt_has(S3O_GCONV_MF2) &&
                    s3_opt_has(S3O_GCONV_MF4);
  const int pos = GT_WAVES * (wide ? 4 : 2) * 16;
  const int nblk = (int)((pw + pos - 1) / pos);
  // (stride-1 only: a residue class of a strided conv sees 1 - 8 taps)
  int ns = nz == 1 ? gconv_splits(ctx, (int64_t)nblk * n_ct, g.k[0] * g.k[1] * g.k[2] * ((g.Cout + 31) / 32)) : 1;
  if (ns > 1 && ensure_scratch(ctx, (size_t)ns * P * g.Cin * sizeof(float)) != S3_OK) ns = 1;
  dim3 grid((unsigned)nblk, (unsigned)n_ct, (unsigned)(nz * ns));
  if (x3)
    hipLaunchKernelGGL((gconv_mfma_kernel<true, 2, true>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, dy,
                       (const unsigned short*)packed_t, nullptr, nullptr, dx, g, P,
                       rows_padded(g.Cin), accumulate, frame, 0, 0, ns, ctx->scratch, x3_lo_offset(g, 1));
  else if (wide)
    hipLaunchKernelGGL((gconv_mfma_kernel<true, 4>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, dy,
                       (const unsigned short*)packed_t, nullptr, nullptr, dx, g, P,
                       rows_padded(g.Cin), accumulate, frame, 0, dy_bf16, ns, ctx->scratch);
  else
    hipLaunchKernelGGL((gconv_mfma_kernel<true, 2>), grid, dim3(GT_WAVES * 64), 0, ctx->stream, dy,
                       (const unsigned short*)packed_t, nullptr, nullptr, dx, g, P,
                       rows_padded(g.Cin), accumulate, frame, 0, dy_bf16, ns, ctx->scratch);
  S3_HIP(ctx, hipGetLastError());
  if (ns > 1) {
    ++ctx->stat[S3_STAT_GCONV_SPLITK];
    const int64_t total = P * g.Cin;
    int eg = (int)((total + 255) / 256);
    if (eg > 2048) eg = 2048;
    hipLaunchKernelGGL(gconv_splitk_epilogue, dim3(eg), dim3(256), 0, ctx->stream, (const float*)ctx->scratch, ns,
                       P, g.Cin, nullptr, nullptr, dx, 0, 1.f, 0, accumulate);
    S3_HIP(ctx, hipGetLastError());
  }
  return S3_OK;
}
