// Data gradient of a stride-2, valid-padded conv with 32 output channels
// (discriminator 32 -> 32 s2 over 13.9 M input positions) on bf16 MFMA with an
// LDS halo — S3_PREC_BF16 training plans.
//
//   dx[i][ci] = sum_{tap : (i - tap) even per axis} sum_co W[tap][ci][co] dPre[(i - tap) / 2][co]
//
// With i = 2 u + p per axis the taps split by the parity class p: p = 0 gets the
// taps {0, 2} (dPre cells u, u - 1), p = 1 the tap {1} (cell u), so each of the
// 8 classes is a small stride-1 correlation over the dPre grid (8, 4, 4, 2, 4,
// 2, 2, 1 taps = 27 in total).  A workgroup owns a 4 x 8 x 16 tile of u (= 8 x
// 16 x 32 positions of x), stages the (4+1) x (8+1) x (16+1) dPre halo once
// (fp32 -> bf16, 64-B cells, the (t >> 1) & 3 chunk swizzle of
// conv_dgrad_c2_kernel) and walks the classes: per class the taps' filter rows
// (A operand, L1-resident packed image) against the shifted halo window (B
// operand), 8 rows x 2 channel fragments of accumulators per wave, stored to
// x's grid at stride 2.  The gather kernel iterates the same taps per residue
// class but re-reads every dPre cell through L1 (1.31 ms at C2 batch 8).
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int ST0 = 4, ST1 = 8, ST2 = 16;
constexpr int SG0 = ST0 + 1, SG1 = ST1 + 1, SG2 = ST2 + 1;
constexpr int SHP = SG0 * SG1 * SG2;          // 765 halo cells
constexpr int SNW = 4;
constexpr int SNT = SNW * 64;
constexpr int SLDS = SHP * 64;                // 48,960 B

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

// fp32 w[tap][cin][32] -> bf16 img[tap][cin_pad][32]  (rows = ci, K = co)
__global__ void dgrad_s2_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                     int cin, int rows_pad) {
  const int total = 27 * rows_pad * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int co = idx & 31, row = (idx >> 5) % rows_pad, tp = idx / (32 * rows_pad);
    const float v = row < cin ? w[((size_t)tp * cin + row) * 32 + co] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

// O16 (NF = 2, C_in = 32): dx is stored as bf16 ONLY — it is dPre of the conv
// below (mask fused, single consumer) whose gradient kernels take bf16.  The
// filter rows are taken in the order (kg, nf, r) so that a lane's two C/D
// fragments are 8 CONSECUTIVE channels 8 kg .. 8 kg + 7: one 16-B store (and
// one 16-B mask read) per position and lane, a whole 64-B row per position
// across the k-groups.  A lane owns the same 8 channels throughout, so their
// sums — the bias gradient of that conv — accumulate in registers and leave as
// one partial row per workgroup (bsum[block][32], for bias_grad_stage2).
// MB (with O16): the activation mask comes as sign BYTES — bit q of byte
// [position][kg] = "channel 8 kg + q of y is > 0", written by the forward
// kernel next to y (gconv_fewch_halo_kernel) — instead of the bf16 tensor y
// itself: 4 B per position instead of 64.  The 64-B rows of y at the 128-B
// pitch of a parity class were fetched 1.7 times each (the other half of a
// line is wanted one class later: FETCH 1.51 GB for 0.89 GB of mask,
// profiles/r03/README.md).
template <int NF, bool O16 = false, bool MB = false>
__global__ __launch_bounds__(SNT) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_dgrad_s2_kernel(
    const float* __restrict__ dy, const unsigned short* __restrict__ img,
    float* __restrict__ dx, ConvGeom g, int rows_pad, int tiles0, int tiles1, int tiles2,
    const void* __restrict__ mask_y, float mask_slope, int mask_bf16, float* __restrict__ bsum) {
  static_assert(!MB || O16, "sign bytes: the bf16-only output path");
  extern __shared__ __attribute__((aligned(16))) char halo[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int ct = blockIdx.y;
  int tr = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int u0 = t0i * ST0, u1 = t1i * ST1, u2 = t2i * ST2;   // tile origin on the u grid
  const int O0 = g.O[0], O1 = g.O[1], O2 = g.O[2];

  // ---- stage the dPre halo: cell (c0, c1, c2) = dPre[u + c - 1], zero outside
  for (int base = tid; base < SHP * 4; base += SNT * 3) {
    float4 va[3], vb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * SNT;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vb[u] = va[u];
      if (item < SHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        int h = hp;
        const int c2 = h % SG2; h /= SG2;
        const int c1 = h % SG1; h /= SG1;
        const int c0 = h;
        const int i0 = u0 + c0 - 1, i1 = u1 + c1 - 1, i2 = u2 + c2 - 1;
        if (i0 >= 0 && i0 < O0 && i1 >= 0 && i1 < O1 && i2 >= 0 && i2 < O2) {
          const float* src = dy + ((((size_t)n * O0 + i0) * O1 + i1) * O2 + i2) * 32 + ch * 8;
          va[u] = *reinterpret_cast<const float4*>(src);
          vb[u] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int item = base + u * SNT;
      if (item < SHP * 4) {
        const int hp = item >> 2, ch = item & 3;
        const int key = ((hp % SG2) >> 1) & 3;
        *reinterpret_cast<uint4*>(halo + hp * 64 + ((ch ^ key) << 4)) =
            make_uint4(pk2(va[u].x, va[u].y), pk2(va[u].z, va[u].w), pk2(vb[u].x, vb[u].y),
                       pk2(vb[u].z, vb[u].w));
      }
    }
  }
  __syncthreads();

  // B-operand offsets of this lane for the two t shifts: halo t index = j + 1 - d, d in {0, 1}
  int off_d[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int th = j + 1 - d;
    off_d[d] = th * 64 + ((kg ^ ((th >> 1) & 3)) << 4);
  }
  // (O16: row (nf, j) of the A operand = channel 8 (j >> 2) + 4 nf + (j & 3))
  const unsigned short* wrow = img + ((size_t)ct * 64 + (O16 ? (j >> 2) * 8 + (j & 3) : j)) * 32 + kg * 8;
  const int R = g.Cin;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // O16: channel sums of what this lane stores
  // O16 with a bf16 mask source (the C2 case): the eight 16-B mask reads of a
  // class do not depend on its MFMAs but sat behind them — waves were parked
  // in s_waitcnt 79 % of the time (profiles/r03/pmc_train_start.txt).  They
  // are issued one class AHEAD now, under the tap loop of the class before.
  const bool mpre = O16 && !MB && mask_y && mask_bf16;
  uint4 mk[MB ? 1 : 8], mk_next[MB ? 1 : 8];
  unsigned mb[MB ? 8 : 1], mb_next[MB ? 8 : 1];        // MB: one sign byte per (position, kg)
  auto mask_fetch = [&](int cls_, uint4* dst8) __attribute__((always_inline)) {
    const int q0 = cls_ >> 2, q1 = (cls_ >> 1) & 1, q2 = cls_ & 1;
    const int i0 = 2 * (u0 + wave) + q0, i2 = 2 * (u2 + j) + q2;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int i1 = 2 * (u1 + m) + q1;
      if (i0 < g.D[0] && i1 < g.D[1] && i2 < g.D[2]) {
        const size_t pos = (((size_t)n * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + i2;
        if constexpr (MB)
          reinterpret_cast<unsigned*>(dst8)[m] = reinterpret_cast<const unsigned char*>(mask_y)[pos * 4 + kg];
        else
          dst8[m] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(mask_y) + pos * 32 + kg * 8);
      } else {
        if constexpr (MB) reinterpret_cast<unsigned*>(dst8)[m] = 0u;
        else dst8[m] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  if (mpre) mask_fetch(0, mk);
  if (MB) mask_fetch(0, reinterpret_cast<uint4*>(mb));
  // O16: the two t-parity classes of a (p0, p1) pair are adjacent positions of
  // x — 64 B each.  Stored class by class they left every 128-B line half
  // written for the length of a class (the tile's 262 KB of output per
  // workgroup do not stay in L2 that long); the even class is now held packed
  // (8 x 16 B per lane) and goes out together with the odd one: both halves of
  // a line within one pair of store instructions.
  // (with the sign-byte mask only: the bf16-mask variant has no registers left)
  constexpr bool PAIR = O16 && MB;
  uint4 keep[PAIR ? 8 : 1];
  // all 16 position pairs of the tile's rows exist (then i2 < D2 for every lane and the
  // pair stores can be exchanged across the wave: see the odd class below)
  const bool lin_t = PAIR && 2 * (u2 + 15) + 1 < g.D[2];
  // wave w owns the u rows (r0 = w, r1 = 0..7)
#pragma unroll 1
  for (int cls = 0; cls < 8; ++cls) {
    const int p0 = cls >> 2, p1 = (cls >> 1) & 1, p2 = cls & 1;
    const int n0 = p0 ? 1 : 2, n1 = p1 ? 1 : 2, n2 = p2 ? 1 : 2;   // taps per axis
    if (mpre && cls < 7) mask_fetch(cls + 1, mk_next);
    if (MB && cls < 7) mask_fetch(cls + 1, reinterpret_cast<uint4*>(mb_next));
    f32x4 acc[8][NF];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < n0; ++a)
      for (int b = 0; b < n1; ++b)
        for (int c = 0; c < n2; ++c) {
          // tap index per axis: parity 1 -> tap 1 (shift 0); parity 0 -> taps 0 / 2 (shift 0 / 1)
          const int ta = p0 ? 1 : 2 * a, tb = p1 ? 1 : 2 * b, tc = p2 ? 1 : 2 * c;
          const int d0 = p0 ? 0 : a, d1 = p1 ? 0 : b, d2 = p2 ? 0 : c;
          const int tp = (ta * 3 + tb) * 3 + tc;
          bf16x8 afr[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            afr[nf] = *reinterpret_cast<const bf16x8*>(wrow + ((size_t)tp * rows_pad + (O16 ? nf * 4 : nf * 16)) * 32);
          const char* hb = halo + (((wave + 1 - d0) * SG1 + (1 - d1)) * SG2) * 64 + off_d[d2];
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(hb + m * SG2 * 64);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
              acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[nf], bfr, acc[m][nf], 0, 0, 0);
          }
        }
    // ---- store the class: x position i = 2 u + p
    const int i0 = 2 * (u0 + wave) + p0, i2 = 2 * (u2 + j) + p2;
    if constexpr (O16) {
      // (no FMA contraction here: the masked value is stored AND summed — with
      // v * mask folded into the channel sum's add in one template variant and
      // not in the other, the bias gradient differed in its last bits between
      // the mask sources)
#pragma clang fp contract(off)
      unsigned short* dx16 = reinterpret_cast<unsigned short*>(dx);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int i1 = 2 * (u1 + m) + p1;
        if (i0 >= g.D[0] || i1 >= g.D[1] || i2 >= g.D[2]) continue;
        const size_t e = ((((size_t)n * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + i2) * 32 + kg * 8;
        float v[8];
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) v[q8] = acc[m][q8 >> 2][q8 & 3];
        if constexpr (MB) {
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) v[q8] *= ((mb[m] >> q8) & 1u) ? 1.f : mask_slope;
        } else if (mask_y) {
          float yv[8];
          if (mask_bf16) {
            const uint4 h = mpre ? mk[m] : *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(mask_y) + e);
            yv[0] = __uint_as_float(h.x << 16); yv[1] = __uint_as_float(h.x & 0xFFFF0000u);
            yv[2] = __uint_as_float(h.y << 16); yv[3] = __uint_as_float(h.y & 0xFFFF0000u);
            yv[4] = __uint_as_float(h.z << 16); yv[5] = __uint_as_float(h.z & 0xFFFF0000u);
            yv[6] = __uint_as_float(h.w << 16); yv[7] = __uint_as_float(h.w & 0xFFFF0000u);
          } else {
            const float4 a4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + e);
            const float4 b4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + e + 4);
            yv[0] = a4.x; yv[1] = a4.y; yv[2] = a4.z; yv[3] = a4.w;
            yv[4] = b4.x; yv[5] = b4.y; yv[6] = b4.z; yv[7] = b4.w;
          }
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) v[q8] *= yv[q8] > 0.f ? 1.f : mask_slope;
        }
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) csum[q8] += v[q8];
        const uint4 pk = make_uint4(pk2(v[0], v[1]), pk2(v[2], v[3]), pk2(v[4], v[5]), pk2(v[6], v[7]));
        if constexpr (!PAIR) {
          *reinterpret_cast<uint4*>(dx16 + e) = pk;
        } else if (p2 == 0) {
          // (the last even position of an odd extent has no partner)
          if (i2 + 1 < g.D[2]) keep[m] = pk;
          else *reinterpret_cast<uint4*>(dx16 + e) = pk;
        } else if (!lin_t) {
          *reinterpret_cast<uint4*>(dx16 + e - 32) = keep[m];
          *reinterpret_cast<uint4*>(dx16 + e) = pk;
        } else {
          // Round 6: the 16 position pairs of this row are 2 KB of consecutive bytes
          // ([even 64 B][odd 64 B] x 16).  Stored from where the values sit, an instruction
          // wrote sixteen 64-B pieces at a 128-B pitch; exchanged across the wave first
          // (lane L of instruction A takes 16-B piece L of the first KB: pair L >> 3, half
          // (L >> 2) & 1, chunk L & 3, i.e. from lane (L & 3) 16 + (L >> 3); instruction B
          // the pairs 8 .. 15), each instruction writes ONE KB in lane order: 385 -> 30x us
          // at C2 batch 8 (an ablation with made-up linear addresses had shown 302).
          const int srcA = (((lane & 3) << 4) + (lane >> 3)) << 2, srcB = srcA + (8 << 2);
          const bool odd_half = (lane >> 2) & 1;
          const unsigned ke[4] = {keep[m].x, keep[m].y, keep[m].z, keep[m].w};
          const unsigned po[4] = {pk.x, pk.y, pk.z, pk.w};
          unsigned oa[4], ob[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const unsigned ea = (unsigned)__builtin_amdgcn_ds_bpermute(srcA, (int)ke[q4]);
            const unsigned da = (unsigned)__builtin_amdgcn_ds_bpermute(srcA, (int)po[q4]);
            const unsigned eb = (unsigned)__builtin_amdgcn_ds_bpermute(srcB, (int)ke[q4]);
            const unsigned db = (unsigned)__builtin_amdgcn_ds_bpermute(srcB, (int)po[q4]);
            oa[q4] = odd_half ? da : ea;
            ob[q4] = odd_half ? db : eb;
          }
          // (every lane of the wave is here: the row test above is wave-uniform in this mode)
          const size_t e0 = ((((size_t)n * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + 2 * u2) * 32 + lane * 8;
          *reinterpret_cast<uint4*>(dx16 + e0) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
          *reinterpret_cast<uint4*>(dx16 + e0 + 512) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
        }
      }
      if (mpre) {
#pragma unroll
        for (int m = 0; m < 8; ++m) mk[m] = mk_next[m];
      }
      if constexpr (MB) {
#pragma unroll
        for (int m = 0; m < 8; ++m) mb[m] = mb_next[m];
      }
      continue;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int ch = ct * 64 + nf * 16 + kg * 4;
      if (ch >= R) continue;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int i1 = 2 * (u1 + m) + p1;
        if (i0 >= g.D[0] || i1 >= g.D[1] || i2 >= g.D[2]) continue;
        const size_t e = ((((size_t)n * g.D[0] + i0) * g.D[1] + i1) * g.D[2] + i2) * R + ch;
        float4 v = make_float4(acc[m][nf][0], acc[m][nf][1], acc[m][nf][2], acc[m][nf][3]);
        if (mask_y) {
          // fused activation adjoint of the PRODUCER of x (x = act(pre)): dPre = dx * act'(x)
          float4 yv;
          if (mask_bf16) {
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(mask_y) + e);
            yv = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xFFFF0000u),
                             __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xFFFF0000u));
          } else {
            yv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(mask_y) + e);
          }
          v.x *= yv.x > 0.f ? 1.f : mask_slope; v.y *= yv.y > 0.f ? 1.f : mask_slope;
          v.z *= yv.z > 0.f ? 1.f : mask_slope; v.w *= yv.w > 0.f ? 1.f : mask_slope;
        }
        *reinterpret_cast<float4*>(dx + e) = v;
      }
    }
  }
  if constexpr (O16) {
    if (bsum) {
      // lanes (j, kg) of all four waves -> channel 8 kg + q8: fixed-order sum in LDS
      __syncthreads();                       // every wave is done with the halo
      float* red = reinterpret_cast<float*>(halo);      // [256 threads][8]
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) red[tid * 8 + q8] = csum[q8];
      __syncthreads();
      if (tid < 32) {
        const int kq = tid >> 3, q8 = tid & 7;
        float t = 0.f;
        for (int w = 0; w < SNW; ++w)
          for (int jj = 0; jj < 16; ++jj) t += red[((w * 64) + kq * 16 + jj) * 8 + q8];
        bsum[(size_t)blockIdx.x * 32 + tid] = t;
      }
    }
  }
}

int s2_rows_pad(int cin) { return (cin + 63) / 64 * 64; }

}  // namespace

bool conv_dgrad_s2_supported(const s3_ctx* ctx, const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_DGRAD_S2)) return false;
  if (g.Cout != 32 || g.Cin % 16 != 0 || g.Cin < 16 || g.d2s != 1) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.k[d] != 3 || g.s[d] != 2 || g.lo[d] != 0) return false;
    // valid padding — or TF 'same' on an even extent (lo = 0, one zero cell past
    // the end): every x position i < D has its sources inside or zero; u grid covers D
    if ((g.O[d] - 1) * 2 + 3 > g.D[d] + 1) return false;
  }
  const int64_t min_tiles = s3_opt_has(S3O_DGRAD_S2_MIN_TILES) ? s3_opt_int(S3O_DGRAD_S2_MIN_TILES, 0)
                                                                   : ctx->num_cu;
  int64_t tiles = g.N;
  const int T[3] = {ST0, ST1, ST2};
  for (int d = 0; d < 3; ++d) tiles *= ((g.D[d] + 1) / 2 + T[d] - 1) / T[d];
  return tiles >= min_tiles;
}

size_t conv_dgrad_s2_packed_bytes(const ConvGeom& g) { return (size_t)27 * s2_rows_pad(g.Cin) * 32 * 2; }

int launch_conv_dgrad_s2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  const int rp = s2_rows_pad(g.Cin);
  hipLaunchKernelGGL(dgrad_s2_pack_kernel, dim3((27 * rp * 32 + 255) / 256), dim3(256), 0, ctx->stream, w,
                     (unsigned short*)img, g.Cin, rp);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// dx as bf16 only (+ per-workgroup channel sums): C_in = 32, one cout tile
bool conv_dgrad_s2_out16_ok(const ConvGeom& g) { return g.Cin == 32; }
int conv_dgrad_s2_blocks(const ConvGeom& g) {
  const int U0 = (g.D[0] + 1) / 2, U1 = (g.D[1] + 1) / 2, U2 = (g.D[2] + 1) / 2;
  return g.N * ((U0 + ST0 - 1) / ST0) * ((U1 + ST1 - 1) / ST1) * ((U2 + ST2 - 1) / ST2);
}

int launch_conv_dgrad_s2(s3_ctx* ctx, const ConvGeom& g, const float* dy, const void* img, float* dx,
                         const void* mask_y, float mask_slope, int mask_bf16, int out_bf16, float* bsum,
                         const void* mask_bits) {
  const int U0 = (g.D[0] + 1) / 2, U1 = (g.D[1] + 1) / 2, U2 = (g.D[2] + 1) / 2;
  const int tiles0 = (U0 + ST0 - 1) / ST0, tiles1 = (U1 + ST1 - 1) / ST1, tiles2 = (U2 + ST2 - 1) / ST2;
  const int n_ct = (g.Cin + 63) / 64;
  dim3 grid((unsigned)(g.N * tiles0 * tiles1 * tiles2), (unsigned)n_ct);
  const int rp = s2_rows_pad(g.Cin);
  if (out_bf16) {
    if (!conv_dgrad_s2_out16_ok(g)) S3_FAIL(ctx, S3_EINVAL, "dgrad_s2: bf16 output needs C_in = 32");
    if (mask_bits)
      hipLaunchKernelGGL((conv_dgrad_s2_kernel<2, true, true>), grid, dim3(SNT), SLDS, ctx->stream, dy,
                         (const unsigned short*)img, dx, g, rp, tiles0, tiles1, tiles2, mask_bits, mask_slope, 1,
                         bsum);
    else
    hipLaunchKernelGGL((conv_dgrad_s2_kernel<2, true>), grid, dim3(SNT), SLDS, ctx->stream, dy,
                       (const unsigned short*)img, dx, g, rp, tiles0, tiles1, tiles2, mask_y, mask_slope, mask_bf16,
                       bsum);
  } else if (g.Cin <= 32)
    hipLaunchKernelGGL(conv_dgrad_s2_kernel<2>, grid, dim3(SNT), SLDS, ctx->stream, dy,
                       (const unsigned short*)img, dx, g, rp, tiles0, tiles1, tiles2, mask_y, mask_slope, mask_bf16,
                       (float*)nullptr);
  else
    hipLaunchKernelGGL(conv_dgrad_s2_kernel<4>, grid, dim3(SNT), SLDS, ctx->stream, dy,
                       (const unsigned short*)img, dx, g, rp, tiles0, tiles1, tiles2, mask_y, mask_slope, mask_bf16,
                       (float*)nullptr);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
