// Stride-2 valid conv with 32 input channels on bf16 MFMA with an LDS halo —
// the discriminator's second layer (32 -> 32, stride 2, over the 13.9 M
// positions x 32 channels of the first activation; S3_PREC_BF16 plans, bf16
// cells in, fp32 or bf16 out).
//
// The gather kernel re-reads every input cell 27 / 8 times through L1 with
// per-tap index math (0.63 ms at C2 batch 8; the layer's input is 0.89 GB).
// Here ONE persistent 8-wave workgroup per CU (a wave per output row: two
// waves per SIMD hide each other's LDS latency) walks a contiguous range of
// 2 x 4 x 16 output tiles:
//
//   * the tile's 5 x 9 x 33 input halo (64-B bf16 cells, 95 KB) sits in LDS
//     with the t axis DE-INTERLEAVED per row — 17 even cells, then 16 odd —
//     so the 16 positions t = 0..15 of a fragment read 16 CONTIGUOUS cells
//     for each of the three t-taps (c = 0: even j, c = 1: odd j, c = 2: even
//     j + 1) under the stride-1 chunk swizzle of conv_halo32_kernel (PMC: a
//     third of the LDS-active cycles are still bank conflicts — the odd row
//     pitch of 33 cells — at 25 % LDS utilisation: not what bounds the layer);
//   * the whole 27 x 32 x 32 filter (55 KB bf16) is staged ONCE per workgroup
//     into LDS, 64-B rows: a wave's A fragment is one contiguous KB;
//   * the next tile's halo is fetched into registers (12 x 16 B per lane)
//     before the 27-tap loop and dropped into LDS after it: the per-CU L2 -> CU
//     path (~10 B / clk) that bounds this layer runs under the MFMAs.
//
// Roles as in conv_halo32_kernel: A = filter rows, B = positions; lane
// (t, kg) of the C/D fragment owns 4 consecutive output channels.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int ST0 = 2, ST1 = 4, ST2 = 16;                 // output tile
constexpr int SH0 = 2 * ST0 + 1, SH1 = 2 * ST1 + 1, SH2 = 2 * ST2 + 1;   // 5 x 9 x 33
constexpr int SHP = SH0 * SH1 * SH2;                      // 1485 halo cells
constexpr int SNW = 8, SNT = SNW * 64;      // one (s0, s1) output row per wave, two waves per SIMD
constexpr int S_HALO = SHP * 64;                          // 95,040 B
constexpr int S_FILT = 27 * 32 * 64;                      // 55,296 B
constexpr int S_LDS = S_HALO + S_FILT;                    // 150,336 B
constexpr int SNCH = (SHP * 4 + SNT - 1) / SNT;           // 12 chunks per lane
constexpr int NEVEN = ST2 + 1;                            // even cells of a row

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}

// in-row slot of input cell c2 (de-interleaved) and the chunk swizzle key
__device__ __forceinline__ int row_slot(int c2) { return (c2 & 1) ? NEVEN + (c2 >> 1) : (c2 >> 1); }
__device__ __forceinline__ int slot_key(int e) { return (e >> 1) & 3; }

// fp32 w[tap][32][cout] -> bf16 img[tap][32 rows (cout, zero past it)][32 ci]
__global__ void halo_s2_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                    int cout) {
  const int total = 27 * 32 * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ci = idx & 31, row = (idx >> 5) & 31, tp = idx >> 10;
    const float v = row < cout ? w[((size_t)tp * 32 + ci) * cout + row] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

template <int NF>
__global__ __launch_bounds__(SNT) void conv_halo_s2_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ img,
    const float* __restrict__ bias, void* __restrict__ yv, ConvGeom g,
    int tiles0, int tiles1, int tiles2, int n_tiles, int out16) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* filt = smem + S_HALO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  // The XCD of this workgroup owns a contiguous share of the tile list and
  // its workgroups walk it INTERLEAVED (tile lo + k, lo + k + nk, ...), tiles
  // numbered along s0 first: at any moment the XCD's 32 workgroups sit on 32
  // neighbouring tiles, so the halo cells two of them share (a fifth of a
  // tile along s0, a ninth along s1) are fetched from HBM once and hit in the
  // XCD's L2 the second time.  (Contiguous ranges per workgroup put 9 - 90
  // tiles = 1 - 8 MB of other workgroups' traffic between the two uses:
  // counter FETCH 1.34 x the tensor.)
  int64_t xs_lo, xs_hi;
  int xs_k, xs_nk;
  s3_xcd_share(n_tiles, xs_lo, xs_hi, xs_k, xs_nk);
  const int t_first = (int)xs_lo + xs_k, t_end = (int)xs_hi, t_step = xs_nk;
  if (t_first >= t_end) return;

  // ---- filter image -> LDS (once)
  for (int i = tid; i < S_FILT / 16; i += SNT)
    reinterpret_cast<uint4*>(filt)[i] = reinterpret_cast<const uint4*>(img)[i];

  // ---- per-lane halo chunk table (tile-invariant): chunk u of this lane is
  // item tid + u * SNT = (cell hp, 16-B chunk ch)
  int c_pack[SNCH];              // c0 | c1 << 8 | c2 << 16 | ch << 24, -1 past the end
  unsigned l_off[SNCH];          // LDS byte offset
#pragma unroll
  for (int u = 0; u < SNCH; ++u) {
    const int item = tid + u * SNT;
    c_pack[u] = -1; l_off[u] = 0;
    if (item < SHP * 4) {
      const int hp = item >> 2, ch = item & 3;
      int h = hp;
      const int c2 = h % SH2; h /= SH2;
      const int c1 = h % SH1; h /= SH1;
      const int c0 = h;
      const int e = row_slot(c2);
      c_pack[u] = c0 | (c1 << 8) | (c2 << 16) | (ch << 24);
      l_off[u] = (unsigned)((((c0 * SH1 + c1) * SH2) + e) * 64 + ((ch ^ slot_key(e)) << 4));
    }
  }
  auto tile_org = [&](int tile, int& n, int& o0, int& o1, int& o2) {
    int tr = tile;
    o0 = (tr % tiles0) * ST0; tr /= tiles0;
    o1 = (tr % tiles1) * ST1; tr /= tiles1;
    o2 = (tr % tiles2) * ST2; tr /= tiles2;
    n = tr;
  };
  uint4 pre[SNCH];
  auto halo_fetch = [&](int tile) {
    int n, o0, o1, o2;
    tile_org(tile, n, o0, o1, o2);
    const unsigned short* xb = x + (size_t)n * D0 * D1 * D2 * 32;
#pragma unroll
    for (int u = 0; u < SNCH; ++u) {
      pre[u] = make_uint4(0u, 0u, 0u, 0u);
      const int cp = c_pack[u];
      const int i0 = 2 * o0 + (cp & 255), i1 = 2 * o1 + ((cp >> 8) & 255), i2 = 2 * o2 + ((cp >> 16) & 255);
      // (cells past the tensor feed only the masked overhang of ragged tiles)
      if (cp >= 0 && i0 < D0 && i1 < D1 && i2 < D2)
        pre[u] = *reinterpret_cast<const uint4*>(xb + (((size_t)i0 * D1 + i1) * D2 + i2) * 32 + (cp >> 24) * 8);
    }
  };
  auto halo_put = [&]() {
#pragma unroll
    for (int u = 0; u < SNCH; ++u)
      if (c_pack[u] >= 0) *reinterpret_cast<uint4*>(halo + l_off[u]) = pre[u];
  };

  halo_fetch(t_first);
  halo_put();
  __syncthreads();

  // B (position) fragment addresses: t-tap c reads slots e0(c) + j
  int off_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int e = (c == 1 ? NEVEN : (c >> 1)) + j;
    off_c[c] = e * 64 + ((kg ^ slot_key(e)) << 4);
  }
  // this wave's (s0, s1) output row
  const int rowb = ((2 * (wave / ST1)) * SH1 + 2 * (wave % ST1)) * SH2 * 64;
  const char* fa = filt + j * 64 + kg * 16;
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  const int R = g.Cout;

  for (int tile = t_first; tile < t_end; tile += t_step) {
    const bool has_next = tile + t_step < t_end;
    if (has_next) halo_fetch(tile + t_step);
    f32x4 acc[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) {
      const int a = tp / 9, b = (tp / 3) % 3, c = tp % 3;
      bf16x8 afr[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        afr[nf] = *reinterpret_cast<const bf16x8*>(fa + (tp * 32 + nf * 16) * 64);
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(halo + rowb + (a * SH1 + b) * SH2 * 64 + off_c[c]);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[nf], bfr, acc[nf], 0, 0, 0);
    }

    // ---- C/D: col = t, row = 4 kg + r (channel 16 nf + 4 kg + r)
    int n, o0b, o1b, o2b;
    tile_org(tile, n, o0b, o1b, o2b);
    const int o2 = o2b + j;
    {
      const int o0 = o0b + wave / ST1, o1 = o1b + wave % ST1;
      const bool ok = o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2];
      const size_t oi = ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * R;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int ch = nf * 16 + kg * 4;
        if (ch >= R || !ok) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[nf][r] + ((bias && ch + r < R) ? bias[ch + r] : 0.f);
          v[r] = v[r] > 0.f ? v[r] : slope * v[r];
        }
        if (out16)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yv) + oi + ch) =
              make_uint2(pk2(v[0], v[1]), pk2(v[2], v[3]));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + oi + ch) =
              make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();          // every wave is past its last halo read
    if (has_next) halo_put();
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------
// C_in = C_out = 64 (the discriminator's 64 -> 64 stride-2 layer: 35 GFLOP at
// C2 batch 8, 263 us on the gather kernel with every input cell fetched 2.3
// times).  A 64-channel halo of a useful tile and the 221 KB filter do not fit
// LDS together, so the contraction is split over the two 32-channel halves of
// the input: pass 0 leaves raw fp32 sums, pass 1 adds them and finishes (bias,
// activation, bf16 / fp32 store).  Per pass a persistent 8-wave workgroup keeps
// its half filter (27 x 64 rows x 64 B = 110,592 B) and the de-interleaved
// halo of a 2 x 2 x 16 output tile (5 x 5 x 33 cells x 64 B = 52,800 B) in LDS
// — 163,392 of 163,840 B — and prefetches the next halo into registers under
// the taps; waves 0-3 / 4-7 take the output channels 0-31 / 32-63 of the four
// (s0, s1) rows.
constexpr int KT0 = 2, KT1 = 2, KT2 = 16;
constexpr int KH0 = 2 * KT0 + 1, KH1 = 2 * KT1 + 1, KH2 = 2 * KT2 + 1;   // 5 x 5 x 33
constexpr int KHP = KH0 * KH1 * KH2;                       // 825 halo cells
constexpr int KNW = 8, KNT = KNW * 64;
constexpr int K_HALO = KHP * 64;                           // 52,800 B
constexpr int K_FILT = 27 * 64 * 64;                       // 110,592 B
constexpr int K_LDS = K_HALO + K_FILT;                     // 163,392 B
constexpr int KNCH = (KHP * 4 + KNT - 1) / KNT;            // 7 chunks per lane

// fp32 w[tap][64][cout 64] -> bf16 img[pass 2][tap][64 rows (cout)][32 ci of that half]
__global__ void halo_s2_k64_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img) {
  const int total = 2 * 27 * 64 * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx & 31, row = (idx >> 5) & 63;
    const int tp = (idx >> 11) % 27, ps = idx / (27 * 2048);
    img[idx] = (unsigned short)(pk2(w[((size_t)tp * 64 + ps * 32 + k) * 64 + row], 0.f) & 0xFFFFu);
  }
}

// ps = 0: part[pos][64] = raw sums over input channels 0..31;  ps = 1: y =
// act(part + sums over channels 32..63 + bias)
__global__ __launch_bounds__(KNT) void conv_halo_s2_k64_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ img,
    const float* __restrict__ bias, float* __restrict__ part, void* __restrict__ yv, ConvGeom g,
    int tiles0, int tiles1, int tiles2, int n_tiles, int out16, int ps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* filt = smem + K_HALO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kg = lane >> 4;
  const int row = wave & 3, nh = wave >> 2;                // (s0, s1) row of the tile, cout half
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  const int rank = s3_xcd_tile(blockIdx.x, gridDim.x);
  const int t_first = (int)((int64_t)rank * n_tiles / gridDim.x);
  const int t_end = (int)((int64_t)(rank + 1) * n_tiles / gridDim.x);
  if (t_first >= t_end) return;

  for (int i = tid; i < K_FILT / 16; i += KNT)
    reinterpret_cast<uint4*>(filt)[i] = reinterpret_cast<const uint4*>(img + (size_t)ps * 27 * 2048)[i];

  int c_pack[KNCH];              // c0 | c1 << 8 | c2 << 16 | ch << 24, -1 past the end
  unsigned l_off[KNCH];
#pragma unroll
  for (int u = 0; u < KNCH; ++u) {
    const int item = tid + u * KNT;
    c_pack[u] = -1; l_off[u] = 0;
    if (item < KHP * 4) {
      const int hp = item >> 2, ch = item & 3;
      int h = hp;
      const int c2 = h % KH2; h /= KH2;
      const int c1 = h % KH1; h /= KH1;
      const int c0 = h;
      const int e = row_slot(c2);
      c_pack[u] = c0 | (c1 << 8) | (c2 << 16) | (ch << 24);
      l_off[u] = (unsigned)((((c0 * KH1 + c1) * KH2) + e) * 64 + ((ch ^ slot_key(e)) << 4));
    }
  }
  auto tile_org = [&](int tile, int& n, int& o0, int& o1, int& o2) {
    int tr = tile;
    o2 = (tr % tiles2) * KT2; tr /= tiles2;
    o1 = (tr % tiles1) * KT1; tr /= tiles1;
    o0 = (tr % tiles0) * KT0; tr /= tiles0;
    n = tr;
  };
  uint4 pre[KNCH];
  auto halo_fetch = [&](int tile) {
    int n, o0, o1, o2;
    tile_org(tile, n, o0, o1, o2);
    const unsigned short* xb = x + (size_t)n * D0 * D1 * D2 * 64 + ps * 32;
#pragma unroll
    for (int u = 0; u < KNCH; ++u) {
      pre[u] = make_uint4(0u, 0u, 0u, 0u);
      const int cp = c_pack[u];
      const int i0 = 2 * o0 + (cp & 255), i1 = 2 * o1 + ((cp >> 8) & 255), i2 = 2 * o2 + ((cp >> 16) & 255);
      // (cells past the tensor feed only the masked overhang of ragged tiles)
      if (cp >= 0 && i0 < D0 && i1 < D1 && i2 < D2)
        pre[u] = *reinterpret_cast<const uint4*>(xb + (((size_t)i0 * D1 + i1) * D2 + i2) * 64 + (cp >> 24) * 8);
    }
  };
  auto halo_put = [&]() {
#pragma unroll
    for (int u = 0; u < KNCH; ++u)
      if (c_pack[u] >= 0) *reinterpret_cast<uint4*>(halo + l_off[u]) = pre[u];
  };
  halo_fetch(t_first);
  halo_put();
  __syncthreads();

  int off_c[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int e = (c == 1 ? NEVEN : (c >> 1)) + j;
    off_c[c] = e * 64 + ((kg ^ slot_key(e)) << 4);
  }
  const int rowb = ((2 * (row / KT1)) * KH1 + 2 * (row % KT1)) * KH2 * 64;
  const char* fa = filt + (nh * 32 + j) * 64 + kg * 16;
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);

  for (int tile = t_first; tile < t_end; ++tile) {
    const bool has_next = tile + 1 < t_end;
    if (has_next) halo_fetch(tile + 1);
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = acc[0];
#pragma unroll
    for (int tp = 0; tp < 27; ++tp) {
      const int a = tp / 9, b = (tp / 3) % 3, c = tp % 3;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(fa + (tp * 64) * 64);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(fa + (tp * 64 + 16) * 64);
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(halo + rowb + (a * KH1 + b) * KH2 * 64 + off_c[c]);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bfr, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bfr, acc[1], 0, 0, 0);
    }
    // ---- C/D: col = t, row = 4 kg + r (channel 32 nh + 16 nf + 4 kg + r)
    int n, o0b, o1b, o2b;
    tile_org(tile, n, o0b, o1b, o2b);
    const int o0 = o0b + row / KT1, o1 = o1b + row % KT1, o2 = o2b + j;
    if (o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2]) {
      const size_t oi = ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * 64;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int ch = nh * 32 + nf * 16 + kg * 4;
        float4 v = make_float4(acc[nf][0], acc[nf][1], acc[nf][2], acc[nf][3]);
        if (ps == 0) {
          *reinterpret_cast<float4*>(part + oi + ch) = v;
          continue;
        }
        const float4 pv = *reinterpret_cast<const float4*>(part + oi + ch);
        float o[4] = {v.x + pv.x, v.y + pv.y, v.z + pv.z, v.w + pv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] += bias ? bias[ch + r] : 0.f;
          o[r] = o[r] > 0.f ? o[r] : slope * o[r];
        }
        if (out16)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yv) + oi + ch) =
              make_uint2(pk2(o[0], o[1]), pk2(o[2], o[3]));
        else
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + oi + ch) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    __syncthreads();          // every wave is past its last halo read
    if (has_next) halo_put();
    __syncthreads();
  }
}

}  // namespace

static bool halo_s2_k64_geom(const ConvGeom& g) { return g.Cin == 64 && g.Cout == 64; }

bool conv_halo_s2_supported(const s3_ctx* ctx, const ConvGeom& g, int precision) {
  if (precision != S3_PREC_BF16 || s3_opt_has(S3O_NO_HALO_S2)) return false;
  const bool k64 = halo_s2_k64_geom(g) && !s3_opt_has(S3O_NO_HALO_S2_K64);
  if (!k64 && (g.Cin != 32 || g.Cout % 4 != 0 || g.Cout < 16 || g.Cout > 32)) return false;
  if (g.d2s != 1) return false;
  if (g.pad_mode == S3_PAD_REFLECT) return false;
  for (int d = 0; d < 3; ++d)
    // valid padding, or TF 'same' on an even extent (one zero cell past the end,
    // none in front: the halo fetch zero-fills cells past the tensor anyway)
    if (g.k[d] != 3 || g.s[d] != 2 || g.lo[d] != 0 || (g.O[d] - 1) * 2 + 3 > g.D[d] + 1) return false;
  if ((int64_t)g.D[0] * g.D[1] * g.D[2] * g.Cin >= ((int64_t)1 << 31)) return false;
  const int64_t min_tiles = s3_opt_has(S3O_HALO_S2_MIN_TILES) ? s3_opt_int(S3O_HALO_S2_MIN_TILES, 0)
                                                                  : 4 * (int64_t)ctx->num_cu;
  const int t0 = k64 ? KT0 : ST0, t1 = k64 ? KT1 : ST1;
  return g.O[2] >= 8 &&
         (int64_t)g.N * ((g.O[0] + t0 - 1) / t0) * ((g.O[1] + t1 - 1) / t1) *
                 ((g.O[2] + ST2 - 1) / ST2) >= min_tiles;
}

size_t conv_halo_s2_packed_bytes(const ConvGeom& g) { return halo_s2_k64_geom(g) ? (size_t)2 * K_FILT : (size_t)S_FILT; }

int launch_conv_halo_s2_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* img) {
  if (halo_s2_k64_geom(g)) {
    hipLaunchKernelGGL(halo_s2_k64_pack_kernel, dim3(216), dim3(256), 0, ctx->stream, w, (unsigned short*)img);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  hipLaunchKernelGGL(halo_s2_pack_kernel, dim3(108), dim3(256), 0, ctx->stream, w, (unsigned short*)img, g.Cout);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_halo_s2_fwd(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* img,
                            const float* bias, void* y, int out_bf16) {
  static S3DeviceOnce attr_set;
  if (halo_s2_k64_geom(g)) {
    static S3DeviceOnce k_attr;
    if (!k_attr.done(ctx->device)) {
      std::lock_guard<std::mutex> lk_k_attr(k_attr.m);
      S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_s2_k64_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, K_LDS));
      k_attr.mark(ctx->device);
    }
    const int t0 = (g.O[0] + KT0 - 1) / KT0, t1 = (g.O[1] + KT1 - 1) / KT1, t2 = (g.O[2] + KT2 - 1) / KT2;
    const int n_tiles = g.N * t0 * t1 * t2;
    int grid = ctx->num_cu;
    if (grid > n_tiles) grid = n_tiles;
    const size_t pbytes = (size_t)g.N * g.O[0] * g.O[1] * g.O[2] * 64 * sizeof(float);
    int rc = ensure_scratch(ctx, pbytes);
    if (rc) return rc;
    for (int ps = 0; ps < 2; ++ps)
      hipLaunchKernelGGL(conv_halo_s2_k64_kernel, dim3(grid), dim3(KNT), K_LDS, ctx->stream,
                         (const unsigned short*)x, (const unsigned short*)img, bias, (float*)ctx->scratch, y, g,
                         t0, t1, t2, n_tiles, out_bf16, ps);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_s2_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo_s2_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS));
    attr_set.mark(ctx->device);
  }
  const int tiles0 = (g.O[0] + ST0 - 1) / ST0, tiles1 = (g.O[1] + ST1 - 1) / ST1,
            tiles2 = (g.O[2] + ST2 - 1) / ST2;
  const int n_tiles = g.N * tiles0 * tiles1 * tiles2;
  int grid = ctx->num_cu;
  if (grid > n_tiles) grid = n_tiles;
  if (g.Cout <= 16)
    hipLaunchKernelGGL(conv_halo_s2_kernel<1>, dim3(grid), dim3(SNT), S_LDS, ctx->stream,
                       (const unsigned short*)x, (const unsigned short*)img, bias, y, g, tiles0, tiles1,
                       tiles2, n_tiles, out_bf16);
  else
    hipLaunchKernelGGL(conv_halo_s2_kernel<2>, dim3(grid), dim3(SNT), S_LDS, ctx->stream,
                       (const unsigned short*)x, (const unsigned short*)img, bias, y, g, tiles0, tiles1,
                       tiles2, n_tiles, out_bf16);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
