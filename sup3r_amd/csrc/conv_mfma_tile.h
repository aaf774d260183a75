// The halo-tile implicit-GEMM conv kernel template shared by
// kernels_conv_mfma.hip (the 64 -> C_out 3x3x3 trunk instantiations) and
// kernels_conv_mfma_gen.hip (logical-axes / any-channel-count instantiations).
// See kernels_conv_mfma.hip for the design notes.
#pragma once
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int TS2 = 16;
constexpr int H2 = TS2 + 2;
constexpr int CIN = 64;
constexpr int CT = 64;          // cout tile
constexpr int F32_ROW = 66;     // halo row stride (dwords), f32 mode
constexpr int F32_BROW = 80;    // filter row stride (dwords), f32 mode

// two fp32 -> packed bf16x2 (v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ inline unsigned pack_bf16(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

__device__ inline unsigned short f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);   // round to nearest even
  return (unsigned short)(u >> 16);
}

__device__ inline float act_f(float v, int act, float alpha) {
  // one select for every kind (slope 1 = identity, 0 = ReLU, alpha = Leaky):
  // testing the kind per element compiles to two scalar branches per value
  const float s = act == S3_ACT_LEAKY ? alpha : (act == S3_ACT_RELU ? 0.f : 1.f);
  return v > 0.f ? v : s * v;
}

// KA: taps along logical axis 0 (3, or 1 for the 2-D nets whose axis 0 is the
// batch / a k = 1 axis: no halo rows there)
template <int TS0, int TS1, int NW = 4, int KA = 3>
struct Tile {
  static constexpr int H0 = TS0 + KA - 1, H1 = TS1 + 2;
  static constexpr int HP = H0 * H1 * H2;        // halo positions
  static constexpr int MFW = TS0 * TS1 / NW;     // M fragments per wave
  static constexpr int NT = NW * 64;             // threads per workgroup
  static constexpr int NPOS = TS0 * TS1 * TS2;
  static constexpr size_t stage_bytes = (size_t)NPOS * (CT + 4) * 4;
  static constexpr size_t lds_bf16_raw = (size_t)HP * 128 + 2 * 8192;
  static constexpr size_t lds_bf16 = lds_bf16_raw > stage_bytes ? lds_bf16_raw : stage_bytes;
  static constexpr size_t lds_f32 = (size_t)HP * F32_ROW * 4 + 2 * CIN * F32_BROW * 4;
};

// NFV: N fragments of the 64-wide cout tile that are computed (bf16 mode; 2 when
// C_out <= 32 — the data gradient of the discriminator's 32 -> 64 conv — so that
// half of the MFMAs and filter-fragment reads are not spent on zero rows)
// GEN (kernels_conv_mfma_gen.hip): the geometry is LOGICAL — axes (a0, a1, a2)
// with cell strides g.xs / g.ys (any permutation of the tensor's (n, s1, s2, t);
// a2 is the 16-position run), C_in any count (K passes of 64 channels, 32 in
// BF16X3; missing channels are zero cells), C_out any count (N fragments past
// it are zero rows of the packed filter; channel groups that straddle a
// depth-to-space block or the end of the channel axis are stored one by one).
template <int PREC, int TS0, int TS1, int NW, bool IN16, bool OUT16, int NFV = 4, int KA = 3, bool GEN = false>
__global__ __launch_bounds__(NW * 64) void conv3_mfma_kernel(
    const void* __restrict__ xv, const void* __restrict__ wpk,
    const float* __restrict__ bias, const void* __restrict__ resv,
    void* __restrict__ yv, ConvGeom g, int tiles0, int tiles1, int tiles2,
    int res16, int dbg) {
  using T = Tile<TS0, TS1, NW, KA>;
  constexpr int H1 = T::H1, HP = T::HP, MFW = T::MFW, NT = T::NT;
  static_assert(MFW >= 1 && MFW * NW == TS0 * TS1, "tile / wave split");
  static_assert(PREC == S3_PREC_BF16 || (!IN16 && !OUT16), "bf16 I/O needs bf16 MFMA");
  constexpr bool X3 = PREC == S3_PREC_BF16X3;
  constexpr bool BF = PREC == S3_PREC_BF16 || X3;   // bf16 MFMA, 128-B LDS cells
  static_assert(BF || (!GEN && KA == 3), "logical-axes geometry: bf16 MFMA modes only");
  constexpr int TAPS = KA * 9;
  constexpr int KCH = X3 ? 32 : 64;                 // channels per K pass
  const int npass = GEN ? (g.Cin + KCH - 1) / KCH : (X3 ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // XCD-aware tile order: dispatcher places block b on XCD b % 8; hand each
  // XCD a contiguous run of tiles so neighbouring halos share its private L2.
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, k = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int ct = blockIdx.y;
  int tr = bid;
  const int t2i = tr % tiles2; tr /= tiles2;
  const int t1i = tr % tiles1; tr /= tiles1;
  const int t0i = tr % tiles0; tr /= tiles0;
  const int n = tr;
  const int org0 = t0i * TS0, org1 = t1i * TS1, org2 = t2i * TS2;
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int KK1 = g.k[1], KK2 = g.k[2];

  char* halo = smem;
  char* bslab = smem + (BF ? (size_t)HP * 128 : (size_t)HP * F32_ROW * 4);
  constexpr int BSLAB_BYTES = BF ? 8192 : CIN * F32_BROW * 4;

  // ---- B slab register prefetch helpers (one slab = 512 x 16 B in bf16,
  // 1024 x 16 B in f32; NT threads share it)
  constexpr int SLAB16 = BF ? 512 : 1024;  // 16-B units
  constexpr int NBQ = SLAB16 >= NT ? SLAB16 / NT : 1;
  uint4 breg[NBQ];
  // (BF16X3: `tap` runs over 2 x 27 slabs, pass-major)
  auto b_issue = [&](int tap) {
    if (BF) {
      const uint4* src = reinterpret_cast<const uint4*>(
          (const char*)wpk + ((size_t)ct * (GEN ? npass * TAPS : (X3 ? 2 * taps : taps)) + tap) * 8192);
#pragma unroll
      for (int q = 0; q < NBQ; ++q)
        if (tid + q * NT < SLAB16) breg[q] = src[tid + q * NT];
    } else {
      const float* w = (const float*)wpk + (size_t)tap * CIN * g.Cout + ct * CT;
#pragma unroll
      for (int q = 0; q < NBQ; ++q) {
        const int idx = tid + q * NT;
        const int ci = idx >> 4, co4 = (idx & 15) * 4;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ct * CT + co4 < g.Cout)
          v = *reinterpret_cast<const uint4*>(w + (size_t)ci * g.Cout + co4);
        breg[q] = v;
      }
    }
  };
  auto b_commit = [&](int buf) {
    char* dst = bslab + buf * BSLAB_BYTES;
    if (BF) {
#pragma unroll
      for (int q = 0; q < NBQ; ++q)
        if (tid + q * NT < SLAB16) reinterpret_cast<uint4*>(dst)[tid + q * NT] = breg[q];
    } else {
#pragma unroll
      for (int q = 0; q < NBQ; ++q) {
        const int idx = tid + q * NT;
        const int ci = idx >> 4, co4 = (idx & 15) * 4;
        *reinterpret_cast<uint4*>(dst + ((size_t)ci * F32_BROW + co4) * 4) = breg[q];
      }
    }
  };

  b_issue(0);

  // ---- stage the input halo (boundary handled here, once per element).
  // UN items per thread per trip: all global loads of a trip are issued before
  // the first convert/ds_write so many 16-B loads per lane are in flight.
  // BF16X3: `pass` selects the 32-channel half; an item is 8 fp32 channels that
  // become one hi and one lo 16-B chunk of the cell.
  auto stage_halo = [&](int pass) __attribute__((always_inline)) {
    constexpr int CHUNKS = X3 ? 4 : (BF ? 8 : 16);  // items per position
    constexpr int ITEMS = HP * CHUNKS;
    constexpr int UN = IN16 ? (ITEMS + NT - 1) / NT : 4;   // bf16 in: one trip
    for (int base = tid; base < ((dbg & 1) ? 0 : ITEMS); base += NT * UN) {
      uint4 va[UN], vb[IN16 ? 1 : UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int item = base + u * NT;
        va[u] = make_uint4(0, 0, 0, 0);
        if (!IN16) vb[u] = va[u];
        if (item < ITEMS) {
          const int hp = item / CHUNKS, ch = item % CHUNKS;
          int h = hp;
          const int c2 = h % H2; h /= H2;
          const int c1 = h % H1; h /= H1;
          const int c0 = h;
          int i0 = org0 + c0 - g.lo[0], i1 = org1 + c1 - g.lo[1], i2 = org2 + c2 - g.lo[2];
          bool valid = true;
          if (g.pad_mode == S3_PAD_REFLECT) {
            i0 = s3_reflect(i0, D0); i1 = s3_reflect(i1, D1); i2 = s3_reflect(i2, D2);
          } else {
            valid = i0 >= 0 && i0 < D0 && i1 >= 0 && i1 < D1 && i2 >= 0 && i2 < D2;
          }
          // ragged tiles: keep addresses legal (results are masked at the store)
          i0 = i0 < 0 ? 0 : (i0 > D0 - 1 ? D0 - 1 : i0);
          i1 = i1 < 0 ? 0 : (i1 > D1 - 1 ? D1 - 1 : i1);
          i2 = i2 < 0 ? 0 : (i2 > D2 - 1 ? D2 - 1 : i2);
          const size_t pos = GEN ? (size_t)n * g.xn + (size_t)i0 * g.xs[0] + (size_t)i1 * g.xs[1] +
                                       (size_t)i2 * g.xs[2]
                                 : (((size_t)n * D0 + i0) * D1 + i1) * D2 + i2;
          if (GEN && valid) {
            // channels [c0, c0 + 8) of this K pass: whole 16-B chunks where the
            // cell stride keeps them aligned, element by element otherwise
            const int c0 = pass * KCH + ch * 8;
            const int Ci = g.Cin;
            if (IN16) {
              if (c0 < Ci)     // (bf16 tensors have C % 8 == 0)
                va[u] = *reinterpret_cast<const uint4*>(
                    reinterpret_cast<const unsigned short*>(xv) + pos * Ci + c0);
            } else {
              const float* src = reinterpret_cast<const float*>(xv) + pos * Ci + c0;
              if (c0 + 8 <= Ci && (Ci & 3) == 0) {
                va[u] = *reinterpret_cast<const uint4*>(src);
                vb[u] = *reinterpret_cast<const uint4*>(src + 4);
              } else if (c0 < Ci) {
                float t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = c0 + e < Ci ? src[e] : 0.f;
                va[u] = make_uint4(__float_as_uint(t[0]), __float_as_uint(t[1]), __float_as_uint(t[2]), __float_as_uint(t[3]));
                vb[u] = make_uint4(__float_as_uint(t[4]), __float_as_uint(t[5]), __float_as_uint(t[6]), __float_as_uint(t[7]));
              }
            }
          } else if (valid) {
            if (IN16) {
              va[u] = *reinterpret_cast<const uint4*>(
                  reinterpret_cast<const unsigned short*>(xv) + pos * CIN + ch * 8);
            } else if (X3) {
              // (channel slice of a wider tensor — the chunked data gradient of
              // the 64 -> 200 conv: chunks past in_cvalid stay zero)
              const int cstr = g.in_cstride ? g.in_cstride : CIN;
              if (!g.in_cstride || pass * 32 + ch * 8 < g.in_cvalid) {
                const float* src = reinterpret_cast<const float*>(xv) + pos * cstr + pass * 32 + ch * 8;
                va[u] = *reinterpret_cast<const uint4*>(src);
                vb[u] = *reinterpret_cast<const uint4*>(src + 4);
              }
            } else if (BF) {
              // (channel slice of a wider tensor: chunks past in_cvalid stay zero)
              const int cstr = g.in_cstride ? g.in_cstride : CIN;
              if (!g.in_cstride || ch * 8 < g.in_cvalid) {
                const float* src = reinterpret_cast<const float*>(xv) + pos * cstr + ch * 8;
                va[u] = *reinterpret_cast<const uint4*>(src);
                vb[u] = *reinterpret_cast<const uint4*>(src + 4);
              }
            } else {
              va[u] = *reinterpret_cast<const uint4*>(
                  reinterpret_cast<const float*>(xv) + pos * CIN + ch * 4);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int item = base + u * NT;
        if (item < ITEMS) {
          const int hp = item / CHUNKS, ch = item % CHUNKS;
          if (BF) {
            uint4 o;
            if (IN16) {
              o = va[u];
            } else {
              const uint4 a = va[u], b = vb[IN16 ? 0 : u];
              o.x = pack_bf16(__uint_as_float(a.x), __uint_as_float(a.y));
              o.y = pack_bf16(__uint_as_float(a.z), __uint_as_float(a.w));
              o.z = pack_bf16(__uint_as_float(b.x), __uint_as_float(b.y));
              o.w = pack_bf16(__uint_as_float(b.z), __uint_as_float(b.w));
              if (X3) {
                // lo = bf16(v - hi): the residue of the first rounding
                uint4 l;
                l.x = pack_bf16(__uint_as_float(a.x) - bf_lo(o.x), __uint_as_float(a.y) - bf_hi(o.x));
                l.y = pack_bf16(__uint_as_float(a.z) - bf_lo(o.y), __uint_as_float(a.w) - bf_hi(o.y));
                l.z = pack_bf16(__uint_as_float(b.x) - bf_lo(o.z), __uint_as_float(b.y) - bf_hi(o.z));
                l.w = pack_bf16(__uint_as_float(b.z) - bf_lo(o.w), __uint_as_float(b.w) - bf_hi(o.w));
                const int slot_lo = (4 + ch) ^ ((hp % H2) & 7);
                *reinterpret_cast<uint4*>(halo + (size_t)hp * 128 + slot_lo * 16) = l;
              }
            }
            // swizzle keyed on the t coordinate of the halo cell so that the
            // read-side key depends on the tap's t-shift only (3 variants)
            const int slot = ch ^ ((hp % H2) & 7);
            *reinterpret_cast<uint4*>(halo + (size_t)hp * 128 + slot * 16) = o;
          } else {
            const uint4 a = va[u];
            uint2* d = reinterpret_cast<uint2*>(halo + ((size_t)hp * F32_ROW + ch * 4) * 4);
            d[0] = make_uint2(a.x, a.y);
            d[1] = make_uint2(a.z, a.w);
          }
        }
      }
    }
  };
  stage_halo(0);
  b_commit(0);
  __syncthreads();

  // ---- epilogue geometry (per thread: CPT consecutive channels of NIT
  // positions) and residual prefetch: the residual rows are fetched NOW, into
  // registers, so their HBM latency hides under the 27-tap MFMA loop
  constexpr int CPT = OUT16 ? 8 : 4;     // channels per thread (16-B store)
  constexpr int GPP = CT / CPT;          // thread groups per position
  constexpr int PPP = NT / GPP;          // positions per pass
  constexpr int NIT = T::NPOS / PPP;
  const int e_c0 = (tid % GPP) * CPT;
  const int e_co = ct * CT + e_c0;
  const bool e_co_ok = e_co < g.Cout;    // C_out % CPT == 0 (checked at dispatch)
  const int e_b = g.d2s;
  const int e_cpo = g.Cout / (e_b * e_b);
  const int e_blk = e_co / e_cpo, e_cc = e_co % e_cpo;
  // GEN: element offset of channel co of local position j through the cell
  // strides of the final (depth-to-space'd) tensor
  auto e_dst_c = [&](int j, int co, bool& ok) -> size_t {
    const int pl = (tid / GPP) + PPP * j;
    const int mf = pl / TS2, o2 = org2 + pl % TS2;
    const int o0 = org0 + mf / TS1, o1 = org1 + mf % TS1;
    ok = co < g.Cout && o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2];
    if (!ok) return 0;
    const int blk = co / e_cpo, cc = co % e_cpo;
    const size_t cell = (size_t)n * g.yn + (size_t)o0 * g.ys[0] + (size_t)o1 * g.ys[1] + (size_t)o2 * g.ys[2] +
                        (size_t)(blk / e_b) * g.yb[0] + (size_t)(blk % e_b) * g.yb[1];
    return cell * e_cpo + cc;
  };
  // whole CPT-channel groups inside one cell, 16-B aligned: the vector epilogue
  const bool e_vec = !GEN || (e_cpo % CPT == 0 && g.Cout % CPT == 0);
  auto e_dst = [&](int j, bool& ok) -> size_t {
    if (GEN) return e_dst_c(j, e_co, ok);
    const int pl = (tid / GPP) + PPP * j;     // local position
    const int mf = pl / TS2, o2 = org2 + pl % TS2;
    const int o0 = org0 + mf / TS1, o1 = org1 + mf % TS1;
    ok = e_co_ok && o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2];
    if (!ok) return 0;
    if (e_b == 1)
      return ((((size_t)n * g.O[0] + o0) * g.O[1] + o1) * g.O[2] + o2) * g.Cout + e_co;
    return ((((size_t)n * g.O[0] * e_b + o0 * e_b + e_blk / e_b) * (g.O[1] * e_b) +
             o1 * e_b + e_blk % e_b) * g.O[2] + o2) * e_cpo + e_cc;
  };
  // (BF16X3 is register-bound: its residual rows are read in the epilogue)
  constexpr bool RES_PRE = !X3;
  uint4 rres[RES_PRE ? NIT : 1];
  if (resv && RES_PRE && e_vec) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      bool ok;
      const size_t dst = e_dst(j, ok);
      rres[j] = make_uint4(0, 0, 0, 0);
      if (ok) {
        if (res16) {
          const unsigned short* rp = reinterpret_cast<const unsigned short*>(resv) + dst;
          if (CPT == 8) {
            rres[j] = *reinterpret_cast<const uint4*>(rp);
          } else {
            const uint2 r2 = *reinterpret_cast<const uint2*>(rp);
            rres[j].x = r2.x; rres[j].y = r2.y;
          }
        } else if (CPT == 4) {
          rres[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(resv) + dst);
        }
      }
    }
  }

  // ---- per-wave fragment coordinates
  const int frow = lane & 15, kq = lane >> 4;
  f32x4 acc[MFW][4];
#pragma unroll
  for (int m = 0; m < MFW; ++m)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (BF) {
    // All LDS read addresses are (per-lane register) + (compile-time
    // immediate): the 27-tap loop below is nothing but ds_read_b128 + MFMA
    // (+ the filter prefetch).  a_addr[c][ks]: byte offset of this lane's
    // 16-B A chunk in the halo row of M-fragment 0, tap t-shift c, k-step ks;
    // b_addr[nf][ks]: the same for the B fragment rows of the filter slab.
    // BF16X3: "k-step" 0 is the hi half of the 128-B cell / row, 1 the lo half.
    static_assert(MFW <= TS1 ? (TS1 % MFW == 0) : (MFW % TS1 == 0), "tile/wave split");
    const int mf0 = wave * MFW;
    const int row0 = (mf0 / TS1) * H1 + (mf0 % TS1);
    unsigned a_addr[3][2], b_addr[4][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sw = (frow + c) & 7;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        a_addr[c][ks] = (unsigned)((row0 * H2 + frow + c) * 128 + (((ks * 4 + kq) ^ sw) << 4));
    }
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int row = nf * 16 + frow;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        b_addr[nf][ks] = (unsigned)(HP * 128 + row * 128 + (((ks * 4 + kq) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll 1
    for (int pass = 0; pass < ((X3 || GEN) ? npass : 1); ++pass) {
      if ((X3 || GEN) && pass) {
        // every wave is past its last read of the previous channel group
        // (barrier of its last tap); the first slab of this pass is committed
        stage_halo(pass);
        __syncthreads();
      }
#pragma unroll
      for (int ta = 0; ta < ((dbg & 4) ? 0 : KA); ++ta) {
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
          for (int tc = 0; tc < 3; ++tc) {
            const int tap = (ta * 3 + tb) * 3 + tc;
            // (TAPS is odd: the ring slot of step TAPS pass + tap is (tap + pass) & 1)
            const int slot = (X3 || GEN) ? ((tap + pass) & 1) : (tap & 1);
            const bool more = (X3 || GEN) ? (pass + 1 < npass || tap + 1 < TAPS) : (tap + 1 < TAPS);
            if (more) b_issue(pass * TAPS + tap + 1);
            if constexpr (X3) {
              bf16x8 bh[NFV], bl[NFV];
#pragma unroll
              for (int nf = 0; nf < NFV; ++nf) {
                bh[nf] = *reinterpret_cast<const bf16x8*>(smem + b_addr[nf][0] + slot * 8192);
                bl[nf] = *reinterpret_cast<const bf16x8*>(smem + b_addr[nf][1] + slot * 8192);
              }
#pragma unroll
              for (int m = 0; m < MFW; ++m) {
                const int roff = ((m / TS1) * H1 + (m % TS1) + ta * H1 + tb) * H2 * 128;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(smem + a_addr[tc][0] + roff);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(smem + a_addr[tc][1] + roff);
                // small terms first
#pragma unroll
                for (int nf = 0; nf < NFV; ++nf)
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nf], acc[m][nf], 0, 0, 0);
#pragma unroll
                for (int nf = 0; nf < NFV; ++nf)
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nf], acc[m][nf], 0, 0, 0);
#pragma unroll
                for (int nf = 0; nf < NFV; ++nf)
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nf], acc[m][nf], 0, 0, 0);
              }
            } else {
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bfr[NFV];
#pragma unroll
                for (int nf = 0; nf < NFV; ++nf)
                  bfr[nf] = *reinterpret_cast<const bf16x8*>(smem + b_addr[nf][ks] + slot * 8192);
#pragma unroll
                for (int m = 0; m < MFW; ++m) {
                  const int roff = ((m / TS1) * H1 + (m % TS1) + ta * H1 + tb) * H2 * 128;
                  const bf16x8 afr = *reinterpret_cast<const bf16x8*>(smem + a_addr[tc][ks] + roff);
#pragma unroll
                  for (int nf = 0; nf < NFV; ++nf)
                    acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[nf], acc[m][nf], 0, 0, 0);
                }
              }
            }
            if (more) b_commit(slot ^ 1);
            __syncthreads();
          }
        }
      }
    }
  } else {
    int hp_base[MFW];
#pragma unroll
    for (int m = 0; m < MFW; ++m) {
      const int mf = wave * MFW + m;       // (s1, s2) pair inside the tile
      hp_base[m] = ((mf / TS1) * H1 + mf % TS1) * H2 + frow;
    }
    for (int tap = 0; tap < taps; ++tap) {
      if (tap + 1 < taps) b_issue(tap + 1);
      const int ta = tap / (KK1 * KK2), tb = (tap / KK2) % KK1, tc = tap % KK2;
      const int tap_off = (ta * H1 + tb) * H2 + tc;
      const char* bs = bslab + (tap & 1) * BSLAB_BYTES;
      const float* hf = reinterpret_cast<const float*>(halo);
      const float* bf = reinterpret_cast<const float*>(bs);
#pragma unroll 4
      for (int ks = 0; ks < CIN / 4; ++ks) {
        float bv[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
          bv[nf] = bf[(ks * 4 + kq) * F32_BROW + nf * 16 + frow];
#pragma unroll
        for (int m = 0; m < MFW; ++m) {
          const float av = hf[(size_t)(hp_base[m] + tap_off) * F32_ROW + ks * 4 + kq];
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)
            acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nf], acc[m][nf], 0, 0, 0);
        }
      }
      if (tap + 1 < taps) b_commit((tap + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue.  The accumulators go through LDS (halo + slabs are dead
  // after the last barrier) so that every thread handles CPT consecutive
  // output channels of one position: 16-B coalesced residual loads / stores,
  // all independent, instead of 64 scalar 4-B accesses per lane.
  if (dbg & 2) return;   // probe: tap loop only
  constexpr int SROW = CT + 4;   // 68: keeps float4 alignment, conflict-free
  float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int m = 0; m < MFW; ++m) {
    const int mf = wave * MFW + m;
#pragma unroll
    for (int nf = 0; nf < (GEN ? NFV : 4); ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stage[(mf * TS2 + kq * 4 + r) * SROW + nf * 16 + frow] = acc[m][nf][r];
  }
  __syncthreads();
  {
    float bv[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) bv[q] = (bias && (GEN ? e_co + q < g.Cout : e_co_ok)) ? bias[e_co + q] : 0.f;
    const int act = g.act;
    const float alpha = g.alpha;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if (GEN && !e_vec) {
        // channel groups that straddle a depth-to-space block or the end of
        // the channel axis (C_out = 1, 2, 6, 14, 72 / 4 = 18 ...): one by one
        const int plq = (tid / GPP) + PPP * j;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          bool okc;
          const size_t dq = e_dst_c(j, e_co + q, okc);
          if (!okc) continue;
          float vq = act_f(stage[plq * SROW + e_c0 + q] + bv[q], act, alpha);
          if (resv)
            vq += res16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(resv)[dq] << 16)
                        : reinterpret_cast<const float*>(resv)[dq];
          if (OUT16) reinterpret_cast<unsigned short*>(yv)[dq] = f2bf(vq);
          else reinterpret_cast<float*>(yv)[dq] = vq;
        }
        continue;
      }
      bool ok;
      const size_t dst = e_dst(j, ok);
      if (!ok) continue;
      const int pl = (tid / GPP) + PPP * j;
      float v[CPT];
#pragma unroll
      for (int q = 0; q < CPT; q += 4) {
        const float4 t = *reinterpret_cast<const float4*>(stage + pl * SROW + e_c0 + q);
        v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w;
      }
#pragma unroll
      for (int q = 0; q < CPT; ++q) v[q] = act_f(v[q] + bv[q], act, alpha);
      if (resv) {
        uint4 r = rres[RES_PRE ? j : 0];
        if (!RES_PRE) r = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(resv) + dst);
        if (res16) {
          v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
          if (CPT == 8) {
            v[4 % CPT] += bf_lo(r.z); v[5 % CPT] += bf_hi(r.z);
            v[6 % CPT] += bf_lo(r.w); v[7 % CPT] += bf_hi(r.w);
          }
        } else if (CPT == 4) {
          v[0] += __uint_as_float(r.x); v[1] += __uint_as_float(r.y);
          v[2] += __uint_as_float(r.z); v[3] += __uint_as_float(r.w);
        } else {
          // fp32 residual with a bf16 store: 8 floats, read here
          const float* rp = reinterpret_cast<const float*>(resv) + dst;
#pragma unroll
          for (int q = 0; q < CPT; q += 4) {
            const float4 r4 = *reinterpret_cast<const float4*>(rp + q);
            v[q] += r4.x; v[q + 1] += r4.y; v[q + 2] += r4.z; v[q + 3] += r4.w;
          }
        }
      }
      if (OUT16) {
        uint4 o;
        o.x = pack_bf16(v[0], v[1]); o.y = pack_bf16(v[2], v[3]);
        o.z = pack_bf16(v[4 % CPT], v[5 % CPT]); o.w = pack_bf16(v[6 % CPT], v[7 % CPT]);
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(yv) + dst) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yv) + dst) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <int PREC, int TS0, int TS1, int NW, bool IN16, bool OUT16, int NFV = 4, int KA = 3, bool GEN = false>
int launch_io(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* wpk,
              const float* bias, const void* res, void* y, int res16) {
  using T = Tile<TS0, TS1, NW, KA>;
  const size_t lds = PREC == S3_PREC_F32 ? T::lds_f32 : T::lds_bf16;
  auto kern = conv3_mfma_kernel<PREC, TS0, TS1, NW, IN16, OUT16, NFV, KA, GEN>;
  static bool attr_set = false;
  if (!attr_set) {
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int tiles0 = (g.O[0] + TS0 - 1) / TS0, tiles1 = (g.O[1] + TS1 - 1) / TS1,
            tiles2 = (g.O[2] + TS2 - 1) / TS2;
  dim3 grid((unsigned)(g.N * tiles0 * tiles1 * tiles2), (unsigned)((g.Cout + CT - 1) / CT));
  const int dbg = (int)s3_opt_int(S3O_MFMA_DBG, 0);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, ctx->stream, x, wpk, bias, res, y, g, tiles0, tiles1, tiles2, res16, dbg);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

}  // namespace
