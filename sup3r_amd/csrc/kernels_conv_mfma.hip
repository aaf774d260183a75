// Halo-tile implicit-GEMM Conv3D for the generator body of Sup3rGan
// (K2: 33 x [REFLECT pad 3 -> Conv3D 64->64 k3 -> crop 2] + the 64->200
// expansion conv = 99.7 % of the generator FLOPs), written for gfx950.
//
//   y[n, p, co] = act(b[co] + sum_{tap, ci} x[n, reflect(p + tap - 1), ci] *
//                 w[tap][ci][co]) (+ residual) (store optionally permuted by
//                 depth-to-space)
//
// im2col-free: a workgroup owns TS0 x TS1 x 16 output positions (16 = run along
// t, the innermost spatial axis) and all 64 output channels of one cout tile.
// The (TS0+2)(TS1+2)(18) x 64-channel input halo is staged ONCE in LDS with the
// reflect/zero boundary evaluated as index math in the load; every one of the
// 27 taps then reads its shifted window straight out of LDS as MFMA A
// fragments.  The 64x64 filter slab of tap+1 is prefetched into registers while
// tap is on the matrix cores and lands in the other half of a 2-slab LDS ring
// (one barrier per tap).  K = 27 * 64 = 1728 is contracted on MFMA:
//
//   S3_PREC_BF16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Halo and filter
//                  rows are 128 B (64 x bf16); 16-B chunks are XOR-swizzled
//                  (halo: chunk ^ (cell_t & 7) — conflict-free for all three
//                  tap t-shifts, checked by brute force over the four 16-lane
//                  groups; filters: chunk ^ ((row >> 1) & 7)) so
//                  that the 16-lane groups of ds_read_b128 hit 16 distinct
//                  16-B slots of the 256-B bank row.  Every LDS read address
//                  in the 27-tap loop is (per-lane register) + (immediate):
//                  the loop is ds_read_b128 + MFMA only.  Activations may be
//                  fp32 or bf16 in HBM (IN16 / OUT16): inference plans keep
//                  the 64-channel trunk in bf16, which halves the halo and
//                  epilogue bytes and removes the convert from the staging.
//   S3_PREC_BF16X3 : fp32 activations and weights split on the fly into bf16
//                  pairs (hi = bf16(v), lo = bf16(v - hi)); every product is
//                  hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate —
//                  the dropped lo*lo term and the split residue are ~2^-16
//                  relative, i.e. fp32-class results at a third of the bf16
//                  rate (5x the fp32-MFMA rate).  A 128-B LDS cell holds 32
//                  channels as [hi x 32 | lo x 32], so the tile, the swizzles
//                  and the read addresses are those of the bf16 mode; the
//                  K = 64 channels are contracted in two passes of 32 with the
//                  halo re-staged in between (accumulators stay in registers).
//   S3_PREC_F32  : v_mfma_f32_16x16x4_f32 (exact fp32, == fmaf chain): parity
//                  mode.  Halo rows padded to 66 dwords, filter rows to 80, so
//                  the per-lane ds_read_b32 of the (row, k) fragments are
//                  conflict-free.
//
// Fragment maps (guide §3): A lane l: row l&15, k-group l>>4; B lane l: col
// l&15, k-group l>>4; C/D lane l reg r: col l&15, row (l>>4)*4 + r.
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

template <int TS0, int TS1, int NW = 4>
struct Tile {
  static constexpr int H0 = TS0 + 2, H1 = TS1 + 2;
  static constexpr int HP = H0 * H1 * H2;        // halo positions
  static constexpr int MFW = TS0 * TS1 / NW;     // M fragments per wave
  static constexpr int NT = NW * 64;             // threads per workgroup
  static constexpr int NPOS = TS0 * TS1 * TS2;
  static constexpr size_t stage_bytes = (size_t)NPOS * (CT + 4) * 4;
  static constexpr size_t lds_bf16_raw = (size_t)HP * 128 + 2 * 8192;
  static constexpr size_t lds_bf16 = lds_bf16_raw > stage_bytes ? lds_bf16_raw : stage_bytes;
  static constexpr size_t lds_f32 = (size_t)HP * F32_ROW * 4 + 2 * CIN * F32_BROW * 4;
};

// pack canonical fp32 w[tap][ci][co] -> bf16 slabs [ct][tap][co 64][ci 64],
// 16-B chunks pre-swizzled (the slab is the LDS image)
__global__ void pack_bf16_kernel(const float* __restrict__ w,
                                 unsigned short* __restrict__ out, int taps,
                                 int cout, int n_ct) {
  const int64_t total = (int64_t)n_ct * taps * CT * CIN;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int ci = (int)(r % CIN); r /= CIN;
    const int row = (int)(r % CT); r /= CT;
    const int tap = (int)(r % taps); r /= taps;
    const int ct = (int)r;
    const int co = ct * CT + row;
    const float v = co < cout ? w[((int64_t)tap * CIN + ci) * cout + co] : 0.f;
    const int chunk = ci >> 3, e = ci & 7;
    const int slot = chunk ^ ((row >> 1) & 7);
    out[(((int64_t)ct * taps + tap) * CT + row) * CIN + slot * 8 + e] = f2bf(v);
  }
}

// BF16X3: canonical fp32 w[tap][ci][co] -> slabs [ct][pass 2][tap][co 64][128 B],
// a row = channels 32 pass .. 32 pass + 31 as [hi x 32 | lo x 32], 16-B chunks
// pre-swizzled like the bf16 slabs
__global__ void pack_bf16x3_kernel(const float* __restrict__ w,
                                   unsigned short* __restrict__ out, int taps,
                                   int cout, int n_ct) {
  const int64_t total = (int64_t)n_ct * 2 * taps * CT * 32;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int cl = (int)(r % 32); r /= 32;
    const int row = (int)(r % CT); r /= CT;
    const int tap = (int)(r % taps); r /= taps;
    const int pass = (int)(r % 2); r /= 2;
    const int ct = (int)r;
    const int co = ct * CT + row, ci = pass * 32 + cl;
    const float v = co < cout ? w[((int64_t)tap * CIN + ci) * cout + co] : 0.f;
    const unsigned short hi = f2bf(v);
    const unsigned short lo = f2bf(v - __uint_as_float((unsigned)hi << 16));
    const int sw = (row >> 1) & 7;
    unsigned short* o = out + ((((int64_t)ct * 2 + pass) * taps + tap) * CT + row) * CIN;
    o[(((cl >> 3)) ^ sw) * 8 + (cl & 7)] = hi;
    o[((4 + (cl >> 3)) ^ sw) * 8 + (cl & 7)] = lo;
  }
}

// NFV: N fragments of the 64-wide cout tile that are computed (bf16 mode; 2 when
// C_out <= 32 — the data gradient of the discriminator's 32 -> 64 conv — so that
// half of the MFMAs and filter-fragment reads are not spent on zero rows)
template <int PREC, int TS0, int TS1, int NW, bool IN16, bool OUT16, int NFV = 4>
__global__ __launch_bounds__(NW * 64) void conv3_mfma_kernel(
    const void* __restrict__ xv, const void* __restrict__ wpk,
    const float* __restrict__ bias, const void* __restrict__ resv,
    void* __restrict__ yv, ConvGeom g, int tiles0, int tiles1, int tiles2,
    int res16, int dbg) {
  using T = Tile<TS0, TS1, NW>;
  constexpr int H1 = T::H1, HP = T::HP, MFW = T::MFW, NT = T::NT;
  static_assert(MFW >= 1 && MFW * NW == TS0 * TS1, "tile / wave split");
  static_assert(PREC == S3_PREC_BF16 || (!IN16 && !OUT16), "bf16 I/O needs bf16 MFMA");
  constexpr bool X3 = PREC == S3_PREC_BF16X3;
  constexpr bool BF = PREC == S3_PREC_BF16 || X3;   // bf16 MFMA, 128-B LDS cells
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
          (const char*)wpk + ((size_t)ct * (X3 ? 2 * taps : taps) + tap) * 8192);
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
          const size_t pos = (((size_t)n * D0 + i0) * D1 + i1) * D2 + i2;
          if (valid) {
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
  auto e_dst = [&](int j, bool& ok) -> size_t {
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
  if (resv && RES_PRE) {
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
    for (int pass = 0; pass < (X3 ? 2 : 1); ++pass) {
      if (X3 && pass) {
        // every wave is past its last read of the first channel half (barrier
        // of tap 26); the slab of step 27 is already committed
        stage_halo(1);
        __syncthreads();
      }
#pragma unroll
      for (int ta = 0; ta < ((dbg & 4) ? 0 : 3); ++ta) {
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
          for (int tc = 0; tc < 3; ++tc) {
            const int tap = (ta * 3 + tb) * 3 + tc;
            // slab step: 27 = odd, so the ring slot of step 27 pass + tap is
            // (tap + pass) & 1
            const int slot = X3 ? ((tap + pass) & 1) : (tap & 1);
            if (X3 ? (pass == 0 || tap + 1 < 27) : (tap + 1 < 27)) b_issue(pass * 27 + tap + 1);
            if constexpr (X3) {
              bf16x8 bh[4], bl[4];
#pragma unroll
              for (int nf = 0; nf < 4; ++nf) {
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
                for (int nf = 0; nf < 4; ++nf)
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nf], acc[m][nf], 0, 0, 0);
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                  acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nf], acc[m][nf], 0, 0, 0);
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
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
            if (X3 ? (pass == 0 || tap + 1 < 27) : (tap + 1 < 27)) b_commit(slot ^ 1);
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
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stage[(mf * TS2 + kq * 4 + r) * SROW + nf * 16 + frow] = acc[m][nf][r];
  }
  __syncthreads();
  {
    float bv[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) bv[q] = (bias && e_co_ok) ? bias[e_co + q] : 0.f;
    const int act = g.act;
    const float alpha = g.alpha;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
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

template <int PREC, int TS0, int TS1, int NW, bool IN16, bool OUT16, int NFV = 4>
int launch_io(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* wpk,
              const float* bias, const void* res, void* y, int res16) {
  using T = Tile<TS0, TS1, NW>;
  const size_t lds = PREC == S3_PREC_F32 ? T::lds_f32 : T::lds_bf16;
  auto kern = conv3_mfma_kernel<PREC, TS0, TS1, NW, IN16, OUT16, NFV>;
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

template <int TS0, int TS1, int NW>
int launch_bf16(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* wpk,
                const float* bias, const void* res, void* y, ConvIO io) {
  if (io.in_bf16 && io.out_bf16)
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, true, true>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  // (fp32-out data gradients with <= 32 output channels: two N fragments)
  if (g.Cout <= 32 && !io.out_bf16 && TS0 == 4 && TS1 == 8 && NW == 16 && !s3_opt_has(S3O_NO_TILE_NF2)) {
    if (io.in_bf16)
      return launch_io<S3_PREC_BF16, TS0, TS1, NW, true, false, 2>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, false, false, 2>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  }
  if (io.in_bf16)
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, true, false>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  if (io.out_bf16)
    return launch_io<S3_PREC_BF16, TS0, TS1, NW, false, true>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
  return launch_io<S3_PREC_BF16, TS0, TS1, NW, false, false>(ctx, g, x, wpk, bias, res, y, io.res_bf16);
}

}  // namespace

bool conv_mfma_supported(const ConvGeom& g, int precision) {
  if (precision != S3_PREC_F32 && precision != S3_PREC_BF16 && precision != S3_PREC_BF16X3) return false;
  if (g.Cin != CIN) return false;
  if (g.Cout % 4 != 0 || g.Cout < 16) return false;
  if (g.k[0] != 3 || g.k[1] != 3 || (g.k[2] != 3 && g.k[2] != 1)) return false;
  for (int d = 0; d < 3; ++d) {
    if (g.s[d] != 1) return false;
    const int lo = g.k[d] == 3 ? 1 : 0;
    const bool same = g.lo[d] == lo && g.O[d] == g.D[d];
    // dgrad frame: full correlation over the padded extent, zero boundary
    const bool full = g.k[d] == 3 && g.lo[d] == 2 && g.O[d] == g.D[d] + 2 &&
                      g.pad_mode == S3_PAD_ZERO;
    // valid padding (the discriminator's 64 -> 128 conv): the halo never
    // leaves the tensor except under the masked overhang of ragged tiles
    const bool valid = g.k[d] == 3 && g.lo[d] == 0 && g.O[d] == g.D[d] - 2 &&
                       g.pad_mode != S3_PAD_REFLECT;
    if (!same && !full && !valid) return false;
  }
  if (g.d2s > 1 && (g.Cout / (g.d2s * g.d2s)) % 4 != 0) return false;
  if (g.k[2] == 1) return false;   // 2-D nets stay on the direct kernel for now
  if (g.D[2] < 8) return false;    // 16-long t runs would be mostly masked
  return true;
}

// transposed + flipped filter of the data gradient:
// wt[tap'][co][ci] = w[26 - tap'][ci][co]
__global__ void pack_dgrad_kernel(const float* __restrict__ w,
                                  float* __restrict__ wt, int cin, int cout) {
  const int64_t total = (int64_t)27 * cin * cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int ci = (int)(r % cin); r /= cin;
    const int co = (int)(r % cout); r /= cout;
    const int tp = (int)r;
    wt[idx] = w[((int64_t)(26 - tp) * cin + ci) * cout + co];
  }
}

// chunk k of the data-gradient filter of a 64 -> C_out conv (C_out > 64):
// wt[tap'][ci' = co - 64 k][co' = ci] = w[26 - tap'][ci][co], zero rows past C_out
__global__ void pack_dgrad_chunk_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                        int cout, int k) {
  const int total = 27 * 64 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ci = idx & 63, cr = (idx >> 6) & 63, tp = idx >> 12;
    const int co = 64 * k + cr;
    wt[idx] = co < cout ? w[((size_t)(26 - tp) * 64 + ci) * cout + co] : 0.f;
  }
}

ConvGeom conv_dgrad_geom(const ConvGeom& g) {
  ConvGeom d = g;
  for (int q = 0; q < 3; ++q) {
    d.D[q] = g.O[q];            // input of the dgrad conv = dPre
    d.O[q] = g.D[q] + 2;        // padded frame of x
    d.lo[q] = 2;
  }
  d.Cin = g.Cout; d.Cout = g.Cin;
  d.pad_mode = S3_PAD_ZERO; d.act = S3_ACT_NONE; d.alpha = 0.f; d.d2s = 1;
  return d;
}

bool conv_dgrad_mfma_supported(const ConvGeom& g, int precision) {
  for (int q = 0; q < 3; ++q)
    if (g.k[q] != 3 || g.s[q] != 1 || g.lo[q] != 1 || g.O[q] != g.D[q]) return false;
  return conv_mfma_supported(conv_dgrad_geom(g), precision);
}

// 64 -> C_out 'same' conv with C_out > 64 (the 64 -> 200 expansion conv): its
// data gradient contracts over C_out; run it as ceil(C_out / 64) passes of the
// 64 -> 64 halo-tile kernel over 64-channel slices of dPre, accumulating in place
ConvGeom conv_dgrad_valid_geom(const ConvGeom& g);

// (a valid-padded conv's gradient lands on x's own grid: no frame, no fold)
ConvGeom conv_dgrad_chunk_geom(const ConvGeom& g, int k) {
  ConvGeom d = g.lo[0] == 0 ? conv_dgrad_valid_geom(g) : conv_dgrad_geom(g);
  d.Cin = 64; d.Cout = 64;
  d.in_cstride = g.Cout;
  d.in_cvalid = g.Cout - 64 * k < 64 ? g.Cout - 64 * k : 64;
  return d;
}

bool conv_dgrad_chunked_supported(const ConvGeom& g, int precision) {
  // (BF16X3 plans since round 4: the split-bf16 tile kernel over fp32 slices)
  if ((precision != S3_PREC_BF16 && (precision != S3_PREC_BF16X3 || s3_opt_has(S3O_NO_DGRAD_X3))) ||
      s3_opt_has(S3O_NO_DGRAD_CHUNKED))
    return false;
  if (g.Cin != 64 || g.Cout <= 64 || g.Cout % 8 != 0 || g.Cout > 512) return false;
  bool same = true, valid = g.pad_mode != S3_PAD_REFLECT && g.d2s == 1;
  for (int q = 0; q < 3; ++q) {
    if (g.k[q] != 3 || g.s[q] != 1) return false;
    same = same && g.lo[q] == 1 && g.O[q] == g.D[q];
    valid = valid && g.lo[q] == 0 && g.O[q] == g.D[q] - 2;
  }
  if (!same && !valid) return false;
  ConvGeom base = g;
  base.Cout = 64; base.d2s = 1;
  return conv_mfma_supported(valid ? conv_dgrad_valid_geom(base) : conv_dgrad_geom(base), precision);
}

int launch_conv_dgrad_chunk_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, float* wt, int k) {
  hipLaunchKernelGGL(pack_dgrad_chunk_kernel, dim3(432), dim3(256), 0, ctx->stream, w, wt, g.Cout, k);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// valid-padded forward conv (lo = 0, O = D - 2): its data gradient is the full
// correlation of dPre with the flipped filter and lands directly on x's grid
// (no padded frame, no fold)
ConvGeom conv_dgrad_valid_geom(const ConvGeom& g) {
  ConvGeom d = conv_dgrad_geom(g);
  for (int q = 0; q < 3; ++q) d.O[q] = g.D[q];
  return d;
}

bool conv_dgrad_mfma_valid_supported(const ConvGeom& g, int precision) {
  if (g.pad_mode == S3_PAD_REFLECT || g.d2s != 1) return false;
  for (int q = 0; q < 3; ++q)
    if (g.k[q] != 3 || g.s[q] != 1 || g.lo[q] != 0 || g.O[q] != g.D[q] - 2) return false;
  return conv_mfma_supported(conv_dgrad_valid_geom(g), precision);
}

int launch_conv_dgrad_pack(s3_ctx* ctx, const ConvGeom& g, const float* w,
                           float* wt) {
  const int64_t total = (int64_t)27 * g.Cin * g.Cout;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, wt, g.Cin, g.Cout);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// the bf16 store handles 8 consecutive channels per thread
bool conv_mfma_bf16_out_ok(const ConvGeom& g) {
  return g.Cout % 8 == 0 && (g.Cout / (g.d2s * g.d2s)) % 8 == 0;
}

size_t conv_mfma_packed_bytes(const ConvGeom& g, int precision) {
  if (precision == S3_PREC_BF16) {
    const int n_ct = (g.Cout + CT - 1) / CT;
    size_t b = (size_t)n_ct * g.k[0] * g.k[1] * g.k[2] * CT * CIN * 2;
    if (conv_mfma_persist_geom_ok(g) || conv_mfma_persist_dgrad_geom_ok(g)) b += conv_mfma_persist_image_bytes(g);
    return b;
  }
  if (precision == S3_PREC_BF16X3)   // hi | lo images of both channel halves
    return (size_t)((g.Cout + CT - 1) / CT) * 2 * g.k[0] * g.k[1] * g.k[2] * CT * CIN * 2;
  return 16;  // f32 mode reads the canonical weights directly
}

int launch_conv_mfma_pack(s3_ctx* ctx, const ConvGeom& g, int precision,
                          const float* w, void* packed) {
  if (precision == S3_PREC_BF16X3) {
    const int taps3 = g.k[0] * g.k[1] * g.k[2];
    const int n_ct3 = (g.Cout + CT - 1) / CT;
    const int64_t total3 = (int64_t)n_ct3 * 2 * taps3 * CT * 32;
    int grid3 = (int)((total3 + 255) / 256);
    if (grid3 > 2048) grid3 = 2048;
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(grid3), dim3(256), 0, ctx->stream, w, (unsigned short*)packed, taps3, g.Cout, n_ct3);
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  if (precision != S3_PREC_BF16) return S3_OK;
  const int taps = g.k[0] * g.k[1] * g.k[2];
  const int n_ct = (g.Cout + CT - 1) / CT;
  const int64_t total = (int64_t)n_ct * taps * CT * CIN;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, (unsigned short*)packed, taps, g.Cout, n_ct);
  S3_HIP(ctx, hipGetLastError());
  if (conv_mfma_persist_geom_ok(g) || conv_mfma_persist_dgrad_geom_ok(g))
    return launch_conv_mfma_persist_pack(ctx, g, w, (char*)packed + total * 2);
  return S3_OK;
}

int launch_conv_mfma_fwd(s3_ctx* ctx, const ConvGeom& g, int precision,
                         const void* x, const void* packed, const float* bias,
                         const void* res, void* y, ConvIO io) {
  if (precision == S3_PREC_BF16) {
    if (!g.in_cstride && conv_mfma_persist2_supported(ctx, g, io, res != nullptr))
      return launch_conv_mfma_persist2(ctx, g, x, (const char*)packed + (size_t)((g.Cout + CT - 1) / CT) * 27 * CT * CIN * 2, bias, res, y);
    if (!g.in_cstride && conv_mfma_persist_supported(ctx, g, io, res != nullptr))
      return launch_conv_mfma_persist(ctx, g, x, (const char*)packed + (size_t)((g.Cout + CT - 1) / CT) * 27 * CT * CIN * 2, bias, res, y);
    if (g.in_rep > 1 || g.res_rep > 1) S3_FAIL(ctx, S3_ESTATE, "conv with a fused temporal repeat off the persistent kernel");
    // tile / wave configuration (SUP3R_AMD_MFMA_TILE overrides for A/B probes)
    const int tile_env = (int)s3_opt_int(S3O_MFMA_TILE, -1);
    int tile = tile_env;
    if (tile < 0) {
      // 512-position workgroups (16 waves) amortise the filter slabs best;
      // fall back to 128-position ones when the grid would not fill the chip
      const int64_t big = (int64_t)g.N * ((g.O[0] + 3) / 4) * ((g.O[1] + 7) / 8) * ((g.O[2] + 15) / 16);
      tile = big >= ctx->num_cu ? 4 : 3;
      // extents that are multiples of 6 rather than of 4 / 8 (the 18 x 18 x 290
      // padded frame of the trunk's data gradient): 6 x 6 x 16 tiles waste
      // 5 % of the positions instead of 55 %; compare rounds x tile size
      const int64_t six = (int64_t)g.N * ((g.O[0] + 5) / 6) * ((g.O[1] + 5) / 6) * ((g.O[2] + 15) / 16);
      const int64_t ncu = ctx->num_cu;
      if (tile == 4 && six >= ncu && ((six + ncu - 1) / ncu) * 576 < ((big + ncu - 1) / ncu) * 512 &&
          !s3_opt_has(S3O_NO_TILE66))
        tile = 7;
    }
    if (tile == 1) return launch_bf16<2, 4, 4>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 2) return launch_bf16<4, 4, 8>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 3) return launch_bf16<2, 4, 8>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 4) return launch_bf16<4, 8, 16>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 6) return launch_bf16<4, 8, 8>(ctx, g, x, packed, bias, res, y, io);
    if (tile == 7) return launch_bf16<6, 6, 12>(ctx, g, x, packed, bias, res, y, io);
    return launch_bf16<4, 4, 4>(ctx, g, x, packed, bias, res, y, io);
  }
  if (precision == S3_PREC_BF16X3) {
    // fp32 activations, split on the fly; the 512-position tile when it fills
    // the chip, the 128-position one otherwise
    const int64_t big = (int64_t)g.N * ((g.O[0] + 3) / 4) * ((g.O[1] + 7) / 8) * ((g.O[2] + 15) / 16);
    if (big >= ctx->num_cu)
      return launch_io<S3_PREC_BF16X3, 4, 8, 8, false, false>(ctx, g, x, packed, bias, res, y, 0);
    return launch_io<S3_PREC_BF16X3, 2, 4, 8, false, false>(ctx, g, x, packed, bias, res, y, 0);
  }
  // f32: the filters are read in canonical layout; `packed` is unused
  return launch_io<S3_PREC_F32, 2, 4, 4, false, false>(ctx, g, x, packed, bias, res, y, 0);
}
