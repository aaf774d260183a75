// Weights-stationary Conv2D in the BF16X3 mode (the mode that owns north_star's
// L-inf < 1e-3: fp32 cells in and out, every fp32 product as three bf16
// products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate),
// gfx950 only.  Same layers as kernels_conv2d_ws.hip: 3 x 3, stride 1, 'same'
// extents with REFLECT boundary (the fused pad / conv / crop group of the 2-D
// generators, /root/reference/sup3r/configs/spatial/gen_*: Conv2DTranspose
// 64 -> 64 x 33 + the 64 -> 256 / 1600 expansion convs), C_in = 64, C_out a
// multiple of 64 (8 per depth-to-space cell).
//
// Why a second kernel.  A BF16X3 operand is twice a bf16 one (hi | lo): nine
// [64 x 64] filter slabs are 144 KB and a 2 x 18 x 18 halo 166 KB — the bf16
// kernel's weights-stationary arrangement does not fit, and BF16X3 2-D plans ran
// on the logical-axes tile kernel, whose one-barrier-per-tap slab ring is bound
// by nine exposed L2 slab loads per tile (174 TFLOP/s-equivalent,
// profiles/r05/config_census.md: 3.5 x slower than the bf16 plan).  Here the
// contraction is SPLIT IN TWO K PASSES of 32 input channels:
//
//   * a pass's operands have exactly the bf16 kernel's LDS shapes — halo cell =
//     [hi x 32 | lo x 32] = 128 B (648 cells = 81 KB), filter slab row = [hi x 32
//     | lo x 32] (nine slabs = 72 KB) — and the same swizzles, fragment
//     addresses and 4 rows x 4 channel fragments register blocking; the two
//     "k-steps" of the bf16 tap loop become the hi and lo halves, and a tap is
//     W_hi X_hi + W_hi X_lo + W_lo X_hi: 16 fragment reads for 48 MFMAs;
//   * the accumulators stay in registers across the two passes of a tile;
//   * the 72 KB filter image is swapped between the passes (from L2: every
//     workgroup reads the same two images), the halo half of the next pass is
//     fetched into registers under the current pass's tap loop (fp32 -> hi / lo
//     split on its way into LDS).
//
// One 8-wave workgroup per CU walks a contiguous, XCD-major run of 2 images x 16
// rows x 16 columns tiles.  MFMA time per tile and SIMD: 2 waves x 864 MFMAs x 16
// clocks = 13.8 us at 2 GHz; the bf16 kernel moves the same cells in 4.6 us.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int XT_I = 2, XT_R = 16, XT_C = 16;       // tile: images x rows x columns
constexpr int XH_R = XT_R + 2, XH_C = XT_C + 2;     // 18 x 18 halo per image
constexpr int XHP = XT_I * XH_R * XH_C;             // 648 cells
constexpr int X_HALO_BYTES = XHP * 128;             // 82,944
constexpr int X_SLAB_OFF = X_HALO_BYTES;
constexpr int X_BIAS_OFF = X_SLAB_OFF + 9 * 8192;   // 156,672
constexpr int X_LDS = X_BIAS_OFF + 256;             // 156,928
constexpr int X_NT = 512;
constexpr int X_PASS_BYTES = 9 * 8192;              // one pass's filter image

__device__ inline unsigned x3_pk(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float x3_lo16(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float x3_hi16(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// eight fp32 channels -> one 16-B chunk of bf16 roundings (hi) and one of the
// bf16-rounded residues v - hi (lo)
__device__ inline void x3_split8(const uint4 a, const uint4 b, uint4& hi, uint4& lo) {
  const float v[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                      __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
  unsigned h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = x3_pk(v[2 * e], v[2 * e + 1]);
    l[e] = x3_pk(v[2 * e] - x3_lo16(h[e]), v[2 * e + 1] - x3_hi16(h[e]));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Orders chosen for the fp32 cells on both sides of the kernel:
//  * slab row rho IS output channel rho of the tile (no permutation as in the bf16
//    kernel): the accumulators of fragment nf in lane kq are channels nf 16 + kq 4 ..
//    + 3, so one 16-B store instruction of a wave writes whole 64-B runs (the four
//    kq lanes of a column), not every other 16 B;
//  * the 32 channels of a pass sit in the K order k -> channel x3_kperm(k): chunk c
//    (8 k's) = channels 4 c .. 4 c + 3 and 16 + 4 c .. 16 + 4 c + 3, so that the two
//    16-B loads of a staging lane (chunk c of a cell) each are a quarter of a 64-B
//    run shared with the lanes of chunks c + 1 .. (halo and filter image agree on
//    the order; the contraction does not care).
__device__ __host__ inline int x3_kperm(int k) {
  const int c = k >> 3, e = k & 7;
  return e < 4 ? c * 4 + e : 16 + c * 4 + (e - 4);
}

// canonical fp32 w[tap 9][ci 64][co] -> images [ct][pass 2][tap][rho 64][hi x 32 | lo x 32],
// chunk c (8 channels) of a row at slot c ^ ((rho >> 1) & 7)
__global__ void pack_ws_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int n_ct) {
  const int total = n_ct * 2 * 9 * 64 * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int cl = idx & 31, rho = (idx >> 5) & 63;
    int r = idx >> 11;
    const int tap = r % 9; r /= 9;
    const int pass = r & 1, ct = r >> 1;
    const int co = ct * 64 + rho, ci = pass * 32 + x3_kperm(cl);
    const float v = co < cout ? w[((size_t)tap * 64 + ci) * cout + co] : 0.f;
    const unsigned hi = x3_pk(v, 0.f) & 0xFFFFu;
    const unsigned lo = x3_pk(v - __uint_as_float(hi << 16), 0.f) & 0xFFFFu;
    const int sw = (rho >> 1) & 7;
    unsigned short* o = out + ((((size_t)ct * 2 + pass) * 9 + tap) * 64 + rho) * 64;
    o[((cl >> 3) ^ sw) * 8 + (cl & 7)] = (unsigned short)hi;
    o[((4 + (cl >> 3)) ^ sw) * 8 + (cl & 7)] = (unsigned short)lo;
  }
}

struct X3Geom {
  int N, H, W;         // images, rows, columns
  int Cout, b, cpo;    // output channels of the conv, depth-to-space block, Cout / b^2
  int act;
  float alpha;
  int tiles_i, tiles_r, tiles_c;
  int dbg;             // option MFMA_DBG (ablations): 1 no halo prefetch, 2 no tap loop, 4 no stores
};

template <bool RES>
__global__ __launch_bounds__(X_NT) void conv2d_ws_x3_kernel(
    const float* __restrict__ x, const char* __restrict__ wimg, const float* __restrict__ bias,
    const float* __restrict__ res, float* __restrict__ y, X3Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.y;
  const int frow = lane & 15, kq = lane >> 4;

  // ---- this workgroup's contiguous run of tiles, XCD-major (kernels_conv2d_ws.hip)
  const int T = g.tiles_i * g.tiles_r * g.tiles_c;
  int rank;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int o = (int)((blockIdx.y * gridDim.x) & 7);
    const int xcd = (b + o) & 7;
    rank = 0;
    for (int k = 0; k < ((xcd - o) & 7); ++k) rank += (nb - k + 7) / 8;
    rank += (b - ((xcd - o) & 7)) / 8;
  }
  int t_cur = (int)(((long long)rank * T) / gridDim.x);
  const int t_end = (int)(((long long)(rank + 1) * T) / gridDim.x);
  if (t_cur >= t_end) return;

  auto tile_org = [&](int t, int& i0, int& r0, int& c0) __attribute__((always_inline)) {
    c0 = (t % g.tiles_c) * XT_C; t /= g.tiles_c;
    r0 = (t % g.tiles_r) * XT_R; t /= g.tiles_r;
    i0 = t * XT_I;
  };

  // ---- lane -> halo item.  A trip covers SIX consecutive halo rows of one image:
  // lane = (sub-row j, column, 8-channel group c4 of the pass's 32) = 6 x 18 x 4 =
  // 432 lanes (the other 80 shadow lanes 0 .. 79: same address, same data, same
  // LDS slots); trips 0 - 2 are image 0, 3 - 5 image 1.  Two 16-B loads per trip.
  const int lt = tid < 6 * XH_C * 4 ? tid : tid - 6 * XH_C * 4;
  const int h_j = lt / (XH_C * 4), h_col = (lt - h_j * (XH_C * 4)) >> 2, h_c4 = lt & 3;
  const unsigned lds_hi = (unsigned)((h_j * XH_C + h_col) * 128 + ((h_c4 ^ (h_col & 7)) << 4));
  const unsigned lds_lo = (unsigned)((h_j * XH_C + h_col) * 128 + (((4 + h_c4) ^ (h_col & 7)) << 4));
#define X3_FETCH1(PA, PB, q)                                                                    \
  {                                                                                             \
    constexpr int up_ = (6 * q >= XH_R) ? 1 : 0;                                                \
    int r_ = s3_reflect(r0_ + 6 * q - up_ * XH_R + h_j - 1, g.H);                               \
    int im_ = i0_ + up_;                                                                        \
    im_ = im_ > g.N - 1 ? g.N - 1 : im_;      /* ragged tiles: legal address, masked store */   \
    r_ = r_ < 0 ? 0 : (r_ > g.H - 1 ? g.H - 1 : r_);                                            \
    const unsigned rowcell_ = ((unsigned)im_ * g.H + r_) * g.W;   /* < 2^31 */                  \
    const uint4* s_ = reinterpret_cast<const uint4*>(x + (size_t)rowcell_ * 64 + colel_);       \
    PA = s_[0]; PB = s_[4];   /* channels 4 c4 .. + 3 and 16 + 4 c4 .. + 3 of the pass */       \
  }
#define X3_FETCH(T, PASS)                                                                       \
  {                                                                                             \
    int i0_, r0_, c0_;                                                                          \
    tile_org((T), i0_, r0_, c0_);                                                               \
    int c_ = s3_reflect(c0_ + h_col - 1, g.W);                                                  \
    c_ = c_ < 0 ? 0 : (c_ > g.W - 1 ? g.W - 1 : c_);                                            \
    const unsigned colel_ = (unsigned)c_ * 64 + (PASS) * 32 + h_c4 * 4;                         \
    X3_FETCH1(pa0, pb0, 0) X3_FETCH1(pa1, pb1, 1) X3_FETCH1(pa2, pb2, 2)                        \
    X3_FETCH1(pa3, pb3, 3) X3_FETCH1(pa4, pb4, 4) X3_FETCH1(pa5, pb5, 5)                        \
  }
#define X3_COMMIT1(PA, PB, q)                                                                   \
  {                                                                                             \
    uint4 hi_, lo_;                                                                             \
    x3_split8(PA, PB, hi_, lo_);                                                                \
    *reinterpret_cast<uint4*>(smem + lds_hi + q * (6 * XH_C * 128)) = hi_;                      \
    *reinterpret_cast<uint4*>(smem + lds_lo + q * (6 * XH_C * 128)) = lo_;                      \
  }
#define X3_COMMIT()                                                                             \
  {                                                                                             \
    X3_COMMIT1(pa0, pb0, 0) X3_COMMIT1(pa1, pb1, 1) X3_COMMIT1(pa2, pb2, 2)                     \
    X3_COMMIT1(pa3, pb3, 3) X3_COMMIT1(pa4, pb4, 4) X3_COMMIT1(pa5, pb5, 5)                     \
  }
  static_assert(XT_I == 2 && XH_R % 6 == 0 && X_NT >= 6 * XH_C * 4, "six trips of six halo rows");
  uint4 pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3, pa4, pb4, pa5, pb5;
  // the pass's filter image: global -> registers -> LDS (9 x 16 B per lane)
#define X3_LOAD_IMAGE(PASS)                                                                     \
  {                                                                                             \
    const uint4* src_ = reinterpret_cast<const uint4*>(wimg + ((size_t)ct * 2 + (PASS)) * X_PASS_BYTES); \
    uint4* dst_ = reinterpret_cast<uint4*>(smem + X_SLAB_OFF);                                  \
    uint4 wr_[9];                                                                               \
    _Pragma("unroll") for (int q = 0; q < 9; ++q) wr_[q] = src_[tid + q * X_NT];                \
    _Pragma("unroll") for (int q = 0; q < 9; ++q) dst_[tid + q * X_NT] = wr_[q];                \
  }

  // Every tile runs pass A (channels 0 .. 31) then pass B: a position's sum is bias
  // + A + B in that order whatever the batch size, the grid or the form of the
  // kernel (sample by sample == the batch, lock-step == ping-pong, bit for bit).
  // This form therefore swaps the filter image twice per tile; it is the A/B
  // fallback (option NO_WS_PP) of the ping-pong form below.
  int pass = 0;                     // the pass whose filter image is in LDS
  X3_FETCH(t_cur, pass);
  X3_LOAD_IMAGE(pass);
  if (tid < 64) {
    const int co = ct * 64 + tid;
    reinterpret_cast<float*>(smem + X_BIAS_OFF)[tid] = (bias && co < g.Cout) ? bias[co] : 0.f;
  }
  X3_COMMIT();
  __syncthreads();

  // ---- fragment addresses (kernels_conv2d_ws.hip): wave w = image w >> 2, rows 4 (w & 3) .. + 3;
  // half sel = 0 (hi) / 1 (lo) of a cell or of a slab row are chunks sel 4 + kq
  const int w_img = wave >> 2, w_row = (wave & 3) * 4;
  unsigned p_addr[3][2], f_addr[2];
#pragma unroll
  for (int tc = 0; tc < 3; ++tc)
#pragma unroll
    for (int sel = 0; sel < 2; ++sel)
      p_addr[tc][sel] = (unsigned)(((w_img * XH_R + w_row) * XH_C + frow + tc) * 128 +
                                   (((sel * 4 + kq) ^ ((frow + tc) & 7)) << 4));
#pragma unroll
  for (int sel = 0; sel < 2; ++sel)
    f_addr[sel] = (unsigned)(X_SLAB_OFF + frow * 128 + (((sel * 4 + kq) ^ ((frow >> 1) & 7)) << 4));
  const float* bl = reinterpret_cast<const float*>(smem + X_BIAS_OFF);
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  // output addressing as in the bf16 kernel (elements are fp32 here)
  const unsigned rsd = (unsigned)(g.W * g.b * g.cpo);          // elements per output row
  const unsigned long long IS = (unsigned long long)(g.H * g.b) * rsd;
  const unsigned RS = (unsigned)g.b * rsd, CS = (unsigned)(g.b * g.cpo);
  unsigned off_n[4];
  bool ch_ok[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    const int co_raw = ct * 64 + nf * 16 + kq * 4;
    ch_ok[nf] = co_raw < g.Cout;
    const int co = co_raw > g.Cout - 4 ? (g.Cout >= 4 ? g.Cout - 4 : 0) : co_raw;
    const int blk = co / g.cpo, cq = co % g.cpo;
    off_n[nf] = (unsigned)(blk / g.b) * rsd + (unsigned)((blk % g.b) * g.cpo + cq);
  }

  // one K pass of a tile over the resident halo half and filter image
#define X3_TAPS()                                                                               \
  _Pragma("unroll 1") for (int tb = 0; tb < ((g.dbg & 2) ? 0 : 3); ++tb) {                      \
    _Pragma("unroll") for (int tc = 0; tc < 3; ++tc) {                                          \
      const int tap = tb * 3 + tc;                                                              \
      bf16x8 wh[4], wl[4], ph[4], pl[4];                                                        \
      _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) {                                        \
        wh[nf] = *reinterpret_cast<const bf16x8*>(smem + f_addr[0] + nf * 2048 + tap * 8192);   \
        wl[nf] = *reinterpret_cast<const bf16x8*>(smem + f_addr[1] + nf * 2048 + tap * 8192);   \
      }                                                                                         \
      _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                           \
        ph[m] = *reinterpret_cast<const bf16x8*>(smem + p_addr[tc][0] + (m + tb) * XH_C * 128); \
        pl[m] = *reinterpret_cast<const bf16x8*>(smem + p_addr[tc][1] + (m + tb) * XH_C * 128); \
      }                                                                                         \
      /* the two cross terms first: small + small, then the leading product */                 \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                             \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                        \
          acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nf], ph[m], acc[m][nf], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                             \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                        \
          acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nf], pl[m], acc[m][nf], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                             \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                        \
          acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nf], ph[m], acc[m][nf], 0, 0, 0); \
    }                                                                                           \
  }

  for (; t_cur < t_end; ++t_cur) {
    const bool has_next = t_cur + 1 < t_end;
    // ---- first pass of the tile: its halo half and the image of `pass` are resident;
    // the other half of the SAME tile's halo is fetched under the taps
    if (!(g.dbg & 1)) X3_FETCH(t_cur, 1 - pass);
    f32x4 acc[4][4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int cb = nf * 16 + kq * 4;
      const f32x4 b4 = {bl[cb], bl[cb + 1], bl[cb + 2], bl[cb + 3]};
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][nf] = b4;
    }
    X3_TAPS()
    // every wave's fragment reads are back before the halo and the slabs are rewritten
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    pass = 1 - pass;
    X3_LOAD_IMAGE(pass);
    X3_COMMIT();
    __syncthreads();
    // ---- second pass; the next tile starts with pass A again
    const int next_first = 0;
    if (has_next && !(g.dbg & 1)) X3_FETCH(t_cur + 1, next_first);
    X3_TAPS()
    if (has_next) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (next_first != pass) {
        pass = next_first;
        X3_LOAD_IMAGE(pass);
      }
      X3_COMMIT();
      __syncthreads();
    }
    // ---- epilogue from registers: lane (position column frow, channel group kq):
    // rows m, fragments nf -> 4 consecutive channels, one 16-B store each
    int i0, r0, c0;
    tile_org(t_cur, i0, r0, c0);
    const int im = i0 + w_img, c = c0 + frow;
    const bool pos_ok = im < g.N && c < g.W && !(g.dbg & 4);
    const int imc = im > g.N - 1 ? g.N - 1 : im, cc = c > g.W - 1 ? g.W - 1 : c;
    const unsigned long long tbase = (unsigned long long)imc * IS + (unsigned long long)((unsigned)cc * CS);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = r0 + w_row + m;
      const bool row_ok = pos_ok && r < g.H;
      const int rc = r > g.H - 1 ? g.H - 1 : r;
      const unsigned long long rbase = tbase + (unsigned long long)(unsigned)rc * RS;
      f32x4 q[4];
      if constexpr (RES) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) q[nf] = *reinterpret_cast<const f32x4*>(res + rbase + off_n[nf]);
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        f32x4 v = acc[m][nf];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : slope * v[e];
        if constexpr (RES) v += q[nf];
        if (row_ok && ch_ok[nf]) *reinterpret_cast<f32x4*>(y + rbase + off_n[nf]) = v;
      }
    }
  }
#undef X3_TAPS
#undef X3_LOAD_IMAGE
#undef X3_COMMIT
#undef X3_COMMIT1
#undef X3_FETCH
#undef X3_FETCH1
}


// ---------------------------------------------------------------------------
// Ping-pong form.  In the kernel above all eight waves walk the same phases
// together and a tile costs the SUM of its MFMA time and its memory time
// (ablations, option MFMA_DBG, 48 x 75 x 75: 78 us per conv = 45 us of MFMAs + the
// 42 us the loads, fp32 -> hi / lo commits, skip reads and stores take on their
// own — at the chip's copy rate: fp32 cells are twice the bf16 kernel's bytes).
// Here, as in conv2d_ws_pp_kernel, the two half-workgroups ("groups", waves 0 - 3 /
// 4 - 7) own separate lists of SINGLE-image tiles and separate halves of the LDS
// halo, and run half a round apart:
//
//   round r (K pass p = r & 1):   phase X   | phase Y   | swap
//     group 0:                    T(t, p)   | M         | image of pass p ^ 1 -> LDS
//     group 1:                    M         | T(t', p)  |
//
//   T = the 9-tap loop of one K pass (432 MFMAs per wave, one wave per SIMD);
//   M after pass A: the prefetched second halo half of the tile -> LDS, prefetch of
//                   the next tile's first half;
//   M after pass B: epilogue + stores of the tile, the next tile's first half ->
//                   LDS, prefetch of its second half.
//
// One workgroup barrier per phase.  While one group's waves hold the matrix
// cores the other group's waves wait on memory, write LDS and store.  Both
// groups are in the same K pass during a round, so the one filter image in LDS
// serves both; every tile runs pass A then pass B (a position's sum is bias + A
// + B whatever the batch size or the grid).  The image swap between rounds is
// the one exposed step (72 KB from L2 per round of ~8 us).
constexpr int PXH_BYTES = XH_R * XH_C * 128;        // 41,472: one image's halo

template <bool RES>
__global__ __launch_bounds__(X_NT) void conv2d_ws_x3_pp_kernel(
    const float* __restrict__ x, const char* __restrict__ wimg, const float* __restrict__ bias,
    const float* __restrict__ res, float* __restrict__ y, X3Geom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int gt = tid & 255;
  const int ct = blockIdx.y;
  const int frow = lane & 15, kq = lane >> 4;

  // ---- this workgroup's run of single-image tiles, XCD-major; group g takes
  // t0 + g, t0 + g + 2, ..
  const int T = g.N * g.tiles_r * g.tiles_c;
  int rank;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int o = (int)((blockIdx.y * gridDim.x) & 7);
    const int xcd = (b + o) & 7;
    rank = 0;
    for (int k = 0; k < ((xcd - o) & 7); ++k) rank += (nb - k + 7) / 8;
    rank += (b - ((xcd - o) & 7)) / 8;
  }
  const int t0 = (int)(((long long)rank * T) / gridDim.x);
  const int t1 = (int)(((long long)(rank + 1) * T) / gridDim.x);
  if (t0 >= t1) return;
  const int n_g = (t1 - t0 - grp + 1) / 2;   // this group's tiles
  const int n_max = (t1 - t0 + 1) / 2;       // group 0's (>= group 1's)
  auto tile_org = [&](int t, int& im, int& r0, int& c0) __attribute__((always_inline)) {
    c0 = (t % g.tiles_c) * XT_C; t /= g.tiles_c;
    r0 = (t % g.tiles_r) * XT_R;
    im = t / g.tiles_r;
  };

  // ---- lane -> halo item of the group's 18 x 18 cells: a trip covers THREE halo
  // rows, lane = (sub-row j, column, 8-channel group c4) = 3 x 18 x 4 = 216 of the
  // group's 256 lanes (the others shadow lanes 0 .. 39); six trips, two 16-B loads each
  const int lt = gt < 3 * XH_C * 4 ? gt : gt - 3 * XH_C * 4;
  const int h_j = lt / (XH_C * 4), h_col = (lt - h_j * (XH_C * 4)) >> 2, h_c4 = lt & 3;
  const unsigned gb = (unsigned)(grp * PXH_BYTES);
  const unsigned lds_hi = gb + (unsigned)((h_j * XH_C + h_col) * 128 + ((h_c4 ^ (h_col & 7)) << 4));
  const unsigned lds_lo = gb + (unsigned)((h_j * XH_C + h_col) * 128 + (((4 + h_c4) ^ (h_col & 7)) << 4));
#define PX_FETCH1(PA, PB, q)                                                                    \
  {                                                                                             \
    int r_ = s3_reflect(r0_ + 3 * q + h_j - 1, g.H);                                            \
    r_ = r_ < 0 ? 0 : (r_ > g.H - 1 ? g.H - 1 : r_);                                            \
    const unsigned rowcell_ = ((unsigned)im_ * g.H + r_) * g.W;   /* < 2^31 */                  \
    const uint4* s_ = reinterpret_cast<const uint4*>(x + (size_t)rowcell_ * 64 + colel_);       \
    PA = s_[0]; PB = s_[4];                                                                     \
  }
#define PX_FETCH(T_, PASS)                                                                      \
  {                                                                                             \
    int im_, r0_, c0_;                                                                          \
    tile_org((T_), im_, r0_, c0_);                                                              \
    int c_ = s3_reflect(c0_ + h_col - 1, g.W);                                                  \
    c_ = c_ < 0 ? 0 : (c_ > g.W - 1 ? g.W - 1 : c_);                                            \
    const unsigned colel_ = (unsigned)c_ * 64 + (PASS) * 32 + h_c4 * 4;                         \
    PX_FETCH1(pa0, pb0, 0) PX_FETCH1(pa1, pb1, 1) PX_FETCH1(pa2, pb2, 2)                        \
    PX_FETCH1(pa3, pb3, 3) PX_FETCH1(pa4, pb4, 4) PX_FETCH1(pa5, pb5, 5)                        \
  }
#define PX_COMMIT1(PA, PB, q)                                                                   \
  {                                                                                             \
    uint4 hi_, lo_;                                                                             \
    x3_split8(PA, PB, hi_, lo_);                                                                \
    *reinterpret_cast<uint4*>(smem + lds_hi + q * (3 * XH_C * 128)) = hi_;                      \
    *reinterpret_cast<uint4*>(smem + lds_lo + q * (3 * XH_C * 128)) = lo_;                      \
  }
#define PX_COMMIT()                                                                             \
  {                                                                                             \
    PX_COMMIT1(pa0, pb0, 0) PX_COMMIT1(pa1, pb1, 1) PX_COMMIT1(pa2, pb2, 2)                     \
    PX_COMMIT1(pa3, pb3, 3) PX_COMMIT1(pa4, pb4, 4) PX_COMMIT1(pa5, pb5, 5)                     \
  }
  static_assert(XH_R % 3 == 0 && 256 >= 3 * XH_C * 4, "six trips of three halo rows per group");
  uint4 pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3, pa4, pb4, pa5, pb5;
#define PX_LOAD_IMAGE(PASS)                                                                     \
  {                                                                                             \
    const uint4* src_ = reinterpret_cast<const uint4*>(wimg + ((size_t)ct * 2 + (PASS)) * X_PASS_BYTES); \
    uint4* dst_ = reinterpret_cast<uint4*>(smem + X_SLAB_OFF);                                  \
    uint4 wr_[9];                                                                               \
    _Pragma("unroll") for (int q = 0; q < 9; ++q) wr_[q] = src_[tid + q * X_NT];                \
    _Pragma("unroll") for (int q = 0; q < 9; ++q) dst_[tid + q * X_NT] = wr_[q];                \
  }

  // ---- prologue: image of pass A, the group's first tile: half A -> LDS, half B
  // into the prefetch registers
  const int tg0 = t0 + grp;                  // (group 1 of a one-tile run has n_g == 0)
  const int tfirst = n_g > 0 ? tg0 : t0;
  PX_FETCH(tfirst, 0);
  PX_LOAD_IMAGE(0);
  if (tid < 64) {
    const int co = ct * 64 + tid;
    reinterpret_cast<float*>(smem + X_BIAS_OFF)[tid] = (bias && co < g.Cout) ? bias[co] : 0.f;
  }
  PX_COMMIT();
  PX_FETCH(tfirst, 1);
  __syncthreads();

  const int w_row = (wave & 3) * 4;
  unsigned p_addr[3][2], f_addr[2];
#pragma unroll
  for (int tc = 0; tc < 3; ++tc)
#pragma unroll
    for (int sel = 0; sel < 2; ++sel)
      p_addr[tc][sel] = gb + (unsigned)((w_row * XH_C + frow + tc) * 128 +
                                        (((sel * 4 + kq) ^ ((frow + tc) & 7)) << 4));
#pragma unroll
  for (int sel = 0; sel < 2; ++sel)
    f_addr[sel] = (unsigned)(X_SLAB_OFF + frow * 128 + (((sel * 4 + kq) ^ ((frow >> 1) & 7)) << 4));
  const float* bl = reinterpret_cast<const float*>(smem + X_BIAS_OFF);
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  const unsigned rsd = (unsigned)(g.W * g.b * g.cpo);
  const unsigned long long IS = (unsigned long long)(g.H * g.b) * rsd;
  const unsigned RS = (unsigned)g.b * rsd, CS = (unsigned)(g.b * g.cpo);
  unsigned off_n[4];
  bool ch_ok[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    const int co_raw = ct * 64 + nf * 16 + kq * 4;
    ch_ok[nf] = co_raw < g.Cout;
    const int co = co_raw > g.Cout - 4 ? (g.Cout >= 4 ? g.Cout - 4 : 0) : co_raw;
    const int blk = co / g.cpo, cq = co % g.cpo;
    off_n[nf] = (unsigned)(blk / g.b) * rsd + (unsigned)((blk % g.b) * g.cpo + cq);
  }

#define PX_TAPS()                                                                               \
  _Pragma("unroll 1") for (int tb = 0; tb < ((g.dbg & 2) ? 0 : 3); ++tb) {                      \
    _Pragma("unroll") for (int tc = 0; tc < 3; ++tc) {                                          \
      const int tap = tb * 3 + tc;                                                              \
      bf16x8 wh[4], wl[4], ph[4], pl[4];                                                        \
      _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) {                                        \
        wh[nf] = *reinterpret_cast<const bf16x8*>(smem + f_addr[0] + nf * 2048 + tap * 8192);   \
        wl[nf] = *reinterpret_cast<const bf16x8*>(smem + f_addr[1] + nf * 2048 + tap * 8192);   \
      }                                                                                         \
      _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                           \
        ph[m] = *reinterpret_cast<const bf16x8*>(smem + p_addr[tc][0] + (m + tb) * XH_C * 128); \
        pl[m] = *reinterpret_cast<const bf16x8*>(smem + p_addr[tc][1] + (m + tb) * XH_C * 128); \
      }                                                                                         \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                             \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                        \
          acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nf], ph[m], acc[m][nf], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                             \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                        \
          acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nf], pl[m], acc[m][nf], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                             \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                        \
          acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nf], ph[m], acc[m][nf], 0, 0, 0); \
    }                                                                                           \
  }
#define PX_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[m][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // group g's T of round r happens in phase 2 r + g, its M in phase 2 r + g + 1
  // (group 1's last M is the extra phase behind the last round)
  auto phase_T = [&](int i, int p) __attribute__((always_inline)) {
    if (i >= n_g) return;
    if (p == 0) {
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int cb = nf * 16 + kq * 4;
        const f32x4 b4 = {bl[cb], bl[cb + 1], bl[cb + 2], bl[cb + 3]};
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][nf] = b4;
      }
    }
    PX_TAPS()
  };
  const int n_rounds = 2 * n_max;
  for (int r = 0; r <= n_rounds; ++r) {
    const int p = r & 1;
    // ---------------- phase X: group 0 in T (round r), group 1 in M (after its T of round r - 1)
    if (grp == 0) {
      if (r < n_rounds) phase_T(r >> 1, p);
    } else if (r > 0) {
      const int i = (r - 1) >> 1, pp_ = (r - 1) & 1;
      if (i < n_g) {
        const int t = tg0 + 2 * i;
        const bool more = i + 1 < n_g;
        if (pp_ == 0) {
          if (!(g.dbg & 16)) PX_COMMIT();                // this tile's half B
          if (more && !(g.dbg & 1)) PX_FETCH(t + 2, 0);
        } else {
#include "conv2d_ws_x3_epilogue.inc"
          if (more) {
            if (!(g.dbg & 16)) PX_COMMIT();              // the next tile's half A
            if (!(g.dbg & 1)) PX_FETCH(t + 2, 1);
          }
        }
      }
    }
    if (r == n_rounds) break;                            // (group 1's last M was the tail)
    PX_BARRIER();
    // ---------------- phase Y: group 1 in T (round r), group 0 in M (after its T of round r)
    const bool swap = r + 1 < n_rounds && !(g.dbg & 8);
    if (grp == 1) {
      phase_T(r >> 1, p);
    } else {
      const int i = r >> 1;
      if (i < n_g) {
        const int t = tg0 + 2 * i;
        const bool more = i + 1 < n_g;
        if (p == 0) {
          if (!(g.dbg & 16)) PX_COMMIT();
          if (more && !(g.dbg & 1)) PX_FETCH(t + 2, 0);
        } else {
#include "conv2d_ws_x3_epilogue.inc"
          if (more) {
            if (!(g.dbg & 16)) PX_COMMIT();
            if (!(g.dbg & 1)) PX_FETCH(t + 2, 1);
          }
        }
      }
    }
    PX_BARRIER();
    // ---------------- the image of the next round's pass
    if (swap) {
      PX_LOAD_IMAGE(1 - p);
      PX_BARRIER();
    }
  }
#undef PX_BARRIER
#undef PX_TAPS
#undef PX_LOAD_IMAGE
#undef PX_COMMIT
#undef PX_COMMIT1
#undef PX_FETCH
#undef PX_FETCH1
}

}  // namespace

// ---- host side.  Geometry: the trunk form of kernels_conv2d_ws.hip (no frame, no
// exogenous channel, no few-feature output conv)
bool conv2d_ws_x3_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res) {
  if (precision != S3_PREC_BF16X3 || s3_opt_on(S3O_NO_CONV2D_WS) || s3_opt_on(S3O_NO_WS_X3)) return false;
  if (io.in_bf16 || io.out_bf16 || (has_res && io.res_bf16)) return false;
  if (g.w_cin || g.res2 || (has_res && g.d2s > 1)) return false;
  // per IMAGE (the choice must not depend on the batch size): a 16 x 16 image is ONE tile per
  // image — 60 tiles of two ~12 us rounds each on 256 CUs lose to the logical-axes kernel's
  // finer split (sup3rcc/gen_solar_5x_1x_1f at (60, 16, 16): 0.64 vs 0.97 ms per forward; spatial/gen_10x_2f at
  // (48, 20, 20): 1.30 vs 1.56 ms; from 32 x 32 cells on the weights-stationary form wins)
  const int64_t min_pos = s3_opt_int(S3O_WS_X3_MIN_POS, 1024);
  if ((int64_t)g.D[0] * g.D[1] < min_pos) return false;
  return conv2d_ws_geom_ok(g) && !conv2d_ws_tail_geom_ok(g);
}

size_t conv2d_ws_x3_image_bytes(const ConvGeom& g) { return (size_t)((g.Cout + 63) / 64) * 2 * X_PASS_BYTES; }

int launch_conv2d_ws_x3_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image) {
  const int n_ct = (g.Cout + 63) / 64;
  int grid = (n_ct * 2 * 9 * 64 * 32 + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_ws_x3_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, (unsigned short*)image, g.Cout,
                     n_ct);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv2d_ws_x3(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image, const float* bias,
                        const void* res, void* y) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_x3_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_x3_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_x3_pp_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_x3_pp_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS));
    attr_set.mark(ctx->device);
  }
  X3Geom w;
  w.N = g.N; w.H = g.D[0]; w.W = g.D[1];
  w.Cout = g.Cout; w.b = g.d2s < 1 ? 1 : g.d2s; w.cpo = g.Cout / (w.b * w.b);
  w.act = g.act; w.alpha = g.alpha;
  w.dbg = (int)s3_opt_int(S3O_MFMA_DBG, 0);
  w.tiles_i = (g.N + XT_I - 1) / XT_I;
  w.tiles_r = (w.H + XT_R - 1) / XT_R; w.tiles_c = (w.W + XT_C - 1) / XT_C;
  const int T = w.tiles_i * w.tiles_r * w.tiles_c, n_ct = (g.Cout + 63) / 64;
  int gx = ctx->num_cu / n_ct;
  if (gx > T) gx = T;
  if (gx < 1) gx = 1;
  if (!s3_opt_on(S3O_NO_WS_PP)) {
    // the ping-pong form: single-image tiles, two half-workgroups half a round apart
    const int T1 = g.N * w.tiles_r * w.tiles_c;
    int gp = ctx->num_cu / n_ct;
    if (gp > (T1 + 1) / 2) gp = (T1 + 1) / 2;
    if (gp < 1) gp = 1;
    auto kern = res ? conv2d_ws_x3_pp_kernel<true> : conv2d_ws_x3_pp_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(gp, n_ct), dim3(X_NT), X_LDS, ctx->stream, (const float*)x, (const char*)image,
                       bias, (const float*)res, (float*)y, w);
  } else {
    auto kern = res ? conv2d_ws_x3_kernel<true> : conv2d_ws_x3_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(gx, n_ct), dim3(X_NT), X_LDS, ctx->stream, (const float*)x, (const char*)image,
                       bias, (const float*)res, (float*)y, w);
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
