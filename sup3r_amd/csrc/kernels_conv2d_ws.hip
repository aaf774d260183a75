// Weights-stationary persistent Conv2D for the bf16 trunks of the 2-D
// generators (sup3r/configs/spatial/gen_*: Conv2DTranspose 64 -> 64 x 33 + the
// 64 -> 256 / 1600 expansion convs; sup3rcc/gen_{solar,wind}_5x_1x_*: Conv2D),
// gfx950 only.  bf16 cells in and out, 3 x 3, stride 1, 'same' extents with
// REFLECT boundary (the fused pad / conv / crop group), C_in = 64, C_out a
// multiple of 64.
//
// Why not the halo-tile kernel (conv_mfma_tile.h): with 9 taps instead of 27 a
// tile's MFMA time (0.43 us per tap) is shorter than the L2 latency of the
// next tap's 8 KB filter slab, so its one-barrier-per-tap ring is bound by
// nine exposed slab loads per tile (measured: 269 TFLOP/s on the 75 x 75 x 48
// chunk of config_fwp_spatial.json).  But nine slabs are only 72 KB: they fit
// in LDS next to one halo.  So here
//
//   * ONE 8-wave workgroup per CU loads the 9 x [64 x 64] bf16 filter image of
//     its 64-channel output tile ONCE and keeps it (weights-stationary), then
//     walks a contiguous list of 2 images x 16 rows x 16 columns position
//     tiles (the time axis of a ForwardPass chunk is the batch axis of a 2-D
//     net: /root/reference/sup3r/pipeline/forward_pass.py:274-337);
//   * the NEXT tile's 2 x 18 x 18 halo (81 KB) is fetched into registers (11
//     16-B loads per lane) before the 9-tap loop and dropped into LDS after
//     it: HBM latency hides under the MFMAs, no barrier inside the tap loop;
//   * MFMA operands are swapped as in kernels_conv_mfma_persist.hip (A =
//     filter rows in a permuted order, B = positions): a lane's accumulators
//     are 8 consecutive output channels of one position per pair of N
//     fragments — bias, activation, skip add and the 16-B bf16 store happen
//     from registers (depth-to-space is a store permutation: C_out / b^2 is a
//     multiple of 8 wherever the plan stores bf16).
//
// LDS: halo 648 cells x 128 B | 9 slabs x 8 KB | 64 biases = 156,928 B.
// Swizzles as in conv_mfma_tile.h (halo chunk ^ (cell column & 7), slab chunk
// ^ ((row >> 1) & 7)): every ds_read_b128 is conflict-free.  Per wave and
// (tap, k-step): 4 filter + 4 position fragments for 16 MFMAs — the 1 : 2
// LDS-read : MFMA issue ratio of the 3-D trunk kernel.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int WT_I = 2, WT_R = 16, WT_C = 16;       // tile: images x rows x columns
constexpr int WH_R = WT_R + 2, WH_C = WT_C + 2;     // 18 x 18 halo per image
constexpr int WHP = WT_I * WH_R * WH_C;             // 648 cells
constexpr int W_HALO_BYTES = WHP * 128;             // 82,944
constexpr int W_SLAB_OFF = W_HALO_BYTES;
constexpr int W_BIAS_OFF = W_SLAB_OFF + 9 * 8192;   // 156,672
constexpr int W_LDS = W_BIAS_OFF + 256;             // 156,928
// EXO form: + the exogenous channel's halo (648 fp32) and its 9 x 64 filter taps
constexpr int W_EXO_OFF = W_LDS;
constexpr int W_WE_OFF = W_EXO_OFF + WHP * 4;       // 159,520
constexpr int W_LDS_EXO = W_WE_OFF + 9 * 64 * 4;    // 161,824
constexpr int W_NT = 512;

__device__ inline unsigned ws_pk(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float ws_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float ws_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// LDS slab row rho = nf*16 + kq*4 + r  <->  output channel
// (nf >> 1)*32 + kq*8 + (nf & 1)*4 + r  (kernels_conv_mfma_persist.hip)
__device__ __host__ inline int ws_row_cout(int rho) {
  const int nf = rho >> 4, kq = (rho >> 2) & 3, r = rho & 3;
  return (nf >> 1) * 32 + kq * 8 + (nf & 1) * 4 + r;
}

// canonical fp32 w[tap 9][ci 64][co] -> bf16 images [ct][tap][rho 64][ci 64]
// tail (C_out <= 16, one N fragment): rows 0 .. 15 are the channels in order
// w_cin: channels of the canonical weights' C_in axis (64, or 65 with an
// exogenous channel behind the 64: its taps go out as fp32 values of the bf16
// roundings, [ct][tap][64 channels in natural order], behind the images)
__global__ void pack_ws_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout,
                               int n_ct, int tail, int w_cin) {
  const int total = n_ct * 9 * 64 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ci = idx & 63, rho = (idx >> 6) & 63, tap = (idx >> 12) % 9, ct = (idx >> 12) / 9;
    const int co = tail ? (rho < 16 ? rho : cout) : ct * 64 + ws_row_cout(rho);
    const float v = co < cout ? w[((size_t)tap * w_cin + ci) * cout + co] : 0.f;
    const int slot = (ci >> 3) ^ ((rho >> 1) & 7);
    out[(((size_t)ct * 9 + tap) * 64 + rho) * 64 + slot * 8 + (ci & 7)] = (unsigned short)(ws_pk(v, 0.f) & 0xFFFFu);
  }
  if (w_cin > 64) {
    float* we = reinterpret_cast<float*>(out + (size_t)total);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_ct * 9 * 64; idx += gridDim.x * blockDim.x) {
      const int c = idx & 63, tap = (idx >> 6) % 9, ct = (idx >> 6) / 9;
      const int co = ct * 64 + c;
      const float v = co < cout ? w[((size_t)tap * w_cin + 64) * cout + co] : 0.f;
      we[idx] = __uint_as_float(ws_pk(v, 0.f) << 16);
    }
  }
}

struct WsGeom {
  int N, H, W;         // images, rows, columns
  int Cout, b, cpo;    // output channels of the conv, depth-to-space block, Cout / b^2
  int act;
  float alpha;
  int tiles_i, tiles_r, tiles_c;
  int dbg;             // option MFMA_DBG (ablations): 1 no halo prefetch, 2 no tap loop, 4 no epilogue
  // 1: the data gradient of such a conv — the full correlation of dPre (N, H, W)
  // with the flipped / transposed filter over the frame padded by one cell,
  // zero boundary: out (N, H + 2, W + 2), out[r] = sum_t w[t] in[r - 2 + t]
  int frame;
};

// NF = 4: 64 output channels per workgroup, bf16 cells out (the trunk form).
// NF = 1: the few-feature OUTPUT conv of a 2-D generator (64 -> C_out <= 16, fp32
// out, no skip, no depth-to-space): one filter fragment per tap, channels in
// natural order, scalar fp32 stores — read-bound (5 fragment reads per 4 MFMAs),
// but the 150 x 150 x 48 output conv of gen_2x_2f drops from 178 us on the
// one-barrier-per-tap tile kernel to a pass at the speed its input streams.
// EXO: + one exogenous fp32 channel (see ConvGeom::w_cin): its 2 x 18 x 18 halo
// rides along the cell halo (two more prefetch registers, bf16-rounded when it
// lands in LDS — the rounding the 65-channel conv's staging applied), its nine
// taps per output are 576 fused multiply-adds per lane and tile on the vector
// unit between the MFMA loop and the hand-over.
// RES: a skip operand (res) is added in the epilogue — a template parameter so
// that the loads of its rows and their use are unconditional code (see
// conv2d_ws_tile.inc on why that matters).
template <int NF, bool EXO = false, bool RES = false>
__global__ __launch_bounds__(W_NT) void conv2d_ws_kernel(
    const unsigned short* __restrict__ x, const char* __restrict__ wimg, const float* __restrict__ bias,
    const unsigned short* __restrict__ res, void* __restrict__ yv, WsGeom g, const float* __restrict__ exo,
    const unsigned short* __restrict__ res2) {
  unsigned short* __restrict__ y = reinterpret_cast<unsigned short*>(yv);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.y;
  const int frow = lane & 15, kq = lane >> 4;

  // ---- this workgroup's contiguous run of tiles (neighbours share halo
  // columns: the second read of a column hits this XCD's L2)
  // Ranks are XCD-major: workgroups are dealt to the 8 XCDs round-robin by their
  // linear id, so the workgroups of one XCD take CONSECUTIVE runs — the halo
  // columns two neighbouring runs share are fetched into that XCD's L2 once
  // (with rank = blockIdx.x the neighbour sat on another XCD and fetched its
  // own copy: FETCH_SIZE 1.69 x the input at the config_fwp_spatial chunk).
  const int T = g.tiles_i * g.tiles_r * g.tiles_c;
  int rank;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    const int o = (int)((blockIdx.y * gridDim.x) & 7);   // XCD of this row's first workgroup
    const int xcd = (b + o) & 7;
    rank = 0;
    // workgroups of this row on the XCDs dealt before this one (in the order o, o + 1, ..)
    for (int k = 0; k < ((xcd - o) & 7); ++k) rank += (nb - k + 7) / 8;
    rank += (b - ((xcd - o) & 7)) / 8;
  }
  int t_cur = (int)(((long long)rank * T) / gridDim.x);
  const int t_end = (int)(((long long)(rank + 1) * T) / gridDim.x);
  if (t_cur >= t_end) return;

  // ---- the filter image of this output-channel tile: 72 KB, once.  Its nine
  // 16-B loads per lane are issued here and land in LDS AFTER the first halo's
  // eleven have been issued as well: one memory latency at kernel start, not two.
  uint4 wimg_r[9];
  {
    const uint4* src = reinterpret_cast<const uint4*>(wimg + (size_t)ct * 9 * 8192);
#pragma unroll
    for (int q = 0; q < 9; ++q) wimg_r[q] = src[tid + q * W_NT];
  }

  // ---- origin of tile t (t is uniform: scalar divisions)
  auto tile_org = [&](int t, int& i0, int& r0, int& c0) __attribute__((always_inline)) {
    c0 = (t % g.tiles_c) * WT_C; t /= g.tiles_c;
    r0 = (t % g.tiles_r) * WT_R; t /= g.tiles_r;
    i0 = t * WT_I;
  };
  // (macros, not lambdas: the prefetch buffer is a loop-local array that must
  // stay in registers — captured by a lambda it was demoted to scratch, and the
  // scratch store waited for every load on the spot)
  //
  // Lane -> halo chunk.  A trip covers THREE consecutive halo rows: lane = (sub-row
  // j, column, 16-B chunk) = 3 x 18 x 8 = 432 lanes (the other 80 shadow sub-row 2:
  // same address, same data, same LDS slot), trip q = rows 3 q .. 3 q + 2 of the 36
  // (2 images x 18) — so a lane's COLUMN is the same in all twelve trips: reflect /
  // clamp / zero flag / LDS swizzle of the column are evaluated once per tile, the
  // image of a trip is a compile-time constant (18 = 6 x 3), and a trip costs the
  // row's reflect and one multiply-add.  (Before: chunk id tid + 512 q decoded by
  // two divisions per trip, ~50 instructions per 16-B load — with the epilogue's
  // index arithmetic 3 us of vector-ALU work per 10 us tile, serial with the MFMAs.)
  const int lt = tid < 3 * WH_C * 8 ? tid : tid - WH_C * 8;
  const int h_j = lt / (WH_C * 8), h_col = (lt - h_j * (WH_C * 8)) >> 3;
  const unsigned lds_lane = (unsigned)((h_j * WH_C + h_col) * 128 + (((tid & 7) ^ (h_col & 7)) << 4));
#define WS_FETCH1(P, q)                                                                         \
  {                                                                                             \
    constexpr int up_ = (3 * q >= WH_R) ? 1 : 0;            /* second image of the tile */      \
    const int rv_ = r0_ + 3 * q - up_ * WH_R + h_j - 1 - g.frame;                               \
    int r_ = g.frame ? rv_ : s3_reflect(rv_, g.H);                                              \
    const bool z_ = zc_ || (g.frame && (rv_ < 0 || rv_ >= g.H));                               \
    /* ragged tiles: the address stays legal (masked at the store) */                          \
    int im_ = i0_ + up_;                                                                        \
    im_ = im_ > g.N - 1 ? g.N - 1 : im_;                                                        \
    r_ = r_ < 0 ? 0 : (r_ > g.H - 1 ? g.H - 1 : r_);                                            \
    const unsigned rowcell_ = ((unsigned)im_ * g.H + r_) * g.W;   /* < 2^31 */                  \
    P = *reinterpret_cast<const uint4*>(x + (size_t)rowcell_ * 64 + colel_);                    \
    /* (zero boundary of the frame: only FLAGGED here and zeroed by WS_COMMIT1 — a select on */ \
    /* the loaded value at this point is a wait for the load in front of the tap loop) */       \
    zmask |= z_ ? (1u << q) : 0u;                                                               \
  }
  // (named registers, not an array: an array that lives across the tap loop
  // was demoted to scratch, and each scratch store waited for its load)
#define WS_FETCH(T)                                                                             \
  {                                                                                             \
    int i0_, r0_, c0_;                                                                          \
    tile_org((T), i0_, r0_, c0_);                                                               \
    const int cv_ = c0_ + h_col - 1 - g.frame;                                                  \
    int c_ = g.frame ? cv_ : s3_reflect(cv_, g.W);                                              \
    const bool zc_ = g.frame && (cv_ < 0 || cv_ >= g.W);                                       \
    c_ = c_ < 0 ? 0 : (c_ > g.W - 1 ? g.W - 1 : c_);                                            \
    const unsigned colel_ = (unsigned)c_ * 64 + (tid & 7) * 8;                                  \
    zmask = 0u;                                                                                 \
    WS_FETCH1(p0, 0) WS_FETCH1(p1, 1) WS_FETCH1(p2, 2) WS_FETCH1(p3, 3) WS_FETCH1(p4, 4)        \
    WS_FETCH1(p5, 5) WS_FETCH1(p6, 6) WS_FETCH1(p7, 7) WS_FETCH1(p8, 8) WS_FETCH1(p9, 9)        \
    WS_FETCH1(p10, 10) WS_FETCH1(p11, 11)                                                       \
  }
#define WS_COMMIT1(P, q)                                                                        \
  *reinterpret_cast<uint4*>(smem + lds_lane + q * (3 * WH_C * 128)) =                           \
      ((zmask >> q) & 1u) ? make_uint4(0, 0, 0, 0) : P;
#define WS_COMMIT()                                                                             \
  {                                                                                             \
    WS_COMMIT1(p0, 0) WS_COMMIT1(p1, 1) WS_COMMIT1(p2, 2) WS_COMMIT1(p3, 3) WS_COMMIT1(p4, 4)   \
    WS_COMMIT1(p5, 5) WS_COMMIT1(p6, 6) WS_COMMIT1(p7, 7) WS_COMMIT1(p8, 8) WS_COMMIT1(p9, 9)   \
    WS_COMMIT1(p10, 10) WS_COMMIT1(p11, 11)                                                     \
  }
  static_assert(WT_I == 2 && WH_R % 3 == 0 && W_NT >= 3 * WH_C * 8 && W_NT - 3 * WH_C * 8 <= WH_C * 8,
                "twelve trips of three halo rows: prefetch registers p0 .. p11");
  uint4 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11;
  unsigned zmask = 0u;   // bit q: chunk q of the prefetched halo lies on the frame's zero boundary
  // exogenous channel: cell tid (+ 512 for tid < 136) of the same halo
#define WS_EFETCH1(P, q, i0_, r0_, c0_)                                                         \
  {                                                                                             \
    int cl_ = tid + q * W_NT;                                                                   \
    cl_ = cl_ > WHP - 1 ? WHP - 1 : cl_;                                                        \
    int im_ = i0_ + cl_ / (WH_C * WH_R);                                                        \
    int r_ = s3_reflect(r0_ + (cl_ / WH_C) % WH_R - 1, g.H);                                    \
    int c_ = s3_reflect(c0_ + cl_ % WH_C - 1, g.W);                                             \
    im_ = im_ > g.N - 1 ? g.N - 1 : im_;                                                        \
    r_ = r_ < 0 ? 0 : (r_ > g.H - 1 ? g.H - 1 : r_);                                            \
    c_ = c_ < 0 ? 0 : (c_ > g.W - 1 ? g.W - 1 : c_);                                            \
    P = exo[((size_t)im_ * g.H + r_) * g.W + c_];                                               \
  }
#define WS_EFETCH(T)                                                                            \
  if constexpr (EXO) {                                                                          \
    int i0_, r0_, c0_;                                                                          \
    tile_org((T), i0_, r0_, c0_);                                                               \
    WS_EFETCH1(pe0, 0, i0_, r0_, c0_) WS_EFETCH1(pe1, 1, i0_, r0_, c0_)                         \
  }
#define WS_ECOMMIT()                                                                            \
  if constexpr (EXO) {                                                                          \
    float* E_ = reinterpret_cast<float*>(smem + W_EXO_OFF);                                     \
    E_[tid] = __uint_as_float(ws_pk(pe0, 0.f) << 16);                                           \
    E_[tid + W_NT < WHP ? tid + W_NT : WHP - 1] = __uint_as_float(ws_pk(pe1, 0.f) << 16);       \
  }
  float pe0 = 0.f, pe1 = 0.f;
  WS_FETCH(t_cur);
  WS_EFETCH(t_cur);
  if constexpr (EXO) {
    const float* wsrc = reinterpret_cast<const float*>(wimg + (size_t)gridDim.y * 9 * 8192) + (size_t)ct * 576;
    float* WE = reinterpret_cast<float*>(smem + W_WE_OFF);
    for (int i = tid; i < 576; i += W_NT) WE[i] = wsrc[i];
  }
  {
    uint4* dst = reinterpret_cast<uint4*>(smem + W_SLAB_OFF);
#pragma unroll
    for (int q = 0; q < 9; ++q) dst[tid + q * W_NT] = wimg_r[q];
    if (tid < 64) {
      const int co = ct * 64 + tid;
      reinterpret_cast<float*>(smem + W_BIAS_OFF)[tid] = (bias && co < g.Cout) ? bias[co] : 0.f;
    }
  }
  WS_COMMIT();
  WS_ECOMMIT();
  __syncthreads();

  // ---- fragment addresses: wave w = image w >> 2, rows 4 (w & 3) .. + 3
  // position fragment (B operand) of row m, tap (tb, tc), k-step ks:
  //   cell (img, 4 (w & 3) + m + tb, frow + tc), chunk (ks 4 + kq) ^ ((frow + tc) & 7)
  // filter fragment (A operand) nf of tap: row nf 16 + frow, chunk (ks 4 + kq) ^ ((row >> 1) & 7)
  const int w_img = wave >> 2, w_row = (wave & 3) * 4;
  // (the slab swizzle (row >> 1) & 7 does not depend on nf — rows nf 16 + frow — so fragment nf
  // sits 2048 B behind fragment 0: two address registers instead of eight)
  unsigned p_addr[3][2], f_addr[2];
#pragma unroll
  for (int tc = 0; tc < 3; ++tc)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      p_addr[tc][ks] = (unsigned)(((w_img * WH_R + w_row) * WH_C + frow + tc) * 128 +
                                  (((ks * 4 + kq) ^ ((frow + tc) & 7)) << 4));
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    f_addr[ks] = (unsigned)(W_SLAB_OFF + frow * 128 + (((ks * 4 + kq) ^ ((frow >> 1) & 7)) << 4));
  // this lane's output channels: h 32 + kq 8 .. + 7 for h = 0, 1 (inside the tile)
  const float* bl = reinterpret_cast<const float*>(smem + W_BIAS_OFF);
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  // output addressing (trunk form), set up once: element index of (image im, row r, column c,
  // this lane's chunk h) = im IS + r RS + c CS + off_h[h] — with a depth-to-space store
  // (block b, cpo = C_out / b^2 channels per hi-res cell) chunk h is block (bi, bj), channel cq
  // of the hi-res cell (r b + bi, c b + bj).  Per tile one 64-bit base, per row one multiply-add,
  // per chunk one add (the index arithmetic of the epilogue was ~100 instructions per store).
  const int Ho = g.H + 2 * g.frame, Wo = g.W + 2 * g.frame;   // output extents
  const unsigned rsd = (unsigned)(Wo * g.b * g.cpo);           // elements per output row
  const unsigned long long IS = (unsigned long long)(Ho * g.b) * rsd;
  const unsigned RS = (unsigned)g.b * rsd, CS = (unsigned)(g.b * g.cpo);
  unsigned off_h[2];
  bool ch_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int co_raw = ct * 64 + h * 32 + kq * 8;
    ch_ok[h] = co_raw < g.Cout;
    const int co = co_raw > g.Cout - 8 ? (g.Cout >= 8 ? g.Cout - 8 : 0) : co_raw;
    const int blk = co / g.cpo, cq = co % g.cpo;
    off_h[h] = (unsigned)(blk / g.b) * rsd + (unsigned)((blk % g.b) * g.cpo + cq);
  }

  for (; t_cur + 1 < t_end; ++t_cur) {
#define WS_HAS_NEXT 1
#include "conv2d_ws_tile.inc"
#undef WS_HAS_NEXT
  }
#define WS_HAS_NEXT 0
#include "conv2d_ws_tile.inc"
#undef WS_HAS_NEXT
}

#undef WS_EFETCH
#undef WS_EFETCH1
#undef WS_ECOMMIT
#undef WS_FETCH
#undef WS_FETCH1
#undef WS_COMMIT
#undef WS_COMMIT1

// ---------------------------------------------------------------------------
// Ping-pong form of the trunk conv (64 output channels per workgroup, bf16
// cells out, no exogenous channel).
//
// Why.  In conv2d_ws_kernel all eight waves walk the same phases together:
// prefetch issue, taps, hand-over, epilogue.  Ablations of that kernel at 16
// tiles per CU (tools/dbg/ws_scaling.py, MFMA_DBG): the tap loop alone is 4.6 us
// per 2-image tile (the MFMA pipes' time: 288 MFMAs x 2 waves per SIMD), the
// prefetch adds 1.4 - 1.8 us, the output stores 1.1 us, hand-over + index
// arithmetic 2.7 us, and the whole tile is the SUM, 11.3 us: a wave issues in
// order, so the 96 + 64 vector-memory instructions of a tile (~16 clocks each
// through the CU's one address unit), the 96 ds_write_b128 of the hand-over and
// the vector-ALU work all sit between two tap loops, with the matrix cores idle
// (41 % busy).  Removing waits or instructions from those phases moves little
// (see conv2d_ws_tile.inc); what helps is other waves running MFMAs meanwhile.
//
// How.  The two images of a tile were already independent (waves 0-3 / 4-7,
// separate halves of the LDS halo).  Here each half-workgroup ("group") owns
// its own list of single-image tiles and the groups run half a period apart:
//
//   group 0:  T0 | M0 | T1 | M1 | ...          T = the 9-tap MFMA loop of a tile
//   group 1:     | T0 | M0 | T1 | M1 | ...     M = hand-over of the next halo, epilogue +
//                                                  stores of this tile, prefetch issue
//
// with ONE workgroup barrier per phase (a group's halo is read in its T and
// rewritten in its M, always a barrier apart; the filter image is read-only).
// While one group's wave holds a SIMD's matrix core, the other group's wave on
// that SIMD issues memory / LDS / vector-ALU instructions.
//
// All global loads are hand-ordered (asm volatile, invisible to the compiler's
// waitcnt pass, as in kernels_conv_mfma_persist.hip): prefetch registers are
// written in one M phase and read two phases later behind ONE explicit
// s_waitcnt vmcnt(0) at the top of that M — where everything outstanding (that
// prefetch, the skip rows fetched at the top of T, the previous M's stores) is
// at least a phase old.  tests/test_abi.py checks the compiled kernel for
// scratch use and for any touch of an in-flight register.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline u32x4 pp_ld16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
constexpr int PH_BYTES = WH_R * WH_C * 128;     // 41,472: one image's halo

#define PP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool RES>
__global__ __launch_bounds__(W_NT) void conv2d_ws_pp_kernel(
    const unsigned short* __restrict__ x, const char* __restrict__ wimg, const float* __restrict__ bias,
    const unsigned short* __restrict__ res, unsigned short* __restrict__ y, WsGeom g,
    const unsigned short* __restrict__ res2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                 // half-workgroup: waves 0-3 / 4-7
  const int gt = tid & 255;                  // thread within the group
  const int ct = blockIdx.y;
  const int frow = lane & 15, kq = lane >> 4;

  // ---- this workgroup's run of SINGLE-image tiles, XCD-major (conv2d_ws_kernel);
  // group g takes tiles t0 + g, t0 + g + 2, ..: the groups work on neighbours
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
    c0 = (t % g.tiles_c) * WT_C; t /= g.tiles_c;
    r0 = (t % g.tiles_r) * WT_R;
    im = t / g.tiles_r;
  };

  uint4 wimg_r[9];
  {
    const uint4* src = reinterpret_cast<const uint4*>(wimg + (size_t)ct * 9 * 8192);
#pragma unroll
    for (int q = 0; q < 9; ++q) wimg_r[q] = src[tid + q * W_NT];
  }

  // ---- lane -> halo chunk of the group's 18 x 18 cells x 8 chunks = 2592 = 10.1 x 256:
  //   trips 0 .. 8: rows 2 q + (gt >> 7), columns 0 .. 15   (column (gt >> 3) & 15)
  //   trip 9:       rows 0 .. 15 (gt >> 4), columns 16, 17   (column 16 + ((gt >> 3) & 1))
  //   trip 10:      rows 16, 17, columns 16, 17 — 32 lanes; the others shadow lane gt & 31
  // so that a lane has two columns per tile (reflect / clamp / zero flag once each)
  const int hcA = (gt >> 3) & 15, hcB = 16 + ((gt >> 3) & 1);
  const unsigned gb = (unsigned)(grp * PH_BYTES);
  const unsigned ldsA = gb + (unsigned)(((gt >> 7) * WH_C + hcA) * 128 + (((gt & 7) ^ (hcA & 7)) << 4));
  const unsigned ldsB = gb + (unsigned)(((gt >> 4) * WH_C + hcB) * 128 + (((gt & 7) ^ (hcB & 7)) << 4));
  const unsigned ldsC = gb + (unsigned)(((16 + ((gt & 31) >> 4)) * WH_C + hcB) * 128 + (((gt & 7) ^ (hcB & 7)) << 4));
  u32x4 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10;
  unsigned zmask = 0u;   // bit q: chunk q lies on the frame's zero boundary
#define PP_COL(HC, COLEL, ZC)                                                                   \
    unsigned COLEL;                                                                             \
    bool ZC;                                                                                    \
    {                                                                                           \
      const int cv_ = c0_ + (HC) - 1 - g.frame;                                                 \
      int c_ = g.frame ? cv_ : s3_reflect(cv_, g.W);                                            \
      ZC = g.frame && (cv_ < 0 || cv_ >= g.W);                                                  \
      c_ = c_ < 0 ? 0 : (c_ > g.W - 1 ? g.W - 1 : c_);                                          \
      COLEL = (unsigned)c_ * 64 + (gt & 7) * 8;                                                 \
    }
// address of trip q's chunk (and its zero flag into zmask); the load itself is PP_LOADQ
#define PP_ADDRQ(q)                                                                             \
    const unsigned short* a##q##_;                                                              \
    {                                                                                           \
      const int hr_ = (q) <= 8 ? 2 * (q) + jr_ : ((q) == 9 ? (gt >> 4) : 16 + ((gt & 31) >> 4)); \
      const int rv_ = r0_ + hr_ - 1 - g.frame;                                                  \
      int r_ = g.frame ? rv_ : s3_reflect(rv_, g.H);                                            \
      const bool z_ = ((q) <= 8 ? zcA_ : zcB_) || (g.frame && (rv_ < 0 || rv_ >= g.H));        \
      r_ = r_ < 0 ? 0 : (r_ > g.H - 1 ? g.H - 1 : r_);                                          \
      const unsigned rowcell_ = ((unsigned)im_ * g.H + r_) * g.W;   /* < 2^31 */                \
      a##q##_ = x + (size_t)rowcell_ * 64 + ((q) <= 8 ? colA_ : colB_);                         \
      zmask |= z_ ? (1u << (q)) : 0u;                                                           \
    }
#define PP_LOADQ(q) p##q = pp_ld16(a##q##_);
#define PP_TRIP(q) { PP_ADDRQ(q) PP_LOADQ(q) }
// per-tile part of a prefetch: origin, the lane's two columns (declares im_, r0_, colA_, ..)
#define PP_FETCH_PREP(T_)                                                                       \
    int im_, r0_, c0_;                                                                          \
    tile_org((T_), im_, r0_, c0_);                                                              \
    PP_COL(hcA, colA_, zcA_)                                                                    \
    PP_COL(hcB, colB_, zcB_)                                                                    \
    const int jr_ = gt >> 7;                                                                    \
    zmask = 0u;
#define PP_FETCH(T_)                                                                            \
  {                                                                                             \
    PP_FETCH_PREP(T_)                                                                           \
    PP_TRIP(0) PP_TRIP(1) PP_TRIP(2) PP_TRIP(3) PP_TRIP(4) PP_TRIP(5) PP_TRIP(6) PP_TRIP(7)     \
    PP_TRIP(8) PP_TRIP(9) PP_TRIP(10)                                                           \
  }
#define PP_SEL(P, q) (((zmask >> q) & 1u) ? (u32x4){0u, 0u, 0u, 0u} : P)
#define PP_COMMIT()                                                                             \
  {                                                                                             \
    *reinterpret_cast<u32x4*>(smem + ldsA + 0 * 2 * WH_C * 128) = PP_SEL(p0, 0);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 1 * 2 * WH_C * 128) = PP_SEL(p1, 1);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 2 * 2 * WH_C * 128) = PP_SEL(p2, 2);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 3 * 2 * WH_C * 128) = PP_SEL(p3, 3);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 4 * 2 * WH_C * 128) = PP_SEL(p4, 4);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 5 * 2 * WH_C * 128) = PP_SEL(p5, 5);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 6 * 2 * WH_C * 128) = PP_SEL(p6, 6);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 7 * 2 * WH_C * 128) = PP_SEL(p7, 7);                \
    *reinterpret_cast<u32x4*>(smem + ldsA + 8 * 2 * WH_C * 128) = PP_SEL(p8, 8);                \
    *reinterpret_cast<u32x4*>(smem + ldsB) = PP_SEL(p9, 9);                                     \
    *reinterpret_cast<u32x4*>(smem + ldsC) = PP_SEL(p10, 10);                                   \
  }
  // everything outstanding has landed; the "+v" operands pin every later use of
  // the prefetch / skip registers behind this statement (volatile asm keeps
  // program order with the loads above and with the barriers)
#define PP_WAIT_P()                                                                             \
  asm volatile("s_waitcnt vmcnt(0)"                                                             \
               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6),          \
                 "+v"(p7), "+v"(p8), "+v"(p9), "+v"(p10)                                        \
               :: "memory")
#define PP_WAIT_ALL()                                                                           \
  {                                                                                             \
    PP_WAIT_P();                                                                                \
    if constexpr (RES)                                                                          \
      asm volatile("" : "+v"(rr[0][0]), "+v"(rr[0][1]), "+v"(rr[1][0]), "+v"(rr[1][1]),         \
                        "+v"(rr[2][0]), "+v"(rr[2][1]), "+v"(rr[3][0]), "+v"(rr[3][1])          \
                   :: "memory");                                                                \
  }

  u32x4 rr[4][2];

  // ---- prologue: filter image + biases (all waves), each group's first halo,
  // the prefetch of its second
  // (a group without a tile — group 1 of a one-tile run — fetches group 0's: no
  // conditional definition of the prefetch registers)
  PP_FETCH(t0 + (n_g > 0 ? grp : 0));
  {
    uint4* dst = reinterpret_cast<uint4*>(smem + W_SLAB_OFF);
#pragma unroll
    for (int q = 0; q < 9; ++q) dst[tid + q * W_NT] = wimg_r[q];
    if (tid < 64) {
      const int co = ct * 64 + tid;
      reinterpret_cast<float*>(smem + W_BIAS_OFF)[tid] = (bias && co < g.Cout) ? bias[co] : 0.f;
    }
  }
  PP_WAIT_P();
  PP_COMMIT();
  if (n_g > 1) PP_FETCH(t0 + grp + 2);
  PP_BARRIER();

  // ---- fragment addresses (conv2d_ws_kernel): wave w of the group = rows 4 (w & 3) .. + 3
  const int w_row = (wave & 3) * 4;
  unsigned p_addr[3][2], f_addr[2];
#pragma unroll
  for (int tc = 0; tc < 3; ++tc)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      p_addr[tc][ks] = gb + (unsigned)((w_row * WH_C + frow + tc) * 128 + (((ks * 4 + kq) ^ ((frow + tc) & 7)) << 4));
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    f_addr[ks] = (unsigned)(W_SLAB_OFF + frow * 128 + (((ks * 4 + kq) ^ ((frow >> 1) & 7)) << 4));
  // (opaque to the compiler: with the constant W_SLAB_OFF visible it splits it off and re-adds
  // it per fragment — four v_add per step — instead of using the ds_read offset field)
  asm volatile("" : "+v"(f_addr[0]), "+v"(f_addr[1]));
  const float* bl = reinterpret_cast<const float*>(smem + W_BIAS_OFF);
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  // output addressing as in conv2d_ws_kernel: im IS + r RS + c CS + off_h[h]
  const int Ho = g.H + 2 * g.frame, Wo = g.W + 2 * g.frame;
  const unsigned rsd = (unsigned)(Wo * g.b * g.cpo);
  const unsigned long long IS = (unsigned long long)(Ho * g.b) * rsd;
  const unsigned RS = (unsigned)g.b * rsd, CS = (unsigned)(g.b * g.cpo);
  unsigned off_h[2];
  bool ch_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int co_raw = ct * 64 + h * 32 + kq * 8;
    ch_ok[h] = co_raw < g.Cout;
    const int co = co_raw > g.Cout - 8 ? (g.Cout >= 8 ? g.Cout - 8 : 0) : co_raw;
    const int blk = co / g.cpo, cq = co % g.cpo;
    off_h[h] = (unsigned)(blk / g.b) * rsd + (unsigned)((blk % g.b) * g.cpo + cq);
  }

  const bool lin_cells = g.b == 1 && g.cpo == 64 && g.Cout - ct * 64 >= 64;
  f32x4 acc[4][4];
  if (grp == 1) PP_BARRIER();      // half a period behind group 0
#pragma unroll 1
  for (int k = 0; k < n_max; ++k) {
    if (k >= n_g) {     // (group 1 of an odd run: keeps the barrier count)
      PP_BARRIER();
      PP_BARRIER();
      continue;
    }
    int im, r0, c0;
    tile_org(t0 + grp + 2 * k, im, r0, c0);
    const int c = c0 + frow;
    const bool col_ok = c < Wo;
    const int cc = c > Wo - 1 ? Wo - 1 : c;
    const unsigned long long tbase = (unsigned long long)im * IS + (unsigned long long)((unsigned)cc * CS);
    // (MFMA_DBG bit 128: the row's stores exchanged across the wave, one KB of consecutive bytes per
    // instruction — what took conv_dgrad_s2_kernel from 384 to 319 us is 4 % SLOWER here, 42 600
    // against 44 600 samples/s at the production chunk: this kernel's stores are not what its M phase
    // waits on.  The path stays because with it in the kernel the register allocator puts the position
    // fragments of the tap loop (MFMA operand B) at v[4k] where the filter fragments (A) and most
    // accumulators (C) sit at v[4k + 2]: without it 262 of the 288 MFMAs of a T phase read all three
    // operands from registers of the same residue mod 4 and the kernel runs 31.3 instead of 29.1 us per
    // conv — 42 350 against 44 600 samples/s in one call.  tools/dbg/mfma_banks.py prints that histogram.)
    const bool lin_st = lin_cells && c0 + 16 <= Wo && (g.dbg & 128);

    // =========================================================== T phase
    {
      if constexpr (RES) {
        // this tile's skip rows (d2s == 1, no frame: the output's layout), under the taps
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          int r = r0 + w_row + m;
          r = r > Ho - 1 ? Ho - 1 : r;
          const unsigned long long ridx = tbase + (unsigned long long)(unsigned)r * RS;
#pragma unroll
          for (int h = 0; h < 2; ++h) rr[m][h] = pp_ld16(res + ridx + off_h[h]);
        }
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int cb = (nf >> 1) * 32 + kq * 8 + (nf & 1) * 4;
        const f32x4 b4 = {bl[cb], bl[cb + 1], bl[cb + 2], bl[cb + 3]};
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][nf] = b4;
      }
      // 18 steps (9 taps x 2 k-halves) of 8 fragment reads + 16 MFMAs, unrolled and scheduled
      // BY HAND (sched_group_barrier): the fragments of step s + 1 are read during step s, one
      // ds_read behind each of its first eight MFMAs.  (Left to itself the compiler reads a
      // step's fragments right in front of its MFMAs, and issues them in one burst: with two
      // waves per SIMD in the tap loop the other wave filled those gaps, here a SIMD's matrix
      // core belongs to ONE wave during a T phase — 24 instead of 16 clocks per MFMA.)
      // (Tried: the prefetch of the group's next tile inside this loop — the index arithmetic of
      // one trip behind the last eight MFMAs of a step, its load behind the step: the T phase
      // grows by 0.8 us, more than the M phase loses; the prefetch stays in M.)
      if (!(g.dbg & 2)) {
        bf16x8 wf[2][4], pf[2][4];
#define PP_FRAGS(B, S)                                                                          \
        {                                                                                       \
          constexpr int tb_ = (S) / 6, tc_ = ((S) % 6) >> 1, ks_ = (S) & 1;                     \
          _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                      \
            wf[B][nf] = *reinterpret_cast<const bf16x8*>(smem + f_addr[ks_] + nf * 2048 + (tb_ * 3 + tc_) * 8192); \
          _Pragma("unroll") for (int m = 0; m < 4; ++m)                                         \
            pf[B][m] = *reinterpret_cast<const bf16x8*>(smem + p_addr[tc_][ks_] + (m + tb_) * WH_C * 128); \
        }
#define PP_MFMAS(B)                                                                             \
          _Pragma("unroll") for (int m = 0; m < 4; ++m)                                         \
            _Pragma("unroll") for (int nf = 0; nf < 4; ++nf)                                    \
              acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[B][nf], pf[B][m], acc[m][nf], 0, 0, 0);
#define PP_SCHED(NV)                                                                            \
          _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                  \
          }                                                                                     \
          _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
            if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                         \
          }                                                                                     \
          __builtin_amdgcn_sched_barrier(0);
#define PP_STEP(S)                                                                              \
        {                                                                                       \
          if constexpr ((S) + 1 < 18) PP_FRAGS(((S) + 1) & 1, ((S) + 1 < 18 ? (S) + 1 : 0))     \
          PP_MFMAS((S) & 1)                                                                     \
          PP_SCHED(0)                                                                           \
        }
        __builtin_amdgcn_s_setprio(2);   // over the other group's wave on this SIMD (VALU issue: priority, then age)
        PP_FRAGS(0, 0)
        PP_STEP(0) PP_STEP(1) PP_STEP(2) PP_STEP(3) PP_STEP(4) PP_STEP(5) PP_STEP(6) PP_STEP(7) PP_STEP(8)
        PP_STEP(9) PP_STEP(10) PP_STEP(11) PP_STEP(12) PP_STEP(13) PP_STEP(14) PP_STEP(15) PP_STEP(16) PP_STEP(17)
        __builtin_amdgcn_s_setprio(0);
#undef PP_STEP
#undef PP_SCHED
#undef PP_MFMAS
#undef PP_FRAGS
      }
    }
    PP_BARRIER();

    // =========================================================== M phase
    {
      PP_WAIT_ALL();
      if (k + 1 < n_g && !(g.dbg & 8)) PP_COMMIT();
      if (res2 && !(g.dbg & 16)) {
        // ---- a SECOND skip operand (d2s == 1): its eight chunks go into the prefetch
        // registers the hand-over above has just emptied — all in flight together, under the
        // first stage of the epilogue — instead of one load + wait per chunk inside it
        // (each of those waits also drained the stores before it: 6.9 ms where the
        // neighbouring convs take 4.9 at 96 x 750 x 750).
#define PP_RES2_LOAD(M, H, P)                                                                   \
        {                                                                                       \
          const int r_ = r0 + w_row + (M);                                                      \
          const int rc_ = r_ > Ho - 1 ? Ho - 1 : r_;                                            \
          P = pp_ld16(res2 + tbase + (unsigned long long)(unsigned)rc_ * RS + off_h[H]);        \
        }
        PP_RES2_LOAD(0, 0, p0) PP_RES2_LOAD(0, 1, p1) PP_RES2_LOAD(1, 0, p2) PP_RES2_LOAD(1, 1, p3)
        PP_RES2_LOAD(2, 0, p4) PP_RES2_LOAD(2, 1, p5) PP_RES2_LOAD(3, 0, p6) PP_RES2_LOAD(3, 1, p7)
#undef PP_RES2_LOAD
        // stage 1: activation + first skip, rounded to bf16 (as the separate add of two bf16
        // tensors saw it), parked in the accumulator registers it came from
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[m][2 * h][e]; v[4 + e] = acc[m][2 * h + 1][e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float sa = slope * v[e];
              asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(sa));
            }
            if constexpr (RES) {
              const u32x4 q4 = rr[m][h];
              v[0] += ws_lo(q4[0]); v[1] += ws_hi(q4[0]); v[2] += ws_lo(q4[1]); v[3] += ws_hi(q4[1]);
              v[4] += ws_lo(q4[2]); v[5] += ws_hi(q4[2]); v[6] += ws_lo(q4[3]); v[7] += ws_hi(q4[3]);
            }
            acc[m][2 * h][0] = __uint_as_float(ws_pk(v[0], v[1]));
            acc[m][2 * h][1] = __uint_as_float(ws_pk(v[2], v[3]));
            acc[m][2 * h][2] = __uint_as_float(ws_pk(v[4], v[5]));
            acc[m][2 * h][3] = __uint_as_float(ws_pk(v[6], v[7]));
          }
        PP_WAIT_P();
        // stage 2: + second skip, stores
#define PP_RES2_STORE(M, H, P)                                                                  \
        {                                                                                       \
          const int r_ = r0 + w_row + (M);                                                      \
          const bool row_ok_ = col_ok && r_ < Ho && !(g.dbg & 4);                               \
          const int rc_ = r_ > Ho - 1 ? Ho - 1 : r_;                                            \
          const unsigned o1x = __float_as_uint(acc[M][2 * (H)][0]), o1y = __float_as_uint(acc[M][2 * (H)][1]); \
          const unsigned o1z = __float_as_uint(acc[M][2 * (H)][2]), o1w = __float_as_uint(acc[M][2 * (H)][3]); \
          const u32x4 q4 = P;                                                                   \
          uint4 o;                                                                              \
          o.x = ws_pk(ws_lo(o1x) + ws_lo(q4[0]), ws_hi(o1x) + ws_hi(q4[0]));                    \
          o.y = ws_pk(ws_lo(o1y) + ws_lo(q4[1]), ws_hi(o1y) + ws_hi(q4[1]));                    \
          o.z = ws_pk(ws_lo(o1z) + ws_lo(q4[2]), ws_hi(o1z) + ws_hi(q4[2]));                    \
          o.w = ws_pk(ws_lo(o1w) + ws_lo(q4[3]), ws_hi(o1w) + ws_hi(q4[3]));                    \
          if (row_ok_ && ch_ok[H])                                                              \
            *reinterpret_cast<uint4*>(y + tbase + (unsigned long long)(unsigned)rc_ * RS + off_h[H]) = o; \
        }
        PP_RES2_STORE(0, 0, p0) PP_RES2_STORE(0, 1, p1) PP_RES2_STORE(1, 0, p2) PP_RES2_STORE(1, 1, p3)
        PP_RES2_STORE(2, 0, p4) PP_RES2_STORE(2, 1, p5) PP_RES2_STORE(3, 0, p6) PP_RES2_STORE(3, 1, p7)
#undef PP_RES2_STORE
      } else if (!(g.dbg & 16))
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        uint4 o_lo = make_uint4(0u, 0u, 0u, 0u);
        const int r = r0 + w_row + m;
        const bool row_ok = col_ok && r < Ho && !(g.dbg & 4);
        const int rc = r > Ho - 1 ? Ho - 1 : r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = acc[m][2 * h][e]; v[4 + e] = acc[m][2 * h + 1][e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float sa = slope * v[e];
            asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(sa));
          }
          const unsigned long long dst = tbase + (unsigned long long)(unsigned)rc * RS + off_h[h];
          if constexpr (RES) {
            const u32x4 q4 = rr[m][h];
            v[0] += ws_lo(q4[0]); v[1] += ws_hi(q4[0]); v[2] += ws_lo(q4[1]); v[3] += ws_hi(q4[1]);
            v[4] += ws_lo(q4[2]); v[5] += ws_hi(q4[2]); v[6] += ws_lo(q4[3]); v[7] += ws_hi(q4[3]);
          }
          uint4 o;
          o.x = ws_pk(v[0], v[1]); o.y = ws_pk(v[2], v[3]); o.z = ws_pk(v[4], v[5]); o.w = ws_pk(v[6], v[7]);
          if (!lin_st) {
            if (row_ok && ch_ok[h]) *reinterpret_cast<uint4*>(y + dst) = o;
          } else if (h == 0) {
            o_lo = o;
          } else {
            const int srcA = (((lane & 3) << 4) + (lane >> 3)) << 2, srcB = srcA + (8 << 2);
            const bool hi_half = (lane >> 2) & 1;
            const unsigned lo4[4] = {o_lo.x, o_lo.y, o_lo.z, o_lo.w}, hi4[4] = {o.x, o.y, o.z, o.w};
            unsigned oa[4], ob[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const unsigned la = (unsigned)__builtin_amdgcn_ds_bpermute(srcA, (int)lo4[q4]);
              const unsigned ha = (unsigned)__builtin_amdgcn_ds_bpermute(srcA, (int)hi4[q4]);
              const unsigned lb = (unsigned)__builtin_amdgcn_ds_bpermute(srcB, (int)lo4[q4]);
              const unsigned hb = (unsigned)__builtin_amdgcn_ds_bpermute(srcB, (int)hi4[q4]);
              oa[q4] = hi_half ? ha : la;
              ob[q4] = hi_half ? hb : lb;
            }
            if (r < Ho && !(g.dbg & 4)) {
              const unsigned long long d0 = (unsigned long long)im * IS + (unsigned long long)((unsigned)c0 * CS) +
                                            (unsigned long long)(unsigned)rc * RS + (unsigned)(lane * 8);
              *reinterpret_cast<uint4*>(y + d0) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
              *reinterpret_cast<uint4*>(y + d0 + 512) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
            }
          }
        }
      }
      if (k + 2 < n_g && !(g.dbg & 1)) PP_FETCH(t0 + grp + 2 * (k + 2));
    }
    PP_BARRIER();
  }
  if (grp == 0) PP_BARRIER();
}
#undef PP_WAIT_ALL
#undef PP_WAIT_P
#undef PP_COMMIT
#undef PP_SEL
#undef PP_FETCH
#undef PP_FETCH_PREP
#undef PP_TRIP
#undef PP_LOADQ
#undef PP_ADDRQ
#undef PP_COL
#undef PP_BARRIER

// ---------------------------------------------------------------------------
// The few-feature HEAD conv of a 2-D generator (C_in = 1 or 2 -> 64 channels,
// fp32 field in, bf16 cells out: sup3r/configs/spatial/gen_*_{1,2}f.json,
// sup3rcc/gen_solar_*).  On the logical-axes MFMA kernel it took 60 us at the
// config_fwp_spatial.json chunk — K = 18 padded to a 64-wide k-step, a halo
// staged through LDS for 0.3 GFLOP — as long as two trunk convs.  It is a
// streaming problem (2 MB in, 35 MB out): here a lane keeps the 9 x C_in x 8
// filter taps of ITS eight output channels in registers (bf16-rounded values, as
// the matrix path multiplies them) and slides a 3 x 3 window along a row
// segment: three new cells per position, 72 C_in FMAs, one 16-B store; eight
// lanes = one position's 128-B cell.  Products of two bf16 values are exact in
// fp32 and the accumulation is fp32 as on the matrix cores; only the order of
// the 9 C_in terms differs.
// positions of a row walked by one lane group per work item.  Short segments on purpose: at
// 197 registers a CU holds two workgroups, and 25-position segments at the production chunk
// were 338 workgroups — a third of the CUs ran two of them back to back (27 us for 13 us of work)
constexpr int HD_SEG = 5;
// F32 (BF16X3 plans): exact fp32 operands, fp32 cells out — the 9 C_in terms of an
// output are plain fp32 FMAs, at least as close to the oracle as the three-product
// split of the matrix path this layer used to run on (73 us at the production chunk)
template <int CIN, bool F32 = false>
__global__ __launch_bounds__(256) void conv2d_head_kernel(
    const float* __restrict__ x, const float* __restrict__ wimg, const float* __restrict__ bias,
    void* __restrict__ yv, int N, int H, int W, int act, float alpha) {
  unsigned short* __restrict__ y = reinterpret_cast<unsigned short*>(yv);
  const int tid = threadIdx.x;
  const int cg = tid & 7;                       // output channels cg 8 .. cg 8 + 7
  const int nseg = (W + HD_SEG - 1) / HD_SEG;
  const long long slots = (long long)N * H * nseg;
  float wr[9][CIN][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float4 a = *reinterpret_cast<const float4*>(wimg + (t * CIN + ci) * 64 + cg * 8);
      const float4 b = *reinterpret_cast<const float4*>(wimg + (t * CIN + ci) * 64 + cg * 8 + 4);
      wr[t][ci][0] = a.x; wr[t][ci][1] = a.y; wr[t][ci][2] = a.z; wr[t][ci][3] = a.w;
      wr[t][ci][4] = b.x; wr[t][ci][5] = b.y; wr[t][ci][6] = b.z; wr[t][ci][7] = b.w;
    }
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[cg * 8 + j] : 0.f;
  const float slope = act == S3_ACT_LEAKY ? alpha : (act == S3_ACT_RELU ? 0.f : 1.f);
  auto rnd = [](float v) __attribute__((always_inline)) {
    return F32 ? v : __uint_as_float(ws_pk(v, 0.f) << 16);
  };

  for (long long slot = (long long)blockIdx.x * 32 + (tid >> 3); slot < slots; slot += (long long)gridDim.x * 32) {
    const int seg = (int)(slot % nseg);
    const long long row = slot / nseg;           // n H + r
    const int r = (int)(row % H);
    const long long n = row / H;
    const int c_lo = seg * HD_SEG, c_hi = c_lo + HD_SEG < W ? c_lo + HD_SEG : W;
    const float* xr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) xr[a] = x + ((size_t)n * H + s3_reflect(r + a - 1, H)) * W * CIN;
    // window columns c - 1, c, c + 1 of the three rows (bf16-rounded)
    float win[3][3][CIN];
    auto load_col = [&](int slotc, int c) __attribute__((always_inline)) {
      const int cr = s3_reflect(c, W);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) win[a][slotc][ci] = rnd(xr[a][(size_t)cr * CIN + ci]);
    };
    load_col(0, c_lo - 1);
    load_col(1, c_lo);
    load_col(2, c_lo + 1);
    unsigned short* yo = y + ((((size_t)n * H + r) * W + c_lo) * 64 + cg * 8) * (F32 ? 2 : 1);
    for (int c = c_lo; c < c_hi; ++c) {
      // column c + 2 is fetched one position ahead of its use (two waves per SIMD at 197
      // registers do not cover an L1 round trip per position)
      float nxt[3][CIN];
      {
        const int cr = s3_reflect(c + 2 < W + 1 ? c + 2 : W, W);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) nxt[a][ci] = xr[a][(size_t)cr * CIN + ci];
      }
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bv[j];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const float xv = win[a][b][ci];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(wr[a * 3 + b][ci][j], xv, acc[j]);
          }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sa = slope * acc[j];
        asm("v_max_f32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(sa));
      }
      if constexpr (F32) {
        float4* yf = reinterpret_cast<float4*>(yo);
        yf[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        yf[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        yo += 128;
      } else {
        uint4 o;
        o.x = ws_pk(acc[0], acc[1]); o.y = ws_pk(acc[2], acc[3]); o.z = ws_pk(acc[4], acc[5]); o.w = ws_pk(acc[6], acc[7]);
        *reinterpret_cast<uint4*>(yo) = o;
        yo += 64;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          win[a][0][ci] = win[a][1][ci]; win[a][1][ci] = win[a][2][ci]; win[a][2][ci] = rnd(nxt[a][ci]);
        }
    }
  }
}

// canonical fp32 w[tap 9][ci][co 64] -> the same layout with bf16-rounded values
__global__ void pack_head_kernel(const float* __restrict__ w, float* __restrict__ out, int n, int exact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = exact ? w[i] : __uint_as_float(ws_pk(w[i], 0.f) << 16);
}

}  // namespace

// physical geometry of a 2-D conv: (N, s1, s2, 1, C), k = (3, 3, 1)
bool conv2d_ws_geom_ok(const ConvGeom& g) {
  if (g.Cin != 64 || g.Cout % 64 != 0 || g.Cout < 64) return false;
  if (g.k[0] != 3 || g.k[1] != 3 || g.k[2] != 1 || g.D[2] != 1 || g.O[2] != 1) return false;
  if (g.pad_mode != S3_PAD_REFLECT || g.in_cstride || g.in_rep > 1 || g.res_rep > 1) return false;
  for (int d = 0; d < 2; ++d)
    if (g.s[d] != 1 || g.lo[d] != 1 || g.O[d] != g.D[d] || g.D[d] < 2) return false;
  if (g.s[2] != 1 || g.lo[2] != 0) return false;
  const int b = g.d2s < 1 ? 1 : g.d2s;
  if (g.Cout % (b * b) != 0 || (g.Cout / (b * b)) % 8 != 0) return false;
  if (g.act == S3_ACT_LEAKY && !(g.alpha >= 0.f && g.alpha <= 1.f)) return false;
  // per sample, like the logical-axes kernel it replaces for these layers
  if ((int64_t)g.O[0] * g.O[1] < 256) return false;
  // (cell indices are 32-bit, element offsets 64-bit: the 750 x 750 x 96-image
  // hi-res layers of a 10x chain of spatial steps — 3.5e9 elements — stay here)
  return (int64_t)g.N * g.D[0] * g.D[1] < ((int64_t)1 << 31) &&
         (int64_t)g.N * g.O[0] * g.O[1] * (g.d2s < 1 ? 1 : g.d2s) * (g.d2s < 1 ? 1 : g.d2s) < ((int64_t)1 << 31);
}

// the data gradient of a 2-D reflect-'same' 64 -> 64 k conv as a conv over the
// zero-padded frame (conv_dgrad_gen_geom): bf16 dPre in, bf16 frame out
bool conv2d_ws_frame_geom_ok(const ConvGeom& g) {
  if (g.Cin != 64 || g.Cout != 64 || g.w_cin) return false;
  if (g.k[0] != 3 || g.k[1] != 3 || g.k[2] != 1 || g.D[2] != 1 || g.O[2] != 1) return false;
  if (g.pad_mode != S3_PAD_ZERO || g.in_cstride || g.in_rep > 1 || g.res_rep > 1 || g.d2s > 1) return false;
  if (g.act != S3_ACT_NONE) return false;
  for (int d = 0; d < 2; ++d)
    if (g.s[d] != 1 || g.lo[d] != 2 || g.O[d] != g.D[d] + 2 || g.D[d] < 2) return false;
  if (g.s[2] != 1 || g.lo[2] != 0) return false;
  if ((int64_t)g.D[0] * g.D[1] < 256) return false;
  return (int64_t)g.N * g.O[0] * g.O[1] < ((int64_t)1 << 31);
}

// the few-feature output conv: 64 -> C_out <= 16, no depth-to-space
bool conv2d_ws_tail_geom_ok(const ConvGeom& g) {
  if (g.Cout < 1 || g.Cout > 16 || (g.d2s > 1)) return false;
  ConvGeom t = g;
  t.Cout = 64;
  t.d2s = 1;
  return conv2d_ws_geom_ok(t);
}

bool conv2d_ws_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res) {
  if (precision != S3_PREC_BF16 || s3_opt_on(S3O_NO_CONV2D_WS)) return false;
  if (conv2d_ws_frame_geom_ok(g)) return io.in_bf16 && io.out_bf16 && !has_res;
  if (g.w_cin && (g.w_cin != 65 || conv2d_ws_tail_geom_ok(g))) return false;   // (one exogenous channel, trunk form)
  if (conv2d_ws_tail_geom_ok(g)) return io.in_bf16 && !io.out_bf16 && !has_res;
  if (!io.in_bf16 || !io.out_bf16 || (has_res && !io.res_bf16)) return false;
  if (has_res && g.d2s > 1) return false;
  return conv2d_ws_geom_ok(g);
}

size_t conv2d_ws_image_bytes(const ConvGeom& g) {
  const size_t n_ct = (size_t)((g.Cout + 63) / 64);
  return n_ct * 9 * 8192 + (g.w_cin > 64 ? n_ct * 9 * 64 * sizeof(float) : 0);
}

int launch_conv2d_ws_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image) {
  const int n_ct = (g.Cout + 63) / 64;
  int grid = (n_ct * 9 * 64 * 64 + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_ws_kernel, dim3(grid), dim3(256), 0, ctx->stream, w, (unsigned short*)image, g.Cout, n_ct,
                     conv2d_ws_tail_geom_ok(g) ? 1 : 0, g.w_cin ? g.w_cin : 64);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv2d_ws(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image, const float* bias,
                     const void* res, void* y) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_kernel<4, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_EXO));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_kernel<4, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_kernel<4, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_EXO));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_pp_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv2d_ws_pp_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS));
    attr_set.mark(ctx->device);
  }
  WsGeom w;
  w.N = g.N; w.H = g.D[0]; w.W = g.D[1];
  w.Cout = g.Cout; w.b = g.d2s < 1 ? 1 : g.d2s; w.cpo = g.Cout / (w.b * w.b);
  w.act = g.act; w.alpha = g.alpha;
  w.dbg = (int)s3_opt_int(S3O_MFMA_DBG, 0);
  w.frame = conv2d_ws_frame_geom_ok(g) ? 1 : 0;
  if (w.frame && (res || g.res2 || g.w_cin)) S3_FAIL(ctx, S3_ESTATE, "conv2d_ws: the frame form takes no skip / exo operand");
  w.tiles_i = (g.N + WT_I - 1) / WT_I;
  w.tiles_r = (w.H + 2 * w.frame + WT_R - 1) / WT_R; w.tiles_c = (w.W + 2 * w.frame + WT_C - 1) / WT_C;
  const bool tail = conv2d_ws_tail_geom_ok(g);
  const int T = w.tiles_i * w.tiles_r * w.tiles_c, n_ct = (g.Cout + 63) / 64;
  // at most one workgroup per CU over all output-channel tiles (each keeps ITS image; a
  // workgroup fills a CU's LDS, so one more than there are CUs would run after the others:
  // 25 channel tiles of the 64 -> 1600 conv x 11 = 275 workgroups was two rounds)
  int gx = ctx->num_cu / n_ct;
  if (gx > T) gx = T;
  if (gx < 1) gx = 1;
  if (g.w_cin && !g.exo) S3_FAIL(ctx, S3_ESTATE, "conv2d_ws: the exogenous channel's field is not bound");
  if (g.res2 && (tail || w.b != 1)) S3_FAIL(ctx, S3_ESTATE, "conv2d_ws: a second skip operand needs the trunk form without depth-to-space");
  if (tail)
    hipLaunchKernelGGL(conv2d_ws_kernel<1>, dim3(gx, 1), dim3(W_NT), W_LDS, ctx->stream, (const unsigned short*)x,
                       (const char*)image, bias, (const unsigned short*)nullptr, y, w, (const float*)nullptr,
                       (const unsigned short*)nullptr);
  else if (!g.w_cin && !s3_opt_on(S3O_NO_WS_PP)) {
    // the ping-pong form: single-image tiles, two half-workgroups half a period apart
    const int T1 = g.N * w.tiles_r * w.tiles_c;
    int gp = ctx->num_cu / n_ct;
    if (gp > (T1 + 1) / 2) gp = (T1 + 1) / 2;
    if (gp < 1) gp = 1;
    auto kern = res ? conv2d_ws_pp_kernel<true> : conv2d_ws_pp_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(gp, n_ct), dim3(W_NT), W_LDS, ctx->stream, (const unsigned short*)x,
                       (const char*)image, bias, (const unsigned short*)res, (unsigned short*)y, w,
                       (const unsigned short*)g.res2);
  } else {
    auto kern = g.w_cin ? (res ? conv2d_ws_kernel<4, true, true> : conv2d_ws_kernel<4, true, false>)
                        : (res ? conv2d_ws_kernel<4, false, true> : conv2d_ws_kernel<4, false, false>);
    hipLaunchKernelGGL(kern, dim3(gx, n_ct), dim3(W_NT), g.w_cin ? W_LDS_EXO : W_LDS, ctx->stream,
                       (const unsigned short*)x, (const char*)image, bias, (const unsigned short*)res, y, w,
                       g.w_cin ? g.exo : (const float*)nullptr, (const unsigned short*)g.res2);
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

// ---- the few-feature head conv (conv2d_head_kernel): C_in 1 / 2 -> 64, fp32 in, bf16 out
bool conv2d_head_geom_ok(const ConvGeom& g) {
  if ((g.Cin != 1 && g.Cin != 2) || g.Cout != 64 || g.w_cin || g.d2s > 1) return false;
  if (g.k[0] != 3 || g.k[1] != 3 || g.k[2] != 1 || g.D[2] != 1 || g.O[2] != 1) return false;
  if (g.pad_mode != S3_PAD_REFLECT || g.in_cstride || g.in_rep > 1 || g.res_rep > 1) return false;
  for (int d = 0; d < 2; ++d)
    if (g.s[d] != 1 || g.lo[d] != 1 || g.O[d] != g.D[d] || g.D[d] < 2) return false;
  if (g.s[2] != 1 || g.lo[2] != 0) return false;
  if (g.act == S3_ACT_LEAKY && !(g.alpha >= 0.f && g.alpha <= 1.f)) return false;
  // per image, like conv2d_ws_geom_ok: the choice must not depend on the batch size
  // (chunk by chunk == batched, bit for bit)
  return (int64_t)g.O[0] * g.O[1] >= 256;
}

bool conv2d_head_supported(const ConvGeom& g, int precision, ConvIO io, bool has_res) {
  if (s3_opt_on(S3O_NO_CONV2D_WS) || s3_opt_on(S3O_NO_CONV2D_HEAD)) return false;
  if (!conv2d_head_geom_ok(g) || io.in_bf16 || has_res || g.res2) return false;
  if (precision == S3_PREC_BF16) return io.out_bf16 != 0;
  if (precision == S3_PREC_BF16X3) return io.out_bf16 == 0;     // (exact fp32 form)
  return false;
}

size_t conv2d_head_image_bytes(const ConvGeom& g) { return (size_t)9 * g.Cin * 64 * sizeof(float); }

int launch_conv2d_head_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image, int exact) {
  const int n = 9 * g.Cin * 64;
  hipLaunchKernelGGL(pack_head_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, w, (float*)image, n, exact);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv2d_head(s3_ctx* ctx, const ConvGeom& g, const void* x, const void* image, const float* bias, void* y,
                       int exact) {
  const int nseg = (g.D[1] + HD_SEG - 1) / HD_SEG;
  const int64_t slots = (int64_t)g.N * g.D[0] * nseg;
  int64_t grid = (slots + 31) / 32;
  // (all workgroups resident — two per CU — each walking its share of the items)
  if (grid > (int64_t)ctx->num_cu * 2) grid = (int64_t)ctx->num_cu * 2;
  auto kern = g.Cin == 1 ? (exact ? conv2d_head_kernel<1, true> : conv2d_head_kernel<1, false>)
                         : (exact ? conv2d_head_kernel<2, true> : conv2d_head_kernel<2, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const float*)x, (const float*)image, bias,
                     y, g.N, g.D[0], g.D[1], g.act, g.alpha);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
