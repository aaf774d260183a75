// Persistent, wave-specialised variant of the halo-tile implicit-GEMM Conv3D
// (kernels_conv_mfma.hip) for the bf16 trunk of the generator: 64 -> 64
// channels, 3x3x3, stride 1, bf16 activations in and out (+ optional bf16
// skip tensor), gfx950 only.
//
// The one-tile-per-workgroup kernel runs its three phases back to back —
// halo staging (HBM/L2 latency), 27 taps on MFMA, epilogue — on every CU at
// the same time, so HBM sees bursts and the matrix cores idle in between;
// one 138 KB halo per CU leaves no room for a second workgroup to overlap
// them.  Here ONE 12-wave workgroup per CU walks a list of 4 x 8 x 16
// position tiles with two kinds of waves:
//
//   consumers (waves 0-7): nothing but ds_read_b128 + MFMA in the 27-tap
//     loop — no vector-memory instruction, so no vmcnt wait can ever park
//     them — then the epilogue straight from the accumulators.  The MFMA
//     operands are swapped (A = filter rows, B = positions) and the filter
//     rows of a slab are permuted so that a lane's C/D fragments are 8
//     consecutive output channels of ONE position per pair of N fragments:
//     bias-init, activation, residual add and 16-B stores (4 lanes = 64
//     contiguous bytes) need no LDS round trip.
//   producers (waves 8-11): stream the 8 KB filter slab of tap + 2 into a
//     3-slot LDS ring with LDS-DMA (global_load_lds_dwordx4; the global image
//     IS the swizzled LDS image) and fetch the NEXT tile's halo into registers
//     (34 x 16 B per lane), two chunks per tap, so the HBM traffic of the
//     halo is spread over the tap loop instead of bursting; after the last
//     tap they drop the halo into LDS.  Their vmcnt waits are counted
//     (never 0 inside the loop): slab t+1 has two taps to arrive.
//
// One s_barrier per tap orders slab hand-over both ways (27 = 9 x 3: the ring
// slot tap % 3 is a compile-time immediate and the tap pipeline is continuous
// across tiles).  LDS: halo 1080 cells x 128 B | 3 slabs x 8 KB | 64 biases
// = 163,072 B.  Swizzles as in kernels_conv_mfma.hip (halo chunk ^ (cell_t &
// 7), slab chunk ^ ((row >> 1) & 7)); all ds_read_b128 are conflict-free.
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int TS0 = 4, TS1 = 8, TS2 = 16;
constexpr int H0 = TS0 + 2, H1 = TS1 + 2, H2 = TS2 + 2;
constexpr int HP = H0 * H1 * H2;                 // 1080 halo cells
constexpr int HALO_BYTES = HP * 128;             // 138,240
constexpr int SLAB_OFF = HALO_BYTES;
constexpr int BIAS_OFF = SLAB_OFF + 3 * 8192;    // 162,816
constexpr int LDS_BYTES = BIAS_OFF + 256;        // 163,072

constexpr int NCW = 8;                           // consumer (MFMA) waves
constexpr int NPW = 4;                           // producer (memory) waves
constexpr int NTHR = (NCW + NPW) * 64;           // 768
constexpr int MFW = TS0 * TS1 / NCW;             // 4 M fragments per consumer
constexpr int PT = NPW * 64;                     // producer threads
constexpr int ROWC = H1 * H2;                    // 180 cells per halo row (fixed c0)
constexpr int JR = (ROWC * 8 + PT - 1) / PT;     // 6 chunks per row per producer lane
constexpr int NLATE = (H0 - 2) * JR;             // 24: rows 2..5, parked in registers
static_assert(MFW * NCW == TS0 * TS1 && TS1 % MFW == 0 && MFW == 4, "tile / wave split");
static_assert(JR <= 6 && NLATE <= 24, "producer tap schedule");

// Producer schedule of the NEXT tile's halo over the 27 taps of this one.
// Halo row c0 is last read in tap 9*c0 + 8 for c0 = 0, 1 (output row s0 reads
// halo rows s0 + ta), so rows 0 and 1 go through a 6-chunk register buffer
// and land in LDS in mid-loop; rows 2..5 are parked in 24 chunks of
// registers until the last tap is over.
//   taps 0..5  : one chunk of row 0      taps 9..14 : one chunk of row 1
//   taps 0..23 : one chunk of rows 2..5
constexpr int early_k(int t) { return (t >= 0 && t < JR) || (t >= 9 && t < 9 + JR) ? 1 : 0; }
constexpr int late_k(int t) { return t >= 0 && t < NLATE ? 1 : 0; }
constexpr int halo_k(int t) { return early_k(t) + late_k(t); }

// Round 4: the tile width along s1 is a template parameter TW (8, or 6 for the
// last column strip of an extent like the C3 chunk's 22 = 8 + 8 + 6: three
// 8-column tiles compute 24 columns).  PGeo<TW> carries what depends on it; the
// kernel body names them as before (local constants shadow the TW = 8 ones
// above, which the launch code keeps using).  A 6-column tile is 4 rows x 2
// groups of 3 position fragments per consumer wave.
template <int TW>
struct PGeo {
  static constexpr int TS1 = TW, H1 = TW + 2;
  static constexpr int HP = H0 * H1 * H2, HALO_BYTES = HP * 128, SLAB_OFF = HALO_BYTES;
  static constexpr int BIAS_OFF = SLAB_OFF + 3 * 8192, LDS_BYTES = BIAS_OFF + 256;
  static constexpr int MFW = TS0 * TW / NCW;
  static constexpr int ROWC = H1 * H2;
  static constexpr int JR = (ROWC * 8 + PT - 1) / PT;
  static constexpr int NLATE = (H0 - 2) * JR;
  static_assert(MFW * NCW == TS0 * TW && TW % MFW == 0 && (MFW == 4 || MFW == 3), "tile / wave split");
  static_assert(JR <= 6 && NLATE <= 24 && H0 + H1 + H2 <= 64, "producer tap schedule / halo table");
  static constexpr int early_k(int t) { return (t >= 0 && t < JR) || (t >= 9 && t < 9 + JR) ? 1 : 0; }
  static constexpr int late_k(int t) { return t >= 0 && t < NLATE ? 1 : 0; }
  static constexpr int halo_k(int t) { return early_k(t) + late_k(t); }
};

__device__ inline unsigned pk_bf16(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ inline float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float hi_f(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// LDS slab row rho = nf*16 + kq*4 + r  <->  output channel
//   (nf >> 1)*32 + kq*8 + (nf & 1)*4 + r
// so that lane (pos, kq) owns channels h*32 + kq*8 .. +7 for h = nf >> 1.
__device__ __host__ inline int slab_row_cout(int rho) {
  const int nf = rho >> 4, kq = (rho >> 2) & 3, r = rho & 3;
  return (nf >> 1) * 32 + kq * 8 + (nf & 1) * 4 + r;
}

// workgroup barrier without the fence of __syncthreads(): LDS hand-over is
// ordered by the explicit waits next to it ("memory": no compiler motion)
#define WG_BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int N>
__device__ inline void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// 16-B global load the compiler can neither sink nor wait on: the producer
// loop orders it by hand (asm volatile keeps program order with the counted
// s_waitcnt / s_barrier statements).  saddr + 32-bit byte offset.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline u32x4 ld16_async(const void* sbase, unsigned voff) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
// n is a constant after unrolling: the switch folds to one s_waitcnt
__device__ inline void wait_vm_n(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    default: wait_vm<6>(); break;
  }
}

// canonical fp32 w[tap][ci][co] -> bf16 LDS images [tap][rho 64][ci 64] with
// rows in rho order and 16-B chunks swizzled by rho
__global__ void pack_persist_kernel(const float* __restrict__ w,
                                    unsigned short* __restrict__ out, int cout, int n_ct) {
  const int total = n_ct * 27 * 64 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += gridDim.x * blockDim.x) {
    const int ci = idx & 63, rho = (idx >> 6) & 63, tap = (idx >> 12) % 27, ct = (idx >> 12) / 27;
    const int co = ct * 64 + slab_row_cout(rho);
    const float v = co < cout ? w[((size_t)tap * 64 + ci) * cout + co] : 0.f;
    const unsigned u = pk_bf16(v, 0.f);
    const int slot = (ci >> 3) ^ ((rho >> 1) & 7);
    out[(((size_t)ct * 27 + tap) * 64 + rho) * 64 + slot * 8 + (ci & 7)] = (unsigned short)(u & 0xFFFFu);
  }
}

// ---- batched re-pack: after an optimizer step every bf16 conv of a training
// plan needs its filter images again — three launches of ~5 us per conv and
// direction (tile image, persistent image, flipped fp32 filter of the data
// gradient: 220+ launches per C2 step).  One launch walks a table of jobs:
// blockIdx.y = job, a thread computes one filter element once and writes it to
// the halo-tile image (kernels_conv_mfma.hip: rows = cout, 16-B chunks
// swizzled by (row >> 1) & 7) and, when the job has one, to the persistent
// image (rows in rho order).  dgrad jobs read the forward filter through the
// flip / transpose map  v'(tap', ci', co') = w[26 - tap'][co'][ci'].
__global__ void pack_jobs_kernel(const S3PackJob* __restrict__ jobs) {
  const S3PackJob jb = jobs[blockIdx.y];
  const int total = jb.n_ct * 27 * 64 * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int ci = idx & 63, row = (idx >> 6) & 63, tap = (idx >> 12) % 27, ct = (idx >> 12) / 27;
    const int co = ct * 64 + row;
    float v = 0.f;
    if (co < jb.cout)
      v = jb.dgrad ? jb.w[((size_t)(26 - tap) * jb.cout + co) * 64 + ci]      // forward w[tap][ci_f = co'][co_f = ci']
                   : jb.w[((size_t)tap * 64 + ci) * jb.cout + co];
    const unsigned short h = (unsigned short)(pk_bf16(v, 0.f) & 0xFFFFu);
    const size_t base = (((size_t)ct * 27 + tap) * 64) * 64;
    const int sw = (ci >> 3);
    jb.tile[base + (size_t)row * 64 + ((sw ^ ((row >> 1) & 7)) << 3) + (ci & 7)] = h;
    if (jb.persist) {
      const int nf = ((row >> 5) << 1) | ((row >> 2) & 1), kq = (row >> 3) & 3, r = row & 3;
      const int rho = nf * 16 + kq * 4 + r;      // slab_row_cout(rho) == row
      jb.persist[base + (size_t)rho * 64 + ((sw ^ ((rho >> 1) & 7)) << 3) + (ci & 7)] = h;
    }
  }
}

// NFV: N fragments computed (4 = all 64 channels of the tile; 2 = the first
// 32, for a last tile holding <= 32 valid channels)
//
// DG: the DATA GRADIENT of a reflect-padded 64 -> 64 trunk conv — the full
// correlation of bf16 dPre (g.D) with the flipped filter over the padded frame
// g.O = g.D + 2, zero boundary, fp32 out, no bias / activation / skip.  The
// frames of the N samples are STACKED into one gs0 x gs1 grid of frames along
// s0 and s1 (stacked coordinate S = q E + u, E = frame extent): with E = 18 no
// multiple of the 4 x 8 tile fits one frame, but 2 x 4 frames are 36 x 72 =
// 9 x 9 tiles exactly.  In stacked coordinates the gradient is a 'same' conv
// over Z[q E + j] = dPre_q[j - 1] for 1 <= j <= E - 2, zero for j = 0, E - 1:
// the zero separator rows stand for both neighbours' zero boundaries, so
// a tile may straddle frames.  The halo tables carry a "zero row" flag (bit 30
// of the element offset; a chunk is zeroed before it lands in LDS if any of
// its three axes is flagged) instead of the reflect rule.
// F16 (DG only): the padded frame is stored as bf16 — 96 instead of 192 MB per
// trunk conv at C2 batch 8, written once here and read once by the fold; the
// fold's sum of <= 8 frame cells is then the sum of bf16-rounded terms (the
// rounding point tests/helpers.emulate_plan installs in the oracle's pad
// adjoint).
template <int NFV, bool DG, int REP = 0, bool RIN = false, bool F16 = false, int TW = 8>
__global__ __launch_bounds__(NTHR) void conv3_mfma_persist_kernel(
    const unsigned short* __restrict__ x, const char* __restrict__ wimg,
    const float* __restrict__ bias, const unsigned short* __restrict__ res,
    unsigned short* __restrict__ y, ConvGeom g, int tiles0, int tiles1,
    int tiles2, int n_tiles, int ct, int gs0, int gs1) {
  // ct: which 64-wide output-channel tile this launch computes (C_out > 64:
  // one launch per tile; wimg / bias already point at the tile's image)
  // gs1 (forward launches, !DG): first s1 column of this launch's strip
  using PG = PGeo<TW>;
  constexpr int TS1 = PG::TS1, H1 = PG::H1, HP = PG::HP, SLAB_OFF = PG::SLAB_OFF, BIAS_OFF = PG::BIAS_OFF;
  constexpr int MFW = PG::MFW, ROWC = PG::ROWC, JR = PG::JR, NLATE = PG::NLATE;
  (void)HP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D0 = g.D[0], D1 = g.D[1], D2 = g.D[2];

  // ---- this workgroup's work list.  The unit is HALF a tile (two s0 rows =
  // the four consumer waves of one row pair, one per SIMD), and the half-tiles
  // of one (n, s1, t) column of tiles are numbered along s0 FIRST: index
  // h = ((n tiles1 + t1) tiles2 + t2) nh0 + hs0 with nh0 = ceil(rows / 2),
  // origin row 2 hs0.  Workgroup rank w owns [w H / G, (w+1) H / G) and walks
  // it in whole tiles (two consecutive half rows of one column, at ANY even
  // origin) where it can.  That balances the tail (1152 tiles on 256 CUs is 4.5
  // tiles each, not 5 rounds), puts neighbouring workgroups half a tile out of
  // phase, so their epilogue / halo traffic does not hit HBM in the same
  // microsecond — and an extent that is not a multiple of 4 rows (the 22 x 22
  // chunks of the C3 executor: 11 half rows) costs no edge tile along s0.
  // Ranks are XCD-major (block b sits on XCD b % 8): each XCD owns a
  // contiguous run of items and neighbouring halos share its L2.
  // (kernel argument tiles0 = nh0, n_tiles = H)
  int h_cur, h_end;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int xcd = b % 8, k = b / 8;
    int rank = k;
    for (int xx = 0; xx < xcd; ++xx) rank += (nblk - xx + 7) / 8;
    const long long H = n_tiles;
    h_cur = (int)((rank * H) / nblk);
    h_end = (int)(((rank + 1) * H) / nblk);
  }
  // origin of the item starting at half-tile h
  auto tile_org = [&](int h, int& n, int& o0, int& o1, int& o2) __attribute__((always_inline)) {
    int tr = h;
    o0 = (tr % tiles0) * 2; tr /= tiles0;
    o2 = (tr % tiles2) * TS2; tr /= tiles2;
    o1 = (tr % tiles1) * TS1 + (DG ? 0 : gs1); tr /= tiles1;
    n = tr;
  };
  // work item starting at half-tile h, whose half-row index hs0 = h % nh0 is
  // carried along (no division per item): rows (2 or 4); returns the next h
  auto item_at = [&](int h, int& hs0, int& nr) __attribute__((always_inline)) {
    const bool whole = hs0 + 1 < tiles0 && h + 1 < h_end;
    nr = whole ? 4 : 2;
    hs0 += whole ? 2 : 1;
    if (hs0 >= tiles0) hs0 = 0;
    return h + (whole ? 2 : 1);
  };

  if (wave >= NCW) {
    // =================================================== producer waves
    const int pt = tid - NCW * 64;               // 0 .. 255
    const int pw = wave - NCW;                   // 0 .. 3
    // slab DMA: wave pw copies bytes [pw*2048, pw*2048 + 2048) of the 8 KB
    // image (scalar base + one per-lane offset register)
    const unsigned dma_voff = (unsigned)(pw * 2048 + lane * 16);
    auto dma_slab = [&](int tap, int slot) __attribute__((always_inline)) {
      // opaque per-call copy: 54 hoisted 64-bit addresses would not fit
      unsigned vo;
      asm volatile("v_mov_b32 %0, %1" : "=v"(vo) : "v"(dma_voff));
      const char* sb = wimg + (size_t)tap * 8192 + vo;
      // (pw * 2048 re-read from lane 0 of the opaque copy: a scalar of its own
      // held across the tap loop is one SGPR more than the producers have)
      char* d = smem + SLAB_OFF + slot * 8192 + __builtin_amdgcn_readfirstlane((int)vo);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)sb,
          (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(sb + 1024),
          (__attribute__((address_space(3))) void*)(d + 1024), 16, 0, 0);
    };
    // Halo chunk (row r, j) of this lane: cell r*180 + (pt >> 3) + 32 j,
    // 16-B chunk pt & 7.  The reflect rule is evaluated ONCE per tile and
    // axis into a 34-entry element-offset table held across the lanes of one
    // VGPR (lanes 0-5: axis 0, 6-15: axis 1, 16-33: axis 2); the row offset
    // is a scalar (readlane), the in-row offset of chunk j two ds_bpermute
    // lookups per tile, so a chunk's address is one add.
    const int pcell = pt >> 3, pch = pt & 7;
    // (DG over a channel slice: 16-B chunks past the slice's valid channels
    // load the cell's first chunk — a legal address — and land as zeros)
    const bool cdead = DG && g.in_cstride && pch * 8 >= g.in_cvalid;
    u32x4 hlate[NLATE], hrow[JR];
    unsigned in_off[JR];      // element offset of chunk j inside a halo row
    unsigned lds_off[JR];     // byte offset of chunk j inside a halo row (swizzled)
    int htab = 0;
    const unsigned short* hx = x;
#pragma unroll
    for (int j = 0; j < JR; ++j) {
      int cell = pcell + 32 * j;
      if (cell > ROWC - 1) cell = ROWC - 1;      // tail lanes duplicate the last cell
      lds_off[j] = (unsigned)(cell * 128 + ((pch ^ ((cell % H2) & 7)) << 4));
    }
    // (a macro, not a lambda: state written through a by-reference capture
    // would be pinned to scratch memory by the "memory" clobbers below)
#define HALO_TABLE(tile_expr)                                                          \
    {                                                                                  \
      int n_, o0_, o1_, o2_;                                                           \
      tile_org((tile_expr), n_, o0_, o1_, o2_);                                        \
      n_ = __builtin_amdgcn_readfirstlane(n_);                                         \
      const int ax = lane < H0 ? 0 : (lane < H0 + H1 ? 1 : 2);                         \
      const int c = lane - (ax == 0 ? 0 : (ax == 1 ? H0 : H0 + H1));                   \
      const int org = ax == 0 ? o0_ : (ax == 1 ? o1_ : o2_);                           \
      /* RIN (input through a fused temporal repeat): g.D[2] is the SOURCE extent */   \
      constexpr bool rp = REP > 1 && RIN && !DG;                                       \
      /* (per-lane VALU arithmetic with immediates on purpose: the producer waves */   \
      /* are at the SGPR limit here — one more scalar is a spill into a VGPR, and */   \
      /* at 168 VGPRs that spills eight more) */                                       \
      const int is2 = rp ? (ax >> 1) : 0;                    /* 1 on the t axis */     \
      const int D = (ax == 0 ? D0 : (ax == 1 ? D1 : D2)) * (1 + (REP > 1 ? REP - 1 : 0) * is2); \
      const int cs = (DG && g.in_cstride) ? g.in_cstride : 64;  /* channel slice of a wider dPre */ \
      const int stride = ax == 0 ? D1 * D2 * cs : (ax == 1 ? D2 * cs : cs);            \
      if (DG) {                                                                        \
        /* stacked frames on axes 0 / 1 (extent E = D + 2, gs frames), plain zero */  \
        /* boundary on axis 2; flagged rows load a legal address and are zeroed */    \
        const int E = D + 2, gsx = ax == 0 ? gs0 : (ax == 1 ? gs1 : 1);                \
        const int R = org + c - 1;                                                     \
        const int q = R >= 0 ? R / E : 0, j = R - q * E;                               \
        const bool zero = R < 0 || R >= gsx * E || j == 0 || j == E - 1;               \
        const int sstr = D0 * D1 * D2 * cs;                                            \
        const int qoff = ax == 0 ? q * gs1 * sstr : (ax == 1 ? q * sstr : 0);          \
        htab = zero ? 0x40000000 : qoff + (j - 1) * stride;                            \
        hx = x;                                                                        \
      } else {                                                                         \
      /* (repeat variants: one padding for all axes, checked by the plan — two scalars less) */ \
      int i = s3_reflect(org + c - (REP > 1 ? g.lo[0] : g.lo[ax]), D);                 \
      /* ragged tiles: keep addresses legal (results are masked at the store) */      \
      i = i < 0 ? 0 : (i > D - 1 ? D - 1 : i);                                         \
      if (rp) {                                                                        \
        if (REP == 2) i >>= is2;                                                       \
        else if (REP == 4) i >>= 2 * is2;                                              \
        else {   /* i / 3 = (i * 0x5556) >> 16 for i < 2^15, by shifts and adds with */ \
                 /* inline constants; blended in by is2 instead of a lane mask */      \
          const unsigned u = (unsigned)i;                                              \
          unsigned t = u + (u << 2);                                                   \
          t += t << 4;                                                                 \
          t += t << 8;                                                                 \
          i = (int)(u + (unsigned)is2 * (((t + u) >> 16) - u));                        \
        }                                                                              \
      }                                                                                \
      htab = i * stride;                                                               \
      hx = x + (size_t)n_ * D0 * D1 * D2 * 64;                                         \
      }                                                                                \
      _Pragma("unroll") for (int j = 0; j < JR; ++j) {                                 \
        int cell = pcell + 32 * j;                                                     \
        if (cell > ROWC - 1) cell = ROWC - 1;                                          \
        const int c1 = cell / H2, c2 = cell - c1 * H2;                                 \
        in_off[j] = (unsigned)__builtin_amdgcn_ds_bpermute((H0 + c1) << 2, htab) +     \
                    (unsigned)__builtin_amdgcn_ds_bpermute((H0 + H1 + c2) << 2, htab) + \
                    (cdead ? 0 : pch * 8);                                             \
      }                                                                                \
    }
    // in-loop loads are hand-ordered (ld16_async): every tap's counted wait
    // retires all but the youngest <= 6 vector-memory ops, so a chunk issued
    // >= 2 taps before its halo_put has landed without a wait of its own
    auto halo_load = [&](int r, int j) __attribute__((always_inline)) {
      const unsigned row = (unsigned)__builtin_amdgcn_readlane(htab, r);
      // (DG: bits 30.. of the sum count the flagged axes; the rest is legal)
      const unsigned eo = DG ? ((row + in_off[j]) & 0x3FFFFFFFu) : row + in_off[j];
      return ld16_async(hx, eo * 2);
    };
    auto halo_put = [&](int r, int j, const u32x4& v) __attribute__((always_inline)) {
      if (pcell + 32 * j < ROWC) {
        u32x4 w = v;
        if (DG) {
          // v may still be in flight when this lambda is inlined: nothing may
          // touch it before the counted wait that precedes the call.  The
          // zeroing select below is plain C — an empty volatile asm (ordered
          // after that wait, like every volatile asm) pins it behind the wait
          asm volatile("" : "+v"(w));
          const unsigned row = (unsigned)__builtin_amdgcn_readlane(htab, r);
          if (((row + in_off[j]) >> 30) || cdead) w = (u32x4){0u, 0u, 0u, 0u};
        }
        unsigned lo = lds_off[j];
        if constexpr (REP > 1) {
          // (the repeat variants recompute it: six registers less in the
          // producer waves keeps them inside the 168-register budget without
          // a spill — tests/test_abi.py::test_persistent_kernel_has_no_scratch)
          int cell = pcell + 32 * j;
          if (cell > ROWC - 1) cell = ROWC - 1;
          lo = (unsigned)(cell * 128 + ((pch ^ ((cell % H2) & 7)) << 4));
        }
        *reinterpret_cast<u32x4*>(smem + r * (ROWC * 128) + lo) = w;
      }
    };

    // ---- prologue: biases (rho order), first halo, slabs of taps 0 and 1
    if (pt < 64)
      reinterpret_cast<float*>(smem + BIAS_OFF)[pt] =
          (bias && ct * 64 + slab_row_cout(pt) < g.Cout) ? bias[ct * 64 + slab_row_cout(pt)] : 0.f;
    dma_slab(0, 0);
    dma_slab(1, 1);
    if (h_cur < h_end) {
      HALO_TABLE(h_cur);
#pragma unroll
      for (int r = 0; r < H0; ++r) {
#pragma unroll
        for (int j = 0; j < JR; ++j) hrow[j] = halo_load(r, j);
        wait_vm<0>();
#pragma unroll
        for (int j = 0; j < JR; ++j) halo_put(r, j, hrow[j]);
      }
    }
    wait_vm<0>();
    WAIT_LGKM0();
    WG_BARRIER();

    for (int h = h_cur; h < h_end;) {
      // (h % nh0 per item rather than a carried scalar: the producer waves
      // are at the SGPR limit)
      int nr_, hs0_ = h % tiles0;
      h = item_at(h, hs0_, nr_);
      // (re-evaluated at each use instead of one more live scalar)
#define has_next (h < h_end)
      // without a next item the prefetch re-reads this rank's last tile (same
      // op count)
      HALO_TABLE(h < h_end - 1 ? h : h_end - 1);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        // rows 0 / 1 of the current halo were last read in taps 8 / 17
        if ((tap == 9 || tap == 18) && has_next) {
#pragma unroll
          for (int j = 0; j < JR; ++j) halo_put(tap == 9 ? 0 : 1, j, hrow[j]);
        }
        // slot (tap+2)%3 was read during tap-1: free since the last barrier
        dma_slab((tap + 2) % 27, (tap + 2) % 3);
        if (tap < JR) hrow[tap % JR] = halo_load(0, tap % JR);
        if (tap >= 9 && tap < 9 + JR) hrow[(tap - 9) % JR] = halo_load(1, (tap - 9) % JR);
        if (tap < NLATE) hlate[tap % NLATE] = halo_load(2 + (tap % NLATE) / JR, (tap % NLATE) % JR);
        // slab tap+1 (DMA issued one tap ago) must have landed before the
        // consumers pass this barrier.  Younger vector-memory ops: the halo
        // chunks of the previous tap (issued after its DMA), this tap's 2 DMA
        // pieces and its chunks.
        wait_vm_n(2 + PG::halo_k(tap) + PG::halo_k(tap - 1));
        WG_BARRIER();
      }
      // every consumer is past its last halo read: rows 2..5 may land
      if (has_next) {
#pragma unroll
        for (int u = 0; u < NLATE; ++u) halo_put(2 + u / JR, u % JR, hlate[u]);
      }
      WAIT_LGKM0();
      WG_BARRIER();
#undef has_next
    }
#undef HALO_TABLE
    return;
  }

  // ===================================================== consumer waves
  const int frow = lane & 15, kq = lane >> 4;
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
    const int rho = nf * 16 + frow;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      b_addr[nf][ks] = (unsigned)(SLAB_OFF + rho * 128 + (((ks * 4 + kq) ^ ((rho >> 1) & 7)) << 4));
  }
  const float slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
  // this lane's two 8-channel chunks: block (bi, bj) and channel offset cc of
  // the depth-to-space store (b = 1: bi = bj = 0, cc = the channel itself)
  // (the repeat variants are plain 64 -> 64 trunk convs: no depth-to-space
  // store — two scalars less, which is what keeps them free of spills)
  const int db = REP > 1 ? 1 : g.d2s, cpo = REP > 1 ? g.Cout : g.Cout / (db * db);
  unsigned c_off[2];
  bool c_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int co = ct * 64 + h * 32 + kq * 8;
    c_ok[h] = co < g.Cout;
    const int blk = co / cpo;
    c_off[h] = (unsigned)((((blk / db) * (g.O[1] * db) + blk % db) * g.O[2]) * cpo + co % cpo);
  }
  WG_BARRIER();   // prologue
  // the matrix-core waves outrank the producer wave sharing their SIMD
  // (A/B in one run: 0.1228 -> 0.1055 ms per 64->64 conv launch)
  __builtin_amdgcn_s_setprio(2);

  const int my_row = mf0 / TS1;                 // s0 row of this wave's fragments
  for (int h = h_cur; h < h_end;) {
    int nr, n, org0, org1, org2;
    const int tile = h;                         // (the item's first half-tile)
    tile_org(tile, n, org0, org1, org2);
    {
      int hs0 = org0 >> 1;
      h = item_at(h, hs0, nr);
    }
    if (my_row >= nr) {
      // half-tile item: the second row pair idles, keeping the barrier count
#pragma unroll 1
      for (int t = 0; t < 28; ++t) WG_BARRIER();
      continue;
    }
    f32x4 acc[MFW][NFV];
#pragma unroll
    for (int nf = 0; nf < NFV; ++nf) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (nf * 16 + kq * 4) * 4);
#pragma unroll
      for (int m = 0; m < MFW; ++m) acc[m][nf] = bv;
    }

    // The halo is static during a tile, so the A (position) fragments of the
    // NEXT tap's first k-step are fetched before this tap's barrier: after the
    // barrier only the four filter fragments stand between a wave and its
    // first MFMA.
    bf16x8 apre[MFW];
#pragma unroll
    for (int m = 0; m < MFW; ++m)
      apre[m] = *reinterpret_cast<const bf16x8*>(smem + a_addr[0][0] + (m * H2) * 128);
#pragma unroll 1
    for (int ta = 0; ta < 3; ++ta) {
      const unsigned ta_off = (unsigned)(ta * H1 * H2 * 128);
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int tc = 0; tc < 3; ++tc) {
          // ring slot of tap = (ta*9 + tb*3 + tc) % 3 = tc
          bf16x8 bfr[NFV];
          // ---- k-step 0: prefetched A, fresh B
#pragma unroll
          for (int nf = 0; nf < NFV; ++nf)
            bfr[nf] = *reinterpret_cast<const bf16x8*>(smem + b_addr[nf][0] + tc * 8192);
#pragma unroll
          for (int m = 0; m < MFW; ++m)
#pragma unroll
            for (int nf = 0; nf < NFV; ++nf)
              acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nf], apre[m], acc[m][nf], 0, 0, 0);
          // ---- k-step 1
#pragma unroll
          for (int nf = 0; nf < NFV; ++nf)
            bfr[nf] = *reinterpret_cast<const bf16x8*>(smem + b_addr[nf][1] + tc * 8192);
#pragma unroll
          for (int m = 0; m < MFW; ++m) {
            const bf16x8 afr = *reinterpret_cast<const bf16x8*>(
                smem + a_addr[tc][1] + ta_off + ((m + tb) * H2) * 128);
#pragma unroll
            for (int nf = 0; nf < NFV; ++nf)
              acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nf], afr, acc[m][nf], 0, 0, 0);
          }
          // ---- A fragments of the next tap (k-step 0)
          {
            const int ntc = tc < 2 ? tc + 1 : 0;
            const int ntb = tc < 2 ? tb : (tb < 2 ? tb + 1 : 0);
            const unsigned nta_off = (tc == 2 && tb == 2) ? ta_off + H1 * H2 * 128 : ta_off;
            if (!(tc == 2 && tb == 2) || ta < 2) {
#pragma unroll
              for (int m = 0; m < MFW; ++m)
                apre[m] = *reinterpret_cast<const bf16x8*>(
                    smem + a_addr[ntc][0] + nta_off + ((m + ntb) * H2) * 128);
            }
          }
          WG_BARRIER();
        }
      }
    }

    if constexpr (DG) {
      // ---- fp32 store over the stacked frames: S = q E + u per axis
      float* yf = reinterpret_cast<float*>(y);
      const int E0 = g.O[0], E1 = g.O[1];
#pragma unroll
      for (int m = 0; m < MFW; ++m) {
        const int mf = mf0 + m;
        const int S0 = org0 + mf / TS1, S1 = org1 + mf % TS1, o2 = org2 + frow;
        if (S0 >= gs0 * E0 || S1 >= gs1 * E1 || o2 >= g.O[2]) continue;
        const int q0 = S0 / E0, u0 = S0 - q0 * E0, q1 = S1 / E1, u1 = S1 - q1 * E1;
        const size_t yo = ((((size_t)(q0 * gs1 + q1) * E0 + u0) * E1 + u1) * g.O[2] + o2) * g.Cout + kq * 8;
        float* yp = yf + yo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (2 * h >= NFV) continue;
          f32x4 a0 = acc[m][(2 * h) % NFV], a1 = acc[m][(2 * h + 1) % NFV];
          if constexpr (F16) {
            uint4 o;
            o.x = pk_bf16(a0[0], a0[1]); o.y = pk_bf16(a0[2], a0[3]);
            o.z = pk_bf16(a1[0], a1[1]); o.w = pk_bf16(a1[2], a1[3]);
            *reinterpret_cast<uint4*>(y + yo + h * 32) = o;
            continue;
          }
          if (res) {   // a later channel slice of the contraction: add to the earlier ones
            const float4 p0 = *reinterpret_cast<const float4*>(yp + h * 32);
            const float4 p1 = *reinterpret_cast<const float4*>(yp + h * 32 + 4);
            a0[0] += p0.x; a0[1] += p0.y; a0[2] += p0.z; a0[3] += p0.w;
            a1[0] += p1.x; a1[1] += p1.y; a1[2] += p1.z; a1[3] += p1.w;
          }
          *reinterpret_cast<float4*>(yp + h * 32) = make_float4(a0[0], a0[1], a0[2], a0[3]);
          *reinterpret_cast<float4*>(yp + h * 32 + 4) = make_float4(a1[0], a1[1], a1[2], a1[3]);
        }
      }
      WG_BARRIER();   // next halo visible
      continue;
    }
    // ---- epilogue straight from the accumulators.  A lane owns two
    // 8-channel chunks per position.  With a depth-to-space store (block b,
    // C_out / b^2 channels per hi-res cell, a multiple of 8) a chunk is one
    // hi-res cell's channels: block (i, j) = chunk / (C_out / b^2) lands at
    // (o0 b + i, o1 b + j, o2).  Addresses are element offsets within sample n.
    const size_t e_base = (size_t)n * g.O[0] * g.O[1] * g.O[2] * g.Cout;
    // offset(m, h) = position part (m) + chunk part (h): both per-lane scalars
    unsigned e_pos[MFW];
    bool e_ok[MFW];
#pragma unroll
    for (int m = 0; m < MFW; ++m) {
      const int mf = mf0 + m;
      const int o0 = org0 + mf / TS1, o1 = org1 + mf % TS1, o2 = org2 + frow;
      e_ok[m] = o0 < g.O[0] && o1 < g.O[1] && o2 < g.O[2];
      e_pos[m] = (unsigned)(((o0 * db * (g.O[1] * db) + o1 * db) * g.O[2] + o2) * cpo);
    }
    auto chunk_off = [&](int m, int h, bool& ok) __attribute__((always_inline)) {
      ok = c_ok[h] && e_ok[m] && 2 * h < NFV;
      return e_pos[m] + c_off[h];
    };
    // residual rows first (all loads in flight together), then activation +
    // add + 16-B stores
    uint4 rres[MFW][2];
    if (res) {
      // a residual read through a fused temporal repeat (d2s == 1, cpo = 64):
      // cell o2 of the skip tensor is cell o2 / res_rep of what is stored
      const bool rr = REP > 1 && !DG && g.res_rep > 1;       // (then res_rep == REP)
      const int O2s = rr ? g.O[2] / (REP > 1 ? REP : 1) : g.O[2];
      const size_t r_base = rr ? (size_t)n * g.O[0] * g.O[1] * O2s * g.Cout : e_base;
#pragma unroll
      for (int m = 0; m < MFW; ++m) {
        unsigned r_pos = e_pos[m];
        if (rr) {
          const int mf = mf0 + m;
          const int o0 = org0 + mf / TS1, o1 = org1 + mf % TS1, o2 = org2 + frow;
          r_pos = (unsigned)(((o0 * g.O[1] + o1) * O2s + o2 / (REP > 1 ? REP : 1)) * cpo);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          bool ok;
          chunk_off(m, h, ok);
          rres[m][h] = make_uint4(0, 0, 0, 0);
          if (ok) rres[m][h] = *reinterpret_cast<const uint4*>(res + r_base + r_pos + c_off[h]);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MFW; ++m) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bool ok;
        const unsigned off = chunk_off(m, h, ok);
        if (!ok) continue;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          // branch-free activation: max(v, slope v) is identity (slope 1),
          // ReLU (0) or LeakyReLU (0 <= alpha <= 1); a per-element test of
          // the kind compiles to two scalar branches per value
          const float a = acc[m][(2 * h + (q >> 2)) % NFV][q & 3];
          // (v_max_f32 by hand: fmaxf() canonicalises the accumulator first —
          // a second v_max per value)
          const float sa = slope * a;
          asm("v_max_f32 %0, %1, %2" : "=v"(v[q]) : "v"(a), "v"(sa));
        }
        if (res) {
          const uint4 r = rres[m][h];
          v[0] += lo_f(r.x); v[1] += hi_f(r.x); v[2] += lo_f(r.y); v[3] += hi_f(r.y);
          v[4] += lo_f(r.z); v[5] += hi_f(r.z); v[6] += lo_f(r.w); v[7] += hi_f(r.w);
        }
        uint4 o;
        o.x = pk_bf16(v[0], v[1]); o.y = pk_bf16(v[2], v[3]);
        o.z = pk_bf16(v[4], v[5]); o.w = pk_bf16(v[6], v[7]);
        *reinterpret_cast<uint4*>(y + e_base + off) = o;
      }
    }
    WG_BARRIER();   // next halo visible
  }
}

}  // namespace

bool conv_mfma_persist_geom_ok(const ConvGeom& g) {
  if (g.Cin != 64 || g.Cout % 8 != 0 || g.Cout < 64 || g.d2s < 1) return false;
  if (g.D[2] < 8) return false;   // (few time steps: the logical-axes tile kernel, a2 = s2)
  if (g.d2s > 1 && (g.Cout % (g.d2s * g.d2s) != 0 || (g.Cout / (g.d2s * g.d2s)) % 8 != 0))
    return false;
  if (g.pad_mode != S3_PAD_REFLECT) return false;
  // the epilogue's branch-free max(v, alpha v)
  if (g.act == S3_ACT_LEAKY && !(g.alpha >= 0.f && g.alpha <= 1.f)) return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1) return false;
  // 32-bit element offsets inside one sample
  return (int64_t)g.D[0] * g.D[1] * g.D[2] * 64 < (int64_t)1 << 31 &&
         (int64_t)g.O[0] * g.O[1] * g.O[2] * g.Cout < (int64_t)1 << 31;
}

// data-gradient geometry of a 64 -> 64 'same' k3 conv (conv_dgrad_geom): full
// correlation over the padded frame, zero boundary
bool conv_mfma_persist_dgrad_geom_ok(const ConvGeom& g) {
  // (C_out = 32: the data gradient of a valid 32 -> 64 conv, two N fragments)
  if (g.Cin != 64 || (g.Cout != 64 && g.Cout != 32) || g.d2s != 1 || g.pad_mode != S3_PAD_ZERO ||
      g.act != S3_ACT_NONE)
    return false;
  for (int d = 0; d < 3; ++d)
    if (g.k[d] != 3 || g.s[d] != 1 || g.lo[d] != 2 || g.O[d] != g.D[d] + 2) return false;
  // few time steps: the logical-axes kernel, as in the forward (conv_mfma_persist_geom_ok).  Found at the end of
  // round 6: sup3rcc/gen_solar_1x_8x_1f (3 time steps) trained at 8 / 16 samples of (54, 54, 3) — enough frame
  // tiles for this kernel — gave non-finite gradients / a memory access fault; 4 and 6 steps likewise, 8 not
  // (tools/dbg/train_plan_probe.py; the census trains at a quarter of its inference batch and never got here)
  if (g.D[2] < 8) return false;
  // a 64-channel slice of a wider dPre (the 64 -> 200 conv): whole 16-B chunks
  if (g.in_cstride && ((g.in_cstride & 7) || (g.in_cvalid & 7) || g.in_cvalid < 8 || g.in_cvalid > 64)) return false;
  // 30-bit element offsets over the whole batch (the zero flag is bit 30)
  return (int64_t)g.N * g.D[0] * g.D[1] * g.D[2] * (g.in_cstride ? g.in_cstride : 64) < (int64_t)1 << 28;
}

// frames per stacked axis: N = gs0 * gs1 with the least tile overhang
static void persist_dgrad_grid(const ConvGeom& g, int* gs0, int* gs1) {
  int64_t best = -1;
  for (int a = 1; a <= g.N; ++a) {
    if (g.N % a) continue;
    const int b = g.N / a;
    // (half-tile units: two s0 rows x TS1 columns)
    const int64_t t = (int64_t)((a * g.O[0] + 1) / 2) * ((b * g.O[1] + TS1 - 1) / TS1);
    if (best < 0 || t < best) { best = t; *gs0 = a; *gs1 = b; }
  }
}

bool conv_mfma_persist_dgrad_supported(const s3_ctx* ctx, const ConvGeom& g) {
  if (s3_opt_on(S3O_NO_PERSIST_DGRAD)) return false;
  if (!conv_mfma_persist_dgrad_geom_ok(g)) return false;
  int gs0 = 1, gs1 = 1;
  persist_dgrad_grid(g, &gs0, &gs1);
  const int64_t halves = (int64_t)((gs0 * g.O[0] + 1) / 2) * ((gs1 * g.O[1] + TS1 - 1) / TS1) *
                         ((g.O[2] + TS2 - 1) / TS2);
  // (the "fills the chip" threshold keeps counting whole 4-row tiles, as the
  // plans and their tests were tuned with)
  const int64_t tiles = (int64_t)((gs0 * g.O[0] + TS0 - 1) / TS0) * ((gs1 * g.O[1] + TS1 - 1) / TS1) *
                        ((g.O[2] + TS2 - 1) / TS2);
  // worth it unless the stacked 2 x 8 x 16 half-tiles cover clearly more than the
  // halo-tile kernel's 6 x 6 x 16 ones would (both share the overhang along t)
  const int64_t covered = halves * 2 * TS1 * TS2;
  const int64_t six = (int64_t)g.N * ((g.O[0] + 5) / 6) * ((g.O[1] + 5) / 6) * ((g.O[2] + 15) / 16) * 576;
  const int64_t min_tiles = s3_opt_has(S3O_PERSIST_DGRAD_MIN_TILES)
                                ? s3_opt_int(S3O_PERSIST_DGRAD_MIN_TILES, 0) : ctx->num_cu;
  // (not lowered with the forward's threshold: 7/16 per CU was 8 % faster on gen_3x_4x_2f at batch 3 and nothing
  // at the BASELINE shapes; the fault tools/config_census.py found with it was the few-time-step case above)
  return tiles >= min_tiles && covered * 10 <= six * 11;
}

int launch_conv_mfma_persist_dgrad(s3_ctx* ctx, const ConvGeom& g, const void* dpre16, const void* image,
                                   float* dxp, int accumulate, int frame16) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<2, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, true, 0, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set.mark(ctx->device);
  }
  if (frame16 && (accumulate || g.Cout != 64))
    S3_FAIL(ctx, S3_ESTATE, "persistent data gradient: a bf16 frame is written once, 64 channels wide");
  int gs0 = 1, gs1 = 1;
  persist_dgrad_grid(g, &gs0, &gs1);
  // (tiles0 = half rows along the stacked s0 axis, n_tiles = half-tiles)
  const int tiles0 = (gs0 * g.O[0] + 1) / 2, tiles1 = (gs1 * g.O[1] + TS1 - 1) / TS1,
            tiles2 = (g.O[2] + TS2 - 1) / TS2;
  const int n_tiles = tiles0 * tiles1 * tiles2;
  int grid = ctx->num_cu;
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  auto kern = g.Cout <= 32 ? conv3_mfma_persist_kernel<2, true> : conv3_mfma_persist_kernel<4, true>;
  if (frame16) kern = conv3_mfma_persist_kernel<4, true, 0, false, true>;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), LDS_BYTES, ctx->stream,
                     (const unsigned short*)dpre16, (const char*)image, (const float*)nullptr,
                     (const unsigned short*)(accumulate ? dxp : nullptr), (unsigned short*)dxp, g, tiles0, tiles1,
                     tiles2, n_tiles, 0, gs0, gs1);
  S3_HIP(ctx, hipGetLastError());
  ++ctx->stat[S3_STAT_PERSIST_DGRAD];
  return S3_OK;
}

bool conv_mfma_persist_rep_ok(int rep) { return rep == 2 || rep == 3 || rep == 4; }

bool conv_mfma_persist_supported(const s3_ctx* ctx, const ConvGeom& g, ConvIO io,
                                 bool has_res) {
  // read per call: the parity tests flip it between two forwards
  if (s3_opt_on(S3O_NO_PERSIST)) return false;
  if (!io.in_bf16 || !io.out_bf16 || (has_res && !io.res_bf16)) return false;
  if (!conv_mfma_persist_geom_ok(g)) return false;
  if (has_res && g.d2s != 1) return false;
  // (counted in whole 4-row tiles, as before the half-row work list)
  const int64_t tiles = (int64_t)g.N * ((g.O[0] + TS0 - 1) / TS0) *
                        ((g.O[1] + TS1 - 1) / TS1) * ((g.O[2] + TS2 - 1) / TS2);
  // Below ~ 7/16 of a tile per CU the one-tile-per-workgroup kernel with its 64-position tiles
  // fills the chip better (48 / 96 tiles: 20 - 27 us against 26 - 29); from there on a whole
  // 512-position tile per workgroup wins (144 / 192 tiles: 29 - 32 us against 37 - 40: the
  // trunk of gen_3x_4x_2f at lr (4, 16, 16, 24), profiles/r06/README.md).  Was: one per CU.
  return tiles >= s3_opt_int(S3O_PERSIST_MIN_TILES, ctx->num_cu * 7 / 16);
}

int launch_pack_jobs(s3_ctx* ctx, const S3PackJob* jobs_dev, int n_jobs, int max_ct) {
  if (n_jobs <= 0) return S3_OK;
  hipLaunchKernelGGL(pack_jobs_kernel, dim3(108 * max_ct, n_jobs), dim3(256), 0, ctx->stream, jobs_dev);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

size_t conv_mfma_persist_image_bytes(const ConvGeom& g) {
  return (size_t)((g.Cout + 63) / 64) * 27 * 64 * 64 * 2;
}

int launch_conv_mfma_persist_pack(s3_ctx* ctx, const ConvGeom& g, const float* w, void* image) {
  const int n_ct = (g.Cout + 63) / 64;
  hipLaunchKernelGGL(pack_persist_kernel, dim3(108 * n_ct), dim3(256), 0, ctx->stream,
                     w, (unsigned short*)image, g.Cout, n_ct);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}

int launch_conv_mfma_persist(s3_ctx* ctx, const ConvGeom& g, const void* x,
                             const void* image, const float* bias,
                             const void* res, void* y) {
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<2, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#define S3_REP_ATTR(R)                                                                                          \
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, false, R, true>),  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));                     \
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, false, R, false>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));                     \
    S3_HIP(ctx, hipFuncSetAttribute(                                                                              \
                    reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, false, R, true, false, 6>),        \
                    hipFuncAttributeMaxDynamicSharedMemorySize, PGeo<6>::LDS_BYTES));                             \
    S3_HIP(ctx, hipFuncSetAttribute(                                                                              \
                    reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, false, R, false, false, 6>),       \
                    hipFuncAttributeMaxDynamicSharedMemorySize, PGeo<6>::LDS_BYTES));
    S3_REP_ATTR(2) S3_REP_ATTR(3) S3_REP_ATTR(4)
#undef S3_REP_ATTR
    S3_HIP(ctx, hipFuncSetAttribute(
                    reinterpret_cast<const void*>(conv3_mfma_persist_kernel<4, false, 0, false, false, 6>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, PGeo<6>::LDS_BYTES));
    S3_HIP(ctx, hipFuncSetAttribute(
                    reinterpret_cast<const void*>(conv3_mfma_persist_kernel<2, false, 0, false, false, 6>),
                    hipFuncAttributeMaxDynamicSharedMemorySize, PGeo<6>::LDS_BYTES));
    attr_set.mark(ctx->device);
  }
  // (tiles0 = half rows along s0, n_tiles = half-tiles: the kernel's work units)
  const int tiles0 = (g.O[0] + 1) / 2, tiles1 = (g.O[1] + TS1 - 1) / TS1,
            tiles2 = (g.O[2] + TS2 - 1) / TS2;
  const int n_tiles = g.N * tiles0 * tiles1 * tiles2;
  int grid = ctx->num_cu;
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  const int n_ct = (g.Cout + 63) / 64;
  // (operands read through a fused temporal repeat: variants of their own with
  // the factor a compile-time constant — the index arithmetic with a run-time
  // factor does not fit the 168-register budget: 8 VGPRs spilled, and a spill
  // next to the hand-ordered in-flight loads is not safe)
  const int rep = g.in_rep > 1 ? g.in_rep : (g.res_rep > 1 ? g.res_rep : 0);
  const bool rin = g.in_rep > 1;
  ConvGeom gk = g;                 // what the kernel sees
  if (rep) {
    bool ok = conv_mfma_persist_rep_ok(rep) && g.Cout == 64 && g.d2s == 1 &&
              !(g.in_rep > 1 && g.res_rep > 1 && g.in_rep != g.res_rep);
    for (int d = 0; d < 3; ++d) ok = ok && g.lo[d] == g.lo[0] && g.O[d] == g.D[d];
    if (!ok) S3_FAIL(ctx, S3_ESTATE, "persistent conv: unsupported fused temporal repeat");
    if (rin) gk.D[2] = g.D[2] / rep;
  }
  // Column strips (round 4): an s1 extent with 1 .. 6 columns beyond a multiple
  // of 8 runs its last columns as ONE strip of 6-column tiles in a launch of
  // its own (TW = 6: 22 columns = 8 + 8 + 6 computed, not 24) — plain 64 -> 64
  // trunk convs only, and only when that strip still fills the chip.
  const int rem1 = g.O[1] % TS1;
  const int64_t strip_halves = (int64_t)g.N * tiles0 * tiles2;
  // (the fused-repeat variants too since the end of round 4: the three convs
  // behind the temporal repeats of a C3 chunk computed 24 columns for 22)
  if (rem1 >= 1 && rem1 <= 6 && g.O[1] > TS1 && strip_halves >= 2 * (int64_t)ctx->num_cu &&
      (!rep || n_ct == 1) && !s3_opt_on(S3O_NO_PERSIST_STRIP)) {
    const int t1a = g.O[1] / TS1;
    const int na = g.N * tiles0 * t1a * tiles2, nb = (int)strip_halves;
    int ga = ctx->num_cu, gb = ctx->num_cu;
    if (ga > (na + 1) / 2) ga = (na + 1) / 2;
    if (gb > (nb + 1) / 2) gb = (nb + 1) / 2;
    for (int ct = 0; ct < n_ct; ++ct) {          // (C_out > 64: one pair of launches per channel tile)
      const bool half = g.Cout - ct * 64 <= 32;
      const char* img = (const char*)image + (size_t)ct * 27 * 8192;
      auto k8 = half ? conv3_mfma_persist_kernel<2, false> : conv3_mfma_persist_kernel<4, false>;
      auto k6 = half ? conv3_mfma_persist_kernel<2, false, 0, false, false, 6>
                     : conv3_mfma_persist_kernel<4, false, 0, false, false, 6>;
      if (rep == 2) {
        k8 = rin ? conv3_mfma_persist_kernel<4, false, 2, true> : conv3_mfma_persist_kernel<4, false, 2, false>;
        k6 = rin ? conv3_mfma_persist_kernel<4, false, 2, true, false, 6>
                 : conv3_mfma_persist_kernel<4, false, 2, false, false, 6>;
      } else if (rep == 3) {
        k8 = rin ? conv3_mfma_persist_kernel<4, false, 3, true> : conv3_mfma_persist_kernel<4, false, 3, false>;
        k6 = rin ? conv3_mfma_persist_kernel<4, false, 3, true, false, 6>
                 : conv3_mfma_persist_kernel<4, false, 3, false, false, 6>;
      } else if (rep == 4) {
        k8 = rin ? conv3_mfma_persist_kernel<4, false, 4, true> : conv3_mfma_persist_kernel<4, false, 4, false>;
        k6 = rin ? conv3_mfma_persist_kernel<4, false, 4, true, false, 6>
                 : conv3_mfma_persist_kernel<4, false, 4, false, false, 6>;
      }
      hipLaunchKernelGGL(k8, dim3(ga), dim3(NTHR), LDS_BYTES, ctx->stream, (const unsigned short*)x, img, bias,
                         (const unsigned short*)res, (unsigned short*)y, gk, tiles0, t1a, tiles2, na, ct, 1, 0);
      hipLaunchKernelGGL(k6, dim3(gb), dim3(NTHR), PGeo<6>::LDS_BYTES, ctx->stream, (const unsigned short*)x,
                         img, bias, (const unsigned short*)res, (unsigned short*)y, gk, tiles0, 1, tiles2, nb,
                         ct, 1, t1a * TS1);
    }
    S3_HIP(ctx, hipGetLastError());
    return S3_OK;
  }
  for (int ct = 0; ct < n_ct; ++ct) {
    // a last tile with <= 32 valid channels computes two N fragments only
    const bool half = g.Cout - ct * 64 <= 32;
    auto kern = half ? conv3_mfma_persist_kernel<2, false> : conv3_mfma_persist_kernel<4, false>;
    if (rep == 2) kern = rin ? conv3_mfma_persist_kernel<4, false, 2, true> : conv3_mfma_persist_kernel<4, false, 2, false>;
    else if (rep == 3) kern = rin ? conv3_mfma_persist_kernel<4, false, 3, true> : conv3_mfma_persist_kernel<4, false, 3, false>;
    else if (rep == 4) kern = rin ? conv3_mfma_persist_kernel<4, false, 4, true> : conv3_mfma_persist_kernel<4, false, 4, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), LDS_BYTES, ctx->stream,
                       (const unsigned short*)x, (const char*)image + (size_t)ct * 27 * 8192, bias,
                       (const unsigned short*)res, (unsigned short*)y, gk, tiles0,
                       tiles1, tiles2, n_tiles, ct, 1, 0);
  }
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
