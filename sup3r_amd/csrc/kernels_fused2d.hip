// Whole-network kernel for small 2-D (spatial) generators — BASELINE config C1
// (`spatial/gen_2x_2f.json`: 36 x [REFLECT pad 3 -> Conv2DTranspose k3 -> crop
// 4] on a 10 x 10 field, 0.27 GFLOP per observation).  As 36 dependent
// launches the forward costs 0.52 ms whatever the batch (every launch is ~14 us
// of latency for ~1 us of work; hipGraph replay does not change that, DESIGN
// §5.3).  Here ONE launch runs the whole layer list: a workgroup per
// observation keeps every activation in LDS (a 10 x 10 x 64 bf16 tensor with
// its reflect border is 18 KB, the 20 x 20 hi-res one 62 KB), the filters
// (2.7 MB as bf16, L2-resident) stream from global memory straight into MFMA
// A fragments with a 3-deep register prefetch, and the only HBM traffic is the
// lo-res input and the hi-res output.
//
// Layout.  A tensor (H, W, C <= 64) lives in an LDS slot as (H + 2) x (W + 2)
// cells of 128 B (64 bf16 channels, 16-B chunks XOR-swizzled by cell & 7) WITH
// its padding border materialised, so a tap is a constant cell offset and a
// 3 x 3 conv over the image is a 1-D conv over the flattened padded array:
// output "positions" are 16 consecutive flattened cells per MFMA fragment
// (border / wrap-around cells are computed and dropped at the store; a 10 x 10
// image needs 8 fragments instead of 10 row-fragments).  MFMA operands: A =
// 16 output channels x 32 input channels of one tap (global, 1 KB per wave
// load), B = 16 positions x 32 channels (one ds_read_b128 per lane), D[co][pos]:
// a lane owns 4 consecutive channels of one position -> one ds_write_b64 into
// the destination slot, with bias, activation, residual and the depth-to-space
// permutation applied on the way.  After every layer the border of the new
// tensor is filled (reflect / zero) for its consumer.
#include <cstdlib>
#include <vector>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x2 __attribute__((ext_vector_type(2)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));

constexpr int FNW = 8;            // waves per workgroup
constexpr int FNT = FNW * 64;
constexpr int FMG = 4;            // M fragments (16 positions each) per work item
constexpr int FMAX_LAYERS = 96;

// one fused conv as the device sees it
struct FL {
  int ck;          // k-steps of 32 input channels per tap
  int cout;        // output channels (before depth-to-space)
  int n_nf;        // ceil(cout / 16)
  int src, dst, res;   // LDS byte offsets of the slots; dst < 0: global output; res < 0: none
  int H, W;        // image the conv runs over (= source tensor's interior)
  int d2s;         // 1 | block size b: dst is (H b, W b, cout / b^2)
  int act;
  float slope;
  int border;      // how the consumer pads dst: 0 zero, 1 reflect (dst in LDS)
  unsigned img;    // byte offset of this layer's packed filter image
  int bias;        // float offset into the weight buffer, -1: none
  int cin;         // input channels
  int w_off;       // float offset of the canonical filter in the weight buffer
};

__device__ __forceinline__ unsigned pk2(float a, float b) {
  hf32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hbf16x2));
}
__device__ __forceinline__ float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// canonical fp32 w[tap 9][cin][cout] -> bf16 [nf][tap][ks][row 16][k 32]
__global__ void fused2d_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ img,
                                    int cin, int cout, int ck, int n_nf) {
  const int total = n_nf * 9 * ck * 512;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx & 31, row = (idx >> 5) & 15;
    int r = idx >> 9;
    const int ks = r % ck; r /= ck;
    const int tap = r % 9; r /= 9;
    const int nf = r;
    const int ci = ks * 32 + k, co = nf * 16 + row;
    const float v = (ci < cin && co < cout) ? w[((size_t)tap * cin + ci) * cout + co] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

// ... every layer of the stack in ONE launch (blockIdx.y = layer): a training
// mini-batch re-packs after each weight update, and 36 dependent 4.6 us
// launches were as long as the fused forward itself
__global__ void fused2d_pack_all_kernel(const float* __restrict__ W, char* __restrict__ img_base,
                                        const FL* __restrict__ L) {
  const FL f = L[blockIdx.y];
  const float* w = W + f.w_off;
  unsigned short* img = reinterpret_cast<unsigned short*>(img_base + f.img);
  const int total = f.n_nf * 9 * f.ck * 512;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx & 31, row = (idx >> 5) & 15;
    int r = idx >> 9;
    const int ks = r % f.ck; r /= f.ck;
    const int tap = r % 9; r /= 9;
    const int nf = r;
    const int ci = ks * 32 + k, co = nf * 16 + row;
    const float v = (ci < f.cin && co < f.cout) ? w[((size_t)tap * f.cin + ci) * f.cout + co] : 0.f;
    img[idx] = (unsigned short)(pk2(v, 0.f) & 0xFFFFu);
  }
}

// border of the (H + 2) x (W + 2) tensor in `slot`: reflect (pad 1) or zero
__device__ __forceinline__ void fill_border(char* slot, int H, int W, int reflect, int tid) {
  const int Wp = W + 2;
  const int nb = 2 * Wp + 2 * H;
  for (int item = tid; item < nb * 8; item += FNT) {
    const int bc = item >> 3, ch = item & 7;
    int r, c;
    if (bc < Wp) { r = 0; c = bc; }
    else if (bc < 2 * Wp) { r = H + 1; c = bc - Wp; }
    else { const int k = bc - 2 * Wp; r = 1 + (k >> 1); c = (k & 1) ? W + 1 : 0; }
    const int sr = r == 0 ? 2 : (r == H + 1 ? H - 1 : r);
    const int sc = c == 0 ? 2 : (c == W + 1 ? W - 1 : c);
    const int dcell = r * Wp + c, scell = sr * Wp + sc;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (reflect) v = *reinterpret_cast<const uint4*>(slot + scell * 128 + ((ch ^ (scell & 7)) << 4));
    *reinterpret_cast<uint4*>(slot + dcell * 128 + ((ch ^ (dcell & 7)) << 4)) = v;
  }
}

template <int CK>
__device__ __forceinline__ void run_layer(const FL& L, const char* __restrict__ img,
                                          const float* __restrict__ wbuf, char* smem,
                                          float* __restrict__ yn, int tid) {
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, kq = lane >> 4;
  const int Wp = L.W + 2;
  const int q0 = Wp + 1;
  const int n_mf = ((L.H - 1) * Wp + L.W + 15) >> 4;
  const int n_mg = (n_mf + FMG - 1) / FMG;
  const int items = L.n_nf * n_mg;
  constexpr int KT = 9 * CK;
  const char* src = smem + L.src;
  for (int item = wave; item < items; item += FNW) {
    const int nf = item / n_mg, mg = item - nf * n_mg;
    f32x4 acc[FMG];
#pragma unroll
    for (int m = 0; m < FMG; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* ab = img + L.img + (size_t)nf * KT * 1024 + (lane & 15) * 64 + (lane >> 4) * 16;
    bf16x8 ring[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ring[i] = *reinterpret_cast<const bf16x8*>(ab + i * 1024);
    const int cbase = q0 + mg * (FMG * 16) + p;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      const int tap = i / CK, ks = i % CK;
      const int off = (tap / 3 - 1) * Wp + (tap % 3 - 1);
      const bf16x8 a = ring[i % 3];
      if (i + 3 < KT) ring[i % 3] = *reinterpret_cast<const bf16x8*>(ab + (i + 3) * 1024);
#pragma unroll
      for (int m = 0; m < FMG; ++m) {
        const int cell = cbase + m * 16 + off;
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(
            src + cell * 128 + (((ks * 4 + kq) ^ (cell & 7)) << 4));
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
      }
    }
    // ---- epilogue: lane (position p, channels nf*16 + 4 kq .. + 3)
    const int co0 = nf * 16 + kq * 4;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (L.bias >= 0 && co0 + r < L.cout) ? wbuf[L.bias + co0 + r] : 0.f;
#pragma unroll
    for (int m = 0; m < FMG; ++m) {
      const int q = cbase + m * 16;
      const int r = q / Wp, c = q - r * Wp;
      if (r < 1 || r > L.H || c < 1 || c > L.W || co0 >= L.cout) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = acc[m][e] + bv[e];
        v[e] = t > 0.f ? t : L.slope * t;
      }
      if (L.dst < 0) {
        float* yp = yn + ((size_t)(r - 1) * L.W + (c - 1)) * L.cout + co0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co0 + e < L.cout) yp[e] = v[e];
        continue;
      }
      int dcell, cc0;
      if (L.d2s > 1) {
        const int b = L.d2s, cpo = L.cout / (b * b);
        const int blk = co0 / cpo;
        cc0 = co0 - blk * cpo;
        dcell = ((r - 1) * b + blk / b + 1) * (L.W * b + 2) + (c - 1) * b + blk % b + 1;
      } else {
        cc0 = co0;
        dcell = q;
      }
      const unsigned daddr = (unsigned)(dcell * 128 + (((cc0 >> 3) ^ (dcell & 7)) << 4) + (cc0 & 7) * 2);
      if (L.res >= 0) {
        const uint2 rr = *reinterpret_cast<const uint2*>(smem + L.res + daddr);
        v[0] += lo_f(rr.x); v[1] += hi_f(rr.x); v[2] += lo_f(rr.y); v[3] += hi_f(rr.y);
      }
      *reinterpret_cast<uint2*>(smem + L.dst + daddr) = make_uint2(pk2(v[0], v[1]), pk2(v[2], v[3]));
    }
  }
}

__global__ __launch_bounds__(FNT) void fused2d_kernel(
    const float* __restrict__ x, float* __restrict__ y, const char* __restrict__ img,
    const float* __restrict__ wbuf, const FL* __restrict__ layers, int n_layers,
    int H0, int W0, int C0, int slot0, int border0, int slot0_cells, int64_t y_per_sample) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int n = blockIdx.x;
  // ---- the observation into slot 0: channels beyond C0 are zero (the first
  // conv contracts over 32)
  for (int i = tid; i < slot0_cells * 8; i += FNT)
    *reinterpret_cast<uint4*>(smem + slot0 + i * 16) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  {
    const float* xn = x + (size_t)n * H0 * W0 * C0;
    const int Wp = W0 + 2;
    for (int i = tid; i < H0 * W0 * C0; i += FNT) {
      const int ch = i % C0, pos = i / C0;
      const int r = pos / W0, c = pos - r * W0;
      const int cell = (r + 1) * Wp + c + 1;
      *reinterpret_cast<unsigned short*>(smem + slot0 + cell * 128 + (((ch >> 3) ^ (cell & 7)) << 4) +
                                         (ch & 7) * 2) = (unsigned short)(pk2(xn[i], 0.f) & 0xFFFFu);
    }
  }
  __syncthreads();
  fill_border(smem + slot0, H0, W0, border0, tid);
  __syncthreads();
  float* yn = y + (size_t)n * y_per_sample;
  for (int li = 0; li < n_layers; ++li) {
    const FL L = layers[li];
    if (L.ck == 1) run_layer<1>(L, img, wbuf, smem, yn, tid);
    else run_layer<2>(L, img, wbuf, smem, yn, tid);
    if (L.dst >= 0) {
      __syncthreads();
      const int b = L.d2s;
      fill_border(smem + L.dst, L.H * b, L.W * b, L.border, tid);
      __syncthreads();
    }
  }
}

}  // namespace

struct Fused2dPlan {
  std::vector<FL> host;
  std::vector<Fused2dLayer> in;
  FL* dev = nullptr;
  char* img = nullptr;
  size_t img_bytes = 0;
  uint64_t version = 0;
  int lds = 0, slot0 = 0, slot0_cells = 0, border0 = 1;
  int N = 0, H0 = 0, W0 = 0, C0 = 0;
  int64_t y_per_sample = 0;
};

// Eligible: an inference op list made of 2-D 3 x 3 'same' stride-1 convs only
// (virtual pad lo = 1 in both axes, t = 1), C_in <= 64, LDS-destined outputs
// with 16 | C_out' <= 64, one padding mode per tensor, everything resident in
// 160 KB of LDS.
Fused2dPlan* fused2d_build(s3_ctx* ctx, const std::vector<Fused2dLayer>& L, int n_tensors,
                           int in_tensor, int out_tensor) {
  if (s3_opt_has(S3O_NO_FUSED2D) || L.empty() || (int)L.size() > FMAX_LAYERS) return nullptr;
  std::vector<int> th(n_tensors, 0), tw(n_tensors, 0), tc(n_tensors, 0), border(n_tensors, -1),
      last_use(n_tensors, -1), prod(n_tensors, -1);
  const int nl = (int)L.size();
  for (int i = 0; i < nl; ++i) {
    const ConvGeom& g = L[i].g;
    if (g.k[0] != 3 || g.k[1] != 3 || g.k[2] != 1 || g.D[2] != 1 || g.O[2] != 1) return nullptr;
    if (g.s[0] != 1 || g.s[1] != 1 || g.lo[0] != 1 || g.lo[1] != 1 || g.lo[2] != 0) return nullptr;
    if (g.O[0] != g.D[0] || g.O[1] != g.D[1] || g.D[0] < 3 || g.D[1] < 3) return nullptr;
    if (g.Cin > 64 || g.Cout < 1 || g.Cout > 1024) return nullptr;
    if (g.act == S3_ACT_LEAKY && !(g.alpha >= 0.f)) return nullptr;
    const int b = g.d2s < 1 ? 1 : g.d2s;
    const bool to_global = L[i].out_t == out_tensor;
    if (to_global && (i != nl - 1 || b != 1 || L[i].res_t >= 0)) return nullptr;
    if (!to_global) {
      const int cpo = g.Cout / (b * b);
      // (a consumer contracts over whole 32-channel k-steps: no partial cells)
      if (g.Cout % (b * b) || cpo % 32 || cpo > 64) return nullptr;
      if (L[i].res_t >= 0 && b != 1) return nullptr;
    }
    const int mode = g.pad_mode == S3_PAD_REFLECT ? 1 : 0;
    const int ti = L[i].in_t;
    if (border[ti] >= 0 && border[ti] != mode) return nullptr;   // two consumers, two paddings
    border[ti] = mode;
    th[ti] = g.D[0]; tw[ti] = g.D[1]; tc[ti] = g.Cin;
    last_use[ti] = i;
    if (L[i].res_t >= 0) last_use[L[i].res_t] = i;
    prod[L[i].out_t] = i;
    if (!to_global) { th[L[i].out_t] = g.D[0] * b; tw[L[i].out_t] = g.D[1] * b; }
  }
  if (L[nl - 1].out_t != out_tensor) return nullptr;
  if (L[0].in_t != in_tensor) return nullptr;
  // every tensor read must be the graph input or produced by an earlier fused conv
  for (int i = 0; i < nl; ++i) {
    const int ids[2] = {L[i].in_t, L[i].res_t};
    for (int q = 0; q < 2; ++q) {
      const int t = ids[q];
      if (t < 0) continue;
      if (t != in_tensor && (prod[t] < 0 || prod[t] >= i)) return nullptr;
      if (q == 1 && (th[t] != L[i].g.D[0] || tw[t] != L[i].g.D[1])) return nullptr;
    }
  }
  // ---- LDS slots by liveness (greedy first fit); a slot = the padded cells.
  // The fragments that run past a tensor's last row read into whatever follows
  // (another slot, or the slack after the last one): garbage that only reaches
  // positions dropped at the store.
  constexpr int kSlackCells = FMG * 16 + 32;
  auto cells_of = [&](int t) { return (th[t] + 2) * (tw[t] + 2); };
  std::vector<int> slot_off(n_tensors, -1);
  struct Slot { int off, bytes, free_at; };
  std::vector<Slot> slots;
  int total = 0;
  auto place = [&](int t, int at) {
    const int need = cells_of(t) * 128;
    for (auto& s : slots)
      if (s.free_at < at && s.bytes >= need) { s.free_at = last_use[t]; slot_off[t] = s.off; return; }
    slots.push_back({total, need, last_use[t]});
    slot_off[t] = total;
    total += need;
  };
  place(in_tensor, -1);
  for (int i = 0; i < nl; ++i)
    if (L[i].out_t != out_tensor) {
      if (last_use[L[i].out_t] < 0) return nullptr;   // dead tensor: not worth handling
      place(L[i].out_t, i);
    }
  total += kSlackCells * 128;
  if (total > 158 * 1024) return nullptr;
  Fused2dPlan* P = new Fused2dPlan();
  P->in = L;
  P->lds = total;
  P->slot0 = slot_off[in_tensor];
  P->slot0_cells = cells_of(in_tensor);
  P->border0 = border[in_tensor];
  P->N = L[0].g.N; P->H0 = L[0].g.D[0]; P->W0 = L[0].g.D[1]; P->C0 = L[0].g.Cin;
  const ConvGeom& gl = L[nl - 1].g;
  P->y_per_sample = (int64_t)gl.D[0] * gl.D[1] * gl.Cout;
  size_t img = 0;
  for (int i = 0; i < nl; ++i) {
    const ConvGeom& g = L[i].g;
    FL f;
    f.ck = (g.Cin + 31) / 32;
    f.cout = g.Cout;
    f.n_nf = (g.Cout + 15) / 16;
    f.src = slot_off[L[i].in_t];
    f.dst = L[i].out_t == out_tensor ? -1 : slot_off[L[i].out_t];
    f.res = L[i].res_t >= 0 ? slot_off[L[i].res_t] : -1;
    f.H = g.D[0]; f.W = g.D[1];
    f.d2s = g.d2s < 1 ? 1 : g.d2s;
    f.act = g.act;
    f.slope = g.act == S3_ACT_LEAKY ? g.alpha : (g.act == S3_ACT_RELU ? 0.f : 1.f);
    f.border = f.dst >= 0 ? (border[L[i].out_t] < 0 ? 0 : border[L[i].out_t]) : 0;
    f.img = (unsigned)img;
    f.bias = L[i].b_off >= 0 ? (int)L[i].b_off : -1;
    f.cin = g.Cin;
    f.w_off = (int)L[i].w_off;
    img += (size_t)f.n_nf * 9 * f.ck * 1024;
    P->host.push_back(f);
  }
  P->img_bytes = img;
  if (hipMalloc((void**)&P->dev, sizeof(FL) * nl) != hipSuccess ||
      hipMalloc((void**)&P->img, img + 4096) != hipSuccess) {
    fused2d_free(P);
    return nullptr;
  }
  (void)hipMemsetAsync(P->img, 0, img + 4096, ctx->stream);   // (the prefetch reads 2 KB past a layer)
  (void)hipMemcpyAsync(P->dev, P->host.data(), sizeof(FL) * nl, hipMemcpyHostToDevice, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  return P;
}

void fused2d_free(Fused2dPlan* P) {
  if (!P) return;
  if (P->dev) (void)hipFree(P->dev);
  if (P->img) (void)hipFree(P->img);
  delete P;
}

int fused2d_run(s3_ctx* ctx, Fused2dPlan* P, const float* W, uint64_t wversion, const float* x,
                float* y) {
  if (P->version != wversion) {
    int max_total = 0;
    for (const FL& f : P->host) max_total = std::max(max_total, f.n_nf * 9 * f.ck * 512);
    int gx = (max_total + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(fused2d_pack_all_kernel, dim3(gx, (unsigned)P->host.size()), dim3(256), 0, ctx->stream,
                       W, P->img, (const FL*)P->dev);
    S3_HIP(ctx, hipGetLastError());
    P->version = wversion;
  }
  static S3DeviceOnce attr_set;
  if (!attr_set.done(ctx->device)) {
    std::lock_guard<std::mutex> lk_attr_set(attr_set.m);
    S3_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fused2d_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.mark(ctx->device);
  }
  hipLaunchKernelGGL(fused2d_kernel, dim3(P->N), dim3(FNT), P->lds, ctx->stream, x, y,
                     (const char*)P->img, W, (const FL*)P->dev, (int)P->host.size(), P->H0, P->W0,
                     P->C0, P->slot0, P->border0, P->slot0_cells, P->y_per_sample);
  S3_HIP(ctx, hipGetLastError());
  return S3_OK;
}
