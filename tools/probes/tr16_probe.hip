// Probe of ds_read_b64_tr_b16 lane/element semantics on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o gpurun_out/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* __restrict__ addr_elems, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int a = addr_elems[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int sc = 0; sc < 2; ++sc) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, q = l & 15;
      // scenario 0: linear 8 B per lane; scenario 1: rows scattered (stride 100 elems, group base 1000*g)
      h_addr[l] = sc == 0 ? l * 4 : g * 1000 + (q >> 2) * 100 + (q & 3) * 4;
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, q = l & 15;
      for (int j = 0; j < 4; ++j) {
        const int src_lane = g * 16 + 4 * j + (q >> 2);
        const int expect = h_addr[src_lane] + (q & 3);
        if (h_out[l * 4 + j] != expect) ++bad;
      }
    }
    printf("scenario %d: mismatches vs model = %d\n", sc, bad);
    for (int l = 0; l < 64; l += 1)
      if (l < 20 || bad) printf("  lane %2d addr %5d -> %5d %5d %5d %5d\n", l, h_addr[l],
             h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
