// tr_probe.hip — semantics probe for ds_read_b64_tr_b16 (hipcc tools/tr_probe.hip -o /tmp/trp && /tmp/trp)
// LDS holds a [k][ROWS] uint16 image with value = k*256 + row.  Lane l supplies the address of 4 contiguous elements
// (k = kbase + (l&15)/4, rows r0 + ((l&15)%4)*4 ..+3); we print what each lane gets back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short short4_t __attribute__((ext_vector_type(4)));
constexpr int ROWS = 64, STRIDE = ROWS + 32;   // padded row stride (elements)
__global__ void k(uint2* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[16 * STRIDE];
  for (int i = threadIdx.x; i < 16 * STRIDE; i += 64) { int kk = i / STRIDE, r = i % STRIDE; lds[i] = (uint16_t)(kk * 256 + r); }
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, li = l & 15;
  const int rowbase = (g & 1) * 16, kbase = (g >> 1) * 8;
  auto p = (__attribute__((address_space(3))) short4_t*)(lds + (kbase + li / 4) * STRIDE + rowbase + (li % 4) * 4);
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  out[l] = __builtin_bit_cast(uint2, v);
}
int main() {
  uint2* d; hipMalloc(&d, 64 * sizeof(uint2));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint2 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    uint16_t e[4] = {(uint16_t)(h[l].x & 0xffff), (uint16_t)(h[l].x >> 16), (uint16_t)(h[l].y & 0xffff), (uint16_t)(h[l].y >> 16)};
    int g = l >> 4, li = l & 15, row = (g & 1) * 16 + li, kb = (g >> 1) * 8;
    for (int j = 0; j < 4; ++j) if (e[j] != (kb + j) * 256 + row) bad++;
    if (l < 4 || l == 17 || l == 35 || l == 63) printf("lane %2d: (k,row) = (%d,%d) (%d,%d) (%d,%d) (%d,%d)   expect row %d k %d..%d\n", l, e[0] >> 8, e[0] & 255, e[1] >> 8, e[1] & 255, e[2] >> 8, e[2] & 255, e[3] >> 8, e[3] & 255, row, kb, kb + 3);
  }
  printf("mismatches vs hypothesis (lane gets 4 consecutive k of ITS row): %d\n", bad);
  return 0;
}
