// store_probe2.hip - how many waves per CU, and how much un-drained store traffic, does the HBM write rate need?
// One workgroup (256 threads) walks `ntile` tiles of 128 rows x 256 B (the GEMM epilogue's pattern: 16 B per lane,
// a row = 16 lanes) per panel; `drain` = s_waitcnt vmcnt(0) after every tile (what a load issued after the stores costs).
// hipcc --offload-arch=gfx950 -O3 tools/store_probe2.hip -o /tmp/sp2 && /tmp/sp2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int DRAIN>
__global__ __launch_bounds__(256) void panel_store(char* out, long rowbytes, int panels, int ntile, int nout, int spin) {
  const int t = threadIdx.x;
  for (int pn = blockIdx.x; pn < panels; pn += gridDim.x) {
    for (int tm = 0; tm < ntile; ++tm) {
      for (int o = 0; o < nout; ++o) {                   // nout outputs (Y and Z) per tile
        char* base = out + (long)o * panels * 128 * rowbytes + (long)pn * 128 * rowbytes + (long)tm * 256;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = t + 256 * k, r = i >> 4, c = i & 15;
          uint4 v = make_uint4(pn, tm, k, t);
          *reinterpret_cast<uint4*>(base + (long)r * rowbytes + c * 16) = v;
        }
      }
      if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);    // stands in for the next tile's main loop
    }
  }
}
int main(int argc, char** argv) {
  const long R = argc > 1 ? atol(argv[1]) : 64000;
  const int ntile = 8, nout = 2;
  const long rowbytes = ntile * 256;
  const int panels = R / 128;
  char* d; hipMalloc(&d, (long)nout * panels * 128 * rowbytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int drain = 0; drain < 2; ++drain)
    for (int grid : {256, 512, 768, 1024, 2048})
      for (int spin : {0, 4}) {
        auto go = [&]() {
          if (drain) hipLaunchKernelGGL(panel_store<1>, dim3(grid), dim3(256), 0, 0, d, rowbytes, panels, ntile, nout, spin);
          else hipLaunchKernelGGL(panel_store<0>, dim3(grid), dim3(256), 0, 0, d, rowbytes, panels, ntile, nout, spin);
        };
        for (int w = 0; w < 3; ++w) go();
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) go();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)nout * panels * 128 * rowbytes;
        printf("drain=%d grid=%5d spin=%d : %8.1f us  %7.1f GB/s\n", drain, grid, spin, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e9);
      }
  return 0;
}
