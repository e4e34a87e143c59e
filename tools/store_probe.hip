// store_probe.hip — pure store-pattern bandwidth probe (standalone; hipcc tools/store_probe.hip -o /tmp/sp && /tmp/sp)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// Each block writes a tile of `rows` rows x `seg` bytes; tiles are laid out over a (R x rowbytes) matrix.
__global__ __launch_bounds__(256) void tile_store(char* out, long rowbytes, int rows, int seg, int tiles_m, int swz) {
  int bid = blockIdx.x;
  if (swz) { int nt = gridDim.x; int q = nt >> 3, r = nt & 7, x = bid & 7, i = bid >> 3; bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
  int tn = bid / tiles_m, tm = bid % tiles_m;
  int cpr = seg / 16;
  uint4 v = make_uint4(bid, 1, 2, 3);
  for (int i = threadIdx.x; i < rows * cpr; i += 256) {
    int r = i / cpr, c = i % cpr;
    *reinterpret_cast<uint4*>(out + ((long)tn * rows + r) * rowbytes + (long)tm * seg + c * 16) = v;
  }
}
int main(int argc, char** argv) {
  // default 131 MB (= Y and Z of the FFN up-projection at 32000 frames: fits the 256 MB MALL); pass a larger row count
  // (e.g. 256000 -> 1 GB) for the HBM-resident write rate
  const long R = argc > 1 ? atol(argv[1]) : 32000, rowbytes = 4096;
  char* d; hipMalloc(&d, R * rowbytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { int rows, seg, swz; } cfg[] = {{128, 256, 1}, {128, 256, 0}, {128, 512, 1}, {64, 1024, 1}, {32, 4096, 1}, {8, 4096, 0}, {128, 4096, 0}};
  for (auto c : cfg) {
    int tiles_m = rowbytes / c.seg, tiles_n = (R + c.rows - 1) / c.rows;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(tile_store, dim3(tiles_m * tiles_n), dim3(256), 0, 0, d, rowbytes, c.rows, c.seg, tiles_m, c.swz);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(tile_store, dim3(tiles_m * tiles_n), dim3(256), 0, 0, d, rowbytes, c.rows, c.seg, tiles_m, c.swz);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("R=%ld tile %4d rows x %5d B  swz=%d  blocks=%6d : %8.1f us  %7.1f GB/s\n", R, c.rows, c.seg, c.swz, tiles_m * tiles_n, ms * 1e3 / 20, R * rowbytes / (ms * 1e-3 / 20) / 1e9);
  }
  return 0;
}
