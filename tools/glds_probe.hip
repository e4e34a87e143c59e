// glds_probe.hip — how fast can one workgroup layout stream a (N x K) bf16 operand panel HBM -> LDS?
// Compares the product GEMM's staging (global -> registers -> ds_write, ONE stage in flight per workgroup) with an
// LDS-DMA ring (global_load_lds_dwordx4, S stages in flight, no staging registers) at 1-3 workgroups per CU.
// Standalone:  hipcc --offload-arch=gfx950 -O3 tools/glds_probe.hip -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define TILE_ROWS 128
#define BK 64                       // bf16 elements per row per stage = 128 B
#define STAGE_BYTES (TILE_ROWS * BK * 2)

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// register staging, one stage in flight (the product kernel's scheme)
__global__ __launch_bounds__(256) void stream_reg(const uint16_t* __restrict__ A, long lda, int K, float* out) {
  __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES];
  const int t = threadIdx.x;
  const long row0 = (long)blockIdx.x * TILE_ROWS;
  uint4 r[4];
  float acc = 0.f;
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int v = t + 256 * i, row = v >> 3, c = v & 7; r[i] = *reinterpret_cast<const uint4*>(A + (row0 + row) * lda + k0 + c * 8); }
  };
  load(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int v = t + 256 * i; *reinterpret_cast<uint4*>(lds + v * 16) = r[i]; }
    __syncthreads();
    if (k0 + BK < K) load(k0 + BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint4 u = *reinterpret_cast<const uint4*>(lds + ((t * 4 + i) & 1023) * 16); acc += __uint_as_float(u.x & 0xffff0000u); }
    __syncthreads();
  }
  if (acc == 123.456f) out[0] = acc;
}

// LDS-DMA ring with S stages: issue stage it+S-1, wait for stage it (counted vmcnt), barrier, consume
template <int S>
__global__ __launch_bounds__(256) void stream_dma(const uint16_t* __restrict__ A, long lda, int K, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long row0 = (long)blockIdx.x * TILE_ROWS;
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
  float acc = 0.f;
  const int niter = K / BK;
  // one stage = 16 wave-instructions of 1 KB (8 rows x 128 B); wave w issues instructions w, w+4, w+8, w+12
  auto issue = [&](int it) {
    const int buf = it % S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ins = w + 4 * j, row = ins * 8 + (lane >> 3), c = lane & 7;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_BYTES + ins * 1024);
      glds16(A + (row0 + row) * lda + (long)it * BK + c * 8, dst);
    }
  };
  for (int s = 0; s < S - 1 && s < niter; ++s) issue(s);
  for (int it = 0; it < niter; ++it) {
    if (it + S - 1 < niter) {
      issue(it + S - 1);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (S - 1)) : "memory");     // stage `it` has landed, S-1 still in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* b = lds + (it % S) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const uint4 u = *reinterpret_cast<const uint4*>(b + ((t * 4 + i) & 1023) * 16); acc += __uint_as_float(u.x & 0xffff0000u); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // buffer (it % S) is free for stage it + S
  }
  if (acc == 123.456f) out[0] = acc;
}

#include <stdlib.h>
int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 64000, K = 1024;   // 64000 rows = 131 MB (MALL resident); 512000 = 1 GB (HBM)
  uint16_t* A; float* out;
  hipMalloc(&A, N * K * 2 + 4096); hipMalloc(&out, 64);
  hipMemset(A, 0x3c, N * K * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = N / TILE_ROWS;
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.1f us  %7.0f GB/s\n", name, ms * 1e3 / 20, N * K * 2 / (ms * 1e-3 / 20) / 1e9);
  };
  timeit("register staging, 1 stage (16 KB LDS)", [&] { hipLaunchKernelGGL(stream_reg, dim3(blocks), dim3(256), 0, 0, A, K, (int)K, out); });
  hipFuncSetAttribute((const void*)stream_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)stream_dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)stream_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)stream_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  timeit("LDS-DMA ring, 2 stages (32 KB)", [&] { hipLaunchKernelGGL(stream_dma<2>, dim3(blocks), dim3(256), 2 * STAGE_BYTES, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 3 stages (48 KB: 3 WG/CU)", [&] { hipLaunchKernelGGL(stream_dma<3>, dim3(blocks), dim3(256), 3 * STAGE_BYTES, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 4 stages (64 KB: 2 WG/CU)", [&] { hipLaunchKernelGGL(stream_dma<4>, dim3(blocks), dim3(256), 4 * STAGE_BYTES, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 8 stages (128 KB: 1 WG/CU)", [&] { hipLaunchKernelGGL(stream_dma<8>, dim3(blocks), dim3(256), 8 * STAGE_BYTES, 0, A, K, (int)K, out); });
  // same rings with the LDS allocation padded to force 3 / 2 / 1 workgroups per CU (the GEMM's occupancy regimes)
  timeit("LDS-DMA ring, 2 stages, 3 WG/CU (53 KB alloc)", [&] { hipLaunchKernelGGL(stream_dma<2>, dim3(blocks), dim3(256), 53 * 1024, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 2 stages, 2 WG/CU (80 KB alloc)", [&] { hipLaunchKernelGGL(stream_dma<2>, dim3(blocks), dim3(256), 80 * 1024, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 3 stages, 2 WG/CU (80 KB alloc)", [&] { hipLaunchKernelGGL(stream_dma<3>, dim3(blocks), dim3(256), 80 * 1024, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 2 stages, 1 WG/CU (160 KB alloc)", [&] { hipLaunchKernelGGL(stream_dma<2>, dim3(blocks), dim3(256), 160 * 1024, 0, A, K, (int)K, out); });
  timeit("LDS-DMA ring, 4 stages, 1 WG/CU (160 KB alloc)", [&] { hipLaunchKernelGGL(stream_dma<4>, dim3(blocks), dim3(256), 160 * 1024, 0, A, K, (int)K, out); });
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
