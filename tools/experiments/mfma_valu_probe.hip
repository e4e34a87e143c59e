// mfma_valu_probe.hip - how much VALU / LDS work hides behind the MFMAs of ONE wave per SIMD on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/experiments/mfma_valu_probe.hip && /tmp/probe
// One 256-thread workgroup per CU (launch_bounds(256, 1)); each wave runs NM MFMAs (32x32x16 bf16) with, behind every MFMA,
// K independent VALU instructions (v_fma_f32 | v_exp_f32 | v_pk_fma_f32) and optionally one ds_read_b128; s_memtime stamps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int K, int KIND, int ACCV, int LDS, int DEP>
__global__ __launch_bounds__(256, 1) void probe(long long* out, float* sink, int nm) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  uint4 u = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  bf16x8 a = __builtin_bit_cast(bf16x8, u), b = a;
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = 1.0f + lane * 1e-3f + i;
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = 1.f;
  __syncthreads();
  uint4 fr = make_uint4(0, 0, 0, 0);
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < nm; it += 4) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int ai = DEP ? (m & 1) : m;
      if (ACCV) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[ai]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(a), "v"(b));
      if (LDS) fr = *reinterpret_cast<const uint4*>(lds + ((lane * 16 + m * 1024 + it * 64) & 32767 & ~15));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
        else if (KIND == 1) v[k] = __builtin_amdgcn_exp2f(v[k]);
        else {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 x = {v[k], v[(k + 6) % 12]};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(x));
          v[k] = x.x;
        }
      }
      if (LDS) asm volatile("" : "+v"(fr.x), "+v"(fr.y), "+v"(fr.z), "+v"(fr.w));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) {
    f32x16 t = acc[i];
    if (ACCV) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(t)); else asm volatile("s_nop 7\n\ts_nop 7" : "+a"(t));
    for (int e = 0; e < 16; ++e) s += t[e];
  }
  s += __uint_as_float(fr.x);
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int KIND, int ACCV, int LDS, int DEP>
void run(const char* name) {
  const int nwg = 256, nm = 1024;
  long long* d;
  float* sink;
  hipMalloc(&d, nwg * 4 * sizeof(long long));
  hipMalloc(&sink, 4);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<K, KIND, ACCV, LDS, DEP>), dim3(nwg), dim3(256), 0, 0, d, sink, nm);
  hipDeviceSynchronize();
  std::vector<long long> h(nwg * 4);
  hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto x : h) mean += x;
  mean /= h.size();
  printf("%-52s %7.1f cycles per MFMA\n", name, mean / nm);
  hipFree(d);
  hipFree(sink);
}

int main() {
  run<0, 0, 0, 0, 0>("MFMA only, 4 accumulators (AGPR)");
  run<0, 0, 1, 0, 0>("MFMA only, 4 accumulators (VGPR)");
  run<0, 0, 1, 0, 1>("MFMA only, 2 accumulators alternating (VGPR)");
  run<2, 0, 0, 0, 0>("+ 2 v_fma per MFMA (AGPR acc)");
  run<4, 0, 0, 0, 0>("+ 4 v_fma per MFMA (AGPR acc)");
  run<6, 0, 0, 0, 0>("+ 6 v_fma per MFMA (AGPR acc)");
  run<8, 0, 0, 0, 0>("+ 8 v_fma per MFMA (AGPR acc)");
  run<12, 0, 0, 0, 0>("+ 12 v_fma per MFMA (AGPR acc)");
  run<8, 0, 1, 0, 0>("+ 8 v_fma per MFMA (VGPR acc)");
  run<2, 1, 0, 0, 0>("+ 2 v_exp per MFMA (AGPR acc)");
  run<4, 1, 0, 0, 0>("+ 4 v_exp per MFMA (AGPR acc)");
  run<8, 1, 0, 0, 0>("+ 8 v_exp per MFMA (AGPR acc)");
  run<4, 2, 0, 0, 0>("+ 4 v_pk_fma per MFMA (AGPR acc)");
  run<8, 2, 0, 0, 0>("+ 8 v_pk_fma per MFMA (AGPR acc)");
  run<0, 0, 0, 1, 0>("+ 1 ds_read_b128 per MFMA");
  run<6, 0, 0, 1, 0>("+ 1 ds_read_b128 + 6 v_fma per MFMA");
  return 0;
}
