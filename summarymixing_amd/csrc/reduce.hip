// reduce.hip — one launch for many small fixed-order reductions.
// Split-K weight-gradient slabs, bias-gradient partials and LayerNorm dgamma/dbeta partial rows all end in
//   dst[i, j] += alpha * sum_s src[s][i, j]
// with tiny outputs; done one launch each they are ~300 latency-bound launches per training step (6 % of the step at
// 64000 frames, 13 % at 16000).  The producers can leave their partials in place (smx_linear_wgrad_partial,
// smx_layernorm_bwd with NULL dgamma/dbeta) and the caller flushes a table of jobs in ONE launch per encoder layer.
#include "smx_common.h"

namespace smx {

// P lanes share one element group (sources s = part, part + P, ..; 4 loads in flight each) and are folded by shuffles
// in a fixed order: bit-reproducible, and no thread walks a long chain of dependent L2 round trips.
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const smx_reduce_job* __restrict__ jobs,
                                                          const int32_t* __restrict__ starts, int njobs) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {                                      // last job whose first block is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const smx_reduce_job j = jobs[lo];
  const long g = (long)((int)blockIdx.x - starts[lo]) * 256 + threadIdx.x;
  // P lanes per element group: 8 for many sources, 4 for a handful (the grouped wgrad writes ~11 slabs per weight: one
  // lane walking them is three dependent rounds of four loads, four lanes have all their loads in flight at once)
  const int pshift = j.nsrc >= 16 ? 3 : (j.nsrc >= 6 ? 2 : 0), P = 1 << pshift;
  const long sld = j.src_ld > 0 ? j.src_ld : j.cols;       // source row stride (elements)
  const long i = g >> pshift;
  const int part = (int)(g & (P - 1));
  if (j.vec) {
    const int cv = j.cols >> 2;
    const long total = (long)j.rows * cv;
    const bool ok = i < total;
    const long ic = ok ? i : 0;
    const int r = (int)(ic / cv), c4 = (int)(ic % cv);
    const float* sp = j.src + (long)r * sld + c4 * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (ok) {
      int s = part;
      for (; s + 3 * P < j.nsrc; s += 4 * P) {
        const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)s * j.src_stride);
        const float4 v1 = *reinterpret_cast<const float4*>(sp + (long)(s + P) * j.src_stride);
        const float4 v2 = *reinterpret_cast<const float4*>(sp + (long)(s + 2 * P) * j.src_stride);
        const float4 v3 = *reinterpret_cast<const float4*>(sp + (long)(s + 3 * P) * j.src_stride);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; s < j.nsrc; s += P) {
        const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)s * j.src_stride);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      }
    }
    float sx = (a0.x + a1.x) + (a2.x + a3.x), sy = (a0.y + a1.y) + (a2.y + a3.y);
    float sz = (a0.z + a1.z) + (a2.z + a3.z), sw = (a0.w + a1.w) + (a2.w + a3.w);
    if (P > 1) {
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        if (off < P) {
          sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64);
          sz += __shfl_xor(sz, off, 64); sw += __shfl_xor(sw, off, 64);
        }
      }
    }
    if (ok && part == 0) {
      float* d = j.dst + (long)r * j.ldd + c4 * 4;
      d[0] += j.alpha * sx; d[1] += j.alpha * sy; d[2] += j.alpha * sz; d[3] += j.alpha * sw;
    }
  } else {
    const long total = (long)j.rows * j.cols;
    const bool ok = i < total;
    const long ic = ok ? i : 0;
    const int r = (int)(ic / j.cols), c = (int)(ic % j.cols);
    const float* sp = j.src + (long)r * sld + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (ok) {
      int s = part;
      for (; s + 3 * P < j.nsrc; s += 4 * P) {
        a0 += sp[(long)s * j.src_stride]; a1 += sp[(long)(s + P) * j.src_stride];
        a2 += sp[(long)(s + 2 * P) * j.src_stride]; a3 += sp[(long)(s + 3 * P) * j.src_stride];
      }
      for (; s < j.nsrc; s += P) a0 += sp[(long)s * j.src_stride];
    }
    float a = (a0 + a1) + (a2 + a3);
    if (P > 1) {
#pragma unroll
      for (int off = 1; off < 8; off <<= 1)
        if (off < P) a += __shfl_xor(a, off, 64);
    }
    if (ok && part == 0) j.dst[(long)r * j.ldd + c] += j.alpha * a;
  }
}

}  // namespace smx

using namespace smx;

extern "C" int smx_reduce_job_blocks(const smx_reduce_job* job_host) {
  if (!job_host || job_host->rows <= 0 || job_host->cols <= 0 || job_host->nsrc <= 0) return 0;
  const long groups = job_host->vec ? (long)job_host->rows * (job_host->cols / 4) : (long)job_host->rows * job_host->cols;
  const long threads = groups * (job_host->nsrc >= 16 ? 8 : (job_host->nsrc >= 6 ? 4 : 1));
  return (int)((threads + 255) / 256);
}

extern "C" int smx_reduce_jobs(const smx_reduce_job* jobs_dev, const int32_t* block_starts_dev, int njobs, int total_blocks,
                               void* stream) {
  SMX_REQUIRE(njobs >= 0 && total_blocks >= 0, "smx_reduce_jobs: bad sizes");
  if (njobs == 0 || total_blocks == 0) return SMX_OK;
  SMX_REQUIRE(jobs_dev && block_starts_dev, "smx_reduce_jobs: null pointer");
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     jobs_dev, block_starts_dev, njobs);
  return check_launch("smx_reduce_jobs");
}
