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
// Round 6 (the recipe batch: 14 launches of 33 us for ~100 MB each): a workgroup found its job by a five-deep binary search of
// dependent loads; now ONE vector load + ballot per 64 jobs.  A thread owns kRU groups (rolled: the first version kept all of them
// in flight at once - 196 registers, two waves per SIMD, 59 us instead of 33; occupancy is what hides these latencies) and
// requests its destination value ahead of the sources.
constexpr int kRU = 1;
__host__ __device__ inline int reduce_pshift(int nsrc) { return nsrc >= 256 ? 6 : (nsrc >= 64 ? 5 : (nsrc >= 16 ? 3 : (nsrc >= 6 ? 2 : 0))); }
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const smx_reduce_job* __restrict__ jobs,
                                                          const int32_t* __restrict__ starts, int njobs) {
  // the last job whose first block is <= blockIdx.x: starts[] is ascending, so that is (number of such jobs) - 1
  int lo = -1;
  {
    const int lane = threadIdx.x & 63;
    for (int k0 = 0; k0 < njobs; k0 += 64) {
      const int k = k0 + lane;
      const bool le = k < njobs && starts[k] <= (int)blockIdx.x;
      const int cnt = __popcll(__ballot(le));
      lo += cnt;
      if (cnt < 64) break;
    }
  }
  const smx_reduce_job j = jobs[lo];
  const long gbase = (long)((int)blockIdx.x - starts[lo]) * (256 * kRU) + threadIdx.x;
  // P lanes per element group: 8 for many sources, 4 for a handful (the grouped wgrad writes ~11 slabs per weight: one
  // lane walking them is three dependent rounds of four loads, four lanes have all their loads in flight at once)
  // (round 6: 64 / 32 lanes for hundreds of sources - the LayerNorm dgamma / dbeta partial rows of a small batch, one per 4 frames:
  //  938 sources at 3750 frames walked by 8 lanes were a 29-deep chain of dependent loads, the critical path of the whole launch)
  const int pshift = reduce_pshift(j.nsrc), P = 1 << pshift;
  const long sld = j.src_ld > 0 ? j.src_ld : j.cols;       // source row stride (elements)
  const int part = (int)(gbase & (P - 1));                 // (the same for a thread's kRU groups: 256 is a multiple of P)
  if (j.vec) {
    const int cv = j.cols >> 2;
    const long total = (long)j.rows * cv;
#pragma unroll 1
    for (int u = 0; u < kRU; ++u) {
      const long i = (gbase + u * 256) >> pshift;
      const bool ok = i < total;
      const long ic = ok ? i : 0;
      const int r = (int)(ic / cv), c4 = (int)(ic % cv);
      const float* sp = j.src + (long)r * sld + c4 * 4;
      float* d = j.dst + (long)r * j.ldd + c4 * 4;
      float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && part == 0) dv = *reinterpret_cast<const float4*>(d);   // (only this lane touches the group: requested ahead)
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      if (ok) {
        int s_ = part;
        for (; s_ + 3 * P < j.nsrc; s_ += 4 * P) {
          const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)s_ * j.src_stride);
          const float4 v1 = *reinterpret_cast<const float4*>(sp + (long)(s_ + P) * j.src_stride);
          const float4 v2 = *reinterpret_cast<const float4*>(sp + (long)(s_ + 2 * P) * j.src_stride);
          const float4 v3 = *reinterpret_cast<const float4*>(sp + (long)(s_ + 3 * P) * j.src_stride);
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
          a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
          a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
          a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        if (s_ + P < j.nsrc && s_ + 2 * P >= j.nsrc) {       // exactly two sources left (two split-K slabs): both in flight
          const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)s_ * j.src_stride);
          const float4 v1 = *reinterpret_cast<const float4*>(sp + (long)(s_ + P) * j.src_stride);
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
          a0.x += v1.x; a0.y += v1.y; a0.z += v1.z; a0.w += v1.w;
          s_ += 2 * P;
        }
        for (; s_ < j.nsrc; s_ += P) {
          const float4 v0 = *reinterpret_cast<const float4*>(sp + (long)s_ * j.src_stride);
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
      }
      float sx = (a0.x + a1.x) + (a2.x + a3.x), sy = (a0.y + a1.y) + (a2.y + a3.y);
      float sz = (a0.z + a1.z) + (a2.z + a3.z), sw = (a0.w + a1.w) + (a2.w + a3.w);
      if (P > 1) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          if (off < P) {
            sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64);
            sz += __shfl_xor(sz, off, 64); sw += __shfl_xor(sw, off, 64);
          }
        }
      }
      if (ok && part == 0)
        *reinterpret_cast<float4*>(d) = make_float4(dv.x + j.alpha * sx, dv.y + j.alpha * sy, dv.z + j.alpha * sz, dv.w + j.alpha * sw);
    }
  } else {
    const long total = (long)j.rows * j.cols;
#pragma unroll 1
    for (int u = 0; u < kRU; ++u) {
      const long i = (gbase + u * 256) >> pshift;
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
        for (int off = 1; off < 64; off <<= 1)
          if (off < P) a += __shfl_xor(a, off, 64);
      }
      if (ok && part == 0) j.dst[(long)r * j.ldd + c] += j.alpha * a;
    }
  }
}

}  // namespace smx

using namespace smx;

extern "C" int smx_reduce_job_blocks(const smx_reduce_job* job_host) {
  if (!job_host || job_host->rows <= 0 || job_host->cols <= 0 || job_host->nsrc <= 0) return 0;
  const long groups = job_host->vec ? (long)job_host->rows * (job_host->cols / 4) : (long)job_host->rows * job_host->cols;
  const long threads = groups << reduce_pshift(job_host->nsrc);
  return (int)((threads + 256 * kRU - 1) / (256 * kRU));
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
