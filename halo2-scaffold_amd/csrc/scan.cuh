// Exclusive scans of 32-bit counters on the device, shared by the MSM's bucket partition and the lookup argument's
// counting sort (own kernels: the library scans took 15-70 us at these sizes, mostly launch latency).
#pragma once
// A/B only (-DH2MI_AB, H2MI_AB_PRIO): issue priority of the partition / scan wavefronts over the accumulation wavefronts they share SIMDs with
#ifdef H2MI_AB
static __device__ uint32_t g_ab_prio = 0;
#define H2_AB_PRIO() do { if (g_ab_prio) __builtin_amdgcn_s_setprio(3); } while (0)
#else
#define H2_AB_PRIO() do {} while (0)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace h2 {

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  return x;
}

// Exclusive scan of m counters (m a multiple of 4), out[m] = total.  One workgroup per segment of SEG
// counters, each thread keeping a contiguous slice in registers (16-byte loads); with more than one segment
// k_scan_segsum runs first and every workgroup adds the sums of the segments before its own.  blockIdx.y
// picks the array.  (The library's look-back scan takes 15-70 us at these sizes, mostly launch latency.)
constexpr uint32_t SCAN_SEG_TASKS = 65536;  // the task counts (<= 32769): one launch, one workgroup per array
constexpr uint32_t SCAN_SEG_BINS = 8192;    // the [bin][tile] matrix: many short segments spread over the CUs
__device__ __forceinline__ uint32_t block_sum_1024(uint32_t v, uint32_t* wsum /* 16 */) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t t = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; w++) t += wsum[w];
  return t;
}
template <uint32_t SEG>
__device__ __forceinline__ void scan_segsum_body(const uint32_t* in, uint32_t m, uint32_t* segsum) {
  H2_AB_PRIO();
  __shared__ uint32_t wsum[16];
  const uint32_t seg0 = blockIdx.x * SEG, len = min(SEG, m - seg0);
  uint32_t sum = 0;
  for (uint32_t j = threadIdx.x * 4; j < len; j += 4096) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + seg0 + j);
    sum += v.x + v.y + v.z + v.w;
  }
  const uint32_t t = block_sum_1024(sum, wsum);
  if (threadIdx.x == 0) segsum[blockIdx.x] = t;
}
template <uint32_t SEG>
__global__ void __launch_bounds__(1024) k_scan_segsum(const uint32_t* in, uint32_t m, uint32_t* segsum) {
  scan_segsum_body<SEG>(in, m, segsum);
}
// one array: segment blockIdx.x of `in_` -> `out_`
template <uint32_t SEG>
__device__ __forceinline__ void scan_seg_body(const uint32_t* in_, uint32_t* out_, uint32_t m, const uint32_t* segsum) {
  H2_AB_PRIO();
  __shared__ uint32_t wsum[16], wsum2[16];
  constexpr uint32_t NV = SEG / 4096;  // 16-byte vectors per thread
  const uint32_t seg = blockIdx.x, seg0 = seg * SEG, len = min(SEG, m - seg0);
  const uint32_t* in = in_ + seg0;
  uint32_t* out = out_ + seg0;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t before = 0;  // sum of the earlier segments
  if (seg) {
    uint32_t part = 0;
    for (uint32_t j = tid; j < seg; j += 1024) part += segsum[j];
    before = block_sum_1024(part, wsum2);
  }
  const uint32_t per = (((len + 1023) / 1024) + 3) & ~3u;  // <= 4 * NV
  const uint32_t lo = min(tid * per, len), hi = min(lo + per, len);
  uint4 r[NV];
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < NV; k++) {
    const uint32_t j = lo + 4 * k;
    r[k] = make_uint4(0, 0, 0, 0);
    if (j < hi) {
      r[k] = *reinterpret_cast<const uint4*>(in + j);
      sum += r[k].x + r[k].y + r[k].z + r[k].w;
    }
  }
  const uint32_t inc = wave_incl_scan(sum);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t a = before + inc - sum;
  for (uint32_t w = 0; w < wave; w++) a += wsum[w];
#pragma unroll
  for (uint32_t k = 0; k < NV; k++) {
    const uint32_t j = lo + 4 * k;
    if (j < hi) {
      uint4 o;
      o.x = a; a += r[k].x;
      o.y = a; a += r[k].y;
      o.z = a; a += r[k].z;
      o.w = a; a += r[k].w;
      *reinterpret_cast<uint4*>(out + j) = o;
    }
  }
  if (tid == 0 && seg0 + len == m) {
    uint32_t t = before;
    for (uint32_t w = 0; w < 16; w++) t += wsum[w];
    out[len] = t;
  }
}
template <uint32_t SEG>
__global__ void __launch_bounds__(1024) k_scan_seg(const uint32_t* in0, uint32_t* out0, const uint32_t* in1, uint32_t* out1, uint32_t m,
                                                   const uint32_t* segsum) {
  scan_seg_body<SEG>(blockIdx.y ? in1 : in0, blockIdx.y ? out1 : out0, m, segsum);
}


}  // namespace h2
