// Calibration kernel for the rocprofv3 FETCH_SIZE counter on gfx950 (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").  Reproduces the access pattern of
// k_msm_accum: every thread gathers 64-byte table entries (4 x 16 B per lane) at random 64-B-aligned
// offsets from a table far larger than the 256 MiB Infinity Cache, plus a coalesced 4-B index stream.
// Known bytes: reads = n*64 (gather) + n*4 (indices); writes = n/64 * 4.
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

extern "C" __global__ void __launch_bounds__(256) calib_gather64(const uint4* table, const uint32_t* idx, uint32_t per_thread, uint32_t* out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t k = 0; k < per_thread; k++) {
    uint32_t e = idx[(size_t)t * per_thread + k];
    const uint4* p = table + (size_t)e * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  out[t] = acc;
}
extern "C" __global__ void __launch_bounds__(256) calib_stream16(const uint4* src, size_t n16, uint32_t* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (size_t i = t; i < n16; i += stride) { uint4 v = src[i]; acc ^= v.x ^ v.w; }
  out[t] = acc;
}
extern "C" __global__ void __launch_bounds__(256) calib_fill(uint4* dst, size_t n16) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = t; i < n16; i += stride) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
  const size_t entries = (size_t)1 << 24;            // 16 Mi x 64 B = 1 GiB table
  const uint32_t threads = 1 << 18, per_thread = 64; // 16 Mi gathers = 1 GiB gathered + 64 MiB indices
  uint4* table; uint32_t *idx, *out;
  CK(hipMalloc(&table, entries * 64)); CK(hipMalloc(&idx, (size_t)threads * per_thread * 4)); CK(hipMalloc(&out, (size_t)threads * 4));
  hipLaunchKernelGGL(calib_fill, dim3(2048), dim3(256), 0, 0, table, entries * 4);
  uint32_t* h = (uint32_t*)malloc((size_t)threads * per_thread * 4);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < (size_t)threads * per_thread; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 20) & (entries - 1); }
  CK(hipMemcpy(idx, h, (size_t)threads * per_thread * 4, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  for (int r = 0; r < 3; r++) {
    hipLaunchKernelGGL(calib_gather64, dim3(threads / 256), dim3(256), 0, 0, table, idx, per_thread, out);
    hipLaunchKernelGGL(calib_stream16, dim3(2048), dim3(256), 0, 0, table, entries * 4, out);
  }
  CK(hipDeviceSynchronize());
  printf("calib_gather64: known read bytes per launch = %zu (gather) + %zu (indices); calib_stream16: %zu\n",
         (size_t)threads * per_thread * 64, (size_t)threads * per_thread * 4, entries * 64);
  return 0;
}
