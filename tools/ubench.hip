// Instruction-throughput microbenchmark for gfx950 integer / f64 paths used by big-integer field arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL32(name, ASM) \
extern "C" __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t s1, uint32_t s2){ \
  uint32_t a[8]; for(int i=0;i<8;i++) a[i]=threadIdx.x*7+i+s1; uint32_t b=s2+threadIdx.x, c=s1^threadIdx.x; \
  for(int k=0;k<ITERS;k++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); } } \
  uint32_t r=0; for(int i=0;i<8;i++) r^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=r; }

KERNEL32(k_add_u32,   "v_add_u32 %0, %0, %1")
KERNEL32(k_mul_lo,    "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi,    "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mad_u24,   "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_u24,   "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mulhi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_add3,      "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_addc,      "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_fma_f32,   "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_alignbit,  "v_alignbit_b32 %0, %0, %1, 30")

#define KERNEL64(name, ASM) \
extern "C" __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t s1, uint32_t s2){ \
  uint64_t a[8]; for(int i=0;i<8;i++) a[i]=threadIdx.x*7+i+s1; uint32_t b=s2+threadIdx.x, c=s1^threadIdx.x; uint64_t d=((uint64_t)b<<20)|c; \
  for(int k=0;k<ITERS;k++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "v"(d) : "vcc"); } } \
  uint64_t r=0; for(int i=0;i<8;i++) r^=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(uint32_t)(r^(r>>32)); }

KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_lshl_add_u64,"v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, 30, %0")
KERNEL64(k_mad_addc,    "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc")

extern "C" __global__ void __launch_bounds__(256) k_fma_f64(uint32_t* out, uint32_t s1, uint32_t s2){
  double a[8]; for(int i=0;i<8;i++) a[i]=threadIdx.x*7+i+s1; double b=1.0+1e-9*s2, c=1e-7*threadIdx.x;
  for(int k=0;k<ITERS;k++){ _Pragma("unroll") for(int i=0;i<8;i++){ asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); } }
  double r=0; for(int i=0;i<8;i++) r+=a[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=(uint32_t)r; }

typedef void (*kfn)(uint32_t*, uint32_t, uint32_t);
struct K { const char* name; kfn f; int per_iter; };

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  K ks[]={{"v_add_u32",k_add_u32,8},{"v_mul_lo_u32",k_mul_lo,8},{"v_mul_hi_u32",k_mul_hi,8},{"v_mad_u32_u24",k_mad_u24,8},{"v_mul_u32_u24",k_mul_u24,8},
          {"v_mul_hi_u32_u24",k_mulhi_u24,8},{"v_add3_u32",k_add3,8},{"v_addc_co_u32",k_addc,8},{"v_fma_f32",k_fma_f32,8},{"v_alignbit_b32",k_alignbit,8},
          {"v_mad_u64_u32",k_mad_u64_u32,8},{"v_lshl_add_u64",k_lshl_add_u64,8},{"v_lshrrev_b64",k_lshrrev_b64,8},{"mad_u64+addc pair",k_mad_addc,8},{"v_fma_f64",k_fma_f64,8}};
  int blocks=p.multiProcessorCount*8; uint32_t* out; CK(hipMalloc(&out, (size_t)blocks*256*4));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(auto& k: ks){
    for(int wpb: {8}){ (void)wpb;
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 1u, 2u); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for(int r=0;r<5;r++) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 1u, 2u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=5;
      double ops=(double)blocks*256*ITERS*k.per_iter; // lane-ops
      double per_cu_clk = ops/(ms*1e-3)/p.multiProcessorCount/2.4e9;
      printf("%-20s %8.3f ms  %8.2f Tlane-op/s  %6.1f lane-ops/clk/CU(@2.4GHz)  => %5.2f cyc/wave-instr/SIMD\n", k.name, ms, ops/(ms*1e-3)/1e12, per_cu_clk, 64.0*4/per_cu_clk);
    }
  }
  return 0;
}
