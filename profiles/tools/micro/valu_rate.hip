// VALU issue rate on one SIMD by number of resident waves: cycles per wave64 instruction for
// independent integer ops (v_add_u32 / v_xor_b32 / v_and_or_b32) and for v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void rate_kernel(unsigned *out, long long *cyc, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a5) : "v"(a7));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(a7));
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a7));
      } else if (KIND == 1) {
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(a7), "v"(a6));
        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(a7), "v"(a6));
        asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a2));
        asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a3) : "v"(a7), "v"(a6));
        asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(a7), "v"(a6));
        asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a5) : "v"(a7));
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a7));
        asm volatile("v_min_i32 %0, %0, %1" : "+v"(a1) : "v"(a7));
      } else {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f2) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f3) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f4) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f5) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f7), "v"(f6));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f1) : "v"(f7), "v"(f6));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + (unsigned)(f0 + f1 + f2 + f3 + f4 + f5);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name) {
  const int iters = 4000;
  unsigned *out;
  long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&cyc, 256 * 8);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU: its waves spread over the four SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<KIND><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e0);
    rate_kernel<KIND><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    const double ninst = (double)iters * 64;
    // s_memtime-style counter runs at 100 MHz: report wall time per instruction per SIMD instead
    printf("%-10s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz), counter %lld\n", name,
           waves_per_simd, ms, ms * 1e6 / (ninst * waves_per_simd), ms * 1e6 / (ninst * waves_per_simd) * 2.4, h[0]);
  }
}

int main() {
  run<0>("add/xor");
  run<1>("vop3 mix");
  run<2>("fma_f32");
  return 0;
}
