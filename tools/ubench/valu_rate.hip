// Micro-benchmark: sustained issue rate of the integer VALU instructions pass A is made of (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
  unsigned c = out[0];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 1) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 2) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[i]));
        if (KIND == 3) asm volatile("v_cmpx_ne_u32_e32 %0, %1" : : "v"(a[i]), "v"(c) : "vcc");  // exec stays full: values differ
        if (KIND == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
        if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  if (s == 0x12345678u) out[1] = s;
}

template <int KIND>
int run(const char* name, unsigned* d, int blocks_per_cu) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 4000, blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10);
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr = (double)blocks * 4 * iters * 64;  // wave-instructions
  const double per_simd_per_s = winstr / (ms * 1e-3) / 1024;
  printf("%-14s %d waves/SIMD: %.3f ms  %.3e wave-instr/s/SIMD  -> %.2f cycles per wave-instr at 2.4 GHz\n", name, blocks_per_cu, ms,
         per_simd_per_s, 2.4e9 / per_simd_per_s);
  return 0;
}

int main() {
  unsigned* d; CHK(hipMalloc(&d, 64)); CHK(hipMemset(d, 0, 64));
  for (int w : {1, 2, 4}) {
    run<0>("v_xor_b32", d, w); run<1>("v_min_u32", d, w); run<2>("v_bfe_u32", d, w); run<3>("v_cmpx_ne_u32", d, w);
    run<4>("v_mad_u32_u24", d, w); run<5>("v_fma_f32", d, w);
  }
  return 0;
}
