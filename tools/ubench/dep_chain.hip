// Micro-benchmark: latency of a DEPENDENT float32 add chain on gfx950 (one wave per SIMD / per CU), bare and with
// LDS reads in between -- the floor of the serial (bit-exact) EM fold.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 1e-3f * i;
  __syncthreads();
  float acc = out[0], c = out[1];
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(c));
    } else if (MODE == 1) {  // two independent chains interleaved
      float acc2 = acc * 2;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(c));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc2) : "v"(c));
      }
      acc += acc2;
    } else if (MODE == 2) {  // chain fed from LDS, 16 quads per iteration
      const f4* src = reinterpret_cast<const f4*>(&buf[(threadIdx.x & 3) * 64]);
      f4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = src[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc += v[i].x; acc += v[i].y; acc += v[i].z; acc += v[i].w; }
    } else if (MODE == 3) {  // v_add with an s_nop between dependent adds
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(acc) : "v"(c));
    } else if (MODE == 4) {  // DPP-free alternative: v_add_f32 with both operands the accumulator chain through v_fma
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %1, 1.0, %0" : "+v"(acc) : "v"(c));
    } else if (MODE == 5) {  // f64 add chain
      double a = acc, cc = c;
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(cc));
      acc = (float)a;
    }
  }
  if (acc == 12345.678f) out[2] = acc;
}

template <int MODE>
int run(const char* name, float* d, int blocks) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10);
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s %4d waves: %.3f ms -> %.2f ns = %.1f cycles (2.4 GHz) per chained add\n", name, blocks, ms, ms * 1e6 / (iters * 64.0),
         ms * 1e-3 * 2.4e9 / (iters * 64.0));
  return 0;
}

int main() {
  float* d; CHK(hipMalloc(&d, 64)); CHK(hipMemset(d, 0, 64));
  for (int b : {256, 1024}) {
    run<0>("v_add_f32 dependent", d, b); run<1>("two chains interleaved (per add)", d, b); run<2>("chain fed by ds_read_b128", d, b);
    run<3>("dependent + s_nop", d, b); run<4>("v_fma_f32 dependent", d, b); run<5>("v_add_f64 dependent", d, b);
  }
  return 0;
}
