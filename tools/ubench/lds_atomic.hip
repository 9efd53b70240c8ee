// Micro-benchmark: cost of LDS atomics on gfx950 as a function of lanes per address (same-address serialisation),
// returning vs non-returning, 4 workgroups of 256 threads per CU (the occupancy of pass A of the k-mer count).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: ds_add_rtn_u32, 1: ds_add_u32 (no return), 2: ds_write_b32, 3: ds_read_b32
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, int lanes_per_addr, int random_addr) {
  __shared__ unsigned cnt[4][64];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  cnt[wave][lane] = 0;
  __syncthreads();
  unsigned acc = 0;
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
    unsigned a;
    if (random_addr) {
      x = x * 1664525u + 1013904223u;
      a = (x >> 20) & (unsigned)(random_addr - 1);  // random_addr distinct addresses (power of two)
    } else {
      a = lane / (unsigned)lanes_per_addr;
    }
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* p = (lds_u32*)&cnt[wave][a];
    if (MODE == 0) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 1) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 2) *(volatile lds_u32*)p = x;
    if (MODE == 3) acc += *(volatile lds_u32*)p;
  }
  if (acc == 0x12345678u) out[1] = acc;
}

template <int MODE>
int run(const char* name, unsigned* d, int lpa, int rnd) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 20000, blocks = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, lpa, rnd);
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, lpa, rnd);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double per_cu = (double)16 * iters;  // wave-instructions per CU
  printf("%-12s %s %2d: %.3f ms  -> %.1f cycles of CU time per wave-instruction (2.4 GHz)\n", name, rnd ? "random addrs" : "lanes/addr  ", rnd ? rnd : lpa, ms,
         ms * 1e-3 * 2.4e9 / per_cu);
  return 0;
}

int main() {
  unsigned* d; CHK(hipMalloc(&d, 64)); CHK(hipMemset(d, 0, 64));
  for (int lpa : {1, 2, 4, 8, 16, 32, 64}) { run<0>("add_rtn", d, lpa, 0); }
  for (int lpa : {1, 2, 4, 8, 64}) { run<1>("add_noret", d, lpa, 0); }
  for (int r : {64, 32, 16, 8}) { run<0>("add_rtn", d, 0, r); run<1>("add_noret", d, 0, r); }
  run<2>("write_b32", d, 1, 0); run<2>("write_b32", d, 0, 32);
  run<3>("read_b32", d, 1, 0); run<3>("read_b32", d, 0, 32);
  return 0;
}
