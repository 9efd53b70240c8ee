// Micro-benchmark: HBM write rate of the partition passes' flush pattern on gfx950 -- every wave appends LINES of B bytes
// (one store instruction per line: 64 lanes x B/64 bytes) to NB of its own slices in turn, as the rings of
// count_scatter_kernel / count_rescatter12_kernel do (csrc/count.hip): consecutive lines of a slice are contiguous, consecutive
// stores of a wave go to different slices.  4 workgroups of 256 threads per CU; non-temporal stores like the product's.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/scatter_write tools/ubench/scatter_write.hip && tools/ubench/scatter_write
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <class T>  // T = uint16_t (128-byte lines), uint32_t (256), u32x2 (512), u32x4 (1024)
__global__ __launch_bounds__(256) void k(T* out, uint32_t nb, uint32_t lines_per_slice, uint32_t rounds) {
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  T* region = out + (size_t)wave * nb * lines_per_slice * 64u;
  T v;
  __builtin_memset(&v, 0x5a, sizeof v);
  uint32_t x = wave * 2654435761u + 12345u;
  for (uint32_t r = 0; r < rounds; ++r)
    for (uint32_t line = 0; line < lines_per_slice; ++line)
      for (uint32_t j = 0; j < nb; ++j) {
        x = x * 1664525u + 1013904223u;                 // (which bucket's ring fills next: any order)
        const uint32_t b = (j + (x >> 28)) % nb;
        __builtin_nontemporal_store(v, &region[((size_t)b * lines_per_slice + line) * 64u + lane]);
      }
}

template <class T>
int run(const char* name, void* d, size_t bytes_total, uint32_t nb) {
  const uint32_t waves = 256u * 4u * 4u;
  const uint32_t lines = (uint32_t)(bytes_total / ((size_t)waves * nb * 64u * sizeof(T)));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<T>, dim3(waves / 4), dim3(256), 0, 0, (T*)d, nb, lines, 1u);
  CHK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<T>, dim3(waves / 4), dim3(256), 0, 0, (T*)d, nb, lines, 1u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double gb = (double)waves * nb * lines * 64.0 * sizeof(T) / 1e9;
  printf("%-28s %2u slices per wave, %6u lines each: %.2f GB in %.3f ms = %.2f TB/s\n", name, nb, lines, gb, best, gb / best);
  return 0;
}

int main() {
  const size_t total = (size_t)4 << 30;
  void* d;
  CHK(hipMalloc(&d, total));
  CHK(hipMemset(d, 0, total));
  for (uint32_t nb : {16u, 32u}) {
    if (run<uint16_t>("128-byte lines (u16 keys)", d, total, nb)) return 1;
    if (run<uint32_t>("256-byte lines (u32 keys)", d, total, nb)) return 1;
    if (run<u32x2>("512-byte lines", d, total, nb)) return 1;
    if (run<u32x4>("1024-byte lines", d, total, nb)) return 1;
  }
  CHK(hipFree(d));
  return 0;
}
