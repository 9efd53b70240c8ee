// runtime_start -- where the HIP runtime's start goes on this box (the 0.1-0.2 s every peng_motif run waits for).
//   hipcc -O2 -o runtime_start runtime_start.cpp ; ./runtime_start [bytes to upload from pageable memory]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)512 << 20;
  double t0 = now(), t = t0;
  auto lap = [&](const char* what) {
    const double n = now();
    printf("  %-44s %8.1f ms\n", what, (n - t) * 1e3);
    t = n;
  };
  hipInit(0);
  lap("hipInit");
  int n = 0;
  hipGetDeviceCount(&n);
  lap("hipGetDeviceCount");
  hipSetDevice(0);
  lap("hipSetDevice");
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  lap("hipStreamCreate (first)");
  void* d = nullptr;
  hipMalloc(&d, bytes);
  lap("hipMalloc");
  char* h = (char*)aligned_alloc(2 << 20, bytes);
  memset(h, 1, bytes);
  lap("host buffer touched");
  hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);
  lap("pageable H2D copy");
  hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);
  lap("pageable H2D copy again");
  hipHostRegister(h, bytes, hipHostRegisterDefault);
  lap("hipHostRegister");
  hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s);
  lap("registered H2D copy");
  hipHostUnregister(h);
  lap("hipHostUnregister");
  void* p = nullptr;
  hipHostMalloc(&p, 4 << 20, hipHostMallocDefault);
  lap("hipHostMalloc 4 MiB");
  hipMemcpy(p, d, 4 << 20, hipMemcpyDeviceToHost);
  lap("D2H 4 MiB into pinned");
  void* q = malloc(4 << 20);
  hipMemcpy(q, d, 4 << 20, hipMemcpyDeviceToHost);
  lap("D2H 4 MiB into pageable");
  printf("  total %.1f ms, %d device(s)\n", (now() - t0) * 1e3, n);
  return 0;
}
