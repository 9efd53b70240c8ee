// runtime_start2 -- follow-up probe: (a) is the 15 ms of the first D2H copy a one-time cost or a property of pinned
// destinations; (b) does a pageable H2D upload of 512 MiB go faster from several threads / streams (the runtime pins the
// source pages on the way: 20-30 ms of the 30-40 ms of a first copy).   ./runtime_start2 [threads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <thread>
#include <vector>
static double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 4;
  const size_t bytes = (size_t)512 << 20;
  double t = now();
  auto lap = [&](const char* what) {
    const double n = now();
    printf("  %-52s %8.1f ms\n", what, (n - t) * 1e3);
    t = n;
  };
  hipInit(0);
  hipSetDevice(0);
  lap("hipInit + hipSetDevice");
  std::vector<hipStream_t> s(nt);
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  lap("streams");
  void* d = nullptr;
  hipMalloc(&d, bytes);
  char* h = (char*)aligned_alloc(2 << 20, bytes);
  memset(h, 1, bytes);
  char* h2 = (char*)aligned_alloc(2 << 20, bytes);
  memset(h2, 2, bytes);
  lap("buffers");
  // (a) D2H first
  void* q = malloc(4 << 20);
  hipMemcpy(q, d, 4 << 20, hipMemcpyDeviceToHost);
  lap("first D2H 4 MiB, pageable destination");
  void* p = nullptr;
  hipHostMalloc(&p, 4 << 20, hipHostMallocDefault);
  lap("hipHostMalloc 4 MiB");
  hipMemcpy(p, d, 4 << 20, hipMemcpyDeviceToHost);
  lap("second D2H 4 MiB, pinned destination");
  hipMemcpy(p, d, 4 << 20, hipMemcpyDeviceToHost);
  lap("third D2H 4 MiB, pinned destination");
  // (b) parallel upload of fresh pageable memory
  {
    std::vector<std::thread> th;
    const size_t part = bytes / nt;
    for (int i = 0; i < nt; ++i)
      th.emplace_back([&, i] {
        hipSetDevice(0);
        hipMemcpyAsync((char*)d + i * part, h + i * part, part, hipMemcpyHostToDevice, s[i]);
        hipStreamSynchronize(s[i]);
      });
    for (auto& x : th) x.join();
  }
  lap("pageable H2D 512 MiB, threads x streams (fresh pages)");
  hipMemcpyAsync(d, h2, bytes, hipMemcpyHostToDevice, s[0]);
  hipStreamSynchronize(s[0]);
  lap("pageable H2D 512 MiB, one thread (fresh pages)");
  return 0;
}
