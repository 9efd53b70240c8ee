#include "device.h"

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

#include "Global.h"

namespace pengk_host {

static pengk_ctx* g_ctx = nullptr;

static int env_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : fallback;
}
int world() {
  static const int w = std::max(1, env_int("WORLD_SIZE", 1));
  return w;
}
bool launched() {
  static const bool l = std::getenv("WORLD_SIZE") != nullptr;
  return l;
}
int rank() {
  static const int r = launched() ? env_int("RANK", 0) : 0;
  return r;
}
static int device_index() { return launched() && std::getenv("LOCAL_RANK") ? env_int("LOCAL_RANK", 0) : Global::device; }

void shard_range(size_t n, int r, int w, size_t* lo, size_t* hi) {
  const size_t base = n / (size_t)w, rem = n % (size_t)w;
  *lo = (size_t)r * base + std::min<size_t>((size_t)r, rem);
  *hi = *lo + base + ((size_t)r < rem ? 1 : 0);
}

void check(int rc, const char* what) {
  if (rc == PENGK_OK) return;
  std::cerr << "Error: " << what << " failed: " << pengk_error_name(rc) << ": " << pengk_last_error() << std::endl;
  exit(1);
}

// Context creation (HIP runtime start-up + code object load, ~0.2 s) can run beside the FASTA reader.
static std::thread g_starter;
static int g_starter_rc = PENGK_OK;
static std::string g_starter_error;

static void join_starter() {
  if (g_starter.joinable()) g_starter.join();
}

void start_context() {
  if (g_ctx || g_starter.joinable()) return;
  g_starter = std::thread([] {
    g_starter_rc = pengk_create(device_index(), &g_ctx);
    if (g_starter_rc != PENGK_OK) g_starter_error = pengk_last_error();  // the message is thread local
  });
  atexit(join_starter);  // an exit() on a FASTA error must not tear the process down under a starting runtime
}

pengk_ctx* context() {
  if (g_starter.joinable()) {
    g_starter.join();
    if (g_starter_rc != PENGK_OK) {
      std::cerr << "Error: pengk_create failed: " << pengk_error_name(g_starter_rc) << ": " << g_starter_error << std::endl;
      exit(1);
    }
  }
  if (!g_ctx) check(pengk_create(device_index(), &g_ctx), "pengk_create");
  static bool options_set = false;
  if (!options_set) {
    options_set = true;
    // The CLI runs the EM in the library's serial mode (em_fast = 2): the reference's float32 arithmetic including its
    // summation order, so that the merge / redundancy decisions downstream -- exact ties in real arithmetic for
    // reverse-complement twins -- fall as the reference's do.  PENGK_EM_FAST=1 (one reciprocal per weight, fp64 tree
    // sums, ~10x faster for large PWM sets) or 0 (reference terms, fp64 tree sums) select the throughput modes.
    int mode = 2;
    if (const char* e = std::getenv("PENGK_EM_FAST")) mode = std::atoi(e);
    check(pengk_set_option(g_ctx, "em_fast", mode), "pengk_set_option");
    if (launched()) {  // RCCL over xGMI; collective: every rank gets here
      // librccl announces its version on stdout while the communicator is built; stdout is the reference's trace
      // (compared byte for byte), so that line goes to stderr
      std::cout.flush();
      fflush(stdout);
      const int saved = dup(1);
      if (saved >= 0) dup2(2, 1);
      const int rc = pengk_comm_init_env(g_ctx);
      fflush(stdout);
      if (saved >= 0) {
        dup2(saved, 1);
        close(saved);
      }
      check(rc, "pengk_comm_init_env");
    }
  }
  return g_ctx;
}

void shutdown() {
  if (g_ctx) pengk_destroy(g_ctx);
  g_ctx = nullptr;
}

}  // namespace pengk_host
