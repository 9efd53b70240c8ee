#include "device.h"

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

#include "Global.h"
#include "shared/SequenceSet.h"

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

static bool shard_allgather(const void* send, void* recv, size_t bytes) {
  return pengk_comm_host_allgather(send, recv, bytes) == PENGK_OK;
}

void start_sharded_ingest() {
  if (world() <= 1) return;
  check(pengk_comm_host_init_env(), "pengk_comm_host_init_env");
  SequenceShardComm sc;
  sc.rank = rank();
  sc.world = world();
  sc.allgather = shard_allgather;
  SequenceSet::setShardComm(sc);
}

void finish_ranks() {
  if (!launched()) return;
  uint64_t one = 1;
  if (world() > 1) check(pengk_comm_host_allreduce_u64(&one, 1), "pengk_comm_host_allreduce_u64");  // nobody leaves early
  if (g_ctx) check(pengk_comm_destroy(g_ctx), "pengk_comm_destroy");
  pengk_comm_host_shutdown();
}

void check(int rc, const char* what) {
  if (rc == PENGK_OK) return;
  std::cerr << "Error: " << what << " failed: " << pengk_error_name(rc) << ": " << pengk_last_error() << std::endl;
  exit(1);
}

// Context creation (HIP runtime start-up + code object load, ~0.2 s) can run beside the FASTA reader.
static std::thread g_starter;
static int g_starter_rc = PENGK_OK;

static void join_starter() {
  if (g_starter.joinable()) g_starter.join();
}

void start_context() {
  if (g_ctx || g_starter.joinable()) return;
  g_starter = std::thread([] {
    g_starter_rc = pengk_create(device_index(), &g_ctx);
    if (g_starter_rc != PENGK_OK) {
      // no usable gfx950 device: there is no CPU path, and no point in reading gigabytes of FASTA first
      std::cerr << "Error: pengk_create failed: " << pengk_error_name(g_starter_rc) << ": " << pengk_last_error() << std::endl;
      _exit(1);
    }
  });
  atexit(join_starter);  // an exit() on a FASTA error must not tear the process down under a starting runtime
}

pengk_ctx* context() {
  if (g_starter.joinable()) {
    g_starter.join();  // (a failed start has already ended the process)
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
