// comm.hip -- C1: the one exchange step of the multi-GPU path, RCCL over xGMI, driven from the C++ side.
//
// One process per GPU.  Sequences shard by whole records, every rank counts its shard (K1 / K1b), then ONE
// all-reduce(sum) of {uint32 counts[4^W], uint64 ltot, uint64 bg[84]} on the context's stream -- the non-overlap
// rule of src/base_pattern.cpp:361-366 never crosses a sequence boundary (:382), so shard counts add exactly -- after
// which the pattern-space sweeps are replicated and the EM splits the PWM list (an all-gather returns the results).
// The reference is a single process: nothing there corresponds to this file.
//
// librccl is opened on first use (dlopen), so single-GPU users of libpengk neither link nor initialise it.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include "pengk_internal.h"

namespace pengk {
namespace {

// the few RCCL entry points used, with the ABI of <rccl/rccl.h> (NCCL 2.x)
typedef struct { char internal[128]; } rccl_unique_id;
static_assert(sizeof(rccl_unique_id) == PENGK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
typedef void* rccl_comm;
enum { RCCL_UINT32 = 3, RCCL_UINT64 = 5, RCCL_UINT8 = 1, RCCL_SUM = 0 };  // ncclDataType_t / ncclRedOp_t values

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return PENGK_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) return fail(PENGK_ERR_DEVICE, "librccl not found: %s", dlerror());
#define PENGK_SYM(field, name)                                            \
  do {                                                                    \
    *(void**)(&g_rccl.field) = dlsym(h, name);                            \
    if (!g_rccl.field) return fail(PENGK_ERR_DEVICE, "librccl lacks %s", name); \
  } while (0)
  PENGK_SYM(GetUniqueId, "ncclGetUniqueId");
  PENGK_SYM(CommInitRank, "ncclCommInitRank");
  PENGK_SYM(CommDestroy, "ncclCommDestroy");
  PENGK_SYM(AllReduce, "ncclAllReduce");
  PENGK_SYM(AllGather, "ncclAllGather");
  PENGK_SYM(GroupStart, "ncclGroupStart");
  PENGK_SYM(GroupEnd, "ncclGroupEnd");
  PENGK_SYM(GetErrorString, "ncclGetErrorString");
#undef PENGK_SYM
  g_rccl.handle = h;
  return PENGK_OK;
}

#define PENGK_RCCL(call)                                                                     \
  do {                                                                                       \
    int r_ = (call);                                                                         \
    if (r_ != 0) return fail(PENGK_ERR_DEVICE, "%s: %s", #call, g_rccl.GetErrorString(r_));   \
  } while (0)

// ---- rendezvous without a launcher library: rank 0 hands the 128-byte id to every other rank over TCP ------------
int send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) return -1;
    c += k;
    n -= (size_t)k;
  }
  return 0;
}
int recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    const ssize_t k = recv(fd, c, n, 0);
    if (k <= 0) return -1;
    c += k;
    n -= (size_t)k;
  }
  return 0;
}

int exchange_id(rccl_unique_id* id, int rank, int world, const char* addr, int port) {
  if (rank == 0) {
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) return fail(PENGK_ERR_DEVICE, "rendezvous: socket() failed");
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_addr.s_addr = htonl(INADDR_ANY);
    sa.sin_port = htons((uint16_t)port);
    if (bind(ls, (sockaddr*)&sa, sizeof sa) != 0 || listen(ls, world) != 0) {
      close(ls);
      return fail(PENGK_ERR_DEVICE, "rendezvous: cannot listen on port %d", port);
    }
    for (int i = 1; i < world; ++i) {
      const int fd = accept(ls, nullptr, nullptr);
      if (fd < 0 || send_all(fd, id, sizeof *id) != 0) {
        if (fd >= 0) close(fd);
        close(ls);
        return fail(PENGK_ERR_DEVICE, "rendezvous: handing the communicator id to a rank failed");
      }
      close(fd);
    }
    close(ls);
    return PENGK_OK;
  }
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char ports[16];
  snprintf(ports, sizeof ports, "%d", port);
  if (getaddrinfo(addr, ports, &hints, &res) != 0 || !res) return fail(PENGK_ERR_DEVICE, "rendezvous: cannot resolve %s", addr);
  int rc = PENGK_ERR_DEVICE;
  for (int attempt = 0; attempt < 600; ++attempt) {  // rank 0 may still be starting: up to a minute
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) break;
    if (connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
      rc = recv_all(fd, id, sizeof *id) == 0 ? PENGK_OK : PENGK_ERR_DEVICE;
      close(fd);
      break;
    }
    close(fd);
    usleep(100000);
  }
  freeaddrinfo(res);
  return rc == PENGK_OK ? PENGK_OK : fail(PENGK_ERR_DEVICE, "rendezvous: no communicator id from %s:%d", addr, port);
}

int env_int(const char* name, int fallback) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : fallback;
}

}  // namespace

void comm_release(pengk_ctx* ctx) {
  if (ctx->comm && g_rccl.handle) (void)g_rccl.CommDestroy((rccl_comm)ctx->comm);
  ctx->comm = nullptr;
  ctx->comm_rank = 0;
  ctx->comm_world = 1;
}

}  // namespace pengk

using namespace pengk;

extern "C" {

int pengk_comm_unique_id(void* id_out) {
  if (!id_out) return fail(PENGK_ERR_ARG, "pengk_comm_unique_id: NULL argument");
  int rc = load_rccl();
  if (rc) return rc;
  rccl_unique_id id;
  PENGK_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return PENGK_OK;
}

int pengk_comm_init(pengk_ctx* ctx, const void* id_bytes, int rank, int world) {
  if (!ctx || !id_bytes) return fail(PENGK_ERR_ARG, "pengk_comm_init: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(PENGK_ERR_ARG, "pengk_comm_init: rank %d of %d", rank, world);
  if (ctx->comm) return fail(PENGK_ERR_ARG, "pengk_comm_init: the context already has a communicator");
  int rc = load_rccl();
  if (rc) return rc;
  rc = enter(ctx);
  if (rc) return rc;
  rccl_unique_id id;
  memcpy(&id, id_bytes, sizeof id);
  rccl_comm comm = nullptr;
  PENGK_RCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return PENGK_OK;
}

int pengk_comm_init_env(pengk_ctx* ctx) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0);
  if (world < 1 || rank < 0 || rank >= world) return fail(PENGK_ERR_ARG, "RANK=%d WORLD_SIZE=%d", rank, world);
  const char* addr = getenv("MASTER_ADDR");
  if (!addr || !*addr) addr = "127.0.0.1";
  // its own port: MASTER_PORT itself belongs to the launcher's store when there is one
  const int port = env_int("PENGK_COMM_PORT", env_int("MASTER_PORT", 29500) + 17);
  int rc = load_rccl();
  if (rc) return rc;
  rccl_unique_id id;
  memset(&id, 0, sizeof id);
  if (rank == 0) PENGK_RCCL(g_rccl.GetUniqueId(&id));
  if (world > 1) {
    rc = exchange_id(&id, rank, world, addr, port);
    if (rc) return rc;
  }
  return pengk_comm_init(ctx, &id, rank, world);
}

int pengk_comm_info(pengk_ctx* ctx, int* rank_out, int* world_out) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  if (rank_out) *rank_out = ctx->comm ? ctx->comm_rank : 0;
  if (world_out) *world_out = ctx->comm ? ctx->comm_world : 1;
  return PENGK_OK;
}

int pengk_comm_destroy(pengk_ctx* ctx) {
  if (!ctx) return PENGK_OK;
  if (ctx->comm) {
    PENGK_HIP(hipSetDevice(ctx->device));
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
  }
  comm_release(ctx);
  return PENGK_OK;
}

int pengk_allreduce_tables(pengk_ctx* ctx, int W, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg) {
  if (!ctx || !d_counts || !d_ltot) return fail(PENGK_ERR_ARG, "pengk_allreduce_tables: NULL argument");
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (!ctx->comm) return PENGK_OK;  // no communicator: the tables are already global (a 1-rank communicator still runs RCCL)
  int rc = enter(ctx);
  if (rc) return rc;
  const size_t np = (size_t)1 << (2 * W);
  rccl_comm comm = (rccl_comm)ctx->comm;
  PENGK_RCCL(g_rccl.GroupStart());
  PENGK_RCCL(g_rccl.AllReduce(d_counts, d_counts, np, RCCL_UINT32, RCCL_SUM, comm, ctx->stream));
  PENGK_RCCL(g_rccl.AllReduce(d_ltot, d_ltot, 1, RCCL_UINT64, RCCL_SUM, comm, ctx->stream));
  if (d_bg) PENGK_RCCL(g_rccl.AllReduce(d_bg, d_bg, 84, RCCL_UINT64, RCCL_SUM, comm, ctx->stream));
  PENGK_RCCL(g_rccl.GroupEnd());
  return PENGK_OK;
}

int pengk_comm_check_bin_bound(pengk_ctx* ctx) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  if (!ctx->d_words) return fail(PENGK_ERR_ARG, "pengk_comm_check_bin_bound: no sequences attached");
  uint64_t bound = ctx->max_bin_bound;
  if (ctx->comm) {
    int rc = enter(ctx);
    if (rc) return rc;
    rc = ensure_scratch(ctx, &ctx->d_misc, &ctx->misc_bytes, sizeof(uint64_t));
    if (rc) return rc;
    PENGK_HIP(hipMemcpyAsync(ctx->d_misc, &bound, sizeof bound, hipMemcpyHostToDevice, ctx->stream));
    PENGK_RCCL(g_rccl.AllReduce(ctx->d_misc, ctx->d_misc, 1, RCCL_UINT64, RCCL_SUM, (rccl_comm)ctx->comm, ctx->stream));
    PENGK_HIP(hipMemcpyAsync(&bound, ctx->d_misc, sizeof bound, hipMemcpyDeviceToHost, ctx->stream));
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (bound >= (1ull << 32))
    return fail(PENGK_ERR_RANGE, "a count bin could reach %llu >= 2^32 over all ranks; use fewer sequences per job",
                (unsigned long long)bound);
  return PENGK_OK;
}

int pengk_allgather(pengk_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  if (!ctx || !d_send || !d_recv) return fail(PENGK_ERR_ARG, "pengk_allgather: NULL argument");
  int rc = enter(ctx);
  if (rc) return rc;
  if (!ctx->comm) {
    if (d_send != d_recv) PENGK_HIP(hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
    return PENGK_OK;
  }
  PENGK_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, RCCL_UINT8, (rccl_comm)ctx->comm, ctx->stream));
  return PENGK_OK;
}

}  // extern "C"
