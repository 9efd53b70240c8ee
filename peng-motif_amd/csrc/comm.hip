// comm.hip -- C1: the one exchange step of the multi-GPU path, driven from the C++ side.
//
// One process per GPU.  Sequences shard by whole records, every rank counts its shard (K1 / K1b), then ONE
// all-reduce(sum) of {uint32 counts[4^W], uint64 ltot, uint64 bg[84]} on the context's stream -- the non-overlap
// rule of src/base_pattern.cpp:361-366 never crosses a sequence boundary (:382), so shard counts add exactly -- after
// which the pattern-space sweeps are replicated and the EM splits the PWM list (an all-gather returns the results).
// The reference is a single process: nothing there corresponds to this file.
//
// Two layers:
//  * the HOST CHANNEL of the process (pengk_comm_host_*): a star of TCP connections to rank 0, set up from the
//    launcher environment.  It carries what the host side of a sharded run has to agree on before any table exists --
//    the number of records, base counts, background counters and warnings of the FASTA shards -- and hands out the
//    RCCL id.  Every socket operation has a deadline (PENGK_COMM_TIMEOUT seconds, default 120): a rank whose peer died
//    fails with PENGK_ERR_DEVICE instead of blocking.  Peers introduce themselves with their rank and a token derived
//    from the launcher environment before they are counted; a stray connection is dropped and never receives anything,
//    and a silent one cannot hold up the real ranks (introductions are read without blocking, against the rendezvous
//    deadline).  The token is a guard against MIX-UPS (two jobs on one host, a stale rank of an earlier run): it is a
//    hash of public launcher values and authenticates nobody.  A job that needs more than that sets PENGK_COMM_TOKEN to
//    a secret; without one rank 0 refuses to listen on anything but MASTER_ADDR's own interface.
//  * the DEVICE TRANSPORT of a context: RCCL over xGMI (default; librccl is opened on first use, so single-GPU users
//    neither link nor initialise it), or -- PENGK_COMM_TRANSPORT=tcp -- the tables staged through host memory and
//    summed over the host channel.  RCCL cannot put two ranks on one GPU; the tcp transport exists so that the whole
//    multi-rank control flow of the CLI can be rehearsed on one card (tests/test_gpu_multirank.py).  It is a test
//    transport, not a fallback: nothing selects it automatically.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

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
  int (*GetVersion)(int*) = nullptr;
  int (*CommAbort)(rccl_comm) = nullptr;  // (optional)
  int version = 0;
};

Rccl g_rccl;
int g_init_abandoned = 0;  // a helper thread of pengk_comm_init is still inside ncclCommInitRank (its deadline passed)

int load_rccl() {
  if (g_rccl.handle) return PENGK_OK;
  void* h = nullptr;
  // A host process that already carries an RCCL (PyTorch ships its own librccl.so and loads it with the process) must
  // not get a second instance beside it: two copies of the library would each keep their own transports, shared-memory
  // segments and IPC state on the same GPUs.  Take the copy that is mapped, if there is one.
  if (FILE* maps = fopen("/proc/self/maps", "r")) {
    char line[1024];
    while (!h && fgets(line, sizeof line, maps)) {
      char* path = strchr(line, '/');
      if (!path || !strstr(path, "librccl")) continue;
      path[strcspn(path, "\n")] = 0;
      h = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    }
    fclose(maps);
  }
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
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
  PENGK_SYM(GetVersion, "ncclGetVersion");
#undef PENGK_SYM
  *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
  // The entry points above are declared by hand with the NCCL 2.x ABI (128-byte id, ncclDataType_t / ncclRedOp_t values):
  // the library says which NCCL API it implements, and anything older than 2.0 is refused.  (ncclGetVersion's code is
  // major * 1000 + minor * 100 + patch up to 2.8 and major * 10000 + minor * 100 + patch from 2.9 on: either way >= 2000.)
  int v = 0;
  if (g_rccl.GetVersion(&v) != 0 || v < 2000)
    return fail(PENGK_ERR_DEVICE, "librccl reports NCCL API version code %d: this library binds the 2.x ABI", v);
  g_rccl.version = v;
  if (!getenv("PENGK_COMM_QUIET"))
    fprintf(stderr, "[pengk] librccl loaded: NCCL API version code %d (%d.%d.%d)%s\n", v, v >= 20000 ? v / 10000 : v / 1000,
            v >= 20000 ? (v / 100) % 100 : (v / 100) % 10, v % 100, g_rccl.CommAbort ? "" : ", no ncclCommAbort");
  g_rccl.handle = h;
  return PENGK_OK;
}

#define PENGK_RCCL(call)                                                                     \
  do {                                                                                       \
    int r_ = (call);                                                                         \
    if (r_ != 0) return fail(PENGK_ERR_DEVICE, "%s: %s", #call, g_rccl.GetErrorString(r_));   \
  } while (0)

int env_int(const char* name, int fallback) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : fallback;
}

int load_rccl_or_test_failure(int rank) {
  const char* t = getenv("PENGK_COMM_TEST_FAIL_LOAD");
  if (t && *t && atoi(t) == rank) return fail(PENGK_ERR_DEVICE, "librccl not found: forced by PENGK_COMM_TEST_FAIL_LOAD");
  return load_rccl();
}

// ---- host channel ------------------------------------------------------------------------------------------------
double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct HostChannel {
  bool up = false;
  int rank = 0, world = 1;
  int timeout_s = 120;
  std::vector<int> fds;  // rank 0: fds[r] for r >= 1; other ranks: fds[0] = the connection to rank 0
  std::mutex mu;
};
HostChannel g_chan;

void set_io_deadline(int fd, int seconds) {
  timeval tv;
  tv.tv_sec = seconds;
  tv.tv_usec = 0;
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
}

int send_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return -1;  // error, peer gone, or the deadline (EAGAIN with SO_SNDTIMEO)
    c += k;
    n -= (size_t)k;
  }
  return 0;
}
int recv_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    const ssize_t k = recv(fd, c, n, 0);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return -1;
    c += k;
    n -= (size_t)k;
  }
  return 0;
}

void chan_close() {
  for (int fd : g_chan.fds)
    if (fd >= 0) close(fd);
  g_chan.fds.clear();
  g_chan.up = false;
  g_chan.rank = 0;
  g_chan.world = 1;
}

struct Hello {
  uint64_t magic, token;
  uint32_t rank, world;
};
constexpr uint64_t HELLO_MAGIC = 0x70656e676b633031ull;  // "pengkc01"

// every rank of one job derives the same token from the launcher environment: it keeps the ranks of DIFFERENT jobs apart
// (a mix-up guard, computable by anyone who knows the launcher values); PENGK_COMM_TOKEN adds a job secret
uint64_t job_token(const char* addr, int port, int world) {
  std::string s = std::string("pengk|") + addr + "|" + std::to_string(port) + "|" + std::to_string(world) + "|";
  for (const char* name : {"TORCHELASTIC_RUN_ID", "PENGK_COMM_TOKEN"}) {
    const char* e = getenv(name);
    s += e ? e : "";
    s += "|";
  }
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
  return h;
}

int chan_fail(const char* fmt, const char* addr, int port) {
  chan_close();
  return fail(PENGK_ERR_DEVICE, fmt, addr, port);
}

int chan_open(int rank, int world, const char* addr, int port, int timeout_s) {
  g_chan.rank = rank;
  g_chan.world = world;
  g_chan.timeout_s = timeout_s;
  if (world == 1) {
    g_chan.up = true;
    return PENGK_OK;
  }
  const uint64_t token = job_token(addr, port, world);
  const double deadline = now_s() + timeout_s;
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char ports[16];
  snprintf(ports, sizeof ports, "%d", port);
  if (getaddrinfo(addr, ports, &hints, &res) != 0 || !res) return fail(PENGK_ERR_DEVICE, "rendezvous: cannot resolve %s", addr);

  if (rank == 0) {
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) {
      freeaddrinfo(res);
      return fail(PENGK_ERR_DEVICE, "rendezvous: socket() failed");
    }
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    // MASTER_ADDR's own interface, not every interface of the host.  A name that resolves to an address this host does
    // not own (NAT, a launcher quirk) may fall back to all interfaces only when the job carries a secret
    // (PENGK_COMM_TOKEN): the derived token alone is computable by anyone who can reach the port.
    bool bound = bind(ls, res->ai_addr, res->ai_addrlen) == 0;
    if (!bound && errno == EADDRNOTAVAIL) {
      const char* secret = getenv("PENGK_COMM_TOKEN");
      if (!secret || !*secret) {
        close(ls);
        freeaddrinfo(res);
        return fail(PENGK_ERR_DEVICE,
                    "rendezvous: %s is not an address of this host; set MASTER_ADDR to one, or PENGK_COMM_TOKEN to a job secret "
                    "to listen on all interfaces", addr);
      }
      sockaddr_in sa{};
      sa.sin_family = AF_INET;
      sa.sin_addr.s_addr = htonl(INADDR_ANY);
      sa.sin_port = htons((uint16_t)port);
      bound = bind(ls, (sockaddr*)&sa, sizeof sa) == 0;
    }
    freeaddrinfo(res);
    if (!bound || listen(ls, world + 8) != 0) {
      close(ls);
      return fail(PENGK_ERR_DEVICE, "rendezvous: cannot listen on %s:%d", addr, port);
    }
    g_chan.fds.assign((size_t)world, -1);
    // Connections that have not introduced themselves yet wait in a pending set and are read without blocking: a silent
    // or slow connection costs the real ranks nothing, and is dropped after 5 s (or when the set is full).
    struct Pending {
      int fd;
      double drop_at;
      size_t got;
      Hello h;
    };
    std::vector<Pending> pending;
    auto drop_pending = [&]() {
      for (Pending& p : pending) close(p.fd);
      pending.clear();
    };
    int have = 0;
    while (have < world - 1) {
      const double now = now_s();
      double wake = deadline;
      for (const Pending& p : pending) wake = p.drop_at < wake ? p.drop_at : wake;
      std::vector<pollfd> pfs(1 + pending.size());
      pfs[0] = pollfd{ls, POLLIN, 0};
      for (size_t i = 0; i < pending.size(); ++i) pfs[1 + i] = pollfd{pending[i].fd, POLLIN, 0};
      const double left = wake - now;
      const int ready = now >= deadline ? -1 : poll(pfs.data(), (nfds_t)pfs.size(), left > 0 ? (int)(left * 1000) + 1 : 0);
      if (ready < 0 && (now >= deadline || errno != EINTR)) {
        close(ls);
        drop_pending();
        chan_close();
        return fail(PENGK_ERR_DEVICE, "rendezvous: only %d of %d ranks reached %s:%d within %d s", have + 1, world, addr, port,
                    timeout_s);
      }
      if (ready < 0) continue;
      std::vector<Pending> keep;
      for (size_t i = 0; i < pending.size(); ++i) {
        Pending p = pending[i];
        bool done = false, bad = false;
        if (pfs[1 + i].revents & (POLLIN | POLLHUP | POLLERR)) {
          const ssize_t k = recv(p.fd, (char*)&p.h + p.got, sizeof(Hello) - p.got, MSG_DONTWAIT);
          if (k > 0) {
            p.got += (size_t)k;
            done = p.got == sizeof(Hello);
          } else if (k == 0 || (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) {
            bad = true;
          }
        }
        if (done) {
          const Hello& h = p.h;
          if (h.magic != HELLO_MAGIC || h.token != token || h.world != (uint32_t)world || h.rank == 0 || h.rank >= (uint32_t)world ||
              g_chan.fds[h.rank] >= 0) {
            close(p.fd);
          } else {
            set_io_deadline(p.fd, timeout_s);
            g_chan.fds[h.rank] = p.fd;
            ++have;
          }
        } else if (bad || now_s() >= p.drop_at) {
          close(p.fd);
        } else {
          keep.push_back(p);
        }
      }
      pending.swap(keep);
      if (pfs[0].revents & POLLIN) {
        const int fd = accept(ls, nullptr, nullptr);
        if (fd >= 0) {
          if (pending.size() >= (size_t)world + 64) {  // more strangers than a job has ranks: the oldest goes
            close(pending.front().fd);
            pending.erase(pending.begin());
          }
          int one = 1;
          setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
          pending.push_back(Pending{fd, now_s() + 5.0, 0, Hello{}});
        }
      }
    }
    drop_pending();
    close(ls);
    for (int r = 1; r < world; ++r)  // everyone is here: release the peers
      if (send_all(g_chan.fds[r], &HELLO_MAGIC, sizeof HELLO_MAGIC) != 0) return chan_fail("rendezvous: a rank left %s:%d", addr, port);
    g_chan.up = true;
    return PENGK_OK;
  }

  int fd = -1;
  while (now_s() < deadline) {  // rank 0 may still be starting
    fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) break;
    if (connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
    close(fd);
    fd = -1;
    usleep(50000);
  }
  freeaddrinfo(res);
  if (fd < 0) return fail(PENGK_ERR_DEVICE, "rendezvous: rank 0 not reachable at %s:%d within %d s", addr, port, timeout_s);
  g_chan.fds.assign(1, fd);
  set_io_deadline(fd, timeout_s);
  const Hello h{HELLO_MAGIC, token, (uint32_t)rank, (uint32_t)world};
  uint64_t ack = 0;
  if (send_all(fd, &h, sizeof h) != 0 || recv_all(fd, &ack, sizeof ack) != 0 || ack != HELLO_MAGIC)
    return chan_fail("rendezvous: rank 0 at %s:%d did not admit this rank (other ranks missing, or a different job)", addr, port);
  g_chan.up = true;
  return PENGK_OK;
}

int chan_env_open() {
  if (g_chan.up) return PENGK_OK;
  const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0);
  if (world < 1 || rank < 0 || rank >= world) return fail(PENGK_ERR_ARG, "RANK=%d WORLD_SIZE=%d", rank, world);
  const char* addr = getenv("MASTER_ADDR");
  if (!addr || !*addr) addr = "127.0.0.1";
  // its own port: MASTER_PORT itself belongs to the launcher's store when there is one
  const int port = env_int("PENGK_COMM_PORT", env_int("MASTER_PORT", 29500) + 17);
  const int timeout_s = env_int("PENGK_COMM_TIMEOUT", 120);
  return chan_open(rank, world, addr, port, timeout_s < 1 ? 1 : timeout_s);
}

int chan_lost() {
  chan_close();
  return fail(PENGK_ERR_DEVICE, "host channel: a rank did not answer (it failed, or the %d s deadline passed)", g_chan.timeout_s);
}

// recv[r * bytes ...) = rank r's send[0 .. bytes)
int chan_allgather(const void* send, void* recv, size_t bytes) {
  if (!g_chan.up) return fail(PENGK_ERR_ARG, "host channel is not open");
  const int w = g_chan.world;
  if (w == 1) {
    if (recv != send) memmove(recv, send, bytes);
    return PENGK_OK;
  }
  if (g_chan.rank == 0) {
    memmove(recv, send, bytes);
    for (int r = 1; r < w; ++r)
      if (recv_all(g_chan.fds[r], (char*)recv + (size_t)r * bytes, bytes) != 0) return chan_lost();
    for (int r = 1; r < w; ++r)
      if (send_all(g_chan.fds[r], recv, (size_t)w * bytes) != 0) return chan_lost();
    return PENGK_OK;
  }
  if (send_all(g_chan.fds[0], send, bytes) != 0 || recv_all(g_chan.fds[0], recv, (size_t)w * bytes) != 0) return chan_lost();
  return PENGK_OK;
}

// in-place sum over the ranks
template <class T>
int chan_allreduce_sum(T* buf, size_t n) {
  if (!g_chan.up) return fail(PENGK_ERR_ARG, "host channel is not open");
  const int w = g_chan.world;
  if (w == 1 || n == 0) return PENGK_OK;
  const size_t bytes = n * sizeof(T);
  if (g_chan.rank == 0) {
    std::vector<T> in(n);
    for (int r = 1; r < w; ++r) {
      if (recv_all(g_chan.fds[r], in.data(), bytes) != 0) return chan_lost();
      for (size_t i = 0; i < n; ++i) buf[i] += in[i];
    }
    for (int r = 1; r < w; ++r)
      if (send_all(g_chan.fds[r], buf, bytes) != 0) return chan_lost();
    return PENGK_OK;
  }
  if (send_all(g_chan.fds[0], buf, bytes) != 0 || recv_all(g_chan.fds[0], buf, bytes) != 0) return chan_lost();
  return PENGK_OK;
}

int chan_bcast(void* buf, size_t bytes) {
  if (!g_chan.up) return fail(PENGK_ERR_ARG, "host channel is not open");
  if (g_chan.world == 1) return PENGK_OK;
  if (g_chan.rank == 0) {
    for (int r = 1; r < g_chan.world; ++r)
      if (send_all(g_chan.fds[r], buf, bytes) != 0) return chan_lost();
    return PENGK_OK;
  }
  return recv_all(g_chan.fds[0], buf, bytes) == 0 ? PENGK_OK : chan_lost();
}

bool use_tcp(const pengk_ctx* ctx) { return ctx->comm_transport == PENGK_TRANSPORT_TCP; }
bool single(const pengk_ctx* ctx) { return ctx->comm_transport == PENGK_TRANSPORT_NONE; }

// tcp transport: a device array summed over the ranks through host memory
template <class T>
int staged_allreduce(pengk_ctx* ctx, T* d_buf, size_t n) {
  std::vector<T> h(n);
  PENGK_HIP(hipMemcpyAsync(h.data(), d_buf, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  int rc = chan_allreduce_sum(h.data(), n);
  if (rc) return rc;
  PENGK_HIP(hipMemcpyAsync(d_buf, h.data(), n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  PENGK_HIP(hipStreamSynchronize(ctx->stream));
  return PENGK_OK;
}

}  // namespace

void comm_release(pengk_ctx* ctx) {
  if (ctx->comm && g_rccl.handle) (void)g_rccl.CommDestroy((rccl_comm)ctx->comm);
  ctx->comm = nullptr;
  ctx->comm_transport = PENGK_TRANSPORT_NONE;
  ctx->comm_rank = 0;
  ctx->comm_world = 1;
}

}  // namespace pengk

using namespace pengk;

extern "C" {

// ---- host channel -------------------------------------------------------------------------------------------------
int pengk_comm_host_init_env(void) {
  std::lock_guard<std::mutex> lock(g_chan.mu);
  return chan_env_open();
}

int pengk_comm_host_info(int* rank_out, int* world_out) {
  std::lock_guard<std::mutex> lock(g_chan.mu);
  if (rank_out) *rank_out = g_chan.up ? g_chan.rank : 0;
  if (world_out) *world_out = g_chan.up ? g_chan.world : 1;
  return PENGK_OK;
}

int pengk_comm_host_allgather(const void* h_send, void* h_recv, size_t bytes_per_rank) {
  if (bytes_per_rank && (!h_send || !h_recv)) return fail(PENGK_ERR_ARG, "pengk_comm_host_allgather: NULL argument");
  std::lock_guard<std::mutex> lock(g_chan.mu);
  return chan_allgather(h_send, h_recv, bytes_per_rank);
}

int pengk_comm_host_allreduce_u64(uint64_t* h_buf, size_t n) {
  if (n && !h_buf) return fail(PENGK_ERR_ARG, "pengk_comm_host_allreduce_u64: NULL argument");
  std::lock_guard<std::mutex> lock(g_chan.mu);
  return chan_allreduce_sum(h_buf, n);
}

int pengk_comm_host_shutdown(void) {
  std::lock_guard<std::mutex> lock(g_chan.mu);
  chan_close();
  return PENGK_OK;
}

// ---- device transport ---------------------------------------------------------------------------------------------
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
  if (!single(ctx)) return fail(PENGK_ERR_ARG, "pengk_comm_init: the context already has a communicator");
  int rc = load_rccl();
  if (rc) return rc;
  rc = enter(ctx);
  if (rc) return rc;
  // ncclCommInitRank blocks until every rank of the id has entered it and has no deadline of its own: a rank that
  // never arrives (it failed to load librccl, died, or belongs to another job) would park the others for good.  The
  // call therefore runs on a helper thread and this one waits for it PENGK_COMM_TIMEOUT seconds; past the deadline
  // the rank reports the error (the helper is left behind: the process is expected to exit, as after any comm error).
  struct InitCall {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int result = 0;
    rccl_comm comm = nullptr;
    rccl_unique_id id;
    int device = 0, rank = 0, world = 1;
  };
  auto call = std::make_shared<InitCall>();
  memcpy(&call->id, id_bytes, sizeof call->id);
  call->device = ctx->device;
  call->rank = rank;
  call->world = world;
  const int timeout_s = env_int("PENGK_COMM_TIMEOUT", 120) < 1 ? 1 : env_int("PENGK_COMM_TIMEOUT", 120);
  std::thread([call]() {
    int r = hipSetDevice(call->device) == hipSuccess ? 0 : -1;
    rccl_comm comm = nullptr;
    if (r == 0) r = g_rccl.CommInitRank(&comm, call->world, call->id, call->rank);
    std::lock_guard<std::mutex> lock(call->mu);
    call->result = r;
    call->comm = comm;
    call->done = true;
    call->cv.notify_all();
  }).detach();
  {
    std::unique_lock<std::mutex> lock(call->mu);
    if (!call->cv.wait_for(lock, std::chrono::seconds(timeout_s), [&] { return call->done; })) {
      // The helper thread is still inside ncclCommInitRank and stays there: there is no communicator yet that ncclCommAbort
      // could be given.  A process in this state must not run its exit handlers (HIP's and RCCL's static destructors
      // under a live thread inside RCCL can crash or hang, which would defeat the deadline): callers ask
      // pengk_comm_init_abandoned() and leave with _exit (host/device.cpp does).
      g_init_abandoned = 1;
      return fail(PENGK_ERR_DEVICE, "ncclCommInitRank: rank %d of %d still waiting for the other ranks after %d s", rank, world,
                  timeout_s);
    }
    if (call->result != 0)
      return fail(PENGK_ERR_DEVICE, "ncclCommInitRank: %s", call->result < 0 ? "hipSetDevice failed" : g_rccl.GetErrorString(call->result));
    ctx->comm = call->comm;
  }
  ctx->comm_transport = PENGK_TRANSPORT_RCCL;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return PENGK_OK;
}

// every rank contributes one status word over the host channel; the first failing rank (if any) is named to all of them
static int chan_agree(int32_t mine, const char* what) {
  std::vector<int32_t> all((size_t)g_chan.world, 0);
  int rc = chan_allgather(&mine, all.data(), sizeof mine);
  if (rc) return rc;
  for (int r = 0; r < g_chan.world; ++r)
    if (all[(size_t)r] != 0) return r == g_chan.rank ? mine : fail(PENGK_ERR_DEVICE, "rank %d %s", r, what);
  return PENGK_OK;
}

int pengk_comm_init_env(pengk_ctx* ctx) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  if (!single(ctx)) return fail(PENGK_ERR_ARG, "pengk_comm_init_env: the context already has a communicator");
  std::unique_lock<std::mutex> lock(g_chan.mu);
  int rc = chan_env_open();
  if (rc) return rc;
  const int rank = g_chan.rank, world = g_chan.world;
  const char* transport = getenv("PENGK_COMM_TRANSPORT");
  if (transport && strcmp(transport, "tcp") == 0) {
    ctx->comm_transport = PENGK_TRANSPORT_TCP;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return PENGK_OK;
  }
  if (transport && *transport && strcmp(transport, "rccl") != 0)
    return fail(PENGK_ERR_ARG, "PENGK_COMM_TRANSPORT=%s (rccl or tcp)", transport);
  // Nobody enters ncclCommInitRank before EVERY rank has said, over the host channel (which has deadlines), that it
  // can: librccl loaded on all of them, the id created on rank 0.  PENGK_COMM_TEST_FAIL_LOAD=<rank> is the test hook
  // that makes one rank fail here (tests/test_gpu_multirank.py).
  int32_t loaded = load_rccl_or_test_failure(rank);
  rc = chan_agree(loaded, "could not load librccl");
  if (rc) return rc;
  struct {
    int32_t rc;
    rccl_unique_id id;
  } msg;
  memset(&msg, 0, sizeof msg);
  if (rank == 0 && g_rccl.GetUniqueId(&msg.id) != 0) msg.rc = fail(PENGK_ERR_DEVICE, "ncclGetUniqueId failed");
  rc = chan_bcast(&msg, sizeof msg);
  if (rc) return rc;
  if (msg.rc) return rank == 0 ? msg.rc : fail(PENGK_ERR_DEVICE, "rank 0 could not create an RCCL id");
  lock.unlock();
  {  // test hook: this rank dies here -- between "everybody can" and ncclCommInitRank (tests/test_gpu_multirank.py)
    const char* t = getenv("PENGK_COMM_TEST_DIE_BEFORE_INIT");
    if (t && *t && atoi(t) == rank) _exit(3);
  }
  const int32_t inited = pengk_comm_init(ctx, &msg.id, rank, world);
  // ... and nobody USES the communicator before every rank has one: a one-sided failure inside ncclCommInitRank
  // surfaces on all ranks here, within the host channel's deadline
  lock.lock();
  rc = chan_agree(inited, "could not create its RCCL communicator");
  lock.unlock();
  if (rc && !inited) {
    // this rank has a communicator, another one has not: it is abandoned, not destroyed (ncclCommDestroy may wait for
    // the peers that never joined); the caller reports the error and the process ends
    ctx->comm = nullptr;
    ctx->comm_transport = PENGK_TRANSPORT_NONE;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
  }
  return rc;
}

int pengk_comm_info(pengk_ctx* ctx, int* rank_out, int* world_out) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  if (rank_out) *rank_out = single(ctx) ? 0 : ctx->comm_rank;
  if (world_out) *world_out = single(ctx) ? 1 : ctx->comm_world;
  return PENGK_OK;
}

int pengk_comm_init_abandoned(void) { return g_init_abandoned; }

int pengk_comm_rccl_version(void) { return g_rccl.handle ? g_rccl.version : 0; }

int pengk_comm_destroy(pengk_ctx* ctx) {
  if (!ctx) return PENGK_OK;
  if (!single(ctx)) {
    PENGK_HIP(hipSetDevice(ctx->device));
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
  }
  comm_release(ctx);
  return PENGK_OK;
}

int pengk_allreduce_tables(pengk_ctx* ctx, int W, uint32_t* d_counts, uint64_t* d_ltot, uint64_t* d_bg) {
  if (!ctx || !d_counts || !d_ltot) return fail(PENGK_ERR_ARG, "pengk_allreduce_tables: NULL argument");
  if (!valid_w(W)) return fail(PENGK_ERR_ARG, "pattern length %d unsupported", W);
  if (single(ctx)) return PENGK_OK;  // no communicator: the tables are already global (a 1-rank communicator still runs RCCL)
  int rc = enter(ctx);
  if (rc) return rc;
  const size_t np = (size_t)1 << (2 * W);
  if (use_tcp(ctx)) {
    std::lock_guard<std::mutex> lock(g_chan.mu);
    rc = staged_allreduce(ctx, d_counts, np);
    if (!rc) rc = staged_allreduce(ctx, d_ltot, 1);
    if (!rc && d_bg) rc = staged_allreduce(ctx, d_bg, 84);
    return rc;
  }
  rccl_comm comm = (rccl_comm)ctx->comm;
  // a group that was opened is always closed, whatever happens in between
  PENGK_RCCL(g_rccl.GroupStart());
  int bad = g_rccl.AllReduce(d_counts, d_counts, np, RCCL_UINT32, RCCL_SUM, comm, ctx->stream);
  if (!bad) bad = g_rccl.AllReduce(d_ltot, d_ltot, 1, RCCL_UINT64, RCCL_SUM, comm, ctx->stream);
  if (!bad && d_bg) bad = g_rccl.AllReduce(d_bg, d_bg, 84, RCCL_UINT64, RCCL_SUM, comm, ctx->stream);
  const int end = g_rccl.GroupEnd();
  if (bad || end) return fail(PENGK_ERR_DEVICE, "grouped ncclAllReduce: %s", g_rccl.GetErrorString(bad ? bad : end));
  return PENGK_OK;
}

int pengk_comm_check_bin_bound(pengk_ctx* ctx) {
  if (!ctx) return fail(PENGK_ERR_ARG, "ctx is NULL");
  if (!ctx->d_words) return fail(PENGK_ERR_ARG, "pengk_comm_check_bin_bound: no sequences attached");
  uint64_t bound = ctx->max_bin_bound;
  if (use_tcp(ctx)) {
    std::lock_guard<std::mutex> lock(g_chan.mu);
    int rc = chan_allreduce_sum(&bound, 1);
    if (rc) return rc;
  } else if (!single(ctx)) {
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
  if (single(ctx)) {
    if (d_send != d_recv) PENGK_HIP(hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
    return PENGK_OK;
  }
  if (use_tcp(ctx)) {
    std::vector<char> mine(bytes_per_rank), all(bytes_per_rank * (size_t)ctx->comm_world);
    PENGK_HIP(hipMemcpyAsync(mine.data(), d_send, bytes_per_rank, hipMemcpyDeviceToHost, ctx->stream));
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
    {
      std::lock_guard<std::mutex> lock(g_chan.mu);
      rc = chan_allgather(mine.data(), all.data(), bytes_per_rank);
    }
    if (rc) return rc;
    PENGK_HIP(hipMemcpyAsync(d_recv, all.data(), all.size(), hipMemcpyHostToDevice, ctx->stream));
    PENGK_HIP(hipStreamSynchronize(ctx->stream));
    return PENGK_OK;
  }
  PENGK_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, RCCL_UINT8, (rccl_comm)ctx->comm, ctx->stream));
  return PENGK_OK;
}

}  // extern "C"
