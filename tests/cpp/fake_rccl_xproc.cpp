// fake_rccl_xproc.cpp — TEST INFRASTRUCTURE: a CROSS-PROCESS stand-in for the RCCL entry points comm.hip binds.
//
// tests/cpp/fake_rccl.cpp serves ranks that are threads of one process.  The path the driver runs
// (`python bench.py --gpus N`: self-spawned ranks, one PROCESS each, gloo rendezvous for the 128-byte id,
// CApiCommunicator -> ah_comm_create -> ah_all_gather_columns_begin/_end) has one process per rank, and real RCCL
// refuses two ranks on one device — so on the one GPU of a test box that path could not run at N > 1 at all.  This
// library lets it: every rank is its own process (all on GPU 0), the "fabric" is a file-backed shared mapping.
//
//   ncclGetUniqueId      rank 0 creates the mapping file (sparse, under $AH_FAKE_RCCL_DIR or /tmp); the id is its path
//   ncclCommInitRank     every rank maps it and waits until `world` ranks have joined (collective, like the real one)
//   ncclSend / ncclRecv  a message travels as <= 1 MiB chunks through a double-buffered slot per ordered pair of ranks:
//                        device -> mapping on the sender, mapping -> device on the receiver.  Operations of one
//                        ncclGroup are progressed TOGETHER (round robin, non-blocking), like RCCL issues a group
//                        together: ranks that each send before they receive cannot deadlock on a full slot
//   ncclAllGather / ncclAllReduce(max, f64)   through a small per-rank collective area + a generation barrier
//   ncclCommAbort        raises a flag every waiting peer observes (their call returns an error instead of hanging)
//
// Every wait is bounded (AH_FAKE_RCCL_TIMEOUT_S, default 120 s): a protocol bug fails the test, it never hangs the box.
// Host staging instead of hipIpc handles on purpose: dmabuf IPC needs a peer-credential hand-off that a test double
// does not have to depend on.  Bandwidth is irrelevant here — what is under test is the product's control flow
// (counts, offsets, pairing of sends and receives, merge) across process boundaries.
// Loaded through AH_RCCL_LIBRARY=<path> (comm.hip); nothing in the product links or ships it.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

extern "C" {
typedef struct XComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;  // 0 = ncclSuccess, 2 = ncclSystemError, 4 = ncclInvalidArgument
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}

namespace {

constexpr int MAX_WORLD = 8;
constexpr size_t CHUNK = 1u << 20;
constexpr size_t COLL_BYTES = 16384;
constexpr uint32_t MAGIC = 0x46524343;  // "FRCC"

struct Slot {  // one direction of one ordered pair (src -> dst)
  std::atomic<uint64_t> posted;  // chunks published by the sender
  std::atomic<uint64_t> taken;   // chunks consumed by the receiver
  uint64_t chunk_bytes[2];
  uint64_t msg_bytes[2];  // size of the whole message the chunk belongs to
};

struct Shm {
  std::atomic<uint32_t> magic;
  std::atomic<int> world, joined, left, aborted;
  std::atomic<uint64_t> coll_arrived, coll_gen;
  Slot slots[MAX_WORLD * MAX_WORLD];
  alignas(64) char coll[MAX_WORLD][COLL_BYTES];
};

constexpr size_t DATA_OFF = (sizeof(Shm) + 4095) & ~size_t(4095);
constexpr size_t MAP_BYTES = DATA_OFF + (size_t)MAX_WORLD * MAX_WORLD * 2 * CHUNK;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double timeout_s() {
  const char* e = getenv("AH_FAKE_RCCL_TIMEOUT_S");
  const double v = e ? atof(e) : 0;
  return v > 0 ? v : 120.0;
}

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return 1;
  }
}

struct PendingOp {
  bool is_send;
  char* ptr;
  size_t bytes;
  int peer;
  hipStream_t stream;
  size_t done = 0;
  bool finished = false;
};

}  // namespace

struct XComm {
  Shm* shm;
  char* data;
  int rank, world;
  std::string path;
  char* chunk(int src, int dst, int k) const { return data + ((size_t)(src * MAX_WORLD + dst) * 2 + k) * CHUNK; }
  Slot& slot(int src, int dst) const { return shm->slots[src * MAX_WORLD + dst]; }
};

namespace {

thread_local int t_group_depth = 0;
thread_local std::vector<std::pair<XComm*, PendingOp>> t_pending;

bool fail(const char* what) {
  fprintf(stderr, "fake_rccl_xproc: %s\n", what);
  fflush(stderr);
  return false;
}

// wait until pred() or abort / timeout
template <typename P>
bool wait_for(XComm* c, P pred, const char* what) {
  const double t0 = now_s(), lim = timeout_s();
  int spins = 0;
  while (!pred()) {
    if (c->shm->aborted.load(std::memory_order_acquire)) return fail("a peer aborted the communicator");
    if (++spins > 64) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (now_s() - t0 > lim) {
        char msg[160];
        snprintf(msg, sizeof msg, "rank %d timed out after %.0f s in %s", c->rank, lim, what);
        c->shm->aborted.store(1, std::memory_order_release);  // take the peers down with us rather than leave them waiting
        return fail(msg);
      }
    }
  }
  return true;
}

bool barrier(XComm* c, const char* what) {
  Shm* s = c->shm;
  const uint64_t gen = s->coll_gen.load(std::memory_order_acquire);
  if (s->coll_arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint64_t)c->world) {
    s->coll_arrived.store(0, std::memory_order_relaxed);
    s->coll_gen.fetch_add(1, std::memory_order_release);
    return true;
  }
  return wait_for(c, [&] { return s->coll_gen.load(std::memory_order_acquire) != gen; }, what);
}

ncclResult_t run_ops(std::vector<std::pair<XComm*, PendingOp>>& ops) {
  // earlier work of the calling stream (the kernels that produced the send buffers) must have finished
  for (auto& e : ops)
    if (e.second.is_send && hipStreamSynchronize(e.second.stream) != hipSuccess) {
      ops.clear();
      return 1;
    }
  const double t0 = now_s(), lim = timeout_s();
  size_t left = ops.size();
  ncclResult_t rc = 0;
  while (left && rc == 0) {
    bool progressed = false;
    // program order per (comm, peer, direction): only the first unfinished op of a queue may move
    std::map<std::tuple<XComm*, int, bool>, bool> busy;
    for (auto& e : ops) {
      PendingOp& op = e.second;
      if (op.finished) continue;
      XComm* c = e.first;
      auto key = std::make_tuple(c, op.peer, op.is_send);
      if (busy[key]) continue;
      busy[key] = true;
      if (op.is_send) {
        Slot& sl = c->slot(c->rank, op.peer);
        const uint64_t p = sl.posted.load(std::memory_order_relaxed);
        if (p - sl.taken.load(std::memory_order_acquire) >= 2) continue;
        const int k = (int)(p & 1);
        const size_t n = std::min(CHUNK, op.bytes - op.done);
        if (n && hipMemcpy(c->chunk(c->rank, op.peer, k), op.ptr + op.done, n, hipMemcpyDeviceToHost) != hipSuccess) {
          rc = 1;
          break;
        }
        sl.chunk_bytes[k] = n;
        sl.msg_bytes[k] = op.bytes;
        sl.posted.store(p + 1, std::memory_order_release);
        op.done += n;
        if (op.done >= op.bytes) op.finished = true, --left;
        progressed = true;
      } else {
        Slot& sl = c->slot(op.peer, c->rank);
        const uint64_t t = sl.taken.load(std::memory_order_relaxed);
        if (sl.posted.load(std::memory_order_acquire) <= t) continue;
        const int k = (int)(t & 1);
        const size_t n = sl.chunk_bytes[k], total = sl.msg_bytes[k];
        if (total != op.bytes) {
          // RCCL would corrupt or hang on a size mismatch between a send and its receive; the checker says so
          fprintf(stderr, "fake_rccl_xproc: rank %d expects %zu bytes from rank %d, which sent %zu\n", c->rank, op.bytes, op.peer,
                  total);
          c->shm->aborted.store(1, std::memory_order_release);
          rc = 4;
          break;
        }
        const size_t room = op.bytes - op.done, m = std::min(n, room);
        if (m) {
          // on the receiver's stream and waited for: the slot is recycled by the ack below, and the stream's next
          // kernels (merge / rebase) read what landed
          if (hipMemcpyAsync(op.ptr + op.done, c->chunk(op.peer, c->rank, k), m, hipMemcpyHostToDevice, op.stream) != hipSuccess ||
              hipStreamSynchronize(op.stream) != hipSuccess) {
            rc = 1;
            break;
          }
        }
        sl.taken.store(t + 1, std::memory_order_release);
        op.done += n;
        if (op.done >= total) op.finished = true, --left;
        progressed = true;
      }
    }
    if (!progressed && left && rc == 0) {
      XComm* c = ops[0].first;
      if (c->shm->aborted.load(std::memory_order_acquire)) {
        fail("a peer aborted the communicator during a grouped exchange");
        rc = 2;
      } else if (now_s() - t0 > lim) {
        c->shm->aborted.store(1, std::memory_order_release);
        fail("timed out inside a grouped exchange (a send without its receive?)");
        rc = 2;
      } else {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
    }
  }
  ops.clear();
  return rc;
}

}  // namespace

#define FAKE_API extern "C" __attribute__((visibility("default")))

FAKE_API ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> counter{0};
  const char* dir = getenv("AH_FAKE_RCCL_DIR");
  if (!dir || !dir[0]) dir = "/tmp";
  memset(id, 0, sizeof *id);
  const long long ns = (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const int n = snprintf(id->internal, sizeof id->internal, "%s/ah_fake_rccl_%d_%d_%llx", dir, (int)getpid(), counter.fetch_add(1), ns);
  if (n <= 0 || n >= (int)sizeof id->internal) return 4;
  const int fd = open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return 2;
  if (ftruncate(fd, (off_t)MAP_BYTES) != 0) {  // sparse: zero-filled, pages appear when touched
    close(fd);
    unlink(id->internal);
    return 2;
  }
  void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  static_cast<Shm*>(p)->magic.store(MAGIC, std::memory_order_release);
  munmap(p, sizeof(Shm));
  return 0;
}

FAKE_API ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_WORLD || rank < 0 || rank >= nranks) return 4;
  std::string path(id.internal, strnlen(id.internal, sizeof id.internal));
  const int fd = open(path.c_str(), O_RDWR);
  if (fd < 0) {
    fprintf(stderr, "fake_rccl_xproc: rank %d cannot open %s\n", rank, path.c_str());
    return 2;
  }
  void* p = mmap(nullptr, MAP_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  XComm* c = new XComm{static_cast<Shm*>(p), static_cast<char*>(p) + DATA_OFF, rank, nranks, path};
  if (c->shm->magic.load(std::memory_order_acquire) != MAGIC) {
    munmap(p, MAP_BYTES);
    delete c;
    return 4;
  }
  int expect = 0;
  c->shm->world.compare_exchange_strong(expect, nranks);
  if (c->shm->world.load() != nranks) {
    munmap(p, MAP_BYTES);
    delete c;
    return 4;
  }
  c->shm->joined.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_for(c, [&] { return c->shm->joined.load(std::memory_order_acquire) >= nranks; }, "ncclCommInitRank")) {
    munmap(p, MAP_BYTES);
    delete c;
    return 2;
  }
  *comm = c;
  return 0;
}

static void leave(XComm* c) {
  if (c->shm->left.fetch_add(1, std::memory_order_acq_rel) + 1 >= c->world) unlink(c->path.c_str());
  munmap(c->shm, MAP_BYTES);
  delete c;
}

FAKE_API ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (comm) leave(comm);
  return 0;
}

FAKE_API ncclResult_t ncclCommAbort(ncclComm_t comm) {
  if (!comm) return 0;
  comm->shm->aborted.store(1, std::memory_order_release);
  unlink(comm->path.c_str());
  leave(comm);
  return 0;
}

FAKE_API ncclResult_t ncclGroupStart() {
  ++t_group_depth;
  return 0;
}

FAKE_API ncclResult_t ncclGroupEnd() {
  if (--t_group_depth == 0) return run_ops(t_pending);
  return 0;
}

FAKE_API ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  if (!comm || peer < 0 || peer >= comm->world || peer == comm->rank) return 4;
  t_pending.push_back({comm, PendingOp{true, (char*)const_cast<void*>(buf), count * type_size(t), peer, stream}});
  return t_group_depth == 0 ? run_ops(t_pending) : 0;
}

FAKE_API ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  if (!comm || peer < 0 || peer >= comm->world || peer == comm->rank) return 4;
  t_pending.push_back({comm, PendingOp{false, (char*)buf, count * type_size(t), peer, stream}});
  return t_group_depth == 0 ? run_ops(t_pending) : 0;
}

FAKE_API ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t stream) {
  const size_t bytes = count * type_size(t);
  if (!c || bytes > COLL_BYTES - 8) return 4;
  if (const char* h = getenv("AH_FAKE_RCCL_HANG")) {  // fault injection: a collective that never returns (bench.py's probe must survive it)
    if (h[0] == '1')
      for (int i = 0; i < 36000; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (bytes && hipMemcpy(c->shm->coll[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  // the send count must be the same on every rank (undefined on real RCCL otherwise): the checker checks
  uint64_t* tag = reinterpret_cast<uint64_t*>(c->shm->coll[c->rank] + COLL_BYTES - 8);
  *tag = bytes;
  if (!barrier(c, "ncclAllGather (arrive)")) return 2;
  for (int r = 0; r < c->world; ++r) {
    const uint64_t theirs = *reinterpret_cast<uint64_t*>(c->shm->coll[r] + COLL_BYTES - 8);
    if (theirs != bytes) {
      fprintf(stderr, "fake_rccl_xproc: ncclAllGather send counts differ: rank %d sends %zu bytes, rank %d sends %llu\n", c->rank, bytes,
              r, (unsigned long long)theirs);
      c->shm->aborted.store(1, std::memory_order_release);
      return 4;
    }
    if (bytes && hipMemcpyAsync((char*)recv + (size_t)r * bytes, c->shm->coll[r], bytes, hipMemcpyHostToDevice, stream) != hipSuccess)
      return 1;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  // nobody may start the next collective (and overwrite its area) before everyone has read this one
  return barrier(c, "ncclAllGather (leave)") ? 0 : 2;
}

FAKE_API ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                                    hipStream_t stream) {
  if (!c || t != 8 || op != 2 || count * 8 > COLL_BYTES - 8) return 4;  // only what comm.hip uses: max over doubles
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (count && hipMemcpy(c->shm->coll[c->rank], send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (!barrier(c, "ncclAllReduce (arrive)")) return 2;
  std::vector<double> out(count);
  for (size_t i = 0; i < count; ++i) {
    double m = reinterpret_cast<double*>(c->shm->coll[0])[i];
    for (int r = 1; r < c->world; ++r) {
      const double v = reinterpret_cast<double*>(c->shm->coll[r])[i];
      m = v > m ? v : m;
    }
    out[i] = m;
  }
  if (count && (hipMemcpyAsync(recv, out.data(), count * 8, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess))
    return 1;
  return barrier(c, "ncclAllReduce (leave)") ? 0 : 2;
}

FAKE_API ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t* async_error) {
  *async_error = (c && c->shm->aborted.load(std::memory_order_acquire)) ? 2 : 0;
  return 0;
}

FAKE_API const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case 0: return "no error";
    case 1: return "fake rccl: HIP failure";
    case 2: return "fake rccl: system error (peer aborted / timed out)";
    case 4: return "fake rccl: invalid argument (mismatched collective)";
    default: return "fake rccl error";
  }
}
