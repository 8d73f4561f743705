// fake_rccl.cpp — TEST INFRASTRUCTURE: an in-process stand-in for the eleven RCCL entry points comm.hip binds, so that
// the REAL exchange code of libarrow_hip.so (ah_all_gatherv / ah_all_gather_columns: count all-gather, one group of
// sends and receives into final offsets, merge kernel) can run at world sizes 2..8 on the ONE GPU a test box has.
// RCCL itself refuses two ranks on one device; here every "rank" is a host thread with its own ah_context, and a
// send / recv pair is a device-to-device copy matched through a process-wide table.  Loaded through
// AH_RCCL_LIBRARY=<path> (comm.hip); nothing in the product links or ships it.
//
// Semantics kept from NCCL: operations between ncclGroupStart / ncclGroupEnd are issued together at GroupEnd;
// send(to p) pairs with recv(from me) at p in program order per (src, dst) pair; all-gather / all-reduce are
// collective over every rank of the communicator.  Everything is made synchronous (the calling thread blocks until
// its part is done): slow and simple, which is what a checker wants.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef struct FakeComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;  // 0 = ncclSuccess
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}

namespace {

struct Msg {
  const void* src;
  size_t bytes;
  bool consumed = false;
};

struct World {
  int size = 0;
  int joined = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Msg*>> box;  // (src, dst) -> messages in program order
  // collectives: one generation at a time
  int coll_arrived = 0, coll_gen = 0;
  std::vector<const void*> coll_ptr;
  std::vector<std::vector<char>> coll_host;
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
int g_next_id = 1;

struct PendingOp {
  bool is_send;
  void* ptr;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case 0: case 1: return 1;          // int8 / char, uint8
    case 2: case 3: case 7: return 4;  // int32, uint32, float32
    case 4: case 5: case 8: return 8;  // int64, uint64, float64
    case 6: case 9: return 2;          // float16, bfloat16
    default: return 1;
  }
}

thread_local int t_group_depth = 0;
thread_local std::vector<std::pair<struct FakeComm*, PendingOp>> t_pending;

}  // namespace

struct FakeComm {
  World* w;
  int rank;
};

namespace {

void run_ops(std::vector<std::pair<FakeComm*, PendingOp>>& ops) {
  // 1. make this thread's earlier stream work visible, post every send
  std::vector<std::pair<World*, Msg*>> mine;
  for (auto& e : ops)
    if (e.second.is_send) hipStreamSynchronize(e.second.stream);
  for (auto& e : ops) {
    if (!e.second.is_send) continue;
    World* w = e.first->w;
    Msg* m = new Msg{e.second.ptr, e.second.bytes};
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->box[{e.first->rank, e.second.peer}].push_back(m);
    }
    w->cv.notify_all();
    mine.push_back({w, m});
  }
  // 2. receives: wait for the matching send, copy device to device, mark it consumed
  for (auto& e : ops) {
    if (e.second.is_send) continue;
    World* w = e.first->w;
    Msg* m = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      auto key = std::make_pair(e.second.peer, e.first->rank);
      w->cv.wait(lk, [&] { return !w->box[key].empty(); });
      m = w->box[key].front();
      w->box[key].pop_front();
    }
    const size_t n = m->bytes < e.second.bytes ? m->bytes : e.second.bytes;
    // on the RECEIVER'S stream and waited for: a plain hipMemcpy of device memory may return before the copy has run,
    // and it runs on the null stream, which the contexts' non-blocking streams do not order against — the receiver's
    // merge / rebase kernels could then read the piece before it had landed (seen once at world 8: a stale offset)
    if (n) hipMemcpyAsync(e.second.ptr, m->src, n, hipMemcpyDeviceToDevice, e.second.stream);
    hipStreamSynchronize(e.second.stream);
    {
      std::lock_guard<std::mutex> lk(w->mu);
      m->consumed = true;
    }
    w->cv.notify_all();
  }
  // 3. a send buffer may be reused once its receiver has copied it
  for (auto& pm : mine) {
    std::unique_lock<std::mutex> lk(pm.first->mu);
    pm.first->cv.wait(lk, [&] { return pm.second->consumed; });
    delete pm.second;
  }
  ops.clear();
}

}  // namespace

// snapshot of the last all-gather, per world
std::map<World*, std::vector<std::vector<char>>>& fake_snapshots() {
  static std::map<World*, std::vector<std::vector<char>>> s;
  return s;
}

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "fake-rccl-%d", g_next_id++);
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  World* w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    auto it = g_worlds.find(key);
    if (it == g_worlds.end()) {
      w = new World();
      w->size = nranks;
      g_worlds[key] = w;
    } else {
      w = it->second;
    }
  }
  if (w->size != nranks || rank < 0 || rank >= nranks) return 4;  // ncclInvalidArgument
  {
    std::unique_lock<std::mutex> lk(w->mu);
    w->joined++;
    w->cv.notify_all();
    w->cv.wait(lk, [&] { return w->joined >= w->size; });  // collective, like the real one
  }
  *comm = new FakeComm{w, rank};
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() {
  ++t_group_depth;
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() {
  if (--t_group_depth == 0) run_ops(t_pending);
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm,
                                                            hipStream_t stream) {
  t_pending.push_back({comm, PendingOp{true, const_cast<void*>(buf), count * type_size(t), peer, stream}});
  if (t_group_depth == 0) run_ops(t_pending);
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm,
                                                            hipStream_t stream) {
  t_pending.push_back({comm, PendingOp{false, buf, count * type_size(t), peer, stream}});
  if (t_group_depth == 0) run_ops(t_pending);
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t,
                                                                 ncclComm_t comm, hipStream_t stream) {
  hipStreamSynchronize(stream);
  const size_t bytes = count * type_size(t);
  World* w = comm->w;
  std::vector<std::vector<char>> all;
  {
    // deposit + wait for everyone (see collective_exchange: the last arriver publishes the snapshot)
    std::vector<char> mine(bytes);
    if (bytes) {
      hipMemcpyAsync(mine.data(), send, bytes, hipMemcpyDeviceToHost, stream);
      hipStreamSynchronize(stream);
    }
    std::unique_lock<std::mutex> lk(w->mu);
    const int gen = w->coll_gen;
    if (w->coll_host.size() != (size_t)w->size) w->coll_host.assign(w->size, {});
    w->coll_host[comm->rank] = std::move(mine);
    if (++w->coll_arrived == w->size) {
      w->coll_arrived = 0;
      fake_snapshots()[w] = w->coll_host;
      w->coll_host.assign(w->size, {});
      w->coll_gen++;
      w->cv.notify_all();
    } else {
      w->cv.wait(lk, [&] { return w->coll_gen != gen; });
    }
    all = fake_snapshots()[w];
  }
  // the send count must be the same on every rank — undefined (a hang or corruption) on real RCCL otherwise; every rank
  // sees the same snapshot, so every rank returns the error
  for (int r = 0; r < w->size; ++r)
    if (all[r].size() != bytes) {
      fprintf(stderr, "fake_rccl: ncclAllGather send counts differ: rank %d sends %zu bytes, rank %d sends %zu\n", comm->rank, bytes, r,
              all[r].size());
      return 4;
    }
  for (int r = 0; r < w->size; ++r)
    if (bytes) hipMemcpyAsync((char*)recv + (size_t)r * bytes, all[r].data(), bytes, hipMemcpyHostToDevice, stream);
  hipStreamSynchronize(stream);  // `all` is pageable: the copies have left it; and the stream's next kernel sees them
  // nobody may start the next collective (and overwrite the snapshot) before everyone has read this one
  {
    std::unique_lock<std::mutex> lk(w->mu);
    const int gen = w->coll_gen;
    if (++w->coll_arrived == w->size) {
      w->coll_arrived = 0;
      w->coll_gen++;
      w->cv.notify_all();
    } else {
      w->cv.wait(lk, [&] { return w->coll_gen != gen; });
    }
  }
  return 0;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op,
                                                                 ncclComm_t comm, hipStream_t stream) {
  if (t != 8 || op != 2) return 4;  // only what comm.hip uses: max over doubles
  // all-gather into a scratch, reduce on the host
  World* w = comm->w;
  void* scratch = nullptr;
  if (hipMalloc(&scratch, count * 8 * (size_t)w->size) != hipSuccess) return 1;
  ncclResult_t r = ncclAllGather(send, scratch, count, t, comm, stream);
  std::vector<double> all(count * (size_t)w->size), out(count);
  hipMemcpyAsync(all.data(), scratch, all.size() * 8, hipMemcpyDeviceToHost, stream);
  hipStreamSynchronize(stream);
  for (size_t i = 0; i < count; ++i) {
    double m = all[i];
    for (int k = 1; k < w->size; ++k) m = all[(size_t)k * count + i] > m ? all[(size_t)k * count + i] : m;
    out[i] = m;
  }
  hipMemcpyAsync(recv, out.data(), count * 8, hipMemcpyHostToDevice, stream);
  hipStreamSynchronize(stream);
  hipFree(scratch);
  return r;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t* async_error) {
  *async_error = 0;
  return 0;
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "fake rccl error"; }

}  // extern "C"
