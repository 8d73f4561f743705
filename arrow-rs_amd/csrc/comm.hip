// comm.hip — the multi-GPU exchange step behind the C ABI: RCCL over xGMI, one process per GPU.
//
// The reference has no parallelism of any kind (its kernels are synchronous pure functions).  The north star
// shards a RecordBatch by contiguous row range over the GPUs of a node (RecordBatch::slice semantics,
// arrow-array/src/record_batch.rs:681); every GPU runs the single-GPU kernels on its range and the global
// result is the in-order concatenation of the shard results — arrow_select::concat
// (arrow-select/src/concat.rs:334-343 primitives, :495 entry), bitmaps merged like bit_mask::set_bits
// (arrow-buffer/src/util/bit_mask.rs:33).  This file is that concatenation ACROSS ranks:
//
//   1. counts      one ncclAllGather of {len, null_count, has_validity} per column; the host reads them back
//                  through the pinned mailbox (its single wait before the exchange, none in between)
//   2. exchange    ONE ncclGroup of point-to-point sends / receives: every peer's value bytes land DIRECTLY at
//                  their final offset of the output buffer, packed validity words in a staging area.  xGMI is a
//                  point-to-point fabric (7 links per GPU): a direct exchange drives every link at once, a ring
//                  all-gather would be bound by one link.  RCCL has no all-gatherv.
//   3. merge       ONE kernel writes every output validity word from the (generally not byte-aligned) pieces
//                  that overlap it — no per-rank launches, no pre-zeroing, no host round trip.
//
// RCCL is bound at run time (dlopen: `librccl.so.1`, reusing a copy the process already loaded — e.g. torch's —
// else ROCm's), so libarrow_hip.so has no link-time dependency and single-GPU hosts never touch it.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <vector>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // AH_RCCL_LIBRARY: an explicit library path (a site's own RCCL build; tests/cpp/fake_rccl.cpp in the test suite)
    const char* names[] = {getenv("AH_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n || !n[0]) continue;
      api.handle = dlopen(n, RTLD_NOW | (n == names[0] ? RTLD_LOCAL : RTLD_GLOBAL));
      if (api.handle) break;
    }
    if (!api.handle) {
      api.error = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?");
      return;
    }
    bool ok = true;
    auto bind = [&](auto& fn, const char* sym) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, sym));
      if (!fn) {
        ok = false;
        api.error = std::string("RCCL symbol missing: ") + sym;
      }
    };
    bind(api.GetUniqueId, "ncclGetUniqueId");
    bind(api.CommInitRank, "ncclCommInitRank");
    bind(api.CommDestroy, "ncclCommDestroy");
    bind(api.AllGather, "ncclAllGather");
    bind(api.AllReduce, "ncclAllReduce");
    bind(api.Send, "ncclSend");
    bind(api.Recv, "ncclRecv");
    bind(api.GroupStart, "ncclGroupStart");
    bind(api.GroupEnd, "ncclGroupEnd");
    bind(api.CommGetAsyncError, "ncclCommGetAsyncError");
    bind(api.GetErrorString, "ncclGetErrorString");
    if (!ok) {
      dlclose(api.handle);
      api.handle = nullptr;
    }
  });
  return api.handle ? &api : nullptr;
}

}  // namespace

struct ah_comm {
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;  // nullptr: world == 1 without RCCL (local copies only)
};

#define AH_NCCL(ctx, expr)                                                                              \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess)                                                                              \
      return ah_fail((ctx), AH_COMM_ERROR, "RCCL error %s at %s:%d (%s)", rccl()->GetErrorString(_r), __FILE__, \
                     __LINE__, #expr);                                                                  \
  } while (0)

namespace {

// SURVEY §5 "failure detection": a collective that died asynchronously (peer gone, link error) shows up here
ah_status check_async(ah_context* ctx, ah_comm* comm) {
  if (!comm->nccl) return AH_OK;
  ncclResult_t async = ncclSuccess;
  AH_NCCL(ctx, rccl()->CommGetAsyncError(comm->nccl, &async));
  if (async != ncclSuccess && async != ncclInProgress)
    return ah_fail(ctx, AH_COMM_ERROR, "RCCL asynchronous error: %s", rccl()->GetErrorString(async));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ah_fail(ctx, AH_HIP_ERROR, "HIP error %s after the collective", hipGetErrorString(e));
  return AH_OK;
}

// a few host words -> device memory through the kernel-argument segment (no copy engine, no pinned staging)
struct Words16 { unsigned long long w[48]; };
__global__ void __launch_bounds__(64) store_words16_kernel(unsigned long long* dst, int n, Words16 v) {
  for (int i = threadIdx.x; i < n; i += 64) dst[i] = v.w[i];
}

// one piece of a concatenated bitmap: `len` bits packed from bit 0 of `words`, destined for rows [row0, row0 + len)
struct Piece {
  const unsigned long long* words;
  long long row0, len;
};

// Every output word is assembled from the pieces that overlap it (funnel shifts); pieces are ordered by row0 and
// contiguous.  One thread per output word, no atomics, no pre-zeroed destination.
__global__ void __launch_bounds__(256) merge_pieces_kernel(unsigned long long* out, long long total_rows, int npieces,
                                                           const Piece* pieces) {
  const long long nwords = (total_rows + 63) >> 6;
  for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (long long)gridDim.x * 256) {
    const long long r0 = w << 6, r1 = r0 + 64;
    unsigned long long acc = 0;
    for (int p = 0; p < npieces; ++p) {
      const Piece pc = pieces[p];
      if (pc.len <= 0 || pc.row0 + pc.len <= r0) continue;
      if (pc.row0 >= r1) break;
      const long long s = r0 - pc.row0;  // index inside the piece of this word's bit 0 (may be negative)
      const BitView bv{(const uint64_t*)pc.words, 0};
      acc |= s >= 0 ? bv_fetch64(bv, s, pc.len) : (bv_fetch64(bv, 0, pc.len) << (-s));
    }
    out[w] = acc;
  }
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" ah_status ah_comm_unique_id(ah_context* ctx, uint8_t* id) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !id) return AH_INVALID_ARGUMENT;
  RcclApi* api = rccl();
  if (!api) return ah_fail(ctx, AH_COMM_ERROR, "RCCL is not available in this process (dlopen librccl.so.1 failed)");
  static_assert(sizeof(ncclUniqueId) == AH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId uid;
  AH_NCCL(ctx, api->GetUniqueId(&uid));
  memcpy(id, &uid, sizeof uid);
  return AH_OK;
}

extern "C" ah_status ah_comm_create(ah_context* ctx, int32_t rank, int32_t world, const uint8_t* id, ah_comm** out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world) return AH_INVALID_ARGUMENT;
  *out = nullptr;
  if (!id && world > 1) return ah_fail(ctx, AH_INVALID_ARGUMENT, "a communicator of %d ranks needs the unique id of rank 0", world);
  hipSetDevice(ctx->device);
  ah_comm* c = new ah_comm();
  c->rank = rank;
  c->world = world;
  if (id) {
    RcclApi* api = rccl();
    if (!api) {
      delete c;
      return ah_fail(ctx, AH_COMM_ERROR, "RCCL is not available in this process");
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = api->CommInitRank(&c->nccl, world, uid, rank);
    if (r != ncclSuccess) {
      delete c;
      return ah_fail(ctx, AH_COMM_ERROR, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api->GetErrorString(r));
    }
  }
  *out = c;
  return AH_OK;
}

extern "C" void ah_comm_destroy(ah_context* ctx, ah_comm* comm) {
  ah_ctx_guard _guard(ctx);
  if (!comm) return;
  if (comm->nccl) {
    if (ctx) hipStreamSynchronize(ctx->stream);
    rccl()->CommDestroy(comm->nccl);
  }
  delete comm;
}

extern "C" int32_t ah_comm_rank(const ah_comm* comm) { return comm ? comm->rank : 0; }
extern "C" int32_t ah_comm_world(const ah_comm* comm) { return comm ? comm->world : 1; }

extern "C" ah_status ah_comm_allreduce_max_f64(ah_context* ctx, ah_comm* comm, double* values, int32_t n) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !comm || (n > 0 && !values) || n < 0 || n > 32) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  if (!comm->nccl || n == 0) {
    AH_HIP(ctx, ah_stream_wait(ctx));
    return AH_OK;
  }
  unsigned long long* d = nullptr;
  AH_TRY(ah_pool_alloc(ctx, 64 * 8, (void**)&d));
  Words16 w{};
  memcpy(w.w, values, (size_t)n * 8);
  store_words16_kernel<<<1, 64, 0, ctx->stream>>>(d, n, w);
  ncclResult_t r = rccl()->AllReduce(d, d, (size_t)n, ncclDouble, ncclMax, comm->nccl, ctx->stream);
  hipError_t e = r == ncclSuccess ? ah_d2h_wait(ctx, ctx->pinned + 32, d, (size_t)n * 8) : hipSuccess;
  ah_pool_free(ctx, d);
  AH_NCCL(ctx, r);
  AH_HIP(ctx, e);
  memcpy(values, ctx->pinned + 32, (size_t)n * 8);
  return check_async(ctx, comm);
}

extern "C" ah_status ah_comm_barrier(ah_context* ctx, ah_comm* comm) {
  ah_ctx_guard _guard(ctx);
  double one = 1.0;
  return ah_comm_allreduce_max_f64(ctx, comm, &one, 1);
}

// The device-side primitive of step 3, exported on its own: the concatenation of `n` bit-packed pieces (each
// (ptr, bit_offset, len)) into `dst` (8-byte aligned, ceil(total / 64) words, written in full).
extern "C" ah_status ah_bitmap_concat(ah_context* ctx, int32_t n, const uint8_t* const* pieces, const int64_t* bit_offsets,
                                      const int64_t* lens, uint8_t* dst, int64_t* total_rows) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || n < 0 || (n > 0 && (!pieces || !lens)) || !dst) return AH_INVALID_ARGUMENT;
  if (((uintptr_t)dst & 7) != 0) return ah_fail(ctx, AH_INVALID_ARGUMENT, "bitmap destination must be 8-byte aligned");
  hipSetDevice(ctx->device);
  // pieces with an offset (or an unaligned pointer) are first re-packed to bit 0 of an 8-byte aligned scratch
  std::vector<Piece> host((size_t)std::max(n, 1));
  std::vector<void*> scratch;
  int64_t total = 0;
  ah_status st = AH_OK;
  for (int i = 0; i < n && st == AH_OK; ++i) {
    const int64_t off = bit_offsets ? bit_offsets[i] : 0;
    host[i].row0 = total;
    host[i].len = lens[i];
    host[i].words = nullptr;
    if (lens[i] <= 0) continue;
    if (pieces[i] && off == 0 && ((uintptr_t)pieces[i] & 7) == 0) {
      host[i].words = (const unsigned long long*)pieces[i];
    } else {  // nullptr reads as all ones (a shard without a null buffer)
      void* p = nullptr;
      st = ah_pool_alloc(ctx, ah_bitmap_bytes(lens[i]), &p);
      if (st != AH_OK) break;
      scratch.push_back(p);
      st = ah_bitmap_op(ctx, BM_COPY, make_bitview(pieces[i], off), BitView{nullptr, 0}, BitView{nullptr, 0}, lens[i],
                        (unsigned long long*)p, nullptr);
      host[i].words = (const unsigned long long*)p;
    }
    total += lens[i];
  }
  Piece* dev = nullptr;
  if (st == AH_OK && total > 0) st = ah_pool_alloc(ctx, sizeof(Piece) * (size_t)n, (void**)&dev);
  if (st == AH_OK && total > 0) {
    hipError_t e = hipMemcpyAsync(dev, host.data(), sizeof(Piece) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      const int64_t nwords = (total + 63) >> 6;
      merge_pieces_kernel<<<(unsigned)std::min<int64_t>(4096, ah_ceil_div(nwords, 256)), 256, 0, ctx->stream>>>(
          (unsigned long long*)dst, total, n, dev);
      e = ah_stream_wait(ctx);  // `host` must outlive the H2D copy; scratch goes back to the pool after the kernel
    }
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "bitmap concat failed: %s", hipGetErrorString(e));
  }
  ah_pool_free(ctx, dev);
  for (void* p : scratch) ah_pool_free(ctx, p);
  if (total_rows) *total_rows = total;
  return st;
}

// concat of every rank's columns, in rank order, materialised on every rank (steps 1-3 above)
extern "C" ah_status ah_all_gather_columns(ah_context* ctx, ah_comm* comm, int32_t n_columns, const ah_array_view* columns,
                                           ah_array_out* outs, ah_exchange_stats* stats) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !comm || n_columns < 1 || n_columns > 16 || !columns || !outs) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  const double t0 = now_ms();
  for (int c = 0; c < n_columns; ++c) ah_out_init(&outs[c]);
  const int R = comm->world, me = comm->rank;
  RcclApi* api = comm->nccl ? rccl() : nullptr;
  std::vector<int> width(n_columns);
  for (int c = 0; c < n_columns; ++c) {
    width[c] = ah_type_width(columns[c].type);
    if (width[c] <= 0)  // Boolean values and strings: bit / offset re-basing across ranks is not wired up
      return ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "all-gather of %s columns", ah_type_name(columns[c].type));
  }
  // ---- 1. counts: {len, null_count, has_validity} per column from every rank
  const int per = 3 * n_columns;
  std::vector<unsigned long long> counts((size_t)per * R);
  Words16 mine{};
  for (int c = 0; c < n_columns; ++c) {
    int64_t nulls = 0;
    AH_TRY(ah_resolve_null_count(ctx, &columns[c], &nulls));
    mine.w[3 * c] = (unsigned long long)columns[c].length;
    mine.w[3 * c + 1] = (unsigned long long)nulls;
    mine.w[3 * c + 2] = (columns[c].validity && nulls > 0) ? 1ull : 0ull;
  }
  if (api) {
    unsigned long long* dcounts = nullptr;
    AH_TRY(ah_pool_alloc(ctx, (size_t)(per * (R + 1)) * 8, (void**)&dcounts));
    store_words16_kernel<<<1, 64, 0, ctx->stream>>>(dcounts, per, mine);
    ncclResult_t r = api->AllGather(dcounts, dcounts + per, (size_t)per, ncclUint64, comm->nccl, ctx->stream);
    hipError_t e = hipSuccess;
    if (r == ncclSuccess) {
      if (per * R <= 200) {
        e = ah_d2h_wait(ctx, ctx->pinned + 16, dcounts + per, (size_t)per * R * 8);
        if (e == hipSuccess) memcpy(counts.data(), ctx->pinned + 16, (size_t)per * R * 8);
      } else {
        e = hipMemcpyAsync(counts.data(), dcounts + per, (size_t)per * R * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
    }
    ah_pool_free(ctx, dcounts);
    AH_NCCL(ctx, r);
    AH_HIP(ctx, e);
    AH_TRY(check_async(ctx, comm));
  } else {
    memcpy(counts.data(), mine.w, (size_t)per * 8);
  }
  const double t_counts = now_ms();
  auto cnt = [&](int r, int c, int k) { return (int64_t)counts[(size_t)r * per + 3 * c + k]; };

  // ---- 2. buffers + ONE grouped exchange
  struct ColPlan {
    int64_t total = 0, nulls = 0;
    bool any_valid = false;
    std::vector<int64_t> row0, bit_off;  // per rank: first output row, byte offset of its packed bits in staging
    void* ov = nullptr;
    void* ob = nullptr;
    unsigned long long* staging = nullptr;
    Piece* dev_pieces = nullptr;
    size_t vbytes = 0, bbytes = 0;
  };
  std::vector<ColPlan> plan(n_columns);
  std::vector<std::vector<Piece>> host_pieces(n_columns);
  ah_status st = AH_OK;
  auto cleanup = [&](ah_status s) {
    for (int c = 0; c < n_columns; ++c) {
      ah_pool_free(ctx, plan[c].staging);
      ah_pool_free(ctx, plan[c].dev_pieces);
      if (s != AH_OK) {
        ah_out_free(ctx, plan[c].ov, plan[c].vbytes);
        ah_out_free(ctx, plan[c].ob, plan[c].bbytes);
        ah_out_init(&outs[c]);
      }
    }
    return s;
  };
  int64_t sent_each = 0, received = 0;
  for (int c = 0; c < n_columns && st == AH_OK; ++c) {
    ColPlan& p = plan[c];
    p.row0.resize(R);
    p.bit_off.resize(R);
    int64_t bits_total = 0;
    for (int r = 0; r < R; ++r) {
      p.row0[r] = p.total;
      p.bit_off[r] = bits_total;
      p.total += cnt(r, c, 0);
      p.nulls += cnt(r, c, 1);
      p.any_valid = p.any_valid || cnt(r, c, 2) != 0;
      bits_total += (int64_t)ah_bitmap_bytes(cnt(r, c, 0));
    }
    if (cnt(me, c, 0) != columns[c].length) return cleanup(ah_fail(ctx, AH_COMM_ERROR, "count exchange returned a foreign length"));
    p.vbytes = (size_t)p.total * width[c];
    st = ah_out_alloc(ctx, p.vbytes, &p.ov);
    if (st == AH_OK && p.any_valid) {
      p.bbytes = ah_bitmap_bytes(p.total);
      st = ah_out_alloc(ctx, p.bbytes, &p.ob);
      if (st == AH_OK) st = ah_pool_alloc(ctx, std::max<size_t>((size_t)bits_total, 8), (void**)&p.staging);
      if (st == AH_OK) st = ah_pool_alloc(ctx, sizeof(Piece) * (size_t)R, (void**)&p.dev_pieces);
      // this rank's validity packed at bit 0, zero padded to whole words, straight into its staging slot
      if (st == AH_OK && columns[c].length > 0) {
        const bool has_v = cnt(me, c, 2) != 0;
        st = ah_bitmap_op(ctx, BM_COPY, has_v ? make_bitview(columns[c].validity, columns[c].validity_bit_offset) : BitView{nullptr, 0},
                          BitView{nullptr, 0}, BitView{nullptr, 0}, columns[c].length,
                          (unsigned long long*)((char*)p.staging + p.bit_off[me]), nullptr);
      }
    }
  }
  if (st != AH_OK) return cleanup(st);

  ncclResult_t gr = api ? api->GroupStart() : ncclSuccess;
  if (gr != ncclSuccess) return cleanup(ah_fail(ctx, AH_COMM_ERROR, "ncclGroupStart failed: %s", api->GetErrorString(gr)));
  for (int c = 0; c < n_columns; ++c) {
    ColPlan& p = plan[c];
    const size_t my_vbytes = (size_t)columns[c].length * width[c];
    if (my_vbytes)  // own piece: a device-to-device copy to its final place
      hipMemcpyAsync((char*)p.ov + (size_t)p.row0[me] * width[c], columns[c].values, my_vbytes, hipMemcpyDeviceToDevice, ctx->stream);
    for (int step = 1; step < R && gr == ncclSuccess; ++step) {  // staggered peers: every link busy in both directions
      const int dst = (me + step) % R, src = (me - step + R) % R;
      if (my_vbytes) gr = api->Send(columns[c].values, my_vbytes, ncclChar, dst, comm->nccl, ctx->stream);
      const size_t src_vbytes = (size_t)cnt(src, c, 0) * width[c];
      if (gr == ncclSuccess && src_vbytes)
        gr = api->Recv((char*)p.ov + (size_t)p.row0[src] * width[c], src_vbytes, ncclChar, src, comm->nccl, ctx->stream);
      if (p.any_valid) {
        const size_t my_b = ah_bitmap_bytes(columns[c].length), src_b = ah_bitmap_bytes(cnt(src, c, 0));
        if (gr == ncclSuccess && columns[c].length > 0)
          gr = api->Send((char*)p.staging + p.bit_off[me], my_b, ncclChar, dst, comm->nccl, ctx->stream);
        if (gr == ncclSuccess && cnt(src, c, 0) > 0)
          gr = api->Recv((char*)p.staging + p.bit_off[src], src_b, ncclChar, src, comm->nccl, ctx->stream);
        sent_each += (int64_t)my_b;
        received += (int64_t)src_b;
      }
      sent_each += (int64_t)my_vbytes;
      received += (int64_t)src_vbytes;
    }
  }
  if (api) {
    ncclResult_t ge = api->GroupEnd();
    if (gr == ncclSuccess) gr = ge;
    if (gr != ncclSuccess) return cleanup(ah_fail(ctx, AH_COMM_ERROR, "RCCL exchange failed: %s", api->GetErrorString(gr)));
  }
  if (R > 1) sent_each /= (R - 1);

  // ---- 3. one merge kernel per column that carries validity
  for (int c = 0; c < n_columns && st == AH_OK; ++c) {
    ColPlan& p = plan[c];
    if (!p.any_valid || p.total == 0) continue;
    host_pieces[c].resize(R);
    for (int r = 0; r < R; ++r)
      host_pieces[c][r] = Piece{(const unsigned long long*)((char*)p.staging + p.bit_off[r]), p.row0[r], cnt(r, c, 0)};
    hipError_t e = hipMemcpyAsync(p.dev_pieces, host_pieces[c].data(), sizeof(Piece) * (size_t)R, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
      st = ah_fail(ctx, AH_HIP_ERROR, "piece table upload failed: %s", hipGetErrorString(e));
      break;
    }
    const int64_t nwords = (p.total + 63) >> 6;
    merge_pieces_kernel<<<(unsigned)std::min<int64_t>(4096, ah_ceil_div(nwords, 256)), 256, 0, ctx->stream>>>(
        (unsigned long long*)p.ob, p.total, R, p.dev_pieces);
  }
  if (st == AH_OK) {
    hipError_t e = ah_stream_wait(ctx);  // results usable at return (synchronous-by-default contract)
    if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "all-gather failed: %s", hipGetErrorString(e));
  }
  if (st == AH_OK) st = check_async(ctx, comm);
  if (st != AH_OK) return cleanup(st);
  for (int c = 0; c < n_columns; ++c) {
    ColPlan& p = plan[c];
    outs[c].type = columns[c].type;
    outs[c].length = p.total;
    outs[c].values = p.ov;
    outs[c].values_bytes = (int64_t)p.vbytes;
    if (p.any_valid) {
      outs[c].validity = (uint8_t*)p.ob;
      outs[c].validity_bytes = (int64_t)p.bbytes;
      outs[c].null_count = p.nulls;
    }
  }
  if (stats) {
    stats->peers = R - 1;
    stats->bytes_to_each_peer = sent_each;
    stats->bytes_received = received;
    stats->counts_ms = t_counts - t0;
    stats->total_ms = now_ms() - t0;
  }
  return cleanup(AH_OK);
}

extern "C" ah_status ah_all_gatherv(ah_context* ctx, ah_comm* comm, const ah_array_view* local, ah_array_out* out,
                                    ah_exchange_stats* stats) {
  ah_ctx_guard _guard(ctx);
  return ah_all_gather_columns(ctx, comm, 1, local, out, stats);
}
