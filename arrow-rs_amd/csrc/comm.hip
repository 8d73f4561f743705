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
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
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
    if (ok) api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
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
  bool dead = false;          // aborted after a local failure inside an exchange: every later call is AH_COMM_ERROR
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
  if (comm->dead) return ah_fail(ctx, AH_COMM_ERROR, "the communicator was aborted by an earlier failure");
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

// ------------------------------------------------------------------------------------------------ the exchange
// Every column contributes up to three byte streams per rank:
//   fixed-width   values  -> land DIRECTLY at row0[rank] * width of the output values
//   Boolean       value bits, re-packed to bit 0 and staged -> merged into the output bit buffer (concat_boolean,
//                 arrow-select/src/concat.rs:345)
//   Utf8 / Large  raw offsets (len + 1 of them), staged -> ONE rebase kernel writes the output offsets
//                 (offset - first + byte base of the rank); text bytes [first, last) -> land DIRECTLY at the rank's
//                 byte base of the output data buffer (concat_bytes, concat.rs:355-368; i32 totals past i32::MAX are
//                 OffsetOverflowError like GenericByteBuilder::append_array, generic_bytes_builder.rs:186)
//   validity      re-packed to bit 0 and staged -> merged (bit_mask.rs:33 set_bits)
//
// The counts payload carries, per rank, a STATUS word and, per column, {len, nulls, has_validity, type, data bytes,
// first offset}.  Everything that can fail locally before the exchange (argument checks, null counts, the payload
// buffer) happens BEFORE the count all-gather, and a rank that failed still takes part in it with status != 0: every
// rank then returns an error together instead of leaving its peers blocked in a collective.  Column
// types are compared across ranks for the same reason (a schema mismatch would pair sends and receives of
// different sizes).  A failure between the count exchange and the grouped exchange (out of memory) aborts the
// communicator (ncclCommAbort) so that peers see an asynchronous error rather than a hang.
namespace {

constexpr int EX_MAX_COLS = 16;
constexpr int EX_HDR = 2;  // status, n_columns
constexpr int EX_PER_COL = 6;

struct CountFill {
  unsigned long long w[EX_HDR + EX_PER_COL * EX_MAX_COLS];
  const void* str_offsets[EX_MAX_COLS];  // non-null: a string column with len > 0; the kernel reads first / last
  long long str_len[EX_MAX_COLS];
  int str_wide[EX_MAX_COLS];
};
__global__ void __launch_bounds__(128) fill_counts_kernel(unsigned long long* dst, int nwords, int ncols, CountFill f) {
  const int t = threadIdx.x;
  if (t < nwords) {
    unsigned long long v = f.w[t];
    const int c = (t - EX_HDR) / EX_PER_COL, k = (t - EX_HDR) % EX_PER_COL;
    if (t >= EX_HDR && c < ncols && f.str_offsets[c] && (k == 4 || k == 5)) {
      long long first, last;
      if (f.str_wide[c]) {
        first = ((const long long*)f.str_offsets[c])[0];
        last = ((const long long*)f.str_offsets[c])[f.str_len[c]];
      } else {
        first = ((const int*)f.str_offsets[c])[0];
        last = ((const int*)f.str_offsets[c])[f.str_len[c]];
      }
      v = (unsigned long long)(k == 4 ? last - first : first);
    }
    dst[t] = v;
  }
}

// The count all-gather always moves the FULL payload (EX_HDR + EX_PER_COL * EX_MAX_COLS words per rank): the send
// count of the collective must not depend on what a rank was handed (ranks that disagree on n_columns — or one whose
// arguments failed the check — would otherwise enter ncclAllGather with different counts, which is undefined on real
// RCCL).  The host only needs the header of every rank and the first `ncols` columns of each: this one-wave kernel
// packs exactly those words so that the usual case still comes back through the pinned mailbox.
__global__ void __launch_bounds__(64) compact_counts_kernel(unsigned long long* dst, const unsigned long long* src, int world,
                                                            int per_full, int per) {
  for (int i = threadIdx.x; i < world * per; i += 64) dst[i] = src[(size_t)(i / per) * per_full + (i % per)];
}

// a piece of a concatenated offsets buffer: rank r's raw offsets (len + 1), its first output row and byte base
struct OffPiece {
  const void* src;
  long long row0, len, base;
};
template <typename O>
__global__ void __launch_bounds__(256) rebase_offsets_kernel(O* out, long long total_rows, long long total_bytes, int npieces,
                                                             const OffPiece* pieces) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i <= total_rows; i += (long long)gridDim.x * 256) {
    if (i == total_rows) {
      out[i] = (O)total_bytes;
      continue;
    }
    for (int p = 0; p < npieces; ++p) {
      const OffPiece pc = pieces[p];
      if (i >= pc.row0 && i < pc.row0 + pc.len) {
        const O* s = (const O*)pc.src;
        out[i] = (O)((long long)s[i - pc.row0] - (long long)s[0] + pc.base);
        break;
      }
    }
  }
}

struct BitJob {
  unsigned long long* out;
  long long total_rows;
  int npieces, first_piece;
};
struct BitJobs {
  BitJob j[2 * EX_MAX_COLS];
};
// merge_pieces_kernel for every bitmap of the exchange in ONE launch (blockIdx.y = bitmap)
__global__ void __launch_bounds__(256) merge_jobs_kernel(BitJobs jobs, const Piece* pieces) {
  const BitJob jb = jobs.j[blockIdx.y];
  const Piece* mine = pieces + jb.first_piece;
  const long long nwords = (jb.total_rows + 63) >> 6;
  for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (long long)gridDim.x * 256) {
    const long long r0 = w << 6, r1 = r0 + 64;
    unsigned long long acc = 0;
    for (int p = 0; p < jb.npieces; ++p) {
      const Piece pc = mine[p];
      if (pc.len <= 0 || pc.row0 + pc.len <= r0) continue;
      if (pc.row0 >= r1) break;
      const long long s = r0 - pc.row0;
      const BitView bv{(const uint64_t*)pc.words, 0};
      acc |= s >= 0 ? bv_fetch64(bv, s, pc.len) : (bv_fetch64(bv, 0, pc.len) << (-s));
    }
    jb.out[w] = acc;
  }
}

struct ColPlan {
  ah_type type = 0;
  int width = 0;  // > 0 fixed, 0 Boolean, -1 strings
  int ow = 0;     // offset width of a string column
  int64_t total = 0, nulls = 0, total_bytes = 0;
  bool any_valid = false;
  std::vector<int64_t> row0, len, bit_off, off_off, byte_base, data_bytes;  // per rank
  int64_t my_first = 0;
  void* ov = nullptr;       // values (fixed) / value bits (Boolean) / text bytes (strings)
  void* oo = nullptr;       // output offsets (strings)
  void* ob = nullptr;       // output validity
  char* vstage = nullptr;   // staged validity words, every rank
  char* bstage = nullptr;   // staged Boolean value words, every rank
  char* ostage = nullptr;   // staged raw offsets, every rank but this one
  size_t vbytes = 0, obytes = 0, bbytes = 0;
};

}  // namespace

struct ah_exchange {
  int n = 0;
  ah_comm* comm = nullptr;
  std::vector<ColPlan> plan;
  std::vector<Piece> pieces;        // tables the merge / rebase kernels read; must outlive their upload
  std::vector<OffPiece> off_pieces;
  Piece* dev_pieces = nullptr;
  OffPiece* dev_off_pieces = nullptr;
  int64_t sent_each = 0, received = 0;
  double t0 = 0, t_counts = 0;
  bool enqueued = false;
};

namespace {

void exchange_free(ah_context* ctx, ah_exchange* x, bool keep_outputs) {
  if (!x) return;
  for (ColPlan& p : x->plan) {
    ah_pool_free(ctx, p.vstage);
    ah_pool_free(ctx, p.bstage);
    ah_pool_free(ctx, p.ostage);
    if (!keep_outputs) {
      ah_out_free(ctx, p.ov, p.vbytes);
      ah_out_free(ctx, p.oo, p.obytes);
      ah_out_free(ctx, p.ob, p.bbytes);
    }
  }
  ah_pool_free(ctx, x->dev_pieces);
  ah_pool_free(ctx, x->dev_off_pieces);
  delete x;
}

// a local failure after the count exchange: peers are (or will be) inside the grouped exchange
ah_status exchange_abort(ah_context* ctx, ah_comm* comm, ah_exchange* x, ah_status st) {
  const std::string msg = ctx->err;  // the frees below may not clobber the message
  if (comm->nccl && comm->world > 1 && rccl()->CommAbort) {
    hipStreamSynchronize(ctx->stream);
    rccl()->CommAbort(comm->nccl);
    comm->nccl = nullptr;
    comm->dead = true;
  }
  exchange_free(ctx, x, false);
  return ah_fail(ctx, st, "%s", msg.c_str());
}

}  // namespace

extern "C" ah_status ah_all_gather_columns_begin(ah_context* ctx, ah_comm* comm, int32_t n_columns, const ah_array_view* columns,
                                                 ah_exchange** handle) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !comm || !handle) return AH_INVALID_ARGUMENT;
  *handle = nullptr;
  if (comm->dead) return ah_fail(ctx, AH_COMM_ERROR, "the communicator was aborted by an earlier failure");
  hipSetDevice(ctx->device);
  const double t0 = now_ms();
  const int R = comm->world, me = comm->rank;
  RcclApi* api = comm->nccl ? rccl() : nullptr;

  // ---- 0. everything that can fail locally, BEFORE the first collective; a failing rank still takes part in it
  ah_status local = AH_OK;
  std::string local_msg;
  auto fail_local = [&](ah_status s) {
    if (local == AH_OK) {
      local = s;
      local_msg = ctx->err;
    }
  };
  if (n_columns < 1 || n_columns > EX_MAX_COLS || !columns) {
    fail_local(ah_fail(ctx, AH_INVALID_ARGUMENT, "all-gather takes 1..%d columns", EX_MAX_COLS));
    n_columns = 0;
  }
  // `per_full` words per rank travel (identical on every rank whatever it was handed); `per` of them are read back
  constexpr int per_full = EX_HDR + EX_PER_COL * EX_MAX_COLS;
  const int per = EX_HDR + EX_PER_COL * n_columns;
  CountFill fill{};
  for (int c = 0; c < n_columns; ++c) {
    const ah_array_view& v = columns[c];
    const bool is_str = v.type == AH_UTF8 || v.type == AH_LARGE_UTF8;
    const int w = is_str ? -1 : ah_type_width(v.type);
    if (!is_str && (w < 0 || v.type == AH_UTF8_VIEW || v.type == AH_BINARY_VIEW)) {
      // views: every rank has its own data-buffer list, which never enters the C ABI (like ah_concat)
      fail_local(ah_fail(ctx, AH_NOT_YET_IMPLEMENTED, "all-gather of %s columns", ah_type_name(v.type)));
      continue;
    }
    if (v.length < 0 || (v.length > 0 && !v.values && !(is_str)) || (is_str && !v.offsets)) {
      fail_local(ah_fail(ctx, AH_INVALID_ARGUMENT, "all-gather: column %d has no buffers", c));
      continue;
    }
    int64_t nulls = 0;
    ah_status s = ah_resolve_null_count(ctx, &v, &nulls);
    if (s != AH_OK) {
      fail_local(s);
      continue;
    }
    unsigned long long* w6 = fill.w + EX_HDR + EX_PER_COL * c;
    w6[0] = (unsigned long long)v.length;
    w6[1] = (unsigned long long)nulls;
    w6[2] = (v.validity && nulls > 0) ? 1ull : 0ull;
    w6[3] = (unsigned long long)v.type;
    if (is_str && v.length > 0) {
      fill.str_offsets[c] = v.offsets;
      fill.str_len[c] = v.length;
      fill.str_wide[c] = v.type == AH_LARGE_UTF8;
    }
  }
  unsigned long long* dcounts = nullptr;
  {
    ah_status s = ah_pool_alloc(ctx, (size_t)(per_full * (R + 1) + per * R) * 8, (void**)&dcounts);
    if (s != AH_OK) fail_local(s);
  }
  if (!api && local != AH_OK) {  // no peers to keep in step
    ah_pool_free(ctx, dcounts);
    return ah_fail(ctx, local, "%s", local_msg.c_str());
  }
  if (!dcounts) {
    // not even the payload buffer: this rank cannot take part in the collective.  Abort the communicator so that
    // peers fail instead of waiting for it.
    if (api && api->CommAbort) {
      api->CommAbort(comm->nccl);
      comm->nccl = nullptr;
      comm->dead = true;
    }
    return ah_fail(ctx, local, "%s", local_msg.c_str());
  }
  fill.w[0] = (unsigned long long)local;
  fill.w[1] = (unsigned long long)n_columns;

  // ---- 1. counts
  std::vector<unsigned long long> counts((size_t)per * R);
  fill_counts_kernel<<<1, 128, 0, ctx->stream>>>(dcounts, per_full, n_columns, fill);
  {
    ncclResult_t r = api ? api->AllGather(dcounts, dcounts + per_full, (size_t)per_full, ncclUint64, comm->nccl, ctx->stream) : ncclSuccess;
    unsigned long long* src = dcounts + (size_t)per_full * (R + 1);
    hipError_t e = hipSuccess;
    if (r == ncclSuccess) {
      compact_counts_kernel<<<1, 64, 0, ctx->stream>>>(src, api ? dcounts + per_full : dcounts, R, per_full, per);
      if (per * R <= 200) {
        e = ah_d2h_wait(ctx, ctx->pinned + 16, src, (size_t)per * R * 8);
        if (e == hipSuccess) memcpy(counts.data(), ctx->pinned + 16, (size_t)per * R * 8);
      } else {
        e = hipMemcpyAsync(counts.data(), src, (size_t)per * R * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
    }
    ah_pool_free(ctx, dcounts);
    AH_NCCL(ctx, r);
    AH_HIP(ctx, e);
    AH_TRY(check_async(ctx, comm));
  }
  const double t_counts = now_ms();
  auto cnt = [&](int r, int c, int k) { return (int64_t)counts[(size_t)r * per + EX_HDR + EX_PER_COL * c + k]; };
  // every rank reads the same table: these failures are common to all of them
  if (local != AH_OK) return ah_fail(ctx, local, "%s", local_msg.c_str());
  for (int r = 0; r < R; ++r)
    if (counts[(size_t)r * per] != 0)
      return ah_fail(ctx, AH_COMM_ERROR, "all-gather: rank %d failed before the exchange (status %lld)", r,
                     (long long)counts[(size_t)r * per]);
  for (int r = 0; r < R; ++r) {
    if ((int64_t)counts[(size_t)r * per + 1] != n_columns)
      return ah_fail(ctx, AH_INVALID_ARGUMENT, "all-gather: rank %d passed %lld columns, rank %d passed %d", r,
                     (long long)counts[(size_t)r * per + 1], me, n_columns);
    for (int c = 0; c < n_columns; ++c)
      if (cnt(r, c, 3) != cnt(0, c, 3))
        return ah_fail(ctx, AH_INVALID_ARGUMENT, "all-gather: column %d is %s on rank %d but %s on rank 0", c,
                       ah_type_name((ah_type)cnt(r, c, 3)), r, ah_type_name((ah_type)cnt(0, c, 3)));
  }
  for (int c = 0; c < n_columns; ++c)
    if (cnt(me, c, 0) != columns[c].length) return ah_fail(ctx, AH_COMM_ERROR, "count exchange returned a foreign length");

  // ---- 2. plan + buffers
  ah_exchange* x = new ah_exchange();
  x->n = n_columns;
  x->comm = comm;
  x->t0 = t0;
  x->t_counts = t_counts;
  x->plan.resize(n_columns);
  for (int c = 0; c < n_columns; ++c) {
    ColPlan& p = x->plan[c];
    p.type = columns[c].type;
    const bool is_str = p.type == AH_UTF8 || p.type == AH_LARGE_UTF8;
    p.width = is_str ? -1 : ah_type_width(p.type);
    p.ow = p.type == AH_UTF8 ? 4 : 8;
    p.row0.resize(R), p.len.resize(R), p.bit_off.resize(R), p.off_off.resize(R), p.byte_base.resize(R), p.data_bytes.resize(R);
    int64_t bits_total = 0, offs_total = 0;
    for (int r = 0; r < R; ++r) {
      p.row0[r] = p.total;
      p.len[r] = cnt(r, c, 0);
      p.bit_off[r] = bits_total;
      p.off_off[r] = offs_total;
      p.byte_base[r] = p.total_bytes;
      p.data_bytes[r] = is_str ? cnt(r, c, 4) : 0;
      p.total += p.len[r];
      p.nulls += cnt(r, c, 1);
      p.any_valid = p.any_valid || cnt(r, c, 2) != 0;
      p.total_bytes += p.data_bytes[r];
      bits_total += (int64_t)ah_bitmap_bytes(p.len[r]);
      if (r != me && p.len[r] > 0) offs_total += ((p.len[r] + 1) * p.ow + 15) & ~15ll;
    }
    p.my_first = is_str ? cnt(me, c, 5) : 0;
    if (is_str && p.ow == 4 && p.total_bytes > INT32_MAX) {  // the same on every rank: nobody enters the exchange
      const long long tb = (long long)p.total_bytes;
      exchange_free(ctx, x, false);
      return ah_fail(ctx, AH_OFFSET_OVERFLOW_ERROR, "%lld", tb);
    }
  }
  ah_status st = AH_OK;
  for (int c = 0; c < n_columns && st == AH_OK; ++c) {
    ColPlan& p = x->plan[c];
    const ah_array_view& v = columns[c];
    int64_t bits_total = 0, offs_total = 0;
    for (int r = 0; r < R; ++r) {
      bits_total += (int64_t)ah_bitmap_bytes(p.len[r]);
      if (r != me && p.len[r] > 0) offs_total += ((p.len[r] + 1) * p.ow + 15) & ~15ll;
    }
    if (p.width > 0) p.vbytes = (size_t)p.total * p.width;
    else if (p.width == 0) p.vbytes = ah_bitmap_bytes(p.total);
    else p.vbytes = (size_t)std::max<int64_t>(p.total_bytes, 8);
    st = ah_out_alloc(ctx, p.vbytes, &p.ov);
    if (st == AH_OK && p.width < 0) {
      p.obytes = (size_t)(p.total + 1) * p.ow;
      st = ah_out_alloc(ctx, p.obytes, &p.oo);
      if (st == AH_OK && offs_total > 0) st = ah_pool_alloc(ctx, (size_t)offs_total, (void**)&p.ostage);
    }
    if (st == AH_OK && p.width == 0 && p.total > 0) {
      st = ah_pool_alloc(ctx, std::max<size_t>((size_t)bits_total, 8), (void**)&p.bstage);
      if (st == AH_OK && v.length > 0)
        st = ah_bitmap_op(ctx, BM_COPY, make_bitview(v.values, v.values_bit_offset), BitView{nullptr, 0}, BitView{nullptr, 0}, v.length,
                          (unsigned long long*)(p.bstage + p.bit_off[me]), nullptr);
    }
    if (st == AH_OK && p.any_valid) {
      p.bbytes = ah_bitmap_bytes(p.total);
      st = ah_out_alloc(ctx, p.bbytes, &p.ob);
      if (st == AH_OK) st = ah_pool_alloc(ctx, std::max<size_t>((size_t)bits_total, 8), (void**)&p.vstage);
      // this rank's validity packed at bit 0, zero padded to whole words, straight into its staging slot
      if (st == AH_OK && v.length > 0) {
        const bool has_v = cnt(me, c, 2) != 0;
        st = ah_bitmap_op(ctx, BM_COPY, has_v ? make_bitview(v.validity, v.validity_bit_offset) : BitView{nullptr, 0},
                          BitView{nullptr, 0}, BitView{nullptr, 0}, v.length, (unsigned long long*)(p.vstage + p.bit_off[me]), nullptr);
      }
    }
  }
  // piece tables of the merge / rebase kernels
  BitJobs jobs{};
  int njobs = 0;
  int64_t max_words = 0;
  for (int c = 0; c < n_columns && st == AH_OK; ++c) {
    ColPlan& p = x->plan[c];
    if (p.total == 0) continue;
    for (int kind = 0; kind < 2; ++kind) {
      char* stage = kind == 0 ? p.bstage : p.vstage;
      if (!stage) continue;
      BitJob& jb = jobs.j[njobs++];
      jb.out = (unsigned long long*)(kind == 0 ? p.ov : p.ob);
      jb.total_rows = p.total;
      jb.npieces = R;
      jb.first_piece = (int)x->pieces.size();
      for (int r = 0; r < R; ++r) x->pieces.push_back(Piece{(const unsigned long long*)(stage + p.bit_off[r]), p.row0[r], p.len[r]});
      max_words = std::max<int64_t>(max_words, (p.total + 63) >> 6);
    }
  }
  if (st == AH_OK && !x->pieces.empty()) st = ah_pool_alloc(ctx, sizeof(Piece) * x->pieces.size(), (void**)&x->dev_pieces);
  std::vector<int> off_first(n_columns, -1);
  for (int c = 0; c < n_columns && st == AH_OK; ++c) {
    ColPlan& p = x->plan[c];
    if (p.width >= 0) continue;
    off_first[c] = (int)x->off_pieces.size();
    for (int r = 0; r < R; ++r)
      x->off_pieces.push_back(OffPiece{r == me ? columns[c].offsets : (const void*)(p.ostage + p.off_off[r]), p.row0[r], p.len[r], p.byte_base[r]});
  }
  if (st == AH_OK && !x->off_pieces.empty()) st = ah_pool_alloc(ctx, sizeof(OffPiece) * x->off_pieces.size(), (void**)&x->dev_off_pieces);
  if (st != AH_OK) return exchange_abort(ctx, comm, x, st);

  // ---- 3. ONE grouped exchange: per (column, peer) the streams go in a fixed order — primary, offsets, validity —
  //         on both sides, so sends and receives of one pair match up
  ncclResult_t gr = api ? api->GroupStart() : ncclSuccess;
  if (gr != ncclSuccess) {
    ah_fail(ctx, AH_COMM_ERROR, "ncclGroupStart failed: %s", api->GetErrorString(gr));
    return exchange_abort(ctx, comm, x, AH_COMM_ERROR);
  }
  int64_t sent_each = 0, received = 0;
  for (int c = 0; c < n_columns; ++c) {
    ColPlan& p = x->plan[c];
    const ah_array_view& v = columns[c];
    const void* my_ptr;   // primary stream of this rank
    size_t my_bytes;
    char* my_dst;
    if (p.width > 0) {
      my_ptr = v.values, my_bytes = (size_t)v.length * p.width, my_dst = (char*)p.ov + (size_t)p.row0[me] * p.width;
    } else if (p.width == 0) {
      my_ptr = p.bstage ? p.bstage + p.bit_off[me] : nullptr, my_bytes = v.length > 0 ? ah_bitmap_bytes(v.length) : 0, my_dst = nullptr;
    } else {
      my_ptr = (const char*)v.values + p.my_first, my_bytes = (size_t)p.data_bytes[me], my_dst = (char*)p.ov + p.byte_base[me];
    }
    if (my_bytes && my_dst)  // own piece: a device-to-device copy to its final place
      hipMemcpyAsync(my_dst, my_ptr, my_bytes, hipMemcpyDeviceToDevice, ctx->stream);
    const size_t my_vb = (p.any_valid && v.length > 0) ? ah_bitmap_bytes(v.length) : 0;
    const size_t my_ob = (p.width < 0 && v.length > 0) ? (size_t)(v.length + 1) * p.ow : 0;
    for (int step = 1; step < R && gr == ncclSuccess; ++step) {  // staggered peers: every link busy in both directions
      const int dst = (me + step) % R, src = (me - step + R) % R;
      size_t src_bytes;
      char* src_dst;
      if (p.width > 0) src_bytes = (size_t)p.len[src] * p.width, src_dst = (char*)p.ov + (size_t)p.row0[src] * p.width;
      else if (p.width == 0) src_bytes = p.len[src] > 0 ? ah_bitmap_bytes(p.len[src]) : 0, src_dst = p.bstage + p.bit_off[src];
      else src_bytes = (size_t)p.data_bytes[src], src_dst = (char*)p.ov + p.byte_base[src];
      if (my_bytes) gr = api->Send(my_ptr, my_bytes, ncclChar, dst, comm->nccl, ctx->stream);
      if (gr == ncclSuccess && src_bytes) gr = api->Recv(src_dst, src_bytes, ncclChar, src, comm->nccl, ctx->stream);
      const size_t src_ob = (p.width < 0 && p.len[src] > 0) ? (size_t)(p.len[src] + 1) * p.ow : 0;
      if (gr == ncclSuccess && my_ob) gr = api->Send(v.offsets, my_ob, ncclChar, dst, comm->nccl, ctx->stream);
      if (gr == ncclSuccess && src_ob) gr = api->Recv(p.ostage + p.off_off[src], src_ob, ncclChar, src, comm->nccl, ctx->stream);
      const size_t src_vb = (p.any_valid && p.len[src] > 0) ? ah_bitmap_bytes(p.len[src]) : 0;
      if (gr == ncclSuccess && my_vb) gr = api->Send(p.vstage + p.bit_off[me], my_vb, ncclChar, dst, comm->nccl, ctx->stream);
      if (gr == ncclSuccess && src_vb) gr = api->Recv(p.vstage + p.bit_off[src], src_vb, ncclChar, src, comm->nccl, ctx->stream);
      sent_each += (int64_t)(my_bytes + my_ob + my_vb);
      received += (int64_t)(src_bytes + src_ob + src_vb);
    }
  }
  if (api) {
    ncclResult_t ge = api->GroupEnd();
    if (gr == ncclSuccess) gr = ge;
    if (gr != ncclSuccess) {
      ah_fail(ctx, AH_COMM_ERROR, "RCCL exchange failed: %s", api->GetErrorString(gr));
      return exchange_abort(ctx, comm, x, AH_COMM_ERROR);
    }
  }
  if (R > 1) sent_each /= (R - 1);
  x->sent_each = sent_each;
  x->received = received;

  // ---- 4. every bitmap merged by ONE launch, one rebase launch per string column
  hipError_t e = hipSuccess;
  if (njobs > 0) {
    e = hipMemcpyAsync(x->dev_pieces, x->pieces.data(), sizeof(Piece) * x->pieces.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(4096, ah_ceil_div(max_words, 256)));
      merge_jobs_kernel<<<dim3(gx, (unsigned)njobs), 256, 0, ctx->stream>>>(jobs, x->dev_pieces);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess && !x->off_pieces.empty()) {
    e = hipMemcpyAsync(x->dev_off_pieces, x->off_pieces.data(), sizeof(OffPiece) * x->off_pieces.size(), hipMemcpyHostToDevice,
                       ctx->stream);
    for (int c = 0; c < n_columns && e == hipSuccess; ++c) {
      ColPlan& p = x->plan[c];
      if (p.width >= 0) continue;
      const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, ah_ceil_div(p.total + 1, 256)));
      if (p.ow == 4)
        rebase_offsets_kernel<int32_t><<<gx, 256, 0, ctx->stream>>>((int32_t*)p.oo, p.total, p.total_bytes, R, x->dev_off_pieces + off_first[c]);
      else
        rebase_offsets_kernel<int64_t><<<gx, 256, 0, ctx->stream>>>((int64_t*)p.oo, p.total, p.total_bytes, R, x->dev_off_pieces + off_first[c]);
      e = hipGetLastError();
    }
  }
  if (e != hipSuccess) {
    ah_fail(ctx, AH_HIP_ERROR, "all-gather merge failed: %s", hipGetErrorString(e));
    // the exchange itself was issued on every rank: no abort needed, but the stream must drain before the buffers go
    hipStreamSynchronize(ctx->stream);
    const std::string msg = ctx->err;
    exchange_free(ctx, x, false);
    return ah_fail(ctx, AH_HIP_ERROR, "%s", msg.c_str());
  }
  x->enqueued = true;
  ctx->inflight = true;  // the caller's columns are still being read
  *handle = x;
  return AH_OK;
}

extern "C" ah_status ah_all_gather_columns_end(ah_context* ctx, ah_comm* comm, ah_exchange* x, ah_array_out* outs,
                                               ah_exchange_stats* stats) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !comm || !x || !outs || x->comm != comm) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  for (int c = 0; c < x->n; ++c) ah_out_init(&outs[c]);
  ah_status st = AH_OK;
  hipError_t e = ah_stream_wait(ctx);  // results usable at return (synchronous-by-default contract)
  if (e != hipSuccess) st = ah_fail(ctx, AH_HIP_ERROR, "all-gather failed: %s", hipGetErrorString(e));
  if (st == AH_OK) st = check_async(ctx, comm);
  if (st != AH_OK) {
    const std::string msg = ctx->err;
    exchange_free(ctx, x, false);
    return ah_fail(ctx, st, "%s", msg.c_str());
  }
  for (int c = 0; c < x->n; ++c) {
    ColPlan& p = x->plan[c];
    outs[c].type = p.type;
    outs[c].length = p.total;
    outs[c].values = p.ov;
    outs[c].values_bytes = (int64_t)p.vbytes;
    outs[c].offsets = p.oo;
    outs[c].offsets_bytes = (int64_t)p.obytes;
    if (p.any_valid) {
      outs[c].validity = (uint8_t*)p.ob;
      outs[c].validity_bytes = (int64_t)p.bbytes;
      outs[c].null_count = p.nulls;
    }
  }
  if (stats) {
    stats->peers = comm->world - 1;
    stats->bytes_to_each_peer = x->sent_each;
    stats->bytes_received = x->received;
    stats->counts_ms = x->t_counts - x->t0;
    stats->total_ms = now_ms() - x->t0;
  }
  exchange_free(ctx, x, true);
  return AH_OK;
}

// concat of every rank's columns, in rank order, materialised on every rank
extern "C" ah_status ah_all_gather_columns(ah_context* ctx, ah_comm* comm, int32_t n_columns, const ah_array_view* columns,
                                           ah_array_out* outs, ah_exchange_stats* stats) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !comm || !outs) return AH_INVALID_ARGUMENT;
  for (int c = 0; c < n_columns && c < EX_MAX_COLS; ++c) ah_out_init(&outs[c]);
  ah_exchange* x = nullptr;
  AH_TRY(ah_all_gather_columns_begin(ctx, comm, n_columns, columns, &x));
  return ah_all_gather_columns_end(ctx, comm, x, outs, stats);
}

extern "C" ah_status ah_all_gatherv(ah_context* ctx, ah_comm* comm, const ah_array_view* local, ah_array_out* out,
                                    ah_exchange_stats* stats) {
  ah_ctx_guard _guard(ctx);
  return ah_all_gather_columns(ctx, comm, 1, local, out, stats);
}
