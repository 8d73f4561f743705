// common.hpp — shared host/device plumbing for libarrow_hip.so (gfx950 only).
//
// Nothing here mirrors a reference file; it is the runtime the reference gets
// from Rust's allocator/Vec (arrow-buffer/src/buffer/mutable.rs) re-thought for
// HBM: a pooled device allocator, pinned read-back slots and HIP-event kernel
// timing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/arrow_hip.h"

// ------------------------------------------------------------------ context
struct ah_prof_entry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0.0;
  int64_t launches = 0;
};

struct ah_context {
  // Entry points lock the context for the duration of the call (recursive: they call each other), so ONE context
  // may be shared by many host threads — the reference's kernels are re-entrant over Send + Sync arrays
  // (arrow-array/src/array/mod.rs:100) and a Rust wrapper can be `Send + Sync` without a mutex of its own.  Calls on
  // one context serialise (they share its stream and read-back slots); threads that want overlap use one context each.
  std::recursive_mutex mu;
  const char* last_entry = nullptr;  // the entry point that last took the context (fault reporter)
  uint64_t entry_calls = 0;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // created on first use (coalesce.hip: table uploads beside the previous push's scatter); a HIP stream costs milliseconds to create
  std::string err;
  // output allocator hook
  ah_alloc_fn alloc = nullptr;
  ah_free_fn free_ = nullptr;
  void* user = nullptr;
  // built-in pool: exact-size free lists of hipMalloc'd blocks
  std::map<size_t, std::vector<void*>> pool_free;
  std::unordered_map<void*, size_t> pool_live;  // ptr -> rounded size
  std::unordered_map<void*, size_t> redzones;   // AH_DEBUG_REDZONE: ptr -> requested size
  // AH_DEBUG_GUARD (context.hip): every pool block is its own mapping, flush against an unmapped granule
  struct guard_block {
    void* va;
    size_t va_bytes, map_bytes;
    hipMemGenericAllocationHandle_t handle;
  };
  std::unordered_map<void*, guard_block> guard_live;  // buffer pointer -> its reservation
  // output buffers carved out of ONE pool block (BatchCoalescer's slab push: thousands of 8192-row output batches per
  // allocation): slice pointer -> its slab; the block goes back to the pool when the last slice (and the creator) let go
  // (found by ADDRESS RANGE: a pointer inside a live slab's block is a slice of it — no per-slice bookkeeping: a slab push
  // hands out 30 000 output batches per 1e9 rows at the reference's batch sizes)
  std::map<uintptr_t, struct ah_slab*> slabs;  // block base -> slab
  // A fault a DEFERRED call found on the device (ah_take's out-of-bounds index: the reference's panic) stays there until the
  // host next waits for the stream: 4 device words {position (~0 = none), index value, values length, kind}; the first fault
  // in stream order wins.  ah_synchronize / ah_array_resolve read them when `fault_armed` and raise the error.
  unsigned long long* fault_dev = nullptr;
  bool fault_armed = false, capture_fault_armed_before = false;
  struct hook_entry {
    ah_free_fn free_;
    void* user;
  };
  std::unordered_map<void*, hook_entry> hook_live;  // outputs handed out by the host allocator hook, with THEIR free
  // pinned host read-back slots: 255 x u64 of payload + the mailbox sequence word [255].  Device kernels write
  // the payload and then the sequence word (system-scope release); the host spins on the word instead of
  // entering hipStreamSynchronize, whose interrupt-driven wake-up costs 0.03-0.8 ms per wait depending on the box
  // (2.4 ms of the 7.7 ms driver-timed step).
  uint64_t* pinned = nullptr;      // host address, 256 x u64
  uint64_t* pinned_dev = nullptr;  // the same memory as the device sees it
  uint64_t mail_seq = 0;           // last sequence number handed to a kernel
  int wait_mode = 0;               // 0 = spin on the mailbox (default), 1 = hipStreamSynchronize (AH_WAIT=block)
  // persistent self-cleaning device scratch (AH_SCRATCH_WORDS u64): [0,512) zero between calls (valid-row
  // counters), [64,72) all-ones between calls (first-failing-position words); the finish kernels that read
  // them back also restore them, so no per-call hipMemsetAsync
  unsigned long long* scratch = nullptr;
  std::vector<hipEvent_t> event_pool;  // recycled profiling events
  // deferred mode (ah_context_set_deferred): infallible fixed-shape kernels skip the end-of-call sync
  bool deferred = false;
  // "unwaited work in flight": set by every entry point that returns while its kernels may still be reading the
  // caller's buffers (the *_acc forms, the coalescer's pushes, ah_all_gather_columns_begin); cleared by any successful
  // host wait on the stream.  ah_out_free drains the stream before a HOST-allocator free while it is set (the host's
  // free is not stream-ordered).
  bool inflight = false;
  // small pinned (host-coherent, device-mapped) blocks handed to objects that own mailboxes of their own (a
  // BatchCoalescer's null-count ring and count words).  hipHostMalloc costs 0.1-0.3 ms, which showed up as idle GPU at
  // the start of every coalescer's life (profiles/r04_coalesce_gaps.md): released blocks are kept here and reused.
  std::vector<std::pair<size_t, void*>> pinned_cache;
  // hipGraph capture (ah_graph_begin / _end): while `capturing`, every launch of the context's stream is recorded instead
  // of run; pooled blocks released meanwhile are NOT recycled (a replay writes them again) but parked in `capture_hold`,
  // which the finished graph takes over; any host wait is refused (the kernels it would wait for are not running)
  bool capturing = false, capture_was_deferred = false;
  std::vector<void*> capture_hold;
  // memory accounting (ah_context_stats; SURVEY 5 "allocation high-water / bytes moved", the analogue of the reference's
  // MemoryPool::used / TrackingMemoryPool, arrow-buffer/src/pool.rs:73-93): device bytes held by live pool blocks
  // (rounded sizes), their high-water mark, cumulative bytes and calls, and what the free lists cache
  ah_context_stats_t stats{};
  // profiling
  bool profiling = false;
  std::map<std::string, ah_prof_entry> prof;
};

// context.hip: the fault reporter's view of "what was the library doing" (see ah_fault_note)
void ah_fault_enter(ah_context* c, const char* entry_point);

struct ah_ctx_guard {
  std::unique_lock<std::recursive_mutex> lk;
  // `fn`: the name of the C-ABI entry point that constructs the guard (every one does, first thing)
  explicit ah_ctx_guard(ah_context* c, const char* fn = __builtin_FUNCTION()) {
    if (c) {
      lk = std::unique_lock<std::recursive_mutex>(c->mu);
      ah_fault_enter(c, fn);
    }
  }
};

// Fault reporter (context.hip): every device / pinned block the library hands out or holds is noted in a fixed lock-free
// table; when the GPU raises a memory fault the HSA runtime calls the library's system-event handler BEFORE it prints
// "Memory access fault ..." and aborts, and the handler says which block the address belongs to (or lies just outside of),
// whether that block is live, cached or released, and which entry point was running.
enum ah_fault_kind { AH_FK_POOL = 1, AH_FK_PINNED = 2, AH_FK_CONTEXT = 3 };
enum ah_fault_state { AH_FS_LIVE = 1, AH_FS_CACHED = 2, AH_FS_RELEASED = 3 };
void ah_fault_note(const void* p, size_t bytes, int kind, int state, const char* what);

ah_status ah_fail(ah_context* ctx, ah_status st, const char* fmt, ...);
// a pinned, device-mapped host block of at least `bytes` (zeroed); ah_pinned_free returns it to the context's cache
ah_status ah_pinned_alloc(ah_context* ctx, size_t bytes, void** host, void** dev);
void ah_pinned_free(ah_context* ctx, void* host, size_t bytes);

// (hipErrorAssert is what the library's own waits return while a deferred ah_take's out-of-bounds fault is outstanding:
// context.hip fault_peek)
#define AH_HIP(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e == hipErrorAssert)                                                              \
      return ah_fail((ctx), AH_PANIC, "%s", AH_DEFERRED_FAULT_TEXT);                       \
    if (_e != hipSuccess)                                                                  \
      return ah_fail((ctx), _e == hipErrorOutOfMemory ? AH_OUT_OF_MEMORY : AH_HIP_ERROR,   \
                     "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__,        \
                     __LINE__, #expr);                                                     \
  } while (0)
#define AH_DEFERRED_FAULT_TEXT                                                                                           \
  "a deferred ah_take on this context met an out-of-bounds index: its result, and everything computed from it, is invalid " \
  "(ah_synchronize reports the index with the reference's panic text)"

#define AH_TRY(expr)            \
  do {                          \
    ah_status _s = (expr);      \
    if (_s != AH_OK) return _s; \
  } while (0)

constexpr int AH_MAIL_FLAG = 255;     // index of the sequence word in ctx->pinned
constexpr int AH_SCRATCH_WORDS = 640;  // 8 x 64 zero-state counters (one set per column of a fused launch) + 8 ones-state position words + 40 + 64 zero-state counter words
constexpr int AH_SCRATCH_ONES = 512;     // [512, 520): all ones between calls
constexpr int AH_SCRATCH_TICKETS = 520;  // [520, 560): zero between calls (filter_small.hip's completion tickets)

// enqueue a device -> pinned-slot copy (a one-wave kernel, stream-ordered like hipMemcpyAsync); `pinned_dst` must
// point into ctx->pinned and `bytes` is a multiple of 8
hipError_t ah_d2h(ah_context* ctx, void* pinned_dst, const void* dev_src, size_t bytes);
// wait until everything enqueued on the context's stream so far has finished (a flag kernel + a host spin)
hipError_t ah_stream_wait(ah_context* ctx);
// both in one launch; `reset`: after reading, the device words are overwritten with `reset_value`
hipError_t ah_d2h_wait(ah_context* ctx, void* pinned_dst, const void* dev_src, size_t bytes, bool reset = false,
                       uint64_t reset_value = 0);
// for kernels that post the mailbox themselves (ah_mail_post): reserve the next sequence number, wait for it
static inline uint64_t ah_mail_next(ah_context* ctx) { return ++ctx->mail_seq; }
hipError_t ah_mail_wait(ah_context* ctx, uint64_t seq);
hipError_t ah_mail_post_async(ah_context* ctx, uint64_t* seq_out);  // post only; pair with ah_mail_wait

// Deferred mode plumbing.  A call that can run deferred hands AH_COUNT(ctx, &n) to the helpers that
// would read a popcount back (nullptr = no read-back), ends with ah_end_of_call_sync() and reports
// ah_nulls(): len - set_bits, or -1 when nothing was read back.  Scratch blocks go back to the
// context's pool right after the enqueue: the pool is per context = per stream, so reuse is stream-ordered.
#define AH_COUNT(ctx, ptr) ((ctx)->deferred ? nullptr : (ptr))
static inline hipError_t ah_end_of_call_sync(ah_context* ctx) {
  return ctx->deferred ? hipSuccess : ah_stream_wait(ctx);
}
static inline int64_t ah_nulls(const ah_context* ctx, int64_t len, int64_t set_bits) {
  return ctx->deferred ? -1 : len - set_bits;
}

bool ah_guard_mode();  // AH_DEBUG_GUARD=1 (context.hip)
// pooled device memory (internal + default output allocator)
ah_status ah_pool_alloc(ah_context* ctx, size_t bytes, void** out);
void ah_pool_free(ah_context* ctx, void* p);
// output buffers honour the host allocator hook
ah_status ah_out_alloc(ah_context* ctx, size_t bytes, void** out);
void ah_out_free(ah_context* ctx, void* p, size_t bytes);

// HIP-event bracketing of hot kernels on the launch stream
struct ah_prof_scope {
  ah_context* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  ah_prof_scope(ah_context* c, const char* n) : ctx(c), name(n) {
    if (ctx->profiling) {
      a = take_event();
      b = take_event();
      hipEventRecord(a, ctx->stream);
    }
  }
  hipEvent_t take_event() {
    hipEvent_t e = nullptr;
    if (!ctx->event_pool.empty()) {
      e = ctx->event_pool.back();
      ctx->event_pool.pop_back();
    } else {
      hipEventCreate(&e);
    }
    return e;
  }
  ~ah_prof_scope() {
    if (ctx->profiling) {
      hipEventRecord(b, ctx->stream);
      ctx->prof[name].pending.emplace_back(a, b);
    }
  }
};

static inline int64_t ah_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t ah_bitmap_bytes(int64_t bits) {  // whole u64 words, like collect_bool
  return (size_t)ah_ceil_div(bits, 64) * 8;
}

int ah_type_width(ah_type t);  // bytes per value; 0 for AH_BOOL; -1 if not fixed-width
const char* ah_type_name(ah_type t);  // DataType Display text ("Int64", "Float64", ...)
bool ah_type_is_integer(ah_type t);
bool ah_type_is_signed(ah_type t);
bool ah_type_is_float(ah_type t);

void ah_out_init(ah_array_out* out);
// context.hip: after the stream has been waited for — the fault a deferred call left on the device, if any (AH_PANIC with the
// reference's text), else AH_OK
ah_status ah_check_deferred_fault(ah_context* ctx);

// context.hip: a pool block whose slices are handed out as results (released through ah_array_release / ah_out_free like
// any other output buffer)
struct ah_slab {
  void* block = nullptr;
  size_t bytes = 0;
  int64_t refs = 0;
};
ah_status ah_slab_create(ah_context* ctx, size_t bytes, ah_slab** out);  // refs = 1: the creator's
void ah_slab_slice(ah_context* ctx, ah_slab* s, void* ptr);              // `ptr` (inside the block) leaves as an output buffer: one more reference
void ah_slab_unref(ah_context* ctx, ah_slab* s);                         // the creator (or a slice) lets go

// strings.hip: take for Utf8 / LargeUtf8
ah_status ah_take_bytes(ah_context* ctx, const ah_array_view* values, const ah_array_view* indices,
                        ah_array_out* out);
// bitmap.hip: ah_bitmap_set_bits without a read-back — *nulls_acc (device) += len - popcount(copied bits)
ah_status ah_bitmap_set_bits_acc(ah_context* ctx, uint8_t* dst, int64_t dst_bit_offset, const uint8_t* src,
                                 int64_t src_bit_offset, int64_t len, unsigned long long* nulls_acc);

// count valid bits of a view's validity (or trust view->null_count >= 0)
ah_status ah_resolve_null_count(ah_context* ctx, const ah_array_view* v, int64_t* nulls);

// ---------------------------------------------------------------- device side
#ifdef __HIPCC__

// A bit-packed LSB-first stream with a bit offset (BooleanBuffer,
// arrow-buffer/src/buffer/boolean.rs:97-104) normalised to 8-byte words:
// bit i of the stream is bit (off + i) of `words`, 0 <= off < 64.
// words == nullptr encodes "all ones" (no null buffer).
struct BitView {
  const uint64_t* words;
  int64_t off;
};

static inline BitView make_bitview(const void* bytes, int64_t bit_offset) {
  BitView v;
  if (!bytes) {
    v.words = nullptr;
    v.off = 0;
    return v;
  }
  uintptr_t p = (uintptr_t)bytes + (uintptr_t)(bit_offset >> 3);
  int64_t off = bit_offset & 7;
  uintptr_t al = p & ~(uintptr_t)7;
  off += (int64_t)(p - al) * 8;
  v.words = (const uint64_t*)al;
  v.off = off;
  return v;
}

// bits [s, s+64) of the stream (s may be any value >= 0), bits at index >= len
// read as 0.  Never touches a word that holds no bit below `len`.
__device__ __forceinline__ uint64_t bv_fetch64(const BitView& v, int64_t s, int64_t len) {
  if (s >= len) return 0;
  if (!v.words) {
    int64_t rem = len - s;
    return rem >= 64 ? ~0ull : ((1ull << rem) - 1);
  }
  int64_t pos = v.off + s;
  int64_t w = pos >> 6;
  int sh = (int)(pos & 63);
  uint64_t lo = v.words[w] >> sh;
  if (sh) {
    int64_t last_word = (v.off + len - 1) >> 6;
    uint64_t hi = (w + 1 <= last_word) ? v.words[w + 1] : 0;
    lo |= hi << (64 - sh);
  }
  int64_t rem = len - s;
  if (rem < 64) lo &= (1ull << rem) - 1;
  return lo;
}

// bv_fetch64 split in two so that SEVERAL bitmaps' words can be in flight together: bv_issue only issues loads (both
// candidate words, no branch, no use of the data), bv_finish assembles the 64 bits.  With bv_fetch64 itself hipcc
// waits (s_waitcnt vmcnt(0)) inside each call: mask, mask validity and value validity cost three serialized memory
// round trips per workgroup.
struct BvRaw {
  uint64_t lo, hi;
  int sh;
  bool has_hi;
};
__device__ __forceinline__ BvRaw bv_issue(const BitView& v, int64_t s, int64_t len) {  // needs v.words and s < len
  BvRaw r;
  const int64_t pos = v.off + s, w = pos >> 6, last_word = (v.off + len - 1) >> 6;
  r.sh = (int)(pos & 63);
  r.has_hi = r.sh != 0 && w + 1 <= last_word;
  r.lo = v.words[w];
  r.hi = v.words[w + 1 <= last_word ? w + 1 : last_word];
  return r;
}
__device__ __forceinline__ uint64_t bv_finish(const BvRaw& r, int64_t s, int64_t len) {
  uint64_t x = r.lo >> r.sh;
  if (r.has_hi) x |= r.hi << (64 - r.sh);
  const int64_t rem = len - s;
  if (rem < 64) x &= (1ull << rem) - 1;
  return x;
}

// Bit p of the low 64 / V bits of x -> bit p * V, written so that it runs on the VECTOR unit when x depends on the lane
// (cmp.hip / cast.hip: lane k < V assembles output word k of a 64 * V-row group from the V ballots).  Built on the scalar
// unit — all V words per wave, ~15 scalar instructions per spread — these interleaves made kernels scalar-issue bound:
// one scalar unit serves the four SIMDs of a CU.
__device__ __forceinline__ uint32_t vspread16x2(uint32_t x) {  // 16 bits -> 32, stride 2
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}
__device__ __forceinline__ uint32_t vspread8x4(uint32_t x) {  // 8 bits -> 32, stride 4
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x;
}
template <int V> __device__ __forceinline__ uint64_t vspread(uint64_t x);
template <> __device__ __forceinline__ uint64_t vspread<1>(uint64_t x) { return x; }
template <> __device__ __forceinline__ uint64_t vspread<2>(uint64_t x) {  // low 32 -> 64
  const uint32_t h = (uint32_t)x;
  return (uint64_t)vspread16x2(h & 0xFFFFu) | ((uint64_t)vspread16x2(h >> 16) << 32);
}
template <> __device__ __forceinline__ uint64_t vspread<4>(uint64_t x) {  // low 16 -> 64
  const uint32_t h = (uint32_t)x;
  return (uint64_t)vspread8x4(h & 0xFFu) | ((uint64_t)vspread8x4((h >> 8) & 0xFFu) << 32);
}


// A wave-uniform value the compiler can see is uniform (scalar register): addresses built from it become scalar
// loads (s_load, counted by lgkmcnt), which do not make the vector loads already in flight wait.
__device__ __forceinline__ int ah_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t ah_uniform64(int64_t x) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Validity bits of the V consecutive rows a lane owns — rows base + lane * V .. + V - 1 — when `base` is
// wave-uniform (made so with ah_uniform) and a multiple of 64: the V words of the wave's 64 * V rows come through
// the scalar unit and each lane picks its own.  A per-lane bv_fetch64 made hipcc put `s_waitcnt vmcnt(0)` right
// behind the fetch, i.e. one full memory round trip per group with nothing else in flight.
template <int V>
__device__ __forceinline__ uint32_t bv_lane_bits(const BitView& v, int64_t base, int lane, int64_t len) {
  uint64_t mine = 0;
  if (v.words && (v.off & 63) == 0 && base + 64 * V <= len) {  // whole words: one wide scalar load
    const uint64_t* w = v.words + ((v.off + base) >> 6);
    uint64_t ww[V];
#pragma unroll
    for (int j = 0; j < V; ++j) ww[j] = w[j];
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (((lane * V) >> 6) == j) mine = ww[j];
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const uint64_t w = bv_fetch64(v, base + 64 * j, len);
      if (((lane * V) >> 6) == j) mine = w;
    }
  }
  return (uint32_t)(mine >> ((lane * V) & 63)) & ((1u << V) - 1u);
}

__device__ __forceinline__ int bv_get(const BitView& v, int64_t i) {
  if (!v.words) return 1;
  int64_t pos = v.off + i;
  return (int)((((const uint8_t*)v.words)[pos >> 3] >> (pos & 7)) & 1);
}

__device__ __forceinline__ unsigned long long wave_reduce_add64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// ---------------------------------------------------------------------------------------------------------- Float16
// `half::f16` as arrow-rs sees it on its x86-64 host (crate `half` 2.7.1, software conversions: arrow-buffer/Cargo.toml:46 takes
// it with default-features = false): f16 -> f32 exact, a NaN keeps sign and payload (<< 13) and gets the quiet bit; f32 -> f16
// round-to-nearest-even, a NaN becomes sign | 0x7C00 | 0x0200 | (mantissa >> 13); arithmetic is "to f32, one f32 operation,
// one rounding back" — never a native half-precision instruction (v_add_f16 would round once where the reference rounds twice).
// Device: the hardware converts (v_cvt_f32_f16 / v_cvt_f16_f32, round-to-nearest-even, denormals kept) carry the non-NaN
// values, the NaN rule is applied explicitly; host (error texts only): integer arithmetic.
struct ah_f16 {
  uint16_t bits;
};
__host__ __device__ __forceinline__ float ah_f16_to_f32(ah_f16 h) {
  const uint32_t sign = (uint32_t)(h.bits & 0x8000u) << 16, exp = h.bits & 0x7C00u, man = h.bits & 0x03FFu;
#ifdef __HIP_DEVICE_COMPILE__
  if (exp == 0x7C00u && man) return __uint_as_float(sign | 0x7FC00000u | (man << 13));
  _Float16 x;
  __builtin_memcpy(&x, &h.bits, 2);
  return (float)x;
#else
  uint32_t x;
  if (exp == 0x7C00u) x = man ? (sign | 0x7FC00000u | (man << 13)) : (sign | 0x7F800000u);
  else if (exp == 0) {
    if (man == 0) x = sign;
    else {
      int e = 0;
      uint32_t m = man;
      while (!(m & 0x0400u)) { m <<= 1; ++e; }
      x = sign | ((uint32_t)(127 - 15 - e + 1) << 23) | ((m & 0x03FFu) << 13);
    }
  } else x = sign | (((exp >> 10) + (127 - 15)) << 23) | (man << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
#endif
}
__device__ __forceinline__ ah_f16 ah_f32_to_f16(float f) {
  const uint32_t x = __float_as_uint(f);
  ah_f16 r;
  if ((x & 0x7FFFFFFFu) > 0x7F800000u) {
    r.bits = (uint16_t)(((x >> 16) & 0x8000u) | 0x7E00u | ((x & 0x007FFFFFu) >> 13));
    return r;
  }
  const _Float16 h = (_Float16)f;
  __builtin_memcpy(&r.bits, &h, 2);
  return r;
}
template <typename T> struct ah_is_fp { static constexpr bool value = std::is_floating_point<T>::value; };
template <> struct ah_is_fp<ah_f16> { static constexpr bool value = true; };

// inclusive scan across the 64 lanes of a wave
// Six `v_add_u32_dpp` (row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3; lanes without a source add 0).  The __shfl_up form this replaces compiled to six
// ds_bpermute round trips with ~7 vector instructions each (42 + 6 LDS waits in every scatter tile's table build).
__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}


// Mailbox post: ONE thread calls this after the payload stores it (or its wave) made to `mail` (= ctx->pinned_dev).
// The fence orders the payload before the sequence word at system scope; the host polls the word.
__device__ __forceinline__ void ah_mail_post(uint64_t* mail, uint64_t seq) {
  __threadfence_system();
  __hip_atomic_store(mail + AH_MAIL_FLAG, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Once-read / once-written streams: `ah_ld_stream<NT>` / `ah_st_stream<NT>` are plain vector
// accesses (NT = false) or the non-temporal forms (`global_load ... nt` / `global_store ... nt`).  Which kernel uses
// which was settled by an A/B on MI355X (profiles/r03_nt_ablation.md: -DAH_NT_LOADS=1 / -DAH_NT_STORES=1 builds of all
// four streaming kernels): the numeric cast gains 7 % with both, the filter scatter 3-4 % with nt stores, arithmetic and
// compare gain nothing (and lose 1-5 % with nt stores alone).  The per-kernel defaults below are that outcome; the -D
// overrides remain for re-running the ablation.
#ifndef AH_NT_LOADS
#define AH_NT_LOADS -1  // -1: per-kernel default; 0 / 1: force for every streaming kernel
#endif
#ifndef AH_NT_STORES
#define AH_NT_STORES -1
#endif
constexpr bool ah_nt_l(bool kernel_default) { return AH_NT_LOADS < 0 ? kernel_default : AH_NT_LOADS != 0; }
constexpr bool ah_nt_s(bool kernel_default) { return AH_NT_STORES < 0 ? kernel_default : AH_NT_STORES != 0; }
template <int BYTES> struct ah_raw_vec;
template <> struct ah_raw_vec<1> { typedef uint8_t type; };
template <> struct ah_raw_vec<2> { typedef uint16_t type; };
template <> struct ah_raw_vec<4> { typedef uint32_t type; };
template <> struct ah_raw_vec<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct ah_raw_vec<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <bool NT, typename VT> __device__ __forceinline__ VT ah_ld_stream(const VT* p) {
  if constexpr (NT && (sizeof(VT) == 1 || sizeof(VT) == 2 || sizeof(VT) == 4 || sizeof(VT) == 8 || sizeof(VT) == 16)) {
    using R = typename ah_raw_vec<sizeof(VT)>::type;
    const R x = __builtin_nontemporal_load((const R*)p);
    VT r;
    __builtin_memcpy(&r, &x, sizeof(VT));
    return r;
  } else {
    return *p;
  }
}
template <bool NT, typename VT> __device__ __forceinline__ void ah_st_stream(VT* p, const VT& v) {
  if constexpr (NT && (sizeof(VT) == 1 || sizeof(VT) == 2 || sizeof(VT) == 4 || sizeof(VT) == 8 || sizeof(VT) == 16)) {
    using R = typename ah_raw_vec<sizeof(VT)>::type;
    R x;
    __builtin_memcpy(&x, &v, sizeof(VT));
    __builtin_nontemporal_store(x, (R*)p);
  } else {
    *p = v;
  }
}

// A counting kernel adds its per-block partials into a zero-between-calls word of ctx->scratch (one atomicAdd per block,
// none for an empty partial); ONE small kernel behind it (`ah_d2h_wait(..., reset = true)`: copy the word(s) into the pinned
// slots, zero them, post the mailbox) is the call's read-back.  Two launches per counted bitmap op where round 2 had four
// (partials buffer, one-block sum, copy kernel).  Posting from INSIDE the counting kernel (last block to arrive) was
// tried: it saves the second launch (filter 14.8 us at 10^4 rows), but the host then returns while the kernel that wrote
// the OUTPUT is still retiring — its end-of-kernel release is what makes the output visible to other streams and copy
// engines, and waiting for the runtime to report the kernel complete (hipStreamQuery) costs 13 us.  A separate posting
// kernel starts only after the producing kernel has ended, so "synchronous at return" holds for any consumer.
// (64 counters, picked by block index: 4096 blocks finishing together on ONE word serialize at ~12 ns per atomic — 50 us)
__device__ __forceinline__ void ah_count_add(unsigned long long* words, unsigned long long count) {
  if (count) atomicAdd(words + (blockIdx.x & 63), count);
}
constexpr int AH_TICKET_COUNT = AH_SCRATCH_TICKETS + 40;  // [560, 624): the 64 counter words of the counting kernels
// the read-back of ah_count_add: 64 counters -> pinned slots, counters zeroed, mailbox posted, host sum
hipError_t ah_count_read(ah_context* ctx, int64_t* total);
// a caller that bails out AFTER enqueueing a counting kernel without reading it: stream drained, counters back to zero
void ah_count_reset(ah_context* ctx);

struct ah_filter_predicate;
// strings.hip: filter_bytes + filter_nulls of a Utf8 / LargeUtf8 column — offsets, data and (vvalid.words != nullptr) the
// compacted validity of the rows the predicate selects: ranges (one pass over the offsets) -> tile scan -> gather, one host
// wait for {byte total, valid rows} in between
ah_status ah_string_filter_bytes(ah_context* ctx, const ah_filter_predicate* p, const ah_array_view* values, BitView vvalid,
                                 ah_array_out* out);

// ---- shared bitmap machinery (bitmap.hip)
enum ah_bitmap_opcode {
  BM_COPY = 0,            // a
  BM_NOT = 1,             // ~a
  BM_AND = 2,             // a & b
  BM_DISTINCT_BOTH = 3,   // (a ^ b) | (a & b & c)     cmp.rs:329-335
  BM_NOT_DISTINCT_BOTH = 4,  // ~(a | b) | (a & b & c) cmp.rs:336-343
  BM_ORNOT = 5,           // ~a | b                    cmp.rs:366-372
  BM_OR = 6,              // a | b
  BM_ANDNOT = 7,          // a & ~b                    boolean.rs:291 and_not
  BM_OR_NOTB = 8,         // a | ~b                    boolean.rs:88 (and_kleene, one nullable side)
  BM_KLEENE_AND_NULLS = 9,   // (a | (c & ~d)) & (c | (a & ~b))   boolean.rs:121
  BM_KLEENE_OR_NULLS = 10,   // (a | (c & d)) & (c | (a & b))     boolean.rs:213
  BM_NULLIF = 11,            // a & ~(b & c)                      nullif.rs:54-56
};
// out_words[w] = op(a, b, c) over `len` bits (each input a BitView with its own
// offset; words == nullptr reads as all-ones); bits past len are zeroed.
// *set_bits (optional) receives the popcount of the result.  Synchronous only
// when set_bits != nullptr.
ah_status ah_bitmap_op(ah_context* ctx, int op, BitView a, BitView b, BitView c, int64_t len,
                       unsigned long long* out_words, int64_t* set_bits, BitView d = BitView{nullptr, 0});

#endif  // __HIPCC__
