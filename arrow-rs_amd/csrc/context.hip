// context.hip — context, pooled HBM allocator, read-back slots, HIP-event
// profiling and bitmap popcount.  Runtime plumbing: the reference's analogue is
// Rust's global allocator + `MutableBuffer` (arrow-buffer/src/buffer/mutable.rs)
// and `BooleanBuffer::count_set_bits` (arrow-buffer/src/buffer/boolean.rs).
#include "common.hpp"
#include <sched.h>

#include <ctime>

// The message of a failing call is kept per THREAD as well as per context: with one context shared by threads,
// `ah_last_error` right after a failing call must not return another thread's later failure.
namespace {
struct TlsError {
  const ah_context* ctx = nullptr;
  std::string msg;
};
thread_local TlsError tls_error;
}  // namespace

ah_status ah_fail(ah_context* ctx, ah_status st, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  tls_error.ctx = ctx;
  tls_error.msg = buf;
  return st;
}

int ah_type_width(ah_type t) {
  switch (t) {
    case AH_BOOL: return 0;
    case AH_INT8: case AH_UINT8: return 1;
    case AH_INT16: case AH_UINT16: case AH_FLOAT16: return 2;
    case AH_INT32: case AH_UINT32: case AH_FLOAT32: return 4;
    case AH_INT64: case AH_UINT64: case AH_FLOAT64: return 8;
    case AH_FIXED16: case AH_UTF8_VIEW: case AH_BINARY_VIEW: return 16;
    case AH_FIXED32: return 32;
    default: return -1;
  }
}
const char* ah_type_name(ah_type t) {
  switch (t) {
    case AH_BOOL: return "Boolean";
    case AH_INT8: return "Int8"; case AH_INT16: return "Int16";
    case AH_INT32: return "Int32"; case AH_INT64: return "Int64";
    case AH_UINT8: return "UInt8"; case AH_UINT16: return "UInt16";
    case AH_UINT32: return "UInt32"; case AH_UINT64: return "UInt64";
    case AH_FLOAT16: return "Float16"; case AH_FLOAT32: return "Float32";
    case AH_FLOAT64: return "Float64";
    case AH_FIXED16: return "FixedWidth16"; case AH_FIXED32: return "FixedWidth32";
    case AH_UTF8: return "Utf8"; case AH_LARGE_UTF8: return "LargeUtf8";
    case AH_UTF8_VIEW: return "Utf8View"; case AH_BINARY_VIEW: return "BinaryView";
    default: return "?";
  }
}
bool ah_type_is_integer(ah_type t) { return t >= AH_INT8 && t <= AH_UINT64; }
bool ah_type_is_signed(ah_type t) { return t >= AH_INT8 && t <= AH_INT64; }
bool ah_type_is_float(ah_type t) { return t == AH_FLOAT32 || t == AH_FLOAT64; }

void ah_out_init(ah_array_out* out) { memset(out, 0, sizeof *out); }

// ------------------------------------------------------------------- fault reporter
// GPUTEST_r05: "Memory access fault by GPU node-2 ... on address 0x77e74c552000. Reason: Unknown." twice on one box, nothing
// else — no test id, no stage, no way to tell whether the address was a buffer of this library, the byte behind one, a block
// already given back, or nothing of ours at all.  The table below is written with plain atomic stores (no lock: the handler
// runs on the HSA runtime's event thread while the faulting call may hold the context's mutex spinning on a mailbox) and is
// only ever READ when a fault has already doomed the process.
#include <dlfcn.h>
#include <hsa/hsa_ext_amd.h>
#include <atomic>
#include <unistd.h>
namespace {
struct FaultSlot {
  std::atomic<uintptr_t> ptr{0};
  std::atomic<size_t> bytes{0};
  std::atomic<int> kind{0}, state{0};
  std::atomic<const char*> what{nullptr};
  std::atomic<uint64_t> stamp{0};
};
constexpr size_t FAULT_SLOTS = 1 << 16;
FaultSlot g_fault_slots[FAULT_SLOTS];
std::atomic<uint64_t> g_fault_stamp{0};
std::atomic<ah_context*> g_fault_ctx[16];
std::atomic<bool> g_fault_registered{false};

size_t fault_hash(uintptr_t p) { return (size_t)((p >> 8) * 0x9E3779B97F4A7C15ull >> 48) & (FAULT_SLOTS - 1); }

void fault_line(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n > (int)sizeof buf - 1) n = (int)sizeof buf - 1;
  if (n > 0) (void)!write(2, buf, (size_t)n);
}
const char* kind_name(int k) { return k == AH_FK_POOL ? "pool block" : k == AH_FK_PINNED ? "pinned host block" : "context block"; }
const char* state_name(int s) { return s == AH_FS_LIVE ? "LIVE" : s == AH_FS_CACHED ? "released to the pool's free list" : "given back to the driver (unmapped)"; }

hsa_status_t fault_handler(const hsa_amd_event_t* ev, void*) {
  if (!ev || ev->event_type != HSA_AMD_GPU_MEMORY_FAULT_EVENT) return HSA_STATUS_ERROR;
  const uintptr_t va = (uintptr_t)ev->memory_fault.virtual_address;
  const uint32_t why = ev->memory_fault.fault_reason_mask;
  fault_line("\n[arrow_hip] GPU memory fault at %p, reason mask 0x%x%s%s%s (the runtime aborts the process after this report)\n", (void*)va, why,
             (why & HSA_AMD_MEMORY_FAULT_PAGE_NOT_PRESENT) ? " page-not-present" : "", (why & HSA_AMD_MEMORY_FAULT_READ_ONLY) ? " read-only" : "",
             (why & HSA_AMD_MEMORY_FAULT_HOST_ONLY) ? " host-only" : "");
  for (auto& c : g_fault_ctx) {
    ah_context* ctx = c.load(std::memory_order_relaxed);
    if (ctx)
      fault_line("[arrow_hip]   context %p (device %d): last entry point %s, call #%llu, mailbox sequence %llu\n", (void*)ctx, ctx->device,
                 ctx->last_entry ? ctx->last_entry : "(none)", (unsigned long long)ctx->entry_calls, (unsigned long long)ctx->mail_seq);
  }
  // the block that holds the address, else the nearest block on either side (the hardware reports faults per 4 KiB page: an
  // address on the first page behind a block's last byte is that block's overrun)
  const FaultSlot *in = nullptr, *below = nullptr, *above = nullptr;
  size_t noted = 0;
  for (const FaultSlot& s : g_fault_slots) {
    const uintptr_t p = s.ptr.load(std::memory_order_relaxed);
    if (!p) continue;
    ++noted;
    const uintptr_t e = p + s.bytes.load(std::memory_order_relaxed);
    if (va >= p && va < e) {
      if (!in || s.stamp.load() > in->stamp.load()) in = &s;
    } else if (e <= va) {
      if (!below || e > below->ptr.load() + below->bytes.load()) below = &s;
    } else if (!above || p < above->ptr.load()) {
      above = &s;
    }
  }
  auto show = [&](const char* rel, const FaultSlot* s) {
    if (!s) return;
    const uintptr_t p = s->ptr.load(), e = p + s->bytes.load();
    const long long d = va >= e ? (long long)(va - e) : (va < p ? (long long)(p - va) : (long long)(va - p));
    fault_line("[arrow_hip]   %s: %s %p + %zu bytes, %s, noted by %s (#%llu); the address is %lld bytes %s\n", rel, kind_name(s->kind.load()),
               (void*)p, s->bytes.load(), state_name(s->state.load()), s->what.load() ? s->what.load() : "?", (unsigned long long)s->stamp.load(), d,
               va >= e ? "past its end" : (va < p ? "before its start" : "into it"));
  };
  if (in) show("INSIDE", in);
  else fault_line("[arrow_hip]   the address is inside none of the %zu blocks this library has noted\n", noted);
  show("nearest block below", below);
  show("nearest block above", above);
  return HSA_STATUS_ERROR;  // not handled: the runtime prints its own line and aborts, exactly as without this handler
}

void fault_register(ah_context* ctx) {
  for (auto& c : g_fault_ctx) {
    ah_context* expect = nullptr;
    if (c.compare_exchange_strong(expect, ctx)) break;
  }
  bool was = false;
  if (!g_fault_registered.compare_exchange_strong(was, true)) return;
  const char* off = getenv("AH_FAULT_REPORT");
  if (off && off[0] == '0') return;
  using reg_fn = hsa_status_t (*)(hsa_amd_system_event_callback_t, void*);
  auto reg = (reg_fn)dlsym(RTLD_DEFAULT, "hsa_amd_register_system_event_handler");
  if (reg) (void)reg(fault_handler, nullptr);
}
void fault_unregister(ah_context* ctx) {
  for (auto& c : g_fault_ctx) {
    ah_context* expect = ctx;
    if (c.compare_exchange_strong(expect, nullptr)) break;
  }
}
}  // namespace

void ah_fault_enter(ah_context* c, const char* entry_point) {
  c->last_entry = entry_point;
  c->entry_calls += 1;
}

void ah_fault_note(const void* p, size_t bytes, int kind, int state, const char* what) {
  if (!p) return;
  const uintptr_t u = (uintptr_t)p;
  size_t h = fault_hash(u);
  FaultSlot* victim = nullptr;
  for (int probe = 0; probe < 64; ++probe, h = (h + 1) & (FAULT_SLOTS - 1)) {
    FaultSlot& s = g_fault_slots[h];
    const uintptr_t cur = s.ptr.load(std::memory_order_relaxed);
    if (cur == u || cur == 0) {
      victim = &s;
      break;
    }
    // a full neighbourhood: the oldest entry that is no longer live makes room
    if (s.state.load(std::memory_order_relaxed) != AH_FS_LIVE && (!victim || s.stamp.load() < victim->stamp.load())) victim = &s;
  }
  if (!victim) return;  // 64 live neighbours: this block goes unrecorded (the report says how many blocks it knows)
  victim->bytes.store(bytes, std::memory_order_relaxed);
  victim->kind.store(kind, std::memory_order_relaxed);
  victim->state.store(state, std::memory_order_relaxed);
  if (what) victim->what.store(what, std::memory_order_relaxed);
  victim->stamp.store(++g_fault_stamp, std::memory_order_relaxed);
  victim->ptr.store(u, std::memory_order_release);
}

// ------------------------------------------------------------------- pool
static size_t pool_round(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes <= (1u << 20)) {  // next power of two
    size_t r = 256;
    while (r < bytes) r <<= 1;
    return r;
  }
  return (bytes + ((1u << 20) - 1)) & ~(size_t)((1u << 20) - 1);
}

static inline void stats_on_alloc(ah_context* ctx, size_t r, bool from_cache) {
  ah_context_stats_t& s = ctx->stats;
  s.live_bytes += (int64_t)r;
  s.allocated_bytes_total += (int64_t)r;
  s.alloc_calls += 1;
  if (from_cache) {
    s.pool_hits += 1;
    s.cached_bytes -= (int64_t)r;
  } else {
    s.device_malloc_calls += 1;
  }
  if (s.live_bytes > s.high_water_bytes) s.high_water_bytes = s.live_bytes;
  if (s.live_bytes + s.cached_bytes > s.reserved_high_water_bytes) s.reserved_high_water_bytes = s.live_bytes + s.cached_bytes;
}

// Guard mode (AH_DEBUG_GUARD=1): the over-READ detector the redzone canaries are not (the reference's
// equivalent is Miri over its `unsafe` gathers, .github/workflows/miri.sh, arrow-select/src/filter.rs:756-761, take.rs:442-454).
// Every pool block — inputs uploaded through ah_device_alloc, outputs, scratch, slabs — is its own virtual-memory mapping
// (hipMemAddressReserve + hipMemCreate + hipMemMap) with the buffer's END flush against the end of the mapping and the
// next granule of the reservation left UNMAPPED: a kernel that touches anything at or past the first aligned unit behind
// the buffer raises "Memory access fault" on every box, instead of reading whatever the pool's power-of-two rounding or a
// neighbouring allocation happened to put there.  Nothing is cached or recycled: a released block is unmapped after the
// device has drained, so a kernel that uses a block after its release faults as well.
// AH_DEBUG_GUARD_ALIGN (default 16, a power of two >= 8) is the alignment of the buffer's start = the slack the end can
// have: the widest per-lane access is 16 bytes, and an ALIGNED 16-byte access never crosses a page, so up to 15 bytes of
// aligned overhang cannot fault on any mapping and are not an error; everything wider is.
static int guard_align_env() {
  const char* e = getenv("AH_DEBUG_GUARD_ALIGN");
  long a = e ? atol(e) : 16;
  if (a < 8 || (a & (a - 1))) a = 16;
  return (int)a;
}
bool ah_guard_mode() {
  static const bool on = [] {
    const char* e = getenv("AH_DEBUG_GUARD");
    return e && e[0] == '1';
  }();
  return on;
}
static ah_status guard_alloc(ah_context* ctx, size_t bytes, void** out) {
  static const size_t align = (size_t)guard_align_env();
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ctx->device;
  static size_t gran = 0;
  if (!gran) {
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !g)
      return ah_fail(ctx, AH_HIP_ERROR, "AH_DEBUG_GUARD=1: hipMemGetAllocationGranularity failed (no virtual memory management on this device?)");
    gran = g;
  }
  const size_t padded = (bytes + align - 1) & ~(align - 1);
  const size_t map_bytes = (padded + gran - 1) / gran * gran;
  ah_context::guard_block b{};
  b.va_bytes = map_bytes + gran;  // the last granule stays unmapped: the guard
  b.map_bytes = map_bytes;
  // Every reservation asks for the next NEVER-USED address (a process-wide cursor handed to hipMemAddressReserve as its
  // hint; honoured every time in tools/probes/vmm_probe.hip): see guard_unmap for why no address may come back.
  static std::mutex hint_mu;
  static char* hint = (char*)0x200000000000ull;
  hipError_t e = hipSuccess;
  {
    std::lock_guard<std::mutex> lk(hint_mu);
    for (int attempt = 0; attempt < 8; ++attempt) {
      e = hipMemAddressReserve(&b.va, b.va_bytes, gran, hint, 0);
      if (e != hipSuccess) break;
      if ((char*)b.va >= hint) {
        hint = (char*)b.va + b.va_bytes;
        break;
      }
      if (attempt == 7) break;  // (an address below the cursor eight times over: take it rather than fail the allocation)
      (void)hipMemAddressFree(b.va, b.va_bytes);
      hint += (size_t)1 << 30;
    }
  }
  if (e == hipSuccess) {
    e = hipMemCreate(&b.handle, map_bytes, &prop, 0);
    if (e == hipSuccess) {
      e = hipMemMap(b.va, map_bytes, 0, b.handle, 0);
      if (e == hipSuccess) {
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(b.va, map_bytes, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(b.va, map_bytes);
      }
      if (e != hipSuccess) (void)hipMemRelease(b.handle);
    }
    if (e != hipSuccess) (void)hipMemAddressFree(b.va, b.va_bytes);
  }
  if (e != hipSuccess)
    return ah_fail(ctx, e == hipErrorOutOfMemory ? AH_OUT_OF_MEMORY : AH_HIP_ERROR, "AH_DEBUG_GUARD=1: mapping %zu bytes failed: %s", map_bytes,
                   hipGetErrorString(e));
  void* p = (char*)b.va + map_bytes - padded;
  ah_fault_note(p, bytes, AH_FK_POOL, AH_FS_LIVE, ctx->last_entry);
  // fresh mappings read as zero (the driver clears VRAM), recycled pool blocks do not: poison the whole mapping so that a kernel
  // which relies on zeroed scratch or output memory sees garbage here on every box (AH_DEBUG_GUARD_FILL=hex byte, default CD)
  static const int fill = [] {
    const char* f = getenv("AH_DEBUG_GUARD_FILL");
    return f ? (int)strtol(f, nullptr, 16) & 0xFF : 0xCD;
  }();
  if (!ctx->capturing && hipMemsetAsync(b.va, fill, map_bytes, ctx->stream) != hipSuccess) (void)hipGetLastError();  // (a recorded memset would poison every replay)
  ctx->guard_live[p] = b;
  ctx->pool_live[p] = map_bytes;
  *out = p;
  stats_on_alloc(ctx, map_bytes, false);
  return AH_OK;
}
// No ADDRESS is ever used twice (guard_alloc's cursor): (1) a released buffer's addresses then stay unmapped for the life of
// the process, so a use-after-release faults however late it comes; (2) on this stack (ROCm 7.2 user space, amdgpu of Linux
// 6.18) a range that is hipMemAddressFree'd, reserved again at the same address and mapped to new memory intermittently serves
// the OLD translation to one of the engines — tools/probes/vmm_probe.hip shows blocks reading back another block's bytes after
// 15-120 map / unmap cycles with address reuse and none in 6 000 cycles without (profiles/r06_crash.md).  The reservation
// itself is given back (a kept one costs two host mappings: a long test process ran into vm.max_map_count); 47 bits of
// address space outlast any test run.
static void guard_unmap(const ah_context::guard_block& b) {
  (void)hipMemUnmap(b.va, b.map_bytes);
  (void)hipMemRelease(b.handle);
  (void)hipMemAddressFree(b.va, b.va_bytes);
}

ah_status ah_pool_alloc(ah_context* ctx, size_t bytes, void** out) {
  if (ah_guard_mode()) return guard_alloc(ctx, bytes ? bytes : 1, out);
  size_t r = pool_round(bytes);
  auto it = ctx->pool_free.find(r);
  if (it != ctx->pool_free.end() && !it->second.empty()) {
    *out = it->second.back();
    it->second.pop_back();
    ctx->pool_live[*out] = r;
    stats_on_alloc(ctx, r, true);
    ah_fault_note(*out, r, AH_FK_POOL, AH_FS_LIVE, ctx->last_entry);
    return AH_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, r);
  if (e != hipSuccess) {
    // drop the cache and retry once
    ah_pool_trim(ctx);
    e = hipMalloc(&p, r);
    if (e != hipSuccess)
      return ah_fail(ctx, AH_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", r, hipGetErrorString(e));
  }
  ctx->pool_live[p] = r;
  *out = p;
  stats_on_alloc(ctx, r, false);
  ah_fault_note(p, r, AH_FK_POOL, AH_FS_LIVE, ctx->last_entry);
  return AH_OK;
}

void ah_pool_free(ah_context* ctx, void* p) {
  if (!p) return;
  auto it = ctx->pool_live.find(p);
  if (it == ctx->pool_live.end()) return;  // not ours
  if (ctx->capturing) {  // a captured kernel writes this block on every replay: it stays out of circulation with the graph
    ctx->capture_hold.push_back(p);
    return;
  }
  auto gb = ctx->guard_live.find(p);
  if (gb != ctx->guard_live.end()) {  // guard mode: never recycled — unmapped once nothing enqueued can still be using it
    (void)hipDeviceSynchronize();
    // Unmapping does not reliably invalidate the translations the GPU has cached (tools/fault_demo.py use_after_release reads a
    // released block without a fault on this stack): the block is overwritten with 0xDD first, so that a late reader at least
    // gets bytes no parity check accepts.
    if (hipMemsetAsync(gb->second.va, 0xDD, gb->second.map_bytes, ctx->stream) == hipSuccess) (void)hipStreamSynchronize(ctx->stream);
    else (void)hipGetLastError();
    guard_unmap(gb->second);
    ah_fault_note(p, 0, AH_FK_POOL, AH_FS_RELEASED, nullptr);
    ctx->guard_live.erase(gb);
    ctx->stats.live_bytes -= (int64_t)it->second;
    ctx->stats.freed_bytes_total += (int64_t)it->second;
    ctx->stats.free_calls += 1;
    ctx->pool_live.erase(it);
    return;
  }
  ctx->pool_free[it->second].push_back(p);
  ah_fault_note(p, it->second, AH_FK_POOL, AH_FS_CACHED, nullptr);
  ctx->stats.live_bytes -= (int64_t)it->second;
  ctx->stats.freed_bytes_total += (int64_t)it->second;
  ctx->stats.cached_bytes += (int64_t)it->second;
  ctx->stats.free_calls += 1;
  ctx->pool_live.erase(it);
}

extern "C" void ah_pool_trim(ah_context* ctx) {
  ah_ctx_guard _guard(ctx);
  if (ctx->capturing) return;  // (cached blocks stay cached; a sync would invalidate the capture)
  hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->pool_free)
    for (void* p : kv.second) {
      hipFree(p);
      ah_fault_note(p, kv.first, AH_FK_POOL, AH_FS_RELEASED, nullptr);
    }
  ctx->pool_free.clear();
  ctx->stats.cached_bytes = 0;
}

// MemoryPool::used() and friends for this context (arrow-buffer/src/pool.rs:73-93)
extern "C" ah_status ah_context_stats(ah_context* ctx, ah_context_stats_t* out, int32_t reset_peaks) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out) return AH_INVALID_ARGUMENT;
  *out = ctx->stats;
  if (reset_peaks) {
    ctx->stats.high_water_bytes = ctx->stats.live_bytes;
    ctx->stats.reserved_high_water_bytes = ctx->stats.live_bytes + ctx->stats.cached_bytes;
  }
  return AH_OK;
}

// ------------------------------------------------------------------- pinned blocks
ah_status ah_pinned_alloc(ah_context* ctx, size_t bytes, void** host, void** dev) {
  const size_t r = (std::max<size_t>(bytes, 8) + 4095) & ~(size_t)4095;
  void* h = nullptr;
  for (size_t i = 0; i < ctx->pinned_cache.size(); ++i)
    if (ctx->pinned_cache[i].first == r) {
      h = ctx->pinned_cache[i].second;
      ctx->pinned_cache.erase(ctx->pinned_cache.begin() + (long)i);
      break;
    }
  if (!h && hipHostMalloc(&h, r, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
    return ah_fail(ctx, AH_OUT_OF_MEMORY, "pinned host block of %zu bytes", r);
  memset(h, 0, r);
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) d = h;  // unified addressing
  *host = h;
  *dev = d;
  ah_fault_note(d, r, AH_FK_PINNED, AH_FS_LIVE, ctx->last_entry);
  return AH_OK;
}
void ah_pinned_free(ah_context* ctx, void* host, size_t bytes) {
  if (!host) return;
  const size_t r = (std::max<size_t>(bytes, 8) + 4095) & ~(size_t)4095;
  if (ctx->pinned_cache.size() < 32) {
    ctx->pinned_cache.emplace_back(r, host);
    ah_fault_note(host, r, AH_FK_PINNED, AH_FS_CACHED, nullptr);
  } else {
    hipHostFree(host);
    ah_fault_note(host, r, AH_FK_PINNED, AH_FS_RELEASED, nullptr);
  }
}

// Debug redzones (AH_DEBUG_REDZONE=1): every pooled output buffer gets a 256-byte canary right
// after its requested size (the pool rounds sizes up, which would otherwise hide small kernel
// overruns); the canary is verified when the buffer is released.  The GPU-side analogue of the
// reference's Miri / force_validate runs.
static bool redzone_on() {
  static const char* e = getenv("AH_DEBUG_REDZONE");
  return e && e[0] == '1';
}
static constexpr size_t RZ = 256;

ah_status ah_out_alloc(ah_context* ctx, size_t bytes, void** out) {
  if (bytes == 0) bytes = 8;
  if (!ctx->alloc && redzone_on()) {
    AH_TRY(ah_pool_alloc(ctx, bytes + RZ, out));
    ctx->redzones[*out] = bytes;
    AH_HIP(ctx, hipMemsetAsync((char*)*out + bytes, 0xA5, RZ, ctx->stream));
    return AH_OK;
  }
  if (ctx->alloc) {
    void* p = ctx->alloc(ctx->user, bytes);
    if (!p) return ah_fail(ctx, AH_OUT_OF_MEMORY, "host allocator returned NULL for %zu bytes", bytes);
    // remembered explicitly: the host may carve its memory out of a block it got from ah_device_alloc, so "not a
    // pool pointer" is not a safe test for "came from the hook"
    ctx->hook_live[p] = ah_context::hook_entry{ctx->free_, ctx->user};
    *out = p;
    return AH_OK;
  }
  return ah_pool_alloc(ctx, bytes, out);
}

ah_status ah_slab_create(ah_context* ctx, size_t bytes, ah_slab** out) {
  *out = nullptr;
  void* b = nullptr;
  if (!bytes) bytes = 8;
  const bool rz = redzone_on();  // AH_DEBUG_REDZONE: a canary behind the slab as behind every other output buffer
  AH_TRY(ah_pool_alloc(ctx, bytes + (rz ? RZ : 0), &b));
  if (rz) AH_HIP(ctx, hipMemsetAsync((char*)b + bytes, 0xA5, RZ, ctx->stream));
  auto* s = new ah_slab();
  s->block = b, s->bytes = bytes, s->refs = 1;
  ctx->slabs[(uintptr_t)b] = s;
  *out = s;
  return AH_OK;
}
void ah_slab_unref(ah_context* ctx, ah_slab* s) {
  if (!s || --s->refs > 0) return;
  if (redzone_on()) {
    unsigned char tail[RZ];
    hipStreamSynchronize(ctx->stream);
    if (hipMemcpy(tail, (char*)s->block + s->bytes, RZ, hipMemcpyDeviceToHost) == hipSuccess)
      for (size_t i = 0; i < RZ; ++i)
        if (tail[i] != 0xA5) {
          fprintf(stderr, "arrow_hip: REDZONE CORRUPTED: slab %p of %zu bytes overrun at +%zu\n", s->block, s->bytes, i);
          abort();
        }
  }
  ctx->slabs.erase((uintptr_t)s->block);
  ah_pool_free(ctx, s->block);
  delete s;
}
void ah_slab_slice(ah_context* ctx, ah_slab* s, void* ptr) {
  (void)ctx, (void)ptr;
  s->refs += 1;
}

void ah_out_free(ah_context* ctx, void* p, size_t bytes) {
  if (!p) return;
  if (!ctx->slabs.empty()) {  // a slice of a slab (any pointer inside a live slab's block): the block outlives it while others are out
    auto it = ctx->slabs.upper_bound((uintptr_t)p);
    if (it != ctx->slabs.begin()) {
      --it;
      if ((uintptr_t)p < it->first + it->second->bytes) {
        ah_slab_unref(ctx, it->second);
        return;
      }
    }
  }
  auto rz = ctx->redzones.find(p);
  if (rz != ctx->redzones.end()) {
    unsigned char tail[RZ];
    hipStreamSynchronize(ctx->stream);
    if (hipMemcpy(tail, (char*)p + rz->second, RZ, hipMemcpyDeviceToHost) == hipSuccess) {
      for (size_t i = 0; i < RZ; ++i)
        if (tail[i] != 0xA5) {
          fprintf(stderr, "arrow_hip: REDZONE CORRUPTED: buffer %p of %zu bytes overrun at +%zu\n", p, rz->second, i);
          abort();
        }
    }
    ctx->redzones.erase(rz);
  }
  auto hk = ctx->hook_live.find(p);
  if (hk != ctx->hook_live.end()) {
    const ah_context::hook_entry e = hk->second;  // freed by the allocator that made it, even if the hook changed since
    ctx->hook_live.erase(hk);
    // the host's free is not stream-ordered: in deferred mode, or after a no-wait call (ctx->inflight), kernels that
    // still use the buffer may be in flight, so drain the stream before handing the memory back
    if (ctx->deferred || ctx->inflight) (void)ah_stream_wait(ctx);
    if (e.free_) e.free_(e.user, p, bytes);
    return;
  }
  ah_pool_free(ctx, p);
}

// ---------------------------------------------------------------- mailbox
// One wave: copy `nwords` device words into the pinned slots, optionally restore the device words to
// `reset_value`, then (seq != 0) publish the sequence word.  nwords == 0 is the pure "stream reached here" flag.
__global__ void __launch_bounds__(64) mail_kernel(const unsigned long long* src, int nwords, uint64_t* dst,
                                                  uint64_t* mail, uint64_t seq, int reset,
                                                  unsigned long long reset_value) {
  for (int i = threadIdx.x; i < nwords; i += 64) {
    unsigned long long v = src[i];
    __hip_atomic_store(dst + i, (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (reset) ((unsigned long long*)src)[i] = reset_value;
  }
  if (seq) {
    __threadfence_system();  // every lane's payload stores, before lane 0's sequence store
    if (threadIdx.x == 0) __hip_atomic_store(mail + AH_MAIL_FLAG, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

// A DEFERRED ah_take that met an out-of-bounds index has left its fault on the device; until the host calls ah_synchronize /
// ah_array_resolve (which raise it with the reference's panic text) its output — and whatever was computed from it — is garbage.
// Every internal host wait therefore peeks at the fault word while a deferred take is outstanding (`fault_armed`): a
// synchronous call that would otherwise hand such garbage on fails instead (AH_HIP turns hipErrorAssert into AH_PANIC with a
// message that points at ah_synchronize).  The fault stays on the device and `fault_armed` stays set: ah_synchronize still
// reports it in full.  (waits used to consume a faulted take's output silently.)
static hipError_t fault_peek(ah_context* ctx, hipError_t e) {
  if (e != hipSuccess || !ctx->fault_armed || !ctx->fault_dev || ctx->capturing) return e;
  unsigned long long w0 = ~0ull;
  if (hipMemcpy(&w0, ctx->fault_dev, sizeof w0, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return e;
  }
  return w0 == ~0ull ? e : hipErrorAssert;
}

static hipError_t mail_wait_impl(ah_context* ctx, uint64_t seq);
hipError_t ah_mail_wait(ah_context* ctx, uint64_t seq) { return fault_peek(ctx, mail_wait_impl(ctx, seq)); }

static hipError_t mail_wait_impl(ah_context* ctx, uint64_t seq) {
  if (ctx->capturing) return hipErrorStreamCaptureUnsupported;  // the posting kernel is being recorded, not run: never spin
  volatile uint64_t* flag = ctx->pinned + AH_MAIL_FLAG;
  if (ctx->wait_mode == 1) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) e = hipErrorUnknown;
    if (e == hipSuccess) ctx->inflight = false;
    return e;
  }
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t spins = 1;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= seq) {
      ctx->inflight = false;  // the stream has passed everything enqueued before the posting kernel
      return hipSuccess;
    }
    cpu_relax();
    if ((spins & 4095) == 0) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
      if (us > 20000.0) {  // a long kernel, or a fault that will never post: ask the runtime
        hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) {
          ctx->inflight = false;
          return __atomic_load_n(flag, __ATOMIC_ACQUIRE) >= seq ? hipSuccess : hipErrorUnknown;
        }
        if (q != hipErrorNotReady) return q;
        sched_yield();  // a kernel of tens of milliseconds: let threads queued on this context's host side run
        if (us > 500000.0) {  // half a second of spinning: sleep in the runtime instead
          hipError_t e = hipStreamSynchronize(ctx->stream);
          if (e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) e = hipErrorUnknown;
          return e;
        }
      }
    }
  }
}

static inline bool in_pinned(const ah_context* ctx, const void* p, size_t bytes) {
  const char* b = (const char*)ctx->pinned;
  return (const char*)p >= b && (const char*)p + bytes <= b + AH_MAIL_FLAG * 8;
}

hipError_t ah_d2h(ah_context* ctx, void* pinned_dst, const void* dev_src, size_t bytes) {
  if (!in_pinned(ctx, pinned_dst, bytes) || (bytes & 7) || ((uintptr_t)dev_src & 7))
    return hipMemcpyAsync(pinned_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  uint64_t* d = ctx->pinned_dev + ((uint64_t*)pinned_dst - ctx->pinned);
  mail_kernel<<<1, 64, 0, ctx->stream>>>((const unsigned long long*)dev_src, (int)(bytes >> 3), d, ctx->pinned_dev, 0, 0, 0);
  return hipGetLastError();
}

// the first half of ah_stream_wait: post the next mailbox sequence number behind everything enqueued so far and return
// it; ah_mail_wait(ctx, seq) is the second half.  A caller that keeps enqueueing in between overlaps its own wait.
hipError_t ah_mail_post_async(ah_context* ctx, uint64_t* seq_out) {
  const uint64_t seq = ah_mail_next(ctx);
  mail_kernel<<<1, 64, 0, ctx->stream>>>(nullptr, 0, ctx->pinned_dev, ctx->pinned_dev, seq, 0, 0);
  *seq_out = seq;
  return hipGetLastError();
}

hipError_t ah_stream_wait(ah_context* ctx) {
  if (ctx->capturing) return hipErrorStreamCaptureUnsupported;
  if (ctx->wait_mode == 1) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) ctx->inflight = false;
    return fault_peek(ctx, e);
  }
  const uint64_t seq = ah_mail_next(ctx);
  mail_kernel<<<1, 64, 0, ctx->stream>>>(nullptr, 0, ctx->pinned_dev, ctx->pinned_dev, seq, 0, 0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ah_mail_wait(ctx, seq) : e;
}

// The 64 counters are "zero between calls": only a successful read-back restores that.  When the read (or the call
// around an already enqueued counting kernel) fails, drain the stream and zero them here — otherwise every later
// counted op of the context would report the leftovers as null counts.
void ah_count_reset(ah_context* ctx) {
  if (ctx->capturing) return;  // recorded kernels never ran: the counters are clean, and a sync would invalidate the capture
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipGetLastError();
  if (hipMemsetAsync(ctx->scratch + AH_TICKET_COUNT, 0, 64 * 8, ctx->stream) == hipSuccess) (void)hipStreamSynchronize(ctx->stream);
}

hipError_t ah_count_read(ah_context* ctx, int64_t* total) {
  hipError_t e = ah_d2h_wait(ctx, ctx->pinned + 64, ctx->scratch + AH_TICKET_COUNT, 64 * 8, /*reset=*/true);
  if (e != hipSuccess) {
    ah_count_reset(ctx);
    return e;
  }
  uint64_t t = 0;
  for (int i = 0; i < 64; ++i) t += ctx->pinned[64 + i];
  *total = (int64_t)t;
  return hipSuccess;
}

hipError_t ah_d2h_wait(ah_context* ctx, void* pinned_dst, const void* dev_src, size_t bytes, bool reset,
                       uint64_t reset_value) {
  if (ctx->capturing) return hipErrorStreamCaptureUnsupported;  // (before ANY HIP call: a sync would invalidate the capture)
  if (!in_pinned(ctx, pinned_dst, bytes) || (bytes & 7) || ((uintptr_t)dev_src & 7)) {
    hipError_t e = hipMemcpyAsync(pinned_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && reset) e = hipMemsetAsync((void*)dev_src, (int)(reset_value & 0xFF), bytes, ctx->stream);
    return e == hipSuccess ? fault_peek(ctx, hipStreamSynchronize(ctx->stream)) : e;
  }
  const uint64_t seq = ah_mail_next(ctx);
  uint64_t* d = ctx->pinned_dev + ((uint64_t*)pinned_dst - ctx->pinned);
  mail_kernel<<<1, 64, 0, ctx->stream>>>((const unsigned long long*)dev_src, (int)(bytes >> 3), d, ctx->pinned_dev, seq,
                                         reset ? 1 : 0, reset_value);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? ah_mail_wait(ctx, seq) : e;
}

// ---------------------------------------------------------------- context
extern "C" int32_t ah_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

extern "C" ah_status ah_context_create(int device, ah_context** out) {
  if (!out) return AH_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return AH_HIP_ERROR;  // fail loudly: no CPU fallback
  if (device < 0 || device >= n) return AH_INVALID_ARGUMENT;
  if (hipSetDevice(device) != hipSuccess) return AH_HIP_ERROR;
  ah_context* c = new ah_context();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return AH_HIP_ERROR;
  }
  c->stream = c->own_stream;
  // mapped + coherent: kernels store results straight into these slots and the host polls them
  if (hipHostMalloc((void**)&c->pinned, 256 * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    hipStreamDestroy(c->own_stream);
    delete c;
    return AH_HIP_ERROR;
  }
  memset(c->pinned, 0, 256 * sizeof(uint64_t));
  void* dp = nullptr;
  if (hipHostGetDevicePointer(&dp, c->pinned, 0) != hipSuccess || !dp) dp = c->pinned;  // unified addressing
  c->pinned_dev = (uint64_t*)dp;
  const char* wm = getenv("AH_WAIT");
  c->wait_mode = (wm && strcmp(wm, "block") == 0) ? 1 : 0;
  uint64_t init[AH_SCRATCH_WORDS];
  for (int i = 0; i < AH_SCRATCH_WORDS; ++i) init[i] = (i >= AH_SCRATCH_ONES && i < AH_SCRATCH_TICKETS) ? ~0ull : 0ull;
  if (hipMalloc((void**)&c->scratch, sizeof init) != hipSuccess ||
      hipMemcpy(c->scratch, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) {
    if (c->scratch) hipFree(c->scratch);
    hipHostFree(c->pinned);
    hipStreamDestroy(c->own_stream);
    delete c;
    return AH_HIP_ERROR;
  }
  const unsigned long long fault_init[4] = {~0ull, 0, 0, 0};
  if (hipMalloc((void**)&c->fault_dev, sizeof fault_init) != hipSuccess ||
      hipMemcpy(c->fault_dev, fault_init, sizeof fault_init, hipMemcpyHostToDevice) != hipSuccess) {
    if (c->fault_dev) hipFree(c->fault_dev);
    hipFree(c->scratch);
    hipHostFree(c->pinned);
    hipStreamDestroy(c->own_stream);
    delete c;
    return AH_HIP_ERROR;
  }
  fault_register(c);
  ah_fault_note(c->pinned_dev, 256 * sizeof(uint64_t), AH_FK_PINNED, AH_FS_LIVE, "ah_context_create (mailbox)");
  ah_fault_note(c->scratch, sizeof init, AH_FK_CONTEXT, AH_FS_LIVE, "ah_context_create (counter scratch)");
  ah_fault_note(c->fault_dev, sizeof fault_init, AH_FK_CONTEXT, AH_FS_LIVE, "ah_context_create (deferred fault words)");
  *out = c;
  return AH_OK;
}

extern "C" void ah_context_destroy(ah_context* ctx) {
  if (!ctx) return;
  fault_unregister(ctx);
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
  ah_profile_reset(ctx);
  ah_pool_trim(ctx);
  for (auto& kv : ctx->pool_live) {
    auto gb = ctx->guard_live.find(kv.first);
    if (gb != ctx->guard_live.end()) guard_unmap(gb->second);
    else hipFree(kv.first);
  }
  for (hipEvent_t e : ctx->event_pool) hipEventDestroy(e);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->fault_dev) hipFree(ctx->fault_dev);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  for (auto& pb : ctx->pinned_cache) hipHostFree(pb.second);
  if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
  if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

extern "C" void ah_context_set_allocator(ah_context* ctx, ah_alloc_fn a, ah_free_fn f, void* user) {
  ah_ctx_guard _guard(ctx);
  ctx->alloc = a;
  ctx->free_ = f;
  ctx->user = user;
}
extern "C" void ah_context_set_stream(ah_context* ctx, void* s) {
  ah_ctx_guard _guard(ctx);
  hipStream_t next = s ? (hipStream_t)s : ctx->own_stream;
  // pooled scratch and released outputs are reused in STREAM order; when the stream really changes, work still
  // running on the old one could otherwise overlap a block's next owner: drain the old stream first
  if (next != ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  ctx->stream = next;
}
extern "C" void* ah_context_stream(ah_context* ctx) {
  ah_ctx_guard _guard(ctx); return (void*)ctx->stream; }
extern "C" void ah_context_set_deferred(ah_context* ctx, int32_t on) {
  ah_ctx_guard _guard(ctx);
  if (!ctx) return;
  if (ctx->capturing) {  // the capture owns the mode until ah_graph_end; remember what to restore
    ctx->capture_was_deferred = on != 0;
    return;
  }
  if (ctx->deferred && !on) hipStreamSynchronize(ctx->stream);  // leaving deferred mode: everything enqueued is done
  ctx->deferred = on != 0;
}
extern "C" int32_t ah_context_deferred(const ah_context* ctx) { return ctx && ctx->deferred; }
extern "C" const char* ah_last_error(ah_context* ctx) {
  if (!ctx) return "no context";
  if (tls_error.ctx == ctx) return tls_error.msg.c_str();  // this thread's own last failure on this context
  ah_ctx_guard _guard(ctx);
  return ctx->err.c_str();
}
extern "C" const char* ah_version(void) { return "arrow_hip 0.1.0 (gfx950)"; }

extern "C" void ah_array_release(ah_context* ctx, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!out) return;
  if (!(out->flags & AH_OUT_BORROWED)) {
    if (!(out->flags & AH_OUT_BORROWED_VALUES)) {
      ah_out_free(ctx, out->values, (size_t)out->values_bytes);
      ah_out_free(ctx, out->offsets, (size_t)out->offsets_bytes);
    }
    ah_out_free(ctx, out->validity, (size_t)out->validity_bytes);
  }
  ah_out_init(out);
}

extern "C" ah_status ah_device_alloc(ah_context* ctx, size_t bytes, void** out) {
  ah_ctx_guard _guard(ctx);
  hipSetDevice(ctx->device);
  return ah_pool_alloc(ctx, bytes ? bytes : 8, out);
}
extern "C" void ah_device_free(ah_context* ctx, void* p) {
  ah_ctx_guard _guard(ctx); ah_pool_free(ctx, p); }
extern "C" ah_status ah_memcpy_htod(ah_context* ctx, void* dst, const void* src, size_t bytes) {
  ah_ctx_guard _guard(ctx);
  if (!bytes) return AH_OK;
  if (ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "ah_memcpy_htod while a graph is being recorded: it waits for the stream");
  AH_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stats.host_to_device_bytes += (int64_t)bytes;
  return AH_OK;
}
extern "C" ah_status ah_memcpy_dtod(ah_context* ctx, void* dst, const void* src, size_t bytes) {
  ah_ctx_guard _guard(ctx);
  if (!bytes) return AH_OK;
  if (ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "ah_memcpy_dtod while a graph is being recorded: it waits for the stream");
  AH_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AH_OK;
}
extern "C" ah_status ah_memcpy_dtoh(ah_context* ctx, void* dst, const void* src, size_t bytes) {
  ah_ctx_guard _guard(ctx);
  if (!bytes) return AH_OK;
  if (ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "ah_memcpy_dtoh while a graph is being recorded: it waits for the stream");
  AH_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stats.device_to_host_bytes += (int64_t)bytes;
  return AH_OK;
}
extern "C" ah_status ah_memset(ah_context* ctx, void* dst, int value, size_t bytes) {
  ah_ctx_guard _guard(ctx);
  if (!bytes) return AH_OK;
  AH_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
  return AH_OK;
}
// ------------------------------------------------------------------- hipGraph capture of deferred calls
// The launch-bound regime (batches of 10^3 .. 10^5 rows: a call is 2-4 us of kernels behind 3-4 us of host launch cost
// each) as ONE graph launch: between ah_graph_begin and ah_graph_end the deferred entry points (wrapping / float
// arithmetic, cmp, boolean, safe numeric casts, ah_filter_predicate_apply with a prebuilt predicate) are RECORDED on the
// context's stream instead of run; ah_graph_launch replays the whole sequence over whatever bytes the captured input
// pointers hold by then, into the SAME output buffers the captured calls returned.
struct ah_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<void*> held;  // pooled blocks the captured calls released (scratch): parked until the graph dies
  int nodes = 0;
  bool arms_fault = false;  // a recorded call (ah_take) can leave a fault on the device: every replay re-arms the host's check
};

extern "C" ah_status ah_graph_begin(ah_context* ctx) {
  ah_ctx_guard _guard(ctx);
  if (!ctx) return AH_INVALID_ARGUMENT;
  if (ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "a graph capture is already open on this context");
  hipSetDevice(ctx->device);
  AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // relaxed: a pool miss may still hipMalloc while recording (it is not part of the graph; the block is then simply owned)
  AH_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capture_was_deferred = ctx->deferred;
  ctx->deferred = true;  // only enqueue-only entry points can be recorded; everything else fails fast (no wait is possible)
  ctx->capturing = true;
  ctx->capture_fault_armed_before = ctx->fault_armed;
  ctx->fault_armed = false;
  return AH_OK;
}

extern "C" ah_status ah_graph_end(ah_context* ctx, ah_graph** out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out) return AH_INVALID_ARGUMENT;
  *out = nullptr;
  if (!ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "no graph capture is open on this context");
  hipSetDevice(ctx->device);
  auto* g = new ah_graph();
  hipError_t e = hipStreamEndCapture(ctx->stream, &g->graph);
  ctx->capturing = false;
  g->arms_fault = ctx->fault_armed;  // (nothing ran while recording: the capture itself leaves no fault)
  ctx->fault_armed = ctx->capture_fault_armed_before;
  ctx->deferred = ctx->capture_was_deferred;
  g->held.swap(ctx->capture_hold);
  for (int k = 0; k < 8 && hipGetLastError() != hipSuccess; ++k) {}  // errors raised while recording are sticky per thread
  if (e == hipSuccess && !g->graph) e = hipErrorStreamCaptureInvalidated;
  if (e != hipSuccess) {
    // An INVALIDATED capture (some HIP call that is illegal while recording slipped through) can leave the stream
    // refusing every later launch ("operation failed due to a previous error during capture", seen on ROCm 7.2 even
    // after hipStreamEndCapture returned).  The context's own stream is replaced; a caller-provided stream is the caller's.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool stuck = hipStreamIsCapturing(ctx->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone ||
                       (hipStreamQuery(ctx->stream) != hipSuccess && hipStreamQuery(ctx->stream) != hipErrorNotReady);
    (void)hipGetLastError();
    if (stuck && ctx->stream == ctx->own_stream) {
      hipStream_t fresh = nullptr;
      if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) {
        (void)hipStreamDestroy(ctx->own_stream);
        ctx->own_stream = ctx->stream = fresh;
      }
      (void)hipGetLastError();
    }
  }
  if (e == hipSuccess) {
    size_t n = 0;
    if (hipGraphGetNodes(g->graph, nullptr, &n) == hipSuccess) g->nodes = (int)n;
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (g->graph) hipGraphDestroy(g->graph);
    for (void* p : g->held) ah_pool_free(ctx, p);
    delete g;
    return ah_fail(ctx, AH_HIP_ERROR, "graph capture failed: %s (an entry point that must wait on the device was called while recording?)",
                   hipGetErrorString(e));
  }
  *out = g;
  return AH_OK;
}

extern "C" int32_t ah_graph_node_count(const ah_graph* g) { return g ? g->nodes : 0; }

// one replay, enqueued on the context's stream (no wait: ah_synchronize / the next synchronous call orders behind it)
extern "C" ah_status ah_graph_launch(ah_context* ctx, ah_graph* g) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !g || !g->exec) return AH_INVALID_ARGUMENT;
  if (ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "cannot launch a graph while recording one");
  hipSetDevice(ctx->device);
  ctx->inflight = true;
  if (g->arms_fault) ctx->fault_armed = true;
  AH_HIP(ctx, hipGraphLaunch(g->exec, ctx->stream));
  return AH_OK;
}

extern "C" void ah_graph_destroy(ah_context* ctx, ah_graph* g) {
  ah_ctx_guard _guard(ctx);
  if (!g) return;
  if (ctx) {
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);  // a replay may still be writing the held blocks
  }
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  if (ctx)
    for (void* p : g->held) ah_pool_free(ctx, p);
  delete g;
}

// The fault slot of deferred calls (common.hpp): read and re-armed after a stream wait.
ah_status ah_check_deferred_fault(ah_context* ctx) {
  if (!ctx->fault_armed || !ctx->fault_dev) return AH_OK;
  ctx->fault_armed = false;
  unsigned long long w[4];
  AH_HIP(ctx, hipMemcpy(w, ctx->fault_dev, sizeof w, hipMemcpyDeviceToHost));
  if (w[0] == ~0ull) return AH_OK;
  const unsigned long long rearm[4] = {~0ull, 0, 0, 0};
  AH_HIP(ctx, hipMemcpy(ctx->fault_dev, rearm, sizeof rearm, hipMemcpyHostToDevice));
  // ah_take's three panics (take.rs:447, :454; arrow-buffer/src/buffer/boolean.rs:495), as the synchronous call words them
  if (w[3] == 0) return ah_fail(ctx, AH_PANIC, "assertion failed: idx < self.bit_len");
  if (w[3] == 1) return ah_fail(ctx, AH_PANIC, "Out-of-bounds index %llu", w[1]);
  return ah_fail(ctx, AH_PANIC, "index out of bounds: the len is %lld but the index is %llu", (long long)w[2], w[1]);
}

extern "C" ah_status ah_synchronize(ah_context* ctx) {
  ah_ctx_guard _guard(ctx);
  if (ctx->capturing) return ah_fail(ctx, AH_INVALID_ARGUMENT, "ah_synchronize while a graph is being recorded: end the capture first");
  AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->inflight = false;
  return ah_check_deferred_fault(ctx);
}

extern "C" ah_status ah_array_resolve(ah_context* ctx, ah_array_out* out) {
  ah_ctx_guard _guard(ctx);
  if (!ctx || !out) return AH_INVALID_ARGUMENT;
  hipSetDevice(ctx->device);
  AH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  AH_TRY(ah_check_deferred_fault(ctx));
  if (out->null_count >= 0) return AH_OK;
  if (!out->validity) {
    out->null_count = 0;
    return AH_OK;
  }
  int64_t set = 0;
  AH_TRY(ah_count_set_bits(ctx, out->validity, out->validity_bit_offset, out->length, &set));
  out->null_count = out->length - set;
  return AH_OK;
}

// -------------------------------------------------------------- profiling
extern "C" void ah_profile_enable(ah_context* ctx, int32_t on) {
  ah_ctx_guard _guard(ctx); ctx->profiling = on != 0; }
static void prof_drain(ah_context* ctx, ah_prof_entry& e) {
  for (auto& pr : e.pending) {
    hipEventSynchronize(pr.second);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      e.total_ms += ms;
      e.launches += 1;
    }
    ctx->event_pool.push_back(pr.first);
    ctx->event_pool.push_back(pr.second);
  }
  e.pending.clear();
}
extern "C" void ah_profile_reset(ah_context* ctx) {
  ah_ctx_guard _guard(ctx);
  for (auto& kv : ctx->prof) prof_drain(ctx, kv.second);
  ctx->prof.clear();
}
extern "C" ah_status ah_profile_get(ah_context* ctx, const char* kernel, double* total_ms,
                                    int64_t* launches) {
  ah_ctx_guard _guard(ctx);
  auto it = ctx->prof.find(kernel);
  if (it == ctx->prof.end()) {
    if (total_ms) *total_ms = 0;
    if (launches) *launches = 0;
    return AH_OK;
  }
  prof_drain(ctx, it->second);
  if (total_ms) *total_ms = it->second.total_ms;
  if (launches) *launches = it->second.launches;
  return AH_OK;
}

// ----------------------------------------------------------- popcount kernel
// One u64 partial per block, then a single-block finish: no same-address atomics.
__global__ void __launch_bounds__(256) popcount_partial_kernel(BitView bits, int64_t len, unsigned long long* total) {
  int64_t nwords = (len + 63) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
       w += (int64_t)gridDim.x * blockDim.x)
    acc += __popcll(bv_fetch64(bits, w << 6, len));
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ah_count_add(total, s[0] + s[1] + s[2] + s[3]);
}

__global__ void __launch_bounds__(1024) sum_u64_kernel(const unsigned long long* in, int64_t n,
                                                       unsigned long long* out) {
  unsigned long long acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += in[i];
  acc = wave_reduce_add64(acc);
  __shared__ unsigned long long s[16];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; i++) t += s[i];
    *out = t;
  }
}

extern "C" ah_status ah_count_set_bits(ah_context* ctx, const uint8_t* bits, int64_t bit_offset,
                                       int64_t len, int64_t* count) {
  ah_ctx_guard _guard(ctx);
  if (len <= 0 || !bits) {
    *count = bits ? 0 : (len > 0 ? len : 0);
    return AH_OK;
  }
  hipSetDevice(ctx->device);
  int64_t nwords = (len + 63) >> 6;
  int grid = (int)std::min<int64_t>(2048, ah_ceil_div(nwords, 256));
  BitView bv = make_bitview(bits, bit_offset);
  popcount_partial_kernel<<<grid, 256, 0, ctx->stream>>>(bv, len, ctx->scratch + AH_TICKET_COUNT);
  AH_HIP(ctx, ah_count_read(ctx, count));
  return AH_OK;
}

ah_status ah_resolve_null_count(ah_context* ctx, const ah_array_view* v, int64_t* nulls) {
  if (!v->validity) {
    *nulls = 0;
    return AH_OK;
  }
  if (v->null_count >= 0) {
    *nulls = v->null_count;
    return AH_OK;
  }
  int64_t set = 0;
  AH_TRY(ah_count_set_bits(ctx, v->validity, v->validity_bit_offset, v->length, &set));
  *nulls = v->length - set;
  return AH_OK;
}
